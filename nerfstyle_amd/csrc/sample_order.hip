// Spatial processing order of the marched samples, gfx950.
//
// The march emits samples ray by ray.  A full-frame batch (762 048 rays, 48 M samples) visits every occupied piece of
// space with hundreds of rays, but in ray order two samples that share a hash-table row are millions of samples apart:
// the forward re-fetches the row from HBM, the backward sends one more atomic to the memory side (the 6.3 M-row
// gradient table receives ~1.4 G records per launch).  nsr_sample_order computes a PERMUTATION -- sample indices in
// Morton order of the encoder input, 10 bits per axis (one key = a 4^3 block of finest-level cells), stable, i.e. ray
// order inside a block -- that the fused field kernels walk instead (`perm` argument): consecutive 16-sample tiles
// then touch the same few cells on every level, the forward's gathers hit in L2 and the backward accumulates whole
// lattice tiles in LDS before one merged atomic per corner leaves (table_scatter.hip).  The sample buffers themselves
// stay in ray order (the composite kernels walk them per ray); only the field kernels go through the permutation.
//
// The sort is hand-written (round 3) and CAPTURE-SAFE: every launch goes to the caller's stream, every counter is
// (re)initialised by a kernel of the call itself, nothing is read back, nothing is allocated.  Rounds 1-2 called
// rocPRIM's radix_sort_pairs, whose onesweep passes on gfx942/gfx950 take their tile ids from an atomic counter
// ("ordered block id" hot-fix, rocprim/device/detail/ordered_block_id.hpp:270-281) that is cleared before EVERY pass by
// reset_from_host() = a plain hipMemset that ignores its stream argument (ordered_block_id.hpp:115-118, called from
// device_radix_sort.hpp:274).  That memset is not part of a captured graph: the first replay still finds the zero the
// capture-time call left, the second replay starts from the previous replay's final count, the tile ids run past the
// number of tiles and the kernel indexes its look-back states out of range -- the MEMORY_APERTURE_VIOLATION of round 2.
// It also made every eager call host-synchronise on the null stream four times.
//
// Algorithm: least-significant-digit radix sort of the Morton keys -- 27 significant bits, see k_order_keys -- in three passes of
// 9 bits, the classic
// three-kernel pass (no look-back, no spinning, so no forward-progress assumption):
//   upsweep    block b counts the digits of its contiguous run of 8192-key tiles        -> spine[digit][block]
//   spine      one block per digit: exclusive scan over the blocks, digit total         -> spine, totals[digit]
//   downsweep  block b walks its tiles in order; per tile every wave ranks its 1024 keys 64 at a time (nine ballots give
//              a lane the set of lanes with its digit: rank = per-wave LDS counter + lower lanes in the set; one lane of
//              the set advances the counter -- deterministic, order-preserving), the four waves' counts are prefixed
//              per digit and scanned over the digits, keys then values go through an LDS tile in sorted order so that a
//              digit's run leaves as consecutive lanes' stores; a running base per digit carries over to the next tile.
// Pass 1 reads no value array (value = index), pass 3 writes no keys, the key kernel produces pass 1's counts itself.
// Slots at and past the emitted count (device value) take no part: the permutation is the identity there, which is
// where a stable sort would put keys larger than every Morton code.  Bytes per pair: 16 (keys) + 12 + 20 + 16.
#include <string.h>

#include <hip/hip_runtime.h>

#include "nsr_common.h"
#include "rm_util.h"

#define SO_THREADS 512
#define SO_IPT 16
#define SO_TILE (SO_THREADS * SO_IPT)          // 8192 keys
#define SO_WAVE_ITEMS (64 * SO_IPT)            // 1024 keys per wave and tile
#define SO_BITS 9                              // 27 key bits: see k_order_keys
#define SO_BINS 512
#define SO_BPT (SO_BINS / SO_THREADS)          // digits per thread in the per-digit loops
#define SO_MAX_BLOCKS 1024
#ifndef SO_BLOCKS_TARGET
#define SO_BLOCKS_TARGET 768                   // 3 resident blocks per CU (52 KB of LDS each)
#endif

struct OrderArgs {
    const float *xyzs;
    const int32_t *m_dev;
    uint32_t M, prefix, nb;
    float bmin[3], bsize[3];
    uint32_t *keys, *perm, *spine;
};

struct SortArgs {
    const uint32_t *keys_in, *vals_in;      // vals_in == NULL: value = index
    uint32_t *keys_out, *vals_out;          // keys_out == NULL: last pass
    uint32_t *spine, *totals;               // [SO_BINS][nb], [SO_BINS]
    const int32_t *m_dev;
    uint32_t prefix, nb, shift;
};

__device__ __forceinline__ uint32_t so_count(const int32_t *m_dev, uint32_t prefix) {
    return m_dev ? min((uint32_t)max(m_dev[0], 0), prefix) : prefix;
}
// contiguous run of tiles of block b
__device__ __forceinline__ void so_tile_range(uint32_t n, uint32_t nb, uint32_t b, uint32_t &t0, uint32_t &t1) {
    const uint32_t ntiles = (n + SO_TILE - 1) / SO_TILE;
    const uint32_t per = (ntiles + nb - 1) / nb;
    t0 = min(b * per, ntiles);
    t1 = min(t0 + per, ntiles);
}

__device__ __forceinline__ uint32_t order_spread10(uint32_t v) {
    v &= 0x3FFu;
    v = (v | (v << 16)) & 0x030000FFu;
    v = (v | (v << 8)) & 0x0300F00Fu;
    v = (v | (v << 4)) & 0x030C30C3u;
    v = (v | (v << 2)) & 0x09249249u;
    return v;
}

// keys of the slots below the emitted count + pass 1's digit counts per block; identity permutation for the rest
__global__ void __launch_bounds__(SO_THREADS)
k_order_keys(OrderArgs a) {
    __shared__ uint32_t hist[SO_BINS];
    const uint32_t n = so_count(a.m_dev, a.prefix);
    for (uint32_t d = threadIdx.x; d < SO_BINS; d += SO_THREADS) hist[d] = 0;
    __syncthreads();
    uint32_t t0, t1;
    so_tile_range(n, a.nb, blockIdx.x, t0, t1);
    for (uint32_t t = t0; t < t1; t++) {
#pragma unroll 4
        for (uint32_t k = 0; k < SO_IPT; k++) {
            const uint32_t i = t * SO_TILE + k * SO_THREADS + threadIdx.x;
            if (i < n) {
                uint32_t q[3];
#pragma unroll
                for (int d = 0; d < 3; d++) {
                    // encoder input of the position (BBox.normalize, then GridEncoder's (x + 1) / 2), quantised to 10 bits
                    const float u = ((a.xyzs[(size_t)i * 3 + d] - a.bmin[d]) / a.bsize[d] + 1.0f) * 0.5f;
                    const float s = fminf(fmaxf(u * 1024.0f, 0.0f), 1023.0f);     // NaN -> 0
                    q[d] = (uint32_t)s;
                }
                // The encoder input of a position inside the bounding box is in [0.5, 1] (BBox.normalize to [0, 1], then the
                // encoder's (x + 1) / 2): the top bit of every 10-bit coordinate is set, so the order of the 30-bit Morton codes is
                // the order of the 27-bit codes of the low 9 bits -- three 9-bit digits instead of three 10-bit ones (half the
                // bins: 8-key runs per tile in the scatter instead of 4, 9 ballots per rank instead of 10).  A position outside
                // the box (never produced by the march, which clamps) sorts with the box's face.
#pragma unroll
                for (int d = 0; d < 3; d++) q[d] = q[d] >= 512u ? q[d] - 512u : 0u;
                const uint32_t key = order_spread10(q[0]) | (order_spread10(q[1]) << 1) | (order_spread10(q[2]) << 2);
                a.keys[i] = key;
                atomicAdd(&hist[key & (SO_BINS - 1)], 1u);
            }
        }
    }
    __syncthreads();
    if (blockIdx.x < a.nb)
        for (uint32_t d = threadIdx.x; d < SO_BINS; d += SO_THREADS) a.spine[(size_t)d * a.nb + blockIdx.x] = hist[d];
    // identity for everything the sort does not cover (the grid may be larger than nb for that: blocks past nb own no tiles)
    for (uint32_t i = n + blockIdx.x * SO_THREADS + threadIdx.x; i < a.M; i += gridDim.x * SO_THREADS) a.perm[i] = i;
}

__global__ void __launch_bounds__(SO_THREADS)
k_sort_upsweep(SortArgs a) {
    __shared__ uint32_t hist[SO_BINS];
    const uint32_t n = so_count(a.m_dev, a.prefix);
    for (uint32_t d = threadIdx.x; d < SO_BINS; d += SO_THREADS) hist[d] = 0;
    __syncthreads();
    uint32_t t0, t1;
    so_tile_range(n, a.nb, blockIdx.x, t0, t1);
    for (uint32_t t = t0; t < t1; t++) {
        uint32_t key[SO_IPT];
#pragma unroll
        for (uint32_t k = 0; k < SO_IPT; k++) {
            const uint32_t i = t * SO_TILE + k * SO_THREADS + threadIdx.x;
            key[k] = i < n ? a.keys_in[i] : 0xFFFFFFFFu;
        }
#pragma unroll
        for (uint32_t k = 0; k < SO_IPT; k++) {
            const uint32_t i = t * SO_TILE + k * SO_THREADS + threadIdx.x;
            if (i < n) atomicAdd(&hist[(key[k] >> a.shift) & (SO_BINS - 1)], 1u);
        }
    }
    __syncthreads();
    for (uint32_t d = threadIdx.x; d < SO_BINS; d += SO_THREADS) a.spine[(size_t)d * a.nb + blockIdx.x] = hist[d];
}

// one block per digit: exclusive scan of the digit's per-block counts (nb <= 1024 = 4 per thread)
__global__ void __launch_bounds__(SO_THREADS)
k_sort_spine(SortArgs a) {
    __shared__ uint32_t wsum[SO_THREADS / 64];
    uint32_t *row = a.spine + (size_t)blockIdx.x * a.nb;
    uint32_t v[4], s = 0;
#pragma unroll
    for (uint32_t j = 0; j < 4; j++) {
        const uint32_t i = threadIdx.x * 4 + j;
        v[j] = i < a.nb ? row[i] : 0u;
        s += v[j];
    }
    uint32_t total;
    uint32_t ex = rm_block_exclusive_scan(s, wsum, total);
#pragma unroll
    for (uint32_t j = 0; j < 4; j++) {
        const uint32_t i = threadIdx.x * 4 + j;
        if (i < a.nb) row[i] = ex;
        ex += v[j];
    }
    if (threadIdx.x == 0) a.totals[blockIdx.x] = total;
}

__global__ void __launch_bounds__(SO_THREADS)
k_sort_downsweep(SortArgs a) {
    __shared__ uint32_t cnt[SO_THREADS / 64][SO_BINS];     // per wave: running count of a digit, then its prefix over the waves
    __shared__ uint32_t dstart[SO_BINS];                   // first sorted slot of a digit inside the tile
    __shared__ uint32_t gbase[SO_BINS];                    // next output slot of a digit
    __shared__ uint32_t stage[SO_TILE];
    __shared__ uint32_t wsum[SO_THREADS / 64];

    const uint32_t n = so_count(a.m_dev, a.prefix);
    uint32_t t0, t1;
    so_tile_range(n, a.nb, blockIdx.x, t0, t1);
    if (t0 >= t1) return;
    const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
    const uint64_t lt_mask = (1ull << lane) - 1ull;

    {   // digit starts (scan of the totals) + this block's offset inside each digit
        uint32_t v[SO_BPT], s = 0;
#pragma unroll
        for (uint32_t j = 0; j < SO_BPT; j++) { v[j] = a.totals[tid * SO_BPT + j]; s += v[j]; }
        uint32_t total;
        uint32_t ex = rm_block_exclusive_scan(s, wsum, total);
#pragma unroll
        for (uint32_t j = 0; j < SO_BPT; j++) {
            const uint32_t d = tid * SO_BPT + j;
            gbase[d] = ex + a.spine[(size_t)d * a.nb + blockIdx.x];
            ex += v[j];
        }
    }
#pragma unroll
    for (uint32_t j = 0; j < SO_BINS / 64; j++) cnt[wave][j * 64 + lane] = 0;
    __syncthreads();

    for (uint32_t t = t0; t < t1; t++) {
        const uint32_t base = t * SO_TILE + wave * SO_WAVE_ITEMS + lane;       // item k of this thread: base + 64 k
        uint32_t key[SO_IPT], val[SO_IPT];
#pragma unroll
        for (uint32_t k = 0; k < SO_IPT; k++) {
            const uint32_t i = base + k * 64;
            key[k] = i < n ? a.keys_in[i] : 0xFFFFFFFFu;
        }
        if (a.vals_in) {
#pragma unroll
            for (uint32_t k = 0; k < SO_IPT; k++) {
                const uint32_t i = base + k * 64;
                val[k] = i < n ? a.vals_in[i] : 0u;
            }
        } else {
#pragma unroll
            for (uint32_t k = 0; k < SO_IPT; k++) val[k] = base + k * 64;
        }

        // ---- rank inside the wave's 1024 keys ----
        uint32_t pos[SO_IPT];                                  // digit | rank << 10, later the sorted slot inside the tile
#pragma unroll
        for (uint32_t k = 0; k < SO_IPT; k++) {
            const uint32_t i = base + k * 64;
            const bool valid = i < n;
            const uint32_t d = (key[k] >> a.shift) & (SO_BINS - 1);
            uint64_t peers = __ballot(valid);
#pragma unroll
            for (uint32_t b = 0; b < SO_BITS; b++) {
                const bool bit = (d >> b) & 1u;
                const uint64_t bal = __ballot(bit);
                peers &= bit ? bal : ~bal;
            }
            const uint32_t below = (uint32_t)__popcll(peers & lt_mask);
            uint32_t r = 0;
            if (valid) {
                const uint32_t old = cnt[wave][d];
                if (below == 0) cnt[wave][d] = old + (uint32_t)__popcll(peers);
                r = old + below;
            }
            pos[k] = d | (r << SO_BITS);
        }
        __syncthreads();

        // ---- per digit: prefix over the waves, total; then the scan over the digits ----
        {
            uint32_t tot[SO_BPT], s = 0;
#pragma unroll
            for (uint32_t j = 0; j < SO_BPT; j++) {
                const uint32_t d = tid * SO_BPT + j;
                uint32_t run = 0;
#pragma unroll
                for (uint32_t w = 0; w < SO_THREADS / 64; w++) {
                    const uint32_t c = cnt[w][d];
                    cnt[w][d] = run;
                    run += c;
                }
                tot[j] = run;
                s += run;
            }
            uint32_t total;
            uint32_t ex = rm_block_exclusive_scan(s, wsum, total);    // syncs twice
#pragma unroll
            for (uint32_t j = 0; j < SO_BPT; j++) {
                dstart[tid * SO_BPT + j] = ex;
                ex += tot[j];
            }
        }
        __syncthreads();

        // ---- keys through the tile in sorted order ----
        const uint32_t tile_n = min(n - t * SO_TILE, (uint32_t)SO_TILE);
#pragma unroll
        for (uint32_t k = 0; k < SO_IPT; k++) {
            const uint32_t d = pos[k] & (SO_BINS - 1), r = pos[k] >> SO_BITS;
            const uint32_t p = dstart[d] + cnt[wave][d] + r;
            pos[k] = p;
            if (base + k * 64 < n) stage[p] = key[k];
        }
        __syncthreads();
        uint32_t gpos[SO_IPT];
#pragma unroll
        for (uint32_t k = 0; k < SO_IPT; k++) {
            const uint32_t i = k * SO_THREADS + tid;
            gpos[k] = 0xFFFFFFFFu;
            if (i < tile_n) {
                const uint32_t kk = stage[i];
                const uint32_t d = (kk >> a.shift) & (SO_BINS - 1);
                gpos[k] = gbase[d] + (i - dstart[d]);
                if (a.keys_out) a.keys_out[gpos[k]] = kk;
            }
        }
        __syncthreads();
        // ---- values the same way; the digit bases move on; the wave counters restart ----
#pragma unroll
        for (uint32_t k = 0; k < SO_IPT; k++)
            if (base + k * 64 < n) stage[pos[k]] = val[k];
#pragma unroll
        for (uint32_t j = 0; j < SO_BPT; j++) {
            const uint32_t d = tid * SO_BPT + j;
            const uint32_t next = d + 1 < SO_BINS ? dstart[d + 1] : tile_n;
            gbase[d] += next - dstart[d];
        }
        __syncthreads();
#pragma unroll
        for (uint32_t j = 0; j < SO_BINS / 64; j++) cnt[wave][j * 64 + lane] = 0;
#pragma unroll
        for (uint32_t k = 0; k < SO_IPT; k++) {
            const uint32_t i = k * SO_THREADS + tid;
            if (i < tile_n) a.vals_out[gpos[k]] = stage[i];
        }
        __syncthreads();
    }
}

struct OrderLayout {
    uint64_t keys_a, keys_b, vals_a, spine, totals, total;
};
static OrderLayout order_layout(uint32_t M) {
    OrderLayout l;
    const uint64_t n = ((uint64_t)M * 4 + 255) & ~255ull;
    l.keys_a = 0; l.keys_b = n; l.vals_a = 2 * n;
    l.spine = 3 * n;
    l.totals = l.spine + (uint64_t)SO_BINS * SO_MAX_BLOCKS * 4;
    l.total = l.totals + SO_BINS * 4;
    return l;
}

extern "C" {

uint64_t nsr_sample_order_workspace_bytes(uint32_t M) {
    if (M == 0) return 0;
    return order_layout(M).total;
}

int nsr_sample_order(const float *xyzs, uint32_t M, const int32_t *m_dev, uint32_t sort_prefix, const float *bbox_min,
                     const float *bbox_size, uint32_t *perm, void *workspace, nsr_stream_t stream) {
    if (M == 0) return NSR_OK;
    NSR_CHECK_PTR(xyzs); NSR_CHECK_PTR(bbox_min); NSR_CHECK_PTR(bbox_size); NSR_CHECK_PTR(perm); NSR_CHECK_PTR(workspace);
    if ((uintptr_t)workspace & 255u) return NSR_ERR_INVALID_ARG;
    if (sort_prefix > M) sort_prefix = M;
    hipStream_t s = (hipStream_t)stream;
    const OrderLayout l = order_layout(M);
    char *ws = (char *)workspace;
    uint32_t *keys_a = (uint32_t *)(ws + l.keys_a), *keys_b = (uint32_t *)(ws + l.keys_b), *vals_a = (uint32_t *)(ws + l.vals_a);
    uint32_t *spine = (uint32_t *)(ws + l.spine), *totals = (uint32_t *)(ws + l.totals);
    const uint32_t ntiles = nsr_div_up(sort_prefix, SO_TILE);
    uint32_t nb = ntiles < SO_BLOCKS_TARGET ? ntiles : SO_BLOCKS_TARGET;
    if (nb == 0) nb = 1;

    OrderArgs a;
    a.xyzs = xyzs; a.m_dev = m_dev; a.M = M; a.prefix = sort_prefix; a.nb = nb;
    for (int d = 0; d < 3; d++) { a.bmin[d] = bbox_min[d]; a.bsize[d] = bbox_size[d]; }
    a.keys = keys_a; a.perm = perm; a.spine = spine;
    const uint32_t idg = nsr_grid_1d(M - (m_dev ? 0 : sort_prefix), SO_THREADS);
    hipLaunchKernelGGL(k_order_keys, dim3(nb > idg ? nb : idg), dim3(SO_THREADS), 0, s, a);
    if (sort_prefix > 0) {
        // pass 1: keys_a (value = index) -> keys_b, perm;  pass 2: keys_b, perm -> keys_a, vals_a;  pass 3: keys_a, vals_a -> perm
        SortArgs p;
        p.spine = spine; p.totals = totals; p.m_dev = m_dev; p.prefix = sort_prefix; p.nb = nb;
        for (int pass = 0; pass < 3; pass++) {
            p.shift = pass * SO_BITS;
            p.keys_in = pass == 1 ? keys_b : keys_a;
            p.vals_in = pass == 0 ? nullptr : pass == 1 ? perm : vals_a;
            p.keys_out = pass == 0 ? keys_b : pass == 1 ? keys_a : nullptr;
            p.vals_out = pass == 1 ? vals_a : perm;
            if (pass > 0) hipLaunchKernelGGL(k_sort_upsweep, dim3(nb), dim3(SO_THREADS), 0, s, p);
            hipLaunchKernelGGL(k_sort_spine, dim3(SO_BINS), dim3(SO_THREADS), 0, s, p);
            hipLaunchKernelGGL(k_sort_downsweep, dim3(nb), dim3(SO_THREADS), 0, s, p);
        }
    }
    return nsr_launch_status();
}

}   // extern "C"
