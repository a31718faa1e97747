// Device-side occupancy-grid update for gfx950: the counterpart of Renderer.update_state /
// _compute_occ_sigmas (renderer.py:120-194) without a single host read.
//
// The reference builds the query points with torch ops (meshgrid / randint / nonzero / boolean masks),
// reads `mean_density` and the occupied-cell count back to the host, and scatters through index_put.
// Here one update is
//     nsr_occ_sample_points   cell choice + jittered positions (counter-based RNG)        1-4 launches
//     nsr_field_forward       sigma-only field on those points (the caller's launch)
//     nsr_occ_update          scatter -> max/decay -> mean (block tree) -> packbits        3-5 launches
// with every data-dependent quantity (occupied-cell count, mean density, threshold) left in device
// memory, so the whole update is legal inside hipGraph capture and costs no pipeline bubble.
//
// Cells are addressed by their flat index  cas * H^3 + morton(x, y, z)  (the layout of density_grid,
// renderer.py:62-63).  In the full update a thread's point IS its Morton index, so the sigma queries are
// spatially coherent (neighbouring lanes = neighbouring cells = shared hash-table lines) and the result
// lands in the grid with coalesced stores -- no index tensor, no scatter.
#include "nsr_common.h"
#include "rm_util.h"

#define OCC_BLOCK 256

// ---- Philox4x32-10 (Salmon et al., SC'11): counter-based, so a point's random numbers depend only on
// (seed, update sequence, point index) -- identical on every rank without communication -------------------
struct OccU4 { uint32_t x, y, z, w; };
__device__ __forceinline__ OccU4 occ_philox(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0, uint32_t k1) {
#pragma unroll
    for (int r = 0; r < 10; r++) {
        const uint32_t hi0 = __umulhi(0xD2511F53u, c0), lo0 = 0xD2511F53u * c0;
        const uint32_t hi1 = __umulhi(0xCD9E8D57u, c2), lo1 = 0xCD9E8D57u * c2;
        const uint32_t n0 = hi1 ^ c1 ^ k0, n2 = hi0 ^ c3 ^ k1;
        c0 = n0; c1 = lo1; c2 = n2; c3 = lo0;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    return OccU4{c0, c1, c2, c3};
}
// uniform in [0, 1) with 24 random bits (what torch.rand gives a float32)
__device__ __forceinline__ float occ_u01(uint32_t r) { return (float)(r >> 8) * (1.0f / 16777216.0f); }

struct OccSampleArgs {
    const float *density_grid;
    const float *noise;             // optional [P,3] in [0,1): overrides the generated jitter (tests)
    const uint32_t *sequence_dev;   // optional device counter added to `sequence`
    const int32_t *occ_list;        // partial update: [C][H^3] occupied Morton indices, ascending
    const uint32_t *occ_count;      // partial update: [C]
    float *xyzs;
    int32_t *indices;
    uint32_t C, H, H3, P, n_rand;
    float bound;
    uint32_t seed_lo, seed_hi, sequence;
    int full;
};

// renderer.py:120-136 (_compute_occ_sigmas) for one cell; operation order of the torch expression
__device__ __forceinline__ void occ_position(const OccSampleArgs &a, uint32_t p, uint32_t cas, uint32_t cx, uint32_t cy, uint32_t cz,
                                             float u0, float u1, float u2) {
#pragma clang fp contract(off)
    const float b = fminf((float)(1u << cas), a.bound);        // min(2 ** cas, self.bound)
    const float half = b / (float)a.H;                         // half_grid_size
    const float span = b - half;
    const float gm1 = (float)(a.H - 1);
    const float c[3] = {(float)cx, (float)cy, (float)cz};
    const float u[3] = {u0, u1, u2};
#pragma unroll
    for (int d = 0; d < 3; d++) {
        const float xn = 2.0f * c[d] / gm1 - 1.0f;             // 2 * coords.float() / (gs - 1) - 1   (:155,:180)
        float v = xn * span;                                    // xyzs * (bound - half_grid_size)
        v += (u[d] * 2.0f - 1.0f) * half;                       // += (rand * 2 - 1) * half_grid_size
        a.xyzs[(size_t)p * 3 + d] = v;
    }
}

__global__ void __launch_bounds__(OCC_BLOCK)
k_occ_sample(OccSampleArgs a) {
    const uint32_t seq = a.sequence + (a.sequence_dev ? a.sequence_dev[0] : 0u);
    for (uint32_t p = blockIdx.x * OCC_BLOCK + threadIdx.x; p < a.P; p += gridDim.x * OCC_BLOCK) {
        uint32_t cas, m;
        int32_t flat;
        const OccU4 r = occ_philox(p, seq, 0u, 0u, a.seed_lo, a.seed_hi);
        if (a.full) {
            // renderer.py:143-160: every cell of every cascade, point index == flat cell index
            cas = p / a.H3;
            m = p % a.H3;
            flat = (int32_t)p;
        } else {
            // renderer.py:163-181: per cascade n_rand uniform cells, then n_rand cells drawn (with replacement)
            // from the currently occupied ones
            const uint32_t per = 2u * a.n_rand;
            cas = p / per;
            const uint32_t j = p % per;
            const OccU4 q = occ_philox(p, seq, 1u, 0u, a.seed_lo, a.seed_hi);
            if (j < a.n_rand) {
                const uint32_t x = __umulhi(q.x, a.H), y = __umulhi(q.y, a.H), z = __umulhi(q.z, a.H);   // randint(0, gs)
                m = rm_morton3d(x, y, z);
                flat = (int32_t)(cas * a.H3 + m);
            } else {
                const uint32_t cnt = a.occ_count[cas];
                if (cnt == 0u) {
                    // the reference's torch.randint(0, 0) raises here; an empty grid simply has nothing to re-sample
                    m = 0u;
                    flat = -1;
                } else {
                    m = (uint32_t)a.occ_list[(size_t)cas * a.H3 + __umulhi(q.w, cnt)];
                    flat = (int32_t)(cas * a.H3 + m);
                }
            }
        }
        const uint32_t cx = rm_morton3d_invert(m), cy = rm_morton3d_invert(m >> 1), cz = rm_morton3d_invert(m >> 2);
        float u0 = occ_u01(r.x), u1 = occ_u01(r.y), u2 = occ_u01(r.z);
        if (a.noise) { u0 = a.noise[(size_t)p * 3]; u1 = a.noise[(size_t)p * 3 + 1]; u2 = a.noise[(size_t)p * 3 + 2]; }
        occ_position(a, p, cas, cx, cy, cz, u0, u1, u2);
        a.indices[p] = flat;
    }
}

// ---- occupied-cell list (torch.nonzero(density_grid[cas] > 0), renderer.py:169): count -> scan -> write ----
__global__ void __launch_bounds__(OCC_BLOCK)
k_occ_count(const float *__restrict__ grid, uint32_t total, uint32_t *__restrict__ block_sums) {
    __shared__ uint32_t wave_sums[OCC_BLOCK / 64];
    const uint32_t i = blockIdx.x * OCC_BLOCK + threadIdx.x;
    uint32_t tot;
    rm_block_exclusive_scan((i < total && grid[i] > 0.0f) ? 1u : 0u, wave_sums, tot);
    if (threadIdx.x == 0) block_sums[blockIdx.x] = tot;
}
// one block per cascade: exclusive scan of that cascade's block totals, total -> occ_count[cas]
__global__ void __launch_bounds__(1024)
k_occ_scan(uint32_t *__restrict__ block_sums, uint32_t blocks_per_cas, uint32_t *__restrict__ occ_count) {
    __shared__ uint32_t wave_sums[1024 / 64];
    uint32_t *bs = block_sums + (size_t)blockIdx.x * blocks_per_cas;
    uint32_t carry = 0;
    for (uint32_t start = 0; start < blocks_per_cas; start += 1024) {
        const uint32_t i = start + threadIdx.x;
        const uint32_t v = i < blocks_per_cas ? bs[i] : 0u;
        uint32_t tot;
        const uint32_t ex = rm_block_exclusive_scan(v, wave_sums, tot);
        if (i < blocks_per_cas) bs[i] = carry + ex;
        carry += tot;
    }
    if (threadIdx.x == 0) occ_count[blockIdx.x] = carry;
}
__global__ void __launch_bounds__(OCC_BLOCK)
k_occ_list(const float *__restrict__ grid, uint32_t H3, uint32_t blocks_per_cas, const uint32_t *__restrict__ block_bases,
           int32_t *__restrict__ occ_list) {
    __shared__ uint32_t wave_sums[OCC_BLOCK / 64];
    const uint32_t i = blockIdx.x * OCC_BLOCK + threadIdx.x;          // H3 is a multiple of OCC_BLOCK (checked on the host)
    const uint32_t cas = blockIdx.x / blocks_per_cas;
    const bool occ = grid[i] > 0.0f;
    uint32_t tot;
    const uint32_t ex = rm_block_exclusive_scan(occ ? 1u : 0u, wave_sums, tot);
    if (occ) occ_list[(size_t)cas * H3 + block_bases[blockIdx.x] + ex] = (int32_t)(i - cas * H3);
}

// ---- update ---------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(OCC_BLOCK)
k_occ_fill(float *__restrict__ tmp, uint32_t n, float v) {
    for (uint32_t i = blockIdx.x * OCC_BLOCK + threadIdx.x; i < n; i += gridDim.x * OCC_BLOCK) tmp[i] = v;
}
// tmp_grid[cas, indices] = sigmas (renderer.py:160,181).  A partial update draws cells WITH replacement (H^3/4 uniform draws plus
// the occupied half), so a cell can be written several times; the reference's index_put_ lets any of the writers win.  Here the
// LARGEST density wins -- one of the legal outcomes, and the same one on every run and every rank (sigma >= 0: the bit pattern
// of a non-negative float orders like the float, and beats the -1 the buffer was filled with)
__global__ void __launch_bounds__(OCC_BLOCK)
k_occ_scatter(const float *__restrict__ sigmas, const int32_t *__restrict__ indices, uint32_t P, float *__restrict__ tmp) {
    for (uint32_t p = blockIdx.x * OCC_BLOCK + threadIdx.x; p < P; p += gridDim.x * OCC_BLOCK) {
        const int32_t f = indices[p];
        if (f < 0) continue;
        const float s = sigmas[p];
        if (s >= 0.0f) atomicMax(reinterpret_cast<int *>(tmp) + f, __float_as_int(s));
        else tmp[f] = s;                         // negative or NaN: never produced by the field (exp * density_scale)
    }
}
// renderer.py:183-186: grid = max(grid * decay, tmp) where both are >= 0; block sums of clamp(grid, 0)
__global__ void __launch_bounds__(OCC_BLOCK)
k_occ_decay_max(float *__restrict__ grid, const float *__restrict__ tmp, uint32_t n, float decay, float *__restrict__ partial) {
    __shared__ float wsum[OCC_BLOCK / 64];
    float acc = 0.0f;
    for (uint32_t i = blockIdx.x * OCC_BLOCK + threadIdx.x; i < n; i += gridDim.x * OCC_BLOCK) {
        float g = grid[i];
        const float t = tmp[i];
        if (g >= 0.0f && t >= 0.0f) {
            g = fmaxf(g * decay, t);
            grid[i] = g;
        }
        acc += fmaxf(g, 0.0f);
    }
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) acc += __shfl_xor(acc, off, 64);
    if ((threadIdx.x & 63) == 0) wsum[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) {
        float s = 0.0f;
        for (int w = 0; w < OCC_BLOCK / 64; w++) s += wsum[w];
        partial[blockIdx.x] = s;
    }
}
// one block: fixed-order tree over the per-block sums -> mean (renderer.py:186), threshold (:188), sequence++
__global__ void __launch_bounds__(1024)
k_occ_finalize(const float *__restrict__ partial, uint32_t nblocks, uint32_t n, float density_thresh,
               float *__restrict__ mean_density, float *__restrict__ thresh_dev, uint32_t *__restrict__ sequence_dev) {
    __shared__ double sh[1024];
    double acc = 0.0;
    for (uint32_t i = threadIdx.x; i < nblocks; i += 1024) acc += (double)partial[i];
    sh[threadIdx.x] = acc;
    __syncthreads();
    for (uint32_t s = 512; s >= 1; s >>= 1) {
        if (threadIdx.x < s) sh[threadIdx.x] += sh[threadIdx.x + s];
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        const float mean = (float)(sh[0] / (double)n);
        mean_density[0] = mean;
        thresh_dev[0] = fminf(mean, density_thresh);
        if (sequence_dev) sequence_dev[0] += 1u;
    }
}
// packbits (raymarching.cu:366-388) with the threshold read from device memory
__global__ void __launch_bounds__(OCC_BLOCK)
k_occ_packbits(const float *__restrict__ grid, uint32_t N, const float *__restrict__ thresh_dev, uint8_t *__restrict__ bitfield) {
    const float thresh = thresh_dev[0];
    for (uint32_t n = blockIdx.x * OCC_BLOCK + threadIdx.x; n < N; n += gridDim.x * OCC_BLOCK) {
        const float4 a = reinterpret_cast<const float4 *>(grid)[(size_t)n * 2];
        const float4 b = reinterpret_cast<const float4 *>(grid)[(size_t)n * 2 + 1];
        uint32_t bits = 0;
        bits |= (a.x > thresh) ? 1u : 0u;
        bits |= (a.y > thresh) ? 2u : 0u;
        bits |= (a.z > thresh) ? 4u : 0u;
        bits |= (a.w > thresh) ? 8u : 0u;
        bits |= (b.x > thresh) ? 16u : 0u;
        bits |= (b.y > thresh) ? 32u : 0u;
        bits |= (b.z > thresh) ? 64u : 0u;
        bits |= (b.w > thresh) ? 128u : 0u;
        bitfield[n] = (uint8_t)bits;
    }
}

// ---- workspace layout (bytes from the 16-byte aligned base) ----------------------------------------------
struct OccLayout {
    uint64_t tmp, occ_list, block_sums, occ_count, partial, thresh, total;
    uint32_t nblk_cells, nblk_reduce;
};
static OccLayout occ_layout(uint32_t C, uint32_t H) {
    OccLayout l;
    const uint64_t H3 = (uint64_t)H * H * H, cells = (uint64_t)C * H3;
    l.nblk_cells = (uint32_t)((cells + OCC_BLOCK - 1) / OCC_BLOCK);
    l.nblk_reduce = nsr_grid_1d(cells, OCC_BLOCK);
    uint64_t o = 0;
    l.tmp = o; o += cells * 4;
    l.occ_list = o; o += cells * 4;
    l.block_sums = o; o += (uint64_t)l.nblk_cells * 4;
    l.occ_count = o; o += 16 * 4;
    l.partial = o; o += (uint64_t)l.nblk_reduce * 4;
    l.thresh = o; o += 16;
    l.total = (o + 15) & ~15ull;
    return l;
}
static bool occ_dims_ok(uint32_t C, uint32_t H) {
    // H^3 a multiple of the block size (and of 8 for packbits); cells fit int32 flat indices
    return C >= 1 && C <= 8 && H >= 8 && H <= 512 && ((uint64_t)H * H * H) % OCC_BLOCK == 0 &&
           (uint64_t)C * H * H * H < (1ull << 31);
}

extern "C" {

uint64_t nsr_occ_workspace_bytes(uint32_t C, uint32_t H) {
    if (!occ_dims_ok(C, H)) return 0;
    return occ_layout(C, H).total;
}

uint32_t nsr_occ_num_points(uint32_t C, uint32_t H, int full_update) {
    if (!occ_dims_ok(C, H)) return 0;
    const uint32_t H3 = H * H * H;
    return full_update ? C * H3 : C * 2u * (H3 / 4u);       // renderer.py:165: N = grid_size ** 3 // 4
}

int nsr_occ_sample_points(const float *density_grid, uint32_t C, uint32_t H, float bound, int full_update, uint64_t seed,
                          uint32_t sequence, const uint32_t *sequence_dev, const float *noise, float *xyzs, int32_t *indices,
                          void *workspace, nsr_stream_t stream) {
    NSR_CHECK_PTR(density_grid); NSR_CHECK_PTR(xyzs); NSR_CHECK_PTR(indices); NSR_CHECK_PTR(workspace);
    if (!occ_dims_ok(C, H) || !(bound > 0.0f)) return NSR_ERR_INVALID_ARG;
    if ((uintptr_t)workspace & 15u) return NSR_ERR_INVALID_ARG;
    hipStream_t s = (hipStream_t)stream;
    const OccLayout l = occ_layout(C, H);
    char *ws = (char *)workspace;
    const uint32_t H3 = H * H * H;
    OccSampleArgs a;
    a.density_grid = density_grid; a.noise = noise; a.sequence_dev = sequence_dev;
    a.occ_list = (const int32_t *)(ws + l.occ_list); a.occ_count = (const uint32_t *)(ws + l.occ_count);
    a.xyzs = xyzs; a.indices = indices; a.C = C; a.H = H; a.H3 = H3; a.n_rand = H3 / 4u;
    a.P = nsr_occ_num_points(C, H, full_update);
    a.bound = bound; a.seed_lo = (uint32_t)seed; a.seed_hi = (uint32_t)(seed >> 32); a.sequence = sequence;
    a.full = full_update ? 1 : 0;
    if (!full_update) {
        uint32_t *block_sums = (uint32_t *)(ws + l.block_sums);
        const uint32_t bpc = H3 / OCC_BLOCK;
        hipLaunchKernelGGL(k_occ_count, dim3(l.nblk_cells), dim3(OCC_BLOCK), 0, s, density_grid, C * H3, block_sums);
        hipLaunchKernelGGL(k_occ_scan, dim3(C), dim3(1024), 0, s, block_sums, bpc, (uint32_t *)(ws + l.occ_count));
        hipLaunchKernelGGL(k_occ_list, dim3(l.nblk_cells), dim3(OCC_BLOCK), 0, s, density_grid, H3, bpc, block_sums,
                           (int32_t *)(ws + l.occ_list));
    }
    hipLaunchKernelGGL(k_occ_sample, dim3(nsr_grid_1d(a.P, OCC_BLOCK)), dim3(OCC_BLOCK), 0, s, a);
    return nsr_launch_status();
}

int nsr_occ_update(float *density_grid, const float *sigmas, const int32_t *indices, uint32_t P, uint32_t C, uint32_t H,
                   int full_update, float density_decay, float density_thresh, uint8_t *bitfield, float *mean_density,
                   uint32_t *sequence_dev, void *workspace, nsr_stream_t stream) {
    NSR_CHECK_PTR(density_grid); NSR_CHECK_PTR(sigmas); NSR_CHECK_PTR(bitfield); NSR_CHECK_PTR(mean_density);
    NSR_CHECK_PTR(workspace);
    if (!occ_dims_ok(C, H)) return NSR_ERR_INVALID_ARG;
    if (P != nsr_occ_num_points(C, H, full_update)) return NSR_ERR_INVALID_ARG;
    if (!full_update && indices == nullptr) return NSR_ERR_INVALID_ARG;
    if (((uintptr_t)workspace | (uintptr_t)density_grid) & 15u) return NSR_ERR_INVALID_ARG;
    hipStream_t s = (hipStream_t)stream;
    const OccLayout l = occ_layout(C, H);
    char *ws = (char *)workspace;
    const uint32_t cells = C * H * H * H;
    const float *tmp = sigmas;                     // full update: point index == flat cell index
    if (!full_update) {
        float *t = (float *)(ws + l.tmp);
        hipLaunchKernelGGL(k_occ_fill, dim3(l.nblk_reduce), dim3(OCC_BLOCK), 0, s, t, cells, -1.0f);   // renderer.py:141
        hipLaunchKernelGGL(k_occ_scatter, dim3(nsr_grid_1d(P, OCC_BLOCK)), dim3(OCC_BLOCK), 0, s, sigmas, indices, P, t);
        tmp = t;
    }
    float *partial = (float *)(ws + l.partial), *thresh = (float *)(ws + l.thresh);
    hipLaunchKernelGGL(k_occ_decay_max, dim3(l.nblk_reduce), dim3(OCC_BLOCK), 0, s, density_grid, tmp, cells, density_decay,
                       partial);
    hipLaunchKernelGGL(k_occ_finalize, dim3(1), dim3(1024), 0, s, partial, l.nblk_reduce, cells, density_thresh, mean_density,
                       thresh, sequence_dev);
    hipLaunchKernelGGL(k_occ_packbits, dim3(nsr_grid_1d(cells / 8, OCC_BLOCK)), dim3(OCC_BLOCK), 0, s, density_grid, cells / 8,
                       thresh, bitfield);
    return nsr_launch_status();
}

}   // extern "C"
