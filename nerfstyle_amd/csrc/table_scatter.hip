// Stand-alone table scatter of the fused field backward, gfx950 (the spatially ordered path).
//
// nsr_field_backward with a `perm` (nsr_sample_order) runs the MLP backward kernel in "gradients out" mode -- it writes
// d loss / d (encoder output) of every sample, 256 B, instead of scattering -- and then THIS kernel, which walks the
// samples in the permutation's spatial order and accumulates the table gradient.  Why two kernels: the MLP backward
// needs one wave per SIMD (240 weight-gradient accumulators per lane), and the scatter is a chain of dependent LDS
// round trips and atomics that one wave per SIMD cannot hide (measured fused: MLP 13.8 ms + scatter 32 ms on the bench
// frame).  Here a wave needs 136 registers and ~9.5 KB of LDS, so three of them share a SIMD.
//
// Lattice accumulator.  The order's key is the sample's BLOCK: its encoder input quantised to 10 bits per axis (a 4^3
// group of finest-level cells for the reference's 16-level grid).  Consecutive samples -- of MANY rays -- share a block,
// and a block touches only a handful of cells on every level: at most ceil(res_l / 1024) + 1 per axis.  So the wave
// keeps, per level, a small LATTICE of corner gradients in LDS anchored at the cell of the block's origin (ceil(res / 1024)
// + 2 corners per axis: for the reference's LLFF grid -- 16 levels, resolutions 16 .. 4096 -- 5^3 on the two finest levels,
// 4^3 on the next two and 3^3 below, 702 float4 = 11 KB per wave; grids up to 6^3 per level and 960 slots in all are accepted).  Lane = (level l = lane >> 2,
// y/z corner pair p = lane & 3) walks the samples in order and adds its two x corners' contributions with a plain LDS
// read-modify-write: within a step the 64 lanes touch 128 different slots and steps are sequential, so no atomics are
// needed; the lattice is addressed by cell coordinates, so nothing is hashed per sample.  When the walk enters a new
// block every level re-anchors; a level whose anchor cell did not change (the coarse ones: a block is a fraction of
// their cell) keeps accumulating, the others are flushed cooperatively, one merged record per corner a block touched
// (tools/sorted_scatter_sim.py: ~6 records and ~3 atomic requests per sample on the bench scene against 29.7 / 19.3 of
// the ray-order run tracker).
#include "field_common.h"
#include "table_scatter.h"


__device__ __forceinline__ bool seq_nonzero(const float4 &v) {
    return ((__float_as_uint(v.x) | __float_as_uint(v.y) | __float_as_uint(v.z) | __float_as_uint(v.w)) << 1) != 0u;
}

// nsr_grid_row for style 0 with the cheap cases taken out (the level is wave-uniform here): a power-of-two table is
// masked, a dense level needs neither 32-bit multiplies (index < 2^19) nor the modulo (index < (res + 1)^3 <= size).
__device__ __forceinline__ uint32_t lat_row(const NsrLevel &lv, uint32_t x, uint32_t y, uint32_t z) {
    if (lv.use_hash) {
        const uint32_t index = x ^ (y * 2654435761u) ^ (z * 805459861u);
        if ((lv.size & (lv.size - 1u)) == 0u) return index & (lv.size - 1u);
        const uint32_t t = __umulhi(lv.magic, index);
        const uint32_t q = (t + ((index - t) >> lv.sh1)) >> lv.sh2;
        return index - q * lv.size;
    }
    return __umul24(x, lv.mul[0]) + __umul24(y, lv.mul[1]) + __umul24(z, lv.mul[2]);
}

struct LatGeom {
    uint16_t base[16];      // first slot of the level's lattice
    uint8_t S[16];          // corners per axis
    uint8_t shift[16];      // the level's lattice is anchored at the origin of the 2^shift-block group the walk is in
};
#ifndef NSR_TS_THREADS
#define NSR_TS_THREADS 256        /* measured on the bench frame: 64 -> 35.6 ms, 128 -> 30.6, 256 -> 29.4 (A + B) */
#endif
// float4 slots per wave, from the 64 KB of LDS a workgroup may ask for: level table + per wave (256 B + 16 B per slot), the
// slot count rounded up to 64 (4 waves: 960).  nsr_table_scatter_supported() and the launch share THIS bound, so a grid is
// rejected before anything is launched or never (round 2 rejected totals of 961..1024 after the MLP backward had run)
constexpr int LAT_MAX_SLOTS = (int)(((65536 - 16 * sizeof(NsrLevel)) / (NSR_TS_THREADS / 64) - 256) / 16 / 64 * 64);
constexpr int LAT_KEY_BITS = 10;                     // must match nsr_sample_order's quantisation
constexpr uint32_t LAT_NONE = 0xFFFFFFFFu;
struct LatState {
    uint32_t b0, b1, b2;     // anchor cell of this lane's level (LAT_NONE: nothing accumulated yet)
};

// corners per axis of a lattice that covers every cell a group of 2^shift blocks (each 1/1024 wide) can touch on a level:
// the group spans e = res * 2^shift / 1024 cells, i.e. at most floor(e) + 2 of them (exactly e when the cells tile it)
static uint32_t lat_corners(uint32_t res, uint32_t shift) {
    const uint64_t span = (uint64_t)res << shift, blocks = 1u << LAT_KEY_BITS;
    const uint32_t cells = (span % blocks == 0) ? (uint32_t)(span / blocks) : (uint32_t)(span / blocks) + 2u;
    return cells + 1u;
}

// Host: lattice geometry.  The walk is in Morton order of the blocks, so the 8 (64, ...) blocks of an aligned group follow
// one another; a level whose cells are larger than a block is anchored at the GROUP's origin as long as that costs no
// lattice slots (a group narrower than a cell still touches at most 2 cells per axis: 3^3 corners) -- its lattice then
// survives the block changes inside the group and is flushed that much less often (bench frame, backward pair: 23.5 ->
// 22.3 ms).  Levels finer than that keep the block as their anchor: paying a 4^3 lattice for a group of two on the levels
// with cells of 1 - 2 blocks was measured and lost (22.7 ms: more slots to scan per flush, more LDS).
static bool lat_geometry(const NsrLevel *lv, LatGeom &g) {
    uint32_t total = 0;
    for (int l = 0; l < 16; l++) {
        const uint32_t res = lv[l].resolution;
        uint32_t shift = 0, S = lat_corners(res, 0);
        while (shift < (uint32_t)LAT_KEY_BITS && lat_corners(res, shift + 1) <= (S > 3u ? S : 3u)) shift++;   // free
#ifdef NSR_TS_FINE_GROUPS
        // experiment: the NSR_TS_FINE_GROUPS finest levels pay a larger lattice for a group of 2^3 blocks (bench frame,
        // backward pair: 1 level 22.14 ms, 2 -> 22.09, 3 -> 22.91 against 22.2: nothing to gain)
        if (l >= 16 - NSR_TS_FINE_GROUPS && lat_corners(res, shift + 1) <= 6u) { shift++; S = lat_corners(res, shift); }
#endif
        S = lat_corners(res, shift) > S ? lat_corners(res, shift) : S;
        if (S < 2u || S > 6u) return false;
        g.S[l] = (uint8_t)S;
        g.shift[l] = (uint8_t)shift;
        g.base[l] = (uint16_t)total;
        total += S * S * S;
    }
    return ((total + 63u) & ~63u) <= (uint32_t)LAT_MAX_SLOTS;
}

// Flushes level l's lattice, anchored at cell (b0, b1, b2) -- wave-uniform arguments -- and clears it.
// 64 slots per trip, three phases so that nothing is computed four times and every LDS round trip is shared:
//   1. one lane per slot: read its float4, test it, and (touched slots only) compute the table row ONCE -> rows[lane];
//   2. four groups of 16 slots, skipped when empty: lane (t = lane >> 2, i = lane & 3) reads component i of slot
//      16q + t and its row and issues the atomic -- the four dwords of a row leave as ONE 16-byte request, x-neighbouring
//      corners (consecutive slots) share their 64-byte line;
//   3. the touched slots are cleared.
template <int S>
__device__ __forceinline__ void lat_flush_level(float4 *__restrict__ lat4, uint32_t *__restrict__ rows, uint32_t b0, uint32_t b1, uint32_t b2,
                                                const NsrLevel &lv, float *__restrict__ gt, int lane, bool td, bool tc) {
    constexpr int NC = S * S * S;
    const int t = lane >> 2, i = lane & 3;
    const bool on = (i < 2) ? td : tc;
    const float *lf = reinterpret_cast<const float *>(lat4);
#pragma unroll
    for (int k0 = 0; k0 < NC; k0 += 64) {
        const int k = k0 + lane;
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (k < NC) v = lat4[k];
        const bool nz = seq_nonzero(v);
        const unsigned long long m = __ballot(nz);
        if (m == 0ull) continue;                                          // wave-uniform: nothing touched in these slots
        if (nz) {
            // k < 256: the divisions by S*S and S are 24-bit multiplies by 16-bit reciprocals (exact on this range); a 32-bit
            // integer multiply costs four issue slots on this machine and the scatter is issue bound
            constexpr uint32_t MZ = (65536u + S * S - 1u) / (S * S), MY = (65536u + S - 1u) / S;
            const uint32_t z = __umul24((uint32_t)k, MZ) >> 16, r = (uint32_t)k - __umul24(z, (uint32_t)(S * S));
            const uint32_t y = __umul24(r, MY) >> 16, x = r - __umul24(y, (uint32_t)S);
            rows[lane] = lv.offset + lat_row(lv, b0 + x, b1 + y, b2 + z);
        }
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int q = 0; q < 4; q++) {
            if (k0 + 16 * q >= NC) break;
            if (((m >> (16 * q)) & 0xFFFFull) == 0ull) continue;          // wave-uniform
            const bool rec = (m >> (16 * q + t)) & 1ull;
            if (rec) {
                const float val = lf[(k0 + 16 * q + t) * 4 + i];
                const uint32_t row = rows[16 * q + t];
#ifndef NSR_ABL_NO_ATOMIC
                if (on) atomicAdd(gt + (size_t)row * 4 + i, val);
#else
                if (on && row == 0xFFFFFFFFu) gt[i] = val;
#endif
            }
        }
        __builtin_amdgcn_wave_barrier();
        if (nz) lat4[k] = make_float4(0.f, 0.f, 0.f, 0.f);
    }
}

// The geometry of level fl comes from the LDS copy of the level table (pad_ = S | shift << 4 | base << 8): a read from the
// kernel-argument segment here would be a vector-memory load, and waiting for it means waiting for every atomic in flight.
__device__ __forceinline__ void lat_flush_dispatch(float4 *__restrict__ lat, int fl, const LatState &st,
                                                   const NsrLevel *__restrict__ lds_lv, float *__restrict__ gt, int lane, bool td, bool tc) {
#ifdef NSR_ABL_TS_NOFLUSH
    return;
#endif
    const uint32_t o0 = (uint32_t)__builtin_amdgcn_readlane((int)st.b0, fl * 4);
    if (o0 == LAT_NONE) return;
    const uint32_t o1 = (uint32_t)__builtin_amdgcn_readlane((int)st.b1, fl * 4);
    const uint32_t o2 = (uint32_t)__builtin_amdgcn_readlane((int)st.b2, fl * 4);
    const NsrLevel flv = lds_lv[fl];
    float4 *lf = lat + (flv.pad_ >> 8);
    uint32_t *rows = reinterpret_cast<uint32_t *>(lat) - 64;                  // 64-entry row scratch in front of the lattices
    switch (flv.pad_ & 0xFu) {
    case 3: lat_flush_level<3>(lf, rows, o0, o1, o2, flv, gt, lane, td, tc); break;
    case 4: lat_flush_level<4>(lf, rows, o0, o1, o2, flv, gt, lane, td, tc); break;
    case 5: lat_flush_level<5>(lf, rows, o0, o1, o2, flv, gt, lane, td, tc); break;
    default: lat_flush_level<6>(lf, rows, o0, o1, o2, flv, gt, lane, td, tc); break;
    }
}


struct TableScatterArgs {
    const float *xyzs;
    const uint32_t *perm;
    const int32_t *m_dev;
    uint32_t M;
    const float4 *gin;            // [M][16] float4: d loss / d (density f0, f1, colour f0, f1) per level
    float *grad_tables;
    float bmin[3], bsize[3];
    int td, tc;
    uint32_t lat_slots;           // float4 slots per wave (multiple of 64)
    NsrLevel lv[16];              // pad_ = S | shift << 4 | base << 8
};

#ifndef NSR_TS_THREADS
#define NSR_TS_THREADS 256        /* measured on the bench frame: 64 -> 35.6 ms, 128 -> 30.6, 256 -> 29.4 (A + B) */
#endif
#ifndef NSR_TS_LDS_ATOMIC
#define NSR_TS_LDS_ATOMIC 0
#endif
#ifndef NSR_TS_RING
#define NSR_TS_RING 8           /* gradient rows in flight per wave (0: one-step-ahead prefetch, the r2 first version) */
#endif
#ifndef NSR_TS_UNROLL
#define NSR_TS_UNROLL 4         /* 1 -> 31.7 ms, 2 -> 30.6, 4 -> 29.4, 8 -> 29.6 */
#endif
constexpr int TS_THREADS = NSR_TS_THREADS;
#ifdef NSR_ABL_TS_NOGIN
#define TS_GIN(v, ln) make_float4(1.0f, 2.0f, 3.0f, (float)(ln))
#else
#define TS_GIN(v, ln) a.gin[(size_t)(uint32_t)__builtin_amdgcn_readlane((int)(v), (ln)) * 16 + l]
#endif
static size_t ts_wave_bytes(uint32_t lat_slots) { return 256 + (size_t)lat_slots * 16; }

#ifndef NSR_TS_WAVES_PER_EU
#define NSR_TS_WAVES_PER_EU 0
#endif
__global__ void __launch_bounds__(TS_THREADS)
#if NSR_TS_WAVES_PER_EU
__attribute__((amdgpu_waves_per_eu(NSR_TS_WAVES_PER_EU, NSR_TS_WAVES_PER_EU)))
#endif
k_table_scatter(TableScatterArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    NsrLevel *lds_lv = reinterpret_cast<NsrLevel *>(smem);
    if (threadIdx.x < 16) lds_lv[threadIdx.x] = a.lv[threadIdx.x];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    char *base = smem + 16 * sizeof(NsrLevel) + (size_t)wave * (256 + (size_t)a.lat_slots * 16);
    float4 *lat = reinterpret_cast<float4 *>(base + 256);               // the 64-entry row scratch sits in front of it
    for (uint32_t k = lane; k < a.lat_slots; k += 64) lat[k] = make_float4(0.f, 0.f, 0.f, 0.f);
    __syncthreads();

    const uint32_t Mc = a.m_dev ? min((uint32_t)max(a.m_dev[0], 0), a.M) : a.M;
    if (Mc == 0) return;
    const uint32_t ntiles = (Mc + 15) / 16;
    const uint32_t nwaves = gridDim.x * (TS_THREADS / 64), gw = blockIdx.x * (TS_THREADS / 64) + wave;
    const uint32_t per = (ntiles + nwaves - 1) / nwaves;
    const uint32_t w_begin = min(gw * per, ntiles), w_end = min(w_begin + per, ntiles);
    if (w_begin >= w_end) return;
    const int s = lane & 15;
    const int l = lane >> 2, py = lane & 1, pz = (lane >> 1) & 1;
    const NsrLevel lv = lds_lv[l];
    const uint32_t S = lv.pad_ & 0xFu, ashift = (lv.pad_ >> 4) & 0xFu;
    float4 *const mylat = lat + (lv.pad_ >> 8) + ((uint32_t)pz * S + (uint32_t)py) * S;
    const bool td = a.td != 0, tc = a.tc != 0;
    float *const gt = a.grad_tables;
    const float kq = (float)(1 << LAT_KEY_BITS), rk = 1.0f / kq;
    LatState st;
    st.b0 = st.b1 = st.b2 = LAT_NONE;
    uint32_t cur_key = LAT_NONE;
    // Round 3: the per-step index arithmetic runs in fp32 (cells, anchors and lattice coordinates are integers < 2^24: exact),
    // which saves the three float->int conversions, the 24-bit integer multiplies and a quarter-rate 64-bit multiply-add of
    // the slot address; the corner weights of the lane's fixed (y, z) corner are one FMA each (w = f * s + o with (s, o) =
    // (1, 0) or (-1, 1)).  The kernel is instruction-issue bound: ~300 issue cycles per step per wave, 168 of them VALU.
    float bf0 = 0.f, bf1 = 0.f, bf2 = 0.f;                           // the anchor (st.b*) as floats
    const float lscale = (float)lv.resolution;                        // nsr_grid_locate with align_corners = 1
    const float lcmax = (float)(lv.resolution - 1u);
    const float Sf = (float)S, Sm2 = (float)(S - 2u);
    const float sy = py ? 1.0f : -1.0f, oy = py ? 0.0f : 1.0f, sz = pz ? 1.0f : -1.0f, oz = pz ? 0.0f : 1.0f;

    // position 16 * tile + s of the order -> buffer index; lanes past the count read the last valid entry (masked later)
    auto fetch_idx = [&](uint32_t tile) -> uint32_t { return a.perm[min(tile * 16 + (uint32_t)s, Mc - 1u)]; };
    uint32_t idx = fetch_idx(w_begin);
    uint32_t idx_next = w_begin + 1 < w_end ? fetch_idx(w_begin + 1) : idx;
    float x0 = a.xyzs[(size_t)idx * 3], x1 = a.xyzs[(size_t)idx * 3 + 1], x2 = a.xyzs[(size_t)idx * 3 + 2];
#if NSR_TS_RING
    // The per-level gradients of the next NSR_TS_RING samples, in registers: a 256-byte row gathered by sample index is a
    // full HBM round trip (~1 us under load), far longer than a step, so a row is requested 8 steps before its use.
    float4 gq[NSR_TS_RING];
#pragma unroll
    for (int i = 0; i < NSR_TS_RING; i++) gq[i] = TS_GIN(idx, i);
#endif

    for (uint32_t tile = w_begin; tile < w_end; tile++) {
        // next tile's inputs: the permutation entry was fetched one tile ahead, so these loads depend on nothing in flight
        float nx0 = x0, nx1 = x1, nx2 = x2;
        uint32_t idx_nn = idx_next;
        if (tile + 1 < w_end) {
            nx0 = a.xyzs[(size_t)idx_next * 3]; nx1 = a.xyzs[(size_t)idx_next * 3 + 1]; nx2 = a.xyzs[(size_t)idx_next * 3 + 2];
            if (tile + 2 < w_end) idx_nn = fetch_idx(tile + 2);
        }
        const bool valid = tile * 16 + (uint32_t)s < Mc;
        const float u0 = field_unit(x0, a.bmin[0], a.bsize[0]), u1 = field_unit(x1, a.bmin[1], a.bsize[1]),
                    u2 = field_unit(x2, a.bmin[2], a.bsize[2]);
        const bool live = valid && (u0 >= 0 && u0 <= 1 && u1 >= 0 && u1 <= 1 && u2 >= 0 && u2 <= 1);
        const uint32_t live16 = (uint32_t)(__ballot(live) & 0xFFFFull);      // lanes 0..15 are samples 0..15
        // this lane's sample's block: the sort key's quantisation (sample_order.hip), 10 bits per axis
        const uint32_t q0 = (uint32_t)fminf(fmaxf(u0 * kq, 0.0f), kq - 1.0f), q1 = (uint32_t)fminf(fmaxf(u1 * kq, 0.0f), kq - 1.0f),
                       q2 = (uint32_t)fminf(fmaxf(u2 * kq, 0.0f), kq - 1.0f);
        const uint32_t bkey = q0 | (q1 << LAT_KEY_BITS) | (q2 << (2 * LAT_KEY_BITS));
        // steps at which the block key MAY differ from the previous live sample's: this sample's key against its left
        // neighbour's (lane 0: against the key the walk is in; dead samples carry a sentinel, so the live sample after one is
        // always flagged).  The exact test runs inside the flagged steps only -- one scalar bit test per step instead of a
        // cross-lane read and a compare.
        const uint32_t bkey_e = live ? bkey : LAT_NONE;
        const uint32_t bkey_l = (uint32_t)__builtin_amdgcn_update_dpp((int)cur_key, (int)bkey_e, 0x111, 0xF, 0xF, false);   // row_shr:1
        const uint32_t chg16 = (uint32_t)(__ballot(bkey_e != bkey_l) & 0xFFFFull);
#if NSR_TS_RING
#pragma unroll 1
        for (int part = 0; part < 16 / NSR_TS_RING; part++) {
        // refills: the rest of this tile first, then the head of the next one (its permutation entries are already here)
        const bool wrap = part == 16 / NSR_TS_RING - 1;
        const uint32_t src = wrap ? idx_next : idx;
        const int src_lane0 = wrap ? 0 : (part + 1) * NSR_TS_RING;
#pragma unroll
        for (int ri = 0; ri < NSR_TS_RING; ri++) {
            const int step = part * NSR_TS_RING + ri;
            const float4 gr = gq[ri];
            do {
            if (!((live16 >> step) & 1u)) break;                               // wave-uniform
#else
        // the per-level gradients of a sample are fetched one step ahead of their use
        float4 gr_next = TS_GIN(idx, 0);
#pragma unroll NSR_TS_UNROLL
        for (int step = 0; step < 16; step++) {
            const float4 gr = gr_next;
            if (step < 15) gr_next = TS_GIN(idx, step + 1);
            if (!((live16 >> step) & 1u)) continue;                            // wave-uniform
#endif
            uint32_t key = cur_key;
            if ((chg16 >> step) & 1u) key = (uint32_t)__builtin_amdgcn_readlane((int)bkey, step);       // wave-uniform
            if (key != cur_key) {
                // ---- the walk enters another block: re-anchor every level at the cell of the block's origin ----
                cur_key = key;
                // origin of the group of 2^ashift blocks (this lane's level) the block belongs to
                const uint32_t kq0 = key & ((1u << LAT_KEY_BITS) - 1u), kq1 = (key >> LAT_KEY_BITS) & ((1u << LAT_KEY_BITS) - 1u),
                               kq2 = key >> (2 * LAT_KEY_BITS);
                const float o0 = (float)((kq0 >> ashift) << ashift) * rk, o1 = (float)((kq1 >> ashift) << ashift) * rk,
                            o2 = (float)((kq2 >> ashift) << ashift) * rk;
                float ff;
                uint32_t n0, n1, n2;
                nsr_grid_locate(o0, lv.resolution, 1, ff, n0);
                nsr_grid_locate(o1, lv.resolution, 1, ff, n1);
                nsr_grid_locate(o2, lv.resolution, 1, ff, n2);
                const bool chg = (n0 != st.b0) | (n1 != st.b1) | (n2 != st.b2);
                unsigned long long mm = __ballot(chg);
                while (mm) {
                    const int fl = (int)(__builtin_ctzll(mm) >> 2);
                    mm &= ~(0xFull << (fl * 4));
                    lat_flush_dispatch(lat, fl, st, lds_lv, gt, lane, td, tc);
                }
                if (chg) { st.b0 = n0; st.b1 = n1; st.b2 = n2; bf0 = (float)n0; bf1 = (float)n1; bf2 = (float)n2; }
            }
            const float su0 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, u0), step));
            const float su1 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, u1), step));
            const float su2 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, u2), step));
            // nsr_grid_locate (align_corners = 1) with the cell kept as a float: p = u * res, c = min(floor(p), res - 1), f = p - c
            float f0, f1, f2, cf0, cf1, cf2;
            {
#pragma clang fp contract(off)
                const float p0 = su0 * lscale, p1 = su1 * lscale, p2 = su2 * lscale;
                cf0 = fminf(floorf(p0), lcmax); cf1 = fminf(floorf(p1), lcmax); cf2 = fminf(floorf(p2), lcmax);
                f0 = p0 - cf0; f1 = p1 - cf1; f2 = p2 - cf2;
            }
            // cell relative to the anchor: 0 .. S - 2 by construction (the sample lies in the block the anchor was taken
            // from; floor(u * res) is monotonic in u)
            const float d0 = cf0 - bf0, d1 = cf1 - bf1, d2 = cf2 - bf2;
            const float r0 = fminf(d0, Sm2), r1 = fminf(d1, Sm2), r2 = fminf(d2, Sm2);
            // this sample's contribution to the lane's two x corners
            const float wyz = fmaf(f1, sy, oy) * fmaf(f2, sz, oz);
            float wB = f0 * wyz, wA = wyz - wB;
            if (fmaxf(d0, fmaxf(d1, d2)) > Sm2) {
                const uint32_t c0 = (uint32_t)cf0, c1 = (uint32_t)cf1, c2 = (uint32_t)cf2;
                // fp32 rounding put the cell one past the lattice (u * res of a sample at the very end of its block can
                // round up across a cell boundary that the block's real extent stops short of): this sample's two corners
                // go straight to the table, exactly; the (clamped) lattice slots get nothing
                const uint32_t rowA = lv.offset + lat_row(lv, c0, c1 + (uint32_t)py, c2 + (uint32_t)pz);
                const uint32_t rowB = lv.offset + lat_row(lv, c0 + 1u, c1 + (uint32_t)py, c2 + (uint32_t)pz);
                const float ga[4] = {gr.x, gr.y, gr.z, gr.w};
#pragma unroll
                for (int i = 0; i < 4; i++) {
                    if ((i < 2 ? td : tc) && ga[i] != 0.0f) {
                        atomicAdd(gt + (size_t)rowA * 4 + i, wA * ga[i]);
                        atomicAdd(gt + (size_t)rowB * 4 + i, wB * ga[i]);
                    }
                }
                wA = 0.0f;
                wB = 0.0f;
            }
            float4 *const slot = mylat + (uint32_t)fmaf(fmaf(r2, Sf, r1), Sf, r0);
#if NSR_TS_LDS_ATOMIC
            // fire-and-forget LDS float adds: no read -> fma -> write round trip to wait for (the lattice is private to the wave
            // and the 64 lanes of a step touch 128 different slots, so nothing conflicts)
            float *const sf = reinterpret_cast<float *>(slot);
            __hip_atomic_fetch_add(sf + 0, wA * gr.x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
            __hip_atomic_fetch_add(sf + 1, wA * gr.y, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
            __hip_atomic_fetch_add(sf + 2, wA * gr.z, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
            __hip_atomic_fetch_add(sf + 3, wA * gr.w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
            __hip_atomic_fetch_add(sf + 4, wB * gr.x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
            __hip_atomic_fetch_add(sf + 5, wB * gr.y, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
            __hip_atomic_fetch_add(sf + 6, wB * gr.z, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
            __hip_atomic_fetch_add(sf + 7, wB * gr.w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
#else
#ifdef NSR_ABL_TS_NORMW
            if (wA == 123.0f) slot[0] = gr;
#else
            float4 va = slot[0], vb = slot[1];
            va.x = fmaf(wA, gr.x, va.x); va.y = fmaf(wA, gr.y, va.y); va.z = fmaf(wA, gr.z, va.z); va.w = fmaf(wA, gr.w, va.w);
            vb.x = fmaf(wB, gr.x, vb.x); vb.y = fmaf(wB, gr.y, vb.y); vb.z = fmaf(wB, gr.z, vb.z); vb.w = fmaf(wB, gr.w, vb.w);
            slot[0] = va;
            slot[1] = vb;
#endif
#endif
#if NSR_TS_RING
            } while (0);
            // the slot's refill is issued AFTER the last use of its old value, so every ring slot stays in the same registers
            // (a refill issued earlier gets other registers, and the copies at the loop edge wait for every load in flight)
            asm volatile("" ::: "memory");
            gq[ri] = TS_GIN(src, src_lane0 + ri);
        }
#endif
        }
        x0 = nx0; x1 = nx1; x2 = nx2;
        idx = idx_next;
        idx_next = idx_nn;
    }
    // every level's lattice leaves
    __builtin_amdgcn_wave_barrier();
#pragma unroll 1
    for (int fl = 0; fl < 16; fl++) lat_flush_dispatch(lat, fl, st, lds_lv, gt, lane, td, tc);
}

bool nsr_table_scatter_supported(const NsrLevel *lv) {
    LatGeom g;
    return lat_geometry(lv, g);
}

int nsr_table_scatter_launch(const NsrLevel *levels, const float *bmin, const float *bsize, const float *xyzs, const uint32_t *perm,
                             const int32_t *m_dev, uint32_t M, const void *gin, float *grad_tables, int td, int tc, hipStream_t s) {
    TableScatterArgs a;
    LatGeom g;
    if (!lat_geometry(levels, g)) return NSR_ERR_UNSUPPORTED;
    uint32_t total = 0;
    for (int l = 0; l < 16; l++) {
        a.lv[l] = levels[l];
        a.lv[l].pad_ = (uint32_t)g.S[l] | ((uint32_t)g.shift[l] << 4) | ((uint32_t)g.base[l] << 8);
        total = g.base[l] + (uint32_t)g.S[l] * g.S[l] * g.S[l];
    }
    a.lat_slots = (total + 63u) & ~63u;
    a.xyzs = xyzs; a.perm = perm; a.m_dev = m_dev; a.M = M; a.gin = (const float4 *)gin; a.grad_tables = grad_tables;
    for (int i = 0; i < 3; i++) { a.bmin[i] = bmin[i]; a.bsize[i] = bsize[i]; }
    a.td = td; a.tc = tc;
#ifndef NSR_ABL_TS_PAD_LDS
#define NSR_ABL_TS_PAD_LDS 0      /* ablation: extra LDS bytes per workgroup, to lower the occupancy */
#endif
    const size_t lds = 16 * sizeof(NsrLevel) + (TS_THREADS / 64) * ts_wave_bytes(a.lat_slots) + NSR_ABL_TS_PAD_LDS;
    static bool attr_set[64] = {};
    int dev = 0;
    (void)hipGetDevice(&dev);
    if (!attr_set[dev & 63]) {
        if (hipFuncSetAttribute(reinterpret_cast<const void *>(&k_table_scatter), hipFuncAttributeMaxDynamicSharedMemorySize, 65536) != hipSuccess)
            return NSR_ERR_LAUNCH;
        attr_set[dev & 63] = true;
    }
    if (lds > 65536) return NSR_ERR_UNSUPPORTED;
    // 16 workgroups per CU (4 096 in all), although only 3-4 are resident at a time: the cost of a run of tiles depends on
    // how often its samples change block, and with one workgroup per resident slot the slowest run decides the launch
    // (bench frame, backward pair: 4 per CU 25.5 ms, 6 -> 24.2, 8 -> 23.9, 16 -> 23.55, 32 -> 23.6)
    uint32_t per_cu = 16;
#ifdef NSR_ABL_TS_PER_CU
    per_cu = NSR_ABL_TS_PER_CU;
#endif
    uint32_t nblocks = 256u * per_cu;
    const uint32_t ntiles = (M + 15) / 16;
    if (nblocks > (ntiles + 3) / 4) nblocks = (ntiles + 3) / 4;
    if (nblocks == 0) nblocks = 1;
    hipLaunchKernelGGL(k_table_scatter, dim3(nblocks), dim3(TS_THREADS), lds, s, a);
    return nsr_launch_status();
}
