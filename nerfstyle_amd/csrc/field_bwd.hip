// Fused field backward for gfx950.  One launch does what the reference does with composite's
// saved tensors + 4 tcnn backward launches + 2 grid backward launches (+ two zeros_like fills):
//
//   recompute encode + forward chain (nothing was saved by the forward: no [M,.] activations in
//   HBM) -> trunc_exp' / sigmoid' / ReLU masks -> dgrad chain on MFMA (W^T as the A operand,
//   samples stay on lanes) -> wgrad on MFMA -> table scatter with fp32 atomics.
//
// wgrad needs the sample index on the MFMA k axis, i.e. every activation / gradient block
// transposed.  That is done in registers with one 16x16x16 MFMA against the identity per block
// (exact), not through LDS.  The 60 weight-gradient tiles (15360 fp32) live in the accumulator
// half of the register file for the whole launch: workgroups are 4 waves = one wave per SIMD, so a
// wave owns all 512 registers of its lane slice (240 of them these accumulators) and every wgrad
// MFMA accumulates in place; each wave adds its tiles to grad_mlp once, at the end.
// (A first version accumulated them in LDS with ds_add_f32: that alone cost 59 of 73 ms -- LDS
// float atomics retire at well under one lane per clock -- see profiles/ and DESIGN.md.)
//
// Lane (s = lane&15, g = lane>>4) owns levels {2g, 2g+1, 8+2g, 9+2g} of sample s in both
// directions, so dX arrives from the MFMA on the lane that scatters it.
#include <stdio.h>
#include "field_common.h"

// ---- backward LDS image (units: shorts) ------------------------------------------------------
constexpr int BW_R3T = 0;        // r3^T  [64 x 16]  4 frag16
constexpr int BW_R2T = 1024;     // r2^T  [64 x 64]  8 frag32
constexpr int BW_R1T = 5120;     // r1^T  [16 x 64]  2 frag32
constexpr int BW_C1BT = 6144;    // c1b^T [64 x 16]  4 frag16
constexpr int BW_C1AT = 7168;    // c1a^T [32 x 64]  4 frag32
constexpr int BW_K2T = 9216;     // k2^T  [64 x 16]  4 frag16 (k index = class row + 3)
constexpr int BW_K1T = 10240;    // k1^T  [32 x 64]  4 frag32
constexpr int BW_D2T = 12288;    // d2^T  [64 x 16]  4 frag16
constexpr int BW_D1T = 13312;    // d1^T  [32 x 64]  4 frag32
constexpr int BW_TOTAL = 15360;

constexpr int BWD_THREADS = 256;
constexpr size_t BWD_QUEUE_BYTES_PER_WAVE = 1024 * 16 + 1024 * 4 + 512 * 8;
constexpr size_t BWD_LDS_BYTES = (size_t)FW_TOTAL * 2 + (size_t)BW_TOTAL * 2 + 16 * sizeof(NsrLevel) +
                                 (BWD_THREADS / 64) * BWD_QUEUE_BYTES_PER_WAVE;

struct FieldBwdArgs {
    FieldArgs f;
    const float *grad_sigmas;
    const float *grad_rgbs;
    float *grad_tables;
    float *grad_mlp;
    int train_density, train_color;
    uint32_t nc;
    uint32_t fast_levels;   // bit l: level l is hashed and its size is a power of two
};

template <int CD>
__device__ __forceinline__ void field_build_bw(short *lds, const float *__restrict__ p) {
    mm_build_frags<CD>(lds + BW_R3T, p + P_R3, 16, 64, 4, 16, true, 0, false);
    mm_build_frags<CD>(lds + BW_R2T, p + P_R2, 64, 64, 4, 64, true, 0, true);
    mm_build_frags<CD>(lds + BW_R1T, p + P_R1, 64, 16, 1, 64, true, 0, true);
    mm_build_frags<CD>(lds + BW_C1BT, p + P_C1B, 16, 64, 4, 16, true, 0, false);
    mm_build_frags<CD>(lds + BW_C1AT, p + P_C1A, 64, 32, 2, 64, true, 0, true);
    mm_build_frags<CD>(lds + BW_K2T, p + P_K2, 16, 64, 4, 16, true, CLASS_ROW_SHIFT, false);
    mm_build_frags<CD>(lds + BW_K1T, p + P_K1, 64, 32, 2, 64, true, 0, true);
    mm_build_frags<CD>(lds + BW_D2T, p + P_D2, 16, 64, 4, 16, true, 0, false);
    mm_build_frags<CD>(lds + BW_D1T, p + P_D1, 64, 32, 2, 64, true, 0, true);
}

// wgrad of one layer over this wave's 16 samples, accumulated in registers:
//   dW[o][i] += sum_s G[o][s] * A[i][s]
// Gt / At are the transposed blocks (lane = feature, elements = samples 4g+e).  Tile (ot,it) of
// the result: lane (i = lane&15, g) element e = dW[16ot + 4g + e][16it + i].
template <int CD, int NG, int NA>
__device__ __forceinline__ void field_wgrad(f4v (&acc)[NG * NA], const s4v (&Gt)[NG], const s4v (&At)[NA]) {
#pragma unroll
    for (int ot = 0; ot < NG; ot++) {
#pragma unroll
        for (int it = 0; it < NA; it++) acc[ot * NA + it] = MM<CD>::k16(Gt[ot], At[it], acc[ot * NA + it]);
    }
}

// End of launch: one global atomic per weight per wave.  Rows are shifted by row_shift and clipped
// to [0, row_hi) (rows outside hold zeros by construction: padded outputs get no gradient).
template <int NG, int NA>
__device__ __forceinline__ void field_wgrad_flush(float *__restrict__ gw, int in_p, int row_shift, int row_hi,
                                                  const f4v (&acc)[NG * NA], int lane) {
    const int i = lane & 15, g = lane >> 4;
#pragma unroll
    for (int ot = 0; ot < NG; ot++) {
#pragma unroll
        for (int it = 0; it < NA; it++) {
#pragma unroll
            for (int e = 0; e < 4; e++) {
                const int row = 16 * ot + 4 * g + e - row_shift;
                const float v = acc[ot * NA + it][e];
                if (row >= 0 && row < row_hi && v != 0.0f) atomicAdd(gw + row * in_p + 16 * it + i, v);
            }
        }
    }
}

template <int CD>
__device__ __forceinline__ void field_tr2(const s8v (&x)[1], s4v ident, s4v (&out)[2]) {
    out[0] = mm_transpose16<CD>(mm_lo(x[0]), ident);
    out[1] = mm_transpose16<CD>(mm_hi(x[0]), ident);
}
template <int CD>
__device__ __forceinline__ void field_tr4(const s8v (&x)[2], s4v ident, s4v (&out)[4]) {
    out[0] = mm_transpose16<CD>(mm_lo(x[0]), ident);
    out[1] = mm_transpose16<CD>(mm_hi(x[0]), ident);
    out[2] = mm_transpose16<CD>(mm_lo(x[1]), ident);
    out[3] = mm_transpose16<CD>(mm_hi(x[1]), ident);
}

// 4 gradient tiles -> masked by the forward activation -> two K=32 B fragments
template <int CD>
__device__ __forceinline__ void field_mask_pack(const f4v (&gacc)[4], const s8v (&act)[2], s8v (&out)[2]) {
    out[0] = mm_cat(mm_round4<CD, false>(mm_relu_mask(gacc[0], mm_lo(act[0]))),
                    mm_round4<CD, false>(mm_relu_mask(gacc[1], mm_hi(act[0]))));
    out[1] = mm_cat(mm_round4<CD, false>(mm_relu_mask(gacc[2], mm_lo(act[1]))),
                    mm_round4<CD, false>(mm_relu_mask(gacc[3], mm_hi(act[1]))));
}

// ---- table scatter ----------------------------------------------------------------------------
// Global float atomics are priced per 64-byte request at the memory side (~1e10 requests/s chip
// wide), not per byte: 64 lanes adding one dword each to 64 different rows cost 64 requests.  The
// naive scatter (4 dword atomics per corner per lane) is 512 requests per sample and ran the whole
// backward at 21 M samples/s.  Two reductions of the request count, both exact up to fp32
// summation order:
//   1. the 16 lanes of a DPP row hold 16 CONSECUTIVE samples of one level; consecutive samples of
//      a ray share cells on the coarse and middle levels, so equal rows form runs: a segmented
//      scan over the row (v_mov_dpp row_shr) sums each run and only its last lane emits a record;
//   2. records {row, d0, d1, c0, c1} go through a small per-wave LDS queue and are drained 16 per
//      wave-instruction with 4 lanes per record, so the four dwords of an interleaved row leave as
//      ONE 16-byte request instead of four 4-byte ones.
//   3. the queue is a RING drained in small paced bursts spread over the NEXT tile's compute
//      (SCQ_PACE): atomics are fire-and-forget, but a burst of ~45 back-to-back wave-instructions
//      blocks at issue once the memory side is saturated, and with one wave per SIMD a blocked
//      wave is an idle SIMD.  Pacing lets the atomic service time hide under the MFMA/VALU work.
#ifdef NSR_ABL_STATS
__device__ unsigned long long g_stats[8];
#define NSR_STAT_ALWAYS(i, n) do { if (lane == 0) atomicAdd(&g_stats[i], (unsigned long long)(n)); } while (0)
#ifdef NSR_ABL_COUNTS
#define NSR_STAT(i, n) NSR_STAT_ALWAYS(i, n)
#else
#define NSR_STAT(i, n) do { } while (0)
#endif
#else
#define NSR_STAT(i, n) do { } while (0)
#endif
#ifdef NSR_ABL_PRIV_ROWS
#ifndef NSR_ABL_PRIV_COPIES
#define NSR_ABL_PRIV_COPIES 8
#endif
__device__ float g_priv[NSR_ABL_PRIV_COPIES][(NSR_ABL_PRIV_ROWS + 8) * 4];
#endif
constexpr int SCQ_CAP = 1024;    // records per wave (power of two)
constexpr int SCQ_MASK = SCQ_CAP - 1;
#ifndef NSR_SCQ_KEEP
#define NSR_SCQ_KEEP 512
#endif
constexpr int SCQ_KEEP = NSR_SCQ_KEEP;     // records the paced drain leaves in the ring for cross-tile merging
constexpr int SCATTER_PACE = 0;  // atomic wave-instructions per corner pair in the last two scatter calls (measured: 2 is slower than 0)
constexpr int SCQ_SEEN = 1024;   // direct-mapped "row -> record still in the ring" table per wave (record index only:
                                 // the row is verified against the ring itself)
struct ScatterQueue {
    uint32_t *rows;              // [SCQ_CAP]
    float4 *vals;                // [SCQ_CAP]
    int *seen_idx;               // [SCQ_SEEN] (monotonic) index of the last record pushed under this hash
    int head, tail;              // wave-uniform, monotonically increasing record indices
};

// DPP row shifts with bound_ctrl:1 -- lanes shifted in from outside the 16-lane row read 0, so no
// "old" register has to be initialised and the move can fold into the consuming VOP2
template <int K>
__device__ __forceinline__ float dpp_shr_f(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x110 + K, 0xF, 0xF, true));
}
__device__ __forceinline__ uint32_t dpp_shr1_u(uint32_t v) {
    return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x111, 0xF, 0xF, true);
}
__device__ __forceinline__ uint32_t dpp_shl1_u(uint32_t v) {
    return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x101, 0xF, 0xF, true);
}

// one Hillis-Steele step of a segmented inclusive scan over a 16-lane DPP row.  m is 1.0 while the
// lanes K to the left still belong to this lane's run, 0.0 once a run head lies in between; the zero
// shifted in from outside the row ends every run at the row boundary.
template <int K>
__device__ __forceinline__ void seg_step(float4 &v, float &m) {
    const float a = dpp_shr_f<K>(v.x), b = dpp_shr_f<K>(v.y), c = dpp_shr_f<K>(v.z), d = dpp_shr_f<K>(v.w);
    const float mu = dpp_shr_f<K>(m);
    v.x = fmaf(a, m, v.x); v.y = fmaf(b, m, v.y); v.z = fmaf(c, m, v.z); v.w = fmaf(d, m, v.w);
    m *= mu;
}

// N full groups of 16 records: all LDS reads first, then the N atomic wave-instructions (4 lanes per
// record: the 4 dwords of an interleaved row leave as ONE 16-byte request).  With one wave per SIMD an
// LDS round trip per instruction would be fully exposed.  Ring rows are stored +1 (see the scatter):
// gt1 = grad_tables - 4 floats.
template <int N>
__device__ __forceinline__ void scq_drain(ScatterQueue &q, float *__restrict__ gt1, int t, int i, bool on) {
    uint32_t row[N];
    float v[N];
#pragma unroll
    for (int k = 0; k < N; k++) {
        const int slot = (q.head + 16 * k + t) & SCQ_MASK;
        row[k] = q.rows[slot];
        v[k] = reinterpret_cast<const float *>(q.vals)[slot * 4 + i];
    }
    if (on) {
#pragma unroll
        for (int k = 0; k < N; k++) {
#ifdef NSR_ABL_PRIV_ROWS
            if (row[k] <= NSR_ABL_PRIV_ROWS) atomicAdd(&g_priv[blockIdx.x % NSR_ABL_PRIV_COPIES][row[k] * 4 + i], v[k]);
            else atomicAdd(gt1 + (size_t)row[k] * 4 + i, v[k]);
#elif !defined(NSR_ABL_NO_ATOMIC)
            atomicAdd(gt1 + (size_t)row[k] * 4 + i, v[k]);
#else
            if (row[k] == 0xFFFFFFFFu) gt1[i] = v[k];
#endif
        }
    }
    q.head += 16 * N;
}

// Issues up to `max_instr` atomic wave-instructions of 16 records.  Only full groups unless `flush`.
__device__ __forceinline__ void scq_pace(ScatterQueue &q, float *__restrict__ gt1, int lane, bool td, bool tc, int max_instr,
                                         bool flush, int keep = 0) {
    __builtin_amdgcn_wave_barrier();
    const int t = lane >> 2, i = lane & 3;
    const bool on = (i < 2) ? td : tc;
    // `keep` newest records stay in the ring (paced drains only): the next tile's scatter can still merge
    // into them through the dedupe table
    int full = (q.tail - q.head - keep) >> 4;
    if (full > max_instr) full = max_instr;
    NSR_STAT(2, full);
    for (; full >= 4; full -= 4) scq_drain<4>(q, gt1, t, i, on);
    if (full >= 2) { scq_drain<2>(q, gt1, t, i, on); full -= 2; }
    if (full >= 1) scq_drain<1>(q, gt1, t, i, on);
    if (flush) {
        const int n = q.tail - q.head;          // < 16 when max_instr did not bound the loop above
        if (t < n && on) {
            const int slot = (q.head + t) & SCQ_MASK;
            atomicAdd(gt1 + (size_t)q.rows[slot] * 4 + i, reinterpret_cast<const float *>(q.vals)[slot * 4 + i]);
        }
        q.head += n < 16 ? n : 16;
    }
    __builtin_amdgcn_wave_barrier();
}

// One level, all 8 corners, for the 16-sample tile (this lane = one sample of one level group).
// FAST: every level handled by this call (one per 16-lane group) is hashed with a power-of-two size
// (wave-uniform, from the host's level table), so the row is (x ^ y*P1 ^ z*P2) & (size-1): the same
// value nsr_grid_row computes, without the per-lane dense/hash select and the invariant-divisor modulo.
// Dead lanes arrive with u = 0 and zero gradients: they map to cell 0 with an all-zero value, extend a
// neighbour's run harmlessly or form a zero run that is never pushed.
template <bool FAST, int PACE>
__device__ __forceinline__ void field_scatter_level(const NsrLevel &lv, ScatterQueue &q, float *__restrict__ gt1, float u0, float u1,
                                                    float u2, float gd0, float gd1, float gc0, float gc1, bool swapx,
                                                    int lane, bool td, bool tc) {
#ifdef NSR_ABL_NO_SCATTER
    if (lv.resolution != 0xFFFFFFFFu) return;
#endif
    float f[3];
    uint32_t c[3];
    nsr_grid_locate(u0, lv.resolution, 1, f[0], c[0]);
    nsr_grid_locate(u1, lv.resolution, 1, f[1], c[1]);
    nsr_grid_locate(u2, lv.resolution, 1, f[2], c[2]);
    // (wx*wy)*wz, the product order of the forward interpolation
    const float wxy[4] = {(1 - f[0]) * (1 - f[1]), f[0] * (1 - f[1]), (1 - f[0]) * f[1], f[0] * f[1]};
    const float wz[2] = {1 - f[2], f[2]};
    const uint32_t hy[2] = {c[1] * 2654435761u, (c[1] + 1) * 2654435761u};
    const uint32_t hz[2] = {c[2] * 805459861u, (c[2] + 1) * 805459861u};
    const uint32_t off1 = lv.offset + 1u, msk = lv.size - 1u;
    // stream 0 = the TRAILING x corner with respect to the ray's direction of travel, stream 1 = the leading
    // one: when the ray steps one cell in x, lane s+1's trailing corner is lane s's leading corner
    const uint32_t cx[2] = {c[0] + (swapx ? 1u : 0u), c[0] + (swapx ? 0u : 1u)};
    const float wsel[2][2] = {{swapx ? wxy[1] : wxy[0], swapx ? wxy[3] : wxy[2]}, {swapx ? wxy[0] : wxy[1], swapx ? wxy[2] : wxy[3]}};
    // Corners are handled as x / x+1 PAIRS and a lane's two records are adjacent in the ring: the hash
    // prime for x is 1, so the two rows are neighbours (same 64-byte line 3 times out of 4) and leave in
    // the same 16-record atomic instruction, where the memory side merges them into one request
    // (tools/scatter_sim.py: 26.1 -> 19.1 requests per sample on the bench scene).
#pragma unroll
    for (uint32_t pr = 0; pr < 4; pr++) {
        // make room for this pair's (at most 128) records; rare
        if (q.tail - q.head > SCQ_CAP - 128) { NSR_STAT(3, 1); scq_pace(q, gt1, lane, td, tc, 16, false); }
        uint32_t key[2], hsh[2], seen_r[2];
        int seen_j[2];
        float4 v[2];
        float m[2];
        bool push[2];
#pragma unroll
        for (uint32_t x = 0; x < 2; x++) {
            // run = consecutive samples with the same table row.  Keys are row + 1, so the zero a DPP shift
            // reads outside the row never matches; the drain subtracts the 1 through its base pointer.
            if (FAST) key[x] = off1 + ((cx[x] ^ hy[pr & 1] ^ hz[pr >> 1]) & msk);
            else key[x] = off1 + nsr_grid_row(lv, cx[x], c[1] + (pr & 1), c[2] + (pr >> 1), 0u);
            // dedupe-table lookup issued NOW (it only needs the key) so that the DPP scan below hides the
            // LDS latency.  x-neighbouring rows differ in their low bits: they never evict each other.
            hsh[x] = key[x] & (SCQ_SEEN - 1);
            seen_j[x] = q.seen_idx[hsh[x]];
            seen_r[x] = q.rows[seen_j[x] & SCQ_MASK];      // stale or recycled slots fail the window test below
            const float w = wsel[x][pr & 1] * wz[pr >> 1];
            v[x] = make_float4(w * gd0, w * gd1, w * gc0, w * gc1);
            m[x] = (dpp_shr1_u(key[x]) != key[x]) ? 0.0f : 1.0f;
            push[x] = dpp_shl1_u(key[x]) != key[x];           // run tail
        }
        // leading stream first; where the next lane's trailing row is this lane's leading row (the ray
        // stepped one cell in x) the finished run sum moves over in registers and continues there, instead
        // of becoming a second record with the same address in the same atomic instruction
        const bool absorb = dpp_shr1_u(key[1]) == key[0];      // my trailing run continues lane s-1's leading run
        const bool absorbed = dpp_shl1_u(key[0]) == key[1];    // my leading run is continued by lane s+1
#ifndef NSR_ABL_NO_SCAN
        seg_step<1>(v[1], m[1]); seg_step<2>(v[1], m[1]); seg_step<4>(v[1], m[1]); seg_step<8>(v[1], m[1]);
        {
            const float a = dpp_shr_f<1>(v[1].x), b2 = dpp_shr_f<1>(v[1].y), c2 = dpp_shr_f<1>(v[1].z), d = dpp_shr_f<1>(v[1].w);
            const float t = absorb ? 1.0f : 0.0f;
            v[0].x = fmaf(a, t, v[0].x); v[0].y = fmaf(b2, t, v[0].y); v[0].z = fmaf(c2, t, v[0].z); v[0].w = fmaf(d, t, v[0].w);
        }
        seg_step<1>(v[0], m[0]); seg_step<2>(v[0], m[0]); seg_step<4>(v[0], m[0]); seg_step<8>(v[0], m[0]);
#endif
        push[1] = push[1] && !absorbed;
#pragma unroll
        for (uint32_t x = 0; x < 2; x++) {
            // a run whose summed gradient is exactly zero (e.g. samples behind an opaque surface: the
            // composite backward gives them zero gradient) adds nothing: skip its request
            const uint32_t any = __float_as_uint(v[x].x) | __float_as_uint(v[x].y) | __float_as_uint(v[x].z) | __float_as_uint(v[x].w);
            push[x] = push[x] && (any << 1) != 0u;
#ifdef NSR_ABL_NO_PUSH
            push[x] = push[x] && m[x] == 12345.f;
#endif
#ifdef NSR_ABL_MIN_ROW
            push[x] = push[x] && key[x] > NSR_ABL_MIN_ROW;
#endif
#ifdef NSR_ABL_MAX_ROW
            push[x] = push[x] && key[x] <= NSR_ABL_MAX_ROW;
#endif
#ifndef NSR_ABL_NO_DEDUPE
            // Exact duplicate addresses are the one thing the atomic path never merges (tools/
            // atomic_merge_rule.hip), and they are common: face-adjacent cells share 4 of their 8 corner
            // rows, runs continue across tiles.  If this row was pushed recently and its record is still in
            // the ring, add into that record instead of emitting another request.
            if (push[x] && seen_r[x] == key[x] && (uint32_t)(seen_j[x] - q.head) < (uint32_t)(q.tail - q.head)) {
                float *dst = reinterpret_cast<float *>(q.vals + (seen_j[x] & SCQ_MASK));
                atomicAdd(dst + 0, v[x].x); atomicAdd(dst + 1, v[x].y); atomicAdd(dst + 2, v[x].z); atomicAdd(dst + 3, v[x].w);
                push[x] = false;
            }
#endif
        }
        const unsigned long long mask0 = __ballot(push[0]), mask1 = __ballot(push[1]);
        const int below = (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(mask0 >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mask0, 0u)) +
                          (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(mask1 >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mask1, 0u));
        const int idx0 = q.tail + below, idx1 = idx0 + (push[0] ? 1 : 0);
        if (push[0]) {
            const int slot = idx0 & SCQ_MASK;
            q.rows[slot] = key[0];
            q.vals[slot] = v[0];
            q.seen_idx[hsh[0]] = idx0;
        }
        if (push[1]) {
            const int slot = idx1 & SCQ_MASK;
            q.rows[slot] = key[1];
            q.vals[slot] = v[1];
            q.seen_idx[hsh[1]] = idx1;
        }
        q.tail += (int)__popcll(mask0) + (int)__popcll(mask1);
        NSR_STAT(0, __popcll(mask0) + __popcll(mask1));
        if (PACE > 0) scq_pace(q, gt1, lane, td, tc, PACE, false);
    }
}

template <typename TT, int CD>
__global__ void __launch_bounds__(BWD_THREADS)
k_field_bwd(FieldBwdArgs b) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    short *wl = reinterpret_cast<short *>(smem);
    short *wt = wl + FW_TOTAL;
    NsrLevel *lds_lv = reinterpret_cast<NsrLevel *>(smem + (size_t)(FW_TOTAL + BW_TOTAL) * 2);
    const FieldArgs &a = b.f;
    field_build_fw<CD, false>(wl, a.params);
    field_build_bw<CD>(wt, a.params);
    if (threadIdx.x < 16) lds_lv[threadIdx.x] = a.lv[threadIdx.x];
    __syncthreads();

    const uint32_t Mc = a.m_dev ? min((uint32_t)max(a.m_dev[0], 0), a.M) : a.M;
    const uint32_t ntiles = (Mc + 15) / 16;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int s = lane & 15, g = lane >> 4;
    const TT *tables = reinterpret_cast<const TT *>(a.tables);
    const uint32_t lb = field_logical_block();
    // split by the DEVICE-side sample count (M is only a capacity): every block gets work
    const uint32_t tpb = (ntiles + gridDim.x - 1) / gridDim.x;
    const uint32_t t_begin = lb * tpb;
    const uint32_t t_end = min(t_begin + tpb, ntiles);
    const s4v ident = mm_identity_frag<CD>(lane);
    const int nc = (int)b.nc;
    ScatterQueue q;
    {
        char *qbase = smem + (size_t)(FW_TOTAL + BW_TOTAL) * 2 + 16 * sizeof(NsrLevel) +
                      (size_t)wave * BWD_QUEUE_BYTES_PER_WAVE;
        q.vals = reinterpret_cast<float4 *>(qbase);
        q.rows = reinterpret_cast<uint32_t *>(qbase + SCQ_CAP * 16);
        q.seen_idx = reinterpret_cast<int *>(qbase + SCQ_CAP * 20);
        for (int k = lane; k < SCQ_SEEN; k += 64) q.seen_idx[k] = -1;
        for (int k = lane; k < SCQ_CAP; k += 64) q.rows[k] = 0u;        // keys are row + 1: 0 matches nothing
        q.head = q.tail = 0;
    }
    const bool td = b.train_density != 0, tc = b.train_color != 0;
    float *const gt1 = b.grad_tables - 4;      // ring rows are stored +1 (field_scatter_level)
    // weight-gradient accumulators (60 tiles x 4 regs), resident for the whole launch
    f4v w_r3[4], w_r2[16], w_r1[4], w_c1b[4], w_c1a[8], w_k2[4], w_k1[8], w_d2[4], w_d1[8];
    {
        const f4v z = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int q4 = 0; q4 < 4; q4++) { w_r3[q4] = z; w_r1[q4] = z; w_c1b[q4] = z; w_k2[q4] = z; w_d2[q4] = z; }
#pragma unroll
        for (int q8 = 0; q8 < 8; q8++) { w_c1a[q8] = z; w_k1[q8] = z; w_d1[q8] = z; }
#pragma unroll
        for (int q16 = 0; q16 < 16; q16++) w_r2[q16] = z;
    }

    // Software rotation around the in-order vmcnt counter: a tile's records are pushed to the LDS ring
    // by its (atomic-free) scatter phase, the next tile's inputs are loaded right after it, and the
    // atomics are issued by pace points inside the NEXT tile's dgrad / wgrad section, where no load
    // result is consumed.  In the straightforward order every load-use waits for a full trip of
    // freshly issued atomics to the memory-side atomic unit (41 % of the wave's cycles, SQ_WAIT_ANY).
    // A tile's raw inputs: loads only, nothing here consumes a loaded value (a use would make the
    // compiler wait for the whole memory round trip inside the prefetch).  Lanes past the sample count
    // read sample 0 (always in bounds) and are masked where the values are used.
    struct TileIn {
        float x0, x1, x2;
        s8v xd, xc;
        float gsig;        // grad_sigmas[m] (used by the g == 0 lanes)
        float grgb[4];     // grad_rgbs[m, 4g .. 4g+3]
    };
    auto load_tile = [&](uint32_t tile) {
        TileIn r;
        const uint32_t m = tile * 16 + s;
        const size_t mc = m < Mc ? m : 0u;
        r.x0 = a.xyzs[mc * 3 + 0];
        r.x1 = a.xyzs[mc * 3 + 1];
        r.x2 = a.xyzs[mc * 3 + 2];
        r.gsig = b.grad_sigmas[mc];
        const float *gp = b.grad_rgbs + mc * a.C_ch;
        if (a.C_ch == 8) {
            const float4 t4 = reinterpret_cast<const float4 *>(gp)[g & 1];
            r.grgb[0] = t4.x; r.grgb[1] = t4.y; r.grgb[2] = t4.z; r.grgb[3] = t4.w;
        } else {
#pragma unroll
            for (int e = 0; e < 4; e++) r.grgb[e] = gp[(uint32_t)(4 * g + e) < a.C_ch ? 4 * g + e : 0];
        }
        if (a.feats) {
            // the forward saved this lane's two B fragments: two 16-byte loads instead of 32 gathers
            const s8v *fi = reinterpret_cast<const s8v *>(a.feats) + ((size_t)tile * 64 + lane) * 2;
            r.xd = fi[0];
            r.xc = fi[1];
        }
        return r;
    };
    // Each wave walks a CONTIGUOUS quarter of the block's tiles: consecutive tiles continue the same ray,
    // so the rows of its coarse and middle levels recur and merge with records still held in the ring.
    const uint32_t wchunk = (t_end > t_begin ? (t_end - t_begin + BWD_THREADS / 64 - 1) / (BWD_THREADS / 64) : 0u);
    const uint32_t w_begin = min(t_begin + wave * wchunk, t_end), w_end = min(w_begin + wchunk, t_end);
    TileIn cur;
    if (w_begin < w_end) cur = load_tile(w_begin);

#ifdef NSR_ABL_STATS
    unsigned long long tacc[4] = {0, 0, 0, 0};
#define NSR_TICK(var) const unsigned long long var = __builtin_amdgcn_s_memtime()
#define NSR_TACC(i, a, c) tacc[i] += (c) - (a)
#else
#define NSR_TICK(var) do { } while (0)
#define NSR_TACC(i, a, c) do { } while (0)
#endif
    for (uint32_t tile = w_begin; tile < w_end; tile++) {
        NSR_TICK(tk0);
#ifdef NSR_ABL_STATS
        __builtin_amdgcn_s_waitcnt(0);
#endif
        NSR_TICK(tk1);
        NSR_TACC(0, tk0, tk1);
        const uint32_t m = tile * 16 + s;
        const bool valid = m < Mc;
        const float u0 = valid ? field_unit(cur.x0, a.bmin[0], a.bsize[0]) : 0.f;
        const float u1 = valid ? field_unit(cur.x1, a.bmin[1], a.bsize[1]) : 0.f;
        const float u2 = valid ? field_unit(cur.x2, a.bmin[2], a.bsize[2]) : 0.f;
        const bool live = valid && !(u0 < 0 || u0 > 1 || u1 < 0 || u1 > 1 || u2 < 0 || u2 > 1);
        const float cur_gsig = valid ? cur.gsig : 0.f;
        float cur_grgb[4];
#pragma unroll
        for (int e = 0; e < 4; e++)
            cur_grgb[e] = (valid && (a.C_ch == 8 ? g < 2 : (uint32_t)(4 * g + e) < a.C_ch)) ? cur.grgb[e] : 0.f;
        // no saved features: gather them now (dependent loads, the slow path)
        if (!a.feats) field_encode<TT, CD, false>(lds_lv, tables, u0, u1, u2, live, g, cur.xd, cur.xc);

        // paced drain of the previous tile's records: SCQ_PACE(n) issues <= n atomic wave-instructions
#define SCQ_PACE(n) scq_pace(q, gt1, lane, td, tc, (n), false, SCQ_KEEP)
        // ================= recompute forward, keeping rounded activations ====================
        s8v xd[1] = {cur.xd}, xc[1] = {cur.xc};
        f4v h[4];
        s8v hd[2], hk[2], hc[2], hr1[2], hr2[2];
        f4v logit[1], c1[1], rgb[1];
        mm_layer32<CD, 4, 1>(wl + FW_D1, lane, xd, h);
        mm_pack64<CD, true>(h, hd);
        mm_layer32<CD, 1, 2>(wl + FW_D2, lane, hd, logit);
        mm_layer32<CD, 4, 1>(wl + FW_K1, lane, xc, h);
        mm_pack64<CD, true>(h, hk);
        mm_layer32<CD, 4, 1>(wl + FW_C1A, lane, xc, h);
        mm_pack64<CD, true>(h, hc);
        mm_layer32<CD, 1, 2>(wl + FW_C1B, lane, hc, c1);
        const s4v c1b = mm_round4<CD, false>(c1[0]);
        mm_layer16<CD, 4>(wl + FW_R1, lane, c1b, h);
        mm_pack64<CD, true>(h, hr1);
        SCQ_PACE(4);
        mm_layer32<CD, 4, 2>(wl + FW_R2, lane, hr1, h);
        mm_pack64<CD, true>(h, hr2);
        SCQ_PACE(4);
        mm_layer32<CD, 1, 2>(wl + FW_R3, lane, hr2, rgb);

        // ================= upstream gradients in B-fragment form (row = 4g + e) ===============
        s4v dyd, dyr, dyk;
        {
            float gd[4] = {0.f, 0.f, 0.f, 0.f}, gr[4] = {0.f, 0.f, 0.f, 0.f}, gk[4] = {0.f, 0.f, 0.f, 0.f};
            if (valid) {
                if (g == 0) {
                    // sigma = exp(logit) * density_scale; trunc_exp backward clamps (tcnn_nerf.py:62-66)
                    const float x = logit[0][0];
                    gd[0] = cur_gsig * a.density_scale * expf(fminf(fmaxf(x, -15.0f), 15.0f));
                }
#pragma unroll
                for (int e = 0; e < 4; e++) {
                    const int ch = 4 * g + e;
                    if ((uint32_t)ch < a.C_ch) {
                        const float gv = cur_grgb[e];
                        if (ch < 3) {
                            const float sg = field_sigmoid(rgb[0][e]);
                            gr[e] = gv * sg * (1.0f - sg);
                        } else {
                            gk[e] = gv;
                        }
                    }
                }
            }
#pragma unroll
            for (int e = 0; e < 4; e++) { dyd[e] = MM<CD>::cvt(gd[e]); dyr[e] = MM<CD>::cvt(gr[e]); dyk[e] = MM<CD>::cvt(gk[e]); }
        }

        // ================= color2: 16 -> 64 -> 64 -> 3 =======================================
        s8v g2[2], g1[2];
        s4v gc1;
        {
            mm_layer16<CD, 4>(wt + BW_R3T, lane, dyr, h);
            field_mask_pack<CD>(h, hr2, g2);
            mm_layer32<CD, 4, 2>(wt + BW_R2T, lane, g2, h);
            field_mask_pack<CD>(h, hr1, g1);
            f4v t1[1];
            mm_layer32<CD, 1, 2>(wt + BW_R1T, lane, g1, t1);
            gc1 = mm_round4<CD, false>(t1[0]);
            // wgrads of r3, r2, r1
            s4v hr2t[4], hr1t[4], g2t[4], g1t[4];
            field_tr4<CD>(hr2, ident, hr2t);
            field_tr4<CD>(g2, ident, g2t);
            const s4v dyrt[1] = {mm_transpose16<CD>(dyr, ident)};
            field_wgrad<CD, 1, 4>(w_r3, dyrt, hr2t);
            SCQ_PACE(4);
            field_tr4<CD>(hr1, ident, hr1t);
            field_wgrad<CD, 4, 4>(w_r2, g2t, hr1t);
            SCQ_PACE(4);
            field_tr4<CD>(g1, ident, g1t);
            const s4v c1t[1] = {mm_transpose16<CD>(c1b, ident)};
            field_wgrad<CD, 4, 1>(w_r1, g1t, c1t);
            SCQ_PACE(4);
        }
        // transposed encoder features (shared by the color1 / class / density wgrads)
        s4v xct[2], xdt[2];
        field_tr2<CD>(xc, ident, xct);
        field_tr2<CD>(xd, ident, xdt);

        // ================= color1: 32 -> 64 -> 16, and class: 32 -> 64 -> nc ==================
        f4v gxc[2];
        {
            s8v gh[2];
            s4v ght[4], hct[4];
            mm_layer16<CD, 4>(wt + BW_C1BT, lane, gc1, h);
            field_mask_pack<CD>(h, hc, gh);
            mm_layer32<CD, 2, 2>(wt + BW_C1AT, lane, gh, gxc);
            field_tr4<CD>(hc, ident, hct);
            const s4v gc1t[1] = {mm_transpose16<CD>(gc1, ident)};
            field_wgrad<CD, 1, 4>(w_c1b, gc1t, hct);
            SCQ_PACE(4);
            field_tr4<CD>(gh, ident, ght);
            field_wgrad<CD, 4, 2>(w_c1a, ght, xct);
            SCQ_PACE(4);
        }
        {
            s8v gh[2];
            s4v ght[4], hkt[4];
            mm_layer16<CD, 4>(wt + BW_K2T, lane, dyk, h);
            field_mask_pack<CD>(h, hk, gh);
            mm_layer32_acc<CD, 2, 2>(wt + BW_K1T, lane, gh, gxc);
            field_tr4<CD>(hk, ident, hkt);
            const s4v dykt[1] = {mm_transpose16<CD>(dyk, ident)};
            field_wgrad<CD, 1, 4>(w_k2, dykt, hkt);
            SCQ_PACE(4);
            field_tr4<CD>(gh, ident, ght);
            field_wgrad<CD, 4, 2>(w_k1, ght, xct);
            SCQ_PACE(4);
        }
        // ================= density: 32 -> 64 -> 1 =============================================
        f4v gxd[2];
        {
            s8v gh[2];
            s4v ght[4], hdt[4];
            mm_layer16<CD, 4>(wt + BW_D2T, lane, dyd, h);
            field_mask_pack<CD>(h, hd, gh);
            mm_layer32<CD, 2, 2>(wt + BW_D1T, lane, gh, gxd);
            field_tr4<CD>(hd, ident, hdt);
            const s4v dydt[1] = {mm_transpose16<CD>(dyd, ident)};
            field_wgrad<CD, 1, 4>(w_d2, dydt, hdt);
            SCQ_PACE(4);
            field_tr4<CD>(gh, ident, ght);
            field_wgrad<CD, 4, 2>(w_d1, ght, xdt);
            SCQ_PACE(4);
        }

        // ================= table scatter =======================================================
        // gxd[t][2*(i&1)+f] is d L / d feature f of level lvl[i] (t = i >> 1): same lane<->level map
        // as the forward encode.
        // this tile's scatter: VALU + LDS only (records go to the ring; atomics are issued by the pace
        // points of the NEXT tile's dgrad / wgrad section), then the next tile's loads
        NSR_TICK(tk2);
        NSR_TACC(1, tk1, tk2);
        // Next tile's loads go out BEFORE this tile's scatter: the scatter touches LDS only (its records are
        // turned into atomics by the pace points of the next tile), so by the next loop top both these loads
        // and the atomics issued ahead of them (vmcnt retires in order) have had the whole scatter to land.
        TileIn nxt = cur;
        if (tile + 1 < w_end) nxt = load_tile(tile + 1);
        NSR_TICK(tk3);
        NSR_TACC(3, tk2, tk3);
        if (td || tc) {
            const int lvl[4] = {2 * g, 2 * g + 1, 8 + 2 * g, 9 + 2 * g};
            // x direction of travel of this lane's ray, from its neighbour sample (any value is correct,
            // a consistent one lets field_scatter_level chain the runs of x-adjacent cells)
            const bool swapx = (s == 15 ? u0 - dpp_shr_f<1>(u0) : __builtin_bit_cast(float, dpp_shl1_u(__builtin_bit_cast(uint32_t, u0))) - u0) < 0.f;
            // levels per call (one per lane group): {0,2,4,6} {1,3,5,7} {8,10,12,14} {9,11,13,15}
            const uint32_t call_levels[4] = {0x0055u, 0x00AAu, 0x5500u, 0xAA00u};
#pragma unroll
            for (int i = 0; i < 4; i++) {
                const NsrLevel lv = lds_lv[lvl[i]];
                const int t = i >> 1, e0 = 2 * (i & 1);
                // dead lanes carry zero gradients (see field_scatter_level)
                const float s0 = live ? gxd[t][e0] : 0.f, s1 = live ? gxd[t][e0 + 1] : 0.f;
                const float s2 = live ? gxc[t][e0] : 0.f, s3 = live ? gxc[t][e0 + 1] : 0.f;
#ifdef NSR_ABL_NO_MLP
#define s0 (live ? cur_gsig + 1.f : 0.f)
#define s1 (live ? cur_gsig + 2.f : 0.f)
#define s2 (live ? cur_grgb[0] + 1.f : 0.f)
#define s3 (live ? cur_grgb[1] + 2.f : 0.f)
#endif
                // The first two calls issue no atomics: the prefetch above and the atomics of the dgrad /
                // wgrad pace points land meanwhile.  Then the prefetched registers are touched -- the one
                // place the compiler has to wait for memory (vmcnt(0): gfx9 counts loads and atomics in one
                // counter and may not assume an order between them) -- and the last two calls pace atomics
                // again, which stay in flight across the loop back-edge.
                if (i == 2)
                    asm volatile("" ::"v"(nxt.x0), "v"(nxt.x1), "v"(nxt.x2), "v"(nxt.gsig), "v"(nxt.grgb[0]), "v"(nxt.grgb[1]),
                                 "v"(nxt.grgb[2]), "v"(nxt.grgb[3]), "v"(nxt.xd), "v"(nxt.xc));
                const bool fast = (b.fast_levels & call_levels[i]) == call_levels[i];
                if (i < 2) {
                    if (fast) field_scatter_level<true, 0>(lv, q, gt1, u0, u1, u2, s0, s1, s2, s3, swapx, lane, td, tc);
                    else field_scatter_level<false, 0>(lv, q, gt1, u0, u1, u2, s0, s1, s2, s3, swapx, lane, td, tc);
                } else {
                    if (fast) field_scatter_level<true, SCATTER_PACE>(lv, q, gt1, u0, u1, u2, s0, s1, s2, s3, swapx, lane, td, tc);
                    else field_scatter_level<false, SCATTER_PACE>(lv, q, gt1, u0, u1, u2, s0, s1, s2, s3, swapx, lane, td, tc);
                }
            }
        }
        NSR_TICK(tk4);
        NSR_TACC(2, tk3, tk4);
        cur = nxt;
    }
#ifdef NSR_ABL_STATS
    for (int i = 0; i < 4; i++) NSR_STAT_ALWAYS(4 + i, tacc[i]);
#endif
    if (td || tc) scq_pace(q, gt1, lane, td, tc, 1 << 20, true);

    // ---- flush this wave's weight gradients -------------------------------------------------------
#ifdef NSR_ABL_NO_MLP
    if (false) {
#else
    if (b.grad_mlp) {
#endif
        float *gm = b.grad_mlp;
        field_wgrad_flush<1, 4>(gm + P_R3, 64, 0, 3, w_r3, lane);
        field_wgrad_flush<4, 4>(gm + P_R2, 64, 0, 64, w_r2, lane);
        field_wgrad_flush<4, 1>(gm + P_R1, 16, 0, 64, w_r1, lane);
        field_wgrad_flush<1, 4>(gm + P_C1B, 64, 0, 16, w_c1b, lane);
        field_wgrad_flush<4, 2>(gm + P_C1A, 32, 0, 64, w_c1a, lane);
        field_wgrad_flush<1, 4>(gm + P_K2, 64, CLASS_ROW_SHIFT, nc, w_k2, lane);
        field_wgrad_flush<4, 2>(gm + P_K1, 32, 0, 64, w_k1, lane);
        field_wgrad_flush<1, 4>(gm + P_D2, 64, 0, 1, w_d2, lane);
        field_wgrad_flush<4, 2>(gm + P_D1, 32, 0, 64, w_d1, lane);
    }
}

extern "C" {

int nsr_field_backward(const nsr_field_desc *desc, const void *tables, const float *mlp_params, const float *xyzs, uint32_t M,
                       const int32_t *m_dev, const float *grad_sigmas, const float *grad_rgbs, float *grad_tables,
                       float *grad_mlp, int train_density_table, int train_color_table, const void *feats,
                       nsr_stream_t stream) {
    if (M == 0) return NSR_OK;
    NSR_CHECK_PTR(desc); NSR_CHECK_PTR(tables); NSR_CHECK_PTR(mlp_params); NSR_CHECK_PTR(xyzs);
    NSR_CHECK_PTR(grad_sigmas); NSR_CHECK_PTR(grad_rgbs);
    if ((train_density_table || train_color_table) && grad_tables == nullptr) return NSR_ERR_INVALID_ARG;
    FieldBwdArgs b;
    uint32_t nblocks;
    const int st = field_fill_args(desc, b.f, M, nblocks);
    if (st != NSR_OK) return st;
    if ((uintptr_t)tables & 15u) return NSR_ERR_INVALID_ARG;
    // 4-wave workgroups, one wave per SIMD (each wave needs the 512-register budget): one
    // workgroup per CU is resident, each walks a contiguous range of tiles
    const uint32_t ntiles = (M + 15) / 16;
    nblocks = (ntiles + 3) / 4;
    if (nblocks > 256) nblocks = 256;
    b.f.tiles_per_block = (ntiles + nblocks - 1) / nblocks;
    b.f.tables = tables; b.f.params = mlp_params; b.f.xyzs = xyzs; b.f.m_dev = m_dev; b.f.sigmas = nullptr; b.f.rgbs = nullptr;
    b.f.feats = const_cast<void *>(feats);
    if (feats && ((uintptr_t)feats & 15u)) return NSR_ERR_INVALID_ARG;
    b.grad_sigmas = grad_sigmas; b.grad_rgbs = grad_rgbs; b.grad_tables = grad_tables; b.grad_mlp = grad_mlp;
    b.train_density = train_density_table; b.train_color = train_color_table; b.nc = desc->num_classes;
    b.fast_levels = 0;
    for (uint32_t l = 0; l < desc->L; l++) {
        const NsrLevel &v = b.f.lv[l];
        if (v.use_hash && v.size && (v.size & (v.size - 1)) == 0) b.fast_levels |= 1u << l;
    }
    hipStream_t s = (hipStream_t)stream;
    const dim3 grid(nblocks), block(BWD_THREADS);
#ifdef NSR_ABL_STATS
#define NSR_ABL_REPORT()                                                                                          \
    do {                                                                                                          \
        unsigned long long h[8], z[8] = {0};                                                                      \
        hipDeviceSynchronize();                                                                                   \
        hipMemcpyFromSymbol(h, HIP_SYMBOL(g_stats), sizeof(h));                                                   \
        hipMemcpyToSymbol(HIP_SYMBOL(g_stats), z, sizeof(z));                                                     \
        fprintf(stderr, "[abl] M=%u records=%llu hits=%llu drain_instr=%llu forced=%llu clk wait=%llu mlp=%llu scatter=%llu loads=%llu\n", M, h[0], h[1], h[2], h[3], h[4], h[5], h[6], h[7]); \
    } while (0)
#else
#define NSR_ABL_REPORT() do { } while (0)
#endif
#define NSR_BWD_LAUNCH(TT, CD)                                                                                \
    do {                                                                                                       \
        /* once per process and instantiation (idempotent, so a race is harmless): keeps the call free of   */ \
        /* non-stream API calls, e.g. while the caller captures a hipGraph                                   */ \
        static bool lds_attr_set[64] = {};                                                                     \
        int dev_ = 0;                                                                                          \
        (void)hipGetDevice(&dev_);                                                                             \
        if (!lds_attr_set[dev_ & 63]) {                                                                                 \
            if (hipFuncSetAttribute(reinterpret_cast<const void *>(&k_field_bwd<TT, CD>),                      \
                                    hipFuncAttributeMaxDynamicSharedMemorySize, (int)BWD_LDS_BYTES) != hipSuccess) \
                return NSR_ERR_LAUNCH;                                                                         \
            lds_attr_set[dev_ & 63] = true;                                                                            \
        }                                                                                                      \
        hipLaunchKernelGGL((k_field_bwd<TT, CD>), grid, block, BWD_LDS_BYTES, s, b);                           \
        NSR_ABL_REPORT();                                                                                      \
        return nsr_launch_status();                                                                            \
    } while (0)
    if (desc->table_dtype == NSR_F32 && desc->compute_dtype == NSR_F16) NSR_BWD_LAUNCH(float, NSR_F16);
    if (desc->table_dtype == NSR_F32 && desc->compute_dtype == NSR_BF16) NSR_BWD_LAUNCH(float, NSR_BF16);
    if (desc->table_dtype == NSR_F16 && desc->compute_dtype == NSR_F16) NSR_BWD_LAUNCH(_Float16, NSR_F16);
    if (desc->table_dtype == NSR_F16 && desc->compute_dtype == NSR_BF16) NSR_BWD_LAUNCH(_Float16, NSR_BF16);
#undef NSR_BWD_LAUNCH
    return NSR_ERR_UNSUPPORTED;
}

}   // extern "C"
