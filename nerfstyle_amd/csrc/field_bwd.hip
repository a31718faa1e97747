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
#include "table_scatter.h"

// Schedule knobs, measured on the bench frame (48.6 M samples):
//   NSR_BWD_PKMAX      ReLU on packed halves (v_pk_max_f16): GOUT kernel 13.5 -> 13.0 ms, tracker kernel 49.0 -> 49.5 ms
//   NSR_BWD_NET_ORDER  one net at a time (forward recompute -> dgrad -> wgrad per net): tracker 49.0 -> 48.0 ms, GOUT 13.5 -> 13.4 ms
#ifndef NSR_BWD_PKMAX
#ifdef NSR_BWD_TU_GOUT
#define NSR_BWD_PKMAX true
#else
#define NSR_BWD_PKMAX false
#endif
#endif
#ifndef NSR_BWD_PKMASK
#define NSR_BWD_PKMASK 0           /* ReLU masks of the backward applied to packed 16-bit pairs (GOUT unit) */
#endif
#ifndef NSR_BWD_WQ
#define NSR_BWD_WQ 0               /* straight-order MLP section with queued weight fragments (GOUT unit) */
#endif
#ifndef NSR_BWD_EARLY_NEXT
#define NSR_BWD_EARLY_NEXT 0
#endif
#ifndef NSR_BWD_NET_ORDER
#ifdef NSR_BWD_TU_GOUT
#define NSR_BWD_NET_ORDER 0          /* with PKMAX: 13.0 ms in the straight order, 13.45 ms net by net */
#else
#define NSR_BWD_NET_ORDER 1
#endif
#endif
#ifndef NSR_BWD_ASM_WGRAD
#define NSR_BWD_ASM_WGRAD 0
#endif

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
// per wave: record ring (1024 x {float4 sums, row}) + the [16 levels][16 samples] float4 gradient staging buffer
constexpr size_t BWD_QUEUE_BYTES_PER_WAVE = 1024 * 16 + 1024 * 4 + 16 * 16 * 16;
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
    float4 *gout;              // GOUT kernels: [M][16 levels] float4 = d loss / d (density f0, f1, colour f0, f1) of every sample
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
// The 60 accumulator tiles are pinned to the ACCUMULATOR half of the register file by inline assembly ("+a"), and this file
// is compiled with -mllvm --amdgpu-mfma-vgpr-form: every other MFMA (forward recompute, dgrad, the identity transposes) then
// writes straight to VGPRs.  Left to its heuristics the compiler gives ALL MFMAs of a kernel that needs AGPRs an AGPR
// destination and copies each transient result back (396 v_accvgpr_read per 16-sample tile, 22 % of the loop).
// The s_nop covers the VALU-write -> MFMA-read distance the hazard recogniser cannot see through the asm; the operands
// always come from a VALU rounding step (mm_round4), never directly from another MFMA.
template <int CD>
__device__ __forceinline__ void field_wgrad_mfma(f4v &acc, s4v a, s4v b) {
#if NSR_BWD_ASM_WGRAD
    if (CD == NSR_F16) asm("s_nop 1\n\tv_mfma_f32_16x16x16_f16 %0, %1, %2, %0" : "+a"(acc) : "v"(a), "v"(b));
    else asm("s_nop 1\n\tv_mfma_f32_16x16x16_bf16 %0, %1, %2, %0" : "+a"(acc) : "v"(a), "v"(b));
#else
    acc = MM<CD>::k16(a, b, acc);
#endif
}
template <int CD, int NG, int NA>
__device__ __forceinline__ void field_wgrad(f4v (&acc)[NG * NA], const s4v (&Gt)[NG], const s4v (&At)[NA]) {
#pragma unroll
    for (int ot = 0; ot < NG; ot++) {
#pragma unroll
        for (int it = 0; it < NA; it++) field_wgrad_mfma<CD>(acc[ot * NA + it], Gt[ot], At[it]);
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
#if NSR_BWD_PKMASK
    // Round first, mask afterwards on the packed 16-bit pairs: (act > 0 as a signed 16-bit pattern) -> all-ones / zero by
    // max(act, 0), negate, arithmetic shift -- three packed instructions per PAIR instead of a compare and a select per
    // element, and one packed conversion per pair.  Same bits: a masked element is +0 either way, the others are rounded
    // alike (round to nearest even).  The mask is inline asm: the compiler turns the same arithmetic back into compares and
    // selects.
#pragma unroll
    for (int t = 0; t < 2; t++) {
        const uint4 a4 = __builtin_bit_cast(uint4, act[t]);
        const uint32_t ap[4] = {a4.x, a4.y, a4.z, a4.w};
        uint32_t o[4];
#pragma unroll
        for (int p = 0; p < 4; p++) {                       // pair p of this fragment: elements 2p, 2p + 1
            const f4v &g = gacc[2 * t + (p >> 1)];
            const int e = 2 * (p & 1);
            // (the conversion is left to the compiler: it reads MFMA results, and only the compiler knows how many wait
            // states that read needs -- an inline-asm conversion here returned stale accumulators in one instantiation)
            uint32_t gp, m;
            if (CD == NSR_F16) {
                typedef _Float16 hh2 __attribute__((ext_vector_type(2)));
                const hh2 hp = {(_Float16)g[e], (_Float16)g[e + 1]};
                gp = __builtin_bit_cast(uint32_t, hp);
            } else {
                typedef __bf16 bb2 __attribute__((ext_vector_type(2)));
                const bb2 bp = {(__bf16)g[e], (__bf16)g[e + 1]};
                gp = __builtin_bit_cast(uint32_t, bp);
            }
            asm("v_pk_max_i16 %0, %1, 0\n\tv_pk_sub_i16 %0, 0, %0\n\tv_pk_ashrrev_i16 %0, 15, %0 op_sel_hi:[0,1]" : "=&v"(m) : "v"(ap[p]));
            o[p] = gp & m;
        }
        out[t] = __builtin_bit_cast(s8v, make_uint4(o[0], o[1], o[2], o[3]));
    }
    return;
#endif
    out[0] = mm_cat(mm_round4<CD, false>(mm_relu_mask(gacc[0], mm_lo(act[0]))),
                    mm_round4<CD, false>(mm_relu_mask(gacc[1], mm_hi(act[0]))));
    out[1] = mm_cat(mm_round4<CD, false>(mm_relu_mask(gacc[2], mm_lo(act[1]))),
                    mm_round4<CD, false>(mm_relu_mask(gacc[3], mm_hi(act[1]))));
}

// ---- table scatter ----------------------------------------------------------------------------
// Global float atomics are priced per 64-byte request at the memory side (21 G requests/s chip wide,
// tools/atomic_*_bench.hip), not per byte: 64 lanes adding one dword each to 64 different rows cost 64
// requests.  The naive scatter (4 dword atomics per corner per lane) is 512 requests per sample and ran the
// whole backward at 21 M samples/s.  What is done instead, exact up to fp32 summation order:
//   1. consecutive samples of a ray share table rows (same cell, or the face shared with the next cell) on
//      all but the finest levels: field_scatter_seq keeps every corner's open run in registers and emits one
//      record {row, d0, d1, c0, c1} per finished run (128 corner touches -> ~30 records per sample);
//   2. records go through a per-wave LDS ring and are drained 16 per wave-instruction with 4 lanes per
//      record: the four dwords of an interleaved row leave as ONE 16-byte request, and the two x corners
//      of a lane, adjacent in the ring, usually share a 64-byte line (one request);
//   3. the ring is drained in small paced bursts spread over the NEXT tile's dgrad / wgrad section
//      (SCQ_PACE): atomics are fire-and-forget, but a burst of ~45 back-to-back wave-instructions blocks
//      at issue once the memory side is saturated, and with one wave per SIMD a blocked wave is an idle
//      SIMD; and nothing may wait on vmcnt while fresh atomics are in flight (see the loop comment).
// Ablation builds (never shipped; tools/abl/ + NSR_LIB_PATH): -DNSR_ABL_NO_ATOMIC drops the
// global atomics (everything else runs), -DNSR_ABL_NO_SCATTER the whole scatter, -DNSR_ABL_STATS adds per-phase
// s_memtime sums (+ -DNSR_ABL_COUNTS: record / forced-drain counters) printed after every launch.
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
constexpr int SCQ_CAP = 1024;    // records per wave (power of two)
constexpr int SCQ_MASK = SCQ_CAP - 1;
struct ScatterQueue {
    uint32_t *rows;              // [SCQ_CAP]
    float4 *vals;                // [SCQ_CAP]
    int head, tail;              // wave-uniform, monotonically increasing record indices
};

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
#ifndef NSR_ABL_NO_ATOMIC
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
                                         bool flush) {
    __builtin_amdgcn_wave_barrier();
    const int t = lane >> 2, i = lane & 3;
    const bool on = (i < 2) ? td : tc;
    int full = (q.tail - q.head) >> 4;
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

// ---------------------------------------------------------------------------------------------------------
// Sequential run tracker.  Lane = (level l = lane >> 2,
// y/z corner pair p = lane & 3) owns the two x corners of that pair as two STREAMS A (x0) and B (x0 + 1) and
// walks the tile's 16 samples in order, keeping for each stream the open run {row key, 4 gradient sums} in
// registers -- across tiles too, a wave's tiles being consecutive samples.  A sample that stays in the cell
// adds to both runs; one that moves exactly one cell along one axis hands the still-needed runs over in
// registers (x: between the lane's own two streams; y, z: from the quad neighbour lane p^1 / p^2 by DPP
// quad_perm) -- the decision is geometric (same grid corner => same row), identical in the four lanes of a
// level, so every finished run is emitted exactly once; anything else closes both runs.  Per sample step
// ~180 instructions for 128 corner touches (the first version of this kernel, a DPP segmented scan over the 16
// samples of a level with a hashed "row still in the ring" table, needed ~4000 per tile more).
constexpr int SEQ_K = 4;      // sample steps per ring push
struct SeqState {
    uint32_t c0, c1, c2;     // cell of the previous sample at this lane's level
    uint32_t kA, kB;         // row + 1 of the open runs (0: none)
    float4 aA, aB;           // their gradient sums {d0, d1, c0, c1}
};

__device__ __forceinline__ float4 seq_quad(const float4 &v, bool n1) {
    // value of quad neighbour p^1 (n1) or p^2
    float4 r;
    if (n1) {
        r.x = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v.x), 0xB1, 0xF, 0xF, true));
        r.y = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v.y), 0xB1, 0xF, 0xF, true));
        r.z = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v.z), 0xB1, 0xF, 0xF, true));
        r.w = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v.w), 0xB1, 0xF, 0xF, true));
    } else {
        r.x = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v.x), 0x4E, 0xF, 0xF, true));
        r.y = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v.y), 0x4E, 0xF, 0xF, true));
        r.z = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v.z), 0x4E, 0xF, 0xF, true));
        r.w = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v.w), 0x4E, 0xF, 0xF, true));
    }
    return r;
}

__device__ __forceinline__ float4 seq_sel(bool c, const float4 &a, const float4 &b) {
    return make_float4(c ? a.x : b.x, c ? a.y : b.y, c ? a.z : b.z, c ? a.w : b.w);
}

__device__ __forceinline__ bool seq_nonzero(const float4 &v) {
    return ((__float_as_uint(v.x) | __float_as_uint(v.y) | __float_as_uint(v.z) | __float_as_uint(v.w)) << 1) != 0u;
}

// pushes the (up to NREC, adjacent) records of every lane, lane-major
template <int NREC>
__device__ __forceinline__ void seq_push(ScatterQueue &q, const bool (&p)[NREC], const uint32_t (&k)[NREC], const float4 (&v)[NREC]) {
    int below = 0, total = 0;
#pragma unroll
    for (int r = 0; r < NREC; r++) {
        const unsigned long long m = __ballot(p[r]);
        below += (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
        total += (int)__popcll(m);
    }
    int idx = q.tail + below;
#pragma unroll
    for (int r = 0; r < NREC; r++) {
        if (p[r]) { const int slot = idx & SCQ_MASK; q.rows[slot] = k[r]; q.vals[slot] = v[r]; }
        idx += p[r] ? 1 : 0;
    }
    q.tail += total;
}

// One tile.  G: this wave's [16 levels][16 samples] float4 staging buffer in LDS; (u0,u1,u2): this lane's
// SAMPLE (lane & 15) position, 0 for dead samples; sg[i]: its gradients for level lvl[i] (zero when dead).
__device__ __forceinline__ void field_scatter_seq(SeqState &st, const NsrLevel *__restrict__ lds_lv, float4 *__restrict__ G,
                                                  ScatterQueue &q, float *__restrict__ gt1, float u0, float u1, float u2,
                                                  const float4 (&sg)[4], int lane, bool td, bool tc) {
#ifdef NSR_ABL_NO_SCATTER
    if (lane >= 0) return;
#endif
    const int s = lane & 15, g = lane >> 4;
    const int lvl[4] = {2 * g, 2 * g + 1, 8 + 2 * g, 9 + 2 * g};
#pragma unroll
    for (int i = 0; i < 4; i++) G[lvl[i] * 16 + s] = sg[i];
    __builtin_amdgcn_wave_barrier();
    // this lane's level
    const int l = lane >> 2, py = lane & 1, pz = (lane >> 1) & 1;
    const NsrLevel lv = lds_lv[l];
    const bool hashed = lv.use_hash != 0;
    const uint32_t mulY = hashed ? 2654435761u : lv.mul[1], mulZ = hashed ? 805459861u : lv.mul[2];
    const uint32_t off1 = lv.offset + 1u;
    const float4 zero4 = make_float4(0.f, 0.f, 0.f, 0.f);
    float4 gnext = G[l * 16];
    // The records of SEQ_K consecutive samples are pushed together, lane-major: a level's records of
    // neighbouring samples (x-neighbouring rows, often one 64-byte line) then sit next to each other in the ring
    // and leave in the same atomic instruction (tools/scatter_sim.py: 21.8 -> 19.6 requests/sample for K = 2).
    bool rp[2 * SEQ_K];
    uint32_t rk[2 * SEQ_K];
    float4 rv[2 * SEQ_K];
    for (int step0 = 0; step0 < 16; step0 += SEQ_K) {
#pragma unroll
    for (int sk = 0; sk < SEQ_K; sk++) {
        const int step = step0 + sk;
        const float4 gr = gnext;
        gnext = G[l * 16 + ((step + 1) & 15)];
        const float su0 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, u0), step));
        const float su1 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, u1), step));
        const float su2 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, u2), step));
        float f0, f1, f2;
        uint32_t c0, c1, c2;
        nsr_grid_locate(su0, lv.resolution, 1, f0, c0);
        nsr_grid_locate(su1, lv.resolution, 1, f1, c1);
        nsr_grid_locate(su2, lv.resolution, 1, f2, c2);
        // ---- how did the cell move? (same answer in the 4 lanes of a level) ----
        const int d0 = (int)(c0 - st.c0), d1 = (int)(c1 - st.c1), d2 = (int)(c2 - st.c2);
        const bool same = (d0 | d1 | d2) == 0;
        const bool sx = (d1 | d2) == 0 && (d0 == 1 || d0 == -1);
        const bool sy = (d0 | d2) == 0 && (d1 == 1 || d1 == -1);
        const bool sz = (d0 | d1) == 0 && (d2 == 1 || d2 == -1);
        const bool sxp = sx && d0 == 1, sxm = sx && d0 == -1;
        // y step up: the lanes of the LOW y corner (py = 0) continue the runs their quad neighbour (py = 1) held
        const bool takeY = sy && (py == (d1 > 0 ? 0 : 1)), giveY = sy && !takeY;
        const bool takeZ = sz && (pz == (d2 > 0 ? 0 : 1)), giveZ = sz && !takeZ;
        // ---- records that end here ----
        const bool emitA = !same && !sxm && !giveY && !giveZ && seq_nonzero(st.aA);
        const bool emitB = !same && !sxp && !giveY && !giveZ && seq_nonzero(st.aB);
        rp[2 * sk] = emitA; rk[2 * sk] = st.kA; rv[2 * sk] = st.aA;
        rp[2 * sk + 1] = emitB; rk[2 * sk + 1] = st.kB; rv[2 * sk + 1] = st.aB;
        // ---- runs that continue: where from ----
        const float4 nA = seq_quad(st.aA, true), nB = seq_quad(st.aB, true);
        const float4 mA = seq_quad(st.aA, false), mB = seq_quad(st.aB, false);
        const float4 baseA = seq_sel(same, st.aA, seq_sel(sxp, st.aB, seq_sel(takeY, nA, seq_sel(takeZ, mA, zero4))));
        const float4 baseB = seq_sel(same, st.aB, seq_sel(sxm, st.aA, seq_sel(takeY, nB, seq_sel(takeZ, mB, zero4))));
        // ---- this sample's contribution: (wx*wy)*wz, the product order of the forward interpolation ----
        const float wy = py ? f1 : 1 - f1, wz = pz ? f2 : 1 - f2;
        const float wA = ((1 - f0) * wy) * wz, wB = (f0 * wy) * wz;
        st.aA = make_float4(fmaf(wA, gr.x, baseA.x), fmaf(wA, gr.y, baseA.y), fmaf(wA, gr.z, baseA.z), fmaf(wA, gr.w, baseA.w));
        st.aB = make_float4(fmaf(wB, gr.x, baseB.x), fmaf(wB, gr.y, baseB.y), fmaf(wB, gr.z, baseB.z), fmaf(wB, gr.w, baseB.w));
        // ---- keys of the (possibly unchanged) cell: nsr_grid_row for both x corners ----
        if (!same) {
            const uint32_t ty = (c1 + (uint32_t)py) * mulY, tz = (c2 + (uint32_t)pz) * mulZ;
            const uint32_t comb = hashed ? (ty ^ tz) : (ty + tz);
            const uint32_t iA = hashed ? (c0 ^ comb) : (c0 * lv.mul[0] + comb);
            const uint32_t iB = hashed ? ((c0 + 1u) ^ comb) : ((c0 + 1u) * lv.mul[0] + comb);
            const uint32_t tA = __umulhi(lv.magic, iA), tB = __umulhi(lv.magic, iB);
            const uint32_t qA = (tA + ((iA - tA) >> lv.sh1)) >> lv.sh2, qB = (tB + ((iB - tB) >> lv.sh1)) >> lv.sh2;
            st.kA = off1 + (iA - qA * lv.size);
            st.kB = off1 + (iB - qB * lv.size);
            st.c0 = c0; st.c1 = c1; st.c2 = c2;
        }
    }
        if (q.tail - q.head > SCQ_CAP - 128 * SEQ_K) { NSR_STAT(3, 1); scq_pace(q, gt1, lane, td, tc, 16 * SEQ_K, false); }
        seq_push<2 * SEQ_K>(q, rp, rk, rv);
    }
}

// FEATS: the forward saved the encoder outputs (the default).  Compile-time because the re-gather path, though
// never executed then, costs the one-wave-per-SIMD kernel registers and schedule (measured 20.1 vs 20.4-22 ms).
// GOUT: instead of scattering, the per-level encoder gradients of every sample are written to FieldBwdArgs::gout
// (256 B/sample) for the stand-alone, high-occupancy table scatter (table_scatter.hip) that walks the samples in
// nsr_sample_order's spatial order.  This kernel runs one wave per SIMD (its 240 weight-gradient accumulators): the
// scatter's dependent LDS / atomic chains are exactly what one wave per SIMD cannot hide.
template <typename TT, int CD, bool FEATS, bool GOUT>
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
    q.rows = nullptr; q.vals = nullptr; q.head = q.tail = 0;
    char *qbase_g = nullptr;
    char *const wave_lds = smem + (size_t)(FW_TOTAL + BW_TOTAL) * 2 + 16 * sizeof(NsrLevel);
    if (!GOUT) {
        char *qbase = wave_lds + (size_t)wave * BWD_QUEUE_BYTES_PER_WAVE;
        q.vals = reinterpret_cast<float4 *>(qbase);
        q.rows = reinterpret_cast<uint32_t *>(qbase + SCQ_CAP * 16);
        qbase_g = qbase + SCQ_CAP * 20;
        for (int k = lane; k < SCQ_CAP; k += 64) q.rows[k] = 0u;        // keys are row + 1: 0 matches nothing
    }
    SeqState seq;
    seq.c0 = seq.c1 = seq.c2 = 0x7FFFFFF0u;
    seq.kA = seq.kB = 0u;
    seq.aA = seq.aB = make_float4(0.f, 0.f, 0.f, 0.f);
    float4 *const seqG = reinterpret_cast<float4 *>(qbase_g);        // [16 levels][16 samples] float4 staging, 4 KB
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
    // position `16 * tile + s` of the walk -> index into the sample buffers (a valid one for lanes past the count).  With a
    // permutation (GOUT only: the forward walked the same order, its saved features are tile-major in it) this is a LOAD:
    // the entry of tile t + 2 is requested while tile t runs, so that tile t + 1's loads never wait for their index.
    auto fetch_idx = [&](uint32_t tile) -> uint32_t {
        const uint32_t m = tile * 16 + s;
        if (GOUT && NSR_BWD_EARLY_NEXT && a.perm) return a.perm[min(m, Mc - 1u)];
        return m < Mc ? m : 0u;
    };
    auto load_tile = [&](uint32_t tile, uint32_t buf_idx) {
        TileIn r;
        const size_t mc = buf_idx;
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
        if (FEATS) {
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
    uint32_t idx_cur = 0, idx_next = 0;          // buffer index of this lane's sample in the current / next tile
    if (w_begin < w_end) {
        idx_cur = fetch_idx(w_begin);
        cur = load_tile(w_begin, idx_cur);
#if NSR_BWD_EARLY_NEXT
        idx_next = w_begin + 1 < w_end ? fetch_idx(w_begin + 1) : idx_cur;
#endif
    }
    // GOUT: one tile's per-level encoder gradients, 4 x 16 bytes per lane = 256 contiguous bytes per sample ([16][4] floats)
    float4 gout_v[4];
    uint32_t gout_m = 0;
    bool gout_valid = false;
    auto gout_store = [&]() {
        if (gout_valid) {
#pragma unroll
            for (int i = 0; i < 4; i++) {
                const int lv_i = (i < 2 ? 2 * g : 8 + 2 * g) + (i & 1);
                b.gout[(size_t)gout_m * 16 + lv_i] = gout_v[i];
            }
        }
        gout_valid = false;
    };

#ifdef NSR_ABL_STATS
    unsigned long long tacc[4] = {0, 0, 0, 0};
#define NSR_TICK(var) const unsigned long long var = __builtin_amdgcn_s_memtime()
#define NSR_TACC(i, a, c) tacc[i] += (c) - (a)
#else
#define NSR_TICK(var) do { } while (0)
#define NSR_TACC(i, a, c) do { } while (0)
#endif
#if NSR_BWD_WQ
    // weight-fragment queue of the MLP section (mfma_tiles.h, mm_queue32): holds the next layer's first four fragments
    s8v wq[4];
    s4v wq16[4];
    mm_queue32<4>(wq, wl + FW_D1, lane);
#endif
    for (uint32_t tile = w_begin; tile < w_end; tile++) {
        NSR_TICK(tk0);
#ifdef NSR_ABL_STATS
        __builtin_amdgcn_s_waitcnt(0);
#endif
        NSR_TICK(tk1);
#if !(defined(NSR_ABL_SECT) && NSR_BWD_WQ)
        NSR_TACC(0, tk0, tk1);
#endif
        const uint32_t m = tile * 16 + s;
        const bool valid = m < Mc;
        // GOUT writes no LDS inside this loop, so the compiler would hoist the (loop-invariant) weight-fragment LDS reads
        // out of it and spill them all -- 220 VGPRs to scratch, reloaded every tile.  The barrier keeps them where they are.
        if (GOUT) asm volatile("" ::: "memory");
#if NSR_BWD_EARLY_NEXT
        // Every member of `cur` is made resident HERE, before this tile issues its own stores and loads: left to its first
        // use, a member's wait comes after them and -- one in-order vmcnt, loop-carried -- is emitted as vmcnt(0): the wave
        // then sits out the round trip of the loads it issued a moment ago (seen in the ISA: vmcnt(0) in front of the first
        // MFMA that reads cur.xc).  The loads of `cur` are a whole tile old at this point.
        if (GOUT)
            asm volatile("" :: "v"(cur.x0), "v"(cur.x1), "v"(cur.x2), "v"(cur.gsig), "v"(cur.grgb[0]), "v"(cur.grgb[1]),
                         "v"(cur.grgb[2]), "v"(cur.grgb[3]), "v"(cur.xd), "v"(cur.xc), "v"(idx_next));
#endif
        const float u0 = valid ? field_unit(cur.x0, a.bmin[0], a.bsize[0]) : 0.f;
        const float u1 = valid ? field_unit(cur.x1, a.bmin[1], a.bsize[1]) : 0.f;
        const float u2 = valid ? field_unit(cur.x2, a.bmin[2], a.bsize[2]) : 0.f;
        const bool live = valid && (u0 >= 0 && u0 <= 1 && u1 >= 0 && u1 <= 1 && u2 >= 0 && u2 <= 1);   // NaN -> zeros too
        const float cur_gsig = valid ? cur.gsig : 0.f;
        float cur_grgb[4];
#pragma unroll
        for (int e = 0; e < 4; e++)
            cur_grgb[e] = (valid && (a.C_ch == 8 ? g < 2 : (uint32_t)(4 * g + e) < a.C_ch)) ? cur.grgb[e] : 0.f;
        // no saved features: gather them now (dependent loads, the slow path)
        if (!FEATS) field_encode<TT, CD, false>(lds_lv, tables, u0, u1, u2, live, g, cur.xd, cur.xc, a.fast_levels);

        // paced drain of the previous tile's records: SCQ_PACE(n) issues <= n atomic wave-instructions
#define SCQ_PACE(n) do { if (!GOUT) scq_pace(q, gt1, lane, td, tc, (n), false); } while (0)
#if NSR_BWD_EARLY_NEXT
        // (value-initialised, NOT a copy of cur: copying cur's not-yet-used members here would wait for their loads, and
        // -- one in-order vmcnt -- for the gradient stores issued in between)
        TileIn nxt{};
#endif
#if NSR_BWD_NET_ORDER
        // ================= one net at a time: forward recompute -> dgrad -> wgrad, then its activations are dead ===========
        // (The straight order -- all four forwards, then all backwards -- keeps hd, hk, hc, hr1, hr2 alive together: 40
        // registers that the accumulator-heavy kernel does not have; the compiler then parks MFMA results in AGPRs and
        // copies them back, ~400 v_accvgpr_read per tile.)
        s8v xd[1] = {cur.xd}, xc[1] = {cur.xc};
        f4v h[4];
        s4v xct[2], xdt[2];
        // ---- density: 32 -> 64 -> 1 --------------------------------------------------------------------------------
        f4v gxd[2];
        {
            s8v hd[2];
            f4v logit[1];
            mm_layer32<CD, 4, 1>(wl + FW_D1, lane, xd, h);
            if (GOUT) gout_store();        // the previous tile's encoder gradients: after this tile's inputs have been waited for
            mm_pack64<CD, true, NSR_BWD_PKMAX>(h, hd);
            mm_layer32<CD, 1, 2>(wl + FW_D2, lane, hd, logit);
            s4v dyd;
            {
                // sigma = exp(logit) * density_scale; trunc_exp backward clamps (tcnn_nerf.py:62-66)
                float gd = 0.f;
                if (valid && g == 0) gd = cur_gsig * a.density_scale * expf(fminf(fmaxf(logit[0][0], -15.0f), 15.0f));
                dyd[0] = MM<CD>::cvt(gd); dyd[1] = MM<CD>::cvt(0.f); dyd[2] = dyd[1]; dyd[3] = dyd[1];
            }
            s8v gh[2];
            s4v ght[4], hdt[4];
            mm_layer16<CD, 4>(wt + BW_D2T, lane, dyd, h);
            field_mask_pack<CD>(h, hd, gh);
            mm_layer32<CD, 2, 2>(wt + BW_D1T, lane, gh, gxd);
            SCQ_PACE(4);
            field_tr4<CD>(hd, ident, hdt);
            const s4v dydt[1] = {mm_transpose16<CD>(dyd, ident)};
            field_wgrad<CD, 1, 4>(w_d2, dydt, hdt);
            SCQ_PACE(4);
            field_tr2<CD>(xd, ident, xdt);
            field_tr4<CD>(gh, ident, ght);
            field_wgrad<CD, 4, 2>(w_d1, ght, xdt);
            SCQ_PACE(4);
        }
        field_tr2<CD>(xc, ident, xct);
        // ---- class: 32 -> 64 -> nc (rows 3..) ------------------------------------------------------------------------
        f4v gxc[2];
        {
            s8v hk[2];
            mm_layer32<CD, 4, 1>(wl + FW_K1, lane, xc, h);
            mm_pack64<CD, true, NSR_BWD_PKMAX>(h, hk);
            s4v dyk;
#pragma unroll
            for (int e = 0; e < 4; e++) {
                const int ch = 4 * g + e;
                dyk[e] = MM<CD>::cvt((valid && ch >= 3 && (uint32_t)ch < a.C_ch) ? cur_grgb[e] : 0.f);
            }
            s8v gh[2];
            s4v ght[4], hkt[4];
            mm_layer16<CD, 4>(wt + BW_K2T, lane, dyk, h);
            field_mask_pack<CD>(h, hk, gh);
            mm_layer32<CD, 2, 2>(wt + BW_K1T, lane, gh, gxc);
            SCQ_PACE(4);
            field_tr4<CD>(hk, ident, hkt);
            const s4v dykt[1] = {mm_transpose16<CD>(dyk, ident)};
            field_wgrad<CD, 1, 4>(w_k2, dykt, hkt);
            SCQ_PACE(4);
            field_tr4<CD>(gh, ident, ght);
            field_wgrad<CD, 4, 2>(w_k1, ght, xct);
            SCQ_PACE(4);
        }
        // ---- colour: 32 -> 64 -> 16 -> 64 -> 64 -> 3 (sigmoid) -------------------------------------------------------
        {
            s8v hc[2], hr1[2], hr2[2];
            f4v c1[1], rgb[1];
            mm_layer32<CD, 4, 1>(wl + FW_C1A, lane, xc, h);
            mm_pack64<CD, true, NSR_BWD_PKMAX>(h, hc);
            mm_layer32<CD, 1, 2>(wl + FW_C1B, lane, hc, c1);
            const s4v c1b = mm_round4<CD, false>(c1[0]);
            mm_layer16<CD, 4>(wl + FW_R1, lane, c1b, h);
            mm_pack64<CD, true, NSR_BWD_PKMAX>(h, hr1);
            mm_layer32<CD, 4, 2>(wl + FW_R2, lane, hr1, h);
            mm_pack64<CD, true, NSR_BWD_PKMAX>(h, hr2);
            SCQ_PACE(4);
            mm_layer32<CD, 1, 2>(wl + FW_R3, lane, hr2, rgb);
            s4v dyr;
#pragma unroll
            for (int e = 0; e < 4; e++) {
                const int ch = 4 * g + e;
                float gr = 0.f;
                if (valid && ch < 3) {
                    const float sg = field_sigmoid(rgb[0][e]);
                    gr = cur_grgb[e] * sg * (1.0f - sg);
                }
                dyr[e] = MM<CD>::cvt(gr);
            }
            s8v g2[2], g1[2];
            s4v gc1;
            mm_layer16<CD, 4>(wt + BW_R3T, lane, dyr, h);
            field_mask_pack<CD>(h, hr2, g2);
            mm_layer32<CD, 4, 2>(wt + BW_R2T, lane, g2, h);
            field_mask_pack<CD>(h, hr1, g1);
            f4v t1[1];
            mm_layer32<CD, 1, 2>(wt + BW_R1T, lane, g1, t1);
            gc1 = mm_round4<CD, false>(t1[0]);
            {
                s4v hr2t[4], g2t[4];
                field_tr4<CD>(hr2, ident, hr2t);
                const s4v dyrt[1] = {mm_transpose16<CD>(dyr, ident)};
                field_wgrad<CD, 1, 4>(w_r3, dyrt, hr2t);
                SCQ_PACE(4);
                s4v hr1t[4];
                field_tr4<CD>(g2, ident, g2t);
                field_tr4<CD>(hr1, ident, hr1t);
                field_wgrad<CD, 4, 4>(w_r2, g2t, hr1t);
                SCQ_PACE(4);
            }
            {
                s4v g1t[4];
                field_tr4<CD>(g1, ident, g1t);
                const s4v c1t[1] = {mm_transpose16<CD>(c1b, ident)};
                field_wgrad<CD, 4, 1>(w_r1, g1t, c1t);
                SCQ_PACE(4);
            }
            s8v gh[2];
            s4v ght[4], hct[4];
            mm_layer16<CD, 4>(wt + BW_C1BT, lane, gc1, h);
            field_mask_pack<CD>(h, hc, gh);
            mm_layer32_acc<CD, 2, 2>(wt + BW_C1AT, lane, gh, gxc);
            field_tr4<CD>(hc, ident, hct);
            const s4v gc1t[1] = {mm_transpose16<CD>(gc1, ident)};
            field_wgrad<CD, 1, 4>(w_c1b, gc1t, hct);
            SCQ_PACE(4);
            field_tr4<CD>(gh, ident, ght);
            field_wgrad<CD, 4, 2>(w_c1a, ght, xct);
            SCQ_PACE(4);
        }
#else
#if NSR_BWD_WQ
        // ================= recompute forward, keeping rounded activations ====================
        s8v xd[1] = {cur.xd}, xc[1] = {cur.xc};
        f4v h[4];
        s8v hd[2], hk[2], hc[2], hr1[2], hr2[2];
        f4v logit[1], c1[1], rgb[1];
        mm_layer32_q<CD, 4, 1>(wq, wl + FW_D1, lane, xd, h);          // queued at the end of the previous tile
        mm_queue32<2>(wq, wl + FW_D2, lane);
        if (GOUT) gout_store();            // the previous tile's encoder gradients: after this tile's inputs have been waited for
#if NSR_BWD_EARLY_NEXT
        // GOUT has no scatter between the end of the MLP section and the loop edge: loads issued there are waited for at
        // once (SQ_WAIT_ANY = 51 % of the wave's cycles, profiles/).  The next tile's inputs are requested HERE instead, a
        // whole MLP section ahead, at the price of 16 registers held through it.
        uint32_t idx_nn = idx_next;
        if (GOUT && tile + 1 < w_end) {
            nxt = load_tile(tile + 1, idx_next);
            if (tile + 2 < w_end) idx_nn = fetch_idx(tile + 2);
        }
#endif
        mm_pack64<CD, true, NSR_BWD_PKMAX>(h, hd);
        mm_layer32_q<CD, 1, 2>(wq, wl + FW_D2, lane, hd, logit);
        mm_queue32<4>(wq, wl + FW_K1, lane);
        mm_layer32_q<CD, 4, 1>(wq, wl + FW_K1, lane, xc, h);
        mm_queue32<4>(wq, wl + FW_C1A, lane);
        mm_pack64<CD, true, NSR_BWD_PKMAX>(h, hk);
        mm_layer32_q<CD, 4, 1>(wq, wl + FW_C1A, lane, xc, h);
        mm_queue32<2>(wq, wl + FW_C1B, lane);
        mm_pack64<CD, true, NSR_BWD_PKMAX>(h, hc);
        mm_layer32_q<CD, 1, 2>(wq, wl + FW_C1B, lane, hc, c1);
        mm_queue16<4>(wq16, wl + FW_R1, lane);
        const s4v c1b = mm_round4<CD, false>(c1[0]);
        mm_layer16_q<CD, 4>(wq16, c1b, h);
        mm_queue32<8>(wq, wl + FW_R2, lane);
        mm_pack64<CD, true, NSR_BWD_PKMAX>(h, hr1);
        SCQ_PACE(4);
        mm_layer32_q<CD, 4, 2>(wq, wl + FW_R2, lane, hr1, h);
        mm_queue32<2>(wq, wl + FW_R3, lane);
        mm_pack64<CD, true, NSR_BWD_PKMAX>(h, hr2);
        SCQ_PACE(4);
        mm_layer32_q<CD, 1, 2>(wq, wl + FW_R3, lane, hr2, rgb);
        mm_queue16<4>(wq16, wt + BW_R3T, lane);
#ifdef NSR_ABL_SECT
        NSR_TICK(ts1);
#endif

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
            mm_layer16_q<CD, 4>(wq16, dyr, h);
            mm_queue32<8>(wq, wt + BW_R2T, lane);
            field_mask_pack<CD>(h, hr2, g2);
            mm_layer32_q<CD, 4, 2>(wq, wt + BW_R2T, lane, g2, h);
            mm_queue32<2>(wq, wt + BW_R1T, lane);
            field_mask_pack<CD>(h, hr1, g1);
            f4v t1[1];
            mm_layer32_q<CD, 1, 2>(wq, wt + BW_R1T, lane, g1, t1);
            mm_queue16<4>(wq16, wt + BW_C1BT, lane);
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
#ifdef NSR_ABL_SECT
        NSR_TICK(ts2);
#endif
        // transposed encoder features (shared by the color1 / class / density wgrads)
        s4v xct[2], xdt[2];
        field_tr2<CD>(xc, ident, xct);
        field_tr2<CD>(xd, ident, xdt);

        // ================= color1: 32 -> 64 -> 16, and class: 32 -> 64 -> nc ==================
        f4v gxc[2];
        {
            s8v gh[2];
            s4v ght[4], hct[4];
            mm_layer16_q<CD, 4>(wq16, gc1, h);
            mm_queue32<4>(wq, wt + BW_C1AT, lane);
            field_mask_pack<CD>(h, hc, gh);
            mm_layer32_q<CD, 2, 2>(wq, wt + BW_C1AT, lane, gh, gxc);
            mm_queue16<4>(wq16, wt + BW_K2T, lane);
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
            mm_layer16_q<CD, 4>(wq16, dyk, h);
            mm_queue32<4>(wq, wt + BW_K1T, lane);
            field_mask_pack<CD>(h, hk, gh);
            mm_layer32_q<CD, 2, 2, true>(wq, wt + BW_K1T, lane, gh, gxc);
            mm_queue16<4>(wq16, wt + BW_D2T, lane);
            field_tr4<CD>(hk, ident, hkt);
            const s4v dykt[1] = {mm_transpose16<CD>(dyk, ident)};
            field_wgrad<CD, 1, 4>(w_k2, dykt, hkt);
            SCQ_PACE(4);
            field_tr4<CD>(gh, ident, ght);
            field_wgrad<CD, 4, 2>(w_k1, ght, xct);
            SCQ_PACE(4);
        }
#ifdef NSR_ABL_SECT
        NSR_TICK(ts3);
#endif
        // ================= density: 32 -> 64 -> 1 =============================================
        f4v gxd[2];
        {
            s8v gh[2];
            s4v ght[4], hdt[4];
            mm_layer16_q<CD, 4>(wq16, dyd, h);
            mm_queue32<4>(wq, wt + BW_D1T, lane);
            field_mask_pack<CD>(h, hd, gh);
            mm_layer32_q<CD, 2, 2>(wq, wt + BW_D1T, lane, gh, gxd);
            mm_queue32<4>(wq, wl + FW_D1, lane);          // the next tile's first layer
            field_tr4<CD>(hd, ident, hdt);
            const s4v dydt[1] = {mm_transpose16<CD>(dyd, ident)};
            field_wgrad<CD, 1, 4>(w_d2, dydt, hdt);
            SCQ_PACE(4);
            field_tr4<CD>(gh, ident, ght);
            field_wgrad<CD, 4, 2>(w_d1, ght, xdt);
            SCQ_PACE(4);
        }

#else
        // ================= recompute forward, keeping rounded activations ====================
        s8v xd[1] = {cur.xd}, xc[1] = {cur.xc};
        f4v h[4];
        s8v hd[2], hk[2], hc[2], hr1[2], hr2[2];
        f4v logit[1], c1[1], rgb[1];
        mm_layer32<CD, 4, 1>(wl + FW_D1, lane, xd, h);
        if (GOUT) gout_store();            // the previous tile's encoder gradients: after this tile's inputs have been waited for
#if NSR_BWD_EARLY_NEXT
        // GOUT has no scatter between the end of the MLP section and the loop edge: loads issued there are waited for at
        // once (SQ_WAIT_ANY = 51 % of the wave's cycles, profiles/).  The next tile's inputs are requested HERE instead, a
        // whole MLP section ahead, at the price of 16 registers held through it.
        uint32_t idx_nn = idx_next;
        if (GOUT && tile + 1 < w_end) {
            nxt = load_tile(tile + 1, idx_next);
            if (tile + 2 < w_end) idx_nn = fetch_idx(tile + 2);
        }
#endif
        mm_pack64<CD, true, NSR_BWD_PKMAX>(h, hd);
        mm_layer32<CD, 1, 2>(wl + FW_D2, lane, hd, logit);
        mm_layer32<CD, 4, 1>(wl + FW_K1, lane, xc, h);
        mm_pack64<CD, true, NSR_BWD_PKMAX>(h, hk);
        mm_layer32<CD, 4, 1>(wl + FW_C1A, lane, xc, h);
        mm_pack64<CD, true, NSR_BWD_PKMAX>(h, hc);
        mm_layer32<CD, 1, 2>(wl + FW_C1B, lane, hc, c1);
        const s4v c1b = mm_round4<CD, false>(c1[0]);
        mm_layer16<CD, 4>(wl + FW_R1, lane, c1b, h);
        mm_pack64<CD, true, NSR_BWD_PKMAX>(h, hr1);
        SCQ_PACE(4);
        mm_layer32<CD, 4, 2>(wl + FW_R2, lane, hr1, h);
        mm_pack64<CD, true, NSR_BWD_PKMAX>(h, hr2);
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

#endif
#endif

        // ================= table scatter =======================================================
        // gxd[t][2*(i&1)+f] is d L / d feature f of level lvl[i] (t = i >> 1): same lane<->level map
        // as the forward encode.  The scatter is VALU + LDS only: its records go to the ring and leave as
        // atomics at the pace points of the NEXT tile's dgrad / wgrad section.
        NSR_TICK(tk2);
#if defined(NSR_ABL_SECT) && NSR_BWD_WQ
        NSR_TACC(0, tk1, ts1); NSR_TACC(1, ts1, ts2); NSR_TACC(2, ts2, ts3); NSR_TACC(3, ts3, tk2);
#else
        NSR_TACC(1, tk1, tk2);
#endif
        // Next tile's loads go out BEFORE this tile's scatter: the scatter touches LDS only (its records are
        // turned into atomics by the pace points of the next tile), so by the next loop top both these loads
        // and the atomics issued ahead of them (vmcnt retires in order) have had the whole scatter to land.
#if NSR_BWD_EARLY_NEXT
        if (!GOUT && tile + 1 < w_end) nxt = load_tile(tile + 1, fetch_idx(tile + 1));
#else
        // (the fused tracker unit: exactly this shape -- moving the declaration or the loads costs it several ms)
        TileIn nxt = cur;
        if (tile + 1 < w_end) nxt = load_tile(tile + 1, fetch_idx(tile + 1));
#endif
        NSR_TICK(tk3);
#if !(defined(NSR_ABL_SECT) && NSR_BWD_WQ)
        NSR_TACC(3, tk2, tk3);
#endif
        if (td || tc) {
            float4 sg[4];
#pragma unroll
            for (int i = 0; i < 4; i++) {
                const int t = i >> 1, e0 = 2 * (i & 1);
                sg[i] = live ? make_float4(gxd[t][e0], gxd[t][e0 + 1], gxc[t][e0], gxc[t][e0 + 1]) : make_float4(0.f, 0.f, 0.f, 0.f);
            }
            if (GOUT) {
                // kept in registers over the loop edge and stored early in the NEXT tile (gout_store): a store issued here
                // would be waited for -- one in-order vmcnt -- together with the next tile's loads at the loop top
#pragma unroll
                for (int i = 0; i < 4; i++) gout_v[i] = sg[i];
                gout_m = NSR_BWD_EARLY_NEXT ? idx_cur : m;        // (without the early loads the walk is in buffer order)
                gout_valid = valid;
            }
            else field_scatter_seq(seq, lds_lv, seqG, q, gt1, live ? u0 : 0.f, live ? u1 : 0.f, live ? u2 : 0.f, sg, lane, td, tc);
        }
        NSR_TICK(tk4);
#if !(defined(NSR_ABL_SECT) && NSR_BWD_WQ)
        NSR_TACC(2, tk3, tk4);
#endif
        cur = nxt;
#if NSR_BWD_EARLY_NEXT
        if (GOUT) { idx_cur = idx_next; idx_next = idx_nn; }
#endif
    }
#ifdef NSR_ABL_STATS
    for (int i = 0; i < 4; i++) NSR_STAT_ALWAYS(4 + i, tacc[i]);
#endif
    if (GOUT) gout_store();
    if (!GOUT && (td || tc)) {
        // close the runs still open in registers
        if (q.tail - q.head > SCQ_CAP - 128) scq_pace(q, gt1, lane, td, tc, 16, false);
        const bool fp[2] = {seq_nonzero(seq.aA), seq_nonzero(seq.aB)};
        const uint32_t fk[2] = {seq.kA, seq.kB};
        const float4 fv[2] = {seq.aA, seq.aB};
        seq_push<2>(q, fp, fk, fv);
        scq_pace(q, gt1, lane, td, tc, 1 << 20, true);
    }

    // ---- weight gradients: summed over the workgroup's waves in LDS, then ONE wave adds them to grad_mlp ----------------
    // Every wave holds 60 tiles = 15 360 partial sums.  Flushed wave by wave (rounds 1-2) that is 240 atomic wave-instructions
    // x 4 64-byte requests from each of 1 024 waves onto the same 960 lines -- ~1 M requests at the hot-line rate of the
    // memory-side atomic unit (5.5 G/s, tools/atomic_footprint_bench.hip): 0.18 ms per launch whatever the batch, half of a
    // 4 096-ray step's backward.  The weight-fragment image (61 440 B = exactly 60 tiles x 64 lanes x 16 B) is dead by now and
    // serves as the reduction buffer: four passes of read-add-write, then wave 0 reloads the totals and flushes them --
    // a quarter of the requests, and fp32 sums of four partials instead of four atomics (same value up to rounding order).
    if (b.grad_mlp) {
        f4v *const red = reinterpret_cast<f4v *>(smem);
#define NSR_RED_ALL(OP)                                                                                              \
        OP(w_r3, 0, 4) OP(w_r2, 4, 16) OP(w_r1, 20, 4) OP(w_c1b, 24, 4) OP(w_c1a, 28, 8) OP(w_k2, 36, 4) OP(w_k1, 40, 8)     \
        OP(w_d2, 48, 4) OP(w_d1, 52, 8)
        __syncthreads();                                   // every wave is done with the weight fragments
        for (int w = 0; w < BWD_THREADS / 64; w++) {
            if (wave == w) {
                if (w == 0) {
#define NSR_RED_ST(arr, base, n) _Pragma("unroll") for (int i = 0; i < n; i++) red[((base) + i) * 64 + lane] = arr[i];
                    NSR_RED_ALL(NSR_RED_ST)
#undef NSR_RED_ST
                } else {
#define NSR_RED_ADD(arr, base, n) _Pragma("unroll") for (int i = 0; i < n; i++) red[((base) + i) * 64 + lane] += arr[i];
                    NSR_RED_ALL(NSR_RED_ADD)
#undef NSR_RED_ADD
                }
            }
            __syncthreads();
        }
        if (wave == 0) {
#define NSR_RED_LD(arr, base, n) _Pragma("unroll") for (int i = 0; i < n; i++) arr[i] = red[((base) + i) * 64 + lane];
            NSR_RED_ALL(NSR_RED_LD)
#undef NSR_RED_LD
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
#undef NSR_RED_ALL
    }
}

// ---- launch of one instantiation (LDS attribute set once per process and instantiation: idempotent, so a race is harmless;
// keeps the call free of non-stream API calls, e.g. while the caller captures a hipGraph) ----
#ifdef NSR_ABL_STATS
#define NSR_ABL_REPORT(M)                                                                                          \
    do {                                                                                                          \
        unsigned long long h[8], z[8] = {0};                                                                      \
        hipDeviceSynchronize();                                                                                   \
        hipMemcpyFromSymbol(h, HIP_SYMBOL(g_stats), sizeof(h));                                                   \
        hipMemcpyToSymbol(HIP_SYMBOL(g_stats), z, sizeof(z));                                                     \
        fprintf(stderr, "[abl] M=%u records=%llu hits=%llu drain_instr=%llu forced=%llu clk wait=%llu mlp=%llu scatter=%llu loads=%llu\n", (M), h[0], h[1], h[2], h[3], h[4], h[5], h[6], h[7]); \
    } while (0)
#else
#define NSR_ABL_REPORT(M) do { } while (0)
#endif
template <typename TT, int CD, bool FEATS, bool GOUT>
static int field_bwd_launch_one(const FieldBwdArgs &b, dim3 grid, hipStream_t s) {
    static bool lds_attr_set[64] = {};
    int dev = 0;
    (void)hipGetDevice(&dev);
    if (!lds_attr_set[dev & 63]) {
        if (hipFuncSetAttribute(reinterpret_cast<const void *>(&k_field_bwd<TT, CD, FEATS, GOUT>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)BWD_LDS_BYTES) != hipSuccess)
            return NSR_ERR_LAUNCH;
        lds_attr_set[dev & 63] = true;
    }
    hipLaunchKernelGGL((k_field_bwd<TT, CD, FEATS, GOUT>), grid, dim3(BWD_THREADS), BWD_LDS_BYTES, s, b);
    NSR_ABL_REPORT(b.f.M);
    return nsr_launch_status();
}
template <bool GOUT>
static int field_bwd_launch_variant(const FieldBwdArgs &b, int table_dtype, int compute_dtype, bool feats, dim3 grid, hipStream_t s) {
    if (feats) {                                 // no gather in the kernel: the table type does not matter
        if (compute_dtype == NSR_F16) return field_bwd_launch_one<float, NSR_F16, true, GOUT>(b, grid, s);
        if (compute_dtype == NSR_BF16) return field_bwd_launch_one<float, NSR_BF16, true, GOUT>(b, grid, s);
    } else {
        if (table_dtype == NSR_F32 && compute_dtype == NSR_F16) return field_bwd_launch_one<float, NSR_F16, false, GOUT>(b, grid, s);
        if (table_dtype == NSR_F32 && compute_dtype == NSR_BF16) return field_bwd_launch_one<float, NSR_BF16, false, GOUT>(b, grid, s);
        if (table_dtype == NSR_F16 && compute_dtype == NSR_F16) return field_bwd_launch_one<_Float16, NSR_F16, false, GOUT>(b, grid, s);
        if (table_dtype == NSR_F16 && compute_dtype == NSR_BF16) return field_bwd_launch_one<_Float16, NSR_BF16, false, GOUT>(b, grid, s);
    }
    return NSR_ERR_UNSUPPORTED;
}

// The GOUT instantiations live in their own translation unit (field_bwd_gout.hip includes this file with
// NSR_BWD_TU_GOUT): they are compiled with the accumulators pinned to AGPRs by inline assembly and every other MFMA in
// VGPR form (-DNSR_BWD_ASM_WGRAD=1 -mllvm --amdgpu-mfma-vgpr-form: 1816 -> 1435 instructions per tile, 14.4 -> 13.5 ms);
// the tracker instantiations keep the compiler's own choice -- the same treatment made them slower (49 -> 52 ms: their
// paced atomic drains are tuned to the old schedule).
int nsr_field_bwd_launch_gout(const FieldBwdArgs &b, int table_dtype, int compute_dtype, bool feats, dim3 grid, hipStream_t s);

#ifdef NSR_BWD_TU_GOUT
// Colour-table-only form of the gradients-out backward: the stylisation stage trains `x_color_embedder` alone (trainers/style.py:25),
// so no weight gradient is wanted (grad_mlp == NULL) and nothing behind the density output either.  What is left of the chain is
// the forward recompute of the class and colour nets (for the ReLU masks) and their input gradients -- 60 of the full kernel's
// ~190 MFMAs, none of its 240 accumulators, so the kernel runs at four waves per SIMD (512-thread workgroups around one 60 KB weight image) instead of one
// (1008x756 stylisation iteration, 24 patches on four streams: 41.1 -> 39.4 ms).  Same helper calls in the same order as the full kernel: the
// colour gradients are bit-identical to its.  gout's density components are written as zeros (the scatter ignores them).
constexpr int COLOR_THREADS = 512;       // 8 waves share one 60 KB weight image: two workgroups per CU = four waves per SIMD (90 registers)
template <int CD>
__global__ void __launch_bounds__(COLOR_THREADS)
k_field_bwd_color(FieldBwdArgs b) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    short *wl = reinterpret_cast<short *>(smem);
    short *wt = wl + FW_TOTAL;
    const FieldArgs &a = b.f;
    field_build_fw<CD, false>(wl, a.params);
    field_build_bw<CD>(wt, a.params);
    __syncthreads();
    const uint32_t Mc = a.m_dev ? min((uint32_t)max(a.m_dev[0], 0), a.M) : a.M;
    const uint32_t ntiles = (Mc + 15) / 16;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int s = lane & 15, g = lane >> 4;
    const uint32_t lb = field_logical_block();
    const uint32_t tpb = (ntiles + gridDim.x - 1) / gridDim.x;
    const uint32_t t_begin = lb * tpb;
    const uint32_t t_end = min(t_begin + tpb, ntiles);
    for (uint32_t tile = t_begin + wave; tile < t_end; tile += COLOR_THREADS / 64) {
        const uint32_t mpos = tile * 16 + s;
        const bool valid = mpos < Mc;
        const uint32_t m = a.perm[min(mpos, Mc - 1u)];
        const float u0 = field_unit(a.xyzs[(size_t)m * 3 + 0], a.bmin[0], a.bsize[0]);
        const float u1 = field_unit(a.xyzs[(size_t)m * 3 + 1], a.bmin[1], a.bsize[1]);
        const float u2 = field_unit(a.xyzs[(size_t)m * 3 + 2], a.bmin[2], a.bsize[2]);
        const bool live = valid && (u0 >= 0 && u0 <= 1 && u1 >= 0 && u1 <= 1 && u2 >= 0 && u2 <= 1);
        float grgb[4];
        {
            const float *gp = b.grad_rgbs + (size_t)m * a.C_ch;
            if (a.C_ch == 8) {
                const float4 t4 = reinterpret_cast<const float4 *>(gp)[g & 1];
                grgb[0] = t4.x; grgb[1] = t4.y; grgb[2] = t4.z; grgb[3] = t4.w;
            } else {
#pragma unroll
                for (int e = 0; e < 4; e++) grgb[e] = gp[(uint32_t)(4 * g + e) < a.C_ch ? 4 * g + e : 0];
            }
#pragma unroll
            for (int e = 0; e < 4; e++)
                if (!(valid && (a.C_ch == 8 ? g < 2 : (uint32_t)(4 * g + e) < a.C_ch))) grgb[e] = 0.f;
        }
        const s8v xc[1] = {(reinterpret_cast<const s8v *>(a.feats) + ((size_t)tile * 64 + lane) * 2)[1]};
        // ---- forward recompute (rounded activations) ----
        f4v h[4];
        s8v hk[2], hc[2], hr1[2], hr2[2];
        f4v c1[1], rgb[1];
        mm_layer32<CD, 4, 1>(wl + FW_K1, lane, xc, h);
        mm_pack64<CD, true, NSR_BWD_PKMAX>(h, hk);
        mm_layer32<CD, 4, 1>(wl + FW_C1A, lane, xc, h);
        mm_pack64<CD, true, NSR_BWD_PKMAX>(h, hc);
        mm_layer32<CD, 1, 2>(wl + FW_C1B, lane, hc, c1);
        const s4v c1b = mm_round4<CD, false>(c1[0]);
        mm_layer16<CD, 4>(wl + FW_R1, lane, c1b, h);
        mm_pack64<CD, true, NSR_BWD_PKMAX>(h, hr1);
        mm_layer32<CD, 4, 2>(wl + FW_R2, lane, hr1, h);
        mm_pack64<CD, true, NSR_BWD_PKMAX>(h, hr2);
        mm_layer32<CD, 1, 2>(wl + FW_R3, lane, hr2, rgb);
        // ---- upstream gradients in B-fragment form (row = 4g + e) ----
        s4v dyr, dyk;
#pragma unroll
        for (int e = 0; e < 4; e++) {
            const int ch = 4 * g + e;
            float gr = 0.f, gk = 0.f;
            if (valid && (uint32_t)ch < a.C_ch) {
                if (ch < 3) {
                    const float sg = field_sigmoid(rgb[0][e]);
                    gr = grgb[e] * sg * (1.0f - sg);
                } else {
                    gk = grgb[e];
                }
            }
            dyr[e] = MM<CD>::cvt(gr);
            dyk[e] = MM<CD>::cvt(gk);
        }
        // ---- colour-2, colour-1, class: input gradients ----
        s8v g2[2], g1[2], gh[2];
        f4v t1[1], gxc[2];
        mm_layer16<CD, 4>(wt + BW_R3T, lane, dyr, h);
        field_mask_pack<CD>(h, hr2, g2);
        mm_layer32<CD, 4, 2>(wt + BW_R2T, lane, g2, h);
        field_mask_pack<CD>(h, hr1, g1);
        mm_layer32<CD, 1, 2>(wt + BW_R1T, lane, g1, t1);
        const s4v gc1 = mm_round4<CD, false>(t1[0]);
        mm_layer16<CD, 4>(wt + BW_C1BT, lane, gc1, h);
        field_mask_pack<CD>(h, hc, gh);
        mm_layer32<CD, 2, 2>(wt + BW_C1AT, lane, gh, gxc);
        mm_layer16<CD, 4>(wt + BW_K2T, lane, dyk, h);
        field_mask_pack<CD>(h, hk, gh);
        mm_layer32_acc<CD, 2, 2>(wt + BW_K1T, lane, gh, gxc);
        if (valid) {
#pragma unroll
            for (int i = 0; i < 4; i++) {
                const int t = i >> 1, e0 = 2 * (i & 1);
                const int lv_i = (i < 2 ? 2 * g : 8 + 2 * g) + (i & 1);
                b.gout[(size_t)m * 16 + lv_i] = live ? make_float4(0.f, 0.f, gxc[t][e0], gxc[t][e0 + 1]) : make_float4(0.f, 0.f, 0.f, 0.f);
            }
        }
    }
}
template <int CD>
static int field_bwd_launch_color(const FieldBwdArgs &b, hipStream_t s) {
    const size_t lds = (size_t)(FW_TOTAL + BW_TOTAL) * 2;
    static bool attr_set[64] = {};
    int dev = 0;
    (void)hipGetDevice(&dev);
    if (!attr_set[dev & 63]) {
        if (hipFuncSetAttribute(reinterpret_cast<const void *>(&k_field_bwd_color<CD>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                (int)lds) != hipSuccess)
            return NSR_ERR_LAUNCH;
        attr_set[dev & 63] = true;
    }
    // two resident workgroups per CU; >= 8 tiles per wave so that the weight-image build amortises
    const uint32_t ntiles = (b.f.M + 15) / 16;
    uint32_t nb = (ntiles + 63) / 64;
    if (nb > 512) nb = 512;
    if (nb == 0) nb = 1;
    hipLaunchKernelGGL((k_field_bwd_color<CD>), dim3(nb), dim3(COLOR_THREADS), lds, s, b);
    return nsr_launch_status();
}

int nsr_field_bwd_launch_gout(const FieldBwdArgs &b, int table_dtype, int compute_dtype, bool feats, dim3 grid, hipStream_t s) {
#if !NSR_BWD_EARLY_NEXT
    if (b.f.perm != nullptr) return NSR_ERR_UNSUPPORTED;      // an ablation build without the index prefetch cannot walk a permutation
#endif
    static const int color_env = [] { const char *e = getenv("NSR_BWD_COLOR_ONLY"); return e ? atoi(e) : 1; }();
    if (color_env && b.grad_mlp == nullptr && !b.train_density && b.train_color && feats && b.f.perm != nullptr) {
        if (compute_dtype == NSR_F16) return field_bwd_launch_color<NSR_F16>(b, s);
        if (compute_dtype == NSR_BF16) return field_bwd_launch_color<NSR_BF16>(b, s);
    }
    return field_bwd_launch_variant<true>(b, table_dtype, compute_dtype, feats, grid, s);
}
#else
extern "C" {

uint64_t nsr_field_backward_workspace_bytes(uint32_t M, int with_perm) {
    return with_perm ? (uint64_t)M * 16 * sizeof(float4) : 0;        // the [M][16] float4 encoder-gradient buffer
}

int nsr_field_backward(const nsr_field_desc *desc, const void *tables, const float *mlp_params, const float *xyzs, uint32_t M,
                       const int32_t *m_dev, const float *grad_sigmas, const float *grad_rgbs, float *grad_tables,
                       float *grad_mlp, int train_density_table, int train_color_table, const void *feats,
                       const uint32_t *perm, void *workspace, nsr_stream_t stream) {
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
    const bool gout = perm != nullptr && (train_density_table || train_color_table);
    // `feats` given together with `perm` were written by nsr_field_forward in perm's order (tile-major): the gradients-out
    // kernel walks the same order; the fused tracker kernel walks the buffers and cannot use them (it re-gathers)
    if (perm != nullptr && !gout) feats = nullptr;
    b.f.feats = const_cast<void *>(feats);
    b.f.perm = (gout && feats != nullptr) ? perm : nullptr;      // (the gradients-out unit is built with NSR_BWD_EARLY_NEXT)
    b.gout = nullptr;
    if (gout) {
        if (workspace == nullptr || ((uintptr_t)workspace & 15u)) return NSR_ERR_INVALID_ARG;
        if (!nsr_table_scatter_supported(b.f.lv)) return NSR_ERR_UNSUPPORTED;   // grid too fine for the 10-bit blocks: call without perm
        b.gout = (float4 *)workspace;
    }
    if (feats && ((uintptr_t)feats & 15u)) return NSR_ERR_INVALID_ARG;
    b.grad_sigmas = grad_sigmas; b.grad_rgbs = grad_rgbs; b.grad_tables = grad_tables; b.grad_mlp = grad_mlp;
    b.train_density = train_density_table; b.train_color = train_color_table; b.nc = desc->num_classes;
    hipStream_t s = (hipStream_t)stream;
    const dim3 grid(nblocks);
    if (!gout) return field_bwd_launch_variant<false>(b, desc->table_dtype, desc->compute_dtype, feats != nullptr, grid, s);
    const int st1 = nsr_field_bwd_launch_gout(b, desc->table_dtype, desc->compute_dtype, feats != nullptr, grid, s);
    if (st1 != NSR_OK) return st1;
    // second kernel: the table scatter in the permutation's order, many waves per CU
    return nsr_table_scatter_launch(b.f.lv, b.f.bmin, b.f.bsize, xyzs, perm, m_dev, M, workspace, grad_tables, train_density_table,
                                    train_color_table, s);
}

}   // extern "C"
#endif
