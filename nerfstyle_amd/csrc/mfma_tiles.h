// MFMA building blocks shared by the fused field kernels (field.hip) and the stand-alone MLP
// kernels (mlp.hip).  gfx950 only: v_mfma_f32_16x16x32_{f16,bf16} (+ the 16x16x16 forms for
// K = 16 products and register transposes).
//
// Orientation used everywhere: weights are the A operand, activations the B operand,
//     Y^T[out x samples] = W[out x in] * X^T[in x samples],
// so the 16 samples of a tile sit on lane&15 and never move: a layer's accumulator tile
// (row = 4*(lane>>4) + reg, col = lane&15) is, after activation + rounding, directly the B
// fragment of the next layer's 16-wide k-block (k = 4*(lane>>4) + elem) -- the whole MLP chain
// runs in registers with no LDS round trip and no cross-lane traffic.  Two 16-wide k-blocks are
// concatenated into one 8-element fragment for the K = 32 instruction; the A fragments in LDS are
// packed with the same (k-block, 4*(lane>>4)+elem) order, so no permutation is ever visible.
#pragma once
#include "nsr_common.h"

typedef short s4v __attribute__((ext_vector_type(4)));
typedef short s8v __attribute__((ext_vector_type(8)));
typedef _Float16 h4v __attribute__((ext_vector_type(4)));
typedef _Float16 h8v __attribute__((ext_vector_type(8)));
typedef __bf16 b8v __attribute__((ext_vector_type(8)));
typedef float f4v __attribute__((ext_vector_type(4)));

template <int CD> struct MM;
template <> struct MM<NSR_F16> {
    static __device__ __forceinline__ f4v k32(s8v a, s8v b, f4v c) {
        return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(h8v, a), __builtin_bit_cast(h8v, b), c, 0, 0, 0);
    }
    static __device__ __forceinline__ f4v k16(s4v a, s4v b, f4v c) {
        return __builtin_amdgcn_mfma_f32_16x16x16f16(__builtin_bit_cast(h4v, a), __builtin_bit_cast(h4v, b), c, 0, 0, 0);
    }
    static __device__ __forceinline__ short cvt(float x) { return __builtin_bit_cast(short, (_Float16)x); }
    static __device__ __forceinline__ float up(short v) { return (float)__builtin_bit_cast(_Float16, v); }
    static __device__ __forceinline__ short one() { return (short)0x3C00; }
};
template <> struct MM<NSR_BF16> {
    static __device__ __forceinline__ f4v k32(s8v a, s8v b, f4v c) {
        return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(b8v, a), __builtin_bit_cast(b8v, b), c, 0, 0, 0);
    }
    static __device__ __forceinline__ f4v k16(s4v a, s4v b, f4v c) {
        return __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(a, b, c, 0, 0, 0);
    }
    static __device__ __forceinline__ short cvt(float x) { return __builtin_bit_cast(short, (__bf16)x); }
    static __device__ __forceinline__ float up(short v) {
        return __builtin_bit_cast(float, ((uint32_t)(uint16_t)v) << 16);
    }
    static __device__ __forceinline__ short one() { return (short)0x3F80; }
};

// accumulator tile -> 4 rounded elements (optionally ReLU first)
template <int CD, bool RELU, bool PKMAX = false>
__device__ __forceinline__ s4v mm_round4(f4v a) {
    if (CD == NSR_F16 && RELU && PKMAX) {
        // round first, then ReLU on the packed halves (v_pk_max_f16: two elements per instruction).  Rounding to
        // nearest is monotonic and keeps the sign, so max(round(x), 0) == round(max(x, 0)) bit for bit.  Opt-in:
        // it pays in the VALU-issue-bound forward kernel, the register-starved backward schedules worse with it.
        const h4v h = {(_Float16)a[0], (_Float16)a[1], (_Float16)a[2], (_Float16)a[3]};
        const h4v z = {(_Float16)0, (_Float16)0, (_Float16)0, (_Float16)0};
        return __builtin_bit_cast(s4v, __builtin_elementwise_max(h, z));
    }
    s4v r;
#pragma unroll
    for (int i = 0; i < 4; i++) r[i] = MM<CD>::cvt(RELU ? fmaxf(a[i], 0.0f) : a[i]);
    return r;
}
__device__ __forceinline__ s8v mm_cat(s4v lo, s4v hi) {
    s8v r;
    r[0] = lo[0]; r[1] = lo[1]; r[2] = lo[2]; r[3] = lo[3];
    r[4] = hi[0]; r[5] = hi[1]; r[6] = hi[2]; r[7] = hi[3];
    return r;
}
__device__ __forceinline__ s4v mm_lo(s8v v) { s4v r; r[0] = v[0]; r[1] = v[1]; r[2] = v[2]; r[3] = v[3]; return r; }
__device__ __forceinline__ s4v mm_hi(s8v v) { s4v r; r[0] = v[4]; r[1] = v[5]; r[2] = v[6]; r[3] = v[7]; return r; }

// Zero the elements of a gradient tile whose forward activation (post-ReLU, rounded) was not > 0.
// Positive non-zero half/bfloat16 <=> the 16-bit pattern is > 0 as a signed integer.
__device__ __forceinline__ f4v mm_relu_mask(f4v g, s4v act) {
    f4v r;
#pragma unroll
    for (int i = 0; i < 4; i++) r[i] = act[i] > 0 ? g[i] : 0.0f;
    return r;
}

// ---- LDS fragment images -------------------------------------------------------------------
// A "frag32" holds a 16-row x 32-k slab of an effective A matrix for one K=32 MFMA: 64 lanes x
// 8 elements (1 KiB), lane-linear => conflict-free ds_read_b128.  A "frag16" is the 16 x 16
// form (64 x 4 elements, 512 B, ds_read_b64).
//
// The effective matrix is A_eff[row][k] = W[row - row_shift][k]            (transposed == 0)
//                                       = W[k - row_shift][row]            (transposed == 1)
// with W row-major [wrows x wcols] fp32 (the master parameters), zero outside W.
// All threads of the block cooperate; the caller synchronises afterwards.
template <int CD>
__device__ __forceinline__ void mm_build_frags(short *lds, const float *__restrict__ W, int wrows, int wcols, int m_tiles,
                                               int K, bool transposed, int row_shift, bool k32) {
    const int E = k32 ? 8 : 4;
    const int nk = K / (k32 ? 32 : 16);
    const int total = m_tiles * nk * 64 * E;
    for (int idx = threadIdx.x; idx < total; idx += blockDim.x) {
        const int e = idx % E;
        const int lane = (idx / E) & 63;
        const int frag = idx / (E * 64);
        const int m = frag / nk, kb = frag % nk;
        const int g = lane >> 4, r = lane & 15;
        const int row = 16 * m + r;
        const int k = k32 ? (32 * kb + 16 * (e >> 2) + 4 * g + (e & 3)) : (16 * kb + 4 * g + e);
        const int wr = transposed ? (k - row_shift) : (row - row_shift);
        const int wc = transposed ? row : k;
        float v = 0.0f;
        if (wr >= 0 && wr < wrows && wc >= 0 && wc < wcols) v = W[wr * wcols + wc];
        lds[idx] = MM<CD>::cvt(v);
    }
}

// Weight fragments are read from LDS right where they are used.  In a kernel that runs ONE wave per SIMD at the register
// limit (the field backward) the compiler leaves ~3 instructions between a ds_read and the MFMA that needs it, and nothing
// else can run while the wave waits: 45 exposed LDS round trips per 16-sample tile, about half of the kernel's cycles.
// NSR_MM_AHEAD = n keeps n fragment reads in flight ahead of the MFMA stream (a register ring refilled right after the
// MFMA that consumed the slot, pinned by scheduling barriers); 0 = read at the point of use.
#ifndef NSR_MM_AHEAD
#define NSR_MM_AHEAD 0
#endif

// One layer: acc[m] = sum_u A[m][u] * b[u]  (K = 32 per step)
template <int CD, int MT, int KP>
__device__ __forceinline__ void mm_layer32(const short *lds_frags, int lane, const s8v (&b)[KP], f4v (&acc)[MT]) {
    const s8v *F = reinterpret_cast<const s8v *>(lds_frags);
#if NSR_MM_AHEAD
    constexpr int N = MT * KP, A = NSR_MM_AHEAD < N ? NSR_MM_AHEAD : N;
    s8v w[A];
#pragma unroll
    for (int i = 0; i < A; i++) w[i] = F[i * 64 + lane];
    __builtin_amdgcn_sched_barrier(0);
    f4v a = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int i = 0; i < N; i++) {
        const int m = i / KP, u = i % KP;
        if (u == 0) a = f4v{0.f, 0.f, 0.f, 0.f};
        a = MM<CD>::k32(w[i % A], b[u], a);
        if (i + A < N) w[i % A] = F[(i + A) * 64 + lane];
        if (u == KP - 1) acc[m] = a;
        __builtin_amdgcn_sched_barrier(0);
    }
#else
#pragma unroll
    for (int m = 0; m < MT; m++) {
        f4v a = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int u = 0; u < KP; u++) a = MM<CD>::k32(F[(m * KP + u) * 64 + lane], b[u], a);
        acc[m] = a;
    }
#endif
}
// Same, accumulating into acc.
template <int CD, int MT, int KP>
__device__ __forceinline__ void mm_layer32_acc(const short *lds_frags, int lane, const s8v (&b)[KP], f4v (&acc)[MT]) {
    const s8v *F = reinterpret_cast<const s8v *>(lds_frags);
#if NSR_MM_AHEAD
    constexpr int N = MT * KP, A = NSR_MM_AHEAD < N ? NSR_MM_AHEAD : N;
    s8v w[A];
#pragma unroll
    for (int i = 0; i < A; i++) w[i] = F[i * 64 + lane];
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int i = 0; i < N; i++) {
        const int m = i / KP, u = i % KP;
        acc[m] = MM<CD>::k32(w[i % A], b[u], acc[m]);
        if (i + A < N) w[i % A] = F[(i + A) * 64 + lane];
        __builtin_amdgcn_sched_barrier(0);
    }
#else
#pragma unroll
    for (int m = 0; m < MT; m++) {
#pragma unroll
        for (int u = 0; u < KP; u++) acc[m] = MM<CD>::k32(F[(m * KP + u) * 64 + lane], b[u], acc[m]);
    }
#endif
}
// One layer with a single 16-wide k-block.
template <int CD, int MT>
__device__ __forceinline__ void mm_layer16(const short *lds_frags, int lane, s4v b, f4v (&acc)[MT]) {
    const s4v *F = reinterpret_cast<const s4v *>(lds_frags);
#if NSR_MM_AHEAD
    s4v w[MT];                       // 8-byte fragments: all of the layer's reads go out together
#pragma unroll
    for (int m = 0; m < MT; m++) w[m] = F[m * 64 + lane];
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int m = 0; m < MT; m++) {
        f4v a = {0.f, 0.f, 0.f, 0.f};
        acc[m] = MM<CD>::k16(w[m], b, a);
    }
#else
#pragma unroll
    for (int m = 0; m < MT; m++) {
        f4v a = {0.f, 0.f, 0.f, 0.f};
        acc[m] = MM<CD>::k16(F[m * 64 + lane], b, a);
    }
#endif
}

// ---- layers whose first weight fragments were requested earlier ("queued") ---------------------------------------------
// mm_queue32 / mm_queue16 issue the first min(N, 4) fragment reads of a layer into a caller-held register queue; the
// caller places them BEFORE the previous layer's epilogue (conversion / packing VALU work), so that the LDS round trip
// runs under it.  The *_q layer forms consume the queue and read any further fragments four ahead, as NSR_MM_AHEAD does.
template <int N>
__device__ __forceinline__ void mm_queue32(s8v (&wq)[4], const short *lds_frags, int lane) {
    const s8v *F = reinterpret_cast<const s8v *>(lds_frags);
#pragma unroll
    for (int i = 0; i < (N < 4 ? N : 4); i++) wq[i] = F[i * 64 + lane];
    __builtin_amdgcn_sched_barrier(0);
}
template <int N>
__device__ __forceinline__ void mm_queue16(s4v (&wq)[4], const short *lds_frags, int lane) {
    const s4v *F = reinterpret_cast<const s4v *>(lds_frags);
#pragma unroll
    for (int i = 0; i < (N < 4 ? N : 4); i++) wq[i] = F[i * 64 + lane];
    __builtin_amdgcn_sched_barrier(0);
}
template <int CD, int MT, int KP, bool ACC = false>
__device__ __forceinline__ void mm_layer32_q(s8v (&wq)[4], const short *lds_frags, int lane, const s8v (&b)[KP], f4v (&acc)[MT]) {
    const s8v *F = reinterpret_cast<const s8v *>(lds_frags);
    constexpr int N = MT * KP, A = N < 4 ? N : 4;
    __builtin_amdgcn_sched_barrier(0);
    f4v a = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int i = 0; i < N; i++) {
        const int m = i / KP, u = i % KP;
        if (u == 0) a = ACC ? acc[m] : f4v{0.f, 0.f, 0.f, 0.f};
        a = MM<CD>::k32(wq[i % A], b[u], a);
        if (i + A < N) wq[i % A] = F[(i + A) * 64 + lane];
        if (u == KP - 1) acc[m] = a;
        __builtin_amdgcn_sched_barrier(0);
    }
}
template <int CD, int MT>
__device__ __forceinline__ void mm_layer16_q(s4v (&wq)[4], s4v b, f4v (&acc)[MT]) {
    static_assert(MT <= 4, "queue holds four fragments");
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int m = 0; m < MT; m++) {
        f4v a = {0.f, 0.f, 0.f, 0.f};
        acc[m] = MM<CD>::k16(wq[m], b, a);
    }
    __builtin_amdgcn_sched_barrier(0);
}

// 4 accumulator tiles (64 rows) -> two K=32 B fragments, with or without ReLU
template <int CD, bool RELU, bool PKMAX = false>
__device__ __forceinline__ void mm_pack64(const f4v (&acc)[4], s8v (&out)[2]) {
    out[0] = mm_cat(mm_round4<CD, RELU, PKMAX>(acc[0]), mm_round4<CD, RELU, PKMAX>(acc[1]));
    out[1] = mm_cat(mm_round4<CD, RELU, PKMAX>(acc[2]), mm_round4<CD, RELU, PKMAX>(acc[3]));
}

// Register transpose of a 16x16 block by one MFMA with the identity:
//   in : lane (s = lane&15, g) holds Z[4g+e][s]      (feature rows on elements, sample on lane)
//   out: lane (f = lane&15, g) holds Z[f][4g+e]      (feature on lane, samples on elements)
// Exact: every product is z*1 or z*0, accumulated in fp32, re-rounded to the 16-bit type it had.
template <int CD>
__device__ __forceinline__ s4v mm_transpose16(s4v z, s4v ident) {
    f4v zero = {0.f, 0.f, 0.f, 0.f};
    const f4v t = MM<CD>::k16(z, ident, zero);
    return mm_round4<CD, false>(t);
}
template <int CD>
__device__ __forceinline__ s4v mm_identity_frag(int lane) {
    s4v r;
    const int c = lane & 15, g = lane >> 4;
#pragma unroll
    for (int e = 0; e < 4; e++) r[e] = (4 * g + e == c) ? MM<CD>::one() : (short)0;
    return r;
}
