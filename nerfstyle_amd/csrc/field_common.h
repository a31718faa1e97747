// Shared pieces of the fused field kernels (forward: field.hip, backward: field_bwd.hip).
#pragma once
#include <hip/hip_fp16.h>

#include "mfma_tiles.h"

// ---- parameter layout (floats) inside mlp_params: density, color1, color2, class ------------
constexpr int P_D1 = 0;        // 64 x 32
constexpr int P_D2 = 2048;     // 16 x 64 (row 0 = logit)
constexpr int P_C1A = 3072;    // 64 x 32
constexpr int P_C1B = 5120;    // 16 x 64
constexpr int P_R1 = 6144;     // 64 x 16
constexpr int P_R2 = 7168;     // 64 x 64
constexpr int P_R3 = 11264;    // 16 x 64 (rows 0..2 = rgb)
constexpr int P_K1 = 12288;    // 64 x 32
constexpr int P_K2 = 14336;    // 16 x 64 (rows 0..nc-1 = classes)
constexpr int P_TOTAL = 15360;

// ---- forward LDS image (units: shorts).  frag32 = 512 shorts, frag16 = 256 shorts ----------
constexpr int FW_D1 = 0, FW_D2 = 2048, FW_C1A = 3072, FW_C1B = 5120, FW_K1 = 6144, FW_K2 = 8192, FW_R2 = 9216,
              FW_R3 = 13312, FW_R1 = 14336, FW_TOTAL = 15360, FW_SIGMA_TOTAL = 3072;
// class logits live in rows 3..3+nc-1 of their output tile so that (row == output channel) and
// lane (s,g) stores channels 4g..4g+3 of rgbs[m, :] as one 16-byte piece.
constexpr int CLASS_ROW_SHIFT = 3;

struct FieldArgs {
    const void *tables;
    const float *params;
    const float *xyzs;
    const int32_t *m_dev;
    uint32_t M;
    float *sigmas;
    float *rgbs;
    void *feats;          // optional [ntiles][64 lanes][2] x 16 B: (xd, xc) B fragments per lane
    const uint32_t *perm; // optional [M]: tile t works on samples perm[16t .. 16t+15] (nsr_sample_order); feats stay tile-major
    float bmin[3], bsize[3];
    float density_scale;
    uint32_t C_ch;
    uint32_t tiles_per_block;
    uint32_t fast_levels;     // bit l: level l is hashed and its size is a power of two
    NsrLevel lv[16];
};

template <int CD, bool SIGMA_ONLY>
__device__ __forceinline__ void field_build_fw(short *lds, const float *__restrict__ p) {
    mm_build_frags<CD>(lds + FW_D1, p + P_D1, 64, 32, 4, 32, false, 0, true);
    mm_build_frags<CD>(lds + FW_D2, p + P_D2, 16, 64, 1, 64, false, 0, true);
    if (!SIGMA_ONLY) {
        mm_build_frags<CD>(lds + FW_C1A, p + P_C1A, 64, 32, 4, 32, false, 0, true);
        mm_build_frags<CD>(lds + FW_C1B, p + P_C1B, 16, 64, 1, 64, false, 0, true);
        mm_build_frags<CD>(lds + FW_K1, p + P_K1, 64, 32, 4, 32, false, 0, true);
        mm_build_frags<CD>(lds + FW_K2, p + P_K2, 16, 64, 1, 64, false, CLASS_ROW_SHIFT, true);
        mm_build_frags<CD>(lds + FW_R2, p + P_R2, 64, 64, 4, 64, false, 0, true);
        mm_build_frags<CD>(lds + FW_R3, p + P_R3, 16, 64, 1, 64, false, 0, true);
        mm_build_frags<CD>(lds + FW_R1, p + P_R1, 64, 16, 4, 16, false, 0, false);
    }
}

// Encoder input of a world position: BBox.normalize (common.py:276-288) then GridEncoder's
// (x + bound) / (2 * bound) with bound = 1 (grid.py:177).  Same op order as the oracle.
__device__ __forceinline__ float field_unit(float x, float mn, float sz) {
#pragma clang fp contract(off)
    const float xn = (x - mn) / sz;
    return (xn + 1.0f) / 2.0f;
}

template <typename TT> struct RowLd;
template <> struct RowLd<float> {
    // one interleaved row = [d0 d1 c0 c1] fp32 = 16 bytes
    static __device__ __forceinline__ float4 full(const float *t, uint32_t row) {
        return reinterpret_cast<const float4 *>(t)[row];
    }
    static __device__ __forceinline__ float2 dens(const float *t, uint32_t row) {
        return reinterpret_cast<const float2 *>(t)[(size_t)row * 2];
    }
    // acc += sum over the 8 corners of w * row (all gathers issued first)
    static __device__ __forceinline__ void accumulate(const float *t, const uint32_t (&rows)[8], const float (&w)[8], float4 &acc) {
        float4 v[8];
#pragma unroll
        for (int idx = 0; idx < 8; idx++) v[idx] = full(t, rows[idx]);
#pragma unroll
        for (int idx = 0; idx < 8; idx++) {
            acc.x += w[idx] * v[idx].x; acc.y += w[idx] * v[idx].y;
            acc.z += w[idx] * v[idx].z; acc.w += w[idx] * v[idx].w;
        }
    }
};
template <> struct RowLd<_Float16> {
    static __device__ __forceinline__ float4 full(const _Float16 *t, uint32_t row) {
        const h4v v = reinterpret_cast<const h4v *>(t)[row];
        return make_float4((float)v[0], (float)v[1], (float)v[2], (float)v[3]);
    }
    static __device__ __forceinline__ float2 dens(const _Float16 *t, uint32_t row) {
        typedef _Float16 h2v __attribute__((ext_vector_type(2)));
        const h2v v = reinterpret_cast<const h2v *>(t)[(size_t)row * 2];
        return make_float2((float)v[0], (float)v[1]);
    }
    // fp32 accumulation straight from the packed halves: v_fma_mix_f32 widens its f16 operand inside the
    // multiply-add (same value as v_cvt_f32_f16 followed by v_fma_f32: the widening is exact), so the 32
    // conversions per level disappear from a kernel that is VALU-issue-bound half of the time
    static __device__ __forceinline__ void accumulate(const _Float16 *t, const uint32_t (&rows)[8], const float (&w)[8], float4 &acc) {
        uint2 v[8];
#ifdef NSR_ABL_FWD_HALFGATHER
#pragma unroll
        for (int idx = 0; idx < 8; idx += 2) v[idx] = v[idx + 1] = reinterpret_cast<const uint2 *>(t)[rows[idx]];
#else
#pragma unroll
        for (int idx = 0; idx < 8; idx++) v[idx] = reinterpret_cast<const uint2 *>(t)[rows[idx]];
#endif
#pragma unroll
        for (int idx = 0; idx < 8; idx++) {
            asm("v_fma_mix_f32 %0, %1, %2, %0 op_sel:[0,0,0] op_sel_hi:[0,1,0]" : "+v"(acc.x) : "v"(w[idx]), "v"(v[idx].x));
            asm("v_fma_mix_f32 %0, %1, %2, %0 op_sel:[0,1,0] op_sel_hi:[0,1,0]" : "+v"(acc.y) : "v"(w[idx]), "v"(v[idx].x));
            asm("v_fma_mix_f32 %0, %1, %2, %0 op_sel:[0,0,0] op_sel_hi:[0,1,0]" : "+v"(acc.z) : "v"(w[idx]), "v"(v[idx].y));
            asm("v_fma_mix_f32 %0, %1, %2, %0 op_sel:[0,1,0] op_sel_hi:[0,1,0]" : "+v"(acc.w) : "v"(w[idx]), "v"(v[idx].y));
        }
    }
};

// Trilinear interpolation of one level for both encoders (gridencoder.cu:134-181, align_corners
// = True, style = 0 on this path: networks/tcnn_nerf.py:26-35).
template <typename TT, bool SIGMA_ONLY, bool FAST>
__device__ __forceinline__ float4 field_encode_level(const NsrLevel &lv, const TT *__restrict__ tables, float u0, float u1,
                                                     float u2, bool live) {
    // FAST (wave-uniform, from the host's level table): the four levels this call handles, one per 16-lane
    // group, are all hashed with a power-of-two size, so row = (x ^ y*P1 ^ z*P2) & (size - 1) -- the value
    // nsr_grid_row gives -- without the per-lane hash/dense select and the invariant-divisor modulo.  Either
    // way the y and z products are formed once per level, not once per corner.
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    if (live) {
        float f[3];
        uint32_t c[3];
        nsr_grid_locate(u0, lv.resolution, 1, f[0], c[0]);
        nsr_grid_locate(u1, lv.resolution, 1, f[1], c[1]);
        nsr_grid_locate(u2, lv.resolution, 1, f[2], c[2]);
        // (wx*wy)*wz: the reference's product order (gridencoder.cu:160-175)
        const float wxy[4] = {(1 - f[0]) * (1 - f[1]), f[0] * (1 - f[1]), (1 - f[0]) * f[1], f[0] * f[1]};
        const float wz[2] = {1 - f[2], f[2]};
        const bool hashed = FAST || lv.use_hash != 0;
        const uint32_t mulY = hashed ? 2654435761u : lv.mul[1], mulZ = hashed ? 805459861u : lv.mul[2];
        const uint32_t ty[2] = {c[1] * mulY, (c[1] + 1u) * mulY}, tz[2] = {c[2] * mulZ, (c[2] + 1u) * mulZ};
        uint32_t rows[8];
        float w[8];
#pragma unroll
        for (uint32_t idx = 0; idx < 8; idx++) {
            w[idx] = wxy[idx & 3] * wz[idx >> 2];
            const uint32_t x = c[0] + (idx & 1u), a = ty[(idx >> 1) & 1], b = tz[idx >> 2];
            if (FAST) {
                rows[idx] = lv.offset + ((x ^ a ^ b) & (lv.size - 1u));
            } else {
                const uint32_t index = hashed ? (x ^ a ^ b) : (x * lv.mul[0] + a + b);
                const uint32_t t = __umulhi(lv.magic, index);
                const uint32_t q = (t + ((index - t) >> lv.sh1)) >> lv.sh2;
                rows[idx] = lv.offset + (index - q * lv.size);
            }
        }
        if (SIGMA_ONLY) {
            float2 v[8];
#pragma unroll
            for (int idx = 0; idx < 8; idx++) v[idx] = RowLd<TT>::dens(tables, rows[idx]);
#pragma unroll
            for (int idx = 0; idx < 8; idx++) { acc.x += w[idx] * v[idx].x; acc.y += w[idx] * v[idx].y; }
        } else {
            RowLd<TT>::accumulate(tables, rows, w, acc);
        }
    }
    return acc;
}

#ifndef NSR_FWD_BATCH_GATHER
#define NSR_FWD_BATCH_GATHER 0     /* all 32 gathers of a lane's four levels in flight before the first is used (set by field.hip) */
#endif
#if NSR_FWD_BATCH_GATHER
// rows and weights of one level (the first half of field_encode_level)
template <bool FAST>
__device__ __forceinline__ void field_level_rows(const NsrLevel &lv, float u0, float u1, float u2, uint32_t (&rows)[8], float (&w)[8]) {
    float f[3];
    uint32_t c[3];
    nsr_grid_locate(u0, lv.resolution, 1, f[0], c[0]);
    nsr_grid_locate(u1, lv.resolution, 1, f[1], c[1]);
    nsr_grid_locate(u2, lv.resolution, 1, f[2], c[2]);
    const float wxy[4] = {(1 - f[0]) * (1 - f[1]), f[0] * (1 - f[1]), (1 - f[0]) * f[1], f[0] * f[1]};
    const float wz[2] = {1 - f[2], f[2]};
    const bool hashed = FAST || lv.use_hash != 0;
    const uint32_t mulY = hashed ? 2654435761u : lv.mul[1], mulZ = hashed ? 805459861u : lv.mul[2];
    const uint32_t ty[2] = {c[1] * mulY, (c[1] + 1u) * mulY}, tz[2] = {c[2] * mulZ, (c[2] + 1u) * mulZ};
#pragma unroll
    for (uint32_t idx = 0; idx < 8; idx++) {
        w[idx] = wxy[idx & 3] * wz[idx >> 2];
        const uint32_t x = c[0] + (idx & 1u), a = ty[(idx >> 1) & 1], b = tz[idx >> 2];
        if (FAST) {
            rows[idx] = lv.offset + ((x ^ a ^ b) & (lv.size - 1u));
        } else {
            const uint32_t index = hashed ? (x ^ a ^ b) : (x * lv.mul[0] + a + b);
            const uint32_t t = __umulhi(lv.magic, index);
            const uint32_t q = (t + ((index - t) >> lv.sh1)) >> lv.sh2;
            rows[idx] = lv.offset + (index - q * lv.size);
        }
    }
}
#endif

// Encodes this lane's four levels; returns the two K=32 B fragments (density, colour).
template <typename TT, int CD, bool SIGMA_ONLY>
__device__ __forceinline__ void field_encode(const NsrLevel *lds_lv, const TT *__restrict__ tables, float u0, float u1, float u2,
                                             bool live, int g, s8v &xd, s8v &xc, uint32_t fast_levels) {
    const int lvl[4] = {2 * g, 2 * g + 1, 8 + 2 * g, 9 + 2 * g};
    // levels per call (one per lane group): {0,2,4,6} {1,3,5,7} {8,10,12,14} {9,11,13,15}
    const uint32_t call_levels[4] = {0x0055u, 0x00AAu, 0x5500u, 0xAA00u};
#if NSR_FWD_BATCH_GATHER
    if (!SIGMA_ONLY && sizeof(TT) == 2) {
        uint2 v[4][8];
        float w[4][8];
        if (live) {
#pragma unroll
            for (int i = 0; i < 4; i++) {
                const NsrLevel lv = lds_lv[lvl[i]];
                uint32_t rows[8];
                if ((fast_levels & call_levels[i]) == call_levels[i]) field_level_rows<true>(lv, u0, u1, u2, rows, w[i]);
                else field_level_rows<false>(lv, u0, u1, u2, rows, w[i]);
#pragma unroll
                for (int idx = 0; idx < 8; idx++) v[i][idx] = reinterpret_cast<const uint2 *>(tables)[rows[idx]];
            }
        }
#pragma unroll
        for (int i = 0; i < 4; i++) {
            float4 a = make_float4(0.f, 0.f, 0.f, 0.f);
            if (live) {
#pragma unroll
                for (int idx = 0; idx < 8; idx++) {
                    asm("v_fma_mix_f32 %0, %1, %2, %0 op_sel:[0,0,0] op_sel_hi:[0,1,0]" : "+v"(a.x) : "v"(w[i][idx]), "v"(v[i][idx].x));
                    asm("v_fma_mix_f32 %0, %1, %2, %0 op_sel:[0,1,0] op_sel_hi:[0,1,0]" : "+v"(a.y) : "v"(w[i][idx]), "v"(v[i][idx].x));
                    asm("v_fma_mix_f32 %0, %1, %2, %0 op_sel:[0,0,0] op_sel_hi:[0,1,0]" : "+v"(a.z) : "v"(w[i][idx]), "v"(v[i][idx].y));
                    asm("v_fma_mix_f32 %0, %1, %2, %0 op_sel:[0,1,0] op_sel_hi:[0,1,0]" : "+v"(a.w) : "v"(w[i][idx]), "v"(v[i][idx].y));
                }
            }
            xd[2 * i + 0] = MM<CD>::cvt(a.x);
            xd[2 * i + 1] = MM<CD>::cvt(a.y);
            xc[2 * i + 0] = MM<CD>::cvt(a.z);
            xc[2 * i + 1] = MM<CD>::cvt(a.w);
        }
        return;
    }
#endif
#pragma unroll
    for (int i = 0; i < 4; i++) {
        const NsrLevel lv = lds_lv[lvl[i]];
        const float4 a = (fast_levels & call_levels[i]) == call_levels[i]
                             ? field_encode_level<TT, SIGMA_ONLY, true>(lv, tables, u0, u1, u2, live)
                             : field_encode_level<TT, SIGMA_ONLY, false>(lv, tables, u0, u1, u2, live);
        xd[2 * i + 0] = MM<CD>::cvt(a.x);
        xd[2 * i + 1] = MM<CD>::cvt(a.y);
        xc[2 * i + 0] = MM<CD>::cvt(a.z);
        xc[2 * i + 1] = MM<CD>::cvt(a.w);
    }
}

__device__ __forceinline__ float field_sigmoid(float x) { return 1.0f / (1.0f + expf(-x)); }

// XCD-aware logical block id: hardware deals blocks round-robin over the 8 XCDs, so blocks b and
// b+8 share an L2.  Give each XCD one contiguous eighth of the tiles (neighbouring tiles =
// neighbouring samples/rays = shared coarse-level table lines).  Bijective for any grid size.
__device__ __forceinline__ uint32_t field_logical_block() {
    const uint32_t nb = gridDim.x, b = blockIdx.x;
    const uint32_t q = nb / 8, r = nb % 8, xcd = b % 8, i = b / 8;
    return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + i;
}


static int field_fill_args(const nsr_field_desc *d, FieldArgs &a, uint32_t M, uint32_t &nblocks) {
    if (d->L != 16) return NSR_ERR_UNSUPPORTED;             // 32 features = two K=32 halves per encoder
    if (d->num_classes > 13) return NSR_ERR_UNSUPPORTED;    // class rows 3..15 of one 16-row tile
    if (d->offsets == nullptr) return NSR_ERR_INVALID_ARG;
    NsrLevels lv;
    nsr_fill_levels(&lv, d->offsets, 16, d->S, d->H, 0u);
    a.fast_levels = 0;
    for (int l = 0; l < 16; l++) {
        a.lv[l] = lv.lv[l];
        if (lv.lv[l].size == 0) return NSR_ERR_INVALID_ARG;
        if (lv.lv[l].use_hash && (lv.lv[l].size & (lv.lv[l].size - 1u)) == 0) a.fast_levels |= 1u << l;
    }
    for (int i = 0; i < 3; i++) { a.bmin[i] = d->bbox_min[i]; a.bsize[i] = d->bbox_size[i]; }
    a.density_scale = d->density_scale;
    a.C_ch = 3 + d->num_classes;
    a.M = M;
    const uint32_t ntiles = (M + 15) / 16;
    // >= 8 tiles per wave so the per-block weight-image build amortises; <= 8 blocks per CU
    uint32_t nb = (ntiles + 31) / 32;
#ifndef NSR_FIELD_MAX_BLOCKS
#define NSR_FIELD_MAX_BLOCKS 2048
#endif
    if (nb > NSR_FIELD_MAX_BLOCKS) nb = NSR_FIELD_MAX_BLOCKS;
    if (nb == 0) nb = 1;
    nblocks = nb;
    a.tiles_per_block = (ntiles + nb - 1) / nb;
    return NSR_OK;
}

