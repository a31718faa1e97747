// Stand-alone multiresolution hash-grid encode (forward / backward) for gfx950: the C-ABI
// replacement of the reference's _gridencoder module (gridencoder/src/gridencoder.cu).  The hot
// path does not go through these kernels -- it uses the fused field kernels (field.hip), which
// share the index/locate helpers in nsr_common.h -- but GridEncoder as a stand-alone module does.
//
// Differences in structure from the reference (behaviour is the same):
//   * the per-level resolution / stride / hash decision is computed once on the host and passed
//     by value (no exp2f, no offsets[] loads, no per-corner stride loop on the device);
//   * outputs can be written directly in [B, L*C] (what grid.py:58 produces with a permute copy);
//   * fp32 accumulation of the interpolation, one rounding to the table type at the end
//     (the reference accumulates in scalar_t); backward always accumulates in fp32.
#include <hip/hip_fp16.h>

#include "nsr_common.h"

template <typename T> struct GeIO;
template <> struct GeIO<float> {
    __device__ static float ld(const float *p) { return *p; }
    __device__ static void st(float *p, float v) { *p = v; }
};
template <> struct GeIO<_Float16> {
    __device__ static float ld(const _Float16 *p) { return (float)*p; }
    __device__ static void st(_Float16 *p, float v) { *p = (_Float16)v; }
};

// gridencoder.cu:83-187.  One thread per (sample, level); samples are the fast index so that a
// wave's 64 lanes walk 64 consecutive samples of one level (consecutive samples of a ray share
// cells on the coarse levels -> identical addresses are merged by the texture addresser).
template <typename T, int C>
__global__ void __launch_bounds__(256)
k_grid_fwd(const float *__restrict__ inputs, const T *__restrict__ grid, T *__restrict__ outputs, uint32_t B, uint32_t L,
           NsrLevels levels, int align_corners, uint32_t style, int out_blc) {
    const uint32_t b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= B) return;
    const uint32_t level = blockIdx.y;
    const NsrLevel lv = levels.lv[level];
    T *out = out_blc ? outputs + ((size_t)b * L + level) * C : outputs + ((size_t)level * B + b) * C;
    const float x0 = inputs[(size_t)b * 3 + 0], x1 = inputs[(size_t)b * 3 + 1], x2 = inputs[(size_t)b * 3 + 2];
    // :107-132 out-of-range inputs produce zeros
    if (x0 < 0 || x0 > 1 || x1 < 0 || x1 > 1 || x2 < 0 || x2 > 1) {
#pragma unroll
        for (int ch = 0; ch < C; ch++) GeIO<T>::st(out + ch, 0.0f);
        return;
    }
    float f[3];
    uint32_t g[3];
    nsr_grid_locate(x0, lv.resolution, align_corners, f[0], g[0]);
    nsr_grid_locate(x1, lv.resolution, align_corners, f[1], g[1]);
    nsr_grid_locate(x2, lv.resolution, align_corners, f[2], g[2]);
    const T *tab = grid + (size_t)lv.offset * C;
    float acc[C];
#pragma unroll
    for (int ch = 0; ch < C; ch++) acc[ch] = 0.0f;
#pragma unroll
    for (uint32_t idx = 0; idx < 8; idx++) {
        float w = 1;
        uint32_t p[3];
#pragma unroll
        for (uint32_t d = 0; d < 3; d++) {
            if ((idx & (1u << d)) == 0) { w *= 1 - f[d]; p[d] = g[d]; }
            else { w *= f[d]; p[d] = g[d] + 1; }
        }
        const uint32_t row = nsr_grid_row(lv, p[0], p[1], p[2], style);
#pragma unroll
        for (int ch = 0; ch < C; ch++) acc[ch] += w * GeIO<T>::ld(tab + (size_t)row * C + ch);
    }
#pragma unroll
    for (int ch = 0; ch < C; ch++) GeIO<T>::st(out + ch, acc[ch]);
}

// gridencoder.cu:238-328.  fp32 atomics into grad_grid.
template <typename T, int C>
__global__ void __launch_bounds__(256)
k_grid_bwd(const T *__restrict__ grad, const float *__restrict__ inputs, float *__restrict__ grad_grid, uint32_t B, uint32_t L,
           NsrLevels levels, int align_corners, uint32_t style, int grad_blc) {
    const uint32_t b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= B) return;
    const uint32_t level = blockIdx.y;
    const NsrLevel lv = levels.lv[level];
    const float x0 = inputs[(size_t)b * 3 + 0], x1 = inputs[(size_t)b * 3 + 1], x2 = inputs[(size_t)b * 3 + 2];
    if (x0 < 0 || x0 > 1 || x1 < 0 || x1 > 1 || x2 < 0 || x2 > 1) return;   // :268-273
    const T *gp = grad_blc ? grad + ((size_t)b * L + level) * C : grad + ((size_t)level * B + b) * C;
    float gcur[C];
#pragma unroll
    for (int ch = 0; ch < C; ch++) gcur[ch] = GeIO<T>::ld(gp + ch);
    float f[3];
    uint32_t g[3];
    nsr_grid_locate(x0, lv.resolution, align_corners, f[0], g[0]);
    nsr_grid_locate(x1, lv.resolution, align_corners, f[1], g[1]);
    nsr_grid_locate(x2, lv.resolution, align_corners, f[2], g[2]);
    float *tab = grad_grid + (size_t)lv.offset * C;
#pragma unroll
    for (uint32_t idx = 0; idx < 8; idx++) {
        float w = 1;
        uint32_t p[3];
#pragma unroll
        for (uint32_t d = 0; d < 3; d++) {
            if ((idx & (1u << d)) == 0) { w *= 1 - f[d]; p[d] = g[d]; }
            else { w *= f[d]; p[d] = g[d] + 1; }
        }
        const uint32_t row = nsr_grid_row(lv, p[0], p[1], p[2], style);
#pragma unroll
        for (int ch = 0; ch < C; ch++) atomicAdd(tab + (size_t)row * C + ch, w * gcur[ch]);
    }
}

__global__ void k_cast_f32_f16(const float *__restrict__ src, _Float16 *__restrict__ dst, uint64_t n) {
    // 8 scalars per thread per step: two 16-byte loads, one 16-byte store
    const uint64_t n8 = n / 8;
    for (uint64_t i = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; i < n8; i += (uint64_t)gridDim.x * blockDim.x) {
        const float4 a = reinterpret_cast<const float4 *>(src)[i * 2];
        const float4 b = reinterpret_cast<const float4 *>(src)[i * 2 + 1];
        typedef _Float16 h8 __attribute__((ext_vector_type(8)));
        h8 o;
        o[0] = (_Float16)a.x; o[1] = (_Float16)a.y; o[2] = (_Float16)a.z; o[3] = (_Float16)a.w;
        o[4] = (_Float16)b.x; o[5] = (_Float16)b.y; o[6] = (_Float16)b.z; o[7] = (_Float16)b.w;
        reinterpret_cast<h8 *>(dst)[i] = o;
    }
    if (blockIdx.x == 0 && threadIdx.x < (n & 7u)) dst[n8 * 8 + threadIdx.x] = (_Float16)src[n8 * 8 + threadIdx.x];
}

template <typename T>
static int launch_fwd(const float *inputs, const void *emb, void *out, uint32_t B, uint32_t C, uint32_t L,
                      const NsrLevels &lv, int align_corners, uint32_t style, int out_blc, hipStream_t s) {
    const dim3 grid(nsr_div_up(B, 256), L, 1), block(256);
    switch (C) {
        case 1: hipLaunchKernelGGL((k_grid_fwd<T, 1>), grid, block, 0, s, inputs, (const T *)emb, (T *)out, B, L, lv, align_corners, style, out_blc); break;
        case 2: hipLaunchKernelGGL((k_grid_fwd<T, 2>), grid, block, 0, s, inputs, (const T *)emb, (T *)out, B, L, lv, align_corners, style, out_blc); break;
        case 4: hipLaunchKernelGGL((k_grid_fwd<T, 4>), grid, block, 0, s, inputs, (const T *)emb, (T *)out, B, L, lv, align_corners, style, out_blc); break;
        case 8: hipLaunchKernelGGL((k_grid_fwd<T, 8>), grid, block, 0, s, inputs, (const T *)emb, (T *)out, B, L, lv, align_corners, style, out_blc); break;
        default: return NSR_ERR_UNSUPPORTED;   // gridencoder.cu:369
    }
    return nsr_launch_status();
}

template <typename T>
static int launch_bwd(const void *grad, const float *inputs, float *gg, uint32_t B, uint32_t C, uint32_t L, const NsrLevels &lv,
                      int align_corners, uint32_t style, int grad_blc, hipStream_t s) {
    const dim3 grid(nsr_div_up(B, 256), L, 1), block(256);
    switch (C) {
        case 1: hipLaunchKernelGGL((k_grid_bwd<T, 1>), grid, block, 0, s, (const T *)grad, inputs, gg, B, L, lv, align_corners, style, grad_blc); break;
        case 2: hipLaunchKernelGGL((k_grid_bwd<T, 2>), grid, block, 0, s, (const T *)grad, inputs, gg, B, L, lv, align_corners, style, grad_blc); break;
        case 4: hipLaunchKernelGGL((k_grid_bwd<T, 4>), grid, block, 0, s, (const T *)grad, inputs, gg, B, L, lv, align_corners, style, grad_blc); break;
        case 8: hipLaunchKernelGGL((k_grid_bwd<T, 8>), grid, block, 0, s, (const T *)grad, inputs, gg, B, L, lv, align_corners, style, grad_blc); break;
        default: return NSR_ERR_UNSUPPORTED;
    }
    return nsr_launch_status();
}

extern "C" {

int nsr_grid_resolutions(uint32_t L, float S, uint32_t H, uint32_t *res_out) {
    NSR_CHECK_PTR(res_out);
    for (uint32_t l = 0; l < L; l++) res_out[l] = (uint32_t)floorf(exp2f((float)l * S) * (float)H);
    return NSR_OK;
}

int nsr_grid_encode_forward(const float *inputs, const void *embeddings, int emb_dtype, const int32_t *offsets, void *outputs,
                            uint32_t B, uint32_t D, uint32_t C, uint32_t L, float S, uint32_t H, int calc_grad_inputs,
                            uint32_t gridtype, int align_corners, uint32_t style, int out_blc, nsr_stream_t stream) {
    if (B == 0) return NSR_OK;
    NSR_CHECK_PTR(inputs); NSR_CHECK_PTR(embeddings); NSR_CHECK_PTR(offsets); NSR_CHECK_PTR(outputs);
    if (D != 3 || calc_grad_inputs) return NSR_ERR_UNSUPPORTED;   // gridencoder.cu:387 raises for D outside 1..5
    if (L == 0 || L > NSR_MAX_LEVELS || gridtype > 1) return NSR_ERR_INVALID_ARG;
    NsrLevels lv;
    nsr_fill_levels(&lv, offsets, L, S, H, gridtype);
    if (emb_dtype == NSR_F32) return launch_fwd<float>(inputs, embeddings, outputs, B, C, L, lv, align_corners, style, out_blc, (hipStream_t)stream);
    if (emb_dtype == NSR_F16) return launch_fwd<_Float16>(inputs, embeddings, outputs, B, C, L, lv, align_corners, style, out_blc, (hipStream_t)stream);
    return NSR_ERR_UNSUPPORTED;
}

int nsr_grid_encode_backward(const void *grad, int grad_dtype, const float *inputs, const int32_t *offsets,
                             float *grad_embeddings, uint32_t B, uint32_t D, uint32_t C, uint32_t L, float S, uint32_t H,
                             uint32_t gridtype, int align_corners, uint32_t style, int grad_blc, nsr_stream_t stream) {
    if (B == 0) return NSR_OK;
    NSR_CHECK_PTR(grad); NSR_CHECK_PTR(inputs); NSR_CHECK_PTR(offsets); NSR_CHECK_PTR(grad_embeddings);
    if (D != 3) return NSR_ERR_UNSUPPORTED;
    if (L == 0 || L > NSR_MAX_LEVELS || gridtype > 1) return NSR_ERR_INVALID_ARG;
    NsrLevels lv;
    nsr_fill_levels(&lv, offsets, L, S, H, gridtype);
    if (grad_dtype == NSR_F32) return launch_bwd<float>(grad, inputs, grad_embeddings, B, C, L, lv, align_corners, style, grad_blc, (hipStream_t)stream);
    if (grad_dtype == NSR_F16) return launch_bwd<_Float16>(grad, inputs, grad_embeddings, B, C, L, lv, align_corners, style, grad_blc, (hipStream_t)stream);
    return NSR_ERR_UNSUPPORTED;
}

int nsr_cast_f32_to_f16(const float *src, void *dst, uint64_t n, nsr_stream_t stream) {
    if (n == 0) return NSR_OK;
    NSR_CHECK_PTR(src); NSR_CHECK_PTR(dst);
    if ((((uintptr_t)src | (uintptr_t)dst) & 15u) != 0) return NSR_ERR_INVALID_ARG;
    hipLaunchKernelGGL(k_cast_f32_f16, dim3(nsr_grid_1d(n / 8 + 1, 256)), dim3(256), 0, (hipStream_t)stream, src, (_Float16 *)dst, n);
    return nsr_launch_status();
}

}   // extern "C"
