// Training composite (raymarching.cu:806-997), the render_train epilogue (renderer.py:225-233) and the reconstruction
// loss (trainers/base.py:251-304: MSE + lambda * cross-entropy) as three gfx950 kernels -- round 3.
//
// Rounds 1-2 gave a ray to a group of 4/8/16 lanes (one channel per lane) that walked the ray's samples one after the
// other: a serial dependence chain per ray, waves as long as their longest ray, 1.25 + 1.85 ms on the bench frame for
// 2.5 + 4.3 GB (0.25 of the HBM roofline), followed by ~50 small torch kernels for the white background, the slices, the
// MSE, the log-sum-exp and their backward (1.7 ms).  Here a WAVE owns a ray and a LANE owns a sample:
//   * 64 consecutive samples of the ray per trip: sigma / delta / colour rows are whole contiguous pieces of the sample
//     buffers (16-byte loads for C = 4 or 8);
//   * the recurrences become wave scans on the DPP network (row_shr 1/2/4/8, row_bcast 15/31): transmittance = exclusive
//     product scan of (1 - alpha), t = inclusive sum scan of the step lengths, the backward's running colour = inclusive
//     sum scans of weight * colour; sums over the ray are lane 63 of a scan; a carry (T, t, sums) links the trips of a
//     ray longer than 64 samples; the early stop (T < T_thresh, :862 / :961) is a ballot;
//   * persistent waves (grid-stride over rays), the next ray's header in flight while this one is evaluated;
//   * the forward also writes what Renderer.render_train returns (image[:, :3] + (1 - weights_sum), the class channels,
//     (depth - near) clamped / (far - near)), the backward takes the gradients of exactly those outputs.
// The scans reassociate the reference's serial fp32 products and sums: results differ from the oracle in the last bits
// (tests: <= 2e-5 absolute), and are bit-identical from run to run.
//
// The loss kernel is one thread per ray (value + gradient in one pass, targets gathered through the pixel indices), a
// block tree for the value and a second one-block kernel that adds the block partials in a fixed order.
#include "nsr_common.h"

#define CP_MAXC 16
#define CP_BLOCK 256

#define CP_DPP(old, v, ctrl, rm, bm)                                                                                  \
    __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, (float)(old)), __builtin_bit_cast(int, (float)(v)), \
                                                          (ctrl), (rm), (bm), false))

// inclusive wave64 scans: lanes without a source keep the identity (`old` operand, bound_ctrl off)
__device__ __forceinline__ float cp_scan_add(float v) {
    v += CP_DPP(0.0f, v, 0x111, 0xF, 0xF);      // row_shr:1
    v += CP_DPP(0.0f, v, 0x112, 0xF, 0xF);      // row_shr:2
    v += CP_DPP(0.0f, v, 0x114, 0xF, 0xF);      // row_shr:4
    v += CP_DPP(0.0f, v, 0x118, 0xF, 0xF);      // row_shr:8
    v += CP_DPP(0.0f, v, 0x142, 0xA, 0xF);      // row_bcast:15 -> rows 1, 3
    v += CP_DPP(0.0f, v, 0x143, 0xC, 0xF);      // row_bcast:31 -> rows 2, 3
    return v;
}
__device__ __forceinline__ float cp_scan_mul(float v) {
    v *= CP_DPP(1.0f, v, 0x111, 0xF, 0xF);
    v *= CP_DPP(1.0f, v, 0x112, 0xF, 0xF);
    v *= CP_DPP(1.0f, v, 0x114, 0xF, 0xF);
    v *= CP_DPP(1.0f, v, 0x118, 0xF, 0xF);
    v *= CP_DPP(1.0f, v, 0x142, 0xA, 0xF);
    v *= CP_DPP(1.0f, v, 0x143, 0xC, 0xF);
    return v;
}
// lane i <- lane i - 1 (lane 0 <- `first`)
__device__ __forceinline__ float cp_shift_up(float v, float first) { return CP_DPP(first, v, 0x138, 0xF, 0xF); }   // wave_shr:1
__device__ __forceinline__ float cp_lane(float v, int lane) {
    return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), lane));
}
__device__ __forceinline__ float cp_sum(float v) { return cp_lane(cp_scan_add(v), 63); }

struct CompArgs {
    const float *sigmas, *rgbs, *deltas;
    const int32_t *rays;
    uint32_t M, N, C;
    float T_thresh;
    int is_ndc;
    // forward outputs
    float *weights_sum, *depth, *image;
    const float *nears, *fars;             // epilogue (both or neither)
    float *rgb_map, *depth_norm, *classes; // epilogue outputs: [N,3], [N], [N, C-3] (classes may be NULL)
    // backward inputs / outputs
    const float *g_ws, *g_image;           // legacy form: d/d weights_sum [N], d/d image [N,C]
    const float *g_rgb_map, *g_classes;    // epilogue form: d/d rgb_map [N,3], d/d classes [N,C-3] (each may be NULL)
    const float *ws_in, *image_in;
    float *g_sigmas, *g_rgbs;
    int epilogue, zero_fill;
};

// one sample's colour row into registers; CT > 0: compile-time C with 16-byte loads, CT == 0: runtime C, scalar loads
template <int CT>
__device__ __forceinline__ void cp_load_row(const float *__restrict__ rgbs, size_t p, uint32_t C, float (&c)[CT ? CT : CP_MAXC]) {
    if (CT > 0) {
#pragma unroll
        for (int q = 0; q < CT / 4; q++) {
            const float4 v = reinterpret_cast<const float4 *>(rgbs + p * CT)[q];
            c[4 * q] = v.x; c[4 * q + 1] = v.y; c[4 * q + 2] = v.z; c[4 * q + 3] = v.w;
        }
    } else {
#pragma unroll
        for (int k = 0; k < CP_MAXC; k++) c[k] = (uint32_t)k < C ? rgbs[p * C + k] : 0.0f;
    }
}

template <int CT>
__global__ void __launch_bounds__(CP_BLOCK)
k_comp_fwd(CompArgs a) {
    constexpr int NC = CT ? CT : CP_MAXC;
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t nwaves = gridDim.x * (CP_BLOCK / 64), gw = blockIdx.x * (CP_BLOCK / 64) + (threadIdx.x >> 6);
    const uint32_t C = CT ? (uint32_t)CT : a.C;
    uint32_t n = gw;
    int32_t h0 = 0, h1 = 0, h2 = 0;
    if (n < a.N) { h0 = a.rays[n * 3]; h1 = a.rays[n * 3 + 1]; h2 = a.rays[n * 3 + 2]; }
    for (; n < a.N; n += nwaves) {
        const uint32_t index = (uint32_t)__builtin_amdgcn_readfirstlane(h0), offset = (uint32_t)__builtin_amdgcn_readfirstlane(h1),
                       steps = (uint32_t)__builtin_amdgcn_readfirstlane(h2);
        const uint32_t nn = n + nwaves;
        if (nn < a.N) { h0 = a.rays[nn * 3]; h1 = a.rays[nn * 3 + 1]; h2 = a.rays[nn * 3 + 2]; }     // next header in flight
        float T = 1.0f, t = 0.0f, ws = 0.0f, d = 0.0f;
        float acc[NC];
#pragma unroll
        for (int k = 0; k < NC; k++) acc[k] = 0.0f;
        if (steps != 0 && offset + steps < a.M) {                                                   // :830
            for (uint32_t base = 0; base < steps; base += 64) {
                const uint32_t i = base + lane;
                const bool in = i < steps;
                const size_t p = (size_t)offset + min(i, steps - 1u);
                const float sg = a.sigmas[p];
                const float2 dd = *reinterpret_cast<const float2 *>(a.deltas + p * 4 + (a.is_ndc ? 2 : 0));
                float c[NC];
                cp_load_row<CT>(a.rgbs, p, C, c);
                const float alpha = in ? 1.0f - __expf(-sg * dd.x) : 0.0f;
                const float incl = cp_scan_mul(1.0f - alpha);
                const float Tb = T * cp_shift_up(incl, 1.0f);              // transmittance in front of the sample
                const float tc = t + cp_scan_add(in ? dd.y : 0.0f);        // t += deltas[1] (:857)
                // the serial loop reaches sample i iff every T after the samples before it stayed >= T_thresh (:862)
                const bool live = in && (i == 0u || Tb >= a.T_thresh);
                const float w = live ? alpha * Tb : 0.0f;
                ws += cp_sum(w);
                d += cp_sum(w * tc);
#pragma unroll
                for (int k = 0; k < NC; k++)
                    if (CT || (uint32_t)k < C) acc[k] += cp_sum(w * c[k]);
                T *= cp_lane(incl, 63);
                t = cp_lane(tc, 63);
                if (__ballot(in && !live) != 0ull || T < a.T_thresh) break;
            }
        }
        // lane k holds channel k
        float mine = acc[0];
#pragma unroll
        for (int k = 1; k < NC; k++) mine = lane == (uint32_t)k ? acc[k] : mine;
        if (lane < C) a.image[(size_t)index * C + lane] = mine;
        if (lane == 0) { a.weights_sum[index] = ws; a.depth[index] = d; }
        if (a.epilogue) {
            // renderer.py:229-233: white background, depth normalised to [near, far]
            if (lane < 3u) a.rgb_map[(size_t)index * 3 + lane] = mine + (1.0f - ws);
            else if (lane < C && a.classes) a.classes[(size_t)index * (C - 3u) + (lane - 3u)] = mine;
            if (lane == 0) {
                const float nr = a.nears[index], fr = a.fars[index];
                a.depth_norm[index] = fmaxf(d - nr, 0.0f) / (fr - nr);
            }
        }
    }
}

template <int CT>
__global__ void __launch_bounds__(CP_BLOCK)
k_comp_bwd(CompArgs a) {
    constexpr int NC = CT ? CT : CP_MAXC;
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t nwaves = gridDim.x * (CP_BLOCK / 64), gw = blockIdx.x * (CP_BLOCK / 64) + (threadIdx.x >> 6);
    const uint32_t C = CT ? (uint32_t)CT : a.C;
    uint32_t n = gw;
    int32_t h0 = 0, h1 = 0, h2 = 0;
    if (n < a.N) { h0 = a.rays[n * 3]; h1 = a.rays[n * 3 + 1]; h2 = a.rays[n * 3 + 2]; }
    for (; n < a.N; n += nwaves) {
        const uint32_t index = (uint32_t)__builtin_amdgcn_readfirstlane(h0), offset = (uint32_t)__builtin_amdgcn_readfirstlane(h1),
                       steps = (uint32_t)__builtin_amdgcn_readfirstlane(h2);
        const uint32_t nn = n + nwaves;
        if (nn < a.N) { h0 = a.rays[nn * 3]; h1 = a.rays[nn * 3 + 1]; h2 = a.rays[nn * 3 + 2]; }
        if (steps == 0) continue;
        uint32_t base = 0;
        if (offset + steps < a.M) {
            // per-ray gradients of the outputs, in lane k for channel k, then wave-uniform
            float gl = 0.0f, il = 0.0f;
            if (lane < C) {
                il = a.image_in[(size_t)index * C + lane];
                if (a.epilogue) {
                    if (lane < 3u) gl = a.g_rgb_map ? a.g_rgb_map[(size_t)index * 3 + lane] : 0.0f;
                    else gl = a.g_classes ? a.g_classes[(size_t)index * (C - 3u) + (lane - 3u)] : 0.0f;
                } else {
                    gl = a.g_image[(size_t)index * C + lane];
                }
            }
            float gim[NC], im[NC];
#pragma unroll
            for (int k = 0; k < NC; k++) { gim[k] = cp_lane(gl, k); im[k] = cp_lane(il, k); }
            float gws = a.g_ws ? a.g_ws[index] : 0.0f;
            // rgb_map = image[:3] + (1 - ws): d/d ws collects minus the three colour gradients
            if (a.epilogue) gws -= gim[0] + gim[1] + gim[2];
            const float ws_final = a.ws_in[index];
            const float tail = gws * (1.0f - ws_final);
            float T = 1.0f;
            float buf[NC];
#pragma unroll
            for (int k = 0; k < NC; k++) buf[k] = 0.0f;
            for (; base < steps; base += 64) {
                const uint32_t i = base + lane;
                const bool in = i < steps;
                const size_t p = (size_t)offset + min(i, steps - 1u);
                const float sg = a.sigmas[p];
                const float dt = a.deltas[p * 4 + (a.is_ndc ? 2 : 0)];
                float c[NC];
                cp_load_row<CT>(a.rgbs, p, C, c);
                const float alpha = in ? 1.0f - __expf(-sg * dt) : 0.0f;
                const float incl = cp_scan_mul(1.0f - alpha);
                const float Tb = T * cp_shift_up(incl, 1.0f);
                const float Ta = T * incl;                                  // T after the sample's update (:959)
                const float w = alpha * Tb;
                const bool live = in && Ta >= a.T_thresh;                   // :961 -- the stopping sample gets no gradient
                float gsum = 0.0f;
                float gc[NC];
#pragma unroll
                for (int k = 0; k < NC; k++) {
                    if (CT || (uint32_t)k < C) {
                        const float run = buf[k] + cp_scan_add(w * c[k]);   // rgbs_buf after this sample (:957)
                        gsum += gim[k] * (Ta * c[k] - (im[k] - run));
                        gc[k] = live ? gim[k] * w : 0.0f;
                        buf[k] = cp_lane(run, 63);
                    } else {
                        gc[k] = 0.0f;
                    }
                }
                if (in && (live || a.zero_fill)) {
                    a.g_sigmas[p] = live ? dt * (gsum + tail) : 0.0f;
                    if (CT > 0) {
#pragma unroll
                        for (int q = 0; q < CT / 4; q++)
                            reinterpret_cast<float4 *>(a.g_rgbs + p * CT)[q] = make_float4(gc[4 * q], gc[4 * q + 1], gc[4 * q + 2], gc[4 * q + 3]);
                    } else {
#pragma unroll
                        for (int k = 0; k < NC; k++)
                            if ((uint32_t)k < C) a.g_rgbs[p * C + k] = gc[k];
                    }
                }
                T *= cp_lane(incl, 63);
                if (__ballot(in && !live) != 0ull) { base += 64; break; }
            }
        }
        // dropped ray (:929, inside the buffer only) or the samples behind an early stop: the zero the reference pre-fills
        if (a.zero_fill) {
            for (; base < steps; base += 64) {
                const uint32_t i = base + lane;
                const size_t p = (size_t)offset + i;
                if (i < steps && p < a.M) {
                    a.g_sigmas[p] = 0.0f;
                    for (uint32_t k = 0; k < C; k++) a.g_rgbs[p * C + k] = 0.0f;
                }
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// reconstruction loss: value and gradient in one pass
// ---------------------------------------------------------------------------------------------------------------------
struct LossArgs {
    const float *rgb_map, *classes;       // [N,3], [N,nc] (classes may be NULL: MSE only)
    const float *target_rgb;              // [P,3]
    const int64_t *target_cls;            // [P] (may be NULL)
    const int64_t *pix;                   // [N] rows of the targets (NULL: row n)
    uint32_t N, nc;
    float ce_lambda, factor;
    const float *scale;                   // device scalar (loss scale) or NULL
    float *g_rgb, *g_classes;             // [N,3], [N,nc]
    float *partials;                      // [blocks][2]
    float *out;                           // [3]: scaled total, mse, ce (both unscaled)
    uint32_t nblocks;
};

__global__ void __launch_bounds__(CP_BLOCK)
k_recon_loss(LossArgs a) {
    __shared__ float red[2][CP_BLOCK / 64];
    const float s = (a.scale ? a.scale[0] : 1.0f) * a.factor;
    const float inv3n = 1.0f / (3.0f * (float)a.N), invn = 1.0f / (float)a.N;
    float mse = 0.0f, ce = 0.0f;
    for (uint32_t n = blockIdx.x * CP_BLOCK + threadIdx.x; n < a.N; n += gridDim.x * CP_BLOCK) {
        const size_t row = a.pix ? (size_t)a.pix[n] : (size_t)n;
#pragma unroll
        for (int k = 0; k < 3; k++) {
            const float df = a.rgb_map[(size_t)n * 3 + k] - a.target_rgb[row * 3 + k];
            mse += df * df;
            a.g_rgb[(size_t)n * 3 + k] = 2.0f * df * inv3n * s;
        }
        if (a.classes && a.nc) {
            // nn.CrossEntropyLoss (trainers/base.py:281): logsumexp(logits) - logits[label], mean over the rays
            const float *lg = a.classes + (size_t)n * a.nc;
            const uint32_t label = (uint32_t)a.target_cls[row];
            float mx = lg[0];
            for (uint32_t k = 1; k < a.nc; k++) mx = fmaxf(mx, lg[k]);
            float se = 0.0f;
            for (uint32_t k = 0; k < a.nc; k++) se += expf(lg[k] - mx);
            const float lse = mx + logf(se);
            ce += lse - lg[min(label, a.nc - 1u)];
            const float gk = a.ce_lambda * invn * s;
            for (uint32_t k = 0; k < a.nc; k++) a.g_classes[(size_t)n * a.nc + k] = (expf(lg[k] - lse) - (k == label ? 1.0f : 0.0f)) * gk;
        }
    }
    // fixed-order block tree
    for (int off = 32; off >= 1; off >>= 1) { mse += __shfl_xor(mse, off, 64); ce += __shfl_xor(ce, off, 64); }
    if ((threadIdx.x & 63) == 0) { red[0][threadIdx.x >> 6] = mse; red[1][threadIdx.x >> 6] = ce; }
    __syncthreads();
    if (threadIdx.x == 0) {
        float m = 0.0f, c = 0.0f;
        for (int w = 0; w < CP_BLOCK / 64; w++) { m += red[0][w]; c += red[1][w]; }
        a.partials[blockIdx.x * 2] = m;
        a.partials[blockIdx.x * 2 + 1] = c;
    }
}

__global__ void __launch_bounds__(CP_BLOCK)
k_recon_loss_final(LossArgs a) {
    __shared__ float red[2][CP_BLOCK / 64];
    float mse = 0.0f, ce = 0.0f;
    for (uint32_t b = threadIdx.x; b < a.nblocks; b += CP_BLOCK) { mse += a.partials[b * 2]; ce += a.partials[b * 2 + 1]; }
    for (int off = 32; off >= 1; off >>= 1) { mse += __shfl_xor(mse, off, 64); ce += __shfl_xor(ce, off, 64); }
    if ((threadIdx.x & 63) == 0) { red[0][threadIdx.x >> 6] = mse; red[1][threadIdx.x >> 6] = ce; }
    __syncthreads();
    if (threadIdx.x == 0) {
        float m = 0.0f, c = 0.0f;
        for (int w = 0; w < CP_BLOCK / 64; w++) { m += red[0][w]; c += red[1][w]; }
        m /= 3.0f * (float)a.N;
        c = c / (float)a.N * a.ce_lambda;
        const float s = (a.scale ? a.scale[0] : 1.0f) * a.factor;
        a.out[0] = (m + c) * s;
        a.out[1] = m;
        a.out[2] = c;
    }
}

static uint32_t cp_grid(uint32_t N) {
    // persistent waves: 256 CUs x 8 blocks of 4 waves; fewer when there are fewer rays
    const uint32_t want = nsr_div_up(N, CP_BLOCK / 64);
    return want < 2048u ? (want ? want : 1u) : 2048u;
}

template <bool BWD>
static int cp_launch(const CompArgs &a, hipStream_t s) {
    const dim3 g(cp_grid(a.N)), b(CP_BLOCK);
    const bool al16 = (((uintptr_t)a.rgbs | (uintptr_t)(BWD ? a.g_rgbs : a.rgbs)) & 15u) == 0;
    if (a.C == 8 && al16) {
        if (BWD) hipLaunchKernelGGL((k_comp_bwd<8>), g, b, 0, s, a); else hipLaunchKernelGGL((k_comp_fwd<8>), g, b, 0, s, a);
    } else if (a.C == 4 && al16) {
        if (BWD) hipLaunchKernelGGL((k_comp_bwd<4>), g, b, 0, s, a); else hipLaunchKernelGGL((k_comp_fwd<4>), g, b, 0, s, a);
    } else {
        if (BWD) hipLaunchKernelGGL((k_comp_bwd<0>), g, b, 0, s, a); else hipLaunchKernelGGL((k_comp_fwd<0>), g, b, 0, s, a);
    }
    return nsr_launch_status();
}

extern "C" {

int nsr_composite_rays_train_forward(const float *sigmas, const float *rgbs, const float *deltas, const int32_t *rays,
                                     uint32_t M, uint32_t N, uint32_t C, float T_thresh, int is_ndc, float *weights_sum,
                                     float *depth, float *image, nsr_stream_t stream) {
    if (N == 0) return NSR_OK;
    NSR_CHECK_PTR(sigmas); NSR_CHECK_PTR(rgbs); NSR_CHECK_PTR(deltas); NSR_CHECK_PTR(rays);
    NSR_CHECK_PTR(weights_sum); NSR_CHECK_PTR(depth); NSR_CHECK_PTR(image);
    if (C == 0 || C > CP_MAXC) return NSR_ERR_UNSUPPORTED;
    if (((uintptr_t)deltas & 7u) != 0) return NSR_ERR_INVALID_ARG;
    CompArgs a = {};
    a.sigmas = sigmas; a.rgbs = rgbs; a.deltas = deltas; a.rays = rays; a.M = M; a.N = N; a.C = C; a.T_thresh = T_thresh;
    a.is_ndc = is_ndc; a.weights_sum = weights_sum; a.depth = depth; a.image = image;
    return cp_launch<false>(a, (hipStream_t)stream);
}

int nsr_composite_rays_train_backward(const float *grad_weights_sum, const float *grad_image, const float *sigmas,
                                      const float *rgbs, const float *deltas, const int32_t *rays, int is_ndc,
                                      const float *weights_sum, const float *image, uint32_t M, uint32_t N, uint32_t C,
                                      float T_thresh, float *grad_sigmas, float *grad_rgbs, nsr_stream_t stream) {
    if (N == 0) return NSR_OK;
    NSR_CHECK_PTR(grad_weights_sum); NSR_CHECK_PTR(grad_image); NSR_CHECK_PTR(sigmas); NSR_CHECK_PTR(rgbs);
    NSR_CHECK_PTR(deltas); NSR_CHECK_PTR(rays); NSR_CHECK_PTR(weights_sum); NSR_CHECK_PTR(image);
    NSR_CHECK_PTR(grad_sigmas); NSR_CHECK_PTR(grad_rgbs);
    if (C == 0 || C > CP_MAXC) return NSR_ERR_UNSUPPORTED;
    CompArgs a = {};
    a.sigmas = sigmas; a.rgbs = rgbs; a.deltas = deltas; a.rays = rays; a.M = M; a.N = N; a.C = C; a.T_thresh = T_thresh;
    a.is_ndc = is_ndc; a.g_ws = grad_weights_sum; a.g_image = grad_image; a.ws_in = weights_sum; a.image_in = image;
    a.g_sigmas = grad_sigmas; a.g_rgbs = grad_rgbs; a.zero_fill = 1;
    return cp_launch<true>(a, (hipStream_t)stream);
}

int nsr_render_train_forward(const float *sigmas, const float *rgbs, const float *deltas, const int32_t *rays, const float *nears,
                             const float *fars, uint32_t M, uint32_t N, uint32_t C, float T_thresh, float *weights_sum,
                             float *depth, float *image, float *rgb_map, float *depth_norm, float *classes, nsr_stream_t stream) {
    if (N == 0) return NSR_OK;
    NSR_CHECK_PTR(sigmas); NSR_CHECK_PTR(rgbs); NSR_CHECK_PTR(deltas); NSR_CHECK_PTR(rays); NSR_CHECK_PTR(nears); NSR_CHECK_PTR(fars);
    NSR_CHECK_PTR(weights_sum); NSR_CHECK_PTR(depth); NSR_CHECK_PTR(image); NSR_CHECK_PTR(rgb_map); NSR_CHECK_PTR(depth_norm);
    if (C < 3 || C > CP_MAXC) return NSR_ERR_UNSUPPORTED;
    if (C > 3 && classes == nullptr) return NSR_ERR_INVALID_ARG;
    if (((uintptr_t)deltas & 7u) != 0) return NSR_ERR_INVALID_ARG;
    CompArgs a = {};
    a.sigmas = sigmas; a.rgbs = rgbs; a.deltas = deltas; a.rays = rays; a.M = M; a.N = N; a.C = C; a.T_thresh = T_thresh;
    a.weights_sum = weights_sum; a.depth = depth; a.image = image; a.nears = nears; a.fars = fars; a.rgb_map = rgb_map;
    a.depth_norm = depth_norm; a.classes = classes; a.epilogue = 1;
    return cp_launch<false>(a, (hipStream_t)stream);
}

int nsr_render_train_backward(const float *grad_rgb_map, const float *grad_classes, const float *grad_weights_sum, const float *sigmas,
                              const float *rgbs, const float *deltas, const int32_t *rays, const float *weights_sum,
                              const float *image, uint32_t M, uint32_t N, uint32_t C, float T_thresh, float *grad_sigmas,
                              float *grad_rgbs, nsr_stream_t stream) {
    if (N == 0) return NSR_OK;
    NSR_CHECK_PTR(sigmas); NSR_CHECK_PTR(rgbs); NSR_CHECK_PTR(deltas); NSR_CHECK_PTR(rays); NSR_CHECK_PTR(weights_sum);
    NSR_CHECK_PTR(image); NSR_CHECK_PTR(grad_sigmas); NSR_CHECK_PTR(grad_rgbs);
    if (C < 3 || C > CP_MAXC) return NSR_ERR_UNSUPPORTED;
    CompArgs a = {};
    a.sigmas = sigmas; a.rgbs = rgbs; a.deltas = deltas; a.rays = rays; a.M = M; a.N = N; a.C = C; a.T_thresh = T_thresh;
    a.g_ws = grad_weights_sum; a.g_rgb_map = grad_rgb_map; a.g_classes = grad_classes; a.ws_in = weights_sum; a.image_in = image;
    a.g_sigmas = grad_sigmas; a.g_rgbs = grad_rgbs; a.zero_fill = 1; a.epilogue = 1;
    return cp_launch<true>(a, (hipStream_t)stream);
}

uint64_t nsr_recon_loss_workspace_bytes(uint32_t N) {
    (void)N;
    return 2048ull * 2 * sizeof(float);
}

int nsr_recon_loss(const float *rgb_map, const float *classes, uint32_t N, uint32_t nc, const float *target_rgb,
                   const int64_t *target_cls, const int64_t *pix, float ce_lambda, float factor, const float *scale,
                   float *grad_rgb_map, float *grad_classes, float *loss_out, void *workspace, nsr_stream_t stream) {
    NSR_CHECK_PTR(loss_out);
    if (N == 0) return NSR_ERR_INVALID_ARG;
    NSR_CHECK_PTR(rgb_map); NSR_CHECK_PTR(target_rgb); NSR_CHECK_PTR(grad_rgb_map); NSR_CHECK_PTR(workspace);
    const bool with_ce = classes != nullptr && nc > 0 && ce_lambda != 0.0f;
    if (with_ce && (target_cls == nullptr || grad_classes == nullptr)) return NSR_ERR_INVALID_ARG;
    LossArgs a;
    a.rgb_map = rgb_map; a.classes = with_ce ? classes : nullptr; a.target_rgb = target_rgb; a.target_cls = target_cls; a.pix = pix;
    a.N = N; a.nc = with_ce ? nc : 0u; a.ce_lambda = ce_lambda; a.factor = factor; a.scale = scale;
    a.g_rgb = grad_rgb_map; a.g_classes = grad_classes; a.partials = (float *)workspace; a.out = loss_out;
    uint32_t nb = nsr_div_up(N, CP_BLOCK);
    if (nb > 2048u) nb = 2048u;
    a.nblocks = nb;
    hipLaunchKernelGGL(k_recon_loss, dim3(nb), dim3(CP_BLOCK), 0, (hipStream_t)stream, a);
    hipLaunchKernelGGL(k_recon_loss_final, dim3(1), dim3(CP_BLOCK), 0, (hipStream_t)stream, a);
    return nsr_launch_status();
}

}   // extern "C"
