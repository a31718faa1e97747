// Fused Adam (+ GradScaler unscale, + EMA shadow, + grad zero-fill, + half copy) over one flat
// fp32 arena, gfx950.  Replaces torch.optim.Adam(eps=1e-15) + zero_grad + torch_ema on the hot
// path (trainers/base.py:216-229,420-426, utils/__init__.py:116-142): one streaming pass of
// 16-byte accesses instead of ~10 elementwise launches over 25 M parameters.
#include "nsr_common.h"

struct AdamArgs {
    float *p, *g, *m, *v, *ema;
    _Float16 *half_copy;
    uint64_t n;
    float beta1, beta2, eps, step_size, inv_sqrt_bc2, grad_scale_inv, ema_decay;
    uint32_t mask4;
};

__device__ __forceinline__ float adam_one(float &p, float g, float &m, float &v, const AdamArgs &a) {
    g *= a.grad_scale_inv;
    m = a.beta1 * m + (1.0f - a.beta1) * g;          // torch: exp_avg.lerp_(grad, 1 - beta1)
    v = a.beta2 * v + (1.0f - a.beta2) * g * g;      // exp_avg_sq.mul_(beta2).addcmul_(g, g, 1 - beta2)
    const float denom = sqrtf(v) * a.inv_sqrt_bc2 + a.eps;
    p -= a.step_size * (m / denom);
    return p;
}

__global__ void __launch_bounds__(256)
k_adam(AdamArgs a) {
    const uint64_t n4 = a.n / 4;
    for (uint64_t i = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; i < n4; i += (uint64_t)gridDim.x * blockDim.x) {
        float4 p = reinterpret_cast<float4 *>(a.p)[i];
        const float4 g = reinterpret_cast<float4 *>(a.g)[i];
        float4 m = reinterpret_cast<float4 *>(a.m)[i];
        float4 v = reinterpret_cast<float4 *>(a.v)[i];
        if (a.mask4 & 1u) adam_one(p.x, g.x, m.x, v.x, a);
        if (a.mask4 & 2u) adam_one(p.y, g.y, m.y, v.y, a);
        if (a.mask4 & 4u) adam_one(p.z, g.z, m.z, v.z, a);
        if (a.mask4 & 8u) adam_one(p.w, g.w, m.w, v.w, a);
        reinterpret_cast<float4 *>(a.p)[i] = p;
        reinterpret_cast<float4 *>(a.m)[i] = m;
        reinterpret_cast<float4 *>(a.v)[i] = v;
        reinterpret_cast<float4 *>(a.g)[i] = make_float4(0.f, 0.f, 0.f, 0.f);
        if (a.ema) {
            float4 e = reinterpret_cast<float4 *>(a.ema)[i];
            const float k = 1.0f - a.ema_decay;   // torch_ema: shadow.sub_((1 - decay) * (shadow - param))
            e.x -= k * (e.x - p.x); e.y -= k * (e.y - p.y); e.z -= k * (e.z - p.z); e.w -= k * (e.w - p.w);
            reinterpret_cast<float4 *>(a.ema)[i] = e;
        }
        if (a.half_copy) {
            typedef _Float16 h4 __attribute__((ext_vector_type(4)));
            h4 h;
            h[0] = (_Float16)p.x; h[1] = (_Float16)p.y; h[2] = (_Float16)p.z; h[3] = (_Float16)p.w;
            reinterpret_cast<h4 *>(a.half_copy)[i] = h;
        }
    }
    if (blockIdx.x == 0 && threadIdx.x < (a.n & 3u)) {
        const uint64_t i = n4 * 4 + threadIdx.x;
        float p = a.p[i], m = a.m[i], v = a.v[i];
        if (a.mask4 & (1u << (i & 3u))) adam_one(p, a.g[i], m, v, a);
        a.p[i] = p; a.m[i] = m; a.v[i] = v; a.g[i] = 0.0f;
        if (a.ema) a.ema[i] -= (1.0f - a.ema_decay) * (a.ema[i] - p);
        if (a.half_copy) a.half_copy[i] = (_Float16)p;
    }
}

extern "C" int nsr_adam_step(float *params, float *grads, float *exp_avg, float *exp_avg_sq, float *ema, void *half_copy,
                             uint64_t n, float lr, float beta1, float beta2, float eps, float grad_scale_inv, float ema_decay,
                             uint32_t step, uint32_t elem_mask4, nsr_stream_t stream) {
    if (n == 0) return NSR_OK;
    NSR_CHECK_PTR(params); NSR_CHECK_PTR(grads); NSR_CHECK_PTR(exp_avg); NSR_CHECK_PTR(exp_avg_sq);
    if (step == 0) return NSR_ERR_INVALID_ARG;
    const uintptr_t al = (uintptr_t)params | (uintptr_t)grads | (uintptr_t)exp_avg | (uintptr_t)exp_avg_sq |
                         (uintptr_t)ema | (uintptr_t)half_copy;
    if (al & 15u) return NSR_ERR_INVALID_ARG;
    AdamArgs a;
    a.p = params; a.g = grads; a.m = exp_avg; a.v = exp_avg_sq; a.ema = ema; a.half_copy = (_Float16 *)half_copy; a.n = n;
    a.mask4 = elem_mask4 & 0xFu;
    a.beta1 = beta1; a.beta2 = beta2; a.eps = eps; a.grad_scale_inv = grad_scale_inv; a.ema_decay = ema_decay;
    const double bc1 = 1.0 - pow((double)beta1, (double)step), bc2 = 1.0 - pow((double)beta2, (double)step);
    a.step_size = (float)((double)lr / bc1);
    a.inv_sqrt_bc2 = (float)(1.0 / sqrt(bc2));
    hipLaunchKernelGGL(k_adam, dim3(nsr_grid_1d(n / 4 + 1, 256)), dim3(256), 0, (hipStream_t)stream, a);
    return nsr_launch_status();
}
