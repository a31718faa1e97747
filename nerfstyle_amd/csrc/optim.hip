// Fused Adam (+ GradScaler unscale, + EMA shadow, + grad zero-fill, + half copy) over one flat
// fp32 arena, gfx950.  Replaces torch.optim.Adam(eps=1e-15) + zero_grad + torch_ema on the hot
// path (trainers/base.py:216-229,420-426, utils/__init__.py:116-142): one streaming pass of
// 16-byte accesses instead of ~10 elementwise launches over 25 M parameters.
#include "nsr_common.h"

struct AdamArgs {
    float *p, *g, *m, *v, *ema;
    _Float16 *half_copy;
    uint64_t n, half_n;              // half_copy covers the first half_n elements (the tables of a whole-arena launch)
    float beta1, beta2, eps, step_size, inv_sqrt_bc2, grad_scale_inv, ema_decay;
    uint32_t mask4;
    const uint32_t *dyn;             // device-side scaler state (nsr_scaler_update) or NULL: host-side scalars above
};

// Device-side GradScaler + step bookkeeping (torch.cuda.amp.GradScaler's policy, trainers/base.py:228,420-425), 16 words:
//   [0] f32 scale   [1] i32 growth tracker   [2] u32 found_inf (set by k_grad_check, consumed by k_scaler_update)
//   [3] u32 optimiser steps taken (skipped steps do not count: bias corrections and the LambdaLR schedule follow it)
//   [4] u32 steps skipped so far   [5] u32 EMA updates made (torch_ema's num_updates)   [6..7] reserved
//   [8] u32 skip this step   [9] f32 step_size = lr / (1 - beta1^t)   [10] f32 1 / sqrt(1 - beta2^t)
//   [11] f32 1 / scale (the scale the gradients carry)   [12] f32 lr   [13] f32 EMA decay of this step   [14..15] reserved
enum { SC_SCALE = 0, SC_TRACKER = 1, SC_FOUND = 2, SC_STEP = 3, SC_SKIPPED = 4, SC_EMA_N = 5, SC_SKIP = 8, SC_STEP_SIZE = 9,
       SC_INV_BC2 = 10, SC_INV_SCALE = 11, SC_LR = 12, SC_EMA_DECAY = 13 };

__device__ __forceinline__ float adam_one(float &p, float g, float &m, float &v, const AdamArgs &a) {
    g *= a.grad_scale_inv;
    m = a.beta1 * m + (1.0f - a.beta1) * g;          // torch: exp_avg.lerp_(grad, 1 - beta1)
    v = a.beta2 * v + (1.0f - a.beta2) * g * g;      // exp_avg_sq.mul_(beta2).addcmul_(g, g, 1 - beta2)
    const float denom = sqrtf(v) * a.inv_sqrt_bc2 + a.eps;
    p -= a.step_size * (m / denom);
    return p;
}

// any non-finite value among the trained elements -> found_inf (GradScaler.unscale_'s check, one streaming pass)
__global__ void __launch_bounds__(256)
k_grad_check(const float *__restrict__ g, uint64_t n, uint32_t mask4, uint32_t *__restrict__ found) {
    const uint64_t n4 = n / 4;
    uint32_t bad = 0;
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    uint64_t i = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x;
    // exponent all ones <=> inf or nan
    auto chk = [&](const float4 &v) {
        const uint32_t e0 = __float_as_uint(v.x) & 0x7F800000u, e1 = __float_as_uint(v.y) & 0x7F800000u;
        const uint32_t e2 = __float_as_uint(v.z) & 0x7F800000u, e3 = __float_as_uint(v.w) & 0x7F800000u;
        bad |= ((mask4 & 1u) && e0 == 0x7F800000u) | ((mask4 & 2u) && e1 == 0x7F800000u) |
               ((mask4 & 4u) && e2 == 0x7F800000u) | ((mask4 & 8u) && e3 == 0x7F800000u);
    };
    for (; i + 3 * stride < n4; i += 4 * stride) {          // four 16-byte loads in flight per lane
        const float4 a0 = reinterpret_cast<const float4 *>(g)[i], a1 = reinterpret_cast<const float4 *>(g)[i + stride];
        const float4 a2 = reinterpret_cast<const float4 *>(g)[i + 2 * stride], a3 = reinterpret_cast<const float4 *>(g)[i + 3 * stride];
        chk(a0); chk(a1); chk(a2); chk(a3);
    }
    for (; i < n4; i += stride) chk(reinterpret_cast<const float4 *>(g)[i]);
    if (blockIdx.x == 0 && threadIdx.x < (n & 3u)) {
        const uint64_t j = n4 * 4 + threadIdx.x;
        if ((mask4 & (1u << (j & 3u))) && (__float_as_uint(g[j]) & 0x7F800000u) == 0x7F800000u) bad = 1;
    }
    if (__ballot(bad != 0) != 0ull && (threadIdx.x & 63u) == 0) atomicOr(found, 1u);
}

// one thread: GradScaler.step's decision + GradScaler.update + the step counter, the LambdaLR value and Adam's bias
// corrections (double precision, as torch computes them on the host)
__global__ void k_scaler_update(uint32_t *st, float lr_base, float lr_decay_steps, float beta1, float beta2, float growth,
                                float backoff, uint32_t growth_interval, int enabled, float ema_decay_max) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    if (ema_decay_max >= 0.0f) {
        // torch_ema (utils/__init__.py:116-142): num_updates += 1; decay = min(decay, (1 + n) / (10 + n)) -- on every call,
        // skipped optimiser steps included (ema.update() is unconditional, trainers/base.py:426)
        const uint32_t n = st[SC_EMA_N] + 1u;
        st[SC_EMA_N] = n;
        st[SC_EMA_DECAY] = __float_as_uint(fminf(ema_decay_max, (1.0f + (float)n) / (10.0f + (float)n)));
    }
    float scale = __uint_as_float(st[SC_SCALE]);
    const bool inf = enabled && st[SC_FOUND] != 0u;
    st[SC_FOUND] = 0u;
    st[SC_INV_SCALE] = __float_as_uint(enabled ? 1.0f / scale : 1.0f);         // the scale THESE gradients carry
    st[SC_SKIP] = inf ? 1u : 0u;
    if (inf) {
        st[SC_SKIPPED] += 1u;
        st[SC_SCALE] = __float_as_uint(scale * backoff);
        st[SC_TRACKER] = 0u;
        return;
    }
    const uint32_t t = st[SC_STEP] + 1u;
    st[SC_STEP] = t;
    // LambdaLR: k scheduler steps have been taken before optimiser step k + 1 (trainers/base.py:223-226,424-425)
    const double lr = lr_decay_steps > 0.0f ? (double)lr_base * pow(0.1, (double)(t - 1u) / (double)lr_decay_steps) : (double)lr_base;
    const double bc1 = 1.0 - pow((double)beta1, (double)t), bc2 = 1.0 - pow((double)beta2, (double)t);
    st[SC_LR] = __float_as_uint((float)lr);
    st[SC_STEP_SIZE] = __float_as_uint((float)(lr / bc1));
    st[SC_INV_BC2] = __float_as_uint((float)(1.0 / sqrt(bc2)));
    if (enabled) {
        const uint32_t tr = st[SC_TRACKER] + 1u;
        if (tr >= growth_interval) {
            st[SC_SCALE] = __float_as_uint(scale * growth);
            st[SC_TRACKER] = 0u;
        } else {
            st[SC_TRACKER] = tr;
        }
    }
}

__global__ void __launch_bounds__(256)
k_adam(AdamArgs a) {
    const uint64_t n4 = a.n / 4;
    bool skip = false;
    if (a.dyn) {
        skip = a.dyn[SC_SKIP] != 0u;
        a.step_size = __uint_as_float(a.dyn[SC_STEP_SIZE]);
        a.inv_sqrt_bc2 = __uint_as_float(a.dyn[SC_INV_BC2]);
        a.grad_scale_inv = __uint_as_float(a.dyn[SC_INV_SCALE]);
        if (a.ema) a.ema_decay = __uint_as_float(a.dyn[SC_EMA_DECAY]);
    }
    if (skip) {
        // GradScaler skipped optimizer.step(): parameters and moments stay; the gradient is cleared (the reference's
        // zero_grad at the top of the next iteration) and the EMA still moves (ema.update() is unconditional, base.py:426)
        for (uint64_t i = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; i < n4; i += (uint64_t)gridDim.x * blockDim.x) {
            reinterpret_cast<float4 *>(a.g)[i] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (a.ema) {
                const float4 p = reinterpret_cast<float4 *>(a.p)[i];
                float4 e = reinterpret_cast<float4 *>(a.ema)[i];
                const float k = 1.0f - a.ema_decay;
                e.x -= k * (e.x - p.x); e.y -= k * (e.y - p.y); e.z -= k * (e.z - p.z); e.w -= k * (e.w - p.w);
                reinterpret_cast<float4 *>(a.ema)[i] = e;
            }
        }
        if (blockIdx.x == 0 && threadIdx.x < (a.n & 3u)) {
            const uint64_t i = n4 * 4 + threadIdx.x;
            a.g[i] = 0.0f;
            if (a.ema) a.ema[i] -= (1.0f - a.ema_decay) * (a.ema[i] - a.p[i]);
        }
        return;
    }
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
        if (a.half_copy && i * 4 < a.half_n) {
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
        if (a.half_copy && i < a.half_n) a.half_copy[i] = (_Float16)p;
    }
}

static int adam_launch(AdamArgs &a, hipStream_t s) {
    hipLaunchKernelGGL(k_adam, dim3(nsr_grid_1d(a.n / 4 + 1, 256)), dim3(256), 0, s, a);
    return nsr_launch_status();
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
    a.half_n = n; a.dyn = nullptr;
    a.mask4 = elem_mask4 & 0xFu;
    a.beta1 = beta1; a.beta2 = beta2; a.eps = eps; a.grad_scale_inv = grad_scale_inv; a.ema_decay = ema_decay;
    const double bc1 = 1.0 - pow((double)beta1, (double)step), bc2 = 1.0 - pow((double)beta2, (double)step);
    a.step_size = (float)((double)lr / bc1);
    a.inv_sqrt_bc2 = (float)(1.0 / sqrt(bc2));
    return adam_launch(a, (hipStream_t)stream);
}

extern "C" int nsr_grad_check(const float *grads, uint64_t n, uint32_t elem_mask4, void *scaler_state, nsr_stream_t stream) {
    if (n == 0) return NSR_OK;
    NSR_CHECK_PTR(grads); NSR_CHECK_PTR(scaler_state);
    if (((uintptr_t)grads & 15u) || ((uintptr_t)scaler_state & 3u)) return NSR_ERR_INVALID_ARG;
    uint64_t blocks = (n / 4 + 256 * 4 - 1) / (256 * 4);
    if (blocks > 2048) blocks = 2048;
    if (blocks == 0) blocks = 1;
    hipLaunchKernelGGL(k_grad_check, dim3((uint32_t)blocks), dim3(256), 0, (hipStream_t)stream, grads, n, elem_mask4 & 0xFu,
                       (uint32_t *)scaler_state + SC_FOUND);
    return nsr_launch_status();
}

extern "C" int nsr_scaler_update(void *scaler_state, float lr_base, float lr_decay_steps, float beta1, float beta2, float growth_factor,
                                 float backoff_factor, uint32_t growth_interval, int enabled, float ema_decay_max,
                                 nsr_stream_t stream) {
    NSR_CHECK_PTR(scaler_state);
    if (((uintptr_t)scaler_state & 3u) || growth_interval == 0) return NSR_ERR_INVALID_ARG;
    hipLaunchKernelGGL(k_scaler_update, dim3(1), dim3(64), 0, (hipStream_t)stream, (uint32_t *)scaler_state, lr_base, lr_decay_steps, beta1,
                       beta2, growth_factor, backoff_factor, growth_interval, enabled, ema_decay_max);
    return nsr_launch_status();
}

extern "C" int nsr_adam_step_scaled(float *params, float *grads, float *exp_avg, float *exp_avg_sq, float *ema, void *half_copy,
                                    uint64_t n, uint64_t half_n, float beta1, float beta2, float eps, uint32_t elem_mask4,
                                    const void *scaler_state, nsr_stream_t stream) {
    if (n == 0) return NSR_OK;
    NSR_CHECK_PTR(params); NSR_CHECK_PTR(grads); NSR_CHECK_PTR(exp_avg); NSR_CHECK_PTR(exp_avg_sq); NSR_CHECK_PTR(scaler_state);
    const uintptr_t al = (uintptr_t)params | (uintptr_t)grads | (uintptr_t)exp_avg | (uintptr_t)exp_avg_sq |
                         (uintptr_t)ema | (uintptr_t)half_copy;
    if ((al & 15u) || ((uintptr_t)scaler_state & 3u) || half_n > n || (half_n & 3u)) return NSR_ERR_INVALID_ARG;
    AdamArgs a;
    a.p = params; a.g = grads; a.m = exp_avg; a.v = exp_avg_sq; a.ema = ema; a.half_copy = (_Float16 *)half_copy; a.n = n;
    a.half_n = half_n; a.dyn = (const uint32_t *)scaler_state;
    a.mask4 = elem_mask4 & 0xFu;
    a.beta1 = beta1; a.beta2 = beta2; a.eps = eps; a.ema_decay = 0.0f;           // (read from the state in the kernel)
    a.step_size = 0.0f; a.inv_sqrt_bc2 = 1.0f; a.grad_scale_inv = 1.0f;
    return adam_launch(a, (hipStream_t)stream);
}
