// Stand-alone bias-free fully fused MLP on MFMA (gfx950): the C-ABI replacement of the
// tcnn.Network modules the reference builds in networks/style_nerf.py:44-98.  The hot path uses
// the fused field kernels; these serve `Network` used as a module of its own and share the same
// MFMA chain (mfma_tiles.h): weights = A operand, 16 samples of a tile on lane&15, activations
// stay in registers between layers.
//
// Supported shapes (the ones the reference instantiates): n_in in {16, 32}, n_neurons = 64,
// n_hidden_layers in {1, 2}, n_out <= 16, output activation None | Sigmoid.
#include "mfma_tiles.h"

struct MlpArgs {
    const float *x;
    const float *params;
    const float *y;
    const float *dy;
    float *out;      // y (forward) or dx (backward)
    float *dparams;
    uint32_t M, n_out, tiles_per_block;
    int out_act;
};

template <int IN, int NH> struct MlpShape {
    static constexpr int P_W1 = 0;                       // [64 x IN]
    static constexpr int P_W2 = 64 * IN;                 // [64 x 64] when NH == 2
    static constexpr int P_WO = 64 * IN + (NH == 2 ? 4096 : 0);   // [16 x 64]
    static constexpr int P_TOTAL = P_WO + 1024;
};

__device__ __forceinline__ float mlp_sigmoid(float x) { return 1.0f / (1.0f + expf(-x)); }

template <int CD, int IN, int NH>
__device__ __forceinline__ void mlp_build_fw(short *lds, const float *__restrict__ p) {
    using S = MlpShape<IN, NH>;
    mm_build_frags<CD>(lds + S::P_W1, p + S::P_W1, 64, IN, 4, IN, false, 0, IN == 32);
    if (NH == 2) mm_build_frags<CD>(lds + S::P_W2, p + S::P_W2, 64, 64, 4, 64, false, 0, true);
    mm_build_frags<CD>(lds + S::P_WO, p + S::P_WO, 16, 64, 1, 64, false, 0, true);
}
template <int CD, int IN, int NH>
__device__ __forceinline__ void mlp_build_bw(short *lds, const float *__restrict__ p) {
    using S = MlpShape<IN, NH>;
    // W1^T [IN x 64]: IN/16 m-tiles, K = 64; W2^T [64 x 64]; WO^T [64 x 16]: 4 frag16
    mm_build_frags<CD>(lds + S::P_W1, p + S::P_W1, 64, IN, IN / 16, 64, true, 0, true);
    if (NH == 2) mm_build_frags<CD>(lds + S::P_W2, p + S::P_W2, 64, 64, 4, 64, true, 0, true);
    mm_build_frags<CD>(lds + S::P_WO, p + S::P_WO, 16, 64, 4, 16, true, 0, false);
}

// loads this lane's features of sample m: k-block t holds features 16t + 4g + e
template <int CD>
__device__ __forceinline__ s4v mlp_load4(const float *x, uint32_t m, int n_in, int t, int g, bool valid) {
    s4v r = {0, 0, 0, 0};
    if (valid) {
        const float4 v = *reinterpret_cast<const float4 *>(x + (size_t)m * n_in + 16 * t + 4 * g);
        r[0] = MM<CD>::cvt(v.x); r[1] = MM<CD>::cvt(v.y); r[2] = MM<CD>::cvt(v.z); r[3] = MM<CD>::cvt(v.w);
    }
    return r;
}

template <int CD, int IN, int NH>
__device__ __forceinline__ void mlp_chain_fwd(const short *wl, int lane, s4v x0, s4v x1, s8v (&h1)[2], s8v (&h2)[2], f4v (&o)[1]) {
    using S = MlpShape<IN, NH>;
    f4v h[4];
    if (IN == 32) {
        const s8v b[1] = {mm_cat(x0, x1)};
        mm_layer32<CD, 4, 1>(wl + S::P_W1, lane, b, h);
    } else {
        mm_layer16<CD, 4>(wl + S::P_W1, lane, x0, h);
    }
    mm_pack64<CD, true>(h, h1);
    if (NH == 2) {
        mm_layer32<CD, 4, 2>(wl + S::P_W2, lane, h1, h);
        mm_pack64<CD, true>(h, h2);
        mm_layer32<CD, 1, 2>(wl + S::P_WO, lane, h2, o);
    } else {
        mm_layer32<CD, 1, 2>(wl + S::P_WO, lane, h1, o);
    }
}

template <int CD, int IN, int NH>
__global__ void __launch_bounds__(256)
k_mlp_fwd(MlpArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    short *wl = reinterpret_cast<short *>(smem);
    mlp_build_fw<CD, IN, NH>(wl, a.params);
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, s = lane & 15, g = lane >> 4;
    const uint32_t ntiles = (a.M + 15) / 16;
    const uint32_t t_begin = blockIdx.x * a.tiles_per_block, t_end = min(t_begin + a.tiles_per_block, ntiles);
    for (uint32_t tile = t_begin + wave; tile < t_end; tile += 4) {
        const uint32_t m = tile * 16 + s;
        const bool valid = m < a.M;
        const s4v x0 = mlp_load4<CD>(a.x, m, IN, 0, g, valid);
        s4v x1 = {0, 0, 0, 0};
        if (IN == 32) x1 = mlp_load4<CD>(a.x, m, IN, 1, g, valid);
        s8v h1[2], h2[2];
        f4v o[1];
        mlp_chain_fwd<CD, IN, NH>(wl, lane, x0, x1, h1, h2, o);
        if (valid) {
#pragma unroll
            for (int e = 0; e < 4; e++) {
                const uint32_t ch = 4 * g + e;
                if (ch < a.n_out) a.out[(size_t)m * a.n_out + ch] = a.out_act == NSR_ACT_SIGMOID ? mlp_sigmoid(o[0][e]) : o[0][e];
            }
        }
    }
}

template <int CD, int NG, int NA>
__device__ __forceinline__ void mlp_wgrad(float *lds_w, int in_p, int row_hi, const s4v (&Gt)[NG], const s4v (&At)[NA], int lane) {
    const int i = lane & 15, g = lane >> 4;
#pragma unroll
    for (int ot = 0; ot < NG; ot++) {
#pragma unroll
        for (int it = 0; it < NA; it++) {
            f4v z = {0.f, 0.f, 0.f, 0.f};
            const f4v d = MM<CD>::k16(Gt[ot], At[it], z);
#pragma unroll
            for (int e = 0; e < 4; e++) {
                const int row = 16 * ot + 4 * g + e;
                if (row < row_hi) atomicAdd(&lds_w[row * in_p + 16 * it + i], d[e]);
            }
        }
    }
}

template <int CD>
__device__ __forceinline__ void mlp_tr4(const s8v (&x)[2], s4v ident, s4v (&out)[4]) {
    out[0] = mm_transpose16<CD>(mm_lo(x[0]), ident);
    out[1] = mm_transpose16<CD>(mm_hi(x[0]), ident);
    out[2] = mm_transpose16<CD>(mm_lo(x[1]), ident);
    out[3] = mm_transpose16<CD>(mm_hi(x[1]), ident);
}
template <int CD>
__device__ __forceinline__ void mlp_mask_pack(const f4v (&gacc)[4], const s8v (&act)[2], s8v (&out)[2]) {
    out[0] = mm_cat(mm_round4<CD, false>(mm_relu_mask(gacc[0], mm_lo(act[0]))),
                    mm_round4<CD, false>(mm_relu_mask(gacc[1], mm_hi(act[0]))));
    out[1] = mm_cat(mm_round4<CD, false>(mm_relu_mask(gacc[2], mm_lo(act[1]))),
                    mm_round4<CD, false>(mm_relu_mask(gacc[3], mm_hi(act[1]))));
}

template <int CD, int IN, int NH>
__global__ void __launch_bounds__(256)
k_mlp_bwd(MlpArgs a) {
    using S = MlpShape<IN, NH>;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    short *wl = reinterpret_cast<short *>(smem);
    short *wt = wl + S::P_TOTAL;
    float *wg = reinterpret_cast<float *>(smem + (size_t)S::P_TOTAL * 4);
    mlp_build_fw<CD, IN, NH>(wl, a.params);
    mlp_build_bw<CD, IN, NH>(wt, a.params);
    for (int i = threadIdx.x; i < S::P_TOTAL; i += 256) wg[i] = 0.0f;
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, s = lane & 15, g = lane >> 4;
    const s4v ident = mm_identity_frag<CD>(lane);
    const uint32_t ntiles = (a.M + 15) / 16;
    const uint32_t t_begin = blockIdx.x * a.tiles_per_block, t_end = min(t_begin + a.tiles_per_block, ntiles);
    for (uint32_t tile = t_begin + wave; tile < t_end; tile += 4) {
        const uint32_t m = tile * 16 + s;
        const bool valid = m < a.M;
        const s4v x0 = mlp_load4<CD>(a.x, m, IN, 0, g, valid);
        s4v x1 = {0, 0, 0, 0};
        if (IN == 32) x1 = mlp_load4<CD>(a.x, m, IN, 1, g, valid);
        s8v h1[2], h2[2];
        f4v o[1];
        mlp_chain_fwd<CD, IN, NH>(wl, lane, x0, x1, h1, h2, o);
        // upstream gradient -> B fragment (row = 4g + e)
        s4v dyf;
#pragma unroll
        for (int e = 0; e < 4; e++) {
            const uint32_t ch = 4 * g + e;
            float gv = 0.0f;
            if (valid && ch < a.n_out) {
                gv = a.dy[(size_t)m * a.n_out + ch];
                if (a.out_act == NSR_ACT_SIGMOID) {
                    const float sg = mlp_sigmoid(o[0][e]);
                    gv *= sg * (1.0f - sg);
                }
            }
            dyf[e] = MM<CD>::cvt(gv);
        }
        f4v h[4];
        s8v gl[2], g1[2];
        mm_layer16<CD, 4>(wt + S::P_WO, lane, dyf, h);
        if (NH == 2) {
            mlp_mask_pack<CD>(h, h2, gl);
            mm_layer32<CD, 4, 2>(wt + S::P_W2, lane, gl, h);
            mlp_mask_pack<CD>(h, h1, g1);
        } else {
            mlp_mask_pack<CD>(h, h1, gl);
            g1[0] = gl[0]; g1[1] = gl[1];
        }
        f4v dx[IN / 16];
        mm_layer32<CD, IN / 16, 2>(wt + S::P_W1, lane, g1, dx);
        if (a.out && valid) {
#pragma unroll
            for (int t = 0; t < IN / 16; t++)
                *reinterpret_cast<float4 *>(a.out + (size_t)m * IN + 16 * t + 4 * g) = make_float4(dx[t][0], dx[t][1], dx[t][2], dx[t][3]);
        }
        // wgrads
        if (a.dparams) {
            s4v glt[4], g1t[4], h1t[4];
            const s4v dyt[1] = {mm_transpose16<CD>(dyf, ident)};
            mlp_tr4<CD>(gl, ident, glt);
            mlp_tr4<CD>(h1, ident, h1t);
            if (NH == 2) {
                s4v h2t[4];
                mlp_tr4<CD>(h2, ident, h2t);
                mlp_wgrad<CD, 1, 4>(wg + S::P_WO, 64, (int)a.n_out, dyt, h2t, lane);
                mlp_wgrad<CD, 4, 4>(wg + S::P_W2, 64, 64, glt, h1t, lane);
                mlp_tr4<CD>(g1, ident, g1t);
            } else {
                mlp_wgrad<CD, 1, 4>(wg + S::P_WO, 64, (int)a.n_out, dyt, h1t, lane);
#pragma unroll
                for (int q = 0; q < 4; q++) g1t[q] = glt[q];
            }
            if (IN == 32) {
                const s4v xt[2] = {mm_transpose16<CD>(x0, ident), mm_transpose16<CD>(x1, ident)};
                mlp_wgrad<CD, 4, 2>(wg + S::P_W1, 32, 64, g1t, xt, lane);
            } else {
                const s4v xt[1] = {mm_transpose16<CD>(x0, ident)};
                mlp_wgrad<CD, 4, 1>(wg + S::P_W1, 16, 64, g1t, xt, lane);
            }
        }
    }
    __syncthreads();
    if (a.dparams) {
        for (int i = threadIdx.x; i < S::P_TOTAL; i += 256) {
            const float v = wg[i];
            if (v != 0.0f) atomicAdd(a.dparams + i, v);
        }
    }
}

template <int CD, int IN, int NH>
static int mlp_launch(MlpArgs &a, bool backward, hipStream_t s) {
    using S = MlpShape<IN, NH>;
    const uint32_t ntiles = (a.M + 15) / 16;
    uint32_t nb = (ntiles + 31) / 32;
    if (nb > 2048) nb = 2048;
    if (nb == 0) nb = 1;
    a.tiles_per_block = (ntiles + nb - 1) / nb;
    if (!backward) {
        hipLaunchKernelGGL((k_mlp_fwd<CD, IN, NH>), dim3(nb), dim3(256), (size_t)S::P_TOTAL * 2, s, a);
    } else {
        const size_t lds = (size_t)S::P_TOTAL * 8;   // fw + bw images (2 B each) + fp32 wgrad
        hipLaunchKernelGGL((k_mlp_bwd<CD, IN, NH>), dim3(nb), dim3(256), lds, s, a);
    }
    return nsr_launch_status();
}

static int mlp_dispatch(MlpArgs &a, uint32_t n_in, uint32_t n_hidden, int cd, bool backward, hipStream_t s) {
#define NSR_MLP_CASE(CDV, INV, NHV) \
    if (cd == CDV && n_in == INV && n_hidden == NHV) return mlp_launch<CDV, INV, NHV>(a, backward, s)
    NSR_MLP_CASE(NSR_F16, 32, 1); NSR_MLP_CASE(NSR_F16, 32, 2); NSR_MLP_CASE(NSR_F16, 16, 1); NSR_MLP_CASE(NSR_F16, 16, 2);
    NSR_MLP_CASE(NSR_BF16, 32, 1); NSR_MLP_CASE(NSR_BF16, 32, 2); NSR_MLP_CASE(NSR_BF16, 16, 1); NSR_MLP_CASE(NSR_BF16, 16, 2);
#undef NSR_MLP_CASE
    return NSR_ERR_UNSUPPORTED;
}

extern "C" {

uint32_t nsr_mlp_param_count(uint32_t n_in, uint32_t n_out, uint32_t n_neurons, uint32_t n_hidden_layers) {
    const uint32_t in_p = (n_in + 15) / 16 * 16, out_p = (n_out + 15) / 16 * 16;
    if (n_hidden_layers == 0) return out_p * in_p;
    return n_neurons * in_p + (n_hidden_layers - 1) * n_neurons * n_neurons + out_p * n_neurons;
}

int nsr_mlp_forward(const float *x, const float *params, uint32_t M, uint32_t n_in, uint32_t n_out, uint32_t n_neurons,
                    uint32_t n_hidden_layers, int out_act, int compute_dtype, float *y, nsr_stream_t stream) {
    if (M == 0) return NSR_OK;
    NSR_CHECK_PTR(x); NSR_CHECK_PTR(params); NSR_CHECK_PTR(y);
    if (n_neurons != 64 || n_out == 0 || n_out > 16) return NSR_ERR_UNSUPPORTED;
    if (out_act != NSR_ACT_NONE && out_act != NSR_ACT_SIGMOID) return NSR_ERR_INVALID_ARG;
    if ((uintptr_t)x & 15u) return NSR_ERR_INVALID_ARG;
    MlpArgs a{};
    a.x = x; a.params = params; a.out = y; a.M = M; a.n_out = n_out; a.out_act = out_act;
    return mlp_dispatch(a, n_in, n_hidden_layers, compute_dtype, false, (hipStream_t)stream);
}

int nsr_mlp_backward(const float *x, const float *params, const float *y, const float *dy, uint32_t M, uint32_t n_in,
                     uint32_t n_out, uint32_t n_neurons, uint32_t n_hidden_layers, int out_act, int compute_dtype, float *dx,
                     float *dparams, nsr_stream_t stream) {
    if (M == 0) return NSR_OK;
    NSR_CHECK_PTR(x); NSR_CHECK_PTR(params); NSR_CHECK_PTR(dy);
    (void)y;   // the forward output is recomputed together with the hidden activations
    if (n_neurons != 64 || n_out == 0 || n_out > 16) return NSR_ERR_UNSUPPORTED;
    if (out_act != NSR_ACT_NONE && out_act != NSR_ACT_SIGMOID) return NSR_ERR_INVALID_ARG;
    if (((uintptr_t)x & 15u) || (dx && ((uintptr_t)dx & 15u))) return NSR_ERR_INVALID_ARG;
    MlpArgs a{};
    a.x = x; a.params = params; a.dy = dy; a.out = dx; a.dparams = dparams; a.M = M; a.n_out = n_out; a.out_act = out_act;
    return mlp_dispatch(a, n_in, n_hidden_layers, compute_dtype, true, (hipStream_t)stream);
}

}   // extern "C"
