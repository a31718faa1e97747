// Fused field kernels for gfx950: BBox.normalize -> hash encode (both tables, one interleaved
// gather) -> density / color1 / color2 / class MLPs on MFMA -> trunc_exp / sigmoid / cat, in one
// launch per direction (reference: networks/style_nerf.py:120-142 with use_dir=False, which is
// 2 encoder launches + 4 tcnn launches + exp + cat, each round-tripping [M,.] through HBM).
//
// Work decomposition: a wave owns a tile of 16 consecutive samples (consecutive samples of one
// ray share cells on the coarse levels, so their gathers coalesce).  Lane (s = lane&15,
// g = lane>>4) encodes levels {2g, 2g+1, 8+2g, 9+2g} of sample s for BOTH encoders -- exactly
// the 8 elements the K=32 MFMA B fragment wants from that lane (features 4g..4g+3 of k-block 0
// and of k-block 1) -- so encode output feeds the matrix cores with no data movement, and in the
// backward the MFMA-produced input gradient lands on the lane that owns those levels' scatter.
// The forward kernel keeps all 32 table gathers of a lane's four levels in flight before it consumes the first (125
// instead of 74 VGPRs, still 4 waves per SIMD): 8.2 -> 8.0 ms on the bench frame.  Forward only -- the backward's re-gather
// path shares field_encode and has no registers to spare.
#ifndef NSR_FWD_BATCH_GATHER
#define NSR_FWD_BATCH_GATHER 1
#endif
#include "field_common.h"

template <typename TT, int CD, bool SIGMA_ONLY>
#ifndef NSR_FWD_WAVES_PER_EU
#define NSR_FWD_WAVES_PER_EU 0
#endif
__global__ void __launch_bounds__(256)
#if NSR_FWD_WAVES_PER_EU
__attribute__((amdgpu_waves_per_eu(NSR_FWD_WAVES_PER_EU, NSR_FWD_WAVES_PER_EU)))
#endif
k_field_fwd(FieldArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    short *wl = reinterpret_cast<short *>(smem);
    NsrLevel *lds_lv = reinterpret_cast<NsrLevel *>(smem + (SIGMA_ONLY ? FW_SIGMA_TOTAL : FW_TOTAL) * 2);
    field_build_fw<CD, SIGMA_ONLY>(wl, a.params);
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

    for (uint32_t tile = t_begin + wave; tile < t_end; tile += 4) {
        const uint32_t mpos = tile * 16 + s;
        const bool valid = mpos < Mc;
        // spatially ordered walk (nsr_sample_order): position `mpos` of the order is sample perm[mpos] of the buffers
        const uint32_t m = (a.perm && valid) ? a.perm[mpos] : mpos;
        float u0 = 0.f, u1 = 0.f, u2 = 0.f;
        if (valid) {
            u0 = field_unit(a.xyzs[(size_t)m * 3 + 0], a.bmin[0], a.bsize[0]);
            u1 = field_unit(a.xyzs[(size_t)m * 3 + 1], a.bmin[1], a.bsize[1]);
            u2 = field_unit(a.xyzs[(size_t)m * 3 + 2], a.bmin[2], a.bsize[2]);
        }
        // gridencoder.cu:107-132: inputs outside [0,1] encode to zeros
        const bool live = valid && (u0 >= 0 && u0 <= 1 && u1 >= 0 && u1 <= 1 && u2 >= 0 && u2 <= 1);   // NaN -> zeros too
        s8v xd, xc;
        field_encode<TT, CD, SIGMA_ONLY>(lds_lv, tables, u0, u1, u2, live, g, xd, xc, a.fast_levels);
        if (!SIGMA_ONLY && a.feats) {
            s8v *fo = reinterpret_cast<s8v *>(a.feats) + ((size_t)tile * 64 + lane) * 2;
            fo[0] = xd;
            fo[1] = xc;
        }

        // ---- density net: 32 -> 64 -> 1 ----------------------------------------------------
        f4v h[4];
        s8v hb[2];
        {
            const s8v b1[1] = {xd};
            mm_layer32<CD, 4, 1>(wl + FW_D1, lane, b1, h);
            mm_pack64<CD, true, true>(h, hb);
        }
        f4v o[1];
        mm_layer32<CD, 1, 2>(wl + FW_D2, lane, hb, o);
        if (valid && g == 0) a.sigmas[m] = expf(o[0][0]) * a.density_scale;   // tcnn_nerf.py:55-60, renderer.py:225
        if (SIGMA_ONLY) continue;

        // ---- class net: 32 -> 64 -> nc (rows 3..) -----------------------------------------
        f4v cls[1];
        {
            const s8v b1[1] = {xc};
            mm_layer32<CD, 4, 1>(wl + FW_K1, lane, b1, h);
            mm_pack64<CD, true, true>(h, hb);
            mm_layer32<CD, 1, 2>(wl + FW_K2, lane, hb, cls);
        }
        // ---- color1 net: 32 -> 64 -> 16 ----------------------------------------------------
        f4v c1[1];
        {
            const s8v b1[1] = {xc};
            mm_layer32<CD, 4, 1>(wl + FW_C1A, lane, b1, h);
            mm_pack64<CD, true, true>(h, hb);
            mm_layer32<CD, 1, 2>(wl + FW_C1B, lane, hb, c1);
        }
        // ---- color2 net: 16 -> 64 -> 64 -> 3, sigmoid --------------------------------------
        f4v rgb[1];
        {
            const s4v c1b = mm_round4<CD, false>(c1[0]);
            mm_layer16<CD, 4>(wl + FW_R1, lane, c1b, h);
            mm_pack64<CD, true, true>(h, hb);
            mm_layer32<CD, 4, 2>(wl + FW_R2, lane, hb, h);
            mm_pack64<CD, true, true>(h, hb);
            mm_layer32<CD, 1, 2>(wl + FW_R3, lane, hb, rgb);
        }
        // ---- cat(rgb, classes): channel ch = 4g + e (style_nerf.py:141) ---------------------
        if (valid) {
            float v[4];
#pragma unroll
            for (int e = 0; e < 4; e++) {
                const int ch = 4 * g + e;
                v[e] = ch < 3 ? field_sigmoid(rgb[0][e]) : cls[0][e];
            }
            float *dst = a.rgbs + (size_t)m * a.C_ch;
            if (a.C_ch == 8) {
                if (g < 2) reinterpret_cast<float4 *>(dst)[g] = make_float4(v[0], v[1], v[2], v[3]);
            } else {
#pragma unroll
                for (int e = 0; e < 4; e++)
                    if ((uint32_t)(4 * g + e) < a.C_ch) dst[4 * g + e] = v[e];
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------------------------
template <typename TT, int CD>
static int field_launch_fwd(const FieldArgs &a, uint32_t nblocks, bool sigma_only, hipStream_t s) {
    if (sigma_only) {
        const size_t lds = FW_SIGMA_TOTAL * 2 + 16 * sizeof(NsrLevel);
        hipLaunchKernelGGL((k_field_fwd<TT, CD, true>), dim3(nblocks), dim3(256), lds, s, a);
    } else {
        const size_t lds = FW_TOTAL * 2 + 16 * sizeof(NsrLevel);
        hipLaunchKernelGGL((k_field_fwd<TT, CD, false>), dim3(nblocks), dim3(256), lds, s, a);
    }
    return nsr_launch_status();
}

extern "C" {

int nsr_field_forward(const nsr_field_desc *desc, const void *tables, const float *mlp_params, const float *xyzs, uint32_t M,
                      const int32_t *m_dev, float *sigmas, float *rgbs, void *feats, const uint32_t *perm, nsr_stream_t stream) {
    if (M == 0) return NSR_OK;
    NSR_CHECK_PTR(desc); NSR_CHECK_PTR(tables); NSR_CHECK_PTR(mlp_params); NSR_CHECK_PTR(xyzs); NSR_CHECK_PTR(sigmas);
    FieldArgs a;
    uint32_t nblocks;
    const int st = field_fill_args(desc, a, M, nblocks);
    if (st != NSR_OK) return st;
    if ((uintptr_t)tables & 15u) return NSR_ERR_INVALID_ARG;
    if (rgbs && a.C_ch == 8 && ((uintptr_t)rgbs & 15u)) return NSR_ERR_INVALID_ARG;
    a.tables = tables; a.params = mlp_params; a.xyzs = xyzs; a.m_dev = m_dev; a.sigmas = sigmas; a.rgbs = rgbs;
    a.feats = feats;
    a.perm = perm;
    if (feats && ((uintptr_t)feats & 15u)) return NSR_ERR_INVALID_ARG;
    const bool so = rgbs == nullptr;
    hipStream_t s = (hipStream_t)stream;
    if (desc->table_dtype == NSR_F32 && desc->compute_dtype == NSR_F16) return field_launch_fwd<float, NSR_F16>(a, nblocks, so, s);
    if (desc->table_dtype == NSR_F32 && desc->compute_dtype == NSR_BF16) return field_launch_fwd<float, NSR_BF16>(a, nblocks, so, s);
    if (desc->table_dtype == NSR_F16 && desc->compute_dtype == NSR_F16) return field_launch_fwd<_Float16, NSR_F16>(a, nblocks, so, s);
    if (desc->table_dtype == NSR_F16 && desc->compute_dtype == NSR_BF16) return field_launch_fwd<_Float16, NSR_BF16>(a, nblocks, so, s);
    return NSR_ERR_UNSUPPORTED;
}

}   // extern "C"
