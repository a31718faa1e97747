// Spatial processing order of the marched samples, gfx950.
//
// The march emits samples ray by ray.  A full-frame batch (762 048 rays, 48 M samples) visits every occupied piece of
// space with hundreds of rays, but in ray order two samples that share a hash-table row are millions of samples apart:
// the forward re-fetches the row from HBM, the backward sends one more atomic to the memory side (the 6.3 M-row
// gradient table receives ~1.4 G records per launch).  nsr_sample_order computes a PERMUTATION -- sample indices in
// Morton order of the encoder input, 10 bits per axis (one key = a 4^3 block of finest-level cells), stable, i.e. ray
// order inside a block -- that the fused field kernels walk instead (`perm` argument): consecutive 16-sample tiles
// then touch the same few cells on every level, the forward's gathers hit in L2 and the backward accumulates whole
// lattice tiles in LDS before one merged atomic per corner leaves (field_bwd.hip).  The sample buffers themselves
// stay in ray order (the composite kernels walk them per ray); only the field kernels go through the permutation
// (forward, MLP backward and table scatter all walk it: the forward's HBM traffic falls from 1230 to 538 B per sample).
//
// The sort is rocPRIM's radix_sort_pairs (a plain library sort of 30-bit keys: 1.9 ms for 47 M pairs on MI355X);
// the keys are made here.  The number of valid samples lives on the device: `sort_prefix` (host value, <= M) says
// how many leading slots take part in the sort -- any value is correct (slots past it keep identity order,
// invalid slots inside it sort to the end); the caller passes an estimate of the emitted count so that a
// capacity-sized buffer is not sorted whole.
#include <string.h>

#include <hip/hip_runtime.h>
#include <rocprim/device/device_radix_sort.hpp>

#include "nsr_common.h"
#include "rm_util.h"

struct OrderArgs {
    const float *xyzs;
    const int32_t *m_dev;
    uint32_t M, prefix;
    float bmin[3], bsize[3];
    uint32_t *keys, *vals, *perm;
};

__device__ __forceinline__ uint32_t order_spread10(uint32_t v) {
    v &= 0x3FFu;
    v = (v | (v << 16)) & 0x030000FFu;
    v = (v | (v << 8)) & 0x0300F00Fu;
    v = (v | (v << 4)) & 0x030C30C3u;
    v = (v | (v << 2)) & 0x09249249u;
    return v;
}

__global__ void __launch_bounds__(256)
k_order_keys(OrderArgs a) {
    const uint32_t Mc = a.m_dev ? min((uint32_t)max(a.m_dev[0], 0), a.M) : a.M;
    for (uint32_t i = blockIdx.x * 256 + threadIdx.x; i < a.M; i += gridDim.x * 256) {
        if (i >= a.prefix) {
            a.perm[i] = i;            // not sorted: identity
            continue;
        }
        uint32_t key = 0xFFFFFFFFu;   // slots past the emitted count sort to the end
        if (i < Mc) {
            uint32_t q[3];
#pragma unroll
            for (int d = 0; d < 3; d++) {
                // encoder input of the position (BBox.normalize, then GridEncoder's (x + 1) / 2), quantised to 10 bits
                const float u = ((a.xyzs[(size_t)i * 3 + d] - a.bmin[d]) / a.bsize[d] + 1.0f) * 0.5f;
                const float s = fminf(fmaxf(u * 1024.0f, 0.0f), 1023.0f);     // NaN -> 0
                q[d] = (uint32_t)s;
            }
            key = order_spread10(q[0]) | (order_spread10(q[1]) << 1) | (order_spread10(q[2]) << 2);
        }
        a.keys[i] = key;
        a.vals[i] = i;
    }
}

struct OrderLayout {
    uint64_t keys_in, keys_out, vals_in, temp, total;
    size_t temp_bytes;
};
static OrderLayout order_layout(uint32_t M) {
    OrderLayout l;
    const uint64_t n = ((uint64_t)M * 4 + 255) & ~255ull;
    l.keys_in = 0; l.keys_out = n; l.vals_in = 2 * n; l.temp = 3 * n;
    size_t tb = 0;
    (void)rocprim::radix_sort_pairs(nullptr, tb, (uint32_t *)nullptr, (uint32_t *)nullptr, (uint32_t *)nullptr,
                                    (uint32_t *)nullptr, (size_t)M, 0u, 32u, (hipStream_t)0);
    l.temp_bytes = tb;
    l.total = l.temp + ((tb + 255) & ~255ull);
    return l;
}

extern "C" {

uint64_t nsr_sample_order_workspace_bytes(uint32_t M) {
    if (M == 0) return 0;
    return order_layout(M).total;
}

int nsr_sample_order(const float *xyzs, uint32_t M, const int32_t *m_dev, uint32_t sort_prefix, const float *bbox_min,
                     const float *bbox_size, uint32_t *perm, void *workspace, nsr_stream_t stream) {
    if (M == 0) return NSR_OK;
    NSR_CHECK_PTR(xyzs); NSR_CHECK_PTR(bbox_min); NSR_CHECK_PTR(bbox_size); NSR_CHECK_PTR(perm); NSR_CHECK_PTR(workspace);
    if ((uintptr_t)workspace & 255u) return NSR_ERR_INVALID_ARG;
    if (sort_prefix > M) sort_prefix = M;
    hipStream_t s = (hipStream_t)stream;
    const OrderLayout l = order_layout(M);
    char *ws = (char *)workspace;
    OrderArgs a;
    a.xyzs = xyzs; a.m_dev = m_dev; a.M = M; a.prefix = sort_prefix;
    for (int d = 0; d < 3; d++) { a.bmin[d] = bbox_min[d]; a.bsize[d] = bbox_size[d]; }
    a.keys = (uint32_t *)(ws + l.keys_in); a.vals = (uint32_t *)(ws + l.vals_in); a.perm = perm;
    hipLaunchKernelGGL(k_order_keys, dim3(nsr_grid_1d(M, 256)), dim3(256), 0, s, a);
    if (hipGetLastError() != hipSuccess) return NSR_ERR_LAUNCH;
    if (sort_prefix > 0) {
        size_t tb = l.temp_bytes;
        // 30 key bits + the all-ones "invalid" key: sort on all 32 bits only when invalid slots can be present
        const hipError_t e = rocprim::radix_sort_pairs((void *)(ws + l.temp), tb, a.keys, (uint32_t *)(ws + l.keys_out), a.vals, perm,
                                                       (size_t)sort_prefix, 0u, 32u, s);
        if (e != hipSuccess) return NSR_ERR_LAUNCH;
    }
    return nsr_launch_status();
}

}   // extern "C"
