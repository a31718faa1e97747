// Morton code and block-scan helpers shared by the ray-march (raymarch.hip) and occupancy-update
// (occupancy.hip) kernels.
#pragma once
#include "nsr_common.h"

// raymarching.cu:56-81
__device__ __forceinline__ uint32_t rm_expand_bits(uint32_t v) {
    // the reference's four multiply-and-mask steps (v * 0x00010001u & 0xFF0000FFu, ...): each product is
    // v + (v << s) with no overlapping bits after the previous mask, i.e. v | (v << s) -- shift-or runs at full
    // rate, 32-bit integer multiplies at a quarter of it
    v = (v | (v << 16)) & 0xFF0000FFu;
    v = (v | (v << 8)) & 0x0F00F00Fu;
    v = (v | (v << 4)) & 0xC30C30C3u;
    v = (v | (v << 2)) & 0x49249249u;
    return v;
}
__device__ __forceinline__ uint32_t rm_morton3d(uint32_t x, uint32_t y, uint32_t z) {
    return rm_expand_bits(x) | (rm_expand_bits(y) << 1) | (rm_expand_bits(z) << 2);
}
__device__ __forceinline__ uint32_t rm_morton3d_invert(uint32_t x) {
    x = x & 0x49249249;
    x = (x | (x >> 2)) & 0xc30c30c3;
    x = (x | (x >> 4)) & 0x0f00f00f;
    x = (x | (x >> 8)) & 0xff0000ff;
    x = (x | (x >> 16)) & 0x0000ffff;
    return x;
}

// wave64 inclusive scan by shuffles, then a block scan over the (<= 16) wave totals in LDS.
// Returns the exclusive prefix of v inside the block and the block total in `total`.
__device__ __forceinline__ uint32_t rm_block_exclusive_scan(uint32_t v, uint32_t *lds_wave_sums, uint32_t &total) {
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    uint32_t incl = v;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const uint32_t up = __shfl_up(incl, off, 64);
        if (lane >= (uint32_t)off) incl += up;
    }
    if (lane == 63) lds_wave_sums[wave] = incl;
    __syncthreads();
    const uint32_t nw = blockDim.x >> 6;
    uint32_t wave_prefix = 0, tot = 0;
    for (uint32_t w = 0; w < nw; w++) {
        const uint32_t s = lds_wave_sums[w];
        if (w < wave) wave_prefix += s;
        tot += s;
    }
    __syncthreads();
    total = tot;
    return wave_prefix + incl - v;
}

