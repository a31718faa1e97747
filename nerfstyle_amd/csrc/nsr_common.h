// Shared host/device helpers for libnsr_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/nsr.h"

#define NSR_CHECK_PTR(p) \
    do { if ((p) == nullptr) return NSR_ERR_INVALID_ARG; } while (0)

static inline int nsr_launch_status() {
    return hipGetLastError() == hipSuccess ? NSR_OK : NSR_ERR_LAUNCH;
}

static inline uint32_t nsr_div_up(uint64_t a, uint64_t b) { return (uint32_t)((a + b - 1) / b); }

// Memory-bound 1-D launches: cap the grid at 256 CUs x 8 blocks and grid-stride the rest.
static inline uint32_t nsr_grid_1d(uint64_t work, uint32_t block) {
    uint64_t g = (work + block - 1) / block;
    if (g > 2048) g = 2048;
    if (g == 0) g = 1;
    return (uint32_t)g;
}

// ---- hash-grid level table (host-computed once per call, passed by value) ----------------
#define NSR_MAX_LEVELS 32
struct NsrLevel {
    uint32_t offset;      // first row of the level
    uint32_t size;        // rows in the level (hashmap_size)
    uint32_t resolution;  // floor(exp2f(level*S)*H), gridencoder.cu:137
    uint32_t use_hash;    // gridencoder.cu:75
    uint32_t mul[3];      // dense-index strides (0 for dims the loop at :62-66 never reaches)
    uint32_t style_mul;   // :68-71
    // index % size without a division (Granlund & Montgomery, "Division by invariant integers using
    // multiplication", fig. 4.1): q = (t + ((n - t) >> sh1)) >> sh2 with t = mulhi(magic, n); exact for
    // every 32-bit n and every size >= 1, powers of two included -> one branchless path
    uint32_t magic, sh1, sh2, pad_;
};
struct NsrLevels {
    NsrLevel lv[NSR_MAX_LEVELS];
};

// Restates the control flow of get_grid_index (gridencoder.cu:55-80), which depends only on
// (resolution, hashmap_size, gridtype), never on the position.
static inline void nsr_fill_levels(NsrLevels *out, const int32_t *offsets, uint32_t L, float S, uint32_t H,
                                   uint32_t gridtype) {
    for (uint32_t l = 0; l < L; l++) {
        NsrLevel &v = out->lv[l];
        v.offset = (uint32_t)offsets[l];
        v.size = (uint32_t)(offsets[l + 1] - offsets[l]);
        v.resolution = (uint32_t)floorf(exp2f((float)l * S) * (float)H);
        uint32_t stride = 1;
        v.mul[0] = v.mul[1] = v.mul[2] = 0;
        for (uint32_t d = 0; d < 3 && stride <= v.size; d++) {
            v.mul[d] = stride;
            stride *= (v.resolution + 1);
        }
        v.style_mul = 0;
        if (stride <= v.size) {
            v.style_mul = stride;
            stride *= 512u;
        }
        v.use_hash = (gridtype == 0 && stride > v.size) ? 1u : 0u;
        uint32_t lg = 0;
        while ((1ull << lg) < (unsigned long long)v.size) lg++;        // lg = ceil(log2(size))
        v.magic = v.size ? (uint32_t)((((1ull << lg) - v.size) << 32) / v.size + 1ull) : 0u;
        v.sh1 = lg < 1 ? lg : 1;
        v.sh2 = lg > 0 ? lg - 1 : 0;
        v.pad_ = 0;
    }
}

#ifdef __HIPCC__
// gridencoder.cu:35-52 (D = 3) and :55-80
__device__ __forceinline__ uint32_t nsr_grid_row(const NsrLevel &lv, uint32_t x, uint32_t y, uint32_t z,
                                                 uint32_t style) {
    uint32_t index;
    if (lv.use_hash) {
        index = (x * 1u) ^ (y * 2654435761u) ^ (z * 805459861u) ^ (style * 3674653429u);
    } else {
        index = x * lv.mul[0] + y * lv.mul[1] + z * lv.mul[2] + style * lv.style_mul;
    }
    const uint32_t t = __umulhi(lv.magic, index);
    const uint32_t q = (t + ((index - t) >> lv.sh1)) >> lv.sh2;
    return index - q * lv.size;                                       // == index % lv.size
}

// gridencoder.cu:138-149.  Contraction is disabled so floor() sees the same value as the oracle.
__device__ __forceinline__ void nsr_grid_locate(float x, uint32_t resolution, int align_corners, float &frac,
                                                uint32_t &cell) {
#pragma clang fp contract(off)
    const float scale = (float)(resolution - (align_corners ? 0u : 1u));
    // (no "+ 0.0f" for align_corners: with contraction off the compiler must keep that add -- x + 0.0 is not x for -0.0 --
    // and the value is the same up to the sign of a zero, which no later product or floor can see)
    float p = x * scale;
    if (!align_corners) p += 0.5f;
    const float c = fminf(floorf(p), (float)(resolution - 1u));
    cell = (uint32_t)c;
    frac = p - c;
}
#endif
