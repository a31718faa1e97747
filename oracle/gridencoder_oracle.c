/*
 * ORACLE (test infrastructure, NOT product code).
 *
 * CPU restatement, in plain C, of the reference's multiresolution hash-grid encoder
 * (hkust-vgd/nerfstyle, gridencoder/src/gridencoder.cu), D = 3 input dims.
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load it.
 *
 * Parity status: the reference kernel is CUDA-only (no nvcc/GPU in the authoring
 * container) and the reference ships no tests for it: "parity unpinned" against an
 * actual reference execution; pinned by the known-answer properties of SURVEY.md
 * section 4 (level locality, OOB -> zeros, backward = transpose of forward) and by an
 * independent pure-PyTorch restatement (oracle/torch_port.py).
 *
 * One deliberate re-statement: the reference evaluates
 *     resolution = (uint32_t)floor(exp2f(level * S) * H)        (gridencoder.cu:137,264)
 * on the device.  A 1-ulp difference in exp2f flips a resolution, so the per-level
 * table is computed once on the host with this function (ora_grid_resolution, fp32
 * arithmetic, libm exp2f) and the same table is used by the oracle and by the HIP
 * kernels (nsr_grid_resolutions in the product library applies the same formula).
 *
 * The embeddings may be fp32 or fp16-valued (passed as fp32 arrays holding
 * fp16-representable numbers).  half_accum != 0 reproduces the reference's
 * scalar_t = at::Half accumulation (results[ch] += w * grid[...] rounds to half after
 * every corner, gridencoder.cu:154,177).
 */
#include <math.h>
#include <stdint.h>
#include <string.h>

/* round-to-nearest-even fp32 -> fp16 -> fp32 (value of the nearest half) */
static float ora_round_f16(float f) {
    uint32_t x;
    memcpy(&x, &f, 4);
    const uint32_t sign = x & 0x80000000u;
    x &= 0x7FFFFFFFu;
    float out;
    if (x >= 0x7F800000u) {            /* inf / nan */
        out = f;
        return out;
    }
    if (x >= 0x477FF000u) {            /* >= 65520 rounds to inf */
        uint32_t inf = sign | 0x7F800000u;
        memcpy(&out, &inf, 4);
        return out;
    }
    if (x < 0x38800000u) {             /* below the smallest normal half (2^-14): subnormal grid 2^-24 */
        float a;
        memcpy(&a, &x, 4);
        /* adding 0.5f (ulp 2^-24 in [0.5,1)) rounds to a multiple of 2^-24, ties-to-even */
        volatile float t = a + 0.5f;
        a = t - 0.5f;
        uint32_t r;
        memcpy(&r, &a, 4);
        r |= sign;
        memcpy(&out, &r, 4);
        return out;
    }
    /* normal: keep 10 mantissa bits, round to nearest even on the 13 dropped bits */
    const uint32_t lsb = (x >> 13) & 1u;
    x += 0xFFFu + lsb;
    x &= ~0x1FFFu;
    x |= sign;
    memcpy(&out, &x, 4);
    return out;
}

void ora_round_f16_array(const float *in, float *out, uint64_t n) {
    for (uint64_t i = 0; i < n; i++) out[i] = ora_round_f16(in[i]);
}

/* gridencoder.cu:137 -- the level's grid resolution, fp32 arithmetic */
uint32_t ora_grid_resolution(uint32_t level, float S, uint32_t H) {
    return (uint32_t)floorf(exp2f((float)level * S) * (float)H);
}

/* gridencoder.cu:35-52 (D = 3; primes[3] multiplies the style id) */
static inline uint32_t ora_fast_hash3(const uint32_t p[3], uint32_t style) {
    uint32_t result = 0;
    result ^= p[0] * 1u;
    result ^= p[1] * 2654435761u;
    result ^= p[2] * 805459861u;
    result ^= style * 3674653429u;
    return result;
}

/* gridencoder.cu:55-80 */
static inline uint32_t ora_get_grid_index(uint32_t gridtype, uint32_t C, uint32_t ch, uint32_t hashmap_size,
                                          uint32_t resolution, const uint32_t pos_grid[3], uint32_t style) {
    uint32_t stride = 1;
    uint32_t index = 0;
    const uint32_t max_styles = 512;
    for (uint32_t d = 0; d < 3 && stride <= hashmap_size; d++) {
        index += pos_grid[d] * stride;
        stride *= (resolution + 1);    /* :65 -- resolution + 1 regardless of align_corners */
    }
    if (stride <= hashmap_size) {
        index += style * stride;
        stride *= max_styles;
    }
    if (gridtype == 0 && stride > hashmap_size) index = ora_fast_hash3(pos_grid, style);
    return (index % hashmap_size) * C + ch;
}

/* Shared front half of kernel_grid / kernel_grid_backward (gridencoder.cu:134-149,261-284).
 * Returns 0 when the input is out of [0,1] (outputs zero / no gradient). */
static inline int ora_locate(const float *in, uint32_t resolution, int align_corners, float pos[3],
                             uint32_t pos_grid[3]) {
    for (int d = 0; d < 3; d++)
        if (in[d] < 0 || in[d] > 1) return 0;
    const float scale = (float)(resolution - (align_corners ? 0 : 1));
    for (int d = 0; d < 3; d++) {
        pos[d] = in[d] * scale + (align_corners ? 0.0f : 0.5f);
        pos_grid[d] = (uint32_t)fminf(floorf(pos[d]), (float)(resolution - 1));
        pos[d] -= (float)pos_grid[d];
    }
    return 1;
}

/* gridencoder.cu:83-187 (forward, calc_grad_inputs = false).
 * outputs: [L, B, C] exactly as the reference kernel writes it (grid.py:58 permutes). */
void ora_grid_encode_forward(const float *inputs, const float *embeddings, const int32_t *offsets,
                             float *outputs, uint32_t B, uint32_t C, uint32_t L, float S, uint32_t H,
                             uint32_t gridtype, int align_corners, uint32_t style, int half_accum) {
    for (uint32_t level = 0; level < L; level++) {
        const float *grid = embeddings + (size_t)(uint32_t)offsets[level] * C;
        const uint32_t hashmap_size = (uint32_t)(offsets[level + 1] - offsets[level]);
        const uint32_t resolution = ora_grid_resolution(level, S, H);
        for (uint32_t b = 0; b < B; b++) {
            float *out = outputs + (size_t)level * B * C + (size_t)b * C;
            float pos[3];
            uint32_t pos_grid[3];
            if (!ora_locate(inputs + (size_t)b * 3, resolution, align_corners, pos, pos_grid)) {
                for (uint32_t ch = 0; ch < C; ch++) out[ch] = 0;
                continue;
            }
            float results[8] = {0};
            for (uint32_t idx = 0; idx < 8; idx++) {
                float w = 1;
                uint32_t pgl[3];
                for (uint32_t d = 0; d < 3; d++) {
                    if ((idx & (1u << d)) == 0) { w *= 1 - pos[d]; pgl[d] = pos_grid[d]; }
                    else { w *= pos[d]; pgl[d] = pos_grid[d] + 1; }
                }
                const uint32_t index = ora_get_grid_index(gridtype, C, 0, hashmap_size, resolution, pgl, style);
                for (uint32_t ch = 0; ch < C; ch++) {
                    results[ch] += w * grid[index + ch];
                    if (half_accum) results[ch] = ora_round_f16(results[ch]);
                }
            }
            for (uint32_t ch = 0; ch < C; ch++) out[ch] = results[ch];
        }
    }
}

/* gridencoder.cu:238-328 (backward; atomics become sequential adds in (level, b, corner)
 * order).  grad: [L, B, C]; grad_embeddings arrives zeroed (grid.py:82), fp32 accumulate. */
void ora_grid_encode_backward(const float *grad, const float *inputs, const int32_t *offsets,
                              float *grad_embeddings, uint32_t B, uint32_t C, uint32_t L, float S,
                              uint32_t H, uint32_t gridtype, int align_corners, uint32_t style) {
    for (uint32_t level = 0; level < L; level++) {
        float *gg = grad_embeddings + (size_t)(uint32_t)offsets[level] * C;
        const uint32_t hashmap_size = (uint32_t)(offsets[level + 1] - offsets[level]);
        const uint32_t resolution = ora_grid_resolution(level, S, H);
        for (uint32_t b = 0; b < B; b++) {
            const float *g = grad + (size_t)level * B * C + (size_t)b * C;
            float pos[3];
            uint32_t pos_grid[3];
            if (!ora_locate(inputs + (size_t)b * 3, resolution, align_corners, pos, pos_grid)) continue;
            for (uint32_t idx = 0; idx < 8; idx++) {
                float w = 1;
                uint32_t pgl[3];
                for (uint32_t d = 0; d < 3; d++) {
                    if ((idx & (1u << d)) == 0) { w *= 1 - pos[d]; pgl[d] = pos_grid[d]; }
                    else { w *= pos[d]; pgl[d] = pos_grid[d] + 1; }
                }
                const uint32_t index = ora_get_grid_index(gridtype, C, 0, hashmap_size, resolution, pgl, style);
                for (uint32_t ch = 0; ch < C; ch++) gg[index + ch] += w * g[ch];
            }
        }
    }
}

/* Emits the (level, b, corner) -> row index table, for index-parity (bit-exact) tests.
 * rows: [L, B, 8] uint32 (row within the level, i.e. before "* C + ch"); 0xFFFFFFFF for OOB. */
void ora_grid_corner_rows(const float *inputs, const int32_t *offsets, uint32_t *rows, uint32_t B,
                          uint32_t L, float S, uint32_t H, uint32_t gridtype, int align_corners,
                          uint32_t style) {
    for (uint32_t level = 0; level < L; level++) {
        const uint32_t hashmap_size = (uint32_t)(offsets[level + 1] - offsets[level]);
        const uint32_t resolution = ora_grid_resolution(level, S, H);
        for (uint32_t b = 0; b < B; b++) {
            uint32_t *r = rows + ((size_t)level * B + b) * 8;
            float pos[3];
            uint32_t pos_grid[3];
            if (!ora_locate(inputs + (size_t)b * 3, resolution, align_corners, pos, pos_grid)) {
                for (int i = 0; i < 8; i++) r[i] = 0xFFFFFFFFu;
                continue;
            }
            for (uint32_t idx = 0; idx < 8; idx++) {
                uint32_t pgl[3];
                for (uint32_t d = 0; d < 3; d++) pgl[d] = pos_grid[d] + (((idx >> d) & 1u) ? 1u : 0u);
                r[idx] = ora_get_grid_index(gridtype, 1, 0, hashmap_size, resolution, pgl, style);
            }
        }
    }
}
