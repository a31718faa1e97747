/*
 * ORACLE (test infrastructure, NOT product code).
 *
 * CPU restatement, in plain C, of the arithmetic of the reference's ray-marching
 * extension (hkust-vgd/nerfstyle, raymarching/src/raymarching.cu).  Every function
 * cites the reference file:line it follows.  Only tests/, __graft_entry__.smoke() and
 * bench.py's cpu_baseline leg may load this library; the product path (nerfstyle_amd/)
 * never does.
 *
 * Parity status: the reference kernels are CUDA-only and cannot be built or run in the
 * authoring container (no nvcc, no GPU) and the reference ships no tests or golden
 * vectors for them, so this restatement is pinned by the known-answer properties the
 * reference code implies -- "parity unpinned" against an actual
 * reference execution.
 *
 * Conventions: fp32 arithmetic exactly as written in the reference (build with
 * -ffp-contract=off so no FMA contraction changes roundings).  The reference's
 * per-ray CUDA threads are replaced by a sequential loop over rays in index order;
 * the reference's atomicAdd arrival order (raymarching.cu:506-507) is therefore the
 * one valid order "ray 0 first, ray 1 second, ...".
 */
#include <float.h>
#include <math.h>
#include <stdint.h>
#include <string.h>

/* raymarching.cu:19 */
#define ORA_SQRT3 1.7320508075688772f

/* raymarching.cu:30-32 */
static inline float ora_signf(const float x) { return copysignf(1.0f, x); }

/* raymarching.cu:34-36 */
static inline float ora_clamp(const float x, const float lo, const float hi) {
    return fminf(hi, fmaxf(lo, x));
}

/* raymarching.cu:42-47 */
static inline int ora_mip_from_pos(const float x, const float y, const float z, const float max_cascade) {
    const float mx = fmaxf(fabsf(x), fmaxf(fabsf(y), fabsf(z)));
    int exponent;
    frexpf(mx, &exponent);
    return (int)fminf(max_cascade - 1, fmaxf(0, (float)exponent));
}

/* raymarching.cu:49-54 (the 0.5 literal is a double in the reference) */
static inline int ora_mip_from_dt(const float dt, const float H, const float max_cascade) {
    const float mx = (float)(dt * H * 0.5);
    int exponent;
    frexpf(mx, &exponent);
    return (int)fminf(max_cascade - 1, fmaxf(0, (float)exponent));
}

/* raymarching.cu:56-63 */
static inline uint32_t ora_expand_bits(uint32_t v) {
    v = (v * 0x00010001u) & 0xFF0000FFu;
    v = (v * 0x00000101u) & 0x0F00F00Fu;
    v = (v * 0x00000011u) & 0xC30C30C3u;
    v = (v * 0x00000005u) & 0x49249249u;
    return v;
}

/* raymarching.cu:65-71 */
static inline uint32_t ora_morton3D_1(uint32_t x, uint32_t y, uint32_t z) {
    return ora_expand_bits(x) | (ora_expand_bits(y) << 1) | (ora_expand_bits(z) << 2);
}

/* raymarching.cu:73-81 */
static inline uint32_t ora_morton3D_invert_1(uint32_t x) {
    x = x & 0x49249249;
    x = (x | (x >> 2)) & 0xc30c30c3;
    x = (x | (x >> 4)) & 0x0f00f00f;
    x = (x | (x >> 8)) & 0xff0000ff;
    x = (x | (x >> 16)) & 0x0000ffff;
    return x;
}

/* raymarching.cu:190-244 */
void ora_near_far_from_aabb(const float *rays_o, const float *rays_d, const float *aabb,
                            uint32_t N, float min_near, float *nears, float *fars) {
    for (uint32_t n = 0; n < N; n++) {
        const float *o = rays_o + (size_t)n * 3, *d = rays_d + (size_t)n * 3;
        const float ox = o[0], oy = o[1], oz = o[2];
        const float dx = d[0], dy = d[1], dz = d[2];
        const float rdx = 1 / dx, rdy = 1 / dy, rdz = 1 / dz;

        float near = (aabb[0] - ox) * rdx;
        float far = (aabb[3] - ox) * rdx;
        if (near > far) { float c = near; near = far; far = c; }

        float near_y = (aabb[1] - oy) * rdy;
        float far_y = (aabb[4] - oy) * rdy;
        if (near_y > far_y) { float c = near_y; near_y = far_y; far_y = c; }

        if (near > far_y || near_y > far) { nears[n] = fars[n] = FLT_MAX; continue; }
        if (near_y > near) near = near_y;
        if (far_y < far) far = far_y;

        float near_z = (aabb[2] - oz) * rdz;
        float far_z = (aabb[5] - oz) * rdz;
        if (near_z > far_z) { float c = near_z; near_z = far_z; far_z = c; }

        if (near > far_z || near_z > far) { nears[n] = fars[n] = FLT_MAX; continue; }
        if (near_z > near) near = near_z;
        if (far_z < far) far = far_z;

        if (near < min_near) near = min_near;
        nears[n] = near;
        fars[n] = far;
    }
}

/* raymarching.cu:313-325 */
void ora_morton3D(const int32_t *coords, uint32_t N, int32_t *indices) {
    for (uint32_t n = 0; n < N; n++)
        indices[n] = (int32_t)ora_morton3D_1((uint32_t)coords[n * 3], (uint32_t)coords[n * 3 + 1],
                                             (uint32_t)coords[n * 3 + 2]);
}

/* raymarching.cu:336-353 */
void ora_morton3D_invert(const int32_t *indices, uint32_t N, int32_t *coords) {
    for (uint32_t n = 0; n < N; n++) {
        const int32_t ind = indices[n];
        coords[n * 3 + 0] = (int32_t)ora_morton3D_invert_1((uint32_t)(ind >> 0));
        coords[n * 3 + 1] = (int32_t)ora_morton3D_invert_1((uint32_t)(ind >> 1));
        coords[n * 3 + 2] = (int32_t)ora_morton3D_invert_1((uint32_t)(ind >> 2));
    }
}

/* raymarching.cu:366-388: bit i of byte n = grid[8n+i] > thresh (strict) */
void ora_packbits(const float *grid, uint32_t N, float density_thresh, uint8_t *bitfield) {
    for (uint32_t n = 0; n < N; n++) {
        const float *g = grid + (size_t)n * 8;
        uint8_t bits = 0;
        for (uint8_t i = 0; i < 8; i++) bits |= (g[i] > density_thresh) ? ((uint8_t)1 << i) : 0;
        bitfield[n] = bits;
    }
}

/* One marching state step shared by the train and inference kernels.
 * Restates raymarching.cu:460-500 (first pass), :530-588 (second pass), :1059-1119. */
typedef struct {
    float ox, oy, oz, dx, dy, dz, rdx, rdy, rdz, rH, H3, bound, dt_gamma, dt_min, dt_max;
    uint32_t C, H;
    const uint8_t *grid;
} ora_march_ctx;

static inline void ora_ctx_init(ora_march_ctx *c, const float *o, const float *d, float bound,
                                float dt_gamma, uint32_t max_steps, uint32_t C, uint32_t H,
                                const uint8_t *grid) {
    c->ox = o[0]; c->oy = o[1]; c->oz = o[2];
    c->dx = d[0]; c->dy = d[1]; c->dz = d[2];
    c->rdx = 1 / c->dx; c->rdy = 1 / c->dy; c->rdz = 1 / c->dz;
    c->rH = 1 / (float)H;
    c->H3 = (float)(H * H * H);
    c->bound = bound; c->dt_gamma = dt_gamma;
    c->dt_min = 2 * ORA_SQRT3 / max_steps;            /* :446 */
    c->dt_max = 2 * ORA_SQRT3 * (1 << (C - 1)) / H;   /* :447 */
    c->C = C; c->H = H; c->grid = grid;
}

/* Evaluates the sample at ray parameter t.  Returns occupancy; outputs the clamped
 * position, dt and (for the empty branch) the parameter tt of the next voxel face. */
static inline int ora_probe(const ora_march_ctx *c, float t, float *px, float *py, float *pz,
                            float *pdt, float *ptt) {
    const float x = ora_clamp(c->ox + t * c->dx, -c->bound, c->bound);
    const float y = ora_clamp(c->oy + t * c->dy, -c->bound, c->bound);
    const float z = ora_clamp(c->oz + t * c->dz, -c->bound, c->bound);
    const float dt = ora_clamp(t * c->dt_gamma, c->dt_min, c->dt_max);
    const int m1 = ora_mip_from_pos(x, y, z, (float)c->C);
    const int m2 = ora_mip_from_dt(dt, (float)c->H, (float)c->C);
    const int level = m1 > m2 ? m1 : m2;
    const float mip_bound = fminf(scalbnf(1.0f, level), c->bound);
    const float mip_rbound = 1 / mip_bound;
    /* :475-477 -- double-precision product (0.5 literal), narrowed to float by clamp(),
     * truncated to int */
    const int nx = (int)ora_clamp((float)(0.5 * (x * mip_rbound + 1) * c->H), 0.0f, (float)(c->H - 1));
    const int ny = (int)ora_clamp((float)(0.5 * (y * mip_rbound + 1) * c->H), 0.0f, (float)(c->H - 1));
    const int nz = (int)ora_clamp((float)(0.5 * (z * mip_rbound + 1) * c->H), 0.0f, (float)(c->H - 1));
    /* :479 -- level * H3 is evaluated in float, the sum converted to uint32 */
    const uint32_t index = (uint32_t)(level * c->H3 + ora_morton3D_1((uint32_t)nx, (uint32_t)ny, (uint32_t)nz));
    const int occ = c->grid[index / 8] & (1 << (index % 8));
    *px = x; *py = y; *pz = z; *pdt = dt;
    if (!occ) {
        /* :491-495 */
        const float tx = (((nx + 0.5f + 0.5f * ora_signf(c->dx)) * c->rH * 2 - 1) * mip_bound - x) * c->rdx;
        const float ty = (((ny + 0.5f + 0.5f * ora_signf(c->dy)) * c->rH * 2 - 1) * mip_bound - y) * c->rdy;
        const float tz = (((nz + 0.5f + 0.5f * ora_signf(c->dz)) * c->rH * 2 - 1) * mip_bound - z) * c->rdz;
        *ptt = t + fmaxf(0.0f, fminf(tx, fminf(ty, tz)));
    }
    return occ != 0;
}

/* raymarching.cu:410-589.  deltas has 4 floats per sample; slots 2,3 are only written
 * when is_ndc (never on the LLFF path: cfgs/renderer/llff.yaml use_ndc false). */
void ora_march_rays_train(const float *rays_o, const float *rays_d, const float *z_hats,
                          const uint8_t *grid, float bound, float dt_gamma, uint32_t max_steps,
                          int is_ndc, uint32_t N, uint32_t C, uint32_t H, uint32_t M,
                          const float *nears, const float *fars, float *xyzs, float *dirs,
                          float *deltas, int32_t *rays, int32_t *counter, const float *noises) {
    for (uint32_t n = 0; n < N; n++) {
        ora_march_ctx c;
        ora_ctx_init(&c, rays_o + (size_t)n * 3, rays_d + (size_t)n * 3, bound, dt_gamma, max_steps, C, H, grid);
        const float near = nears[n], far = fars[n], noise = noises[n];
        (void)near;
        float t0 = nears[n];
        t0 += ora_clamp(t0 * dt_gamma, c.dt_min, c.dt_max) * noise;   /* :452 */

        /* first pass :455-501 */
        float t = t0;
        uint32_t num_steps = 0;
        float x, y, z, dt, tt;
        while (t < far && num_steps < max_steps) {
            if (ora_probe(&c, t, &x, &y, &z, &dt, &tt)) {
                num_steps++;
                t += dt;
            } else {
                do { t += ora_clamp(t * dt_gamma, c.dt_min, c.dt_max); } while (t < tt);
            }
        }

        /* :506-514 -- atomics become sequential adds */
        const uint32_t point_index = (uint32_t)counter[0]; counter[0] += (int32_t)num_steps;
        const uint32_t ray_index = (uint32_t)counter[1]; counter[1] += 1;
        rays[ray_index * 3] = (int32_t)n;
        rays[ray_index * 3 + 1] = (int32_t)point_index;
        rays[ray_index * 3 + 2] = (int32_t)num_steps;

        if (num_steps == 0) continue;
        if (point_index + num_steps >= M) continue;   /* :517 -- '>=' drops an exact fit */

        float *pxyz = xyzs + (size_t)point_index * 3;
        float *pdir = dirs ? dirs + (size_t)point_index * 3 : 0;
        float *pdel = deltas + (size_t)point_index * 4;

        /* second pass :523-588 */
        t = t0;
        uint32_t step = 0;
        float last_t = t;
        float last_z = ora_clamp(c.oz + t * c.dz, -bound, bound);
        while (t < far && step < num_steps) {
            if (ora_probe(&c, t, &x, &y, &z, &dt, &tt)) {
                pxyz[0] = x; pxyz[1] = y; pxyz[2] = z;
                if (pdir) { pdir[0] = c.dx; pdir[1] = c.dy; pdir[2] = c.dz; pdir += 3; }
                t += dt;
                pdel[0] = dt;
                pdel[1] = t - last_t;
                last_t = t;
                if (is_ndc) {
                    const float new_z = ora_clamp(c.oz + t * c.dz, -bound, bound);
                    pdel[2] = (2 / (new_z - 1) - 2 / (z - 1)) / z_hats[n];
                    pdel[3] = (2 / (new_z - 1) - 2 / (last_z - 1)) / z_hats[n];
                    last_z = z;   /* :570 (sic: the train kernel stores z, the inference one new_z) */
                }
                pxyz += 3; pdel += 4;
                step++;
            } else {
                do { t += ora_clamp(t * dt_gamma, c.dt_min, c.dt_max); } while (t < tt);
            }
        }
    }
}

/* raymarching.cu:806-879.  The reference uses __expf; expf is used here and the GPU
 * parity tolerance covers the difference. */
void ora_composite_rays_train_forward(const float *sigmas, const float *rgbs, const float *deltas,
                                      const int32_t *rays, uint32_t M, uint32_t N, uint32_t C,
                                      float T_thresh, int is_ndc, float *weights_sum, float *depth,
                                      float *image) {
    for (uint32_t n = 0; n < N; n++) {
        const uint32_t index = (uint32_t)rays[n * 3];
        const uint32_t offset = (uint32_t)rays[n * 3 + 1];
        const uint32_t num_steps = (uint32_t)rays[n * 3 + 2];
        for (uint32_t i = 0; i < C; ++i) image[(size_t)index * C + i] = 0;
        if (num_steps == 0 || offset + num_steps >= M) {
            weights_sum[index] = 0;
            depth[index] = 0;
            continue;
        }
        const float *s = sigmas + offset, *rgb = rgbs + (size_t)offset * C, *dl = deltas + (size_t)offset * 4;
        uint32_t step = 0;
        float T = 1.0f, ws = 0, t = 0, d = 0;
        while (step < num_steps) {
            const float alpha = 1.0f - expf(-s[0] * (is_ndc ? dl[2] : dl[0]));
            const float weight = alpha * T;
            for (uint32_t i = 0; i < C; ++i) image[(size_t)index * C + i] += weight * rgb[i];
            t += (is_ndc ? dl[3] : dl[1]);
            d += weight * t;
            ws += weight;
            T *= 1.0f - alpha;
            if (T < T_thresh) break;   /* :862 -- after accumulating */
            s++; rgb += C; dl += 4; step++;
        }
        weights_sum[index] = ws;
        depth[index] = d;
    }
}

/* raymarching.cu:904-986.  grad_sigmas / grad_rgbs / rgbs_buf must arrive zeroed
 * (raymarching.py:339-341). */
void ora_composite_rays_train_backward(const float *grad_weights_sum, const float *grad_image,
                                       const float *sigmas, const float *rgbs, const float *deltas,
                                       const int32_t *rays, int is_ndc, const float *weights_sum,
                                       const float *image, uint32_t M, uint32_t N, uint32_t C,
                                       float T_thresh, float *grad_sigmas, float *grad_rgbs,
                                       float *rgbs_buf) {
    for (uint32_t n = 0; n < N; n++) {
        const uint32_t index = (uint32_t)rays[n * 3];
        const uint32_t offset = (uint32_t)rays[n * 3 + 1];
        const uint32_t num_steps = (uint32_t)rays[n * 3 + 2];
        if (num_steps == 0 || offset + num_steps >= M) continue;
        const float gws = grad_weights_sum[index];
        const float *gim = grad_image + (size_t)index * C;
        const float ws_final = weights_sum[index];
        const float *im = image + (size_t)index * C;
        const float *s = sigmas + offset, *rgb = rgbs + (size_t)offset * C, *dl = deltas + (size_t)offset * 4;
        float *gs = grad_sigmas + offset, *grgb = grad_rgbs + (size_t)offset * C;
        float *buf = rgbs_buf + (size_t)index * C;
        uint32_t step = 0;
        float T = 1.0f, ws = 0;
        while (step < num_steps) {
            const float alpha = 1.0f - expf(-s[0] * (is_ndc ? dl[2] : dl[0]));
            const float weight = alpha * T;
            for (uint32_t i = 0; i < C; ++i) buf[i] += weight * rgb[i];
            ws += weight;
            T *= 1.0f - alpha;
            if (T < T_thresh) break;   /* :961 -- before writing this sample's grads */
            for (uint32_t i = 0; i < C; ++i) grgb[i] = gim[i] * weight;
            float grad_image_sum = 0;
            for (uint32_t i = 0; i < C; ++i) grad_image_sum += (gim[i] * (T * rgb[i] - (im[i] - buf[i])));
            gs[0] = (is_ndc ? dl[2] : dl[0]) * (grad_image_sum + gws * (1 - ws_final));
            s++; rgb += C; dl += 4; gs++; grgb += C; step++;
        }
        (void)ws;
    }
}

/* raymarching.cu:1004-1120.  rays_t has 1 float per ray (2 when is_ndc). */
void ora_march_rays(uint32_t n_alive, uint32_t n_step, const int32_t *rays_alive, const float *rays_t,
                    const float *rays_o, const float *rays_d, const float *z_hats, float bound,
                    float dt_gamma, uint32_t max_steps, int is_ndc, uint32_t C, uint32_t H,
                    const uint8_t *grid, const float *nears, const float *fars, float *xyzs,
                    float *dirs, float *deltas, const float *noises) {
    for (uint32_t n = 0; n < n_alive; n++) {
        const int index = rays_alive[n];
        const float noise = noises[n];
        ora_march_ctx c;
        ora_ctx_init(&c, rays_o + (size_t)index * 3, rays_d + (size_t)index * 3, bound, dt_gamma, max_steps, C, H, grid);
        float *pxyz = xyzs + (size_t)n * n_step * 3;
        float *pdir = dirs ? dirs + (size_t)n * n_step * 3 : 0;
        float *pdel = deltas + (size_t)n * n_step * 4;
        float t = rays_t[(size_t)index * (is_ndc ? 2 : 1)];
        const float far = fars[index];
        (void)nears;
        uint32_t step = 0;
        t += ora_clamp(t * dt_gamma, c.dt_min, c.dt_max) * noise;   /* :1053 */
        float last_t = t;
        float last_z = ora_clamp(c.oz + t * c.dz, -bound, bound);
        float x, y, z, dt, tt;
        while (t < far && step < n_step) {
            if (ora_probe(&c, t, &x, &y, &z, &dt, &tt)) {
                pxyz[0] = x; pxyz[1] = y; pxyz[2] = z;
                if (pdir) { pdir[0] = c.dx; pdir[1] = c.dy; pdir[2] = c.dz; pdir += 3; }
                t += dt;
                pdel[0] = dt;
                pdel[1] = t - last_t;
                if (is_ndc) {
                    const float new_z = ora_clamp(c.oz + t * c.dz, -bound, bound);
                    pdel[2] = (2 / (new_z - 1) - 2 / (z - 1)) / z_hats[index];
                    pdel[3] = (2 / (new_z - 1) - 2 / (last_z - 1)) / z_hats[index];
                    last_z = new_z;
                }
                last_t = t;
                pxyz += 3; pdel += 4;
                step++;
            } else {
                do { t += ora_clamp(t * dt_gamma, c.dt_min, c.dt_max); } while (t < tt);
            }
        }
    }
}

/* raymarching.cu:1133-1231 (in-place accumulation; terminated rays get rays_alive = -1). */
void ora_composite_rays(uint32_t n_alive, uint32_t n_step, float T_thresh, int32_t *rays_alive,
                        float *rays_t, const float *sigmas, const float *rgbs, const float *deltas,
                        uint32_t C, int is_ndc, float *weights_sum, float *depth, float *image) {
    for (uint32_t n = 0; n < n_alive; n++) {
        const int index = rays_alive[n];
        const float *s = sigmas + (size_t)n * n_step;
        const float *rgb = rgbs + (size_t)n * n_step * C;
        const float *dl = deltas + (size_t)n * n_step * 4;
        float *rt = rays_t + (size_t)index * (is_ndc ? 2 : 1);
        float *img = image + (size_t)index * C;
        float t_rm = 0, t_phy;
        if (is_ndc) { t_rm = rt[0]; t_phy = rt[1]; } else { t_phy = rt[0]; }
        float weight_sum = weights_sum[index];
        float d = depth[index];
        uint32_t step = 0;
        while (step < n_step) {
            if (dl[0] == 0) break;   /* :1178 */
            const float alpha = 1.0f - expf(-s[0] * (is_ndc ? dl[2] : dl[0]));
            const float T = 1 - weight_sum;
            const float weight = alpha * T;
            weight_sum += weight;
            if (is_ndc) { t_rm += dl[1]; t_phy += dl[3]; } else { t_phy += dl[1]; }
            d += weight * t_phy;
            for (uint32_t i = 0; i < C; ++i) img[i] += weight * rgb[i];
            if (T < T_thresh) break;   /* :1206 -- T *before* this sample */
            s++; rgb += C; dl += 4; step++;
        }
        if (step < n_step) {
            rays_alive[n] = -1;
        } else {
            if (is_ndc) { rt[0] = t_rm; rt[1] = t_phy; } else { rt[0] = t_phy; }
        }
        weights_sum[index] = weight_sum;
        depth[index] = d;
    }
}
