// Occupancy-grid ray march, sample compaction and alpha compositing for gfx950.
// Behaviour follows hkust-vgd/nerfstyle raymarching/src/raymarching.cu (cited per function);
// the structure does not: offsets come from a wave64 shuffle scan + one look-up of per-block
// totals (deterministic, no atomics), the composite backward keeps its running sums in
// registers (no rgbs_buf round trip), and every entry point takes an explicit stream.
#include <stdlib.h>

#include "nsr_common.h"
#include "rm_util.h"

#define RM_BLOCK 256
#define RM_SQRT3 1.7320508075688772f
#define NSR_MARCH_WPR_MAX_RAYS 20480u   // batches up to this size march one wave per ray (k_march_wpr)

// ---------------------------------------------------------------------------------------------
// small device helpers
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ float rm_clamp(float x, float lo, float hi) { return fminf(hi, fmaxf(lo, x)); }
__device__ __forceinline__ float rm_sign(float x) { return copysignf(1.0f, x); }

// ---------------------------------------------------------------------------------------------
// marching core (raymarching.cu:460-500, 530-588, 1059-1119)
// ---------------------------------------------------------------------------------------------
struct RmRay {
    float ox, oy, oz, dx, dy, dz, rdx, rdy, rdz;
};
struct RmCfg {
    float bound, rbound, dt_gamma, dt_min, dt_max, rH, H3, Hf, halfH, Cf;
    uint32_t H;
    const uint8_t *grid;
};

__device__ __forceinline__ RmCfg rm_cfg(float bound, float dt_gamma, uint32_t max_steps, uint32_t C, uint32_t H,
                                        const uint8_t *grid) {
    RmCfg c;
    c.bound = bound;
    c.rbound = 1 / bound;
    c.halfH = 0.5f * (float)H;
    c.dt_gamma = dt_gamma;
    c.dt_min = 2 * RM_SQRT3 / (float)max_steps;               // :446
    c.dt_max = 2 * RM_SQRT3 * (float)(1 << (C - 1)) / (float)H;  // :447
    c.rH = 1 / (float)H;
    c.H3 = (float)(H * H * H);
    c.Hf = (float)H;
    c.Cf = (float)C;
    c.H = H;
    c.grid = grid;
    return c;
}

__device__ __forceinline__ int rm_mip(float v, float max_cascade) {
    int e;
    frexpf(v, &e);
    return (int)fminf(max_cascade - 1, fmaxf(0.0f, (float)e));
}

// Evaluates the sample at parameter t.  Same operation order as the reference (and the oracle);
// contraction is off so that every rounding matches the restatement bit for bit.
__device__ __forceinline__ bool rm_probe(const RmRay &r, const RmCfg &c, float t, float &x, float &y, float &z,
                                         float &dt, float &tt) {
#pragma clang fp contract(off)
    x = rm_clamp(r.ox + t * r.dx, -c.bound, c.bound);
    y = rm_clamp(r.oy + t * r.dy, -c.bound, c.bound);
    z = rm_clamp(r.oz + t * r.dz, -c.bound, c.bound);
    dt = rm_clamp(t * c.dt_gamma, c.dt_min, c.dt_max);
    const int m1 = rm_mip(fmaxf(fabsf(x), fmaxf(fabsf(y), fabsf(z))), c.Cf);   // :42-47
    const int m2 = rm_mip((dt * c.Hf) * 0.5f, c.Cf);                             // :49-54 (x 0.5 is exact in either width)
    const int level = max(m1, m2);
    const float mip_pow = scalbnf(1.0f, level);
    const float mip_bound = fminf(mip_pow, c.bound);
    // 1 / mip_bound (:474): the reciprocal of a power of two is exact, the other case is the loop-invariant
    // 1 / bound -- the same correctly rounded quotients as the division, without a division per probe
    const float mip_rbound = mip_pow < c.bound ? scalbnf(1.0f, -level) : c.rbound;
    // :475-477 computes 0.5 * (double)v * (double)H and narrows once.  v has 24 significant bits, H at most 11:
    // the double product is exact, so the single rounding of the float product v * (0.5f * H) gives the same
    // float (0.5f * H is exact too) -- no fp64 in the probe
    const int nx = (int)rm_clamp((x * mip_rbound + 1) * c.halfH, 0.0f, (float)(c.H - 1));
    const int ny = (int)rm_clamp((y * mip_rbound + 1) * c.halfH, 0.0f, (float)(c.H - 1));
    const int nz = (int)rm_clamp((z * mip_rbound + 1) * c.halfH, 0.0f, (float)(c.H - 1));
    const uint32_t index = (uint32_t)((float)level * c.H3 + (float)rm_morton3d(nx, ny, nz));   // :479
    const bool occ = c.grid[index / 8] & (1 << (index % 8));
    if (!occ) {
        // :491-495
        const float tx = ((((float)nx + 0.5f + 0.5f * rm_sign(r.dx)) * c.rH * 2 - 1) * mip_bound - x) * r.rdx;
        const float ty = ((((float)ny + 0.5f + 0.5f * rm_sign(r.dy)) * c.rH * 2 - 1) * mip_bound - y) * r.rdy;
        const float tz = ((((float)nz + 0.5f + 0.5f * rm_sign(r.dz)) * c.rH * 2 - 1) * mip_bound - z) * r.rdz;
        tt = t + fmaxf(0.0f, fminf(tx, fminf(ty, tz)));
    }
    return occ;
}

// Returns the number of additions made (the march's step index k advances by it: k_march_count's sample mask).
__device__ __forceinline__ uint32_t rm_skip(const RmCfg &c, float &t, float tt) {
#pragma clang fp contract(off)
    // do { t += clamp(t * dt_gamma, dt_min, dt_max); } while (t < tt);  (:497) -- the same additions in the same
    // order, four per trip: the first partial sum that is not below tt is the loop's result
    uint32_t adds = 0;
    for (;;) {
        const float t1 = t + rm_clamp(t * c.dt_gamma, c.dt_min, c.dt_max);
        const float t2 = t1 + rm_clamp(t1 * c.dt_gamma, c.dt_min, c.dt_max);
        const float t3 = t2 + rm_clamp(t2 * c.dt_gamma, c.dt_min, c.dt_max);
        const float t4 = t3 + rm_clamp(t3 * c.dt_gamma, c.dt_min, c.dt_max);
        const bool b1 = t1 < tt, b2 = t2 < tt, b3 = t3 < tt, b4 = t4 < tt;
        t = !b1 ? t1 : (!b2 ? t2 : (!b3 ? t3 : t4));
        adds += !b1 ? 1u : (!b2 ? 2u : (!b3 ? 3u : 4u));
        if (!(b1 && b2 && b3 && b4)) break;
    }
    return adds;
}

__device__ __forceinline__ RmRay rm_load_ray(const float *rays_o, const float *rays_d, uint32_t n) {
    RmRay r;
    r.ox = rays_o[n * 3 + 0]; r.oy = rays_o[n * 3 + 1]; r.oz = rays_o[n * 3 + 2];
    r.dx = rays_d[n * 3 + 0]; r.dy = rays_d[n * 3 + 1]; r.dz = rays_d[n * 3 + 2];
    r.rdx = 1 / r.dx; r.rdy = 1 / r.dy; r.rdz = 1 / r.dz;
    return r;
}

// ---------------------------------------------------------------------------------------------
// utilities
// ---------------------------------------------------------------------------------------------
// raymarching.cu:190-244
__global__ void k_near_far_from_aabb(const float *__restrict__ rays_o, const float *__restrict__ rays_d,
                                     const float *__restrict__ aabb, uint32_t N, float min_near,
                                     float *__restrict__ nears, float *__restrict__ fars) {
#pragma clang fp contract(off)
    const float a0 = aabb[0], a1 = aabb[1], a2 = aabb[2], a3 = aabb[3], a4 = aabb[4], a5 = aabb[5];
    for (uint32_t n = blockIdx.x * blockDim.x + threadIdx.x; n < N; n += gridDim.x * blockDim.x) {
        const RmRay r = rm_load_ray(rays_o, rays_d, n);
        const float big = 3.402823466e+38f;
        float near = (a0 - r.ox) * r.rdx, far = (a3 - r.ox) * r.rdx;
        if (near > far) { const float c = near; near = far; far = c; }
        float near_y = (a1 - r.oy) * r.rdy, far_y = (a4 - r.oy) * r.rdy;
        if (near_y > far_y) { const float c = near_y; near_y = far_y; far_y = c; }
        if (near > far_y || near_y > far) { nears[n] = big; fars[n] = big; continue; }
        if (near_y > near) near = near_y;
        if (far_y < far) far = far_y;
        float near_z = (a2 - r.oz) * r.rdz, far_z = (a5 - r.oz) * r.rdz;
        if (near_z > far_z) { const float c = near_z; near_z = far_z; far_z = c; }
        if (near > far_z || near_z > far) { nears[n] = big; fars[n] = big; continue; }
        if (near_z > near) near = near_z;
        if (far_z < far) far = far_z;
        if (near < min_near) near = min_near;
        nears[n] = near;
        fars[n] = far;
    }
}

// raymarching.cu:313-325
__global__ void k_morton3d(const int32_t *__restrict__ coords, uint32_t N, int32_t *__restrict__ indices) {
    for (uint32_t n = blockIdx.x * blockDim.x + threadIdx.x; n < N; n += gridDim.x * blockDim.x)
        indices[n] = (int32_t)rm_morton3d((uint32_t)coords[n * 3], (uint32_t)coords[n * 3 + 1],
                                          (uint32_t)coords[n * 3 + 2]);
}

// raymarching.cu:336-353
__global__ void k_morton3d_invert(const int32_t *__restrict__ indices, uint32_t N, int32_t *__restrict__ coords) {
    for (uint32_t n = blockIdx.x * blockDim.x + threadIdx.x; n < N; n += gridDim.x * blockDim.x) {
        const int32_t ind = indices[n];
        coords[n * 3 + 0] = (int32_t)rm_morton3d_invert((uint32_t)(ind >> 0));
        coords[n * 3 + 1] = (int32_t)rm_morton3d_invert((uint32_t)(ind >> 1));
        coords[n * 3 + 2] = (int32_t)rm_morton3d_invert((uint32_t)(ind >> 2));
    }
}

// raymarching.cu:366-388.  One thread per output byte, 8 floats in as two 16-byte loads.
__global__ void k_packbits(const float *__restrict__ grid, uint32_t N, float thresh, uint8_t *__restrict__ bitfield) {
    for (uint32_t n = blockIdx.x * blockDim.x + threadIdx.x; n < N; n += gridDim.x * blockDim.x) {
        const float4 a = reinterpret_cast<const float4 *>(grid)[(size_t)n * 2];
        const float4 b = reinterpret_cast<const float4 *>(grid)[(size_t)n * 2 + 1];
        uint32_t bits = 0;
        bits |= (a.x > thresh) ? 1u : 0u;
        bits |= (a.y > thresh) ? 2u : 0u;
        bits |= (a.z > thresh) ? 4u : 0u;
        bits |= (a.w > thresh) ? 8u : 0u;
        bits |= (b.x > thresh) ? 16u : 0u;
        bits |= (b.y > thresh) ? 32u : 0u;
        bits |= (b.z > thresh) ? 64u : 0u;
        bits |= (b.w > thresh) ? 128u : 0u;
        bitfield[n] = (uint8_t)bits;
    }
}

// ---------------------------------------------------------------------------------------------
// training march: count -> scan -> emit
// ---------------------------------------------------------------------------------------------
// pass 1 (raymarching.cu:455-501): per-ray sample count, per-block total.
__global__ void __launch_bounds__(RM_BLOCK)
k_march_count(const float *__restrict__ rays_o, const float *__restrict__ rays_d, const uint8_t *__restrict__ grid,
              float bound, float dt_gamma, uint32_t max_steps, uint32_t N, uint32_t C, uint32_t H,
              const float *__restrict__ nears, const float *__restrict__ fars, const float *__restrict__ noises,
              uint32_t *__restrict__ counts, uint32_t *__restrict__ block_sums, uint32_t *__restrict__ mask, uint32_t kcap) {
    __shared__ uint32_t wave_sums[RM_BLOCK / 64];
    const uint32_t n = blockIdx.x * RM_BLOCK + threadIdx.x;
    uint32_t num_steps = 0;
    if (n < N) {
        const RmCfg c = rm_cfg(bound, dt_gamma, max_steps, C, H, grid);
        const RmRay r = rm_load_ray(rays_o, rays_d, n);
        const float far = fars[n];
        float t = nears[n];
        {
#pragma clang fp contract(off)
            const float noise = noises ? noises[n] : 0.0f;
            t += rm_clamp(t * dt_gamma, c.dt_min, c.dt_max) * noise;   // :452
        }
        float x, y, z, dt, tt;
        if (mask == nullptr) {
            while (t < far && num_steps < max_steps) {
                if (rm_probe(r, c, t, x, y, z, dt, tt)) {
                    num_steps++;
                    t += dt;
                } else {
                    rm_skip(c, t, tt);
                }
            }
        } else {
            // Round 3: the probes are made ONCE.  Every parameter the loop visits is an element of the occupancy-independent
            // sequence t_0, t_{k+1} = t_k + clamp(t_k * dt_gamma, dt_min, dt_max) (both branches advance t by exactly that
            // expression), so the samples are fully described by WHICH k they sit at: a bit mask per ray, word w of ray n at
            // mask[w * N + n].  k_march_emit_mask replays the sequence (one addition per k, no probe) and emits the marked
            // elements -- bit-identical to re-marching, at a twentieth of the instructions.
            uint32_t k = 0, word = 0, wi = 0;
            const uint32_t wmax = (kcap + 31u) / 32u;
            while (t < far && num_steps < max_steps && k < kcap) {
                if (rm_probe(r, c, t, x, y, z, dt, tt)) {
                    word |= 1u << (k & 31u);
                    num_steps++;
                    t += dt;
                    k++;
                } else {
                    k += rm_skip(c, t, tt);
                }
                while (wi < (k >> 5) && wi < wmax) {     // completed words (a skip may pass several: zeros)
                    mask[(size_t)wi * N + n] = word;
                    word = 0;
                    wi++;
                }
            }
            if (wi <= wmax) mask[(size_t)wi * N + n] = word;      // (the mask holds wmax + 1 words per ray)
        }
        counts[n] = num_steps;
    }
    uint32_t total;
    rm_block_exclusive_scan(num_steps, wave_sums, total);
    if (threadIdx.x == 0) block_sums[blockIdx.x] = total;
}

// Exclusive scan of the per-block totals by ONE block; also applies the reference's counter
// semantics (atomicAdd(counter, num_steps) / atomicAdd(counter+1, 1), :506-507): block bases
// start at the incoming counter[0]; counter[0] += total, counter[1] += N.
__global__ void __launch_bounds__(1024)
k_scan_block_sums(uint32_t *__restrict__ block_sums, uint32_t nblocks, int32_t *__restrict__ counter, uint32_t N) {
    __shared__ uint32_t wave_sums[1024 / 64];
    __shared__ uint32_t carry_s;
    if (threadIdx.x == 0) carry_s = counter ? (uint32_t)counter[0] : 0u;
    __syncthreads();
    const uint32_t base0 = carry_s;
    uint32_t carry = base0;
    for (uint32_t start = 0; start < nblocks; start += 1024) {
        const uint32_t i = start + threadIdx.x;
        const uint32_t v = i < nblocks ? block_sums[i] : 0u;
        uint32_t total;
        const uint32_t ex = rm_block_exclusive_scan(v, wave_sums, total);
        if (i < nblocks) block_sums[i] = carry + ex;
        carry += total;
    }
    if (threadIdx.x == 0 && counter) {
        counter[0] = (int32_t)carry;
        counter[1] += (int32_t)N;
    }
}

// pass 2 (raymarching.cu:505-588): offsets from the scan, then re-march and emit.
__global__ void __launch_bounds__(RM_BLOCK)
k_march_emit(const float *__restrict__ rays_o, const float *__restrict__ rays_d, const float *__restrict__ z_hats,
             const uint8_t *__restrict__ grid, float bound, float dt_gamma, uint32_t max_steps, int is_ndc,
             uint32_t N, uint32_t C, uint32_t H, uint32_t M, const float *__restrict__ nears,
             const float *__restrict__ fars, const float *__restrict__ noises, const uint32_t *__restrict__ counts,
             const uint32_t *__restrict__ block_bases, uint32_t ray_base, float *__restrict__ xyzs,
             float *__restrict__ dirs, float *__restrict__ deltas, int32_t *__restrict__ rays) {
    __shared__ uint32_t wave_sums[RM_BLOCK / 64];
    const uint32_t n = blockIdx.x * RM_BLOCK + threadIdx.x;
    const uint32_t num_steps = n < N ? counts[n] : 0u;
    uint32_t total;
    const uint32_t point_index = block_bases[blockIdx.x] + rm_block_exclusive_scan(num_steps, wave_sums, total);
    if (n >= N) return;
    const uint32_t ray_index = ray_base + n;
    rays[ray_index * 3 + 0] = (int32_t)n;
    rays[ray_index * 3 + 1] = (int32_t)point_index;
    rays[ray_index * 3 + 2] = (int32_t)num_steps;
    if (num_steps == 0) return;
    if (point_index + num_steps >= M) {          // :517
        // dropped ray: the reference's buffers are zero-filled before the launch (raymarching.py:238-240), here they
        // are torch.empty -- write the in-buffer part, so that no consumer bounded by min(counter[0], M) ever
        // reads uninitialised positions (a NaN bit pattern would poison the weight gradients as 0 * NaN)
        for (uint32_t i = point_index; i < min(point_index + num_steps, M); i++) {
            xyzs[(size_t)i * 3 + 0] = 0.f; xyzs[(size_t)i * 3 + 1] = 0.f; xyzs[(size_t)i * 3 + 2] = 0.f;
            if (dirs) { dirs[(size_t)i * 3 + 0] = 0.f; dirs[(size_t)i * 3 + 1] = 0.f; dirs[(size_t)i * 3 + 2] = 0.f; }
            reinterpret_cast<float4 *>(deltas)[i] = make_float4(0.f, 0.f, 0.f, 0.f);
        }
        return;
    }

    const RmCfg c = rm_cfg(bound, dt_gamma, max_steps, C, H, grid);
    const RmRay r = rm_load_ray(rays_o, rays_d, n);
    const float far = fars[n];
    float t = nears[n];
    {
#pragma clang fp contract(off)
        const float noise = noises ? noises[n] : 0.0f;
        t += rm_clamp(t * dt_gamma, c.dt_min, c.dt_max) * noise;
    }
    float *pxyz = xyzs + (size_t)point_index * 3;
    float *pdir = dirs ? dirs + (size_t)point_index * 3 : nullptr;
    float *pdel = deltas + (size_t)point_index * 4;
    uint32_t step = 0;
    float last_t = t;
    float last_z = rm_clamp(r.oz + t * r.dz, -bound, bound);
    float x, y, z, dt, tt;
    while (t < far && step < num_steps) {
        if (rm_probe(r, c, t, x, y, z, dt, tt)) {
#pragma clang fp contract(off)
            pxyz[0] = x; pxyz[1] = y; pxyz[2] = z;
            if (pdir) { pdir[0] = r.dx; pdir[1] = r.dy; pdir[2] = r.dz; pdir += 3; }
            t += dt;
            if (is_ndc) {
                const float new_z = rm_clamp(r.oz + t * r.dz, -bound, bound);
                const float zh = z_hats[n];
                reinterpret_cast<float4 *>(pdel)[0] = make_float4(dt, t - last_t, (2 / (new_z - 1) - 2 / (z - 1)) / zh,
                                                                  (2 / (new_z - 1) - 2 / (last_z - 1)) / zh);
                last_z = z;   // :570
            } else {
                reinterpret_cast<float2 *>(pdel)[0] = make_float2(dt, t - last_t);
            }
            last_t = t;
            pxyz += 3; pdel += 4;
            step++;
        } else {
            rm_skip(c, t, tt);
        }
    }
}

// pass 2 without probes: replays the t sequence and emits the elements k_march_count marked (see there).  Not for NDC
// (its deltas need the previous sample's z, k_march_emit keeps that form).
__global__ void __launch_bounds__(RM_BLOCK)
k_march_emit_mask(const float *__restrict__ rays_o, const float *__restrict__ rays_d, float bound, float dt_gamma,
                  uint32_t max_steps, uint32_t N, uint32_t C, uint32_t H, uint32_t M, const float *__restrict__ nears,
                  const float *__restrict__ noises, const uint32_t *__restrict__ counts, const uint32_t *__restrict__ block_bases,
                  const uint32_t *__restrict__ mask, float *__restrict__ xyzs, float *__restrict__ dirs,
                  float *__restrict__ deltas, int32_t *__restrict__ rays, int closed_form) {
    __shared__ uint32_t wave_sums[RM_BLOCK / 64];
    const uint32_t n = blockIdx.x * RM_BLOCK + threadIdx.x;
    const uint32_t num_steps = n < N ? counts[n] : 0u;
    uint32_t total;
    const uint32_t point_index = block_bases[blockIdx.x] + rm_block_exclusive_scan(num_steps, wave_sums, total);
    if (n >= N) return;
    rays[n * 3 + 0] = (int32_t)n;
    rays[n * 3 + 1] = (int32_t)point_index;
    rays[n * 3 + 2] = (int32_t)num_steps;
    if (num_steps == 0) return;
    if (point_index + num_steps >= M) {          // :517, see k_march_emit
        for (uint32_t i = point_index; i < min(point_index + num_steps, M); i++) {
            xyzs[(size_t)i * 3 + 0] = 0.f; xyzs[(size_t)i * 3 + 1] = 0.f; xyzs[(size_t)i * 3 + 2] = 0.f;
            if (dirs) { dirs[(size_t)i * 3 + 0] = 0.f; dirs[(size_t)i * 3 + 1] = 0.f; dirs[(size_t)i * 3 + 2] = 0.f; }
            reinterpret_cast<float4 *>(deltas)[i] = make_float4(0.f, 0.f, 0.f, 0.f);
        }
        return;
    }
    const RmCfg c = rm_cfg(bound, dt_gamma, max_steps, C, H, nullptr);
    const RmRay r = rm_load_ray(rays_o, rays_d, n);
    float t = nears[n];
    {
#pragma clang fp contract(off)
        const float noise = noises ? noises[n] : 0.0f;
        t += rm_clamp(t * dt_gamma, c.dt_min, c.dt_max) * noise;
    }
    float *pxyz = xyzs + (size_t)point_index * 3;
    float *pdir = dirs ? dirs + (size_t)point_index * 3 : nullptr;
    float *pdel = deltas + (size_t)point_index * 4;
    uint32_t step = 0;
    float last_t = t;
    for (uint32_t w = 0; step < num_steps; w++) {
        uint32_t word = mask[(size_t)w * N + n];
        // Constant step (dt_gamma = 0) inside one binade: the word's 32 parameters are t, t_1 = t + dt and t_j = bits(t_1) +
        // (j - 1) (bits(t_2) - bits(t_1)) -- the floats the 32 serial additions give (wpr_block_t has the argument) -- so only the
        // MARKED steps cost instructions.  Words that cross a binade keep the serial form.
        bool closed = false;
        uint32_t b1 = 0, cc = 0;
        if (closed_form && dt_gamma == 0.0f && t > 0.0f) {
#pragma clang fp contract(off)
            const float dt0 = rm_clamp(t * dt_gamma, c.dt_min, c.dt_max);
            const float t1 = t + dt0, t2 = t1 + dt0;
            b1 = __float_as_uint(t1);
            const uint32_t b2 = __float_as_uint(t2);
            cc = b2 - b1;
            const uint32_t b32 = b1 + 31u * cc;
            closed = b2 > b1 && (b1 >> 23) == (b32 >> 23) && cc < (1u << 23);
        }
        if (closed) {
            while (word != 0u && step < num_steps) {
#pragma clang fp contract(off)
                const uint32_t k = (uint32_t)__builtin_ctz(word);
                word &= word - 1u;
                const float tk = k == 0u ? t : __uint_as_float(b1 + (k - 1u) * cc);
                const float dt = rm_clamp(tk * dt_gamma, c.dt_min, c.dt_max);
                const float t_next = tk + dt;
                pxyz[0] = rm_clamp(r.ox + tk * r.dx, -bound, bound);
                pxyz[1] = rm_clamp(r.oy + tk * r.dy, -bound, bound);
                pxyz[2] = rm_clamp(r.oz + tk * r.dz, -bound, bound);
                if (pdir) { pdir[0] = r.dx; pdir[1] = r.dy; pdir[2] = r.dz; pdir += 3; }
                reinterpret_cast<float2 *>(pdel)[0] = make_float2(dt, t_next - last_t);
                last_t = t_next;
                pxyz += 3; pdel += 4;
                step++;
            }
            t = __uint_as_float(b1 + 31u * cc);                  // t_32: the next word's first parameter
            continue;
        }
        for (uint32_t b = 0; b < 32u && step < num_steps; b++, word >>= 1) {
#pragma clang fp contract(off)
            const float dt = rm_clamp(t * dt_gamma, c.dt_min, c.dt_max);
            const float t_next = t + dt;
            if (word & 1u) {
                pxyz[0] = rm_clamp(r.ox + t * r.dx, -bound, bound);
                pxyz[1] = rm_clamp(r.oy + t * r.dy, -bound, bound);
                pxyz[2] = rm_clamp(r.oz + t * r.dz, -bound, bound);
                if (pdir) { pdir[0] = r.dx; pdir[1] = r.dy; pdir[2] = r.dz; pdir += 3; }
                reinterpret_cast<float2 *>(pdel)[0] = make_float2(dt, t_next - last_t);
                last_t = t_next;
                pxyz += 3; pdel += 4;
                step++;
            }
            t = t_next;
        }
    }
}

// ---------------------------------------------------------------------------------------------
// training march, one WAVE per ray (small batches)
// ---------------------------------------------------------------------------------------------
// The thread-per-ray march above is a serial chain of ~200-instruction probes per ray: 0.85 ms for a
// 4096-ray batch however idle the chip is.  The parameters the reference visits are the ray-independent
// sequence t_0 = near (+ noise), t_{k+1} = t_k + clamp(t_k * dt_gamma, dt_min, dt_max) -- both branches of
// the loop (:484-498) advance t by exactly that expression -- and the loop only decides WHICH t_k it
// probes: after an occupied probe the next one, after an empty probe the first t_j >= tt.  So a wave
// takes 64 consecutive t_k, all lanes probe speculatively (same rm_probe, same rounding), every empty
// lane finds its successor j by binary search over the wave's t values, and a scalar walk follows the
// successor links from the entry point: exactly the probes, in exactly the order, of the serial loop.
// Lanes the walk visits and finds occupied are the samples.

// per-block totals of counts[] (the tail of k_march_count)
__global__ void __launch_bounds__(RM_BLOCK)
k_march_block_sums(const uint32_t *__restrict__ counts, uint32_t N, uint32_t *__restrict__ block_sums) {
    __shared__ uint32_t wave_sums[RM_BLOCK / 64];
    const uint32_t n = blockIdx.x * RM_BLOCK + threadIdx.x;
    uint32_t total;
    rm_block_exclusive_scan(n < N ? counts[n] : 0u, wave_sums, total);
    if (threadIdx.x == 0) block_sums[blockIdx.x] = total;
}

// The 64 consecutive march parameters that start at t_block -- lane j gets t_j of t_0 = t_block, t_{j+1} = t_j + clamp(t_j *
// dt_gamma, dt_min, dt_max) -- and t_64 (the next block's start).  Shared by the counting / re-marching kernel and the replaying
// emit, so that both see the same floats.
__device__ __forceinline__ void wpr_block_t(const RmCfg &c, float dt_gamma, float t_block, uint32_t lane, float &my_t, float &t_next) {
    float tc = t_block;
    my_t = t_block;
    bool closed = false;
    if (dt_gamma == 0.0f && t_block > 0.0f) {
        // Constant step (LLFF: dt_gamma = 0): t_{j+1} = fl(t_j + dt).  Inside one binade every t_j is a multiple of the
        // binade's ulp u, so the rounded sum advances the BIT PATTERN by a constant c = dt / u rounded to an integer (ties
        // to even make the very first step the only possible exception: after it the mantissa parity repeats).  Two real
        // additions give t_1 and t_2, c = bits(t_2) - bits(t_1), and t_j = bits(t_1) + (j - 1) c for j = 1..64 -- the same
        // floats as the 64 serial additions below, as long as t_1 .. t_64 share an exponent (else: the serial loop).
#pragma clang fp contract(off)
        const float dt0 = rm_clamp(t_block * dt_gamma, c.dt_min, c.dt_max);
        const float t1 = t_block + dt0, t2 = t1 + dt0;
        const uint32_t b1 = __float_as_uint(t1), b2 = __float_as_uint(t2), cc = b2 - b1;
        const uint32_t b64 = b1 + 63u * cc;
        if (b2 > b1 && (b1 >> 23) == (b64 >> 23) && cc < (1u << 23)) {
            closed = true;
            my_t = lane == 0 ? t_block : __uint_as_float(b1 + (lane - 1u) * cc);
            tc = __uint_as_float(b64);
        }
    }
    if (!closed) {
#pragma clang fp contract(off)
        for (uint32_t j = 0; j < 64; j++) {
            if (lane == j) my_t = tc;
            tc += rm_clamp(tc * dt_gamma, c.dt_min, c.dt_max);
        }
    }
    t_next = tc;
}

template <bool EMIT>
__global__ void __launch_bounds__(256)
k_march_wpr(const float *__restrict__ rays_o, const float *__restrict__ rays_d, const uint8_t *__restrict__ grid, float bound,
            float dt_gamma, uint32_t max_steps, uint32_t N, uint32_t C, uint32_t H, uint32_t M, const float *__restrict__ nears,
            const float *__restrict__ fars, const float *__restrict__ noises, uint32_t *__restrict__ counts,
            const uint32_t *__restrict__ block_bases, float *__restrict__ xyzs, float *__restrict__ dirs,
            float *__restrict__ deltas, int32_t *__restrict__ rays, uint32_t *__restrict__ slots, uint32_t slot_cap) {
    __shared__ float t_lds[4][64];
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    const uint32_t n = blockIdx.x * 4 + wave;
    if (n >= N) return;                                     // wave-uniform
    float *tl = t_lds[wave];
    const RmCfg c = rm_cfg(bound, dt_gamma, max_steps, C, H, grid);
    const RmRay r = rm_load_ray(rays_o, rays_d, n);
    const float far = fars[n];
    float t_block = nears[n];
    {
#pragma clang fp contract(off)
        const float noise = noises ? noises[n] : 0.0f;
        t_block += rm_clamp(t_block * dt_gamma, c.dt_min, c.dt_max) * noise;   // :452
    }
    uint32_t limit = max_steps, point_index = 0;
    if (EMIT) {
        // offsets: base of this ray's 256-ray block + the counts of the rays before it in the block
        const uint32_t blk0 = (n / RM_BLOCK) * RM_BLOCK;
        uint32_t part = 0;
        for (uint32_t i = blk0 + lane; i < n; i += 64) part += counts[i];
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) part += __shfl_xor(part, off, 64);
        point_index = block_bases[n / RM_BLOCK] + part;
        limit = counts[n];
        if (lane == 0) {
            rays[n * 3 + 0] = (int32_t)n;
            rays[n * 3 + 1] = (int32_t)point_index;
            rays[n * 3 + 2] = (int32_t)limit;
        }
        if (limit == 0) return;
        if (point_index + limit >= M) {          // :517; zero the in-buffer part of a dropped ray (see k_march_emit)
            for (uint32_t i = point_index + lane; i < min(point_index + limit, M); i += 64) {
                xyzs[(size_t)i * 3 + 0] = 0.f; xyzs[(size_t)i * 3 + 1] = 0.f; xyzs[(size_t)i * 3 + 2] = 0.f;
                if (dirs) { dirs[(size_t)i * 3 + 0] = 0.f; dirs[(size_t)i * 3 + 1] = 0.f; dirs[(size_t)i * 3 + 2] = 0.f; }
                reinterpret_cast<float4 *>(deltas)[i] = make_float4(0.f, 0.f, 0.f, 0.f);
            }
            return;
        }
    }
    uint32_t steps = 0;
    float carry_tt = -INFINITY;       // the walk enters a block at its first t >= carry_tt
    float last_t = t_block;           // :530, t after the previous sample's step
    bool done = false;
    // (counting pass with `slots`: every block's start and sample mask are recorded -- [slot_cap][3] words per ray after the N
    // block counts -- and k_march_wpr_replay emits from them without probing again; slot_cap * 64 >= march_kcap, the bound the
    // thread-per-ray mask relies on)
    uint32_t nblk = 0;
    while (!done && t_block < far && (EMIT || slots == nullptr || nblk < slot_cap)) {
        // ---- the block's 64 parameters ----
        float my_t, tc;
        wpr_block_t(c, dt_gamma, t_block, lane, my_t, tc);
        const float t_next_block = tc;
        // ---- speculative probe ----
        const bool valid = my_t < far;
        float x = 0, y = 0, z = 0, dt = 0, tt = 0;
        const bool occ = valid && rm_probe(r, c, valid ? my_t : t_block, x, y, z, dt, tt);
        // ---- successor of an empty lane: first j > lane with t_j >= tt (do { t += dt; } while (t < tt), :497) ----
        tl[lane] = my_t;
        __builtin_amdgcn_wave_barrier();
        uint32_t lo = lane + 1, hi = 64;
#pragma unroll
        for (int it = 0; it < 6; it++) {
            const uint32_t mid = (lo + hi) >> 1;
            const float tm = tl[mid < 64 ? mid : 63];
            const bool go = lo < hi && tm < tt;
            const bool stay = lo < hi && !(tm < tt);
            lo = go ? mid + 1 : lo;
            hi = stay ? mid : hi;
        }
        const uint32_t nxt = occ ? lane + 1 : lo;
        // ---- walk the chain ----
        const unsigned long long occ_mask = __ballot(occ), valid_mask = __ballot(valid);
        uint32_t cur = 0;
        if (carry_tt != -INFINITY) {
            const unsigned long long ge = __ballot(my_t >= carry_tt);
            cur = ge ? (uint32_t)__builtin_ctzll(ge) : 64u;
        }
        unsigned long long sample_mask = 0ull;
        uint32_t last = 64;
        while (cur < 64) {
            if (!((valid_mask >> cur) & 1ull)) { done = true; break; }          // t >= far
            if ((occ_mask >> cur) & 1ull) {
                sample_mask |= 1ull << cur;
                steps++;
                if (steps == limit) { done = true; break; }
            }
            last = cur;
            cur = (uint32_t)__builtin_amdgcn_readlane((int)nxt, (int)cur);
        }
        if (!done && last < 64)
            carry_tt = ((occ_mask >> last) & 1ull) ? -INFINITY : __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, tt), (int)last));
        // ---- emit this block's samples ----
        if (EMIT && sample_mask) {
#pragma clang fp contract(off)
            const bool mine = (sample_mask >> lane) & 1ull;
            const uint32_t before = (uint32_t)__popcll(sample_mask & ((1ull << lane) - 1ull));
            const uint32_t first_idx = steps - (uint32_t)__popcll(sample_mask);       // samples emitted before this block
            const float t_after = my_t + dt;                                            // t += dt, :551
            __builtin_amdgcn_wave_barrier();
            tl[lane] = t_after;
            __builtin_amdgcn_wave_barrier();
            // t after the previous sample's step: previous sample lane of this block, or the carried one
            const unsigned long long below = sample_mask & ((1ull << lane) - 1ull);
            const float prev_t = below ? tl[63 - __builtin_clzll(below)] : last_t;
            if (mine) {
                const size_t o = (size_t)point_index + first_idx + before;
                xyzs[o * 3 + 0] = x; xyzs[o * 3 + 1] = y; xyzs[o * 3 + 2] = z;
                if (dirs) { dirs[o * 3 + 0] = r.dx; dirs[o * 3 + 1] = r.dy; dirs[o * 3 + 2] = r.dz; }
                reinterpret_cast<float2 *>(deltas + o * 4)[0] = make_float2(dt, t_after - prev_t);
            }
            last_t = tl[63 - __builtin_clzll(sample_mask)];
        }
        __builtin_amdgcn_wave_barrier();
        if (!EMIT && slots != nullptr) {
            if (lane < 3) {
                const uint32_t wv = lane == 0 ? __float_as_uint(t_block) : (lane == 1 ? (uint32_t)sample_mask : (uint32_t)(sample_mask >> 32));
                slots[(size_t)N + ((size_t)n * slot_cap + nblk) * 3 + lane] = wv;
            }
            nblk++;
        }
        t_block = t_next_block;
    }
    if (!EMIT && lane == 0) {
        counts[n] = steps;
        if (slots != nullptr) slots[n] = nblk;
    }
}

// Emitting pass of the wave-per-ray march from the counting pass's records: no occupancy probe, no successor search, no chain
// walk -- per recorded block with samples: the 64 parameters again (wpr_block_t), positions and step sizes of the marked lanes,
// the same stores as k_march_wpr<true>.  Bit-identical output; 4 096-ray bf16 + graph step 1.09 -> 0.96 ms.
__global__ void __launch_bounds__(256)
k_march_wpr_replay(const float *__restrict__ rays_o, const float *__restrict__ rays_d, float bound, float dt_gamma, uint32_t max_steps,
                   uint32_t N, uint32_t C, uint32_t H, uint32_t M, const uint32_t *__restrict__ counts,
                   const uint32_t *__restrict__ block_bases, const uint32_t *__restrict__ slots, uint32_t slot_cap,
                   float *__restrict__ xyzs, float *__restrict__ dirs, float *__restrict__ deltas, int32_t *__restrict__ rays) {
    __shared__ float t_lds[4][64];
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    const uint32_t n = blockIdx.x * 4 + wave;
    if (n >= N) return;                                     // wave-uniform
    float *tl = t_lds[wave];
    const RmCfg c = rm_cfg(bound, dt_gamma, max_steps, C, H, nullptr);
    const RmRay r = rm_load_ray(rays_o, rays_d, n);
    const uint32_t blk0 = (n / RM_BLOCK) * RM_BLOCK;
    uint32_t part = 0;
    for (uint32_t i = blk0 + lane; i < n; i += 64) part += counts[i];
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) part += __shfl_xor(part, off, 64);
    const uint32_t point_index = block_bases[n / RM_BLOCK] + part;
    const uint32_t limit = counts[n];
    if (lane == 0) {
        rays[n * 3 + 0] = (int32_t)n;
        rays[n * 3 + 1] = (int32_t)point_index;
        rays[n * 3 + 2] = (int32_t)limit;
    }
    if (limit == 0) return;
    if (point_index + limit >= M) {          // :517; zero the in-buffer part of a dropped ray (see k_march_emit)
        for (uint32_t i = point_index + lane; i < min(point_index + limit, M); i += 64) {
            xyzs[(size_t)i * 3 + 0] = 0.f; xyzs[(size_t)i * 3 + 1] = 0.f; xyzs[(size_t)i * 3 + 2] = 0.f;
            if (dirs) { dirs[(size_t)i * 3 + 0] = 0.f; dirs[(size_t)i * 3 + 1] = 0.f; dirs[(size_t)i * 3 + 2] = 0.f; }
            reinterpret_cast<float4 *>(deltas)[i] = make_float4(0.f, 0.f, 0.f, 0.f);
        }
        return;
    }
    const uint32_t nblk = min(slots[n], min(slot_cap, 64u));
    // lane b holds block b's record
    const uint32_t *rec = slots + (size_t)N + (size_t)n * slot_cap * 3;
    uint32_t s_t = 0, s_lo = 0, s_hi = 0;
    if (lane < nblk) { s_t = rec[lane * 3]; s_lo = rec[lane * 3 + 1]; s_hi = rec[lane * 3 + 2]; }
    uint32_t steps = 0;
    float last_t = __uint_as_float((uint32_t)__builtin_amdgcn_readfirstlane((int)s_t));      // block 0 starts at the ray's first parameter
    for (uint32_t b = 0; b < nblk; b++) {
        const unsigned long long sample_mask = (unsigned long long)(uint32_t)__builtin_amdgcn_readlane((int)s_lo, (int)b) |
                                               ((unsigned long long)(uint32_t)__builtin_amdgcn_readlane((int)s_hi, (int)b) << 32);
        if (sample_mask == 0ull) continue;
        const float t_block = __uint_as_float((uint32_t)__builtin_amdgcn_readlane((int)s_t, (int)b));
        float my_t, t_next;
        wpr_block_t(c, dt_gamma, t_block, lane, my_t, t_next);
        steps += (uint32_t)__popcll(sample_mask);
        float x, y, z, dt, t_after;
        {
#pragma clang fp contract(off)
            x = rm_clamp(r.ox + my_t * r.dx, -c.bound, c.bound);         // rm_probe's expressions
            y = rm_clamp(r.oy + my_t * r.dy, -c.bound, c.bound);
            z = rm_clamp(r.oz + my_t * r.dz, -c.bound, c.bound);
            dt = rm_clamp(my_t * c.dt_gamma, c.dt_min, c.dt_max);
            t_after = my_t + dt;                                            // t += dt, :551
        }
        const bool mine = (sample_mask >> lane) & 1ull;
        const uint32_t before = (uint32_t)__popcll(sample_mask & ((1ull << lane) - 1ull));
        const uint32_t first_idx = steps - (uint32_t)__popcll(sample_mask);
        __builtin_amdgcn_wave_barrier();
        tl[lane] = t_after;
        __builtin_amdgcn_wave_barrier();
        const unsigned long long below = sample_mask & ((1ull << lane) - 1ull);
        const float prev_t = below ? tl[63 - __builtin_clzll(below)] : last_t;
        if (mine) {
            const size_t o = (size_t)point_index + first_idx + before;
            xyzs[o * 3 + 0] = x; xyzs[o * 3 + 1] = y; xyzs[o * 3 + 2] = z;
            if (dirs) { dirs[o * 3 + 0] = r.dx; dirs[o * 3 + 1] = r.dy; dirs[o * 3 + 2] = r.dz; }
            float dprev;
            {
#pragma clang fp contract(off)
                dprev = t_after - prev_t;
            }
            reinterpret_cast<float2 *>(deltas + o * 4)[0] = make_float2(dt, dprev);
        }
        last_t = tl[63 - __builtin_clzll(sample_mask)];
        __builtin_amdgcn_wave_barrier();
    }
}

// ---------------------------------------------------------------------------------------------
// compositing: the training composite lives in composite.hip (wave per ray, lane per sample)
// ---------------------------------------------------------------------------------------------
#define RM_MAXC 16

// Single-pass inference composite: the arithmetic of kernel_composite_rays (raymarching.cu:1133-1231)
// -- T = 1 - weight_sum, stop test on T BEFORE the sample (:1206), absolute t starting at near --
// applied to the compacted [offset, offset+count) samples of the training-style march instead of
// up to 1024 host-driven march_rays / composite_rays iterations (renderer.py:266-285).
template <int LPR>
__global__ void __launch_bounds__(RM_BLOCK)
k_composite_infer(const float *__restrict__ sigmas, const float *__restrict__ rgbs, const float *__restrict__ deltas,
                  const int32_t *__restrict__ rays, const float *__restrict__ nears, uint32_t M, uint32_t N, uint32_t C,
                  float T_thresh, float *__restrict__ weights_sum, float *__restrict__ depth, float *__restrict__ image) {
    const uint32_t tid = blockIdx.x * RM_BLOCK + threadIdx.x;
    const uint32_t n = tid / LPR, ch = tid % LPR;
    if (n >= N) return;
    const uint32_t index = (uint32_t)rays[n * 3], offset = (uint32_t)rays[n * 3 + 1], num_steps = (uint32_t)rays[n * 3 + 2];
    const bool has_ch = ch < C;
    float acc = 0.0f, ws = 0.0f, d = 0.0f;
    if (!(num_steps == 0 || offset + num_steps >= M)) {
        float t_phy = nears[index];
        const float *s = sigmas + offset;
        const float *rgb = rgbs + (size_t)offset * C + (has_ch ? ch : 0);
        const float *dl = deltas + (size_t)offset * 4;
        for (uint32_t step = 0; step < num_steps; step++) {
            const float2 dd = *reinterpret_cast<const float2 *>(dl + step * 4);
            const float alpha = 1.0f - __expf(-s[step] * dd.x);
            const float T = 1 - ws;
            const float weight = alpha * T;
            ws += weight;
            t_phy += dd.y;
            d += weight * t_phy;
            if (has_ch) acc += weight * rgb[(size_t)step * C];
            if (T < T_thresh) break;   // :1206
        }
    }
    if (ch == 0) {
        weights_sum[index] = ws;
        depth[index] = d;
    }
    if (has_ch) image[(size_t)index * C + ch] = acc;
}

// ---------------------------------------------------------------------------------------------
// inference march / composite (raymarching.cu:1004-1120, 1133-1231)
// ---------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(RM_BLOCK)
k_march_rays(uint32_t n_alive, uint32_t n_step, const int32_t *__restrict__ rays_alive, const float *__restrict__ rays_t,
             const float *__restrict__ rays_o, const float *__restrict__ rays_d, const float *__restrict__ z_hats,
             float bound, float dt_gamma, uint32_t max_steps, int is_ndc, uint32_t C, uint32_t H,
             const uint8_t *__restrict__ grid, const float *__restrict__ fars, float *__restrict__ xyzs,
             float *__restrict__ dirs, float *__restrict__ deltas, const float *__restrict__ noises) {
    const uint32_t n = blockIdx.x * RM_BLOCK + threadIdx.x;
    if (n >= n_alive) return;
    const int index = rays_alive[n];
    const RmCfg c = rm_cfg(bound, dt_gamma, max_steps, C, H, grid);
    const RmRay r = rm_load_ray(rays_o, rays_d, (uint32_t)index);
    float *pxyz = xyzs + (size_t)n * n_step * 3;
    float *pdir = dirs ? dirs + (size_t)n * n_step * 3 : nullptr;
    float *pdel = deltas + (size_t)n * n_step * 4;
    float t = rays_t[(size_t)index * (is_ndc ? 2 : 1)];
    const float far = fars[index];
    {
#pragma clang fp contract(off)
        const float noise = noises ? noises[n] : 0.0f;
        t += rm_clamp(t * dt_gamma, c.dt_min, c.dt_max) * noise;   // :1053
    }
    uint32_t step = 0;
    float last_t = t;
    float last_z = rm_clamp(r.oz + t * r.dz, -bound, bound);
    float x, y, z, dt, tt;
    while (t < far && step < n_step) {
        if (rm_probe(r, c, t, x, y, z, dt, tt)) {
#pragma clang fp contract(off)
            pxyz[0] = x; pxyz[1] = y; pxyz[2] = z;
            if (pdir) { pdir[0] = r.dx; pdir[1] = r.dy; pdir[2] = r.dz; pdir += 3; }
            t += dt;
            pdel[0] = dt;
            pdel[1] = t - last_t;
            if (is_ndc) {
                const float new_z = rm_clamp(r.oz + t * r.dz, -bound, bound);
                const float zh = z_hats[index];
                pdel[2] = (2 / (new_z - 1) - 2 / (z - 1)) / zh;
                pdel[3] = (2 / (new_z - 1) - 2 / (last_z - 1)) / zh;
                last_z = new_z;
            }
            last_t = t;
            pxyz += 3; pdel += 4;
            step++;
        } else {
            rm_skip(c, t, tt);
        }
    }
    // The reference relies on the caller zero-filling deltas (raymarching.py:409-412) so that an
    // unused tail reads delta == 0 (= "ray terminated", :1178).  Write the terminator here so the
    // caller does not have to memset [n_alive*n_step, 4] floats per iteration.
    for (; step < n_step; step++) {
        pdel[0] = 0.0f;
        pdel += 4;
    }
}

__global__ void __launch_bounds__(RM_BLOCK)
k_composite_rays(uint32_t n_alive, uint32_t n_step, float T_thresh, int32_t *__restrict__ rays_alive,
                 float *__restrict__ rays_t, const float *__restrict__ sigmas, const float *__restrict__ rgbs,
                 const float *__restrict__ deltas, uint32_t C, int is_ndc, float *__restrict__ weights_sum,
                 float *__restrict__ depth, float *__restrict__ image) {
    const uint32_t n = blockIdx.x * RM_BLOCK + threadIdx.x;
    if (n >= n_alive) return;
    const int index = rays_alive[n];
    const float *s = sigmas + (size_t)n * n_step;
    const float *rgb = rgbs + (size_t)n * n_step * C;
    const float *dl = deltas + (size_t)n * n_step * 4;
    float *rt = rays_t + (size_t)index * (is_ndc ? 2 : 1);
    float *img = image + (size_t)index * C;
    float t_rm = 0.0f, t_phy;
    if (is_ndc) { t_rm = rt[0]; t_phy = rt[1]; } else { t_phy = rt[0]; }
    float weight_sum = weights_sum[index];
    float d = depth[index];
    float acc[RM_MAXC];
#pragma unroll
    for (int i = 0; i < RM_MAXC; i++) acc[i] = (uint32_t)i < C ? img[i] : 0.0f;
    uint32_t step = 0;
    while (step < n_step) {
        if (dl[0] == 0) break;   // :1178
        const float alpha = 1.0f - __expf(-s[0] * (is_ndc ? dl[2] : dl[0]));
        const float T = 1 - weight_sum;
        const float weight = alpha * T;
        weight_sum += weight;
        if (is_ndc) { t_rm += dl[1]; t_phy += dl[3]; } else { t_phy += dl[1]; }
        d += weight * t_phy;
#pragma unroll
        for (int i = 0; i < RM_MAXC; i++)
            if ((uint32_t)i < C) acc[i] += weight * rgb[i];
        if (T < T_thresh) break;   // :1206
        s++; rgb += C; dl += 4; step++;
    }
    if (step < n_step) {
        rays_alive[n] = -1;
    } else {
        if (is_ndc) { rt[0] = t_rm; rt[1] = t_phy; } else { rt[0] = t_phy; }
    }
    weights_sum[index] = weight_sum;
    depth[index] = d;
#pragma unroll
    for (int i = 0; i < RM_MAXC; i++)
        if ((uint32_t)i < C) img[i] = acc[i];
}

// ---------------------------------------------------------------------------------------------
// alive-ray compaction (replaces rays_alive[rays_alive >= 0], renderer.py:284)
// ---------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(RM_BLOCK)
k_alive_count(const int32_t *__restrict__ rays_alive, uint32_t n_alive, uint32_t *__restrict__ block_sums) {
    __shared__ uint32_t wave_sums[RM_BLOCK / 64];
    const uint32_t n = blockIdx.x * RM_BLOCK + threadIdx.x;
    const uint32_t keep = (n < n_alive && rays_alive[n] >= 0) ? 1u : 0u;
    // one ballot per wave instead of a shuffle scan: popcount of the 64-bit mask
    const unsigned long long mask = __ballot(keep);
    if ((threadIdx.x & 63u) == 0) wave_sums[threadIdx.x >> 6] = (uint32_t)__popcll(mask);
    __syncthreads();
    if (threadIdx.x == 0) {
        uint32_t t = 0;
        for (uint32_t w = 0; w < RM_BLOCK / 64; w++) t += wave_sums[w];
        block_sums[blockIdx.x] = t;
    }
}

__global__ void __launch_bounds__(RM_BLOCK)
k_alive_write(const int32_t *__restrict__ rays_alive, uint32_t n_alive, const uint32_t *__restrict__ block_bases,
              int32_t *__restrict__ out) {
    __shared__ uint32_t wave_sums[RM_BLOCK / 64];
    const uint32_t n = blockIdx.x * RM_BLOCK + threadIdx.x;
    const int32_t v = n < n_alive ? rays_alive[n] : -1;
    const bool keep = v >= 0;
    const unsigned long long mask = __ballot(keep);
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    if (lane == 0) wave_sums[wave] = (uint32_t)__popcll(mask);
    __syncthreads();
    uint32_t base = block_bases[blockIdx.x];
    for (uint32_t w = 0; w < wave; w++) base += wave_sums[w];
    if (keep) out[base + (uint32_t)__popcll(mask & ((1ull << lane) - 1ull))] = v;
}

__global__ void k_store_scan_total(const int32_t *__restrict__ counter2, int32_t *__restrict__ n_out) {
    n_out[0] = counter2[0];
}

// ---------------------------------------------------------------------------------------------
// device ray generation (nerf_lib.py:69-142, common.py:139-147)
// ---------------------------------------------------------------------------------------------
__global__ void k_generate_rays(const float *__restrict__ pose, uint32_t w, uint32_t h, float fx, float fy, float cx,
                                float cy, int camera_flip, const int32_t *__restrict__ pix, uint32_t N,
                                float *__restrict__ rays_o, float *__restrict__ rays_d) {
#pragma clang fp contract(off)
    const float r00 = pose[0], r01 = pose[1], r02 = pose[2], tx = pose[3];
    const float r10 = pose[4], r11 = pose[5], r12 = pose[6], ty = pose[7];
    const float r20 = pose[8], r21 = pose[9], r22 = pose[10], tz = pose[11];
    const float f0 = (camera_flip >> 2) & 1 ? -1.0f : 1.0f;   // nerf_lib.py:121, bit order [2,1,0]
    const float f1 = (camera_flip >> 1) & 1 ? -1.0f : 1.0f;
    const float f2 = (camera_flip >> 0) & 1 ? -1.0f : 1.0f;
    for (uint32_t n = blockIdx.x * blockDim.x + threadIdx.x; n < N; n += gridDim.x * blockDim.x) {
        const uint32_t p = pix ? (uint32_t)pix[n] : n;
        const uint32_t py = p / w, px = p - py * w;
        // np.linspace(0, w, 2w+1)[1::2] == x + 0.5 exactly in fp32 for w < 2^22
        const float i = (float)px + 0.5f, j = (float)py + 0.5f;
        const float d0 = ((i - cx) / fx) * f0, d1 = ((j - cy) / fy) * f1, d2 = f2;
        const float wx = r00 * d0 + r01 * d1 + r02 * d2;
        const float wy = r10 * d0 + r11 * d1 + r12 * d2;
        const float wz = r20 * d0 + r21 * d1 + r22 * d2;
        const float nrm = sqrtf(wx * wx + wy * wy + wz * wz);
        rays_d[n * 3 + 0] = wx / nrm; rays_d[n * 3 + 1] = wy / nrm; rays_d[n * 3 + 2] = wz / nrm;
        rays_o[n * 3 + 0] = tx; rays_o[n * 3 + 1] = ty; rays_o[n * 3 + 2] = tz;
    }
}

// ---------------------------------------------------------------------------------------------
// C ABI
// ---------------------------------------------------------------------------------------------
extern "C" {

const char *nsr_status_string(int status) {
    switch (status) {
        case NSR_OK: return "ok";
        case NSR_ERR_INVALID_ARG: return "invalid argument (null pointer, bad size or enum)";
        case NSR_ERR_UNSUPPORTED: return "unsupported configuration for the gfx950 kernels";
        case NSR_ERR_LAUNCH: return "HIP kernel launch failed";
        default: return "unknown status";
    }
}
int nsr_abi_version(void) { return 4; }
const char *nsr_target_arch(void) { return "gfx950"; }

int nsr_near_far_from_aabb(const float *rays_o, const float *rays_d, const float *aabb, uint32_t N, float min_near,
                           float *nears, float *fars, nsr_stream_t stream) {
    if (N == 0) return NSR_OK;
    NSR_CHECK_PTR(rays_o); NSR_CHECK_PTR(rays_d); NSR_CHECK_PTR(aabb); NSR_CHECK_PTR(nears); NSR_CHECK_PTR(fars);
    hipLaunchKernelGGL(k_near_far_from_aabb, dim3(nsr_grid_1d(N, 256)), dim3(256), 0, (hipStream_t)stream, rays_o, rays_d,
                       aabb, N, min_near, nears, fars);
    return nsr_launch_status();
}

int nsr_morton3d(const int32_t *coords, uint32_t N, int32_t *indices, nsr_stream_t stream) {
    if (N == 0) return NSR_OK;
    NSR_CHECK_PTR(coords); NSR_CHECK_PTR(indices);
    hipLaunchKernelGGL(k_morton3d, dim3(nsr_grid_1d(N, 256)), dim3(256), 0, (hipStream_t)stream, coords, N, indices);
    return nsr_launch_status();
}

int nsr_morton3d_invert(const int32_t *indices, uint32_t N, int32_t *coords, nsr_stream_t stream) {
    if (N == 0) return NSR_OK;
    NSR_CHECK_PTR(coords); NSR_CHECK_PTR(indices);
    hipLaunchKernelGGL(k_morton3d_invert, dim3(nsr_grid_1d(N, 256)), dim3(256), 0, (hipStream_t)stream, indices, N, coords);
    return nsr_launch_status();
}

int nsr_packbits(const float *grid, uint32_t N, float density_thresh, uint8_t *bitfield, nsr_stream_t stream) {
    if (N == 0) return NSR_OK;
    NSR_CHECK_PTR(grid); NSR_CHECK_PTR(bitfield);
    if (((uintptr_t)grid & 15u) != 0) return NSR_ERR_INVALID_ARG;   // 16-byte loads
    hipLaunchKernelGGL(k_packbits, dim3(nsr_grid_1d(N, 256)), dim3(256), 0, (hipStream_t)stream, grid, N, density_thresh,
                       bitfield);
    return nsr_launch_status();
}

// step-index capacity of the sample mask: t runs from near to far (<= the AABB diagonal 2 sqrt(3) bound) in steps of at
// least dt_min = 2 sqrt(3) / max_steps, i.e. at most bound * max_steps additions; + slack for the rounding of the sums
static uint32_t march_kcap(float bound, uint32_t max_steps) { return (uint32_t)ceilf(fmaxf(bound, 1.0f) * (float)max_steps) + 96u; }
static bool march_uses_mask(uint32_t N, int is_ndc) { return !is_ndc && N > NSR_MARCH_WPR_MAX_RAYS; }
// wave-per-ray path: records per ray (block start + 64-bit sample mask), 0 = too many for a wave's lanes (re-marching emit)
static uint32_t march_wpr_slot_cap(float bound, uint32_t max_steps) {
    const uint32_t cap = (march_kcap(bound, max_steps) + 63u) / 64u + 1u;
    return cap <= 64u ? cap : 0u;
}

uint64_t nsr_march_rays_train_workspace_bytes(uint32_t N, float bound, uint32_t max_steps) {
    const uint64_t nblocks = (N + RM_BLOCK - 1) / RM_BLOCK;
    uint64_t words = (uint64_t)N + nblocks + 64;
    if (march_uses_mask(N, 0)) words += (uint64_t)N * ((march_kcap(bound, max_steps) + 31u) / 32u + 1u);
    else words += (uint64_t)N * (1u + 3u * march_wpr_slot_cap(bound, max_steps));
    return words * sizeof(uint32_t);
}

int nsr_march_rays_train(const float *rays_o, const float *rays_d, const float *z_hats, const uint8_t *grid, float bound,
                         float dt_gamma, uint32_t max_steps, int is_ndc, uint32_t N, uint32_t C, uint32_t H, uint32_t M,
                         const float *nears, const float *fars, float *xyzs, float *dirs, float *deltas, int32_t *rays,
                         int32_t *counter, const float *noises, void *workspace, nsr_stream_t stream) {
    if (N == 0) return NSR_OK;
    NSR_CHECK_PTR(rays_o); NSR_CHECK_PTR(rays_d); NSR_CHECK_PTR(grid); NSR_CHECK_PTR(nears); NSR_CHECK_PTR(fars);
    NSR_CHECK_PTR(xyzs); NSR_CHECK_PTR(deltas); NSR_CHECK_PTR(rays); NSR_CHECK_PTR(counter); NSR_CHECK_PTR(workspace);
    if (is_ndc && z_hats == nullptr) return NSR_ERR_INVALID_ARG;
    if (max_steps == 0 || C == 0 || C > 8 || H == 0 || H > 1024) return NSR_ERR_INVALID_ARG;
    if (((uintptr_t)deltas & 15u) != 0) return NSR_ERR_INVALID_ARG;
    hipStream_t s = (hipStream_t)stream;
    const uint32_t nblocks = (N + RM_BLOCK - 1) / RM_BLOCK;
    uint32_t *counts = (uint32_t *)workspace;
    uint32_t *block_sums = counts + N;
    // small batches: one wave per ray (k_march_wpr), bit-identical results; NSR_MARCH_WPR=0/1 forces a path
    static const int wpr_env = [] { const char *e = getenv("NSR_MARCH_WPR"); return e ? atoi(e) : -1; }();
    const bool wpr = !is_ndc && (wpr_env >= 0 ? wpr_env != 0 : N <= NSR_MARCH_WPR_MAX_RAYS);
    if (wpr) {
        const uint32_t wblocks = (N + 3) / 4;
        // the counting pass records every block's start and sample mask; the emit replays them (no second probe).  The records
        // need the workspace of nsr_march_rays_train_workspace_bytes for THIS N (a forced wave-per-ray march of a large batch,
        // NSR_MARCH_WPR=1, re-marches instead: its workspace was sized for the thread-per-ray mask)
        static const int slots_env = [] { const char *e = getenv("NSR_MARCH_WPR_SLOTS"); return e ? atoi(e) : 1; }();
        const uint32_t slot_cap = (slots_env && !march_uses_mask(N, 0)) ? march_wpr_slot_cap(bound, max_steps) : 0u;
        uint32_t *slots = slot_cap ? block_sums + nblocks + 64 : nullptr;
        hipLaunchKernelGGL((k_march_wpr<false>), dim3(wblocks), dim3(256), 0, s, rays_o, rays_d, grid, bound, dt_gamma, max_steps, N, C,
                           H, M, nears, fars, noises, counts, (const uint32_t *)nullptr, (float *)nullptr, (float *)nullptr,
                           (float *)nullptr, (int32_t *)nullptr, slots, slot_cap);
        hipLaunchKernelGGL(k_march_block_sums, dim3(nblocks), dim3(RM_BLOCK), 0, s, counts, N, block_sums);
        hipLaunchKernelGGL(k_scan_block_sums, dim3(1), dim3(1024), 0, s, block_sums, nblocks, counter, N);
        if (slots)
            hipLaunchKernelGGL(k_march_wpr_replay, dim3(wblocks), dim3(256), 0, s, rays_o, rays_d, bound, dt_gamma, max_steps, N, C, H, M,
                               counts, block_sums, slots, slot_cap, xyzs, dirs, deltas, rays);
        else
            hipLaunchKernelGGL((k_march_wpr<true>), dim3(wblocks), dim3(256), 0, s, rays_o, rays_d, grid, bound, dt_gamma, max_steps, N,
                               C, H, M, nears, fars, noises, counts, block_sums, xyzs, dirs, deltas, rays, (uint32_t *)nullptr, 0u);
        return nsr_launch_status();
    }
    // large batches, thread per ray: the counting pass marks the samples in a per-ray bit mask, the emitting pass replays the
    // t sequence without probing the grid again (NDC keeps the re-marching emit)
    const bool use_mask = !wpr && march_uses_mask(N, is_ndc);
    uint32_t *mask = use_mask ? block_sums + nblocks + 64 : nullptr;
    const uint32_t kcap = march_kcap(bound, max_steps);
    hipLaunchKernelGGL(k_march_count, dim3(nblocks), dim3(RM_BLOCK), 0, s, rays_o, rays_d, grid, bound, dt_gamma, max_steps, N,
                       C, H, nears, fars, noises, counts, block_sums, mask, kcap);
    // the reference's ray slots start at the incoming counter[1]; only 0 is supported without a
    // host read (renderer.py:213-214 zeroes the counter before every call)
    hipLaunchKernelGGL(k_scan_block_sums, dim3(1), dim3(1024), 0, s, block_sums, nblocks, counter, N);
    // The replaying emit is bound by its stores: every lane appends 12 + 8 bytes at a time to its own ray's two runs, and with
    // the ~30 waves per CU its 32 registers allow, more partially written lines are open than the L2 holds.  60 KB of (unused)
    // dynamic LDS per workgroup keeps two workgroups per CU: 1.12-1.20 -> 0.87-1.09 ms on the bench frame (box to box); the
    // closed-form step of k_march_emit_mask takes another ~15 us at that occupancy, nothing at full occupancy.
    static const int emit_lds = [] { const char *e = getenv("NSR_MARCH_EMIT_LDS"); const int v = e ? atoi(e) : 61440; return v < 0 ? 0 : (v > 64000 ? 64000 : v); }();
    static const int closed_env = [] { const char *e = getenv("NSR_MARCH_EMIT_CLOSED"); return e ? atoi(e) : 1; }();
    if (use_mask) {
        hipLaunchKernelGGL(k_march_emit_mask, dim3(nblocks), dim3(RM_BLOCK), (size_t)emit_lds, s, rays_o, rays_d, bound, dt_gamma, max_steps, N, C,
                           H, M, nears, noises, counts, block_sums, mask, xyzs, dirs, deltas, rays, closed_env);
    }
    else
        hipLaunchKernelGGL(k_march_emit, dim3(nblocks), dim3(RM_BLOCK), 0, s, rays_o, rays_d, z_hats, grid, bound, dt_gamma,
                           max_steps, is_ndc, N, C, H, M, nears, fars, noises, counts, block_sums, 0u, xyzs, dirs, deltas, rays);
    return nsr_launch_status();
}

int nsr_march_rays(uint32_t n_alive, uint32_t n_step, const int32_t *rays_alive, const float *rays_t, const float *rays_o,
                   const float *rays_d, const float *z_hats, float bound, float dt_gamma, uint32_t max_steps, int is_ndc,
                   uint32_t C, uint32_t H, const uint8_t *grid, const float *nears, const float *fars, float *xyzs,
                   float *dirs, float *deltas, const float *noises, nsr_stream_t stream) {
    if (n_alive == 0 || n_step == 0) return NSR_OK;
    NSR_CHECK_PTR(rays_alive); NSR_CHECK_PTR(rays_t); NSR_CHECK_PTR(rays_o); NSR_CHECK_PTR(rays_d); NSR_CHECK_PTR(grid);
    NSR_CHECK_PTR(fars); NSR_CHECK_PTR(xyzs); NSR_CHECK_PTR(deltas);
    (void)nears;
    if (is_ndc && z_hats == nullptr) return NSR_ERR_INVALID_ARG;
    if (max_steps == 0 || C == 0 || C > 8 || H == 0 || H > 1024) return NSR_ERR_INVALID_ARG;
    hipLaunchKernelGGL(k_march_rays, dim3(nsr_div_up(n_alive, RM_BLOCK)), dim3(RM_BLOCK), 0, (hipStream_t)stream, n_alive,
                       n_step, rays_alive, rays_t, rays_o, rays_d, z_hats, bound, dt_gamma, max_steps, is_ndc, C, H, grid, fars,
                       xyzs, dirs, deltas, noises);
    return nsr_launch_status();
}

int nsr_composite_rays(uint32_t n_alive, uint32_t n_step, float T_thresh, int32_t *rays_alive, float *rays_t,
                       const float *sigmas, const float *rgbs, const float *deltas, uint32_t C, int is_ndc,
                       float *weights_sum, float *depth, float *image, nsr_stream_t stream) {
    if (n_alive == 0 || n_step == 0) return NSR_OK;
    NSR_CHECK_PTR(rays_alive); NSR_CHECK_PTR(rays_t); NSR_CHECK_PTR(sigmas); NSR_CHECK_PTR(rgbs); NSR_CHECK_PTR(deltas);
    NSR_CHECK_PTR(weights_sum); NSR_CHECK_PTR(depth); NSR_CHECK_PTR(image);
    if (C == 0 || C > RM_MAXC) return NSR_ERR_UNSUPPORTED;
    hipLaunchKernelGGL(k_composite_rays, dim3(nsr_div_up(n_alive, RM_BLOCK)), dim3(RM_BLOCK), 0, (hipStream_t)stream, n_alive,
                       n_step, T_thresh, rays_alive, rays_t, sigmas, rgbs, deltas, C, is_ndc, weights_sum, depth, image);
    return nsr_launch_status();
}

int nsr_composite_rays_infer(const float *sigmas, const float *rgbs, const float *deltas, const int32_t *rays, const float *nears,
                             uint32_t M, uint32_t N, uint32_t C, float T_thresh, float *weights_sum, float *depth, float *image,
                             nsr_stream_t stream) {
    if (N == 0) return NSR_OK;
    NSR_CHECK_PTR(sigmas); NSR_CHECK_PTR(rgbs); NSR_CHECK_PTR(deltas); NSR_CHECK_PTR(rays); NSR_CHECK_PTR(nears);
    NSR_CHECK_PTR(weights_sum); NSR_CHECK_PTR(depth); NSR_CHECK_PTR(image);
    if (C == 0 || C > RM_MAXC) return NSR_ERR_UNSUPPORTED;
    if (((uintptr_t)deltas & 7u) != 0) return NSR_ERR_INVALID_ARG;
    hipStream_t hs = (hipStream_t)stream;
#define NSR_CI(LPR)                                                                                                 \
    hipLaunchKernelGGL((k_composite_infer<LPR>), dim3(nsr_div_up((uint64_t)N * LPR, RM_BLOCK)), dim3(RM_BLOCK), 0, hs, \
                       sigmas, rgbs, deltas, rays, nears, M, N, C, T_thresh, weights_sum, depth, image)
    if (C <= 4) NSR_CI(4); else if (C <= 8) NSR_CI(8); else NSR_CI(16);
#undef NSR_CI
    return nsr_launch_status();
}

uint64_t nsr_compact_alive_workspace_bytes(uint32_t n_alive) {
    const uint64_t nblocks = (n_alive + RM_BLOCK - 1) / RM_BLOCK;
    return (nblocks + 64) * sizeof(uint32_t);
}

int nsr_compact_alive(const int32_t *rays_alive, uint32_t n_alive, int32_t *out, int32_t *n_out, void *workspace,
                      nsr_stream_t stream) {
    NSR_CHECK_PTR(n_out);
    hipStream_t s = (hipStream_t)stream;
    if (n_alive == 0) {
        return hipMemsetAsync(n_out, 0, sizeof(int32_t), s) == hipSuccess ? NSR_OK : NSR_ERR_LAUNCH;
    }
    NSR_CHECK_PTR(rays_alive); NSR_CHECK_PTR(out); NSR_CHECK_PTR(workspace);
    const uint32_t nblocks = (n_alive + RM_BLOCK - 1) / RM_BLOCK;
    uint32_t *block_sums = (uint32_t *)workspace;
    int32_t *counter2 = (int32_t *)(block_sums + nblocks);   // scratch {total, unused}
    if (hipMemsetAsync(counter2, 0, 2 * sizeof(int32_t), s) != hipSuccess) return NSR_ERR_LAUNCH;
    hipLaunchKernelGGL(k_alive_count, dim3(nblocks), dim3(RM_BLOCK), 0, s, rays_alive, n_alive, block_sums);
    hipLaunchKernelGGL(k_scan_block_sums, dim3(1), dim3(1024), 0, s, block_sums, nblocks, counter2, 0u);
    hipLaunchKernelGGL(k_alive_write, dim3(nblocks), dim3(RM_BLOCK), 0, s, rays_alive, n_alive, block_sums, out);
    hipLaunchKernelGGL(k_store_scan_total, dim3(1), dim3(1), 0, s, counter2, n_out);
    return nsr_launch_status();
}

int nsr_generate_rays(const float *pose, uint32_t w, uint32_t h, float fx, float fy, float cx, float cy, int camera_flip,
                      const int32_t *pix, uint32_t N, float *rays_o, float *rays_d, nsr_stream_t stream) {
    if (N == 0) return NSR_OK;
    NSR_CHECK_PTR(pose); NSR_CHECK_PTR(rays_o); NSR_CHECK_PTR(rays_d);
    if (w == 0 || h == 0) return NSR_ERR_INVALID_ARG;
    if (pix == nullptr && (uint64_t)N != (uint64_t)w * h) return NSR_ERR_INVALID_ARG;
    hipLaunchKernelGGL(k_generate_rays, dim3(nsr_grid_1d(N, 256)), dim3(256), 0, (hipStream_t)stream, pose, w, h, fx, fy, cx, cy,
                       camera_flip, pix, N, rays_o, rays_d);
    return nsr_launch_status();
}

}   // extern "C"
