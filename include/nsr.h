/*
 * nsr.h -- C ABI of libnsr_hip.so, the MI355X (gfx950) volume-rendering hot path.
 *
 * Drop-in boundary for the two pybind11 extension modules of hkust-vgd/nerfstyle
 *   _raymarching  (raymarching/src/bindings.cpp:5-21, signatures raymarching/src/raymarching.h:6-38)
 *   _gridencoder  (gridencoder/src/bindings.cpp:5-8,  signatures gridencoder/src/gridencoder.h:12-14)
 * and for the third native surface the reference reaches through tinycudann
 *   tcnn.Network  (networks/style_nerf.py:44-98)
 * plus fused entry points that have no reference counterpart (the reference launches
 * march / encode x2 / MLP x4 / exp / cat / composite separately).
 *
 * Contract (same as the reference's):
 *   - plain C, raw DEVICE pointers + explicit sizes + a HIP stream; no torch types;
 *   - the caller allocates every output and every workspace; the library never allocates,
 *     frees or retains device memory and keeps no state between calls (re-entrant, thread-safe).
 *     The one exception is a per-device "attribute already set" flag beside the two kernels that ask
 *     for more than the default LDS (hipFuncSetAttribute is idempotent: a racing second call is harmless);
 *   - no host synchronisation, no host reads of device data, no library calls (round 3: the sample sort
 *     is the library's own, rocPRIM is gone): every call is legal inside hipGraph stream capture;
 *   - returns an int status (0 = ok, <0 = error) and never throws; the reference returns void
 *     and raises c10::Error / std::runtime_error (gridencoder.cu:369,387,414,433,440-487) --
 *     the Python wrappers turn a negative status into RuntimeError.
 *   - pointer arguments marked "host" are read on the host at call time (small tables).
 *
 * Element-type codes (nsr_dtype): the reference dispatches on at::ScalarType
 * (AT_DISPATCH_FLOATING_TYPES_AND_HALF); this ABI passes a code.
 */
#ifndef NSR_H_
#define NSR_H_

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef void *nsr_stream_t; /* a hipStream_t */

enum nsr_status {
    NSR_OK = 0,
    NSR_ERR_INVALID_ARG = -1, /* null pointer, bad size, bad enum */
    NSR_ERR_UNSUPPORTED = -2, /* configuration outside what the kernels are built for */
    NSR_ERR_LAUNCH = -3       /* hipGetLastError() after the launch was not hipSuccess */
};

enum nsr_dtype { NSR_F32 = 0, NSR_F16 = 1, NSR_BF16 = 2 };
enum nsr_activation { NSR_ACT_NONE = 0, NSR_ACT_SIGMOID = 1 };

const char *nsr_status_string(int status);
/* ABI version; bumped on any signature change. */
int nsr_abi_version(void);
/* "gfx950" -- the only code object in the library. */
const char *nsr_target_arch(void);

/* ------------------------------------------------------------------------------------------
 * _raymarching replacements
 * ------------------------------------------------------------------------------------------ */

/* replaces near_far_from_aabb (raymarching.h:6, raymarching.cu:190-255).
 * rays_o, rays_d [N,3] f32; aabb [6] f32 (device); nears, fars [N] f32. */
int nsr_near_far_from_aabb(const float *rays_o, const float *rays_d, const float *aabb, uint32_t N,
                           float min_near, float *nears, float *fars, nsr_stream_t stream);

/* replaces morton3D (raymarching.h:8, raymarching.cu:313-331). coords [N,3] i32 -> indices [N] i32 */
int nsr_morton3d(const int32_t *coords, uint32_t N, int32_t *indices, nsr_stream_t stream);

/* replaces morton3D_invert (raymarching.h:9, raymarching.cu:336-359). */
int nsr_morton3d_invert(const int32_t *indices, uint32_t N, int32_t *coords, nsr_stream_t stream);

/* replaces packbits (raymarching.h:10, raymarching.cu:366-399).
 * grid f32 [N*8]; bitfield u8 [N]; bit i of byte n = grid[8n+i] > density_thresh. */
int nsr_packbits(const float *grid, uint32_t N, float density_thresh, uint8_t *bitfield,
                 nsr_stream_t stream);

/* replaces march_rays_train (raymarching.h:12, raymarching.cu:410-599).
 *
 * Same outputs as the reference kernel, with one strengthening: sample offsets come from an
 * exclusive scan over ray index instead of atomicAdd arrival order (raymarching.cu:506-507),
 * so `rays` is deterministic: row n = (n, offset_n, count_n).  That is one of the orders the
 * reference can produce.  counter[0] += total samples, counter[1] += N (the reference's
 * atomics), so a non-zero incoming counter offsets the allocation exactly as there.
 *
 * xyzs [M,3], deltas [M,4] f32 (slots 2,3 only written when is_ndc), dirs [M,3] or NULL (the
 * model on this path is built with use_dir=False and never reads it), rays [N,3] i32,
 * counter [2] i32, noises [N] f32 or NULL (= zeros: perturb is force-disabled,
 * raymarching.py:247), z_hats [N] or NULL when !is_ndc.
 * Rays with offset+count >= M are dropped like the reference does (raymarching.cu:517).
 * workspace: nsr_march_rays_train_workspace_bytes(N, bound, max_steps) bytes of device scratch (per-ray counts, block sums
 * and, for batches marched one thread per ray, a bit mask of bound * max_steps step indices per ray: the counting pass marks
 * the samples, the emitting pass replays the t sequence without probing the grid again; batches marched one wave per ray
 * record the start and the 64-bit sample mask of every 64-step block instead, and their emit replays those). */
uint64_t nsr_march_rays_train_workspace_bytes(uint32_t N, float bound, uint32_t max_steps);
int nsr_march_rays_train(const float *rays_o, const float *rays_d, const float *z_hats,
                         const uint8_t *grid, float bound, float dt_gamma, uint32_t max_steps,
                         int is_ndc, uint32_t N, uint32_t C, uint32_t H, uint32_t M,
                         const float *nears, const float *fars, float *xyzs, float *dirs,
                         float *deltas, int32_t *rays, int32_t *counter, const float *noises,
                         void *workspace, nsr_stream_t stream);

/* replaces composite_rays_train_forward (raymarching.h:14, raymarching.cu:806-890).
 * sigmas [M], rgbs [M,C], deltas [M,4], rays [N,3] -> weights_sum [N], depth [N], image [N,C].
 * M is the sample-buffer size the march used for its "offset+count >= M" drop test
 * (raymarching.cu:517,830): pass the same value.  C <= 16. */
int nsr_composite_rays_train_forward(const float *sigmas, const float *rgbs, const float *deltas,
                                     const int32_t *rays, uint32_t M, uint32_t N,
                                     uint32_t C, float T_thresh, int is_ndc, float *weights_sum,
                                     float *depth, float *image, nsr_stream_t stream);

/* replaces composite_rays_train_backward (raymarching.h:15, raymarching.cu:904-997).
 * The reference needs grad_sigmas, grad_rgbs pre-zeroed and an [N,C] scratch rgbs_buf
 * (raymarching.py:339-341); here the running colour sum lives in registers, rgbs_buf is gone,
 * and every sample that belongs to a ray (early-stopped tails and dropped rays included) gets its
 * gradient written -- zeros where the reference leaves the pre-filled zero -- so the buffers need
 * NOT arrive zeroed; only padding samples that belong to no ray are left untouched. */
int nsr_composite_rays_train_backward(const float *grad_weights_sum, const float *grad_image,
                                      const float *sigmas, const float *rgbs, const float *deltas,
                                      const int32_t *rays, int is_ndc, const float *weights_sum,
                                      const float *image, uint32_t M, uint32_t N,
                                      uint32_t C, float T_thresh, float *grad_sigmas, float *grad_rgbs,
                                      nsr_stream_t stream);

/* Training composite + the epilogue of Renderer.render_train (renderer.py:225-233) in one launch (no reference
 * counterpart: the reference runs composite_rays_train, then torch ops for `image + (1 - weights_sum)`, the class slice
 * and `clamp(depth - nears, 0) / (fars - nears)`).  Same arithmetic as nsr_composite_rays_train_forward; additionally
 * writes rgb_map [N,3] = image[:, :3] + (1 - weights_sum), classes [N, C-3] = image[:, 3:] (NULL allowed when C == 3) and
 * depth_norm [N].  weights_sum, depth (raw) and image [N,C] are still written (the backward reads weights_sum / image). */
int nsr_render_train_forward(const float *sigmas, const float *rgbs, const float *deltas, const int32_t *rays,
                             const float *nears, const float *fars, uint32_t M, uint32_t N, uint32_t C, float T_thresh,
                             float *weights_sum, float *depth, float *image, float *rgb_map, float *depth_norm,
                             float *classes, nsr_stream_t stream);
/* Backward of nsr_render_train_forward with respect to sigmas / rgbs, from the gradients of rgb_map [N,3], classes
 * [N, C-3] and weights_sum [N] (each may be NULL = zero; depth_norm carries no gradient, as in the reference whose
 * composite backward ignores grad_depth, raymarching.py:333).  Every sample of every ray gets its gradient written. */
int nsr_render_train_backward(const float *grad_rgb_map, const float *grad_classes, const float *grad_weights_sum,
                              const float *sigmas, const float *rgbs, const float *deltas, const int32_t *rays,
                              const float *weights_sum, const float *image, uint32_t M, uint32_t N, uint32_t C,
                              float T_thresh, float *grad_sigmas, float *grad_rgbs, nsr_stream_t stream);

/* Reconstruction loss of Trainer.calc_loss (trainers/base.py:251-304), value AND gradient in one pass:
 *   loss = mean((rgb_map - target_rgb[pix])^2) + ce_lambda * mean(logsumexp(classes) - classes[target_cls[pix]])
 * (nn.MSELoss + 0.001 * nn.CrossEntropyLoss on the class logits; the reference builds it from ~25 torch kernels and as
 * many again in autograd).  pix [N] int64 = rows of the targets (NULL: row n); target_rgb [P,3] f32; target_cls [P] int64.
 * classes == NULL, nc == 0 or ce_lambda == 0: MSE only (grad_classes untouched).
 * Everything is multiplied by factor * (scale ? scale[0] : 1): `scale` is the device-side loss scale (nsr_scaler_update),
 * `factor` e.g. 1 / world size.  loss_out [3] = {scaled total, unscaled mse, unscaled ce_lambda * ce}.  The value is summed in
 * a fixed order (block tree + one-block final pass): run-to-run identical.
 * workspace: nsr_recon_loss_workspace_bytes(N) bytes. */
uint64_t nsr_recon_loss_workspace_bytes(uint32_t N);
int nsr_recon_loss(const float *rgb_map, const float *classes, uint32_t N, uint32_t nc, const float *target_rgb,
                   const int64_t *target_cls, const int64_t *pix, float ce_lambda, float factor, const float *scale,
                   float *grad_rgb_map, float *grad_classes, float *loss_out, void *workspace, nsr_stream_t stream);

/* replaces march_rays (raymarching.h:17, raymarching.cu:1004-1130), inference. */
int nsr_march_rays(uint32_t n_alive, uint32_t n_step, const int32_t *rays_alive, const float *rays_t,
                   const float *rays_o, const float *rays_d, const float *z_hats, float bound,
                   float dt_gamma, uint32_t max_steps, int is_ndc, uint32_t C, uint32_t H,
                   const uint8_t *grid, const float *nears, const float *fars, float *xyzs, float *dirs,
                   float *deltas, const float *noises, nsr_stream_t stream);

/* replaces composite_rays (raymarching.h:18, raymarching.cu:1133-1240), inference, in place. */
int nsr_composite_rays(uint32_t n_alive, uint32_t n_step, float T_thresh, int32_t *rays_alive,
                       float *rays_t, const float *sigmas, const float *rgbs, const float *deltas,
                       uint32_t C, int is_ndc, float *weights_sum, float *depth, float *image,
                       nsr_stream_t stream);

/* Single-pass inference composite (no reference counterpart as one call): kernel_composite_rays'
 * arithmetic (raymarching.cu:1133-1231: T = 1 - weight_sum, stop test before the sample, absolute t
 * from nears[ray]) over the compacted samples [offset, offset+count) that nsr_march_rays_train
 * emits -- replaces the host loop of up to max_steps march_rays / composite_rays iterations
 * (renderer.py:266-285).  Outputs are written (not accumulated). */
int nsr_composite_rays_infer(const float *sigmas, const float *rgbs, const float *deltas, const int32_t *rays,
                             const float *nears, uint32_t M, uint32_t N, uint32_t C, float T_thresh,
                             float *weights_sum, float *depth, float *image, nsr_stream_t stream);

/* Alive-ray compaction on the device (replaces the boolean-mask `rays_alive[rays_alive >= 0]`
 * of renderer.py:284, which is a host sync): stable, ballot/scan based.
 * out [n_alive] i32, n_out [1] i32 (device).  workspace: nsr_compact_alive_workspace_bytes(n). */
uint64_t nsr_compact_alive_workspace_bytes(uint32_t n_alive);
int nsr_compact_alive(const int32_t *rays_alive, uint32_t n_alive, int32_t *out, int32_t *n_out,
                      void *workspace, nsr_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * _gridencoder replacements
 * ------------------------------------------------------------------------------------------ */

/* gridencoder.cu:137 evaluated once on the host: resolution[l] = floor(exp2f(l*S)*H), fp32.
 * The kernels take this table instead of re-evaluating exp2f per thread. res_out host [L]. */
int nsr_grid_resolutions(uint32_t L, float S, uint32_t H, uint32_t *res_out);

/* replaces grid_encode_forward (gridencoder.h:12, gridencoder.cu:83-235,439-461), D = 3.
 * inputs [B,3] f32 in [0,1]; embeddings [rows,C] of emb_dtype (NSR_F32|NSR_F16);
 * offsets HOST [L+1] i32 (the reference passes a device tensor; it is 17 ints);
 * outputs of emb_dtype: layout [L,B,C] when out_blc == 0 (the reference kernel's) or
 * [B,L*C] when out_blc != 0 (what grid.py:58 returns after its permute+reshape copy).
 * calc_grad_inputs is not supported (never true on this path) -> NSR_ERR_UNSUPPORTED.
 * C in {1,2,4,8} like the reference; L <= 32. */
int nsr_grid_encode_forward(const float *inputs, const void *embeddings, int emb_dtype,
                            const int32_t *offsets, void *outputs, uint32_t B, uint32_t D, uint32_t C,
                            uint32_t L, float S, uint32_t H, int calc_grad_inputs, uint32_t gridtype,
                            int align_corners, uint32_t style, int out_blc, nsr_stream_t stream);

/* replaces grid_encode_backward (gridencoder.h:13, gridencoder.cu:238-328,464-494).
 * grad of grad_dtype in the layout selected by grad_blc; grad_embeddings [rows,C] f32 ALWAYS
 * (fp32 atomics; the reference accumulates in half under AMP) and arrives zeroed or holding a
 * running sum (the kernel accumulates). */
int nsr_grid_encode_backward(const void *grad, int grad_dtype, const float *inputs,
                             const int32_t *offsets, float *grad_embeddings, uint32_t B, uint32_t D,
                             uint32_t C, uint32_t L, float S, uint32_t H, uint32_t gridtype,
                             int align_corners, uint32_t style, int grad_blc, nsr_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * tcnn.Network replacement: bias-free fully fused MLP on MFMA
 * ------------------------------------------------------------------------------------------
 * params: fp32 master weights, flat, row-major [out,in] per layer, layers concatenated, the
 * last layer's rows padded to a multiple of 16 (nsr_mlp_param_count).  Hidden width 64, ReLU.
 * Compute contract: inputs, weights and hidden activations rounded to compute_dtype
 * (NSR_F16 | NSR_BF16), products accumulated in fp32 by v_mfma_f32_16x16x32_{f16,bf16}.
 * x [M,n_in] f32; y [M,n_out] f32 (after out_act).  n_in in {16,32,64}; n_out <= 16;
 * n_hidden_layers in {1,2}; n_neurons == 64. */
uint32_t nsr_mlp_param_count(uint32_t n_in, uint32_t n_out, uint32_t n_neurons, uint32_t n_hidden_layers);
int nsr_mlp_forward(const float *x, const float *params, uint32_t M, uint32_t n_in, uint32_t n_out,
                    uint32_t n_neurons, uint32_t n_hidden_layers, int out_act, int compute_dtype,
                    float *y, nsr_stream_t stream);
/* dy [M,n_out] f32 = dL/dy (post-activation); y [M,n_out] the forward output (for sigmoid').
 * dx [M,n_in] f32 or NULL; dparams [count] f32 is ACCUMULATED into (fp32 atomics, one add per
 * workgroup per weight).  Activations are recomputed, nothing is saved by the forward. */
int nsr_mlp_backward(const float *x, const float *params, const float *y, const float *dy, uint32_t M,
                     uint32_t n_in, uint32_t n_out, uint32_t n_neurons, uint32_t n_hidden_layers,
                     int out_act, int compute_dtype, float *dx, float *dparams, nsr_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * Fused field: BBox.normalize -> 2x hash encode -> 4 MLPs -> trunc_exp / sigmoid / cat
 * (networks/style_nerf.py:120-142, use_dir = False) in ONE launch per direction.
 * ------------------------------------------------------------------------------------------
 * Data layout (MI355X-first): the two hash tables are stored INTERLEAVED,
 *   tables[row][enc][feat], enc 0 = x_density_embedder, enc 1 = x_color_embedder, feat < 2,
 * so that one 16-byte (f32) / 8-byte (f16) load or one 64-byte atomic request serves both
 * encoders (same positions, same hash => same row).  grad_tables has the same layout, f32.
 * mlp_params: the four flat fp32 parameter vectors concatenated in the order
 *   density (32->64->1), color1 (32->64->16), color2 (16->64->64->3), class (32->64->nc);
 * grad_mlp likewise (accumulated into).
 */
typedef struct nsr_field_desc {
    uint32_t L;                /* levels (16) */
    uint32_t H;                /* base resolution (16) */
    float S;                   /* log2(per_level_scale), fp32 (grid.py:36) */
    uint32_t num_classes;      /* nc <= 13; output channels C_ch = 3 + nc */
    int table_dtype;           /* NSR_F32 | NSR_F16: element type of `tables` */
    int compute_dtype;         /* NSR_F16 | NSR_BF16 for the MFMA chain */
    float bbox_min[3];         /* BBox.normalize: x_hat = (x - min) / size (common.py:276-288) */
    float bbox_size[3];
    float density_scale;       /* renderer.py:225, folded into sigma */
    const int32_t *offsets;    /* HOST [L+1], rows (grid.py:129-140) */
} nsr_field_desc;

/* xyzs [M,3] f32 world positions.  sigmas [M] f32 = exp(logit) * density_scale;
 * rgbs [M, 3+nc] f32 = cat(sigmoid(rgb), classes) or NULL for the sigma-only branch
 * (style_nerf.py:125-126).  m_dev: optional device int32 sample count (<= M) -- tiles past it
 * are skipped, so callers can size M as a capacity and never read the count on the host.
 * feats (optional, may be NULL): ceil(M/16)*2048 bytes that receive the encoded features of both
 * encoders as ready-made MFMA B fragments (128 B per sample, the only thing a training forward
 * saves); pass the same buffer to nsr_field_backward to skip its re-gather. */
/* perm (optional, may be NULL): device uint32 [M] from nsr_sample_order -- the kernel's tile t then works on samples
 * perm[16t .. 16t+15] of the buffers (outputs land at those samples' own slots; `feats` stays tile-major, so the
 * backward must be given the same perm).  Results per sample are bit-identical with and without it. */
int nsr_field_forward(const nsr_field_desc *desc, const void *tables, const float *mlp_params,
                      const float *xyzs, uint32_t M, const int32_t *m_dev, float *sigmas, float *rgbs,
                      void *feats, const uint32_t *perm, nsr_stream_t stream);

/* grad_sigmas [M], grad_rgbs [M,3+nc] f32.  Recomputes the forward (nothing saved), then
 * back-propagates: trunc_exp' = exp(clamp(logit,-15,15)) (tcnn_nerf.py:62-66), sigmoid', ReLU
 * masks, MLP dgrad on MFMA, wgrad on MFMA (accumulated into grad_mlp), and scatters the
 * encoder gradient into grad_tables with fp32 atomics (both encoders per request).
 * train_density_table / train_color_table: 0 skips that encoder's scatter (stylisation
 * optimises x_color_embedder only, trainers/style.py:25).  grad_mlp may be NULL: no weight gradient is produced; with perm,
 * saved feats, train_density_table = 0 and grad_mlp = NULL (exactly the stylisation stage) a colour-only kernel runs that
 * computes the class / colour nets' input gradients and nothing else. */
int nsr_field_backward(const nsr_field_desc *desc, const void *tables, const float *mlp_params,
                       const float *xyzs, uint32_t M, const int32_t *m_dev, const float *grad_sigmas,
                       const float *grad_rgbs, float *grad_tables, float *grad_mlp,
                       int train_density_table, int train_color_table, const void *feats,
                       const uint32_t *perm, void *workspace, nsr_stream_t stream);
/* feats given together with perm must come from nsr_field_forward called with the SAME perm (they are tile-major in
 * perm's order; the first kernel below walks that order).
 * perm != NULL (nsr_sample_order): the MLP backward writes every sample's encoder-output gradient (256 B) to
 * `workspace` and a second, high-occupancy kernel accumulates the table gradient walking the samples in perm's
 * spatial order (LDS lattices, one merged atomic record per touched corner; csrc/table_scatter.hip).  Pays on dense
 * (full-frame) batches; results equal the perm == NULL call up to fp32 summation order.
 * workspace: nsr_field_backward_workspace_bytes(M, with_perm) bytes, 16-byte aligned (0 / NULL without perm). */
uint64_t nsr_field_backward_workspace_bytes(uint32_t M, int with_perm);

/* Spatial processing order of marched samples (no reference counterpart; see csrc/sample_order.hip): perm [M] u32 =
 * indices of the first min(m_dev[0], M) samples in Morton order of their encoder input (10 bits per axis, stable: ray
 * order inside a 4^3-finest-cell block), followed by the remaining slots in identity order.  sort_prefix <= M caps the
 * number of leading slots that take part in the sort; the kernels bound themselves by min(m_dev[0], sort_prefix) on the
 * device, so M is the normal value (a smaller one is still correct: the slots past it keep identity order).  The sort is
 * the library's own LSD radix sort (3 x 9 bits: the encoder input of a position inside the box is in [0.5, 1], so the top bit
 * of every coordinate is constant; positions outside the box sort with its faces): stream-ordered, no host call, capture-safe,
 * deterministic.
 * Consumed by nsr_field_forward / nsr_field_backward (spatial walk, table scatter in spatial order).
 * workspace: nsr_sample_order_workspace_bytes(M) bytes, 256-byte aligned.  bbox_min / bbox_size: HOST float[3]. */
uint64_t nsr_sample_order_workspace_bytes(uint32_t M);
int nsr_sample_order(const float *xyzs, uint32_t M, const int32_t *m_dev, uint32_t sort_prefix,
                     const float *bbox_min, const float *bbox_size, uint32_t *perm, void *workspace,
                     nsr_stream_t stream);

/* fp32 master tables -> f16 gather copy (the reference's `embeddings.to(torch.half)` under
 * autocast, grid.py:42-43).  n = number of scalars. */
int nsr_cast_f32_to_f16(const float *src, void *dst, uint64_t n, nsr_stream_t stream);

/* Fused Adam (+ optional EMA shadow) over a flat fp32 arena, one pass, grads zeroed on the way
 * out (replaces torch.optim.Adam + zero_grad + torch_ema, trainers/base.py:216-229,420-426).
 * grad_scale_inv multiplies the gradient first (GradScaler unscale). step >= 1.
 * half_copy (f16, may be NULL) receives the updated parameters rounded to half.
 * elem_mask4: bit (i & 3) set <=> element i is trained; others keep parameter and moments (their
 * gradient is still zeroed).  0xF = everything; on the interleaved tables 0x3 = density table only,
 * 0xC = colour table only (stylisation, trainers/style.py:25). */
int nsr_adam_step(float *params, float *grads, float *exp_avg, float *exp_avg_sq, float *ema,
                  void *half_copy, uint64_t n, float lr, float beta1, float beta2, float eps,
                  float grad_scale_inv, float ema_decay, uint32_t step, uint32_t elem_mask4,
                  nsr_stream_t stream);

/* Device-side GradScaler + optimiser bookkeeping (replaces scaler.step(optim) / scaler.update() / scheduler.step(),
 * trainers/base.py:228,420-425 and trainers/style.py:200-204, which read the inf flag back to the host every step).
 * scaler_state: device buffer of 16 32-bit words, kept by the caller across steps:
 *   [0] f32 loss scale (initialise to 65536, GradScaler's default)   [1] i32 growth tracker   [2] u32 found_inf
 *   [3] u32 optimiser steps taken   [4] u32 steps skipped   [5] u32 EMA updates made   [8..13] this step's values for
 *   nsr_adam_step_scaled (skip flag, lr / (1 - beta1^t), 1 / sqrt(1 - beta2^t), 1 / scale, lr, EMA decay); other words
 *   reserved, initialise to 0.
 * One optimiser step =
 *   nsr_grad_check        per trained region: sets found_inf if any element selected by elem_mask4 is inf / nan
 *                         (GradScaler.unscale_'s check; untrained regions are not looked at, so data-parallel ranks
 *                         that all-reduce only the trained regions take the same decision);
 *   nsr_scaler_update     one thread: skip decision, scale *= backoff and tracker = 0 on inf, else step count += 1,
 *                         tracker += 1, scale *= growth every growth_interval clean steps; lr = lr_base *
 *                         0.1^((t-1) / lr_decay_steps) (LambdaLR advanced only on steps that were not skipped;
 *                         lr_decay_steps <= 0: constant) and Adam's bias corrections for step t, in double precision;
 *                         enabled == 0: never skips, scale stays 1 (GradScaler(enabled=False)); clears found_inf;
 *                         ema_decay_max >= 0: torch_ema's schedule, num_updates += 1 and decay = min(ema_decay_max,
 *                         (1 + n) / (10 + n)) on every call (skipped steps included); < 0: no EMA;
 *   nsr_adam_step_scaled  per trained region: nsr_adam_step with those device-side scalars; on a skipped step the
 *                         parameters and moments stay, the gradient is still zeroed and the EMA still moves
 *                         (ema.update() is unconditional, base.py:426).  half_copy covers the first half_n elements.
 * Nothing is read on the host: legal under stream capture. */
int nsr_grad_check(const float *grads, uint64_t n, uint32_t elem_mask4, void *scaler_state, nsr_stream_t stream);
int nsr_scaler_update(void *scaler_state, float lr_base, float lr_decay_steps, float beta1, float beta2,
                      float growth_factor, float backoff_factor, uint32_t growth_interval, int enabled,
                      float ema_decay_max, nsr_stream_t stream);
int nsr_adam_step_scaled(float *params, float *grads, float *exp_avg, float *exp_avg_sq, float *ema,
                         void *half_copy, uint64_t n, uint64_t half_n, float beta1, float beta2, float eps,
                         uint32_t elem_mask4, const void *scaler_state, nsr_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * Occupancy-grid update on the device: replaces Renderer.update_state / _compute_occ_sigmas
 * (renderer.py:120-194), which is torch glue with two host reads (`mean_density.item()`, the size of
 * `nonzero(density_grid > 0)`).  One update = nsr_occ_sample_points -> nsr_field_forward (sigma only,
 * rgbs = NULL, on the returned positions) -> nsr_occ_update.  Nothing is read on the host; every call is
 * capture-safe.  Cells are flat indices cas * H^3 + morton(x, y, z): the layout of density_grid [C, H^3]
 * (renderer.py:62-63).  H^3 must be a multiple of 256, C <= 8.
 * ------------------------------------------------------------------------------------------ */
/* device scratch for both calls (the same buffer must be passed to both calls of one update) */
uint64_t nsr_occ_workspace_bytes(uint32_t C, uint32_t H);
/* points of one update: full_update (renderer.py:143-160) C*H^3, one per cell in flat-index order;
 * else (renderer.py:163-181) per cascade H^3/4 uniform cells followed by H^3/4 draws from the occupied ones */
uint32_t nsr_occ_num_points(uint32_t C, uint32_t H, int full_update);
/* xyzs [P,3] f32 = cell centre position 2*c/(H-1)-1 scaled by (b - b/H) plus (u*2-1)*b/H, b = min(2^cas, bound)
 * (renderer.py:130-133); indices [P] i32 = flat cell index (-1: no point -- the occupied half of a cascade whose
 * grid is empty, where the reference's randint(0, 0) raises).  u in [0,1): Philox4x32-10 keyed by `seed`, counter =
 * (point, sequence [+ sequence_dev[0]]) -- the same numbers on every rank; `noise` [P,3] (optional, device)
 * overrides u (tests pin the jitter with it). */
int nsr_occ_sample_points(const float *density_grid, uint32_t C, uint32_t H, float bound, int full_update,
                          uint64_t seed, uint32_t sequence, const uint32_t *sequence_dev, const float *noise,
                          float *xyzs, int32_t *indices, void *workspace, nsr_stream_t stream);
/* sigmas [P] f32 = field densities (already x density_scale) at those points.  Applies renderer.py:183-189:
 * tmp_grid = -1 except tmp_grid[indices] = sigmas; where grid >= 0 and tmp >= 0: grid = max(grid * decay, tmp);
 * mean_density[0] (device f32) = mean(clamp(grid, 0)); bitfield = packbits(grid, min(mean, density_thresh)).
 * sequence_dev (optional) is incremented, so a captured graph draws fresh jitter on every replay. */
int nsr_occ_update(float *density_grid, const float *sigmas, const int32_t *indices, uint32_t P, uint32_t C,
                   uint32_t H, int full_update, float density_decay, float density_thresh, uint8_t *bitfield,
                   float *mean_density, uint32_t *sequence_dev, void *workspace, nsr_stream_t stream);

/* Ray generation on the device (nerf_lib.py:69-142 + common.py:139-147): pixel centres
 * (x + 0.5), camera-frame direction ((i-cx)/fx, (j-cy)/fy, 1) * flip, R * d, normalise.
 * pose [4,4] f32 row-major (device).  pix [N] i32 1-D pixel ids (row-major, y * w + x) or NULL
 * for all w*h pixels in order (then N == w*h).  flip bits as camera_flip (nerf_lib.py:121). */
int nsr_generate_rays(const float *pose, uint32_t w, uint32_t h, float fx, float fy, float cx,
                      float cy, int camera_flip, const int32_t *pix, uint32_t N, float *rays_o,
                      float *rays_d, nsr_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* NSR_H_ */
