"""Host orchestration of the hot path: the counterpart of the reference's renderer.py
(`Renderer`: update_state :139-194, render_train :196-235, render_test :237-293, render :295-313,
state_dict / load_state_dict :78-107).  Same public methods, same state keys, same outputs.

What is different underneath (MI355X-first):
  * render_train never reads the sample count on the host: the march emits into a
    capacity-bounded buffer, the count stays on the device and the fused field kernels bound
    themselves by it (the reference does `step_counter[0].item()` + `empty_cache()` per call and
    memsets 40 B x N x max_steps first, raymarching.py:238-283);
  * the model is two fused launches (forward / backward), not 2 encoders + 4 MLPs + exp + cat;
  * the white-background / depth epilogue stays in torch (a few [N]-sized ops).
"""
from math import ceil, log2
from typing import Dict, Optional

import torch

from . import raymarching
from .common import Box2D, Intrinsics, RayBatch
from .config import RendererConfig
from .rays import generate_rays
from .style_nerf import StyleTCNerf

STEP_CTR_SIZE = 16


class Renderer(torch.nn.Module):
    def __init__(self, model: StyleTCNerf, cfg: RendererConfig, intr: Intrinsics, bound: float, name: str = 'Renderer',
                 precrop_frac: float = 1., raymarch_channels: int = 3, samples_per_ray_cap: Optional[int] = None):
        """samples_per_ray_cap: capacity of the sample buffers in samples per ray (None = max_steps,
        i.e. the reference's force_all_rays allocation N * max_steps; smaller values trade memory for
        the reference's mean_count-style "drop rays that do not fit" behaviour, raymarching.py:230-236)."""
        super().__init__()
        self.model = model
        self.cfg = cfg
        self.intr = intr
        self._use_precrop = False
        self.precrop_frac = precrop_frac
        self.raymarch_channels = raymarch_channels
        self.update_occ = True
        self.bound = bound
        self.samples_per_ray_cap = samples_per_ray_cap
        # Spatially ordered table scatter in the backward (nsr_sample_order + the two-kernel nsr_field_backward).  'auto': for
        # dense batches only -- on a full 1008x756 frame the backward drops from 49 to 29 ms (+ 1.8 ms sort); on sparse random
        # batches the blocks hold too few samples (65 536 random rays: 6.2 vs 4.5 ms).  True / False force it.
        self.sort_samples = 'auto'
        self.sort_min_rays = 90000       # any batch of at least this many rays ...
        self.sort_min_dense_rays = 16384  # ... or a DENSE pixel set (full frame / patch / crop: neighbouring pixels) of this many
        self._pinned_bitfield = None
        self.aabb = torch.tensor([-bound, -bound, -bound, bound, bound, bound], dtype=torch.float32)
        self.cascade = 1 + ceil(log2(bound))
        grid_size = self.cfg.grid_size
        bitfield_size = self.cascade * (grid_size ** 3) // 8
        self.density_grid = torch.zeros((self.cascade, grid_size ** 3))
        self.density_bitfield = torch.zeros((bitfield_size, ), dtype=torch.uint8)
        self.step_counter = torch.zeros((STEP_CTR_SIZE, 2), dtype=torch.int32)
        self.local_step = 0
        self._mean_density_dev, self._mean_density_host = None, 0.0
        self._mean_count_host, self._mean_count_stale = 0, False
        self._occ_state = None
        self.occ_seed = 0
        self.device = torch.device('cpu')

    # TensorModule behaviour of the reference (common.py:207-240): plain tensor attributes move too
    def _apply(self, fn, *args, **kwargs):
        for k in ('aabb', 'density_grid', 'density_bitfield', 'step_counter'):
            setattr(self, k, fn(getattr(self, k)))
        super()._apply(fn, *args, **kwargs)
        self.device = self.aabb.device
        if self._mean_density_dev is not None and self._mean_density_dev.device != self.device:
            self.mean_density = float(self._mean_density_dev.item())
        return self

    def state_dict(self):
        """renderer.py:78-91"""
        sd = {'model': self.model.state_dict()}
        for k in ['intr', 'precrop_frac', 'raymarch_channels', 'bound', 'density_grid', 'density_bitfield',
                  'step_counter', 'local_step', 'mean_count', 'mean_density']:
            v = getattr(self, k)
            if torch.is_tensor(v):
                v = v.detach()
            sd[k] = v
        return sd

    def load_state_dict(self, state_dict):
        """renderer.py:93-107"""
        for k in ['intr', 'precrop_frac', 'raymarch_channels', 'bound']:
            if getattr(self, k) != state_dict[k]:
                raise RuntimeError('Values do not match when loading key "{}"'.format(k))
        self.model.load_state_dict(state_dict['model'])
        for k in ['density_grid', 'density_bitfield', 'step_counter', 'local_step', 'mean_count', 'mean_density']:
            v = state_dict[k]
            if torch.is_tensor(v):
                v = v.to(self.device)
            setattr(self, k, v)

    @property
    def use_precrop(self):
        return self._use_precrop

    @use_precrop.setter
    def use_precrop(self, value: bool):
        self._use_precrop = value

    # ---- occupancy grid ------------------------------------------------------------------------
    @property
    def mean_density(self) -> float:
        """renderer.py:186 keeps a host float; here the value lives on the device and is read only when asked for."""
        if self._mean_density_dev is not None:
            return float(self._mean_density_dev.item())
        return self._mean_density_host

    @mean_density.setter
    def mean_density(self, v):
        self._mean_density_host = float(v)
        self._mean_density_dev = None

    @property
    def mean_count(self) -> int:
        """renderer.py:192-194 (mean emitted samples of the last min(16, update_iter) steps).  The reference marches
        with force_all_rays=True (:219-221), so the value is checkpoint state only: computed when read."""
        if self._mean_count_stale and self.step_counter.is_cuda:
            total_step = min(STEP_CTR_SIZE, self.cfg.update_iter)
            self._mean_count_host = int(self.step_counter[:total_step, 0].sum().item() / total_step)
            self._mean_count_stale = False
        return self._mean_count_host

    @mean_count.setter
    def mean_count(self, v):
        self._mean_count_host = int(v)
        self._mean_count_stale = False

    def manual_seed(self, seed: int):
        """Seed of the occupancy jitter / cell draws (counter-based: every rank that uses the same seed draws
        the same points, so data-parallel replicas keep identical grids without a collective)."""
        self.occ_seed = int(seed)
        return self

    def _occ_buffers(self, P):
        """Scratch of one update; allocated once per shape (no allocation in the steady state)."""
        key = (P, str(self.device))
        if getattr(self, '_occ_key', None) != key:
            from . import _lib as L
            dev = self.device
            nbytes = int(L.lib().nsr_occ_workspace_bytes(self.cascade, self.cfg.grid_size))
            if nbytes == 0:
                raise RuntimeError('occupancy grid of size {} x {}^3 is not supported (H^3 must be a multiple of 256)'.format(
                    self.cascade, self.cfg.grid_size))
            self._occ_ws = torch.empty((nbytes + 3) // 4, dtype=torch.int32, device=dev)
            self._occ_xyzs = torch.empty(P, 3, dtype=torch.float32, device=dev)
            self._occ_idx = torch.empty(P, dtype=torch.int32, device=dev)
            self._occ_key = key
        return self._occ_ws, self._occ_xyzs, self._occ_idx

    @torch.no_grad()
    def update_state(self, noise: Optional[torch.Tensor] = None) -> None:
        """renderer.py:139-194 on the device: cell choice + jitter (nsr_occ_sample_points), the sigma-only fused
        field on those points, then scatter / max-decay / mean / packbits (nsr_occ_update).  No host read, no
        allocation after the first call, legal under hipGraph capture.  `noise` [P,3] in [0,1) pins the jitter."""
        from . import _lib as L
        C, H = self.cascade, self.cfg.grid_size
        full = 1 if self.local_step < self.cfg.update_thres else 0
        P = int(L.lib().nsr_occ_num_points(C, H, full))
        ws, xyzs, idx = self._occ_buffers(P)
        if self._occ_state is None or self._occ_state.device != self.device:
            self._occ_state = torch.zeros(4, dtype=torch.int32, device=self.device)    # [0]: update sequence
        if self._mean_density_dev is None:
            self._mean_density_dev = torch.full((1,), self._mean_density_host, dtype=torch.float32, device=self.device)
        if not self.density_grid.is_contiguous():
            self.density_grid = self.density_grid.contiguous()
        L.check(L.lib().nsr_occ_sample_points(L.p(self.density_grid), C, H, float(self.bound), full, self.occ_seed, 0,
                                              L.p(self._occ_state), L.p(noise), L.p(xyzs), L.p(idx), L.p(ws), L.stream()),
                'occ_sample_points')
        sigmas = self.model.field(xyzs, sigma_only=True, density_scale=self.cfg.density_scale)
        L.check(L.lib().nsr_occ_update(L.p(self.density_grid), L.p(sigmas), L.p(idx), P, C, H, full,
                                       float(self.cfg.density_decay), float(self.cfg.density_thresh),
                                       L.p(self.density_bitfield), L.p(self._mean_density_dev), L.p(self._occ_state),
                                       L.p(ws), L.stream()), 'occ_update')
        self._mean_count_stale = True

    def pin_march_bitfield(self, bitfield: Optional[torch.Tensor]):
        """Benchmark / debugging hook: march through THIS bitfield while update_state keeps maintaining density_grid,
        mean_density and density_bitfield from the model as usual (None: back to density_bitfield).  Synthetic
        throughput runs use it: a random-initialised model has no scene, so the seeded synthetic occupancy stands in for
        the converged one while the periodic update still runs at full cost inside the step."""
        self._pinned_bitfield = bitfield
        return self

    @property
    def march_bitfield(self):
        return self.density_bitfield if self._pinned_bitfield is None else self._pinned_bitfield

    # ---- training render -----------------------------------------------------------------------
    def sample_capacity(self, n_rays: int) -> int:
        per_ray = self.cfg.max_steps if self.samples_per_ray_cap is None else min(self.samples_per_ray_cap,
                                                                               self.cfg.max_steps)
        return n_rays * per_ray

    def render_train(self, rays: RayBatch, **kwargs):
        """renderer.py:196-235 -> (image [N,3], depth [N], classes [N,nc]).  Three stages that graph.GraphedPatchBackward
        also drives one by one: march (sync-free), optional spatial order of the samples, shade (field + composite)."""
        if self.occupancy_update_due():
            self.update_state()
        mt = self.march_train(rays)
        perm = None
        if torch.is_grad_enabled() and self._use_spatial_order(mt['N'], bool(kwargs.get('dense', False))):
            perm = self.model.sample_order(mt['xyzs'], mt['counter'])
        return self.shade_train(mt, perm)

    def march_train(self, rays: RayBatch, into: Optional[dict] = None) -> dict:
        """near/far + occupancy-grid march + compaction: capacity-sized sample buffers, device-side counts.
        into: the dict a previous call returned -- its tensors are written in place (static buffers of a captured step)."""
        if into is not None:
            nears, fars = raymarching.near_far_from_aabb_into(rays.origins, rays.dirs, self.aabb, self.cfg.min_near,
                                                              into['nears'], into['fars'])
            counter = into['counter']
            counter.zero_()
            self._last_counter = counter
            N, M = into['N'], into['M']
            assert rays.origins.shape[0] == N
            self._last_capacity = M
            raymarching.march_rays_train_nosync(
                rays.origins, rays.dirs, self.bound, self.march_bitfield, self.cascade, self.cfg.grid_size, nears, fars,
                M, counter, 0., self.cfg.max_steps, out=(into['xyzs'], into['deltas'], into['rays_info']))
            if self.update_occ:
                # the step bookkeeping of the allocating path: the count ring behind mean_count, the occupancy schedule's step
                self.step_counter[self.local_step % STEP_CTR_SIZE].copy_(counter)
                self.local_step += 1
            return into
        nears, fars = raymarching.near_far_from_aabb(rays.origins, rays.dirs, self.aabb, self.cfg.min_near)
        if self.update_occ:
            counter = self.step_counter[self.local_step % STEP_CTR_SIZE]
            counter.zero_()
            self.local_step += 1
        else:
            counter = torch.zeros(2, dtype=torch.int32, device=self.device)

        self._last_counter = counter     # device-side (samples, rays) of this call; never read here
        N = rays.origins.shape[0]
        M = self.sample_capacity(N)
        self._last_capacity = M
        xyzs, _, deltas, rays_info = raymarching.march_rays_train_nosync(
            rays.origins, rays.dirs, self.bound, self.march_bitfield, self.cascade, self.cfg.grid_size, nears, fars,
            M, counter, 0., self.cfg.max_steps)
        return {'N': N, 'M': M, 'nears': nears, 'fars': fars, 'counter': counter, 'xyzs': xyzs, 'deltas': deltas,
                'rays_info': rays_info}

    def shade_train(self, mt: dict, perm=None):
        """fused field on the marched samples + train composite with the epilogue of renderer.py:225-233 folded in
        (nsr_render_train_forward / _backward): two autograd nodes, no torch glue between them"""
        sigmas, rgbs = self.model.field(mt['xyzs'], sigma_only=False, m_dev=mt['counter'], density_scale=self.cfg.density_scale,
                                        perm=perm)
        image, depth, classes, _ = _render_train(sigmas, rgbs, mt['deltas'], mt['rays_info'], mt['nears'], mt['fars'],
                                                 self.cfg.t_thresh)
        return image, depth, classes

    def _use_spatial_order(self, n_rays: int, dense: bool) -> bool:
        if getattr(self.model, '_spatial_scatter_unsupported', False):
            return False                 # learnt from a backward that fell back (style_nerf._field.backward)
        if self.sort_samples != 'auto':
            return bool(self.sort_samples)
        return n_rays >= self.sort_min_rays or (dense and n_rays >= self.sort_min_dense_rays)

    def last_call_overflowed(self) -> torch.Tensor:
        """Device-side flag (0-dim bool tensor, no host sync) of the last render_train call: the march emitted
        at least `capacity` samples, so the rays at the end of the batch were dropped exactly like the reference's
        mean_count path does (raymarching.cu:517).  Their in-buffer samples are zero-filled and carry zero gradient."""
        return self._last_counter[0] >= self._last_capacity

    # ---- inference render ----------------------------------------------------------------------
    @torch.no_grad()
    def render_test(self, rays: RayBatch, **kwargs):
        """renderer.py:237-293.  By default ONE march + ONE fused field launch + ONE composite with the
        inference kernel's arithmetic (nsr_composite_rays_infer) instead of up to max_steps host
        iterations; `self.reference_inference_loop = True` selects the reference's loop structure
        (render_test_loop), which the tests hold equal to this path."""
        if getattr(self, 'reference_inference_loop', False):
            return self.render_test_loop(rays, **kwargs)
        from . import _lib as L
        nears, fars = raymarching.near_far_from_aabb(rays.origins, rays.dirs, self.aabb, self.cfg.min_near)
        N = rays.origins.shape[0]
        M = self.sample_capacity(N)
        counter = torch.zeros(2, dtype=torch.int32, device=self.device)
        xyzs, _, deltas, rays_info = raymarching.march_rays_train_nosync(
            rays.origins, rays.dirs, self.bound, self.march_bitfield, self.cascade, self.cfg.grid_size, nears, fars,
            M, counter, 0., self.cfg.max_steps)
        if self.samples_per_ray_cap is not None and self.samples_per_ray_cap < self.cfg.max_steps:
            # a bounded buffer can overflow, and the march then DROPS the rays that do not fit (raymarching.cu:517);
            # the reference's inference loop never drops a ray, so fall back to its iteration structure.  One host read
            # per frame (the reference reads the alive count every iteration); never taken with the default capacity.
            self.last_test_overflow = int(counter[0].item()) >= M
            if self.last_test_overflow:
                return self.render_test_loop(rays, **kwargs)
        sigmas, rgbs = self.model.field(xyzs, sigma_only=False, m_dev=counter, density_scale=self.cfg.density_scale)
        C = self.raymarch_channels
        weights_sum = torch.empty(N, dtype=torch.float32, device=self.device)
        depth = torch.empty(N, dtype=torch.float32, device=self.device)
        image = torch.empty(N, C, dtype=torch.float32, device=self.device)
        L.check(L.lib().nsr_composite_rays_infer(L.p(sigmas), L.p(rgbs), L.p(deltas), L.p(rays_info), L.p(nears), M, N, C,
                                                 float(self.cfg.t_thresh), L.p(weights_sum), L.p(depth), L.p(image),
                                                 L.stream()), 'composite_rays_infer')
        classes = image[:, 3:]
        image = image[:, :3] + (1 - weights_sum).unsqueeze(-1)
        depth = torch.clamp(depth - nears, min=0) / (fars - nears)
        return image, depth, classes

    @torch.no_grad()
    def render_test_loop(self, rays: RayBatch, **kwargs):
        """renderer.py:237-293.  Same iteration structure (n_step = max(min(N // n_alive, 8), 1));
        alive-ray compaction is a device scan instead of boolean-mask indexing."""
        nears, fars = raymarching.near_far_from_aabb(rays.origins, rays.dirs, self.aabb, self.cfg.min_near)
        N = len(rays)
        dev = self.device
        weights_sum = torch.zeros(N, dtype=torch.float32, device=dev)
        depth = torch.zeros(N, dtype=torch.float32, device=dev)
        image = torch.zeros(N, self.raymarch_channels, dtype=torch.float32, device=dev)
        n_alive = N
        rays_alive = torch.arange(n_alive, dtype=torch.int32, device=dev)
        rays_alive_next = torch.empty_like(rays_alive)
        n_out = torch.empty(1, dtype=torch.int32, device=dev)
        rays_t = nears.clone()[:, None].contiguous()
        step = 0
        while step < self.cfg.max_steps:
            if n_alive <= 0:
                break
            n_step = max(min(N // n_alive, 8), 1)
            xyzs, _, deltas = raymarching.march_rays(
                n_alive, n_step, rays_alive, rays_t, rays.origins, rays.dirs, None, self.bound, self.march_bitfield,
                self.cascade, self.cfg.grid_size, nears, fars, 128, False, 0., self.cfg.max_steps, self.cfg.use_ndc)
            sigmas, rgbs = self.model.field(xyzs, sigma_only=False, density_scale=self.cfg.density_scale)
            raymarching.composite_rays(n_alive, n_step, rays_alive, rays_t, sigmas, rgbs, deltas, self.cfg.use_ndc,
                                       weights_sum, depth, image, self.cfg.t_thresh)
            raymarching.compact_alive(rays_alive, n_alive, rays_alive_next, n_out)
            rays_alive, rays_alive_next = rays_alive_next, rays_alive
            n_alive = int(n_out.item())
            step += n_step
        classes = image[:, 3:]
        image = image[:, :3]
        image = image + (1 - weights_sum).unsqueeze(-1)
        depth = torch.clamp(depth - nears, min=0) / (fars - nears)
        return image, depth, classes

    # ---- a training render in two halves (data-parallel overlap, parallel.py) ---------------------------------------
    def occupancy_update_due(self) -> bool:
        """The next training render starts with update_state (renderer.py:206-207): it reads the parameters."""
        return bool(self.update_occ and (self.local_step % self.cfg.update_iter == 0))

    def begin_train(self, pose, pix_subset, dense: bool = False, into: Optional[dict] = None) -> dict:
        """First half of render(pose, training=True, pix_subset=...): ray generation, [occupancy update when due,] occupancy
        march + compaction and the spatial order of the samples.  Apart from the occupancy update none of it reads the
        PARAMETERS (the march reads the bitfield only), so a data-parallel step may issue it for step i+1 while the gradient
        all-reduce of step i is in flight and the optimiser has not stepped yet -- as long as occupancy_update_due() is
        False.  finish_train(ctx) does the rest."""
        rays, _ = generate_rays(pose, self.intr, None, camera_flip=self.cfg.flip_camera, pix_subset=pix_subset, device=self.device)
        if self.occupancy_update_due():
            self.update_state()
        # into: the ctx a previous call returned; every tensor of it is rewritten in place (graph.GraphedRenderStep(prefetch=True))
        mt = self.march_train(rays, into=into['mt'] if into is not None else None)
        perm = None
        if torch.is_grad_enabled() and self._use_spatial_order(mt['N'], dense):
            perm = self.model.sample_order(mt['xyzs'], mt['counter'], out=into['perm'] if into is not None else None)
        if into is not None:
            assert (perm is None) == (into['perm'] is None)
            return into
        return {'mt': mt, 'perm': perm}

    def begin_train_on(self, stream, pose, pix_subset, dense: bool = False, after=None, into: Optional[dict] = None) -> dict:
        """begin_train issued on `stream` (a side stream) so that it runs BESIDE whatever the current stream is doing -- the
        previous step's backward: its table scatter waits on the memory-side atomic unit with most of its issue slots idle, and
        the march / sort kernels fit next to it.  finish_train(ctx) makes the current stream wait for it.  Only legal when no
        occupancy update is due (occupancy_update_due(): that one reads the parameters).  after: an event the side stream waits
        for first (the producer of `pose` / `pix_subset`, the last occupancy update, the last reader of `into`).
        into: the ctx of an earlier call whose tensors are rewritten in place -- a training loop alternates between two of them
        and allocates nothing per step.  Without it the tensors are allocated by the side stream and consumed -- and released
        -- by the current one: they are marked for the caching allocator accordingly (which then keeps two pools of
        capacity-sized buffers busy; prefer `into`)."""
        assert not self.occupancy_update_due(), 'an occupancy update reads the parameters: march on the training stream'
        cur = torch.cuda.current_stream(self.device)
        if after is not None:
            stream.wait_event(after)
        with torch.cuda.stream(stream):
            ctx = self.begin_train(pose, pix_subset, dense, into=into)
        if into is None:
            for t in list(ctx['mt'].values()) + [ctx['perm']]:
                if torch.is_tensor(t):
                    t.record_stream(cur)
        ctx['stream'] = stream
        return ctx

    def finish_train(self, ctx: dict) -> Dict[str, torch.Tensor]:
        """Second half: fused field + composite + epilogue on the samples begin_train marched (reads the parameters)."""
        if ctx.get('stream') is not None:
            torch.cuda.current_stream(self.device).wait_stream(ctx['stream'])
        out = {'target': None}
        out['rgb_map'], out['trans_map'], out['classes'] = self.shade_train(ctx['mt'], ctx['perm'])
        self._last_counter, self._last_capacity = ctx['mt']['counter'], ctx['mt']['M']
        return out

    def render(self, pose, image=None, patch: Optional[Box2D] = None, num_rays: Optional[int] = None,
               training: bool = False, pix_subset=None, dense: Optional[bool] = None) -> Dict[str, torch.Tensor]:
        """renderer.py:295-313.  `dense` overrides the guess below for callers that pass a patch as `pix_subset`."""
        output = {}
        precrop_frac = self.precrop_frac if self._use_precrop else 1.
        rays, output['target'] = generate_rays(pose, self.intr, image, patch=patch, precrop=precrop_frac, bsize=num_rays,
                                               camera_flip=self.cfg.flip_camera, pix_subset=pix_subset,
                                               device=self.device)
        render_fn = self.render_train if training else self.render_test
        # a full frame, a patch or a centre crop is a dense pixel set: neighbouring rays share hash-table rows
        if dense is None:
            dense = pix_subset is None and num_rays is None
        output['rgb_map'], output['trans_map'], output['classes'] = render_fn(rays, dense=dense)
        return output


class _render_train_fn(torch.autograd.Function):
    """composite_rays_train (raymarching.cu:806-997) + `image[:, :3] + (1 - weights_sum)`, the class slice and the depth
    normalisation (renderer.py:229-233) as ONE kernel each way, over capacity-sized sample buffers.  Outputs
    (rgb_map [N,3], depth [N], classes [N,nc], weights_sum [N]); gradients are produced with torch.empty (the backward kernel
    writes every sample that belongs to a ray, zeros included)."""

    @staticmethod
    def forward(ctx, sigmas, rgbs, deltas, rays, nears, fars, T_thresh):
        ctx.set_materialize_grads(False)     # outputs the loss does not use arrive as None, not as zero tensors (two fills a step)
        from . import _lib as L
        from . import profiling
        M, N, C = sigmas.shape[0], rays.shape[0], rgbs.shape[1]
        dev = sigmas.device
        weights_sum = torch.empty(N, dtype=torch.float32, device=dev)
        depth_raw = torch.empty(N, dtype=torch.float32, device=dev)
        image = torch.empty(N, C, dtype=torch.float32, device=dev)
        rgb_map = torch.empty(N, 3, dtype=torch.float32, device=dev)
        depth = torch.empty(N, dtype=torch.float32, device=dev)
        classes = torch.empty(N, C - 3, dtype=torch.float32, device=dev)
        with profiling.timed('composite_fwd'):
            L.check(L.lib().nsr_render_train_forward(
                L.p(sigmas), L.p(rgbs), L.p(deltas), L.p(rays), L.p(nears), L.p(fars), M, N, C, float(T_thresh), L.p(weights_sum),
                L.p(depth_raw), L.p(image), L.p(rgb_map), L.p(depth), L.p(classes) if C > 3 else None, L.stream()),
                'render_train_forward')
        ctx.save_for_backward(sigmas, rgbs, deltas, rays, weights_sum, image)
        ctx.T_thresh = T_thresh
        ctx.mark_non_differentiable(depth)
        return rgb_map, depth, classes, weights_sum

    @staticmethod
    def backward(ctx, g_rgb, g_depth, g_classes, g_ws):
        from . import _lib as L
        from . import profiling
        sigmas, rgbs, deltas, rays, weights_sum, image = ctx.saved_tensors
        M, N, C = sigmas.shape[0], rays.shape[0], rgbs.shape[1]

        def prep(g):
            return None if g is None else g.to(torch.float32).contiguous()
        g_rgb, g_classes, g_ws = prep(g_rgb), prep(g_classes), prep(g_ws)
        if C == 3:
            g_classes = None
        grad_sigmas = torch.empty_like(sigmas)
        grad_rgbs = torch.empty_like(rgbs)
        with profiling.timed('composite_bwd'):
            L.check(L.lib().nsr_render_train_backward(
                L.p(g_rgb), L.p(g_classes), L.p(g_ws), L.p(sigmas), L.p(rgbs), L.p(deltas), L.p(rays), L.p(weights_sum), L.p(image),
                M, N, C, float(ctx.T_thresh), L.p(grad_sigmas), L.p(grad_rgbs), L.stream()), 'render_train_backward')
        return grad_sigmas, grad_rgbs, None, None, None, None, None


_render_train = _render_train_fn.apply
