"""hipGraph capture of the render + loss + backward part of a training step.

The training path is free of host synchronisation (march compaction counts stay on the device, the
field kernels split their tiles from the device-side count), so the whole forward and backward can be
captured once with torch.cuda.CUDAGraph (hipGraph on ROCm) and replayed: one graph launch instead of
~40 kernel launches and their host-side bookkeeping.  What it buys is launch overhead, i.e. it matters
for the reference's 4096-ray steps, not for full-frame steps.  Given a FusedAdam and a device-side
LossScaler the optimiser step is part of the graph (learning rate, step count, loss scale and the skip
decision are device scalars), and with prefetch=True the graph forks after the backward: the optimiser
on one branch, the next step's ray generation + march on the other.

Reference call stack this replaces: trainers/base.py:367-426 (one `calc_loss` + `backward`)."""
from typing import Callable, Dict

import torch

from .renderer import Renderer


class GraphedRenderStep:
    """graph = GraphedRenderStep(renderer, n_rays, loss_fn); loss = graph(pose, pix)
    (prefetch=True: loss = graph(pose, pix, pose_next, pix_next) -- the next call's pose and pixels, the SAME tensor objects)

    loss_fn(output_dict, pix) -> scalar loss tensor; it must index any per-pixel targets with the `pix`
    tensor it is given (a static buffer refilled before every replay) and must not synchronise.
    Gradients accumulate into model.arena.grad exactly as in eager mode."""

    def __init__(self, renderer: Renderer, n_rays: int, loss_fn: Callable[[Dict[str, torch.Tensor], torch.Tensor], torch.Tensor],
                 warmup: int = 2, dense: bool = False, optimizer=None, scaler=None, lr_decay_steps: float = 0.0,
                 prefetch: bool = False):
        # optimizer (FusedAdam) + scaler (LossScaler): the optimiser step joins the graph.  Possible since round 3: learning rate,
        # step count, bias corrections, loss scale, skip decision and the EMA decay are device-side scalars (nsr_scaler_update),
        # so nothing the captured kernels need changes on the host from step to step.  Without them the caller steps the
        # optimiser after the replay (e.g. data-parallel runs: the gradient all-reduce sits between backward and optimiser).
        assert (optimizer is None) == (scaler is None), 'a captured optimiser step needs the device-side LossScaler'
        self.optimizer, self.scaler, self.lr_decay_steps = optimizer, scaler, lr_decay_steps
        # dense: the pixel sets are patches / crops / whole frames (neighbouring pixels): Renderer._use_spatial_order then picks
        # the spatial sample order + lattice scatter from 16 384 rays on, inside the graph like everything else
        self.dense = dense
        # Occupancy updates keep their schedule (every cfg.update_iter steps, renderer.py:206-207) but run BETWEEN replays:
        # the full / partial update have different launch shapes, and the per-step counter ring of the eager path
        # changes its pointer every step.  The device-side update (nsr_occ_*) needs no host read, so it costs no bubble.
        self.occ_updates = bool(renderer.update_occ)
        self.r = renderer
        dev = renderer.device
        self.pose = torch.zeros(4, 4, dtype=torch.float32, device=dev)
        self.pose[:3, :3] = torch.eye(3, device=dev)
        self.pix = torch.arange(n_rays, dtype=torch.int64, device=dev)
        self.loss_fn = loss_fn
        self.graph = None
        self.loss = None
        self._warmup = warmup
        # prefetch: the captured step is  shade + loss + backward of THIS step's samples  ->  { optimiser step  ||  ray generation +
        # march + compaction (+ sample order) of the NEXT step's pixels }.  The march reads the occupancy bitfield only, never the
        # parameters, so it may run beside the optimiser: at 4 096 rays the ~0.15 ms of latency-bound march kernels disappear
        # under the ~0.21 ms of the HBM-bound inf/nan check + Adam pass.  The marched samples live in static buffers (`_stage`)
        # that the next replay's shade reads; graph(pose, pix, pose_next, pix_next) -- a call whose (pose, pix) are not the
        # objects announced by the previous call, or that follows an occupancy update, marches eagerly first.
        self.prefetch = bool(prefetch)
        self.pose_next = self.pose.clone()
        self.pix_next = self.pix.clone()
        self._stage = None               # ctx of Renderer.begin_train whose tensors are the static sample buffers
        self._announced = (None, None)

    def _restage(self, pose, pix):
        """march (pose, pix) now, on the current stream, into the static sample buffers (allocated by the first call)"""
        keep = self.r.update_occ
        self.r.update_occ = False          # no occupancy update and no step bookkeeping in here: __call__ keeps the schedule
        try:
            self._stage = self.r.begin_train(pose, pix, dense=self.dense, into=self._stage)
        finally:
            self.r.update_occ = keep

    def _body_prefetch(self, with_optimizer=True):
        out = self.r.finish_train(self._stage)
        loss = self.loss_fn(out, self.pix)
        if loss.requires_grad:
            loss.backward()
        main = torch.cuda.current_stream(self.r.device)
        side = self._side
        side.wait_stream(main)                                   # fork: the backward no longer needs the staged samples
        with torch.cuda.stream(side):
            self._restage(self.pose_next, self.pix_next)
            self.pix.copy_(self.pix_next)                        # the loss of the NEXT replay gathers its targets with these
        if with_optimizer and self.optimizer is not None:
            self.optimizer.step(scaler=self.scaler, lr_decay_steps=self.lr_decay_steps)
        main.wait_stream(side)                                   # join
        return loss.detach()

    def _body(self, with_optimizer=True):
        keep = self.r.update_occ
        self.r.update_occ = False          # inside the graph: fixed launch sequence, a private device-side counter
        try:
            out = self.r.render(self.pose, None, training=True, pix_subset=self.pix, dense=self.dense)
        finally:
            self.r.update_occ = keep
        loss = self.loss_fn(out, self.pix)
        if loss.requires_grad:             # (recon_loss(..., backward=True) has back-propagated already)
            loss.backward()
        if with_optimizer and self.optimizer is not None:
            self.optimizer.step(scaler=self.scaler, lr_decay_steps=self.lr_decay_steps)
        return loss.detach()

    def capture(self, pose: torch.Tensor, pix: torch.Tensor, pose_next=None, pix_next=None):
        self.pose.copy_(pose)
        self.pix.copy_(pix)
        if self.prefetch:
            self.pose_next.copy_(pose_next if pose_next is not None else pose)
            self.pix_next.copy_(pix_next if pix_next is not None else pix)
        model = self.r.model
        model._ensure_grad()
        # the f16 gather copy must be current BEFORE capture: a captured replay reads the copy the optimiser keeps in sync
        # (FusedAdam refreshes it in its own pass) and must not bake a cast of stale data into the graph
        if model.table_dtype == torch.float16:
            model._gather_tables()
        # warm-up on a side stream (allocator pools, lazy initialisation), as torch.cuda.graphs asks
        s = torch.cuda.Stream(device=self.r.device)
        s.wait_stream(torch.cuda.current_stream())
        if self.optimizer is not None:
            # the scaler state lives on the device before capture (no allocation / upload inside it), and the EMA count moves there
            self.scaler.state_on(self.r.device)
            if getattr(self.optimizer, '_scaler', None) is not self.scaler:
                self.scaler.adopt_ema_updates(self.optimizer.ema_updates, self.r.device)
                self.optimizer._scaler = self.scaler
        with torch.cuda.stream(s):
            for _ in range(self._warmup):
                self._body(with_optimizer=False)   # warm-up renders only: no parameter update, no step counted
        torch.cuda.current_stream().wait_stream(s)
        model.arena.grad.zero_()           # the warm-up passes accumulated gradients
        if self.prefetch:
            self._side = torch.cuda.Stream(device=self.r.device)
            self._restage(self.pose, self.pix)             # this step's samples: eager, into what become the static buffers
            torch.cuda.synchronize(self.r.device)
        self.graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.graph):
            self.loss = self._body_prefetch() if self.prefetch else self._body()
        self._counter = self._stage['mt']['counter'] if self.prefetch else self.r._last_counter   # the captured march's counter (refilled by every replay)
        model.arena.grad.zero_()           # capture does not execute, but keep the contract explicit
        return self

    def __call__(self, pose: torch.Tensor, pix: torch.Tensor, pose_next=None, pix_next=None) -> torch.Tensor:
        first = self.graph is None
        if first:
            self.capture(pose, pix, pose_next, pix_next)
        r = self.r
        updated = False
        if self.occ_updates:
            if r.local_step % r.cfg.update_iter == 0:
                r.update_state()
                updated = True
            r.local_step += 1
        m = r.model
        if m.table_dtype == torch.float16 and m._half_version != m.arena._version:
            # parameters changed by something other than FusedAdam (load_state_dict, an EMA swap, a stock optimiser):
            # refresh the f16 gather copy the captured kernels read
            m._gather_tables()
        if self.prefetch:
            # the staged samples (and the pixel ids the previous replay left in self.pix) are this step's only if the previous
            # replay marched exactly these pixels of this pose through the bitfield that is current now (never across an
            # occupancy update, like the data-parallel overlap); otherwise: march now
            if first or updated or self._announced[0] is not pose or self._announced[1] is not pix:
                self.pose.copy_(pose)
                self.pix.copy_(pix)
                self._restage(self.pose, self.pix)
            if pose_next is None:
                pose_next, pix_next = pose, pix
            self.pose_next.copy_(pose_next)
            self.pix_next.copy_(pix_next)
            self._announced = (pose_next, pix_next)
        else:
            self.pose.copy_(pose)
            self.pix.copy_(pix)
        self.graph.replay()
        self.r._last_counter = self._counter
        return self.loss

    @property
    def counter(self):
        """device-side (samples, rays) counter of the captured march"""
        return self._counter


class GraphedPatchBackward:
    """One patch of the deferred back-propagation (trainers/style.py:189-198) as ONE graph: render `n_rays` pixels of a frame
    with autograd and back-propagate a given d loss / d rgb into model.arena.grad.

        g = GraphedPatchBackward(renderer, n_rays); g(pose, pix, grad)      # pix [n_rays] positions in the frame, grad [n_rays, 3]

    A 1008x756 iteration re-renders 24 patches, ~45 launches each plus their host-side bookkeeping: eager, the GPU waits for
    the host between kernels (the kernel-event spans of the patch loop are 40 % longer than the kernels).  The graph holds ray
    generation + march + compaction, the spatial order of the samples (nsr_sample_order is capture-safe since round 3: its
    own radix sort, every counter reset by a kernel of the call) and the fused field + composite + their backward.  Patches of
    the same size share the graph (pose, pixel positions and the gradient are static buffers refilled before a replay).  The
    renderer must not update its occupancy grid in this stage (StyleTrainer never does)."""

    def __init__(self, renderer: Renderer, n_rays: int, dense: bool = True, warmup: int = 1):
        self.r = renderer
        dev = renderer.device
        self.pose = torch.zeros(4, 4, dtype=torch.float32, device=dev)
        self.pose[:3, :3] = torch.eye(3, device=dev)
        self.pix = torch.arange(n_rays, dtype=torch.int64, device=dev)
        self.grad = torch.zeros(n_rays, 3, dtype=torch.float32, device=dev)
        self.dense = dense
        self.graph = None
        self.counter = None
        self._warmup = warmup

    def _body(self):
        from .rays import generate_rays
        r = self.r
        keep = r.update_occ
        r.update_occ = False               # a private device-side counter, no step bookkeeping inside the graph
        try:
            rays, _ = generate_rays(self.pose, r.intr, None, camera_flip=r.cfg.flip_camera, pix_subset=self.pix, device=r.device)
            mt = r.march_train(rays)
        finally:
            r.update_occ = keep
        perm = r.model.sample_order(mt['xyzs'], mt['counter']) if r._use_spatial_order(mt['N'], self.dense) else None
        image, _, _ = r.shade_train(mt, perm)
        image.backward(self.grad)
        return mt['counter']

    def capture(self):
        """The static buffers must hold a real patch: the warm-up passes run on them, and their gradient is removed again."""
        model = self.r.model
        model._ensure_grad()
        if model.table_dtype == torch.float16:
            model._gather_tables()
        saved = model.arena.grad.clone()
        s = torch.cuda.Stream(device=self.r.device)
        s.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s):
            for _ in range(self._warmup):
                self._body()
        torch.cuda.current_stream().wait_stream(s)
        self.graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.graph):
            self.counter = self._body()
        model.arena.grad.copy_(saved)
        return self

    def __call__(self, pose: torch.Tensor, pix: torch.Tensor, grad: torch.Tensor) -> None:
        self.pose.copy_(pose)
        self.pix.copy_(pix)
        self.grad.copy_(grad)
        if self.graph is None:
            self.capture()
        m = self.r.model
        if m.table_dtype == torch.float16 and m._half_version != m.arena._version:
            m._gather_tables()
        self.graph.replay()
        self.r._last_counter = self.counter
