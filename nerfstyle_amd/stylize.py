"""Stylisation-stage glue around the HIP renderer: the deferred back-propagation of
StyleTrainer.run_iter (trainers/style.py:162-204).

  pass 1  full-frame render without autograd (style.py:177-179)
  loss    any differentiable image-space loss on rgb_map (the reference: VGG16 relu3 content MSE +
          SemanticStyleLoss, networks/fx.py + loss.py -- stock PyTorch, out of scope here) -> d loss / d rgb
  pass 2  for every patch of `defer_patch_size` pixels (style.py:189-198): render the patch rays with
          autograd and back-propagate the cached patch gradient into the HIP renderer

Only `x_color_embedder` is optimised in this stage (StyleTrainer.OPTIM_KEYS, style.py:25): build the
optimiser with FusedAdam(model, keywords=['x_color_embedder']) -- it also switches the density-table
scatter off in the fused backward.

resident_backprop_step is the same iteration WITHOUT the deferral (one render with autograd, activations resident in HBM):
what a 288 GB device should run; deferred_backprop_step keeps the reference's structure.

Multi-GPU: patches are independent units, so pass 2 shards the patch list across ranks and pass 1 shards
pixel rows (all-gathered, <= 9.1 MB for 1008x756); the colour-table gradient is all-reduced with the
rest of the arena by parallel.sync_gradients.
"""
from typing import Callable, List, Optional

import torch
import torch.nn.functional as F

from . import parallel as P
from .common import Box2D


class StyleCriterion:
    """The image-space loss of StyleTrainer.calc_loss (trainers/style.py:70-105): VGG16 'relu3' features of the rendered
    frame, the ground-truth frame and the style image; content = MSE(rendered, ground-truth features) * content_lambda,
    style = SemanticStyleLoss(rendered features | class predictions) * style_lambda (cfgs/training/style.yaml:
    content_lambda 0.001, style_lambda 1.0).  The features of the fixed images are computed once and cached (the
    reference re-extracts target and style features every iteration, style.py:88-89)."""

    def __init__(self, fx, style_loss, content_lambda: float = 0.001, style_lambda: float = 1.0, content_feat: str = 'relu3',
                 amp_dtype=None):
        """amp_dtype: torch.float16 / torch.bfloat16 runs the loss under torch.autocast, as the reference does with
        enable_amp (cfgs/training/default.yaml:16, style.py:182-184: convolutions and the feature products in half
        precision, reductions in fp32); None = fp32."""
        self.fx, self.style_loss = fx, style_loss
        self.content_lambda, self.style_lambda, self.content_feat = content_lambda, style_lambda, content_feat
        self.amp_dtype = amp_dtype
        self._target_feats = {}

    def _autocast(self, device):
        return torch.autocast(device_type=device.type, dtype=self.amp_dtype, enabled=self.amp_dtype is not None)

    @torch.no_grad()
    def init_style(self, style_image: torch.Tensor, num_classes: int):
        """style_image [3,H,W] in [0,1] (trainers/style.py:66-68)"""
        self.style_loss.init_feats(self.fx(style_image), num_classes=num_classes)

    @torch.no_grad()
    def target_features(self, key, target_chw: torch.Tensor):
        if key not in self._target_feats:
            self._target_feats[key] = self.fx(target_chw)[self.content_feat]
        return self._target_feats[key]

    def __call__(self, rgb_hw3: torch.Tensor, target_chw: torch.Tensor, classes_hwc: torch.Tensor, frame_key=None, it: int = 0):
        """rgb_hw3 [H,W,3] (requires_grad), target_chw [3,H,W], classes_hwc [H,W,nc] logits -> (total, content, style)"""
        with self._autocast(rgb_hw3.device):
            rgb_feats = self.fx(rgb_hw3.permute(2, 0, 1))
            tgt = self.target_features(frame_key, target_chw) if frame_key is not None else self.fx(target_chw)[self.content_feat]
            preds = torch.argmax(classes_hwc, dim=-1)                                   # style.py:85
            content = F.mse_loss(rgb_feats[self.content_feat], tgt) * self.content_lambda
            style = self.style_loss(rgb_feats, None, preds, it) * self.style_lambda
        return content + style, content.detach(), style.detach()


def patch_list(w: int, h: int, patch: int) -> List[Box2D]:
    """Row-major patches as the reference enumerates them (style.py:189-191)."""
    out = []
    for y in range(0, h, patch):
        for x in range(0, w, patch):
            out.append(Box2D(x, y, min(patch, w - x), min(patch, h - y)))
    return out


@torch.no_grad()
def render_full_frame(renderer, pose, rank: int = 0, world: int = 1, with_classes: bool = False):
    """Pass 1.  Returns rgb_map [H, W, 3] (identical on every rank); with_classes: (rgb_map, classes [H, W, nc])."""
    W, H = renderer.intr.size()
    C = renderer.raymarch_channels
    if world == 1:
        out = renderer.render(pose, None, training=True)
        full = torch.cat((out['rgb_map'], out['classes']), dim=1)
    else:
        y0, y1 = P.shard_bounds(H, rank, world)
        if y1 > y0:
            o = renderer.render(pose, None, patch=Box2D(0, y0, W, y1 - y0), training=True)
            part = torch.cat((o['rgb_map'], o['classes']), dim=1)
        else:
            part = torch.empty(0, C, device=renderer.device)
        sizes = [(P.shard_bounds(H, r, world)[1] - P.shard_bounds(H, r, world)[0]) * W for r in range(world)]
        chunks = [torch.empty(n, C, device=renderer.device) for n in sizes]
        if len(set(sizes)) == 1:
            torch.distributed.all_gather(chunks, part.contiguous())
        else:
            _all_gather_ragged(chunks, part)
        full = torch.cat(chunks, 0)
    rgb = full[:, :3].reshape(H, W, 3)
    return (rgb, full[:, 3:].reshape(H, W, C - 3)) if with_classes else rgb


def _all_gather_ragged(chunks, part):
    n = max(c.shape[0] for c in chunks)
    pad = torch.zeros(n, part.shape[1], device=part.device)
    pad[:part.shape[0]] = part
    bufs = [torch.empty_like(pad) for _ in chunks]
    torch.distributed.all_gather(bufs, pad)
    for c, b in zip(chunks, bufs):
        c.copy_(b[:c.shape[0]])


def deferred_backprop_step(renderer, pose, image_loss: Callable, patch_size: int = 200, loss_scale=1.0, rank: int = 0,
                           world: int = 1, optimizer=None, with_classes: bool = False,
                           patch_graphs: Optional[dict] = None):
    """One stylisation iteration up to (not including) the optimiser step.  `image_loss` maps rgb [H, W, 3]
    (requires_grad) -- and, with_classes, the class logits [H, W, nc] of the same pass -- to a scalar.
    Gradients accumulate into model.arena.grad.  Returns (loss value, rgb_map of pass 1).
    loss_scale: float or 0-dim device tensor (optim.LossScaler.scale_tensor).  optimizer: the FusedAdam that will step; at
    world > 1 the all-reduce covers exactly what it trains (parallel.sync_gradients) -- None reduces the whole arena.
    patch_graphs: a dict the caller keeps across iterations; when given, pass 2 replays one hipGraph per patch shape
    (graph.GraphedPatchBackward) instead of launching every patch's kernels from the host."""
    W, H = renderer.intr.size()
    if with_classes:
        rgb, classes = render_full_frame(renderer, pose, rank, world, with_classes=True)
        rgb = rgb.detach().requires_grad_(True)
        loss = image_loss(rgb, classes.detach())
    else:
        rgb = render_full_frame(renderer, pose, rank, world).detach().requires_grad_(True)
        loss = image_loss(rgb)
    (loss * loss_scale).backward()
    grad_map = rgb.grad                                   # [H, W, 3]  (style.py:187)
    patches = patch_list(W, H, patch_size)
    b, e = P.shard_bounds(len(patches), rank, world)
    # Graph replays rotate over the caller's stream and `streams` - 1 side streams (one graph instance per patch shape and
    # stream): a 40 000-ray patch leaves the chip half empty during its march, compaction, sort and composite kernels (157
    # workgroups of the march on 256 CUs, 13-60 us kernels), and the other patches' gather / MLP / scatter kernels fill it.
    # All accumulate into the gradient arena with atomics; the streams are joined before the function returns.  1008x756,
    # 24 patches, whole iteration: 1 stream 48.9 ms, 2 -> 44.2, 3 -> 42.5, 4 -> 40.8, 8 -> 40.7 (default 4; a graph instance
    # holds its own sample buffers, ~3 GB for a 200x200 patch at 160 samples per ray).
    nstreams = 1
    if patch_graphs is not None and patch_graphs.get('concurrent', True) and renderer.device.type == 'cuda':
        nstreams = max(1, int(patch_graphs.get('streams', 4)))
    main, sides = None, []
    if nstreams > 1:
        main = torch.cuda.current_stream(renderer.device)
        sides = patch_graphs.setdefault('side_streams', [])
        while len(sides) < nstreams - 1:
            sides.append(torch.cuda.Stream(device=renderer.device))
        for sd in sides[:nstreams - 1]:
            sd.wait_stream(main)
    pose_t = torch.as_tensor(pose, dtype=torch.float32, device=renderer.device) if patch_graphs is not None else None
    for i, box in enumerate(patches[b:e]):
        if patch_graphs is None:
            g = grad_map[box.y:box.y + box.h, box.x:box.x + box.w].reshape(-1, 3)
            out = renderer.render(pose, None, patch=box, training=True)
            out['rgb_map'].backward(g)                     # style.py:196-198
            continue
        from .graph import GraphedPatchBackward
        slot = i % nstreams
        with torch.cuda.stream(sides[slot - 1] if slot else torch.cuda.current_stream(renderer.device)):
            g = grad_map[box.y:box.y + box.h, box.x:box.x + box.w].reshape(-1, 3)
            rows = torch.arange(box.y, box.y + box.h, device=renderer.device)
            cols = torch.arange(box.x, box.x + box.w, device=renderer.device)
            pix = (rows[:, None] * W + cols[None, :]).reshape(-1)              # positions in the row-major frame
            key = (box.w * box.h, W, H, slot)
            if key not in patch_graphs:
                # capture runs warm-up passes whose gradient it removes again (save / restore of the arena): nothing else may be
                # accumulating meanwhile -- first use only
                torch.cuda.synchronize(renderer.device)
                patch_graphs[key] = GraphedPatchBackward(renderer, box.w * box.h, dense=True)
                patch_graphs[key](pose_t, pix, g)
                torch.cuda.synchronize(renderer.device)
            else:
                patch_graphs[key](pose_t, pix, g)
    for sd in sides[:nstreams - 1]:
        main.wait_stream(sd)
    if world > 1:
        P.sync_gradients(renderer.model, optimizer=optimizer)
    return loss.detach(), rgb.detach()


def resident_backprop_step(renderer, pose, image_loss: Callable, loss_scale=1.0, rank: int = 0, world: int = 1, optimizer=None,
                           with_classes: bool = False):
    """The stylisation iteration WITHOUT the deferral: the frame is rendered once, with autograd, its activations stay in HBM,
    and d loss / d rgb goes back through that one render.

    The reference renders the frame without autograd, takes the image loss's gradient and then re-renders the frame patch by
    patch with autograd to back-propagate it (trainers/style.py:162-204) because the activations of 762 048 rays do not fit
    the GPUs it was written for.  Here they do: marched samples, the sample order, the encoded features (128 B per sample) and
    the composite's sums of a 1008x756 frame are ~10 GB of 288, so nothing is marched, sorted, gathered or composited twice.
    The gradient is the same sum over the same rays (deterministic march, no perturbation): equal to deferred_backprop_step's
    up to fp32 summation order (tests).  1008x756 iteration: 38.3-39.4 -> 29.3 ms.

    Same arguments and return value as deferred_backprop_step (no patch_size / patch_graphs).  world > 1: every rank renders
    (and later back-propagates through) its band of rows; the bands are all-gathered so that every rank evaluates the image loss
    on the whole frame, as in render_full_frame."""
    W, H = renderer.intr.size()
    C = renderer.raymarch_channels
    y0, y1 = (0, H) if world == 1 else P.shard_bounds(H, rank, world)
    out = None
    if y1 > y0:
        out = renderer.render(pose, None, patch=None if world == 1 else Box2D(0, y0, W, y1 - y0), training=True)
        part = torch.cat((out['rgb_map'].detach(), out['classes'].detach()), dim=1)
    else:
        part = torch.empty(0, C, device=renderer.device)
    if world == 1:
        full = part
    else:
        sizes = [(P.shard_bounds(H, r, world)[1] - P.shard_bounds(H, r, world)[0]) * W for r in range(world)]
        chunks = [torch.empty(n, C, device=renderer.device) for n in sizes]
        if len(set(sizes)) == 1:
            torch.distributed.all_gather(chunks, part.contiguous())
        else:
            _all_gather_ragged(chunks, part)
        full = torch.cat(chunks, 0)
    rgb = full[:, :3].reshape(H, W, 3).clone().requires_grad_(True)
    loss = image_loss(rgb, full[:, 3:].reshape(H, W, C - 3)) if with_classes else image_loss(rgb)
    (loss * loss_scale).backward()
    if out is not None:
        out['rgb_map'].backward(rgb.grad.reshape(-1, 3)[y0 * W:y1 * W])
    if world > 1:
        P.sync_gradients(renderer.model, optimizer=optimizer)
    return loss.detach(), rgb.detach()
