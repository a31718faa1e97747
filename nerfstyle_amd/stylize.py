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

Multi-GPU: patches are independent units, so pass 2 shards the patch list across ranks and pass 1 shards
pixel rows (all-gathered, <= 9.1 MB for 1008x756); the colour-table gradient is all-reduced with the
rest of the arena by parallel.sync_gradients.
"""
from typing import Callable, List

import torch

from . import parallel as P
from .common import Box2D


def patch_list(w: int, h: int, patch: int) -> List[Box2D]:
    """Row-major patches as the reference enumerates them (style.py:189-191)."""
    out = []
    for y in range(0, h, patch):
        for x in range(0, w, patch):
            out.append(Box2D(x, y, min(patch, w - x), min(patch, h - y)))
    return out


@torch.no_grad()
def render_full_frame(renderer, pose, rank: int = 0, world: int = 1):
    """Pass 1.  Returns rgb_map [H, W, 3] (identical on every rank)."""
    W, H = renderer.intr.size()
    if world == 1:
        out = renderer.render(pose, None, training=True)
        return out['rgb_map'].view(H, W, 3)
    y0, y1 = P.shard_bounds(H, rank, world)
    part = renderer.render(pose, None, patch=Box2D(0, y0, W, y1 - y0), training=True)['rgb_map'] if y1 > y0 else \
        torch.empty(0, 3, device=renderer.device)
    sizes = [(P.shard_bounds(H, r, world)[1] - P.shard_bounds(H, r, world)[0]) * W for r in range(world)]
    chunks = [torch.empty(n, 3, device=renderer.device) for n in sizes]
    torch.distributed.all_gather(chunks, part.contiguous()) if len(set(sizes)) == 1 else _all_gather_ragged(chunks, part)
    return torch.cat(chunks, 0).view(H, W, 3)


def _all_gather_ragged(chunks, part):
    n = max(c.shape[0] for c in chunks)
    pad = torch.zeros(n, 3, device=part.device)
    pad[:part.shape[0]] = part
    bufs = [torch.empty_like(pad) for _ in chunks]
    torch.distributed.all_gather(bufs, pad)
    for c, b in zip(chunks, bufs):
        c.copy_(b[:c.shape[0]])


def deferred_backprop_step(renderer, pose, image_loss: Callable[[torch.Tensor], torch.Tensor], patch_size: int = 200,
                           loss_scale: float = 1.0, rank: int = 0, world: int = 1, only_color_table: bool = True):
    """One stylisation iteration up to (not including) the optimiser step.  `image_loss` maps
    rgb [H, W, 3] (requires_grad) to a scalar.  Gradients accumulate into model.arena.grad.
    Returns (loss value, rgb_map of pass 1)."""
    W, H = renderer.intr.size()
    rgb = render_full_frame(renderer, pose, rank, world).detach().requires_grad_(True)
    loss = image_loss(rgb)
    (loss * loss_scale).backward()
    grad_map = rgb.grad                                   # [H, W, 3]  (style.py:187)
    patches = patch_list(W, H, patch_size)
    b, e = P.shard_bounds(len(patches), rank, world)
    for box in patches[b:e]:
        out = renderer.render(pose, None, patch=box, training=True)
        g = grad_map[box.y:box.y + box.h, box.x:box.x + box.w].reshape(-1, 3)
        out['rgb_map'].backward(g)                         # style.py:196-198
    if world > 1:
        P.sync_gradients(renderer.model, only_color_table=only_color_table)
    return loss.detach(), rgb.detach()
