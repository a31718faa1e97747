"""Ray generation: counterpart of nerf_lib.generate_rays (nerf_lib.py:69-142) + RayBatch's
normalisation (common.py:139-147), done by one device kernel (nsr_generate_rays) instead of a
NumPy meshgrid + H2D copy + einsum per call."""
from typing import Optional

import numpy as np
import torch

from . import _lib as L
from .common import Box2D, Intrinsics, RayBatch


def pixel_ids(intr: Intrinsics, patch: Optional[Box2D] = None, precrop: float = 1., device=None):
    """Row-major pixel ids (y * W + x) of the full frame / centre crop / patch, as the reference
    slices x_coords / y_coords (nerf_lib.py:106-113).  Returns (ids int32 [w*h], w, h, dx, dy)."""
    assert 0. <= precrop <= 1.
    assert precrop >= 1. or patch is None, 'Using both precrop and patch is not supported'
    W, H = intr.size()
    w, h, dx, dy = W, H, 0, 0
    if precrop < 1.:
        w, h = int(W * precrop), int(H * precrop)
        dx, dy = (W - w) // 2, (H - h) // 2
    if patch is not None:
        dx, dy, w, h = patch.x, patch.y, min(patch.w, W - patch.x), min(patch.h, H - patch.y)
    ys = torch.arange(dy, dy + h, device=device, dtype=torch.int32)
    xs = torch.arange(dx, dx + w, device=device, dtype=torch.int32)
    ids = (ys[:, None] * W + xs[None, :]).reshape(-1)
    return ids, w, h, dx, dy


def generate_rays(pose, intr: Intrinsics, img=None, patch: Optional[Box2D] = None, precrop: float = 1.,
                  bsize: Optional[int] = None, camera_flip: int = 0, pix_subset=None, device=None):
    """Same arguments and returns as the reference (RayBatch, target).  bsize: that many pixels
    drawn without replacement with np.random.choice, exactly like nerf_lib.py:134 (pass
    `pix_subset`, positions into the cropped pixel list, to supply your own / device-side draw).
    img: [C, H, W] tensor -> target [K, C]."""
    device = device or (pose.device if torch.is_tensor(pose) else None)
    pose = torch.as_tensor(pose, dtype=torch.float32, device=device).contiguous()
    device = pose.device
    W, H = intr.size()
    if pix_subset is not None and patch is None and precrop >= 1.:
        # positions into the uncropped frame ARE the pixel ids: no id grid (two aranges, a multiply-add over the frame and a
        # gather -- five launches that a 4 096-ray step notices)
        ids = pix_subset.to(torch.int32).contiguous()
    else:
        ids, w, h, dx, dy = pixel_ids(intr, patch, precrop, device)
        if bsize is not None and pix_subset is None:
            pix_subset = torch.from_numpy(np.random.choice(np.arange(w * h), bsize, replace=False)).to(device)
        if pix_subset is not None:
            ids = ids[pix_subset.long()].contiguous()
    N = ids.shape[0]
    rays_o = torch.empty(N, 3, dtype=torch.float32, device=device)
    rays_d = torch.empty(N, 3, dtype=torch.float32, device=device)
    L.check(L.lib().nsr_generate_rays(L.p(pose), W, H, float(intr.fx), float(intr.fy), float(intr.cx), float(intr.cy),
                                      int(camera_flip), L.p(ids), N, L.p(rays_o), L.p(rays_d), L.stream()),
            'generate_rays')
    target = None
    if img is not None:
        if H != img.shape[-2] or W != img.shape[-1]:
            img = torch.nn.functional.interpolate(img.unsqueeze(0), size=(H, W)).squeeze(0)
        target = img.reshape(img.shape[0], -1).t()[ids.long()]
    rays = RayBatch.__new__(RayBatch)    # directions are already unit length (normalised in-kernel)
    rays.origins, rays.dirs = rays_o, rays_d
    return rays, target


def tile_order(w: int, h: int, tile_w: int = 8, tile_h: int = 8, device=None) -> torch.Tensor:
    """Every pixel id of a w x h frame once, visited tile by tile (tile_w x tile_h pixels, row-major inside a tile and over the
    tiles; edge tiles are partial).  For batches that cover the whole frame: `np.random.choice(w * h, w * h, replace=False)`
    (nerf_lib.py:134) is the whole frame in an order the loss -- a sum over the rays -- does not depend on, and in THIS order the
    64 rays of a marching wave cross the same occupancy cells, so the thread-per-ray march loop stops running both of its
    branches for the longest of 64 unrelated rays (bench frame: march 2.37 -> 1.87 ms, step 36.1 -> 34.8 ms)."""
    import numpy as np
    yy, xx = np.divmod(np.arange(w * h, dtype=np.int64), w)
    key = ((yy // tile_h) * ((w + tile_w - 1) // tile_w) + xx // tile_w) * (tile_w * tile_h) + (yy % tile_h) * tile_w + xx % tile_w
    return torch.as_tensor(np.argsort(key, kind='stable'), device=device)
