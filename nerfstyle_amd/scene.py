"""Synthetic LLFF-shaped workload: the real LLFF 'room' cameras (poses and
intrinsics are data shipped in assets/), a seeded synthetic occupancy bitfield, seeded targets.
No images, segmentation maps or checkpoints of the reference exist offline, so every benchmark
and parity run uses this scene."""
import json
import os

import numpy as np

from .common import Intrinsics

_ASSETS = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'assets')


def load_cameras(scene: str = 'room', scale_res: int = 1):
    """-> (poses float32 [F,4,4] with the dataset's 0.33 translation scale applied, Intrinsics, meta dict) of the LLFF
    scene `scene` ('room': 35 training frames, 'fern': 17; the reference's transforms_train.json, a data file).
    scale_res=2 gives the 1008x756 frames of BASELINE config 2."""
    with open(os.path.join(_ASSETS, 'llff_{}_cameras.json'.format(scene))) as f:
        c = json.load(f)
    poses = np.asarray(c['poses'], dtype=np.float32)
    intr = Intrinsics(h=c['h'] * scale_res, w=c['w'] * scale_res, fx=c['fl_x'] * scale_res, fy=c['fl_y'] * scale_res,
                      cx=c['cx'] * scale_res, cy=c['cy'] * scale_res)
    return poses, intr, c


def load_room_cameras(scale_res: int = 1):
    return load_cameras('room', scale_res)


def _expand_bits(v):
    v = v.astype(np.uint32)
    v = (v * np.uint32(0x00010001)) & np.uint32(0xFF0000FF)
    v = (v * np.uint32(0x00000101)) & np.uint32(0x0F00F00F)
    v = (v * np.uint32(0x00000011)) & np.uint32(0xC30C30C3)
    v = (v * np.uint32(0x00000005)) & np.uint32(0x49249249)
    return v


def morton3d_np(x, y, z):
    return _expand_bits(x) | (_expand_bits(y) << np.uint32(1)) | (_expand_bits(z) << np.uint32(2))


def synthetic_density_grid(bound=2.0, grid_size=128, n_boxes=64, seed=0, extent=1.5, box_min=0.08, box_max=0.35):
    """Union of `n_boxes` random axis-aligned boxes inside [-extent, extent]^3, rasterised into
    every cascade in Morton order.  Returns density_grid float32 [cascade, H^3] holding 1.0 in
    occupied cells and 0.0 elsewhere (packbits with threshold 0.5 gives the bitfield)."""
    rng = np.random.default_rng(seed)
    cascade = 1 + int(np.ceil(np.log2(bound)))
    centers = rng.uniform(-extent, extent, size=(n_boxes, 3))
    half = rng.uniform(box_min, box_max, size=(n_boxes, 3))
    lo, hi = centers - half, centers + half
    H = grid_size
    ii = np.arange(H, dtype=np.uint32)
    X, Y, Z = np.meshgrid(ii, ii, ii, indexing='ij')
    mort = morton3d_np(X.reshape(-1), Y.reshape(-1), Z.reshape(-1)).astype(np.int64)
    grid = np.zeros((cascade, H ** 3), dtype=np.float32)
    for cas in range(cascade):
        b = min(2.0 ** cas, bound)
        cc = (np.arange(H) + 0.5) / H * 2 * b - b            # cell centres
        occ = np.zeros((H, H, H), dtype=bool)
        for k in range(n_boxes):
            mx = (cc >= lo[k, 0] - b / H) & (cc <= hi[k, 0] + b / H)
            my = (cc >= lo[k, 1] - b / H) & (cc <= hi[k, 1] + b / H)
            mz = (cc >= lo[k, 2] - b / H) & (cc <= hi[k, 2] + b / H)
            if mx.any() and my.any() and mz.any():
                occ[np.ix_(mx, my, mz)] = True
        grid[cas, mort] = occ.reshape(-1).astype(np.float32)
    return grid
