"""Shared builders for the parity tests (seeded inputs, small scenes)."""
import json
import os

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def room_cameras():
    with open(os.path.join(ROOT, 'nerfstyle_amd', 'assets', 'llff_room_cameras.json')) as f:
        return json.load(f)


def room_rays(O, n, seed=0, frame=0):
    """n random rays of LLFF room frame `frame` (flip_camera=3), via the oracle's generate_rays."""
    c = room_cameras()
    rng = np.random.default_rng(seed)
    idx = rng.choice(c['w'] * c['h'], size=n, replace=False)
    pose = np.asarray(c['poses'][frame], np.float32)
    return O.generate_rays(pose, c['w'], c['h'], c['fl_x'], c['fl_y'], c['cx'], c['cy'], 3, pix_indices=idx)


def small_scene(seed=0, n_boxes=48):
    """(density_grid [2, 128^3] f32 of 0/1, bitfield u8) for bound=2, H=128."""
    from nerfstyle_amd.scene import synthetic_density_grid
    grid = synthetic_density_grid(2.0, 128, n_boxes, seed)
    bits = np.packbits((grid.reshape(-1, 8) > 0.5)[:, ::-1], axis=1).reshape(-1)   # bit i of byte n = cell 8n+i
    return grid, bits


def rel_l2(a, b):
    a = np.asarray(a, np.float64)
    b = np.asarray(b, np.float64)
    return float(np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-30))
