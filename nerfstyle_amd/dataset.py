"""HBM-resident training targets.

The reference's `generate_rays` slices a host NumPy image per step and copies the picked pixels to the
device (`nerf_lib.py:126-141`).  Here every frame's per-pixel target row -- RGB plus the segment id, the
4-channel layout `generate_rays` hands to `calc_loss` (`trainers/base.py:266-269`) -- and every pose live in
HBM for the whole run (35 frames at 1008x756 are 427 MB of the 288 GB), so a step gathers its targets with
one device index op.

File parsing (transforms JSON, PNG decode, segment .npz) is the caller's: `data/` is outside the hot path
 and is not rebuilt here.  `ResidentDataset` takes the decoded arrays."""
from typing import Optional

import numpy as np
import torch

from .common import BBox, Intrinsics


class ResidentDataset:
    """targets [N, H*W, C] (C = 3, or 4 with the segment id as float like the reference's 4th plane,
    base_dataset.py:140-148) and poses [N,4,4], both on `device`."""

    def __init__(self, images: np.ndarray, poses: np.ndarray, intr: Intrinsics, bound: float, device,
                 seg_maps: Optional[np.ndarray] = None, num_classes: int = 0):
        images = np.asarray(images, np.float32)
        poses = np.asarray(poses, np.float32)
        n, c, h, w = images.shape
        if c != 3:
            raise ValueError('images must be [N,3,H,W] (alpha already composited)')
        if (h, w) != (intr.h, intr.w):
            raise ValueError('images are {}x{}, intrinsics say {}x{}'.format(w, h, intr.w, intr.h))
        if poses.shape != (n, 4, 4):
            raise ValueError('poses must be [N,4,4]')
        planes = [torch.from_numpy(images).reshape(n, 3, h * w)]
        if seg_maps is not None:
            seg = np.asarray(seg_maps, np.float32)
            if seg.shape != (n, h, w):
                raise ValueError('seg_maps must be [N,H,W]')
            planes.append(torch.from_numpy(seg).reshape(n, 1, h * w))
        # pixel-major rows: one index_select row per ray
        self.targets = torch.cat(planes, dim=1).transpose(1, 2).contiguous().to(device)
        self.poses = torch.from_numpy(poses).to(device)
        self.intr, self.num_classes, self.bound = intr, int(num_classes), bound
        self.bbox = BBox.from_radius(bound)

    def __len__(self):
        return self.poses.shape[0]

    def sample(self, frame: int, pix: torch.Tensor):
        """(pose [4,4], target [len(pix), C]) for pixel ids `pix` (row-major y * w + x) of frame `frame`"""
        return self.poses[frame], self.targets[frame].index_select(0, pix)
