"""Data contracts of the hot path, same names and fields as the reference's common.py:
Box2D (:26), Intrinsics (:42), RayBatch (:130, normalises dirs in __post_init__ :139-147),
BBox (:243, normalize :276).  Plain dataclasses / tensors; no native code."""
from __future__ import annotations

from dataclasses import dataclass
from typing import Tuple

import numpy as np
import torch
from torch import Tensor


@dataclass(frozen=True)
class Box2D:
    x: int
    y: int
    w: int
    h: int

    def wrange(self):
        return slice(self.x, self.x + self.w)

    def hrange(self):
        return slice(self.y, self.y + self.h)


@dataclass(frozen=True)
class Intrinsics:
    h: int
    w: int
    fx: float
    fy: float
    cx: float
    cy: float

    def __post_init__(self):
        object.__setattr__(self, 'h', int(self.h))
        object.__setattr__(self, 'w', int(self.w))

    def size(self) -> Tuple[int, int]:
        return self.w, self.h

    def scale(self, w: int, h: int) -> 'Intrinsics':
        """common.py:91-113"""
        cx, cy = w / 2., h / 2.
        old_ar, new_ar = self.w / self.h, w / h
        ratio = h / self.h if new_ar >= old_ar else w / self.w
        return Intrinsics(h, w, self.fx * ratio, self.fy * ratio, cx, cy)


@dataclass
class RayBatch:
    origins: Tensor
    dirs: Tensor

    def __post_init__(self):
        assert len(self.origins.shape) <= 2
        assert len(self.dirs.shape) == 2
        if len(self.origins.shape) == 1:
            self.origins = torch.tile(self.origins, (len(self.dirs), 1))
        assert self.origins.shape == self.dirs.shape
        self.dirs = self.dirs / torch.norm(self.dirs, dim=-1, keepdim=True)

    def __len__(self):
        return len(self.dirs)


class BBox:
    """common.py:243-295 (min_pt / max_pt tensors, normalize)."""

    def __init__(self, bbox_min, bbox_max):
        self.min_pt = torch.tensor(np.asarray(bbox_min), dtype=torch.float32)
        self.max_pt = torch.tensor(np.asarray(bbox_max), dtype=torch.float32)

    @classmethod
    def from_radius(cls, radius: float) -> 'BBox':
        bbox_max = np.array([radius, radius, radius])
        return BBox(-bbox_max, bbox_max)

    @property
    def size(self) -> Tensor:
        return self.max_pt - self.min_pt

    def to(self, device):
        self.min_pt = self.min_pt.to(device)
        self.max_pt = self.max_pt.to(device)
        return self

    def normalize(self, pts: Tensor) -> Tensor:
        return (pts - self.min_pt) / self.size
