"""LLFF dataset (torch-ngp `transforms_{train,val,test}.json` format) and its HBM-resident form.

Mirrors `data/base_dataset.py:35-158` + `data/llff_dataset.py:12-52` of the reference: poses scaled by
`cfg.scale` (`base_dataset.py:64`), images [N,3,H,W] float32 in [0,1] with alpha composited onto white
(`:74-78`), per-pixel segment ids from `<root>/<seg_name>/<frame>_seg.npz['seg_map']` for the train split
(`llff_dataset.py:33-37`, classes must be 0..C-1 with optional -1 = unlabelled, `base_dataset.py:88-95`),
`max_count` uniform frame subset (`:109-120`), `__getitem__` -> (image with the segment plane as 4th
channel, pose) (`:140-148`).  Not mirrored: the colour-transfer option (`ct_image`, `:98-106`: stays in the
reference's utils) and CUDA-only side effects.

`ResidentDataset` is the MI355X-side addition: every frame's RGB+segment target and
pose live in HBM (a 35-frame 1008x756 scene is 427 MB of the 288 GB), so a training step gathers its
targets with one device index op instead of a host NumPy slice + H2D copy (`nerf_lib.py:134-141`)."""
import enum
import json
from pathlib import Path
from typing import List, Optional

import numpy as np
import torch

from .common import BBox, Intrinsics


class DatasetSplit(enum.Enum):
    TRAIN = 0
    VAL = 1
    TEST = 2


def parse_rgb(path) -> np.ndarray:
    """utils/__init__.py:425-441 without the resize branch: [C,H,W] float32 in [0,1]"""
    from PIL import Image
    img = np.array(Image.open(path), dtype=np.float32) / 255.0
    if img.ndim == 2:
        img = img[..., None]
    return np.ascontiguousarray(img.transpose(2, 0, 1))


class LLFFDataset(torch.utils.data.Dataset):
    def __init__(self, root_path, split: DatasetSplit = DatasetSplit.TRAIN, scale: float = 0.33, bound: float = 2.0,
                 seg_name: str = 'seg', max_count: Optional[int] = None):
        self.root = Path(root_path)
        self.split = split
        assert self.root.exists(), 'Root path "{}" does not exist'.format(self.root)
        with open(self.root / 'transforms_{}.json'.format(split.name.lower())) as f:
            self.split_json = json.load(f)
        frames = self.split_json['frames']
        self.poses = np.array([f['transform_matrix'] for f in frames], dtype=np.float32)
        assert self.poses.ndim == 3 and self.poses.shape[1:] == (4, 4)
        self.poses[:, :3, 3] *= scale
        n = len(self.poses)
        image_paths = None if split == DatasetSplit.TEST else [self.root / f['file_path'] for f in frames]
        self.has_gt = image_paths is not None
        if self.has_gt:
            self.fns: List[str] = [p.stem for p in image_paths]
            if len(set(self.fns)) != len(self.fns):
                self.fns = [p.parent.stem + '_' + p.stem for p in image_paths]
            self.images = np.stack([parse_rgb(p) for p in image_paths])
            if self.images.shape[1] == 4:
                rgb, alpha = self.images[:, :3], self.images[:, 3:]
                self.images = rgb * alpha + (1 - alpha)
            assert len(self.images) == n
        else:
            self.images = None
            w = len(str(n))
            self.fns = ['frame_{:0{w}d}'.format(i, w=w) for i in range(n)]
        self.seg_groups, self.num_classes = None, 0
        if split == DatasetSplit.TRAIN:
            self.seg_groups = np.stack([np.load(self.root / seg_name / '{}_seg.npz'.format(fn))['seg_map']
                                        for fn in self.fns]).astype(np.float32)
            groups = np.unique(self.seg_groups)
            if groups[0] < 0:
                groups = groups[1:]
            self.num_classes = len(groups)
            assert self.seg_groups.shape[-2:] == self.images.shape[-2:]
            assert np.all(groups == np.arange(self.num_classes)), 'segment ids must be 0..C-1 (and -1 for unlabelled)'
        if max_count is not None and max_count < n:
            assert max_count > 0, 'Invalid value for "max_count"'
            ids = np.round(np.linspace(0, n, max_count + 1)[:-1]).astype(int)
            self.fns = [self.fns[i] for i in ids]
            self.poses = self.poses[ids]
            if self.has_gt:
                self.images = self.images[ids]
            if self.seg_groups is not None:
                self.seg_groups = self.seg_groups[ids]
        j = self.split_json
        self.intr = Intrinsics(h=int(j['h']), w=int(j['w']), fx=j['fl_x'], fy=j['fl_y'], cx=j['cx'], cy=j['cy'])
        self.bound = bound
        self.bbox = BBox.from_radius(bound)

    def __len__(self):
        return len(self.poses)

    def __getitem__(self, index):
        if self.seg_groups is not None:
            return np.concatenate((self.images[index], self.seg_groups[index][None]), axis=0), self.poses[index]
        if self.has_gt:
            return self.images[index], self.poses[index]
        return None, self.poses[index]


class ResidentDataset:
    """All targets of a dataset in device memory: `targets [N, H*W, 4]` (RGB + segment id, the layout
    `generate_rays` returns per pixel, nerf_lib.py:126-141) and `poses [N,4,4]`."""

    def __init__(self, ds: LLFFDataset, device):
        assert ds.has_gt, 'the test split has no targets'
        n, _, h, w = ds.images.shape
        planes = ds.images if ds.seg_groups is None else np.concatenate((ds.images, ds.seg_groups[:, None]), axis=1)
        self.targets = torch.from_numpy(np.ascontiguousarray(planes.reshape(n, planes.shape[1], h * w).transpose(0, 2, 1))).to(device)
        self.poses = torch.from_numpy(ds.poses).to(device)
        self.intr, self.num_classes, self.bbox, self.bound = ds.intr, ds.num_classes, ds.bbox, ds.bound

    def __len__(self):
        return self.poses.shape[0]

    def sample(self, frame: int, pix: torch.Tensor):
        """(pose [4,4], target [len(pix), C]) for pixel indices `pix` (row-major y * w + x) of frame `frame`"""
        return self.poses[frame], self.targets[frame].index_select(0, pix)
