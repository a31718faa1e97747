"""Image-space losses of the stylisation stage: the counterpart of the reference's loss.py (`cosine_dists` :32-36,
`GramStyleLoss` :45-66, `NNFMStyleLoss` :92-113, `SemanticStyleLoss` :116-214, `labels_downscale` :24-29,
`compute_centroid` :15-21).  They stay stock PyTorch: they run on whatever device their inputs
live on -- the reference's `.cuda()` calls are gone -- and back-propagate into the HIP renderer through
`stylize.deferred_backprop_step`.  Pinned against the reference's own outputs in tests/test_losses_cpu.py.

MI355X-side difference: the nearest-neighbour losses never materialise the [positions x positions] distance matrix
under autograd (11 844^2 floats at 504x378, 47 628^2 = 9 GB at 1008x756 -- plus the copies autograd keeps).  The
nearest style position of every image position is found chunk by chunk without autograd; the loss is then
1 - <f_hat, s_hat[nearest]>, which has the value AND the gradient of `amin(dists, dim=1)` (amin's gradient flows to the
arg-min element only)."""
from typing import Dict, List, Optional, Sequence, Union

import numpy as np
import torch
import torch.nn.functional as F
from torch import nn


def compute_centroid(mask: torch.Tensor) -> torch.Tensor:
    """loss.py:15-21: (row, column) centroid of a boolean mask, normalised by the mask's height / width."""
    H, W = mask.shape
    n = torch.sum(mask)
    r = torch.arange(H, device=mask.device)
    c = torch.arange(W, device=mask.device)
    r_mean = torch.sum(torch.sum(mask, dim=1) * r) / n / H
    c_mean = torch.sum(torch.sum(mask, dim=0) * c) / n / W
    return torch.stack((r_mean, c_mean))


def labels_downscale(labels: torch.Tensor, new_dim) -> torch.Tensor:
    """loss.py:24-29: nearest-sample down-scaling of a label map with linspace indices."""
    H, W = labels.shape
    NH, NW = new_dim
    r = torch.linspace(0, H - 1, NH, device=labels.device).long()
    c = torch.linspace(0, W - 1, NW, device=labels.device).long()
    return labels[r[:, None], c]


def cosine_dists(feats1: torch.Tensor, feats2: torch.Tensor) -> torch.Tensor:
    """loss.py:32-36: [N1, C], [N2, C] -> [N1, N2] of 1 - cos."""
    f1 = feats1 / torch.linalg.norm(feats1, dim=1)[:, None]
    f2 = feats2 / torch.linalg.norm(feats2, dim=1)[:, None]
    return 1.0 - torch.matmul(f1, f2.T)


def nearest_style_index(f1_hat: torch.Tensor, f2_hat: torch.Tensor, row_cluster: Optional[torch.Tensor] = None,
                        style_cluster: Optional[torch.Tensor] = None, chunk: int = 8192, style_groups=None):
    """Arg-min over style positions of the cosine distance, chunked over image positions, no autograd.
    row_cluster [N1] (long, -1 = unrestricted): image position n may only match style positions whose
    style_cluster equals row_cluster[n] (loss.py:201-206 sets the others to inf).  Returns (index [N1], valid [N1]).

    With clusters the positions are GROUPED instead of masked: the image positions of class k are multiplied with the
    style positions of cluster k only -- 1/K of the [N1 x N2] products, no mask, no inf fill (at 1008x756 the masked
    form is 3.5 TFLOP and five passes over a 2.3 G-entry matrix per iteration).  `style_groups` (list of K index tensors,
    the positions of every cluster) can be passed in when the style side is fixed; one host read (the class counts) per call."""
    N1 = f1_hat.shape[0]
    idx = torch.zeros(N1, dtype=torch.long, device=f1_hat.device)
    valid = torch.ones(N1, dtype=torch.bool, device=f1_hat.device)
    with torch.no_grad():
        f1d = f1_hat.detach()
        f2t = f2_hat.detach().T.contiguous()
        if row_cluster is None:
            for b in range(0, N1, chunk):
                idx[b:b + chunk] = torch.argmax(f1d[b:b + chunk] @ f2t, dim=1)      # arg-min of 1 - cos
            return idx, valid
        if style_groups is None:
            K = int(max(int(style_cluster.max()), int(row_cluster.max())) + 1) if style_cluster.numel() else 0
            style_groups = [torch.nonzero(style_cluster == k)[:, 0] for k in range(K)]
        K = len(style_groups)
        rc = torch.clamp(row_cluster, min=-1)
        order = torch.argsort(rc, stable=True)                                      # unrestricted rows first, then class 0, 1, ...
        counts = torch.bincount(rc + 1, minlength=K + 1).tolist()
        pos = 0
        for k in range(-1, len(counts) - 1):
            n = counts[k + 1]
            rows = order[pos:pos + n]
            pos += n
            if n == 0:
                continue
            cand = None if k < 0 else (style_groups[k] if k < K else style_groups[0][:0])
            if cand is not None and cand.numel() == 0:
                valid[rows] = False                                                 # a class without any allowed style position
                continue
            sub = f2t if cand is None else f2t[:, cand]
            for b in range(0, n, chunk):
                r = rows[b:b + chunk]
                j = torch.argmax(f1d[r] @ sub, dim=1)
                idx[r] = j if cand is None else cand[j]
    return idx, valid


def _nn_min_dists(f1: torch.Tensor, f2: torch.Tensor, row_cluster=None, style_cluster=None, style_groups=None) -> torch.Tensor:
    """[N1, C], [N2, C] -> [N1]: min_j (1 - cos(f1[n], f2[j])) with gradients w.r.t. both feature sets."""
    f1_hat = f1 / torch.linalg.norm(f1, dim=1)[:, None]
    f2_hat = f2 / torch.linalg.norm(f2, dim=1)[:, None]
    idx, valid = nearest_style_index(f1_hat, f2_hat, row_cluster, style_cluster, style_groups=style_groups)
    d = 1.0 - torch.sum(f1_hat * f2_hat[idx], dim=1)
    if row_cluster is not None:    # a class without any allowed style position: inf, as in the reference (no host read here)
        d = torch.where(valid, d, torch.full_like(d, float('inf')))
    return d


class StyleLoss(nn.Module):
    def __init__(self, keys: List[str]) -> None:
        super().__init__()
        self.keys = keys


class GramStyleLoss(StyleLoss):
    """loss.py:45-66"""

    @staticmethod
    def _gram_mtx(feats: torch.Tensor):
        H, W = feats.shape[-2:]
        f = feats.reshape(feats.shape[0], feats.shape[1], H * W)
        return torch.matmul(f, f.transpose(-2, -1)) / (H * W)

    def forward(self, feats1: Dict[str, torch.Tensor], feats2: Dict[str, torch.Tensor]) -> torch.Tensor:
        losses = [F.mse_loss(self._gram_mtx(feats1[k].float()), self._gram_mtx(feats2[k].float())) for k in self.keys]
        return torch.sum(torch.stack(losses))


class NNFMStyleLoss(StyleLoss):
    """loss.py:92-113: mean over image positions of the cosine distance to the nearest style feature."""

    def forward(self, feats1: Dict[str, torch.Tensor], feats2: Dict[str, torch.Tensor]) -> torch.Tensor:
        loss = 0
        for k in self.keys:
            f1, f2 = feats1[k].squeeze(0), feats2[k].squeeze(0)
            assert len(f1) == len(f2)
            f1 = f1.reshape(len(f1), -1).T          # [H*W, C]
            f2 = f2.reshape(len(f2), -1).T
            loss = loss + torch.mean(_nn_min_dists(f1, f2))
        return loss


class SemanticStyleLoss(StyleLoss):
    """loss.py:116-214.  `clusters`: the style image's segment map ([H, W] integer array / tensor, ids 0..n-1 with
    optional -1, the content of the reference's `clusters_path` .npz['seg_map']) or None; `matching`: optional fixed
    class -> cluster assignment (trainers/style.py:47-49), else computed once from the first rendered frame
    (update_matching, loss.py:166-184: feature-mean cosine distance + centroid distance, Hungarian assignment)."""

    def __init__(self, keys: Union[str, List[str]], clusters=None, matching: Optional[Sequence[int]] = None) -> None:
        super().__init__([keys] if isinstance(keys, str) else keys)
        self.ready = False
        self._style_groups = None
        self.use_matching = clusters is not None
        self.matching = None
        self.clusters = None
        if self.use_matching:
            c = np.asarray(clusters.cpu() if torch.is_tensor(clusters) else clusters)
            ids = np.unique(c)
            if ids[0] < 0:
                ids = ids[1:]
            self.n_clusters = len(ids)
            assert np.all(np.arange(self.n_clusters) == ids), 'cluster ids must be 0..n-1 (and -1 for unlabelled)'
            self.clusters = torch.as_tensor(c)
            self.matching = None if matching is None else [int(m) for m in matching]

    @torch.no_grad()
    def init_feats(self, all_style_feats: Dict[str, torch.Tensor], num_classes: int):
        """loss.py:144-164"""
        style_feats = all_style_feats[self.keys[0]].squeeze(0)
        self.style_feats = style_feats
        if self.use_matching:
            size = style_feats.shape[1:]
            cl = F.interpolate(self.clusters.to(style_feats.device)[None, None].float(), size)
            self.clusters = cl[0, 0].to(torch.long)
            self.style_feats_mean = torch.stack([torch.mean(style_feats[:, self.clusters == i], dim=1)
                                                 for i in range(self.n_clusters)])
            self.style_centroids = torch.stack([compute_centroid(self.clusters == i) for i in range(self.n_clusters)])
            self.num_classes = num_classes
        self.ready = True

    @torch.no_grad()
    def update_matching(self, image_feats: torch.Tensor, preds: torch.Tensor):
        """loss.py:166-184 (one host round trip: scipy's assignment solver; runs once)"""
        from scipy.optimize import linear_sum_assignment
        preds_small = labels_downscale(preds, image_feats.shape[-2:])
        image_mean = torch.stack([torch.mean(image_feats[:, preds_small == i], dim=1) for i in range(self.num_classes)])
        image_centroids = torch.stack([compute_centroid(preds == i) for i in range(self.num_classes)])
        feat_d = cosine_dists(image_mean, self.style_feats_mean)
        patch_d = torch.linalg.norm(image_centroids[:, None] - self.style_centroids[None], dim=-1)
        cost = np.nan_to_num((feat_d + patch_d).detach().cpu().numpy())
        self.matching = [int(j) for j in linear_sum_assignment(cost)[1]]

    def forward(self, feats1: Dict[str, torch.Tensor], _=None, preds: Optional[torch.Tensor] = None, iter: int = 0) -> torch.Tensor:
        assert self.ready
        image_feat = feats1[self.keys[0]].squeeze(0)
        f1 = image_feat.reshape(image_feat.shape[0], -1).T                 # [(h w), c]
        f2 = self.style_feats.reshape(self.style_feats.shape[0], -1).T
        row_cluster = style_cluster = None
        if self.use_matching:
            if self.matching is None:
                self.update_matching(image_feat, preds)
            preds_small = labels_downscale(preds, image_feat.shape[-2:]).reshape(-1)
            # image position of class i may only match style positions of cluster matching[i]; classes the matching
            # does not cover are unrestricted (loss.py:201-206 loops over range(num_classes) only)
            nc = self.num_classes
            m = torch.full((nc + 1,), -1, dtype=torch.long, device=f1.device)          # slot nc: "unrestricted"
            m[:nc] = torch.as_tensor(self.matching, dtype=torch.long, device=f1.device)[:nc]
            row_cluster = m[torch.where((preds_small < 0) | (preds_small >= nc), torch.full_like(preds_small, nc), preds_small)]
            style_cluster = self.clusters.reshape(-1)
            if self._style_groups is None or self._style_groups[0].device != f1.device:
                self._style_groups = [torch.nonzero(style_cluster.to(f1.device) == k)[:, 0] for k in range(self.n_clusters)]
        return torch.mean(_nn_min_dists(f1, f2, row_cluster, style_cluster, self._style_groups))


def get_style_loss(loss_name: str, keys: Union[List[str], str], **kwargs) -> StyleLoss:
    """loss.py:293-303"""
    ctor = globals()[loss_name]
    assert isinstance(ctor, type) and issubclass(ctor, StyleLoss)
    return ctor([keys] if isinstance(keys, str) else keys, **kwargs)
