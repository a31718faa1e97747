"""ORACLE (test infrastructure, NOT product code).

Pure-PyTorch CPU restatement of the reference's render step, independent of the C restatement:
  * hash-grid encode with integer tensor ops + gathers (gridencoder.cu:35-80,134-181),
    differentiable w.r.t. the tables (so autograd gives the transpose = kernel_grid_backward);
  * the four bias-free MLPs as fp32 matmuls (style_nerf.py:120-142 wiring, use_dir=False);
  * fixed-K sampling + cumprod integration = the reference's legacy pure-PyTorch path
    (nerf_lib.sample_points :145-176 without jitter, nerf_lib.integrate_points :179-219), which is
    what BASELINE.json config 1 ("64 samples/ray, pure-PyTorch CPU path") refers to;
  * segmented compositing of marched samples (raymarching.cu:806-879 without the early stop).

Uses: (1) cross-check of oracle/liboracle.so, (2) fp32 autograd reference for the fused backward
kernel, (3) bench.py's `cpu_baseline` leg ("port").  Only tests/, smoke() and that leg import it.
Pinned by tests/golden/reference_python.npz (integrate_points, BBox.normalize, trunc_exp).
"""
import math

import numpy as np
import torch

PRIMES = (1, 2654435761, 805459861)


def grid_offsets(num_levels=16, per_level_scale=2.0, base_resolution=16, log2_hashmap_size=19, align_corners=True):
    """grid.py:129-140"""
    offsets, offset = [], 0
    for i in range(num_levels):
        resolution = int(np.ceil(base_resolution * per_level_scale ** i))
        n = min(2 ** log2_hashmap_size, (resolution if align_corners else resolution + 1) ** 3)
        n = int(np.ceil(n / 8) * 8)
        offsets.append(offset)
        offset += n
    offsets.append(offset)
    return np.array(offsets, dtype=np.int32)


def level_resolution(level, S, H):
    """gridencoder.cu:137 in fp32"""
    return int(np.floor(np.exp2(np.float32(level) * np.float32(S), dtype=np.float32) * np.float32(H)))


def _row_index(pg, res, size, gridtype=0, style=0):
    """gridencoder.cu:55-80 for int64 tensors pg [..., 3] (uint32 wrap-around emulated with & mask)."""
    M32 = 0xFFFFFFFF
    stride, index = 1, torch.zeros(pg.shape[:-1], dtype=torch.int64)
    for d in range(3):
        if stride > size:
            break
        index = (index + pg[..., d] * stride) & M32
        stride = (stride * (res + 1)) & M32
    if stride <= size:
        index = (index + style * stride) & M32
        stride = (stride * 512) & M32
    if gridtype == 0 and stride > size:
        index = torch.zeros_like(index)
        for d in range(3):
            index = index ^ ((pg[..., d] * PRIMES[d]) & M32)
        index = index ^ ((style * 3674653429) & M32)
    return index % size


_CORNERS = torch.tensor([[(idx >> d) & 1 for d in range(3)] for idx in range(8)], dtype=torch.int64)   # [8,3]


def grid_encode(x, emb, offsets, per_level_scale, base_resolution=16, align_corners=True, gridtype=0, sparse_grad=False):
    """x [B,3] float32 in [0,1]; emb [rows, C] (requires_grad ok) -> [B, L*C].  One gather of
    [B,8] rows per level (the 8 corners are vectorised; weights multiply in d = 0,1,2 order).
    sparse_grad: table gradients as sparse tensors (no dense [rows, C] zero-fill per level)."""
    S = np.float32(np.log2(per_level_scale))
    L = len(offsets) - 1
    outs = []
    oob = ((x < 0) | (x > 1)).any(dim=1)
    cf = _CORNERS.to(x.dtype)
    for l in range(L):
        res = level_resolution(l, S, base_resolution)
        size = int(offsets[l + 1] - offsets[l])
        scale = float(res - (0 if align_corners else 1))
        pos = x * scale + (0.0 if align_corners else 0.5)
        pg = torch.minimum(torch.floor(pos), torch.tensor(float(res - 1)))
        frac = pos - pg
        pgl = pg.to(torch.int64)[:, None, :] + _CORNERS[None]                       # [B,8,3]
        wd = frac[:, None, :] * cf[None] + (1 - frac[:, None, :]) * (1 - cf[None])   # [B,8,3]
        w = wd[..., 0] * wd[..., 1] * wd[..., 2]
        rows = _row_index(pgl, res, size, gridtype) + int(offsets[l])                # [B,8]
        acc = (w[..., None] * torch.nn.functional.embedding(rows, emb, sparse=sparse_grad)).sum(dim=1)
        acc = torch.where(oob[:, None], torch.zeros_like(acc), acc)
        outs.append(acc)
    return torch.cat(outs, dim=1)


def quant(x, half):
    """Round to f16 / bf16 with a straight-through gradient: emulates the kernels' operand rounding
    so that ReLU masks (and therefore gradients) are comparable bit pattern for bit pattern."""
    if half is None:
        return x
    dt = torch.float16 if half == 'f16' else torch.bfloat16
    return x + (x.detach().to(dt).to(x.dtype) - x.detach())


def mlp(x, params, n_in, n_out, n_hidden_layers=1, n_neurons=64, out_act='none', half=None):
    pad16 = lambda v: (v + 15) // 16 * 16
    d, p = pad16(n_in), 0
    shapes = [(n_neurons, d)] + [(n_neurons, n_neurons)] * (n_hidden_layers - 1) + [(pad16(n_out), n_neurons)]
    a = quant(x, half)
    for li, (o, i) in enumerate(shapes):
        w = quant(params[p:p + o * i].view(o, i), half)
        p += o * i
        a = a @ w.t()
        if li < len(shapes) - 1:
            a = quant(torch.relu(a), half)
    a = a[:, :n_out]
    return torch.sigmoid(a) if out_act == 'sigmoid' else a


class TruncExp(torch.autograd.Function):
    """tcnn_nerf.py:55-69"""
    @staticmethod
    def forward(ctx, x):
        ctx.save_for_backward(x)
        return torch.exp(x)

    @staticmethod
    def backward(ctx, g):
        return g * torch.exp(ctx.saved_tensors[0].clamp(-15, 15))


class Field(torch.nn.Module):
    """style_nerf.py:12-142 with use_dir=False, fp32."""

    def __init__(self, num_classes=5, bound=2.0, seed=80000, table_scale=1e-4, sparse_grad=False, min_res=16):
        super().__init__()
        self.bound, self.nc, self.sparse_grad, self.min_res = bound, num_classes, sparse_grad, min_res
        self.pls = float(np.exp2(np.log2(1024 * (2 * bound) / min_res) / 15))   # tcnn_nerf.py:20-22, bbox size 2*bound
        self.offsets = grid_offsets(16, self.pls, min_res, 19, True)
        g = torch.Generator().manual_seed(seed)
        R = int(self.offsets[-1])
        self.emb_density = torch.nn.Parameter((torch.rand(R, 2, generator=g) * 2 - 1) * table_scale)
        self.emb_color = torch.nn.Parameter((torch.rand(R, 2, generator=g) * 2 - 1) * table_scale)

        def xav(shapes):
            return torch.cat([((torch.rand(o, i, generator=g) * 2 - 1) * math.sqrt(6.0 / (o + i))).reshape(-1)
                              for (o, i) in shapes])
        self.p_density = torch.nn.Parameter(xav([(64, 32), (16, 64)]))
        self.p_color1 = torch.nn.Parameter(xav([(64, 32), (16, 64)]))
        self.p_color2 = torch.nn.Parameter(xav([(64, 16), (64, 64), (16, 64)]))
        self.p_class = torch.nn.Parameter(xav([(64, 32), (16, 64)]))

    def encoder_input(self, pts):
        x = (pts + self.bound) / (2 * self.bound)      # BBox.normalize, common.py:276-288
        return (x + 1) / 2                             # grid.py:177 with bound = 1

    def forward(self, pts, sigma_only=False, half=None, table_half=False):
        """half in {None,'f16','bf16'}: emulate the kernels' operand rounding (straight-through);
        table_half: gather from f16-rounded tables (the AMP copy)."""
        x = self.encoder_input(pts)
        ed = quant(self.emb_density, 'f16') if table_half else self.emb_density
        xd = grid_encode(x, ed, self.offsets, self.pls, base_resolution=self.min_res, sparse_grad=self.sparse_grad)
        logit = mlp(xd, self.p_density, 32, 1, half=half)
        sigmas = TruncExp.apply(logit)
        if sigma_only:
            return sigmas
        ec = quant(self.emb_color, 'f16') if table_half else self.emb_color
        xc = grid_encode(x, ec, self.offsets, self.pls, base_resolution=self.min_res, sparse_grad=self.sparse_grad)
        classes = mlp(xc, self.p_class, 32, self.nc, half=half)
        c1 = mlp(xc, self.p_color1, 32, 16, half=half)
        rgb = mlp(c1, self.p_color2, 16, 3, n_hidden_layers=2, out_act='sigmoid', half=half)
        return torch.cat((rgb, classes), dim=1), sigmas


def sample_points(rays_o, rays_d, near, far, num_samples):
    """nerf_lib.py:145-176 with the stratified jitter at its midpoint (t_rand = 0.5) so that the
    baseline is deterministic; RayBatch.lerp's 2-D case is restated with origins[:, None, :]
    (the reference's own broadcast there is dead code that raises, common.py:172)."""
    z = torch.linspace(near, far, steps=num_samples + 1)
    z = z.expand(rays_o.shape[0], num_samples + 1)
    lower, upper = z[:, :-1], z[:, 1:]
    z_vals = lower + (upper - lower) * 0.5
    pts = rays_o[:, None, :] + rays_d[:, None, :] * z_vals[..., None]
    dists = z_vals[..., 1:] - z_vals[..., :-1]
    dists = torch.cat([dists, torch.full((dists.shape[0], 1), 1e10)], dim=-1)
    return pts, dists


def integrate_points(dists, rgbs, densities, prev_rgb, prev_acc, prev_trans):
    """nerf_lib.py:179-219 (density2alpha = 1 - exp(-relu(d) * dist), utils/__init__.py:352-353)"""
    alpha = 1.0 - torch.exp(-torch.relu(densities) * dists)
    alpha_tmp = torch.cat([prev_trans, (1. - alpha[:, :-1])], dim=-1)
    trans = torch.cumprod(alpha_tmp, dim=-1)
    weights = alpha * trans
    rgb_map = prev_rgb + (weights[..., None] * rgbs).sum(dim=1)
    acc_map = prev_acc + weights.sum(dim=1, keepdim=True)
    trans_map = (trans[:, -1] * (1. - alpha[:, -1]))[:, None]
    return rgb_map, acc_map, trans_map


def render_fixed_k(field, rays_o, rays_d, near=0.2, far=4.0, num_samples=64):
    """The pure-PyTorch render the reference had before the CUDA extensions (BASELINE config 1)."""
    pts, dists = sample_points(rays_o, rays_d, near, far, num_samples)
    N, K = pts.shape[:2]
    pts = pts.reshape(-1, 3).clamp(-field.bound, field.bound)
    out, sigmas = field(pts)
    rgbs = out.view(N, K, -1)
    C = rgbs.shape[-1]
    rgb_map, acc_map, _ = integrate_points(dists, rgbs, sigmas.view(N, K), torch.zeros(N, C), torch.zeros(N, 1),
                                           torch.ones(N, 1))
    image = rgb_map[:, :3] + (1 - acc_map)
    return image, rgb_map[:, 3:]


def composite_marched(sigmas, rgbs, deltas, rays, T_thresh=0.0):
    """raymarching.cu:806-879 as a per-ray loop of vector ops (small inputs only)."""
    N, C = rays.shape[0], rgbs.shape[1]
    ws = torch.zeros(N)
    depth = torch.zeros(N)
    image = torch.zeros(N, C)
    for n in range(N):
        idx, off, cnt = int(rays[n, 0]), int(rays[n, 1]), int(rays[n, 2])
        if cnt == 0 or off + cnt >= sigmas.shape[0]:
            continue
        s, c, d = sigmas[off:off + cnt], rgbs[off:off + cnt], deltas[off:off + cnt]
        alpha = 1 - torch.exp(-s * d[:, 0])
        T = torch.cumprod(torch.cat([torch.ones(1), 1 - alpha[:-1]]), 0)
        T_after = T * (1 - alpha)
        keep = torch.ones(cnt, dtype=torch.bool)
        below = (T_after < T_thresh).nonzero()
        if len(below) > 0:
            keep[int(below[0]) + 1:] = False          # the crossing sample is still accumulated (:862)
        w = alpha * T * keep
        t = torch.cumsum(d[:, 1], 0)
        ws[idx] = w.sum()
        depth[idx] = (w * t).sum()
        image[idx] = (w[:, None] * c).sum(0)
    return ws, depth, image
