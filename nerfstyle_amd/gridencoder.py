"""Host-side mirror of the reference's `gridencoder` package (gridencoder/grid.py): `GridEncoder`
with the same constructor, parameters (`embeddings`, buffer `offsets`) and `forward(inputs,
bound=1, style=0)`; `grid_encode` with the same argument list as grid.py:19-25.

Backed by nsr_grid_encode_forward / nsr_grid_encode_backward.  Differences from the reference
(all documented in include/nsr.h): outputs are written directly as [B, L*C] (no permute copy),
the backward accumulates in fp32 whatever the table type, calc_grad_inputs is unsupported (it is
never requested on this path: positions never require grad).
"""
import ctypes

import numpy as np
import torch
import torch.nn as nn
from torch.autograd import Function

from . import _lib as L

_gridtype_to_id = {'hash': 0, 'tiled': 1}


def _offsets_host(offsets):
    off = offsets.detach().to('cpu', torch.int32).contiguous().numpy()
    return off, off.ctypes.data_as(ctypes.POINTER(ctypes.c_int32))


class _grid_encode(Function):
    @staticmethod
    def forward(ctx, inputs, embeddings, offsets, per_level_scale, base_resolution,
                calc_grad_inputs=False, gridtype=0, align_corners=False, style=0):
        """grid.py:19-66.  inputs [B,3] float in [0,1]; embeddings [rows,C]; offsets [L+1] int
        (any device: it is 17 ints and is read on the host).  Returns [B, L*C]."""
        if calc_grad_inputs:
            raise RuntimeError('grid_encode: calc_grad_inputs is not supported (never used on this path)')
        inputs = inputs.detach().to(torch.float32).contiguous()
        B, D = inputs.shape
        Lv = offsets.shape[0] - 1
        C = embeddings.shape[1]
        S = float(np.float32(np.log2(per_level_scale)))   # grid.py:36; narrowed to float at the binding
        H = int(base_resolution)
        emb = embeddings.detach()
        if torch.is_autocast_enabled() and C % 2 == 0:     # grid.py:42-43
            emb = emb.to(torch.half)
        emb = emb.contiguous()
        off_np, off_p = _offsets_host(offsets)
        outputs = torch.empty(B, Lv * C, device=inputs.device, dtype=emb.dtype)
        L.check(L.lib().nsr_grid_encode_forward(
            L.p(inputs), L.p(emb), L.dt(emb.dtype), off_p, L.p(outputs), B, D, C, Lv, S, H, 0, int(gridtype),
            int(bool(align_corners)), int(style), 1, L.stream()), 'grid_encode_forward')
        ctx.save_for_backward(inputs)
        ctx.off_np = off_np
        ctx.dims = [B, D, C, Lv, S, H, gridtype, emb.shape[0]]
        ctx.align_corners = align_corners
        ctx.style = style
        ctx.emb_dtype = embeddings.dtype
        return outputs

    @staticmethod
    def backward(ctx, grad):
        """grid.py:68-97"""
        (inputs,) = ctx.saved_tensors
        B, D, C, Lv, S, H, gridtype, rows = ctx.dims
        grad = grad.contiguous()
        if grad.dtype not in (torch.float32, torch.float16):
            grad = grad.to(torch.float32)
        grad_embeddings = torch.zeros(rows, C, dtype=torch.float32, device=inputs.device)   # grid.py:82
        off_p = ctx.off_np.ctypes.data_as(ctypes.POINTER(ctypes.c_int32))
        L.check(L.lib().nsr_grid_encode_backward(
            L.p(grad), L.dt(grad.dtype), L.p(inputs), off_p, L.p(grad_embeddings), B, D, C, Lv, S, H, int(gridtype),
            int(bool(ctx.align_corners)), int(ctx.style), 1, L.stream()), 'grid_encode_backward')
        return None, grad_embeddings.to(ctx.emb_dtype), None, None, None, None, None, None, None


grid_encode = _grid_encode.apply


class GridEncoder(nn.Module):
    """grid.py:103-191"""

    def __init__(self, input_dim=3, num_levels=16, level_dim=2, per_level_scale=2, base_resolution=16,
                 log2_hashmap_size=19, desired_resolution=None, gridtype='hash', align_corners=False):
        super().__init__()
        if desired_resolution is not None:
            per_level_scale = np.exp2(np.log2(desired_resolution / base_resolution) / (num_levels - 1))
        self.input_dim = input_dim
        self.num_levels = num_levels
        self.level_dim = level_dim
        self.per_level_scale = per_level_scale
        self.log2_hashmap_size = log2_hashmap_size
        self.base_resolution = base_resolution
        self.output_dim = num_levels * level_dim
        self.gridtype = gridtype
        self.gridtype_id = _gridtype_to_id[gridtype]
        self.align_corners = align_corners
        self.n_output_dims = num_levels * level_dim

        offsets, offset = [], 0
        self.max_params = 2 ** log2_hashmap_size
        for i in range(num_levels):
            resolution = int(np.ceil(base_resolution * per_level_scale ** i))
            params_in_level = min(self.max_params, (resolution if align_corners else resolution + 1) ** input_dim)
            params_in_level = int(np.ceil(params_in_level / 8) * 8)
            offsets.append(offset)
            offset += params_in_level
        offsets.append(offset)
        self.register_buffer('offsets', torch.from_numpy(np.array(offsets, dtype=np.int32)))
        self.n_params = offsets[-1] * level_dim
        self.embeddings = nn.Parameter(torch.empty(offset, level_dim))
        self.reset_parameters()

    def reset_parameters(self):
        std = 1e-4
        self.embeddings.data.uniform_(-std, std)

    def __repr__(self):
        return ('GridEncoder: input_dim={} num_levels={} level_dim={} resolution={} per_level_scale={:.4f} '
                'params={} gridtype={} align_corners={}').format(
            self.input_dim, self.num_levels, self.level_dim, self.base_resolution, self.per_level_scale,
            tuple(self.embeddings.shape), self.gridtype, self.align_corners)

    def forward(self, inputs, bound=1, style=0):
        inputs = (inputs + bound) / (2 * bound)   # grid.py:177
        prefix_shape = list(inputs.shape[:-1])
        inputs = inputs.view(-1, self.input_dim)
        outputs = grid_encode(inputs, self.embeddings, self.offsets, self.per_level_scale, self.base_resolution,
                              False, self.gridtype_id, self.align_corners, style)
        return outputs.view(prefix_shape + [self.output_dim])
