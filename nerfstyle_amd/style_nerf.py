"""Model wiring of the hot path: the counterpart of the reference's networks/style_nerf.py
(`StyleTCNerf`) and networks/tcnn_nerf.py (`get_grid_encoder`, `trunc_exp`).

`StyleTCNerf(cfg, bbox, class_dim, enc_dtype, use_dir=False)` keeps the reference constructor and
`forward(pts, dirs=None)` contract (sigma [M,1] when dirs is None, else (rgbs|classes [M,3+nc],
sigma [M,1]); style_nerf.py:120-159) and the reference's state_dict keys
(`x_density_embedder.{embeddings,offsets}`, `x_color_embedder.{...}`, `{density,color1,color2,
class}_net.params`), but the storage is MI355X-first:

  * ONE flat fp32 parameter arena [rows*4 + 15360] = the two hash tables INTERLEAVED as
    tables[row][enc][feat] followed by the four MLP parameter vectors, and one gradient arena of
    the same shape.  One 16-byte gather / one atomic request serves both encoders; the optimiser
    is one streaming pass; the multi-GPU gradient all-reduce is one flat bucket.
  * the whole field is evaluated by two fused launches (nsr_field_forward / nsr_field_backward);
    the backward recomputes the forward instead of saving [M,.] activations and accumulates
    straight into the gradient arena.
"""
import ctypes
import math

import numpy as np
import torch
import torch.nn as nn
from torch.autograd import Function

from . import _lib as L
from . import profiling
from .common import BBox
from .config import NetworkConfig
from .gridencoder import GridEncoder
from .network import init_mlp_params

MLP_PARAMS = 15360
# (name, offset in the MLP block, length); order fixed by include/nsr.h
MLP_LAYOUT = (('density_net', 0, 3072), ('color1_net', 3072, 3072), ('color2_net', 6144, 6144),
              ('class_net', 12288, 3072))


class _trunc_exp(Function):
    """tcnn_nerf.py:55-69"""
    @staticmethod
    def forward(ctx, x):
        x = x.to(torch.float32)
        ctx.save_for_backward(x)
        return torch.exp(x)

    @staticmethod
    def backward(ctx, g):
        x = ctx.saved_tensors[0]
        return g * torch.exp(x.clamp(-15, 15))


trunc_exp = _trunc_exp.apply


def per_level_scale_from_cfg(cfg: NetworkConfig, max_bound: float) -> float:
    """tcnn_nerf.py:20-22"""
    pe = cfg.pos_enc
    max_res = pe.max_res_coeff * max_bound
    return float(np.exp2(np.log2(max_res / pe.min_res) / (pe.n_lvls - 1)))


def get_grid_encoder(cfg: NetworkConfig, max_bound: float, enc_dtype=None, use_custom_impl=True):
    """tcnn_nerf.py:14-52 (only the custom torch-ngp-style encoder exists here)."""
    pe = cfg.pos_enc
    return GridEncoder(input_dim=3, num_levels=pe.n_lvls, level_dim=pe.n_feats_per_lvl,
                       per_level_scale=per_level_scale_from_cfg(cfg, max_bound), base_resolution=pe.min_res,
                       log2_hashmap_size=pe.hashmap_size, gridtype='hash', align_corners=True)


class _field(Function):
    """Fused field: xyzs -> (sigmas [M], rgbs [M, 3+nc] | None).  `arena` is an input only so that
    autograd routes gradients here; the backward accumulates in place into model.grad_arena (which
    is arena.grad) and returns None for it."""

    @staticmethod
    def forward(ctx, xyzs, arena, model, sigma_only, m_dev, density_scale, want_feats=False, perm=None):
        xyzs = xyzs.detach().to(torch.float32).contiguous()
        M = xyzs.shape[0]
        dev = xyzs.device
        sigmas = torch.empty(M, dtype=torch.float32, device=dev)
        rgbs = None if sigma_only else torch.empty(M, model.out_channels, dtype=torch.float32, device=dev)
        desc = model._desc(density_scale)
        tables = model._gather_tables()
        # the one thing a training forward saves: the encoded features as MFMA fragments (128 B/sample)
        feats = None
        # (decided by the caller: grad mode is always off inside Function.forward)
        if (not sigma_only) and want_feats:
            feats = torch.empty(((M + 15) // 16) * 512, dtype=torch.int32, device=dev)
        with profiling.timed('field_fwd_sigma' if sigma_only else 'field_fwd'):
            L.check(L.lib().nsr_field_forward(ctypes.byref(desc), L.p(tables), L.p(model._mlp_flat()),
                                              L.p(xyzs), M, L.p(m_dev), L.p(sigmas), L.p(rgbs), L.p(feats),
                                              L.p(perm) if feats is not None else None, L.stream()),
                    'field_forward')
        ctx.model = model
        ctx.m_dev = m_dev
        ctx.density_scale = density_scale
        ctx.sigma_only = sigma_only
        # feats (15.6 GB at a full-frame capacity) and perm go through save_for_backward, NOT ctx attributes: autograd
        # drops saved tensors when backward() has run, attributes live as long as anything still holds the loss
        ctx.save_for_backward(xyzs, feats, perm)
        if sigma_only:
            return sigmas
        return sigmas, rgbs

    @staticmethod
    def backward(ctx, grad_sigmas, grad_rgbs=None):
        model = ctx.model
        xyzs, feats, perm = ctx.saved_tensors
        M = xyzs.shape[0]
        dev = xyzs.device
        if grad_sigmas is None:
            grad_sigmas = torch.zeros(M, dtype=torch.float32, device=dev)
        if grad_rgbs is None:
            grad_rgbs = torch.zeros(M, model.out_channels, dtype=torch.float32, device=dev)
        grad_sigmas = grad_sigmas.to(torch.float32).contiguous()
        grad_rgbs = grad_rgbs.to(torch.float32).contiguous()
        model._ensure_grad()
        desc = model._desc(ctx.density_scale)
        ga = model.grad_arena
        tables = model._gather_tables()
        ws = None
        if perm is not None:
            # [M][16] float4 of per-level encoder gradients for the spatially ordered table scatter (second kernel): 256 B
            # per sample SLOT (31 GB for a 122 M-slot capacity buffer), kept on the model and only ever grown -- the
            # caching allocator would otherwise be asked for the largest block of the step every step
            need = int(L.lib().nsr_field_backward_workspace_bytes(M, 1)) // 4
            if torch.cuda.is_current_stream_capturing():
                ws = torch.empty(need, dtype=torch.float32, device=dev)      # graph-owned: a captured pointer must never dangle
            else:
                ws = getattr(model, '_bwd_ws', None)
                if ws is None or ws.device != dev or ws.numel() < need:
                    model._bwd_ws = None
                    ws = model._bwd_ws = torch.empty(need, dtype=torch.float32, device=dev)
        with profiling.timed('field_bwd'):
            def call(perm, wsp):
                return L.lib().nsr_field_backward(
                    ctypes.byref(desc), L.p(tables), L.p(model._mlp_flat()), L.p(xyzs), M, L.p(ctx.m_dev),
                    L.p(grad_sigmas), L.p(grad_rgbs), L.p(ga), (ga.data_ptr() + model.table_elems * 4) if model.train_mlps else None,
                    int(model.train_density_table), int(model.train_color_table), L.p(feats), L.p(perm), L.p(wsp),
                    L.stream())
            st = call(perm, ws)
            if st == -2 and perm is not None:
                feats = None             # saved in the permutation's order: useless to a kernel that walks the buffers
                # NSR_ERR_UNSUPPORTED: a grid finer than the spatial scatter's LDS lattices hold (finest resolution above
                # ~5 cells per 1/1024 block) -- nothing was launched; the fused run-tracker backward takes over
                model._spatial_scatter_unsupported = True
                st = call(None, None)
            L.check(st, 'field_backward')
        return None, None, None, None, None, None, None, None


class _EncoderView(nn.Module):
    """Reference-shaped handle on one of the two interleaved tables (names used by checkpoints
    and by the trainers' keyword filters, e.g. OPTIM_KEYS = ['x_color_embedder'], style.py:25)."""

    def __init__(self, owner, enc_index, template: GridEncoder):
        super().__init__()
        self.__dict__['_owner'] = owner
        self.enc_index = enc_index
        for k in ('input_dim', 'num_levels', 'level_dim', 'per_level_scale', 'log2_hashmap_size', 'base_resolution',
                  'output_dim', 'gridtype', 'gridtype_id', 'align_corners', 'n_output_dims'):
            setattr(self, k, getattr(template, k))
        self.register_buffer('offsets', template.offsets.clone())

    @property
    def embeddings(self):
        """[rows, 2] strided view into the arena (not a copy)."""
        return self._owner.tables_view()[:, self.enc_index, :]


class _NetView(nn.Module):
    def __init__(self, owner, name, off, n, n_in, n_out):
        super().__init__()
        self.__dict__['_owner'] = owner
        self.n_input_dims, self.n_output_dims = n_in, n_out
        self._off, self._n = off, n

    @property
    def params(self):
        o = self._owner
        return o.arena.detach()[o.table_elems + self._off: o.table_elems + self._off + self._n]


class StyleTCNerf(nn.Module):
    def __init__(self, cfg: NetworkConfig, bbox: BBox, class_dim: int, enc_dtype=None, use_dir: bool = False,
                 compute_dtype=torch.float16, device=None):
        """enc_dtype: None -> half gather tables (the reference's AMP behaviour, trainers/base.py:148
        + grid.py:42-43); torch.float32 -> fp32 tables."""
        super().__init__()
        if use_dir:
            raise NotImplementedError('use_dir=True (SH direction encoding) is off on this path; both reference entry '
                                      'points pass use_dir=False (trainers/base.py:149-151, render.py:68-69)')
        assert cfg.density_hidden_dims == 64 and cfg.rgb_hidden_dims == 64
        assert cfg.density_hidden_layers == 1 and cfg.rgb_hidden_layers == 2
        assert cfg.pos_enc.n_lvls == 16 and cfg.pos_enc.n_feats_per_lvl == 2
        self.cfg = cfg
        self.bounds_bbox = bbox
        self.use_dir = False
        self.class_dim = class_dim
        self.out_channels = 3 + class_dim
        self.compute_dtype = compute_dtype
        self.table_dtype = torch.float16 if enc_dtype in (None, torch.float16) else torch.float32
        self.train_density_table = True
        self.train_color_table = True
        self.train_mlps = True                   # False (set by an optimiser that trains no net): the backward skips the weight gradients
        self.save_features = True     # forward keeps 128 B/sample of encoded features for the backward

        max_bound = torch.max(bbox.size).item()
        template = get_grid_encoder(cfg, max_bound)
        self.per_level_scale = template.per_level_scale
        self.S = float(np.float32(np.log2(self.per_level_scale)))
        self._offsets_np = template.offsets.numpy().astype(np.int32).copy()
        self.rows = int(self._offsets_np[-1])
        self.table_elems = self.rows * 4

        # ---- the arena -------------------------------------------------------------------------
        flat = torch.empty(self.table_elems + MLP_PARAMS, dtype=torch.float32)
        g = torch.Generator().manual_seed(int(cfg.network_seed or 0))
        flat[:self.table_elems].uniform_(-1e-4, 1e-4, generator=g)   # grid.py:150-152
        seed = int(cfg.network_seed or 0)
        mlp = torch.cat([
            init_mlp_params(32, 1, 64, 1, seed), init_mlp_params(32, 16, 64, 1, seed + 1),
            init_mlp_params(16, 3, 64, 2, seed + 2), init_mlp_params(32, class_dim, 64, 1, seed + 3)])
        flat[self.table_elems:] = mlp
        self.arena = nn.Parameter(flat)
        self.grad_arena = None
        self._half_tables = None
        self._half_version = -1

        self.x_density_embedder = _EncoderView(self, 0, template)
        self.x_color_embedder = _EncoderView(self, 1, template)
        self.density_net = _NetView(self, 'density_net', 0, 3072, 32, 1)
        self.color1_net = _NetView(self, 'color1_net', 3072, 3072, 32, 16)
        self.color2_net = _NetView(self, 'color2_net', 6144, 6144, 16, 3)
        self.class_net = _NetView(self, 'class_net', 12288, 3072, 32, class_dim)
        if device is not None:
            self.to(device)

    # ---- storage helpers -----------------------------------------------------------------------
    @property
    def device(self):
        return self.arena.device

    def _apply(self, fn, *args, **kwargs):
        super()._apply(fn, *args, **kwargs)
        self.bounds_bbox.to(self.arena.device)
        self.grad_arena = None
        self._half_tables = None
        self._half_version = -1
        return self

    def tables_view(self):
        return self.arena.detach()[:self.table_elems].view(self.rows, 2, 2)

    def _mlp_flat(self):
        return self.arena.detach()[self.table_elems:]

    def _ensure_grad(self):
        if self.grad_arena is None or self.grad_arena.device != self.arena.device:
            self.grad_arena = torch.zeros_like(self.arena.detach())
        if self.arena.grad is None:
            self.grad_arena.zero_()
            self.arena.grad = self.grad_arena
        elif self.arena.grad.data_ptr() != self.grad_arena.data_ptr():
            self.grad_arena = self.arena.grad
        return self.grad_arena

    def half_tables(self):
        """f16 gather copy of the tables; storage is stable so fused optimisers can write into it."""
        if self._half_tables is None or self._half_tables.device != self.arena.device:
            self._half_tables = torch.empty(self.table_elems, dtype=torch.float16, device=self.arena.device)
            self._half_version = -1
        return self._half_tables

    def mark_half_synced(self):
        self._half_version = self.arena._version

    def _gather_tables(self):
        if self.table_dtype == torch.float32:
            return self.arena.detach()
        h = self.half_tables()
        if self._half_version != self.arena._version:
            L.check(L.lib().nsr_cast_f32_to_f16(L.p(self.arena.detach()), L.p(h), self.table_elems, L.stream()),
                    'cast_f32_to_f16')
            self._half_version = self.arena._version
        return h

    def _desc(self, density_scale=1.0):
        d = L.FieldDesc()
        d.L, d.H, d.S = 16, int(self.cfg.pos_enc.min_res), self.S
        d.num_classes = self.class_dim
        d.table_dtype = L.dt(self.table_dtype)
        d.compute_dtype = L.dt(self.compute_dtype)
        # host copy of the (constant) bounding box, read back once: a per-call .cpu() would be a device
        # synchronisation in the middle of an otherwise sync-free step
        if getattr(self, '_bbox_host', None) is None:
            self._bbox_host = (self.bounds_bbox.min_pt.detach().cpu().tolist(), self.bounds_bbox.size.detach().cpu().tolist())
        mn, sz = self._bbox_host
        for i in range(3):
            d.bbox_min[i] = mn[i]
            d.bbox_size[i] = sz[i]
        d.density_scale = float(density_scale)
        d.offsets = self._offsets_np.ctypes.data_as(ctypes.POINTER(ctypes.c_int32))
        return d

    # ---- reference-shaped checkpoints ----------------------------------------------------------
    def state_dict(self, *args, **kwargs):
        t = self.tables_view()
        sd = {
            'x_density_embedder.embeddings': t[:, 0, :].clone(),
            'x_density_embedder.offsets': self.x_density_embedder.offsets.clone(),
            'x_color_embedder.embeddings': t[:, 1, :].clone(),
            'x_color_embedder.offsets': self.x_color_embedder.offsets.clone(),
        }
        m = self._mlp_flat()
        for name, off, n in MLP_LAYOUT:
            sd[name + '.params'] = m[off:off + n].clone()
        return sd

    def load_state_dict(self, sd, strict=True):
        with torch.no_grad():
            t = self.tables_view()
            t[:, 0, :].copy_(sd['x_density_embedder.embeddings'])
            t[:, 1, :].copy_(sd['x_color_embedder.embeddings'])
            m = self._mlp_flat()
            for name, off, n in MLP_LAYOUT:
                m[off:off + n].copy_(sd[name + '.params'].reshape(-1))
            self.arena.add_(0)   # bump the version counter: the half copy is stale
        return self

    def named_views(self):
        """(name, arena slice or strided view) pairs in the reference's named_parameters() order."""
        t = self.tables_view()
        out = [('x_density_embedder.embeddings', t[:, 0, :]), ('x_color_embedder.embeddings', t[:, 1, :])]
        m = self._mlp_flat()
        for name, off, n in MLP_LAYOUT:
            out.append((name + '.params', m[off:off + n]))
        return out

    # ---- forward -------------------------------------------------------------------------------
    def field(self, pts, sigma_only=False, m_dev=None, density_scale=1.0, perm=None):
        """Fast path: flat sigmas [M] (and rgbs [M,3+nc]); m_dev = device int32 sample count; perm = spatial order from
        `sample_order` (int32 [M] device tensor): a training forward then walks the samples in that order (its saved features
        are tile-major in it) and the backward follows -- MLP chain first, then the table gradient with the stand-alone
        lattice scatter kernel (same result per sample, table gradient up to fp32 summation order; pays on dense batches)."""
        want_feats = bool(self.save_features and torch.is_grad_enabled() and self.arena.requires_grad and not sigma_only)
        return _field.apply(pts, self.arena, self, sigma_only, m_dev, density_scale, want_feats, perm)

    def sample_order(self, xyzs, m_dev=None, sort_prefix=None, out=None):
        """nsr_sample_order: Morton-order permutation of the samples [M,3] (int32 tensor [M], a uint32 bit pattern;
        written into `out` when given)."""
        M = xyzs.shape[0]
        dev = xyzs.device
        need = (int(L.lib().nsr_sample_order_workspace_bytes(M)) + 3) // 4 + 64
        # a temporary of torch's stream-ordered caching allocator: no hipMalloc in the steady state, correct on any stream,
        # and under capture it belongs to the graph's own pool (a captured launch bakes its pointers in)
        ws = torch.empty(need, dtype=torch.int32, device=dev)
        ws_ptr = (ws.data_ptr() + 255) & ~255
        perm = torch.empty(M, dtype=torch.int32, device=dev) if out is None else out
        assert perm.dtype == torch.int32 and perm.numel() == M and perm.is_contiguous() and perm.device == dev
        if getattr(self, '_bbox_host', None) is None:
            self._desc()
        mn, sz = self._bbox_host
        c3 = ctypes.c_float * 3
        L.check(L.lib().nsr_sample_order(L.p(xyzs), M, L.p(m_dev), M if sort_prefix is None else int(min(sort_prefix, M)),
                                         c3(*mn), c3(*sz), L.p(perm), ws_ptr, L.stream()), 'sample_order')
        return perm

    def forward(self, pts, dirs=None, bsize=1000000):
        """style_nerf.py:144-159.  The >1M-point chunking of the reference (utils.batch_exec) is
        unnecessary here: the fused kernel keeps no per-sample intermediates in HBM."""
        pts = pts.reshape(-1, 3)
        if dirs is None:
            return self.field(pts, sigma_only=True).unsqueeze(-1)
        sigmas, rgbs = self.field(pts, sigma_only=False)
        return rgbs, sigmas.unsqueeze(-1)
