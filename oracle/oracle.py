"""ORACLE (test infrastructure, NOT product code).

numpy-facing bindings of oracle/liboracle.so (the C restatement of the reference's
raymarching.cu / gridencoder.cu arithmetic) plus numpy restatements of the pieces of the
path that are Python in the reference (model wiring, MLP contract, epilogue).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this
module.  Nothing under nerfstyle_amd/ does.

Parity status: see the headers of the .c files ("parity unpinned" against a reference
execution; pinned by known-answer properties + the importable-Python goldens of
tests/golden/).
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, 'liboracle.so')
_lib = None

c_f = ctypes.POINTER(ctypes.c_float)
c_i = ctypes.POINTER(ctypes.c_int32)
c_u = ctypes.POINTER(ctypes.c_uint32)
c_b = ctypes.POINTER(ctypes.c_uint8)


def build(force=False):
    """Compile oracle/liboracle.so with gcc (building the checker is not using it)."""
    srcs = [os.path.join(_HERE, f) for f in ('raymarching_oracle.c', 'gridencoder_oracle.c')]
    if not force and os.path.exists(_LIB_PATH):
        if all(os.path.getmtime(_LIB_PATH) >= os.path.getmtime(s) for s in srcs):
            return _LIB_PATH
    subprocess.check_call(['make', '-C', _HERE, '-B', 'liboracle.so'], stdout=subprocess.DEVNULL)
    return _LIB_PATH


def lib():
    global _lib
    if _lib is None:
        build()
        _lib = ctypes.CDLL(_LIB_PATH)
        _lib.ora_grid_resolution.restype = ctypes.c_uint32
        _lib.ora_grid_resolution.argtypes = [ctypes.c_uint32, ctypes.c_float, ctypes.c_uint32]
    return _lib


def _f(a):
    a = np.ascontiguousarray(a, dtype=np.float32)
    return a, a.ctypes.data_as(c_f)


def _i(a):
    a = np.ascontiguousarray(a, dtype=np.int32)
    return a, a.ctypes.data_as(c_i)


def _u8(a):
    a = np.ascontiguousarray(a, dtype=np.uint8)
    return a, a.ctypes.data_as(c_b)


U = ctypes.c_uint32
F = ctypes.c_float
I = ctypes.c_int


# ---------------------------------------------------------------------------------------------
# raymarching (C restatement)
# ---------------------------------------------------------------------------------------------

def near_far_from_aabb(rays_o, rays_d, aabb, min_near=0.2):
    """raymarching.py:19-52 / raymarching.cu:190-255"""
    o, po = _f(rays_o)
    d, pd = _f(rays_d)
    a, pa = _f(aabb)
    N = o.shape[0]
    nears = np.empty(N, np.float32)
    fars = np.empty(N, np.float32)
    lib().ora_near_far_from_aabb(po, pd, pa, U(N), F(min_near), nears.ctypes.data_as(c_f),
                                 fars.ctypes.data_as(c_f))
    return nears, fars


def morton3D(coords):
    """raymarching.cu:313-331"""
    c, pc = _i(coords)
    N = c.shape[0]
    out = np.empty(N, np.int32)
    lib().ora_morton3D(pc, U(N), out.ctypes.data_as(c_i))
    return out


def morton3D_invert(indices):
    """raymarching.cu:336-359"""
    ind, pi = _i(indices)
    N = ind.shape[0]
    out = np.empty((N, 3), np.int32)
    lib().ora_morton3D_invert(pi, U(N), out.ctypes.data_as(c_i))
    return out


def packbits(grid, thresh):
    """raymarching.py:139-167 / raymarching.cu:366-399"""
    g, pg = _f(grid)
    N = g.size // 8
    out = np.empty(N, np.uint8)
    lib().ora_packbits(pg, U(N), F(thresh), out.ctypes.data_as(c_b))
    return out


def march_rays_train(rays_o, rays_d, bound, bitfield, C, H, nears, fars, max_steps=1024,
                     dt_gamma=0.0, M=None, align=-1, counter=None, want_dirs=True):
    """raymarching.py:174-288 (force_all_rays semantics: M = N * max_steps, perturb disabled
    :247, slice to the emitted count padded to `align` :275-281) / raymarching.cu:410-599."""
    o, po = _f(rays_o)
    d, pd = _f(rays_d)
    g, pg = _u8(bitfield)
    ne, pne = _f(nears)
    fa, pfa = _f(fars)
    N = o.shape[0]
    if M is None:
        M = N * max_steps
    xyzs = np.zeros((M, 3), np.float32)
    dirs = np.zeros((M, 3), np.float32)
    deltas = np.zeros((M, 4), np.float32)
    rays = np.empty((N, 3), np.int32)
    if counter is None:
        counter = np.zeros(2, np.int32)
    noises = np.zeros(N, np.float32)
    lib().ora_march_rays_train(po, pd, None, pg, F(bound), F(dt_gamma), U(max_steps), I(0), U(N), U(C),
                               U(H), U(M), pne, pfa, xyzs.ctypes.data_as(c_f),
                               dirs.ctypes.data_as(c_f) if want_dirs else None,
                               deltas.ctypes.data_as(c_f), rays.ctypes.data_as(c_i),
                               counter.ctypes.data_as(c_i), noises.ctypes.data_as(c_f))
    m = int(counter[0])
    if align > 0:
        m += align - m % align
    m = min(m, M)
    return xyzs[:m], dirs[:m], deltas[:m], rays, counter


def composite_rays_train_forward(sigmas, rgbs, deltas, rays, T_thresh=1e-4):
    """raymarching.py:291-325 / raymarching.cu:806-890"""
    s, ps = _f(sigmas)
    r, pr = _f(rgbs)
    dl, pdl = _f(deltas)
    ry, pry = _i(rays)
    M, N, C = s.shape[0], ry.shape[0], r.shape[1]
    ws = np.empty(N, np.float32)
    depth = np.empty(N, np.float32)
    image = np.empty((N, C), np.float32)
    lib().ora_composite_rays_train_forward(ps, pr, pdl, pry, U(M), U(N), U(C), F(T_thresh), I(0),
                                           ws.ctypes.data_as(c_f), depth.ctypes.data_as(c_f),
                                           image.ctypes.data_as(c_f))
    return ws, depth, image


def composite_rays_train_backward(grad_ws, grad_image, sigmas, rgbs, deltas, rays, ws, image,
                                  T_thresh=1e-4):
    """raymarching.py:327-347 / raymarching.cu:904-997"""
    gws, pgws = _f(grad_ws)
    gim, pgim = _f(grad_image)
    s, ps = _f(sigmas)
    r, pr = _f(rgbs)
    dl, pdl = _f(deltas)
    ry, pry = _i(rays)
    w, pw = _f(ws)
    im, pim = _f(image)
    M, N, C = s.shape[0], ry.shape[0], r.shape[1]
    gs = np.zeros(M, np.float32)
    gr = np.zeros((M, C), np.float32)
    buf = np.zeros((N, C), np.float32)
    lib().ora_composite_rays_train_backward(pgws, pgim, ps, pr, pdl, pry, I(0), pw, pim, U(M), U(N), U(C),
                                            F(T_thresh), gs.ctypes.data_as(c_f), gr.ctypes.data_as(c_f),
                                            buf.ctypes.data_as(c_f))
    return gs, gr


def march_rays(n_alive, n_step, rays_alive, rays_t, rays_o, rays_d, bound, bitfield, C, H, nears, fars,
               align=-1, max_steps=1024, dt_gamma=0.0):
    """raymarching.py:357-424 / raymarching.cu:1004-1130"""
    ra, pra = _i(rays_alive)
    rt, prt = _f(rays_t)
    o, po = _f(rays_o)
    d, pd = _f(rays_d)
    g, pg = _u8(bitfield)
    ne, pne = _f(nears)
    fa, pfa = _f(fars)
    M = n_alive * n_step
    if align > 0:
        M += align - (M % align)
    xyzs = np.zeros((M, 3), np.float32)
    dirs = np.zeros((M, 3), np.float32)
    deltas = np.zeros((M, 4), np.float32)
    noises = np.zeros(n_alive, np.float32)
    lib().ora_march_rays(U(n_alive), U(n_step), pra, prt, po, pd, None, F(bound), F(dt_gamma), U(max_steps),
                         I(0), U(C), U(H), pg, pne, pfa, xyzs.ctypes.data_as(c_f),
                         dirs.ctypes.data_as(c_f), deltas.ctypes.data_as(c_f), noises.ctypes.data_as(c_f))
    return xyzs, dirs, deltas


def composite_rays(n_alive, n_step, rays_alive, rays_t, sigmas, rgbs, deltas, weights_sum, depth, image,
                   T_thresh=1e-2):
    """raymarching.py:430-459 / raymarching.cu:1133-1240.  In-place on rays_alive, rays_t,
    weights_sum, depth, image (must be C-contiguous arrays of the right dtype)."""
    assert rays_alive.dtype == np.int32 and rays_alive.flags.c_contiguous
    for a in (rays_t, weights_sum, depth, image):
        assert a.dtype == np.float32 and a.flags.c_contiguous
    s, ps = _f(sigmas)
    r, pr = _f(rgbs)
    dl, pdl = _f(deltas)
    C = r.shape[-1]
    lib().ora_composite_rays(U(n_alive), U(n_step), F(T_thresh), rays_alive.ctypes.data_as(c_i),
                             rays_t.ctypes.data_as(c_f), ps, pr, pdl, U(C), I(0),
                             weights_sum.ctypes.data_as(c_f), depth.ctypes.data_as(c_f),
                             image.ctypes.data_as(c_f))


# ---------------------------------------------------------------------------------------------
# gridencoder (C restatement + the Python-side sizing of grid.py)
# ---------------------------------------------------------------------------------------------

def grid_offsets(num_levels=16, per_level_scale=2.0, base_resolution=16, log2_hashmap_size=19,
                 align_corners=False, input_dim=3):
    """gridencoder/grid.py:129-140"""
    offsets, offset = [], 0
    max_params = 2 ** log2_hashmap_size
    for i in range(num_levels):
        resolution = int(np.ceil(base_resolution * per_level_scale ** i))
        params_in_level = min(max_params, (resolution if align_corners else resolution + 1) ** input_dim)
        params_in_level = int(np.ceil(params_in_level / 8) * 8)
        offsets.append(offset)
        offset += params_in_level
    offsets.append(offset)
    return np.array(offsets, dtype=np.int32)


def per_level_scale_from_cfg(max_res_coeff=1024, max_bound=4.0, min_res=16, n_lvls=16):
    """networks/tcnn_nerf.py:20-22"""
    max_res = max_res_coeff * max_bound
    return float(np.exp2(np.log2(max_res / min_res) / (n_lvls - 1)))


def grid_S(per_level_scale):
    """grid.py:36 (np.log2 in double, narrowed to float at the binding, gridencoder.h:12)"""
    return float(np.float32(np.log2(per_level_scale)))


def grid_resolutions(L, S, H):
    return np.array([lib().ora_grid_resolution(l, F(S), H) for l in range(L)], dtype=np.uint32)


def round_f16(a):
    a, pa = _f(a)
    out = np.empty_like(a)
    lib().ora_round_f16_array(pa, out.ctypes.data_as(c_f), ctypes.c_uint64(a.size))
    return out


def grid_encode_forward(inputs, embeddings, offsets, per_level_scale, base_resolution, gridtype=0,
                        align_corners=False, style=0, half_accum=False):
    """gridencoder/grid.py:19-66 -> [B, L*C] (the permuted layout the Python wrapper returns)."""
    x, px = _f(inputs)
    e, pe = _f(embeddings)
    off, poff = _i(offsets)
    B, L, C = x.shape[0], off.shape[0] - 1, e.shape[1]
    out = np.empty((L, B, C), np.float32)
    lib().ora_grid_encode_forward(px, pe, poff, out.ctypes.data_as(c_f), U(B), U(C), U(L),
                                  F(grid_S(per_level_scale)), U(base_resolution), U(gridtype),
                                  I(int(align_corners)), U(style), I(int(half_accum)))
    return np.ascontiguousarray(out.transpose(1, 0, 2).reshape(B, L * C))


def grid_encode_backward(grad, inputs, offsets, n_rows, C, per_level_scale, base_resolution, gridtype=0,
                         align_corners=False, style=0):
    """gridencoder/grid.py:68-97: grad [B, L*C] -> grad_embeddings [n_rows, C]"""
    x, px = _f(inputs)
    off, poff = _i(offsets)
    B, L = x.shape[0], off.shape[0] - 1
    g = np.ascontiguousarray(np.asarray(grad, np.float32).reshape(B, L, C).transpose(1, 0, 2))
    ge = np.zeros((n_rows, C), np.float32)
    lib().ora_grid_encode_backward(g.ctypes.data_as(c_f), px, poff, ge.ctypes.data_as(c_f), U(B), U(C), U(L),
                                   F(grid_S(per_level_scale)), U(base_resolution), U(gridtype),
                                   I(int(align_corners)), U(style))
    return ge


def grid_corner_rows(inputs, offsets, per_level_scale, base_resolution, gridtype=0, align_corners=False,
                     style=0):
    x, px = _f(inputs)
    off, poff = _i(offsets)
    B, L = x.shape[0], off.shape[0] - 1
    rows = np.empty((L, B, 8), np.uint32)
    lib().ora_grid_corner_rows(px, poff, rows.ctypes.data_as(c_u), U(B), U(L), F(grid_S(per_level_scale)),
                               U(base_resolution), U(gridtype), I(int(align_corners)), U(style))
    return rows


# ---------------------------------------------------------------------------------------------
# MLP contract (replaces tcnn.Network; tinycudann is an un-vendored, unpinned third-party
# dependency -- README.md:26 -- so its numerics are "parity unpinned"; the contract below is the
# build's own and this is its fp32 / emulated-half restatement)
# ---------------------------------------------------------------------------------------------

def mlp_layer_shapes(n_in, n_out, n_neurons=64, n_hidden_layers=1):
    """Row-major [out, in] weight matrices, no biases; the last layer's rows are padded to a
    multiple of 16 (padded rows are parameters too, their outputs are discarded)."""
    pad16 = lambda v: (v + 15) // 16 * 16
    shapes, d = [], pad16(n_in)
    for _ in range(n_hidden_layers):
        shapes.append((n_neurons, d))
        d = n_neurons
    shapes.append((pad16(n_out), d))
    return shapes


def mlp_split(params, n_in, n_out, n_neurons=64, n_hidden_layers=1):
    ws, p = [], 0
    for (o, i) in mlp_layer_shapes(n_in, n_out, n_neurons, n_hidden_layers):
        ws.append(np.asarray(params[p:p + o * i], np.float32).reshape(o, i))
        p += o * i
    assert p == len(params)
    return ws


def _q(a, half):
    if half is None:
        return a.astype(np.float32)
    if half == 'f16':
        return a.astype(np.float16).astype(np.float32)
    if half == 'bf16':
        u = np.ascontiguousarray(a, np.float32).view(np.uint32).astype(np.uint64)
        r = ((u + 0x7FFF + ((u >> 16) & 1)) >> 16) << 16
        return r.astype(np.uint32).view(np.float32)
    raise ValueError(half)


def mlp_forward(x, params, n_in, n_out, n_neurons=64, n_hidden_layers=1, out_act='none', half=None,
                return_acts=False):
    """networks/style_nerf.py:44-98 network_config semantics: ReLU hidden, output None|Sigmoid.
    half in {None,'f16','bf16'} emulates the kernel contract: inputs, weights and hidden
    activations rounded to that type, products accumulated in fp32."""
    ws = mlp_split(params, n_in, n_out, n_neurons, n_hidden_layers)
    a = _q(np.asarray(x, np.float32), half)
    acts = [a]
    for li, w in enumerate(ws):
        z = a.astype(np.float64) @ _q(w, half).T.astype(np.float64)
        z = z.astype(np.float32)
        if li < len(ws) - 1:
            a = _q(np.maximum(z, 0), half)
            acts.append(a)
        else:
            z = z[:, :n_out]
            if out_act == 'sigmoid':
                z = (1.0 / (1.0 + np.exp(-z.astype(np.float64)))).astype(np.float32)
            a = z
    return (a, acts) if return_acts else a


# ---------------------------------------------------------------------------------------------
# model wiring + epilogue (Python in the reference)
# ---------------------------------------------------------------------------------------------

def bbox_normalize(pts, min_pt, max_pt):
    """common.py:276-288"""
    min_pt = np.asarray(min_pt, np.float32)
    size = np.asarray(max_pt, np.float32) - min_pt
    return ((np.asarray(pts, np.float32) - min_pt) / size).astype(np.float32)


def encoder_inputs(pts, bound_box=2.0):
    """style_nerf.py:121 (BBox.normalize) followed by GridEncoder.forward's (x + 1) / 2 remap
    (grid.py:177, bound=1): already-normalised [0,1] inputs land in [0.5,1]."""
    x = bbox_normalize(pts, [-bound_box] * 3, [bound_box] * 3)
    return ((x + np.float32(1)) / np.float32(2)).astype(np.float32)


class FieldParams:
    """Plain container: two hash tables [R,2] + four flat MLP parameter vectors."""

    def __init__(self, emb_density, emb_color, p_density, p_color1, p_color2, p_class, offsets,
                 per_level_scale, base_resolution=16, num_classes=5, bound=2.0):
        self.emb_density, self.emb_color = emb_density, emb_color
        self.p_density, self.p_color1, self.p_color2, self.p_class = p_density, p_color1, p_color2, p_class
        self.offsets, self.per_level_scale, self.base_resolution = offsets, per_level_scale, base_resolution
        self.num_classes, self.bound = num_classes, bound


def field_forward(fp, pts, sigma_only=False, half=None, table_half=False):
    """networks/style_nerf.py:120-142 with use_dir=False.  Returns (rgbs|classes [M,3+nc], sigmas [M])
    and the raw density logit (for trunc_exp's backward)."""
    x = encoder_inputs(pts, fp.bound)
    ed = round_f16(fp.emb_density) if table_half else fp.emb_density
    xd = grid_encode_forward(x, ed, fp.offsets, fp.per_level_scale, fp.base_resolution, 0, True, 0)
    logit = mlp_forward(xd, fp.p_density, 32, 1, 64, 1, 'none', half)[:, 0]
    sigmas = np.exp(logit.astype(np.float32)).astype(np.float32)     # tcnn_nerf.py:55-60
    if sigma_only:
        return None, sigmas, logit
    ec = round_f16(fp.emb_color) if table_half else fp.emb_color
    xc = grid_encode_forward(x, ec, fp.offsets, fp.per_level_scale, fp.base_resolution, 0, True, 0)
    classes = mlp_forward(xc, fp.p_class, 32, fp.num_classes, 64, 1, 'none', half)
    c1 = mlp_forward(xc, fp.p_color1, 32, 16, 64, 1, 'none', half)
    rgb = mlp_forward(c1, fp.p_color2, 16, 3, 64, 2, 'sigmoid', half)
    return np.concatenate([rgb, classes], axis=1).astype(np.float32), sigmas, logit


def render_epilogue(weights_sum, depth, image, nears, fars):
    """renderer.py:229-233: white background, depth normalisation, class split."""
    classes = image[:, 3:]
    rgb = image[:, :3] + (1 - weights_sum)[:, None]
    d = np.clip(depth - nears, 0, None) / (fars - nears)
    return rgb.astype(np.float32), d.astype(np.float32), classes


def compute_psnr(mse):
    """utils/__init__.py:323-325"""
    return float(-10.0 * np.log(mse) / np.log(10.0))


def generate_rays(pose, w, h, fx, fy, cx, cy, camera_flip=0, patch=None, pix_indices=None):
    """nerf_lib.py:69-142 + common.py:139-147.  pix_indices replaces np.random.choice (:134):
    the caller supplies the 1-D pixel ids so the restatement is RNG-free."""
    x_coords = np.linspace(0, w, num=2 * w + 1, dtype=np.float32)[1::2]
    y_coords = np.linspace(0, h, num=2 * h + 1, dtype=np.float32)[1::2]
    if patch is not None:
        px, py, pw, ph = patch
        x_coords = x_coords[px:px + pw]
        y_coords = y_coords[py:py + ph]
    i, j = np.meshgrid(x_coords, y_coords, indexing='xy')
    k = np.ones_like(i)
    # float32 throughout, as in the reference (float32 arrays op python floats stay float32;
    # the torch einsum is a float32 matmul).  Matches the golden to 1 ulp (fp32 FMA order).
    dirs = np.stack([(i - np.float32(cx)) / np.float32(fx), (j - np.float32(cy)) / np.float32(fy), k],
                    axis=-1).astype(np.float32)
    flip = np.where([(camera_flip >> b) & 1 for b in [2, 1, 0]], -1, 1).astype(np.float32)
    dirs = dirs * flip
    pose = np.asarray(pose, np.float32)
    rays_d = (dirs.reshape(-1, 3) @ pose[:3, :3].T).astype(np.float32)
    if pix_indices is not None:
        rays_d = rays_d[pix_indices]
    rays_d = rays_d / np.sqrt((rays_d * rays_d).sum(-1, keepdims=True, dtype=np.float32))
    rays_o = np.tile(pose[:3, 3], (rays_d.shape[0], 1))
    return rays_o.astype(np.float32), rays_d.astype(np.float32)
