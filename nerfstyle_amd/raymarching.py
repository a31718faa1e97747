"""Host-side mirror of the reference's `raymarching` package (raymarching/raymarching.py): same
function names, argument meaning and return values, backed by libnsr_hip.so through the C ABI.

Reference aliases mirrored: near_far_from_aabb (:52), morton3D (:113), morton3D_invert (:136),
packbits (:167), march_rays_train (:288), composite_rays_train (:350), march_rays (:427),
composite_rays (:462).  Like the reference wrappers, inputs are cast to float32 and made
contiguous; unlike them, nothing is silently moved to the GPU: a CPU tensor raises.

Extra entry points with no reference counterpart (used by the renderer's fast path):
march_rays_train_nosync (capacity-bounded, no `.item()`), compact_alive.
"""
import torch
from torch.autograd import Function

from . import _lib as L
from . import profiling


def _f32(t):
    return t.detach().to(torch.float32).contiguous()


class _near_far_from_aabb(Function):
    @staticmethod
    def forward(ctx, rays_o, rays_d, aabb, min_near=0.2):
        """raymarching.py:19-49"""
        rays_o = _f32(rays_o).view(-1, 3)
        rays_d = _f32(rays_d).view(-1, 3)
        aabb = _f32(aabb)
        N = rays_o.shape[0]
        nears = torch.empty(N, dtype=torch.float32, device=rays_o.device)
        fars = torch.empty(N, dtype=torch.float32, device=rays_o.device)
        L.check(L.lib().nsr_near_far_from_aabb(L.p(rays_o), L.p(rays_d), L.p(aabb), N, float(min_near), L.p(nears),
                                               L.p(fars), L.stream()), 'near_far_from_aabb')
        ctx.mark_non_differentiable(nears, fars)
        return nears, fars


near_far_from_aabb = _near_far_from_aabb.apply


def near_far_from_aabb_into(rays_o, rays_d, aabb, min_near, nears, fars):
    """near_far_from_aabb writing into caller-owned [N] float32 buffers (static buffers of a captured step)"""
    rays_o = _f32(rays_o).view(-1, 3)
    rays_d = _f32(rays_d).view(-1, 3)
    N = rays_o.shape[0]
    assert nears.numel() == N and fars.numel() == N and nears.dtype == fars.dtype == torch.float32
    L.check(L.lib().nsr_near_far_from_aabb(L.p(rays_o), L.p(rays_d), L.p(_f32(aabb)), N, float(min_near), L.p(nears),
                                           L.p(fars), L.stream()), 'near_far_from_aabb')
    return nears, fars


def morton3D(coords):
    """raymarching.py:89-113: coords [N,3] int -> indices [N] int32"""
    coords = coords.int().contiguous()
    N = coords.shape[0]
    indices = torch.empty(N, dtype=torch.int32, device=coords.device)
    L.check(L.lib().nsr_morton3d(L.p(coords), N, L.p(indices), L.stream()), 'morton3D')
    return indices


def morton3D_invert(indices):
    """raymarching.py:116-136"""
    indices = indices.int().contiguous()
    N = indices.shape[0]
    coords = torch.empty(N, 3, dtype=torch.int32, device=indices.device)
    L.check(L.lib().nsr_morton3d_invert(L.p(indices), N, L.p(coords), L.stream()), 'morton3D_invert')
    return coords


def packbits(grid, thresh, bitfield=None):
    """raymarching.py:139-167: grid [C, H^3] float -> bitfield uint8 [C*H^3/8]"""
    grid = _f32(grid)
    C, H3 = grid.shape[0], grid.shape[1]
    N = C * H3 // 8
    if bitfield is None:
        bitfield = torch.empty(N, dtype=torch.uint8, device=grid.device)
    L.check(L.lib().nsr_packbits(L.p(grid), N, float(thresh), L.p(bitfield), L.stream()), 'packbits')
    return bitfield


def _march_workspace(N, device, bound, max_steps):
    nbytes = int(L.lib().nsr_march_rays_train_workspace_bytes(N, float(bound), int(max_steps)))
    return torch.empty((nbytes + 3) // 4, dtype=torch.int32, device=device)


def march_rays_train_nosync(rays_o, rays_d, bound, density_bitfield, C, H, nears, fars, M, step_counter,
                            dt_gamma=0., max_steps=1024, want_dirs=False, out=None):
    """Capacity-bounded march: emits into [M, .] buffers and never reads the sample count on the
    host (the count stays in step_counter[0] on the device).  Samples past the count are NOT
    zero-filled (the reference memsets 40 B x N x max_steps per call, raymarching.py:238-240):
    consumers bound themselves by the device count / by `rays`.
    Returns xyzs [M,3], dirs [M,3] | None, deltas [M,4], rays [N,3]."""
    rays_o = _f32(rays_o).view(-1, 3)
    rays_d = _f32(rays_d).view(-1, 3)
    N = rays_o.shape[0]
    dev = rays_o.device
    if out is not None:
        # caller-owned sample buffers (xyzs [M,3], deltas [M,4] f32, rays [N,3] i32), e.g. the static buffers of a captured step
        xyzs, deltas, rays = out
        assert xyzs.shape == (M, 3) and deltas.shape == (M, 4) and rays.shape == (N, 3) and not want_dirs
        dirs = None
    else:
        xyzs = torch.empty(M, 3, dtype=torch.float32, device=dev)
        dirs = torch.empty(M, 3, dtype=torch.float32, device=dev) if want_dirs else None
        deltas = torch.empty(M, 4, dtype=torch.float32, device=dev)
        rays = torch.empty(N, 3, dtype=torch.int32, device=dev)
    ws = _march_workspace(N, dev, bound, max_steps)
    with profiling.timed('march_train'):
        L.check(L.lib().nsr_march_rays_train(
            L.p(rays_o), L.p(rays_d), None, L.p(density_bitfield), float(bound), float(dt_gamma), int(max_steps), 0,
            N, int(C), int(H), int(M), L.p(nears), L.p(fars), L.p(xyzs), L.p(dirs), L.p(deltas), L.p(rays),
            L.p(step_counter), None, L.p(ws), L.stream()), 'march_rays_train')
    return xyzs, dirs, deltas, rays


def march_rays_train(rays_o, rays_d, z_hats, bound, density_bitfield, C, H, nears, fars,
                     step_counter=None, mean_count=-1, perturb=False, align=-1,
                     force_all_rays=False, dt_gamma=0, max_steps=1024, is_ndc=False):
    """raymarching.py:174-288, same arguments and return values (xyzs, dirs, deltas, rays), same
    host synchronisation on the emitted count (`step_counter[0].item()`, :276) and the same
    align-to-`align` padding with zeroed samples (:238-240,277-281).  perturb is ignored like
    the reference does (:247)."""
    rays_o = _f32(rays_o).view(-1, 3)
    rays_d = _f32(rays_d).view(-1, 3)
    density_bitfield = density_bitfield.contiguous()
    N = rays_o.shape[0]
    dev = rays_o.device
    M = N * max_steps
    if not force_all_rays and mean_count > 0:
        if align > 0:
            mean_count += align - mean_count % align
        M = mean_count
    if is_ndc:
        z_hats = _f32(z_hats).view(-1)
    xyzs = torch.empty(M, 3, dtype=torch.float32, device=dev)
    dirs = torch.empty(M, 3, dtype=torch.float32, device=dev)
    deltas = torch.empty(M, 4, dtype=torch.float32, device=dev)
    rays = torch.empty(N, 3, dtype=torch.int32, device=dev)
    if step_counter is None:
        step_counter = torch.zeros(2, dtype=torch.int32, device=dev)
    ws = _march_workspace(N, dev, bound, max_steps)
    L.check(L.lib().nsr_march_rays_train(
        L.p(rays_o), L.p(rays_d), L.p(z_hats) if is_ndc else None, L.p(density_bitfield), float(bound),
        float(dt_gamma), int(max_steps), int(bool(is_ndc)), N, int(C), int(H), int(M), L.p(nears), L.p(fars),
        L.p(xyzs), L.p(dirs), L.p(deltas), L.p(rays), L.p(step_counter), None, L.p(ws), L.stream()),
        'march_rays_train')
    m_emitted = int(step_counter[0].item())   # D2H sync, as in the reference
    m = m_emitted
    if force_all_rays or mean_count <= 0:
        if align > 0:
            m += align - m % align
        m = min(m, M)
    else:
        m = M
    # the reference hands back zero-initialised tails (torch.zeros allocation); only the tail needs it
    lo = min(m_emitted, m)
    xyzs, dirs, deltas = xyzs[:m], dirs[:m], deltas[:m]
    xyzs[lo:].zero_()
    dirs[lo:].zero_()
    deltas[lo:].zero_()
    if not is_ndc:
        deltas[:, 2:].zero_()
    return xyzs, dirs, deltas, rays


class _composite_rays_train(Function):
    @staticmethod
    def forward(ctx, sigmas, rgbs, deltas, rays, T_thresh=1e-4, is_ndc=False):
        """raymarching.py:291-325"""
        sigmas = sigmas.to(torch.float32).contiguous().view(-1)
        rgbs = rgbs.to(torch.float32).contiguous()
        deltas = deltas.contiguous()
        M, N, C = sigmas.shape[0], rays.shape[0], rgbs.shape[1]
        dev = sigmas.device
        weights_sum = torch.empty(N, dtype=torch.float32, device=dev)
        depth = torch.empty(N, dtype=torch.float32, device=dev)
        image = torch.empty(N, C, dtype=torch.float32, device=dev)
        L.check(L.lib().nsr_composite_rays_train_forward(
            L.p(sigmas), L.p(rgbs), L.p(deltas), L.p(rays), M, N, C, float(T_thresh), int(bool(is_ndc)),
            L.p(weights_sum), L.p(depth), L.p(image), L.stream()), 'composite_rays_train_forward')
        ctx.save_for_backward(sigmas, rgbs, deltas, rays, weights_sum, image)
        ctx.dims = [M, N, C, T_thresh]
        ctx.is_ndc = is_ndc
        ctx.mark_non_differentiable(depth)
        return weights_sum, depth, image

    @staticmethod
    def backward(ctx, grad_weights_sum, grad_depth, grad_image):
        """raymarching.py:327-347 (grad_depth is ignored there too, :331)"""
        sigmas, rgbs, deltas, rays, weights_sum, image = ctx.saved_tensors
        M, N, C, T_thresh = ctx.dims
        grad_weights_sum = grad_weights_sum.to(torch.float32).contiguous()
        grad_image = grad_image.to(torch.float32).contiguous()
        grad_sigmas = torch.zeros_like(sigmas)
        grad_rgbs = torch.zeros_like(rgbs)
        L.check(L.lib().nsr_composite_rays_train_backward(
            L.p(grad_weights_sum), L.p(grad_image), L.p(sigmas), L.p(rgbs), L.p(deltas), L.p(rays),
            int(bool(ctx.is_ndc)), L.p(weights_sum), L.p(image), M, N, C, float(T_thresh), L.p(grad_sigmas),
            L.p(grad_rgbs), L.stream()), 'composite_rays_train_backward')
        return grad_sigmas, grad_rgbs, None, None, None, None


composite_rays_train = _composite_rays_train.apply


def march_rays(n_alive, n_step, rays_alive, rays_t, rays_o, rays_d, z_hats, bound, density_bitfield,
               C, H, near, far, align=-1, perturb=False, dt_gamma=0, max_steps=1024, is_ndc=False):
    """raymarching.py:357-424"""
    rays_o = _f32(rays_o).view(-1, 3)
    rays_d = _f32(rays_d).view(-1, 3)
    dev = rays_o.device
    M = n_alive * n_step
    if align > 0:
        M += align - (M % align)
    xyzs = torch.zeros(M, 3, dtype=torch.float32, device=dev)
    dirs = torch.zeros(M, 3, dtype=torch.float32, device=dev)
    deltas = torch.zeros(M, 4, dtype=torch.float32, device=dev)
    noises = torch.rand(n_alive, dtype=torch.float32, device=dev) if perturb else None
    L.check(L.lib().nsr_march_rays(
        int(n_alive), int(n_step), L.p(rays_alive), L.p(rays_t), L.p(rays_o), L.p(rays_d),
        L.p(_f32(z_hats).view(-1)) if is_ndc else None, float(bound), float(dt_gamma), int(max_steps),
        int(bool(is_ndc)), int(C), int(H), L.p(density_bitfield), L.p(near), L.p(far), L.p(xyzs), L.p(dirs),
        L.p(deltas), L.p(noises), L.stream()), 'march_rays')
    return xyzs, dirs, deltas


def composite_rays(n_alive, n_step, rays_alive, rays_t, sigmas, rgbs, deltas, is_ndc, weights_sum, depth, image,
                   T_thresh=1e-2):
    """raymarching.py:430-459 (in place on rays_alive, rays_t, weights_sum, depth, image)"""
    t_size = 2 if is_ndc else 1
    assert rays_t.shape[-1] == t_size
    sigmas = sigmas.to(torch.float32).contiguous().view(-1)
    rgbs = rgbs.to(torch.float32).contiguous()
    C = rgbs.shape[-1]
    L.check(L.lib().nsr_composite_rays(
        int(n_alive), int(n_step), float(T_thresh), L.p(rays_alive), L.p(rays_t), L.p(sigmas), L.p(rgbs),
        L.p(deltas), C, int(bool(is_ndc)), L.p(weights_sum), L.p(depth), L.p(image), L.stream()), 'composite_rays')
    return tuple()


def compact_alive(rays_alive, n_alive, out=None, n_out=None):
    """Stable device-side compaction of `rays_alive[:n_alive] >= 0` (replaces the boolean-mask
    indexing of renderer.py:284).  Returns (out, n_out) with n_out a 1-element device tensor."""
    dev = rays_alive.device
    if out is None:
        out = torch.empty(max(n_alive, 1), dtype=torch.int32, device=dev)
    if n_out is None:
        n_out = torch.empty(1, dtype=torch.int32, device=dev)
    nbytes = int(L.lib().nsr_compact_alive_workspace_bytes(n_alive))
    ws = torch.empty((nbytes + 3) // 4, dtype=torch.int32, device=dev)
    L.check(L.lib().nsr_compact_alive(L.p(rays_alive), int(n_alive), L.p(out), L.p(n_out), L.p(ws), L.stream()),
            'compact_alive')
    return out, n_out
