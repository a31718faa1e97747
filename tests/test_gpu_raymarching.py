"""GPU parity: the HIP ray-marching family (through the C ABI, via nerfstyle_amd.raymarching)
against the CPU oracle on the same seeded inputs.

Bars: integer/index work (morton, packbits, ray table, sample counts, alive flags) bit-exact;
marched positions/deltas bit-exact (same fp32 operation order, contraction off on both sides);
compositing within 2e-5 absolute of the oracle (the kernels use __expf like the reference,
the oracle uses expf)."""
import numpy as np
import pytest
import torch

from helpers import room_cameras, room_rays, small_scene

pytestmark = pytest.mark.gpu

AABB = np.array([-2, -2, -2, 2, 2, 2], np.float32)


def T(a, dev, dtype=None):
    return torch.as_tensor(np.ascontiguousarray(a), device=dev) if dtype is None else torch.as_tensor(
        np.ascontiguousarray(a), device=dev).to(dtype)


def test_near_far_morton_packbits(O, dev):
    from nerfstyle_amd import raymarching as R
    ro, rd = room_rays(O, 5000, seed=1)
    ro[:7] = 9.0                                               # rays that miss the box
    near, far = O.near_far_from_aabb(ro, rd, AABB, 0.2)
    n2, f2 = R.near_far_from_aabb(T(ro, dev), T(rd, dev), T(AABB, dev), 0.2)
    assert np.array_equal(n2.cpu().numpy(), near) and np.array_equal(f2.cpu().numpy(), far)
    rng = np.random.default_rng(2)
    c = rng.integers(0, 1024, (100003, 3)).astype(np.int32)
    ind = R.morton3D(T(c, dev))
    assert np.array_equal(ind.cpu().numpy(), O.morton3D(c))
    assert np.array_equal(R.morton3D_invert(ind).cpu().numpy(), c)
    g = rng.random((2, 128 ** 3)).astype(np.float32)
    assert np.array_equal(R.packbits(T(g, dev), 0.37).cpu().numpy(), O.packbits(g, 0.37))
    # empty input
    assert R.morton3D(torch.zeros(0, 3, dtype=torch.int32, device=dev)).numel() == 0


@pytest.mark.parametrize('n_rays,max_steps,dt_gamma', [(4096, 1024, 0.), (1000, 512, 0.), (257, 64, 0.), (24001, 1024, 0.),
                                                       (3001, 1024, 1. / 128), (22001, 1024, 1. / 256)])
def test_march_rays_train_bit_exact(O, dev, n_rays, max_steps, dt_gamma):
    """Both march implementations against the sequential oracle, bit for bit: batches up to 20 480 rays run
    one wave per ray (k_march_wpr: speculative probes + successor walk), larger ones one thread per ray;
    dt_gamma != 0 makes the step grow along the ray (cone marching, raymarching.cu:468)."""
    from nerfstyle_amd import raymarching as R
    grid, bits = small_scene()
    ro, rd = room_rays(O, n_rays, seed=n_rays)
    near, far = O.near_far_from_aabb(ro, rd, AABB, 0.2)
    xo, do, dlo, rays_o, cnt_o = O.march_rays_train(ro, rd, 2.0, bits, 2, 128, near, far, max_steps, dt_gamma=dt_gamma, align=128)
    counter = torch.zeros(2, dtype=torch.int32, device=dev)
    x, d, dl, rays = R.march_rays_train(T(ro, dev), T(rd, dev), None, 2.0, T(bits, dev), 2, 128, T(near, dev), T(far, dev),
                                        counter, -1, False, 128, True, dt_gamma, max_steps, False)
    assert np.array_equal(counter.cpu().numpy(), cnt_o)
    assert np.array_equal(rays.cpu().numpy(), rays_o)           # deterministic scan order == sequential oracle order
    assert x.shape == xo.shape
    assert np.array_equal(x.cpu().numpy(), xo)
    assert np.array_equal(dl.cpu().numpy(), dlo)
    assert np.array_equal(d.cpu().numpy(), do)
    assert rays_o[:, 2].sum() > 0


def test_march_overflow_drop_and_nosync(O, dev):
    """Capacity smaller than the emitted count: rays with offset+count >= M are dropped exactly like
    the reference (raymarching.cu:517) and the no-sync path agrees with the synchronising one."""
    from nerfstyle_amd import raymarching as R
    grid, bits = small_scene()
    ro, rd = room_rays(O, 2048, seed=9)
    near, far = O.near_far_from_aabb(ro, rd, AABB, 0.2)
    _, _, _, rays_full, cnt = O.march_rays_train(ro, rd, 2.0, bits, 2, 128, near, far, 1024)
    M = int(cnt[0]) // 2
    xo, _, dlo, rays_o, cnt_o = O.march_rays_train(ro, rd, 2.0, bits, 2, 128, near, far, 1024, M=M)
    counter = torch.zeros(2, dtype=torch.int32, device=dev)
    x, _, dl, rays = R.march_rays_train_nosync(T(ro, dev), T(rd, dev), 2.0, T(bits, dev), 2, 128, T(near, dev), T(far, dev), M,
                                               counter, 0., 1024)
    assert np.array_equal(rays.cpu().numpy(), rays_o) and np.array_equal(counter.cpu().numpy(), cnt_o)
    kept = (rays_o[:, 2] > 0) & (rays_o[:, 1] + rays_o[:, 2] < M)
    assert kept.sum() > 0 and (~kept & (rays_o[:, 2] > 0)).sum() > 0
    xc, dlc = x.cpu().numpy(), dl.cpu().numpy()
    for n in np.nonzero(kept)[0][::37]:
        o, c = rays_o[n, 1], rays_o[n, 2]
        assert np.array_equal(xc[o:o + c], xo[o:o + c]) and np.array_equal(dlc[o:o + c, :2], dlo[o:o + c, :2])


def _composite_inputs(O, n_rays=3000, C=8, seed=0):
    grid, bits = small_scene()
    ro, rd = room_rays(O, n_rays, seed=seed)
    near, far = O.near_far_from_aabb(ro, rd, AABB, 0.2)
    xyzs, _, deltas, rays, cnt = O.march_rays_train(ro, rd, 2.0, bits, 2, 128, near, far, 1024, align=128)
    rng = np.random.default_rng(seed)
    M = len(xyzs)
    sig = (rng.random(M) ** 4 * 400).astype(np.float32)       # mostly small, some opaque -> early stops
    rgb = rng.random((M, C)).astype(np.float32)
    return sig, rgb, deltas, rays, near, far


@pytest.mark.parametrize('C', [8, 3, 11])
def test_composite_train_forward_backward(O, dev, C):
    from nerfstyle_amd import raymarching as R
    sig, rgb, deltas, rays, near, far = _composite_inputs(O, 3000, C, seed=C)
    ws_o, d_o, im_o = O.composite_rays_train_forward(sig, rgb, deltas, rays, 1e-4)
    s_t, r_t = T(sig, dev).requires_grad_(), T(rgb, dev).requires_grad_()
    ws, depth, image = R.composite_rays_train(s_t, r_t, T(deltas, dev), T(rays, dev), 1e-4, False)
    assert np.abs(ws.detach().cpu().numpy() - ws_o).max() < 2e-5
    assert np.abs(image.detach().cpu().numpy() - im_o).max() < 2e-5
    assert np.abs(depth.detach().cpu().numpy() - d_o).max() < 2e-4
    rng = np.random.default_rng(1)
    gws = rng.standard_normal(len(ws_o)).astype(np.float32)
    gim = rng.standard_normal(im_o.shape).astype(np.float32)
    (ws * T(gws, dev)).sum().add((image * T(gim, dev)).sum()).backward()
    gs_o, gr_o = O.composite_rays_train_backward(gws, gim, sig, rgb, deltas, rays, ws_o, im_o, 1e-4)
    gs, gr = s_t.grad.cpu().numpy(), r_t.grad.cpu().numpy()
    assert np.abs(gr - gr_o).max() < 5e-5
    # grad_sigma sums terms of size ~|grad| * delta; compare relative to its scale
    assert np.abs(gs - gs_o).max() < 1e-4 * max(1.0, np.abs(gs_o).max())
    # early-stopped tails really are zero
    assert np.array_equal(gs == 0, gs_o == 0)


def test_inference_march_composite_loop(O, dev):
    """The render_test iteration (renderer.py:266-285) step by step against the oracle: alive sets,
    rays_t and accumulators after every iteration."""
    from nerfstyle_amd import raymarching as R
    grid, bits = small_scene()
    N, C = 1500, 8
    ro, rd = room_rays(O, N, seed=21)
    near, far = O.near_far_from_aabb(ro, rd, AABB, 0.2)
    rng = np.random.default_rng(3)

    def fake_field(xyz):      # deterministic stand-in for the model, identical on both sides
        s = (np.abs(np.sin(xyz.sum(1) * 3.1)) * 60).astype(np.float32)
        c = np.stack([np.abs(np.cos(xyz[:, i % 3] * (i + 1))) for i in range(C)], 1).astype(np.float32)
        return s, c

    alive_o = np.arange(N, dtype=np.int32)
    rt_o = near.copy()[:, None]
    ws_o = np.zeros(N, np.float32); d_o = np.zeros(N, np.float32); im_o = np.zeros((N, C), np.float32)
    alive = T(alive_o, dev); rt = T(rt_o, dev)
    ws = torch.zeros(N, device=dev); dp = torch.zeros(N, device=dev); im = torch.zeros(N, C, device=dev)
    bits_t, ro_t, rd_t, near_t, far_t = T(bits, dev), T(ro, dev), T(rd, dev), T(near, dev), T(far, dev)
    step, it = 0, 0
    while step < 1024 and len(alive_o) > 0 and it < 40:
        n_alive = len(alive_o)
        n_step = max(min(N // n_alive, 8), 1)
        xo, _, dlo = O.march_rays(n_alive, n_step, alive_o, rt_o, ro, rd, 2.0, bits, 2, 128, near, far, 128, 1024)
        x, _, dl = R.march_rays(n_alive, n_step, alive, rt, ro_t, rd_t, None, 2.0, bits_t, 2, 128, near_t, far_t, 128, False, 0.,
                                1024, False)
        assert np.array_equal(x.cpu().numpy(), xo) and np.array_equal(dl.cpu().numpy()[:, :2], dlo[:, :2])
        s, c = fake_field(xo)
        O.composite_rays(n_alive, n_step, alive_o, rt_o, s, c, dlo, ws_o, d_o, im_o, 1e-4)
        R.composite_rays(n_alive, n_step, alive, rt, T(s, dev), T(c, dev), dl, False, ws, dp, im, 1e-4)
        assert np.array_equal(alive.cpu().numpy()[:n_alive], alive_o)
        assert np.abs(ws.cpu().numpy() - ws_o).max() < 2e-5 and np.abs(im.cpu().numpy() - im_o).max() < 2e-5
        assert np.allclose(rt.cpu().numpy(), rt_o, atol=1e-6)
        out, n_out = R.compact_alive(alive, n_alive)
        alive_o = alive_o[alive_o >= 0]
        assert int(n_out.item()) == len(alive_o)
        assert np.array_equal(out.cpu().numpy()[:len(alive_o)], alive_o)
        alive = out[:len(alive_o)].contiguous()
        step += n_step
        it += 1
    assert it > 3


def test_generate_rays_kernel_vs_reference_golden(O, dev, golden):
    from nerfstyle_amd.common import Box2D, Intrinsics
    from nerfstyle_amd.rays import generate_rays
    c = room_cameras()
    intr = Intrinsics(c['h'], c['w'], c['fl_x'], c['fl_y'], c['cx'], c['cy'])
    pose = torch.tensor(golden['pose0'], device=dev)
    rays, _ = generate_rays(pose, intr, camera_flip=3)
    sel = golden['rays_full_sel']
    assert np.array_equal(rays.origins.cpu().numpy()[sel], golden['rays_full_o'])
    assert np.abs(rays.dirs.cpu().numpy()[sel] - golden['rays_full_d']).max() <= 2.4e-7     # 2 ulp: fp32, different sum order
    rays, _ = generate_rays(pose, intr, patch=Box2D(200, 0, 200, 200), camera_flip=3)
    assert np.abs(rays.dirs.cpu().numpy()[golden['rays_patch_sel']] - golden['rays_patch_d']).max() <= 2.4e-7
    np.random.seed(69420)
    img = torch.arange(4 * c['h'] * c['w'], dtype=torch.float32, device=dev).view(4, c['h'], c['w'])
    rays, target = generate_rays(pose, intr, img, bsize=4096, camera_flip=3)
    assert np.abs(rays.dirs.cpu().numpy()[:512] - golden['rays_rand_d']).max() <= 2.4e-7
    idx = golden['rays_rand_idx']
    assert np.array_equal(target[:, 0].cpu().numpy().astype(np.int64), idx)            # channel 0 holds the pixel id
