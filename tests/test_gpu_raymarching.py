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


def test_composite_train_kernel_vs_reference_integrate_points_golden(O, dev, golden):
    """The HIP composite itself (nsr_composite_rays_train_forward, through the C ABI) against output of the REFERENCE:
    nerf_lib.integrate_points run on seeded [N, K = 64] inputs by tests/golden/make_goldens.py (ip_* fixtures).  With no early
    stop (T_thresh = 0) raymarching.cu's composite and integrate_points are the same quadrature.  64 samples per ray is exactly
    one wave trip of the kernel; a second case puts 70 zero-density samples in front of every ray (134 samples = three trips):
    the result must not change, which exercises the carries between trips."""
    from nerfstyle_amd import raymarching as R
    dists, rgbs, dens = golden['ip_dists'], golden['ip_rgbs'], golden['ip_dens']
    N, K = dists.shape
    deltas = np.zeros((N * K + 1, 4), np.float32)
    deltas[:N * K, 0] = dists.reshape(-1)
    deltas[:N * K, 1] = dists.reshape(-1)
    rays = np.stack([np.arange(N), np.arange(N) * K, np.full(N, K)], 1).astype(np.int32)
    sig = np.concatenate([dens.reshape(-1), [0]]).astype(np.float32)
    rgb = np.concatenate([rgbs.reshape(-1, 3), np.zeros((1, 3))]).astype(np.float32)
    ws, depth, image = R.composite_rays_train(T(sig, dev), T(rgb, dev), T(deltas, dev), T(rays, dev), 0.0, False)
    assert np.allclose(image.cpu().numpy(), golden['ip_rgb_map'], atol=3e-6)
    assert np.allclose(ws.cpu().numpy()[:, None], golden['ip_acc_map'], atol=3e-6)
    # the same rays with 70 zero-density samples in front: 134 samples = three wave trips, carries across the trips
    pad = 70
    K2 = K + pad
    d2 = np.zeros((N * K2 + 1, 4), np.float32)
    s2 = np.zeros(N * K2 + 1, np.float32)
    r2 = np.zeros((N * K2 + 1, 3), np.float32)
    dd = np.concatenate([np.full((N, pad), 0.01, np.float32), dists], 1)
    d2[:N * K2, 0] = dd.reshape(-1)
    d2[:N * K2, 1] = dd.reshape(-1)
    s2[:N * K2] = np.concatenate([np.zeros((N, pad), np.float32), dens.reshape(N, K)], 1).reshape(-1)
    r2[:N * K2] = np.concatenate([np.ones((N, pad, 3), np.float32), rgbs.reshape(N, K, 3)], 1).reshape(-1, 3)
    rays2 = np.stack([np.arange(N), np.arange(N) * K2, np.full(N, K2)], 1).astype(np.int32)
    ws2, _, image2 = R.composite_rays_train(T(s2, dev), T(r2, dev), T(d2, dev), T(rays2, dev), 0.0, False)
    assert np.allclose(image2.cpu().numpy(), golden['ip_rgb_map'], atol=3e-6)
    assert np.allclose(ws2.cpu().numpy()[:, None], golden['ip_acc_map'], atol=3e-6)


@pytest.mark.parametrize('C', [8, 4, 5])
def test_render_train_epilogue_fused_equals_composite_plus_torch(O, dev, C):
    """nsr_render_train_forward / _backward (composite + white background + class slice + depth normalisation,
    renderer.py:225-233, in one launch each way) against composite_rays_train followed by those torch expressions, values and
    the gradients w.r.t. sigmas / rgbs, for losses that use rgb_map, classes and weights_sum."""
    from nerfstyle_amd import raymarching as R
    from nerfstyle_amd.renderer import _render_train
    sig, rgb, deltas, rays, near, far = _composite_inputs(O, 3000, C, seed=10 + C)
    g = np.random.default_rng(3)
    w_rgb, w_cls, w_ws = g.standard_normal((len(rays), 3)).astype(np.float32), g.standard_normal((len(rays), C - 3)).astype(np.float32), \
        g.standard_normal(len(rays)).astype(np.float32)

    def run(fused):
        s_t, r_t = T(sig, dev).requires_grad_(), T(rgb, dev).requires_grad_()
        if fused:
            rgb_map, depth, classes, ws = _render_train(s_t, r_t, T(deltas, dev), T(rays, dev), T(near, dev), T(far, dev), 1e-4)
        else:
            ws, d, image = R.composite_rays_train(s_t, r_t, T(deltas, dev), T(rays, dev), 1e-4, False)
            rgb_map = image[:, :3] + (1 - ws).unsqueeze(-1)
            classes = image[:, 3:]
            depth = torch.clamp(d - T(near, dev), min=0) / (T(far, dev) - T(near, dev))
        loss = (rgb_map * T(w_rgb, dev)).sum() + (classes * T(w_cls, dev)).sum() + (ws * T(w_ws, dev)).sum()
        loss.backward()
        return [x.detach().cpu().numpy() for x in (rgb_map, depth, classes, ws, s_t.grad, r_t.grad)]
    a, b = run(True), run(False)
    # sample-buffer slots that belong to no ray (alignment padding) are never written by either path: compare the rays' samples
    owned = np.zeros(len(sig), bool)
    for n in range(len(rays)):
        owned[rays[n, 1]:rays[n, 1] + rays[n, 2]] = True
    a[4], b[4], a[5], b[5] = a[4][owned], b[4][owned], a[5][owned], b[5][owned]
    for x, y, tol in zip(a, b, (1e-6, 1e-6, 1e-6, 1e-6, 1e-5, 1e-6)):
        assert x.shape == y.shape and np.abs(x - y).max() <= tol * max(1.0, np.abs(y).max())
    # a loss that ignores the classes (grad None) and one that only uses weights_sum
    s_t, r_t = T(sig, dev).requires_grad_(), T(rgb, dev).requires_grad_()
    rgb_map, depth, classes, ws = _render_train(s_t, r_t, T(deltas, dev), T(rays, dev), T(near, dev), T(far, dev), 1e-4)
    rgb_map.sum().backward()
    assert float(s_t.grad.abs().sum()) > 0 and float(r_t.grad[:, 3:].abs().sum()) == 0.0


def test_recon_loss_value_and_gradient_equal_torch(dev):
    """nsr_recon_loss (MSE + lambda * cross-entropy of trainers/base.py:251-304, value + gradient in one pass, targets
    gathered through the pixel indices, times a device-side scale) against the torch expressions, to 1e-6 relative."""
    from nerfstyle_amd.recon_loss import last_terms, recon_loss
    g = torch.Generator(device=dev)
    g.manual_seed(8)
    for N, nc in ((5000, 5), (777, 13), (300000, 5)):
        P = 2 * N + 3
        rgb = torch.rand(N, 3, device=dev, generator=g).requires_grad_()
        cls = (torch.randn(N, nc, device=dev, generator=g) * 3).requires_grad_()
        t_rgb = torch.rand(P, 3, device=dev, generator=g)
        t_cls = torch.randint(0, nc, (P,), device=dev, generator=g)
        pix = torch.randperm(P, device=dev, generator=g)[:N]
        scale = torch.tensor(1024.0, device=dev)
        loss = recon_loss(rgb, cls, t_rgb, t_cls, pix, ce_lambda=1e-3, factor=0.5, scale=scale)
        loss.backward()
        rgb2, cls2 = rgb.detach().clone().requires_grad_(), cls.detach().clone().requires_grad_()
        mse = torch.nn.functional.mse_loss(rgb2, t_rgb[pix])
        ce = torch.nn.functional.cross_entropy(cls2, t_cls[pix]) * 1e-3
        ref = (mse + ce) * 0.5 * scale
        ref.backward()
        assert abs(float(loss) - float(ref)) < 2e-6 * abs(float(ref))
        terms = last_terms(loss).cpu().numpy()
        assert abs(terms[0] - float(mse)) < 2e-6 * float(mse) and abs(terms[1] - float(ce)) < 2e-6 * float(ce)
        assert float((rgb.grad - rgb2.grad).abs().max()) < 1e-6 * float(rgb2.grad.abs().max())
        assert float((cls.grad - cls2.grad).abs().max()) < 1e-6 * float(cls2.grad.abs().max()) + 1e-12
        # backward=True: the stored gradient is back-propagated from inside the call (through whatever produced the inputs)
        # and the loss comes back detached -- same value, same terms, bit-identical gradients
        base_r, base_c = rgb.detach().clone().requires_grad_(), cls.detach().clone().requires_grad_()
        l3 = recon_loss(base_r * 1.0, base_c * 1.0, t_rgb, t_cls, pix, ce_lambda=1e-3, factor=0.5, scale=scale, backward=True)
        assert not l3.requires_grad and float(l3) == float(loss)
        assert torch.equal(last_terms(l3), last_terms(loss))
        assert torch.equal(base_r.grad, rgb.grad) and torch.equal(base_c.grad, cls.grad)
    # MSE only, no pixel indirection
    rgb = torch.rand(1000, 3, device=dev, generator=g).requires_grad_()
    tgt = torch.rand(1000, 3, device=dev, generator=g)
    loss = recon_loss(rgb, None, tgt)
    loss.backward()
    assert abs(float(loss) - float(torch.mean((rgb.detach() - tgt) ** 2))) < 1e-7
    assert float((rgb.grad - 2 * (rgb.detach() - tgt) / 3000).abs().max()) < 1e-9


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
