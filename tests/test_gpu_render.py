"""GPU end-to-end: Renderer (march -> fused field -> composite -> epilogue) against the oracle
pipeline on the same checkpoint, the training step, and the inference loop.

PSNR bar (BASELINE.json): >= 40 dB between the build's render and the reference render on identical
inputs; here the "reference render" is the oracle pipeline in fp32 (the reference itself cannot run
offline), PSNR = -10 log10(mean((a-b)^2)) on [0,1] images (utils/__init__.py:323-325)."""
import numpy as np
import pytest
import torch

from helpers import rel_l2, small_scene

pytestmark = pytest.mark.gpu


def _setup(dev, nc=5, table_dtype=torch.float32, table_scale=0.5, cap=None, contrast=1.0):
    from nerfstyle_amd.common import BBox
    from nerfstyle_amd.config import NetworkConfig, RendererConfig
    from nerfstyle_amd.renderer import Renderer
    from nerfstyle_amd.scene import load_room_cameras
    from nerfstyle_amd.style_nerf import StyleTCNerf
    from oracle import torch_port as TP
    ref = TP.Field(num_classes=nc, table_scale=table_scale)
    if contrast != 1.0:
        # the seeded checkpoint is nearly grey (rgb 0.50 +- 0.009, sigma 1.0 +- 0.09): scaling the last layers spreads the colours
        # (std 0.13 at 16) and the densities (0.09 .. 11.7), so that an image comparison can tell a wrong MLP from a right one
        with torch.no_grad():
            ref.p_density[2048:] *= contrast
            ref.p_color2[-1024:] *= contrast
            ref.p_class[2048:] *= contrast
    m = StyleTCNerf(NetworkConfig(), BBox.from_radius(2.0), nc, enc_dtype=table_dtype, use_dir=False)
    sd = m.state_dict()
    sd.update({'x_density_embedder.embeddings': ref.emb_density.detach(), 'x_color_embedder.embeddings': ref.emb_color.detach(),
               'density_net.params': ref.p_density.detach(), 'color1_net.params': ref.p_color1.detach(),
               'color2_net.params': ref.p_color2.detach(), 'class_net.params': ref.p_class.detach()})
    m.load_state_dict(sd)
    poses, intr, _ = load_room_cameras()
    r = Renderer(m, RendererConfig.llff(), intr, 2.0, raymarch_channels=3 + nc, samples_per_ray_cap=cap).to(dev)
    grid, bits = small_scene()
    r.density_grid = torch.tensor(grid, device=dev)
    r.density_bitfield = torch.tensor(bits, device=dev)
    r.update_occ = False                     # fixed synthetic occupancy
    return r, ref, poses, intr, bits


def _oracle_render(O, ref, bits, ro, rd, half=None, density_scale=1.0):
    aabb = np.array([-2, -2, -2, 2, 2, 2], np.float32)
    near, far = O.near_far_from_aabb(ro, rd, aabb, 0.2)
    xyzs, _, deltas, rays, cnt = O.march_rays_train(ro, rd, 2.0, bits, 2, 128, near, far, 1024, align=128)
    fp = O.FieldParams(ref.emb_density.detach().numpy(), ref.emb_color.detach().numpy(), ref.p_density.detach().numpy(),
                       ref.p_color1.detach().numpy(), ref.p_color2.detach().numpy(), ref.p_class.detach().numpy(), ref.offsets,
                       ref.pls, num_classes=ref.nc)
    out, sig, _ = O.field_forward(fp, xyzs, half=half)
    ws, depth, image = O.composite_rays_train_forward((sig * np.float32(density_scale)).astype(np.float32), out, deltas, rays, 1e-4)
    rgb, d, classes = O.render_epilogue(ws, depth, image, near, far)
    return rgb, d, classes, int(cnt[0])


@pytest.mark.parametrize('contrast,n_rays', [(1.0, 4096), (16.0, 2048)])
def test_render_train_matches_oracle_pipeline(O, dev, contrast, n_rays):
    """Training render (march -> fused field -> composite + epilogue) against the fp32 oracle pipeline on the same rays: PSNR > 45 dB
    (bar: 40).  contrast = 16: the same checkpoint with its last layers scaled so that colours and densities really vary across
    the image (the plain seeded checkpoint renders almost uniformly grey, which any MLP would pass)."""
    r, ref, poses, intr, bits = _setup(dev, contrast=contrast)
    if contrast != 1.0:
        r.cfg.density_scale = 40.0                 # opaque surfaces: the image is the scene, not the white background
    np.random.seed(69420)
    pix = np.random.choice(intr.w * intr.h, n_rays, replace=False)
    ro, rd = O.generate_rays(poses[0], intr.w, intr.h, intr.fx, intr.fy, intr.cx, intr.cy, 3, pix_indices=pix)
    rgb_o, d_o, cls_o, m = _oracle_render(O, ref, bits, ro, rd, density_scale=r.cfg.density_scale)
    assert m > n_rays * 8
    out = r.render(torch.tensor(poses[0], device=dev), None, num_rays=None, training=True,
                   pix_subset=torch.tensor(pix, device=dev))
    rgb = out['rgb_map'].detach().cpu().numpy()
    psnr = O.compute_psnr(float(np.mean((rgb - rgb_o) ** 2)))
    assert psnr > 45.0, psnr
    if contrast != 1.0:
        hit = (1.0 - rgb_o.min(1)) > 0.05                                  # rays that see something
        assert hit.mean() > 0.2 and rgb_o[hit].std() > 0.08, (hit.mean(), rgb_o[hit].std())      # the image is not flat
        # ... and a wrong colour net is caught: shuffling the oracle's colour channels costs > 20 dB
        assert O.compute_psnr(float(np.mean((rgb - rgb_o[:, ::-1]) ** 2))) < psnr - 20.0
    assert np.abs(out['classes'].detach().cpu().numpy() - cls_o).max() < 5e-2 * max(1.0, np.abs(cls_o).max())
    ok = np.isfinite(d_o)
    assert np.abs(out['trans_map'].detach().cpu().numpy()[ok] - d_o[ok]).max() < 5e-3


def test_single_pass_inference_equals_reference_loop(O, dev):
    """render_test (one march + one field launch + nsr_composite_rays_infer) against the reference's
    iteration structure (render_test_loop: march_rays / composite_rays / compaction per step), with
    early termination active (opaque scene)."""
    from nerfstyle_amd.common import Box2D
    r, ref, poses, intr, bits = _setup(dev)
    r.cfg.density_scale = 400.0                 # opaque boxes -> rays terminate early (T < 1e-4)
    pose = torch.tensor(poses[5], device=dev)
    patch = Box2D(100, 60, 256, 200)
    fast = r.render(pose, None, patch=patch, training=False)
    r.reference_inference_loop = True
    loop = r.render(pose, None, patch=patch, training=False)
    for k in ('rgb_map', 'classes'):
        assert float((fast[k] - loop[k]).abs().max()) < 2e-4, k
    ok = torch.isfinite(loop['trans_map'])
    assert float((fast['trans_map'][ok] - loop['trans_map'][ok]).abs().max()) < 2e-3
    assert float(loop['rgb_map'].min()) < 0.9        # something was rendered


def test_render_test_matches_render_train(O, dev):
    """Inference loop (march_rays / composite_rays / compaction) vs the training path on a
    200x200 patch: same samples, T = 1 - ws vs running product, stop test one sample later."""
    from nerfstyle_amd.common import Box2D
    r, ref, poses, intr, bits = _setup(dev)
    pose = torch.tensor(poses[3], device=dev)
    patch = Box2D(150, 80, 200, 200)
    a = r.render(pose, None, patch=patch, training=True)['rgb_map'].detach()
    b = r.render(pose, None, patch=patch, training=False)['rgb_map']
    assert a.shape == (40000, 3)
    mse = float(((a - b) ** 2).mean())
    assert -10 * np.log10(max(mse, 1e-12)) > 45.0


def test_training_step_reduces_loss_and_matches_torch_adam(O, dev):
    from nerfstyle_amd.optim import FusedAdam
    r, ref, poses, intr, bits = _setup(dev, table_scale=1e-4, cap=256)
    m = r.model
    opt = FusedAdam(m, lr=1e-2, betas=(0.9, 0.999), eps=1e-15, ema_decay=0.95)
    g = torch.Generator(device='cpu').manual_seed(0)
    target = torch.rand(4096, 3, generator=g).to(dev) * 0.5
    tcls = torch.randint(0, 5, (4096,), generator=g).to(dev)
    pose = torch.tensor(poses[0], device=dev)
    pix = torch.randperm(intr.w * intr.h, generator=g)[:4096].to(dev)
    losses = []
    for it in range(30):
        out = r.render(pose, None, training=True, pix_subset=pix)
        loss = torch.mean((out['rgb_map'] - target) ** 2) + 1e-3 * torch.nn.functional.cross_entropy(out['classes'], tcls)
        loss.backward()
        if it == 0:
            # one fused step == torch.optim.Adam on the same gradient (fp32 elementwise; eps 1e-15)
            p0 = m.arena.detach().clone()
            g0 = m.arena.grad.detach().clone()
            pt = torch.nn.Parameter(p0.clone())
            pt.grad = g0.clone()
            topt = torch.optim.Adam([pt], lr=1e-2, betas=(0.9, 0.999), eps=1e-15)
            topt.step()
        opt.step()
        if it == 0:
            assert float((m.arena.detach() - pt.detach()).abs().max()) < 1e-6
            assert float(m.arena.grad.abs().max()) == 0.0          # zeroed on the way out
            # half copy written by the optimiser equals a fresh cast
            if m.table_dtype == torch.float16:
                assert torch.equal(m.half_tables(), m.arena.detach()[:m.table_elems].half())
        losses.append(float(loss))
    assert losses[-1] < 0.6 * losses[0], losses
    assert np.all(np.isfinite(losses))
    assert int(r.step_counter.max()) == 0            # update_occ is off: the ring of per-step counters is never written


def test_fused_adam_ema_matches_torch_ema_formula(dev):
    """EMA shadow of the fused optimiser vs torch_ema's update as the reference uses it (utils/__init__.py:116-142,
    trainers/base.py:229,426: ExponentialMovingAverage(params, decay=0.95), update() after every optimiser step):
    num_updates += 1; decay = min(decay, (1 + num_updates) / (10 + num_updates)); shadow -= (1 - decay) * (shadow - p).
    torch_ema itself is absent offline: the formula is restated here (parity unpinned against the package)."""
    from nerfstyle_amd.common import BBox
    from nerfstyle_amd.config import NetworkConfig
    from nerfstyle_amd.optim import FusedAdam
    from nerfstyle_amd.style_nerf import StyleTCNerf
    m = StyleTCNerf(NetworkConfig(), BBox.from_radius(2.0), 5, enc_dtype=None, use_dir=False).to(dev)
    opt = FusedAdam(m, lr=1e-2, betas=(0.9, 0.999), eps=1e-15, ema_decay=0.95)
    pt = torch.nn.Parameter(m.arena.detach().clone())
    topt = torch.optim.Adam([pt], lr=1e-2, betas=(0.9, 0.999), eps=1e-15)
    shadow = pt.detach().clone()
    g = torch.Generator(device=dev)
    g.manual_seed(2)
    for n in range(1, 13):
        grad = torch.randn(m.arena.shape, device=dev, generator=g) * 1e-3
        m._ensure_grad().copy_(grad)
        pt.grad = grad.clone()
        opt.step()
        topt.step()
        decay = min(0.95, (1 + n) / (10 + n))
        shadow.sub_((1.0 - decay) * (shadow - pt.detach()))
        assert float((m.arena.detach() - pt.detach()).abs().max()) < 2e-6, n
        assert float((opt.ema - shadow).abs().max()) < 2e-6, n
    assert float((opt.ema - m.arena.detach()).abs().max()) > 1e-4        # the shadow really lags the parameters
    assert torch.equal(m.half_tables(), m.arena.detach()[:m.table_elems].half())


@pytest.mark.parametrize('keywords', [None, ['x_color_embedder']])
def test_device_side_grad_scaler_equals_torch_gradscaler(dev, keywords):
    """LossScaler + FusedAdam.step(scaler=...) -- inf/nan check, skip, back-off / growth, step count, LambdaLR and bias
    corrections all on the device (nsr_grad_check / nsr_scaler_update / nsr_adam_step_scaled) -- against
    torch.cuda.amp.GradScaler + torch.optim.Adam + LambdaLR driven as the reference drives them (trainers/base.py:420-426:
    scaler.step, scaler.update, scheduler.step only if the scale did not drop, ema.update always) on a sequence of finite /
    inf / nan gradients, growth_interval 3 so that growth happens too.  With keywords = colour table only, a non-finite value
    in an UNTRAINED region must not skip the step (the reference's optimiser does not hold those parameters)."""
    from nerfstyle_amd.common import BBox
    from nerfstyle_amd.config import NetworkConfig
    from nerfstyle_amd.optim import FusedAdam, LossScaler
    from nerfstyle_amd.style_nerf import StyleTCNerf
    m = StyleTCNerf(NetworkConfig(), BBox.from_radius(2.0), 5, enc_dtype=None, use_dir=False).to(dev)
    opt = FusedAdam(m, lr=1e-2, betas=(0.9, 0.999), eps=1e-15, keywords=keywords, ema_decay=0.95)
    sc = LossScaler(init_scale=1024.0, growth_interval=3)
    decay_steps = 50.0
    n = m.arena.numel()
    trained = torch.zeros(n, dtype=torch.bool, device=dev)
    if keywords is None:
        trained[:] = True
    else:
        trained[:m.table_elems].view(-1, 4)[:, 2:] = True          # interleaved rows: [density 2 | colour 2]
    pt = torch.nn.Parameter(m.arena.detach().clone())
    topt = torch.optim.Adam([pt], lr=1e-2, betas=(0.9, 0.999), eps=1e-15)
    sched = torch.optim.lr_scheduler.LambdaLR(topt, lambda it: 0.1 ** (it / decay_steps))
    tsc = torch.cuda.amp.GradScaler(init_scale=1024.0, growth_interval=3)
    tsc.scale(torch.zeros(1, device=dev))                       # lazy initialisation of the scale tensor
    shadow = pt.detach().clone()
    g = torch.Generator(device=dev)
    g.manual_seed(4)
    bad_trained = int(torch.nonzero(trained)[12345])
    bad_untrained = int(torch.nonzero(~trained)[777]) if keywords is not None else None
    #            0     1      2     3     4     5      6     7     8     9
    kinds = ['ok', 'ok', 'inf', 'ok', 'ok', 'ok', 'nan', 'ok', 'untrained_inf', 'ok', 'ok', 'ok']
    skipped = 0
    for it, kind in enumerate(kinds):
        scale_now = tsc.get_scale()
        assert sc.get_scale() == scale_now, it
        grad = torch.randn(n, device=dev, generator=g) * 1e-3 * scale_now
        if kind == 'inf':
            grad[bad_trained] = float('inf')
        elif kind == 'nan':
            grad[bad_trained] = float('nan')
        elif kind == 'untrained_inf' and bad_untrained is not None:
            grad[bad_untrained] = float('-inf')
        m._ensure_grad().copy_(grad)
        # the reference's optimiser only holds the trained parameters: their gradients are the only ones it checks / applies
        pt.grad = torch.where(trained, grad, torch.zeros_like(grad))
        before = pt.detach().clone()
        tsc.step(topt)
        old = tsc.get_scale()
        tsc.update()
        if old <= tsc.get_scale():
            sched.step()
        else:
            skipped += 1
        sc.step(opt, lr_decay_steps=decay_steps)
        with torch.no_grad():                                    # untrained entries never move in either optimiser
            pt.copy_(torch.where(trained, pt.detach(), before))
        n_upd = it + 1
        d = min(0.95, (1 + n_upd) / (10 + n_upd))
        shadow.sub_((1.0 - d) * (shadow - pt.detach()))
        assert float((m.arena.detach() - pt.detach()).abs().max()) < 2e-6, (it, kind)
        assert float((opt.ema - shadow).abs().max()) < 2e-6, (it, kind)
        assert float(m.arena.grad.abs().max()) == 0.0, (it, kind)                 # zeroed also on a skipped step
    assert skipped == 2 and sc.steps_skipped() == 2 and opt.steps_taken == len(kinds) - 2
    assert sc.get_scale() == tsc.get_scale()
    st = sc.state_dict()
    assert st['_growth_tracker'] == int(tsc.state_dict()['_growth_tracker'])
    assert abs(float(sc.state[12:13].view(torch.float32)[0]) - topt.param_groups[0]['lr'] / (0.1 ** (1 / decay_steps))) < 1e-7 * 1e-2 + 1e-9
    if m.table_dtype == torch.float16:
        assert torch.equal(m.half_tables(), m.arena.detach()[:m.table_elems].half())


def _occ_renderer(dev, H):
    import nerfstyle_amd.renderer as RM
    from nerfstyle_amd.common import BBox
    from nerfstyle_amd.config import NetworkConfig, RendererConfig
    from nerfstyle_amd.scene import load_room_cameras
    from nerfstyle_amd.style_nerf import StyleTCNerf
    from oracle import torch_port as TP
    ref = TP.Field(num_classes=5, table_scale=0.5)
    m = StyleTCNerf(NetworkConfig(), BBox.from_radius(2.0), 5, enc_dtype=torch.float32, use_dir=False)
    sd = m.state_dict()
    sd.update({'x_density_embedder.embeddings': ref.emb_density.detach(), 'x_color_embedder.embeddings': ref.emb_color.detach(),
               'density_net.params': ref.p_density.detach(), 'color1_net.params': ref.p_color1.detach(),
               'color2_net.params': ref.p_color2.detach(), 'class_net.params': ref.p_class.detach()})
    m.load_state_dict(sd)
    poses, intr, _ = load_room_cameras()
    cfg = RendererConfig.llff()
    cfg.grid_size = H
    r = RM.Renderer(m, cfg, intr, 2.0, raymarch_channels=8).to(dev)
    return r, ref, poses, cfg


def _oracle_cell_positions(O, H, cas, morton_idx, noise):
    """renderer.py:120-136,155: positions of cells (given by Morton index) of cascade `cas`, jitter u in [0,1)"""
    coords = O.morton3D_invert(morton_idx.astype(np.int32)).astype(np.float32)
    xn = (np.float32(2) * coords / np.float32(H - 1) - np.float32(1)).astype(np.float32)
    bound = np.float32(min(2 ** cas, 2.0))
    half = np.float32(bound / np.float32(H))
    pos = (xn * np.float32(bound - half)).astype(np.float32)
    pos = (pos + (noise.astype(np.float32) * np.float32(2) - np.float32(1)) * half).astype(np.float32)
    return pos


@pytest.mark.parametrize('H', [32, 128])
def test_device_occupancy_update_full_matches_oracle(O, dev, H):
    """Full occupancy update (renderer.py:139-160,183-189) on the device with the jitter pinned by an explicit
    noise tensor: positions bit-exact vs the restated torch expression, densities within the f16-MFMA tolerance of
    the oracle field (all cells at H=32, a 40 000-cell sample at H=128), mean within 1e-5 relative of a float64 mean
    of the device grid, bitfield EXACTLY packbits(device grid, min(mean, thresh))."""
    r, ref, poses, cfg = _occ_renderer(dev, H)
    H3 = H ** 3
    rng = np.random.default_rng(5)
    noise = rng.random((2 * H3, 3), dtype=np.float32)
    r.update_state(noise=torch.tensor(noise, device=dev))
    grid = r.density_grid.cpu().numpy()
    xyzs = r._occ_xyzs.cpu().numpy()
    assert np.array_equal(r._occ_idx.cpu().numpy(), np.arange(2 * H3, dtype=np.int32))
    sel = np.arange(2 * H3) if H == 32 else rng.choice(2 * H3, 40000, replace=False)
    fp = O.FieldParams(ref.emb_density.detach().numpy(), ref.emb_color.detach().numpy(), ref.p_density.detach().numpy(),
                       ref.p_color1.detach().numpy(), ref.p_color2.detach().numpy(), ref.p_class.detach().numpy(), ref.offsets,
                       ref.pls, num_classes=5)
    want_sigma = np.zeros(len(sel), np.float32)
    for cas in range(2):
        mk = (sel // H3) == cas
        pos = _oracle_cell_positions(O, H, cas, sel[mk] % H3, noise[sel[mk]])
        assert np.array_equal(xyzs[sel[mk]], pos)                          # bit-exact positions
        _, sig, _ = O.field_forward(fp, pos, sigma_only=True, half='f16')
        want_sigma[mk] = sig
    # density_grid starts at 0: max(0 * decay, sigma) = sigma
    assert rel_l2(grid.reshape(-1)[sel], want_sigma) < 5e-3
    mean64 = float(np.clip(grid.astype(np.float64), 0, None).mean())
    assert abs(r.mean_density - mean64) <= 1e-5 * mean64
    bits_ref = O.packbits(grid, min(np.float32(r.mean_density), np.float32(cfg.density_thresh)))
    assert np.array_equal(r.density_bitfield.cpu().numpy(), bits_ref)
    assert int(r._occ_state[0]) == 1                                      # sequence advanced on the device
    # second full update decays: grid = max(grid * 0.95, new sigma), new jitter drawn from the RNG
    before = grid.copy()
    r.update_state()
    after = r.density_grid.cpu().numpy()
    assert np.all(after >= before * np.float32(cfg.density_decay) - 1e-6)
    # generated jitter: uniform in [-1, 1) half cells around the centre, different from update to update
    centre = np.concatenate([_oracle_cell_positions(O, H, cas, np.arange(H3), np.full((H3, 3), 0.5, np.float32)) for cas in range(2)])
    half = np.repeat(np.array([1.0 / H, 2.0 / H], np.float32), H3)[:, None]
    u = (r._occ_xyzs.cpu().numpy() - centre) / half
    assert np.abs(u).max() <= 1.0 + 1e-4 and abs(float(u.mean())) < 5e-3 and abs(float(u.std()) - 0.57735) < 5e-3
    a1 = r._occ_xyzs.clone()
    r.update_state()
    assert not torch.equal(a1, r._occ_xyzs)
    # same seed + same sequence on a second renderer => the same points (rank-identical replicas)
    r2, _, _, _ = _occ_renderer(dev, H)
    r2.update_state(); r2.update_state(); r2.update_state()
    assert torch.equal(r2._occ_xyzs, r._occ_xyzs)
    r3, _, _, _ = _occ_renderer(dev, H)
    r3.update_state(); r3.update_state(); r3.update_state()
    assert torch.equal(r3.density_grid, r2.density_grid) and torch.equal(r3.density_bitfield, r2.density_bitfield)
    # a training render with the learned bitfield works end to end at this grid size
    out = r.render(torch.tensor(poses[0], device=dev), None, num_rays=None, training=True,
                   pix_subset=torch.arange(0, 4096, device=dev))
    assert torch.isfinite(out['rgb_map']).all()


def test_device_occupancy_update_partial_matches_restatement(O, dev):
    """Partial update (renderer.py:163-181): per cascade H^3/4 uniform cells + H^3/4 draws from the occupied cells;
    only sampled cells change, by max(grid * decay, sigma of one of the points that hit the cell)."""
    from nerfstyle_amd import _lib as L
    H = 32
    r, ref, poses, cfg = _occ_renderer(dev, H)
    H3, N = H ** 3, H ** 3 // 4
    r.update_state()                                         # full update first: a non-trivial grid
    # make cascade 1 sparse so that "draw from the occupied cells" is a real restriction
    g = r.density_grid.clone()
    thr = torch.quantile(g[1], 0.9)
    g[1][g[1] < thr] = 0.0
    r.density_grid.copy_(g)
    before = r.density_grid.cpu().numpy().copy()
    r.local_step = cfg.update_thres
    P = int(L.lib().nsr_occ_num_points(2, H, 0))
    assert P == 2 * 2 * N
    ws, xyzs, idx = r._occ_buffers(P)
    L.check(L.lib().nsr_occ_sample_points(L.p(r.density_grid), 2, H, 2.0, 0, 1234, 7, None, None, L.p(xyzs), L.p(idx), L.p(ws),
                                          L.stream()))
    idx_h, xyz_h = idx.cpu().numpy(), xyzs.cpu().numpy()
    for cas in range(2):
        lo = cas * 2 * N
        ii = idx_h[lo:lo + 2 * N]
        assert ii.min() >= cas * H3 and ii.max() < (cas + 1) * H3
        occ = np.flatnonzero(before[cas] > 0)
        assert np.isin(ii[N:] - cas * H3, occ).all()                       # second half: occupied cells only
        assert len(np.unique(ii[:N])) > 0.7 * min(N, H3) * (1 - np.exp(-1)) # first half: spread over the grid
        cnt = np.bincount(ii[N:] - cas * H3, minlength=H3)[occ]
        assert cnt.min() >= 0 and abs(cnt.mean() - N / len(occ)) < 1e-6     # every draw landed on an occupied cell
        centre = _oracle_cell_positions(O, H, cas, ii - cas * H3, np.full((2 * N, 3), 0.5, np.float32))
        half = min(2 ** cas, 2.0) / H
        assert np.abs(xyz_h[lo:lo + 2 * N] - centre).max() <= half * (1 + 1e-4)
    with torch.no_grad():
        sig = r.model.field(xyzs, sigma_only=True, density_scale=1.0)
    L.check(L.lib().nsr_occ_update(L.p(r.density_grid), L.p(sig), L.p(idx), P, 2, H, 0, 0.95, 10.0, L.p(r.density_bitfield),
                                   L.p(r._mean_density_dev), None, L.p(ws), L.stream()))
    after = r.density_grid.cpu().numpy().reshape(-1)
    b = before.reshape(-1)
    s_h = sig.cpu().numpy()
    touched = np.zeros(2 * H3, bool)
    touched[idx_h] = True
    assert np.array_equal(after[~touched], b[~touched])                    # tmp_grid = -1 there: unchanged, no decay
    smin = np.full(2 * H3, np.inf, np.float32)
    smax = np.full(2 * H3, -np.inf, np.float32)
    np.minimum.at(smin, idx_h, s_h)
    np.maximum.at(smax, idx_h, s_h)
    lo_v = np.maximum(b * np.float32(0.95), smin)[touched]
    hi_v = np.maximum(b * np.float32(0.95), smax)[touched]
    assert np.all(after[touched] >= lo_v) and np.all(after[touched] <= hi_v)
    # duplicates: the largest density wins (k_occ_scatter) -- deterministic on every run and rank
    assert int((smax[touched] > smin[touched]).sum()) > 100 and np.array_equal(after[touched], hi_v)
    mean64 = float(np.clip(after.astype(np.float64), 0, None).mean())
    assert abs(r.mean_density - mean64) <= 1e-5 * mean64
    assert np.array_equal(r.density_bitfield.cpu().numpy(), O.packbits(after.reshape(2, -1), min(np.float32(r.mean_density), np.float32(10.0))))
    # an empty cascade: its occupied half yields no points (index -1) instead of the reference's randint(0, 0) error
    r.density_grid.zero_()
    L.check(L.lib().nsr_occ_sample_points(L.p(r.density_grid), 2, H, 2.0, 0, 1, 0, None, None, L.p(xyzs), L.p(idx), L.p(ws), L.stream()))
    ii = idx.cpu().numpy().reshape(2, 2, N)
    assert (ii[:, 1] == -1).all() and (ii[:, 0] >= 0).all()


def test_sample_overflow_is_safe_and_reported(O, dev):
    """A sample buffer that is too small (samples_per_ray_cap): the rays that do not fit are dropped like the
    reference's mean_count path (raymarching.cu:517) -- their in-buffer samples are zeroed (never uninitialised
    memory), gradients stay finite, the overflow is reported on the device, and inference falls back to the
    reference's loop, which never drops a ray."""
    r, ref, poses, intr, bits = _setup(dev, cap=24)
    m = r.model
    n = 4096
    pix = torch.arange(0, n * 40, 40, device=dev)
    pose = torch.tensor(poses[0], device=dev)
    # poison the allocator's free blocks: torch.empty buffers of the next call come back holding NaNs
    for k in (3, 4, 8, 1):
        junk = torch.full((n * 24 * k,), float('nan'), device=dev)
        del junk
    m._ensure_grad().zero_()
    out = r.render(pose, None, training=True, pix_subset=pix)
    loss = out['rgb_map'].square().mean() + out['classes'].square().mean()
    loss.backward()
    assert bool(r.last_call_overflowed())
    assert torch.isfinite(out['rgb_map']).all() and torch.isfinite(m.arena.grad).all()
    assert float(m.arena.grad.abs().sum()) > 0
    cnt = int(r._last_counter[0])
    assert cnt >= n * 24
    # dropped rays render as background with zero class logits
    big = _setup(dev, cap=None)[0]
    full = big.render(pose, None, training=True, pix_subset=pix)
    assert not bool(big.last_call_overflowed())
    dropped = (out['rgb_map'] - full['rgb_map']).abs().amax(1) > 1e-5
    assert 0 < int(dropped.sum()) < n
    assert float((out['rgb_map'][dropped] - 1.0).abs().max()) == 0.0
    first = int(torch.nonzero(dropped)[0])
    assert not bool(dropped[:first].any())                    # rays before the first dropped one are intact
    # inference with the same cap: falls back to the loop and equals the uncapped single pass
    a = r.render(pose, None, training=False, pix_subset=pix)
    assert r.last_test_overflow
    b = big.render(pose, None, training=False, pix_subset=pix)
    assert float((a['rgb_map'] - b['rgb_map']).abs().max()) < 2e-4


def test_sparsity_term_sigma_only_forward_backward(O, dev):
    """BASELINE config 4 (fern, --sparsity_lambda 0.01): trainers/base.py:409-413 evaluates
    `self.renderer.model(sparsity_pts)` WITH autograd on 50 000 uniform points of the bbox and :285-291 adds
    mean(|1 - exp(-coeff * sigma)|) * lambda.  That is the sigma-only fused forward followed by the fused backward
    with no colour gradient: checked against autograd through the rounding-emulating restatement; the colour /
    class / colour-2 blocks and the colour table must receive exactly zero."""
    from nerfstyle_amd.style_nerf import MLP_LAYOUT
    r, ref, poses, intr, bits = _setup(dev, table_dtype=torch.float32, table_scale=0.5)
    m = r.model
    g = torch.Generator().manual_seed(11)
    n = 50000
    pts = torch.rand(n, 3, generator=g) * 4.0 - 2.0                     # rand * bbox.size + bbox.min_pt  (:411-412)
    coeff, lam, SCALE = 0.05, 0.01, 65536.0
    m._ensure_grad().zero_()
    sig = m(pts.to(dev))                                                 # StyleTCNerf.forward(pts): [n, 1]
    assert sig.shape == (n, 1) and sig.requires_grad
    loss = torch.mean(torch.abs(1 - torch.exp(-coeff * sig))) * lam
    (loss * SCALE).backward()
    grad = m.arena.grad.detach().cpu() / SCALE
    # reference gradient
    for p in ref.parameters():
        p.grad = None
    sig_ref = ref(pts, sigma_only=True, half='f16')
    loss_ref = torch.mean(torch.abs(1 - torch.exp(-coeff * sig_ref))) * lam
    loss_ref.backward()
    assert abs(float(loss) - float(loss_ref)) < 2e-3 * float(loss_ref)
    assert rel_l2(sig.detach().cpu().numpy().reshape(-1), sig_ref.detach().numpy().reshape(-1)) < 2e-3
    gt = grad[:m.table_elems].view(m.rows, 2, 2)
    assert rel_l2(gt[:, 0, :].numpy(), ref.emb_density.grad.numpy()) < 5e-3
    assert float(gt[:, 1, :].abs().max()) == 0.0                         # colour table: exactly zero
    gm = grad[m.table_elems:]
    off = {name: (o, k) for name, o, k in MLP_LAYOUT}
    o, k = off['density_net']
    assert rel_l2(gm[o:o + k].numpy(), ref.p_density.grad.numpy()) < 5e-3
    for name in ('color1_net', 'color2_net', 'class_net'):
        o, k = off[name]
        assert float(gm[o:o + k].abs().max()) == 0.0, name
    # the term composes with a render in the same backward (one optimiser step sees both)
    m.arena.grad.zero_()
    out = r.render(torch.tensor(poses[0], device=dev), None, training=True, pix_subset=torch.arange(0, 2048 * 90, 90, device=dev))
    total = out['rgb_map'].square().mean() + torch.mean(torch.abs(1 - torch.exp(-coeff * m(pts.to(dev))))) * lam
    (total * SCALE).backward()
    assert torch.isfinite(m.arena.grad).all() and float(m.arena.grad[:m.table_elems].view(m.rows, 2, 2)[:, 1].abs().max()) > 0


def test_fern_cameras_step_with_sparsity_term(O, dev):
    """BASELINE config 4 on its own cameras (nerfstyle_amd/assets/llff_fern_cameras.json, from the reference's
    datasets/nerf_llff_data/fern/transforms_train.json): a training render of 4 096 fern rays against the oracle pipeline on
    the same rays (PSNR > 45 dB), then the step's loss = fused reconstruction loss + the sparsity term
    (trainers/base.py:285-291,409-413, --sparsity_lambda 0.01) back-propagated together: the gradient equals the sum of the
    two terms' separate gradients (the sparsity term reaches the density table and density_net only)."""
    from nerfstyle_amd.common import BBox
    from nerfstyle_amd.config import NetworkConfig, RendererConfig
    from nerfstyle_amd.recon_loss import recon_loss
    from nerfstyle_amd.renderer import Renderer
    from nerfstyle_amd.scene import load_cameras
    from nerfstyle_amd.style_nerf import MLP_LAYOUT, StyleTCNerf
    from oracle import torch_port as TP
    nc = 5
    ref = TP.Field(num_classes=nc, table_scale=0.5)
    m = StyleTCNerf(NetworkConfig(), BBox.from_radius(2.0), nc, enc_dtype=torch.float32, use_dir=False)
    sd = m.state_dict()
    sd.update({'x_density_embedder.embeddings': ref.emb_density.detach(), 'x_color_embedder.embeddings': ref.emb_color.detach(),
               'density_net.params': ref.p_density.detach(), 'color1_net.params': ref.p_color1.detach(),
               'color2_net.params': ref.p_color2.detach(), 'class_net.params': ref.p_class.detach()})
    m.load_state_dict(sd)
    poses, intr, meta = load_cameras('fern')
    assert len(poses) == 17 and abs(intr.fx - 407.5657916100737) < 1e-9
    r = Renderer(m, RendererConfig.llff(), intr, 2.0, raymarch_channels=3 + nc).to(dev)
    grid, bits = small_scene()
    r.density_grid = torch.tensor(grid, device=dev)
    r.density_bitfield = torch.tensor(bits, device=dev)
    r.update_occ = False
    rng = np.random.default_rng(17)
    pix = rng.choice(intr.w * intr.h, 4096, replace=False)
    ro, rd = O.generate_rays(poses[5], intr.w, intr.h, intr.fx, intr.fy, intr.cx, intr.cy, 3, pix_indices=pix)
    rgb_o, d_o, cls_o, cnt = _oracle_render(O, ref, bits, ro, rd)
    assert cnt > 4096 * 8
    pose, pix_t = torch.tensor(poses[5], device=dev), torch.tensor(pix, device=dev)
    g = torch.Generator().manual_seed(2)
    t_rgb = torch.rand(intr.w * intr.h, 3, generator=g).to(dev)
    t_cls = torch.randint(0, nc, (intr.w * intr.h,), generator=g).to(dev)
    pts = (torch.rand(50000, 3, generator=g) * 4.0 - 2.0).to(dev)
    SCALE = 4096.0

    def step(with_render, with_sparsity):
        m._ensure_grad().zero_()
        total = 0.0
        out = None
        if with_render:
            out = r.render(pose, None, training=True, pix_subset=pix_t)
            total = recon_loss(out['rgb_map'], out['classes'], t_rgb, t_cls, pix_t, ce_lambda=1e-3, factor=SCALE)
        if with_sparsity:
            sig = m(pts)
            total = total + torch.mean(torch.abs(1 - torch.exp(-0.05 * sig))) * 0.01 * SCALE
        total.backward()
        return m.arena.grad.detach().clone() / SCALE, out
    g_both, out = step(True, True)
    rgb = out['rgb_map'].detach().cpu().numpy()
    assert O.compute_psnr(float(np.mean((rgb - rgb_o) ** 2))) > 45.0
    g_r, _ = step(True, False)
    g_s, _ = step(False, True)
    assert float(g_r.abs().sum()) > 0 and float(g_s.abs().sum()) > 0
    assert rel_l2((g_r + g_s).cpu().numpy(), g_both.cpu().numpy()) < 2e-5
    gt = g_s[:m.table_elems].view(m.rows, 2, 2)
    assert float(gt[:, 1].abs().max()) == 0.0 and float(gt[:, 0].abs().max()) > 0.0
    for name, off, n in MLP_LAYOUT:
        blk = g_s[m.table_elems + off: m.table_elems + off + n]
        assert (float(blk.abs().max()) > 0.0) == (name == 'density_net'), name


def test_checkpoint_roundtrip_on_the_gpu_is_bit_identical(O, dev, tmp_path):
    """Save (nerfstyle_amd.checkpoint, the reference's key layout) after 5 training steps with occupancy updates, device-side
    GradScaler and EMA; restore into freshly constructed objects; the restored renderer renders the same image bit for bit
    and the NEXT training step (same pixels) leaves bit-identical parameters, moments, EMA and scaler state."""
    from nerfstyle_amd import checkpoint as C
    from nerfstyle_amd.common import BBox
    from nerfstyle_amd.config import NetworkConfig, RendererConfig
    from nerfstyle_amd.optim import FusedAdam, LossScaler
    from nerfstyle_amd.recon_loss import recon_loss
    from nerfstyle_amd.renderer import Renderer
    from nerfstyle_amd.scene import load_room_cameras
    from nerfstyle_amd.style_nerf import StyleTCNerf
    poses, intr, _ = load_room_cameras()
    nc = 4

    def build():
        torch.manual_seed(0)
        m = StyleTCNerf(NetworkConfig(), BBox.from_radius(2.0), nc, enc_dtype=None, use_dir=False)
        r = Renderer(m, RendererConfig.llff(), intr, 2.0, raymarch_channels=3 + nc, samples_per_ray_cap=512).to(dev).manual_seed(9)
        return m, r, FusedAdam(m, lr=1e-2, ema_decay=0.95), LossScaler(init_scale=8192.0, growth_interval=4)
    m, r, opt, sc = build()
    g = torch.Generator().manual_seed(6)
    npix = intr.w * intr.h
    t_rgb = torch.rand(npix, 3, generator=g).to(dev)
    t_cls = torch.randint(0, nc, (npix,), generator=g).to(dev)
    pixs = [torch.randperm(npix, generator=g)[:4096].to(dev) for _ in range(6)]
    pose = torch.tensor(poses[3], device=dev)

    def train_step(m, r, opt, sc, pix):
        out = r.render(pose, None, training=True, pix_subset=pix)
        recon_loss(out['rgb_map'], out['classes'], t_rgb, t_cls, pix, scale=sc.scale_tensor(dev)).backward()
        sc.step(opt, lr_decay_steps=100.0)
    for it in range(5):
        train_step(m, r, opt, sc, pixs[it])
    f = tmp_path / 'iter_0005.pth'
    C.save_checkpoint(f, r, optim=opt, scaler=sc, iter_ctr=5)
    m2, r2, opt2, sc2 = build()
    assert C.restore(C.load_checkpoint(f), r2, optim=opt2, scaler=sc2) == 5
    assert sc2.get_scale() == sc.get_scale() == 16384.0 and opt2.step_count == 5       # grew once after 4 clean steps
    eval_pix = torch.arange(0, npix, 37, device=dev)
    with torch.no_grad():
        a = r.render(pose, None, training=False, pix_subset=eval_pix)
        b = r2.render(pose, None, training=False, pix_subset=eval_pix)
    assert torch.equal(a['rgb_map'], b['rgb_map']) and torch.equal(a['classes'], b['classes'])
    assert torch.equal(r.density_bitfield, r2.density_bitfield) and r2.local_step == r.local_step == 5
    assert torch.equal(m.arena.detach(), m2.arena.detach()) and torch.equal(opt.exp_avg, opt2.exp_avg)
    assert torch.equal(opt.exp_avg_sq, opt2.exp_avg_sq) and torch.equal(opt.ema, opt2.ema) and opt2.ema_updates == opt.ema_updates_made == 5
    train_step(m, r, opt, sc, pixs[5])
    train_step(m2, r2, opt2, sc2, pixs[5])
    # the table gradient is summed by float atomics in arrival order: a step is reproducible to fp32 rounding, not bit for bit
    d = (m.arena.detach() - m2.arena.detach()).abs()
    assert float((d > 1e-5).float().mean()) < 1e-3 and sc.state_dict() == sc2.state_dict()
    assert float((opt.exp_avg - opt2.exp_avg).abs().max()) < 1e-6 * max(1.0, float(opt.exp_avg.abs().max()))


def test_deferred_backprop_equals_direct_and_trains_only_colour_table(O, dev):
    """Stylisation glue (trainers/style.py:162-204): per-patch deferred back-propagation gives the
    gradient of one direct full-frame backward; with OPTIM_KEYS = ['x_color_embedder'] (style.py:25)
    only the colour table moves."""
    from nerfstyle_amd.common import Intrinsics
    from nerfstyle_amd.optim import FusedAdam
    from nerfstyle_amd.stylize import deferred_backprop_step, patch_list
    r, ref, poses, intr, bits = _setup(dev, cap=256)
    r.intr = Intrinsics(96, 128, intr.fx * 128 / intr.w, intr.fy * 128 / intr.w, 64., 48.)    # small frame
    pose = torch.tensor(poses[2], device=dev)
    g = torch.Generator().manual_seed(3)
    target = torch.rand(96, 128, 3, generator=g).to(dev)
    kern = torch.ones(3, 1, 5, 5, device=dev) / 25.0

    def image_loss(rgb):            # a stand-in for VGG features: box-blurred image vs target
        x = rgb.permute(2, 0, 1).unsqueeze(0)
        t = target.permute(2, 0, 1).unsqueeze(0)
        return ((torch.nn.functional.conv2d(x, kern, padding=2, groups=3) -
                 torch.nn.functional.conv2d(t, kern, padding=2, groups=3)) ** 2).mean()

    assert len(patch_list(128, 96, 50)) == 6
    m = r.model
    opt = FusedAdam(m, lr=0.1, keywords=['x_color_embedder'])      # cfgs/training/style.yaml:1
    assert m.train_color_table and not m.train_density_table
    # f16 MFMA operands: gradients of a mean-over-pixels loss underflow without loss scaling; the
    # reference trains under GradScaler (init scale 65536, trainers/base.py:228) for the same reason
    SCALE = 65536.0
    loss, _ = deferred_backprop_step(r, pose, image_loss, patch_size=50, loss_scale=SCALE)
    g_def = m.arena.grad.clone()
    m.arena.grad.zero_()
    out = r.render(pose, None, training=True)
    (image_loss(out['rgb_map'].view(96, 128, 3)) * SCALE).backward()
    g_dir = m.arena.grad.clone()
    assert float(g_dir.abs().sum()) > 0
    assert rel_l2(g_def.cpu().numpy(), g_dir.cpu().numpy()) < 2e-3
    gt = g_dir[:m.table_elems].view(m.rows, 2, 2)
    assert float(gt[:, 0, :].abs().max()) == 0.0                     # density table not scattered
    assert float(gt[:, 1, :].abs().max()) > 0.0
    before = m.arena.detach().clone()
    opt.step(grad_scale=SCALE)
    after = m.arena.detach()
    tb, ta = before[:m.table_elems].view(m.rows, 2, 2), after[:m.table_elems].view(m.rows, 2, 2)
    assert torch.equal(tb[:, 0, :], ta[:, 0, :])                     # density table untouched
    assert torch.equal(before[m.table_elems:], after[m.table_elems:])  # MLPs untouched
    assert not torch.equal(tb[:, 1, :], ta[:, 1, :])                 # colour table moved
    assert float(m.arena.grad.abs().max()) == 0.0


def test_stylisation_iteration_vgg_semantic_loss_504x378(O, dev):
    """BASELINE config 3 end to end on one GPU at the shipped LLFF resolution: StyleTrainer.run_iter
    (trainers/style.py:162-204) = full-frame pass without autograd -> VGG16 'relu3' content MSE + SemanticStyleLoss with
    style clusters and Hungarian matching (style.py:70-105, loss.py:116-214; the PyTorch losses pinned against the
    reference in tests/test_losses_cpu.py) -> cached d loss / d pixels -> 6 patch re-renders of 200x200 with autograd ->
    colour-table-only fused Adam (style.py:25).  The deferred gradient must equal one direct full-frame backward."""
    from nerfstyle_amd.losses import SemanticStyleLoss
    from nerfstyle_amd.optim import FusedAdam
    from nerfstyle_amd.stylize import StyleCriterion, deferred_backprop_step, patch_list
    from nerfstyle_amd.vgg import VGG16FeatureExtractor
    r, ref, poses, intr, bits = _setup(dev, cap=256)
    r.cfg.max_steps = 512                                     # README's stylisation command: --max_steps 512
    m = r.model
    W, H = intr.size()
    assert (W, H) == (504, 378) and len(patch_list(W, H, 200)) == 6
    g = torch.Generator().manual_seed(5)
    target = torch.rand(3, H, W, generator=g).to(dev)
    style = torch.rand(3, 378, 504, generator=g).to(dev)
    seg = torch.randint(0, 5, (378, 504), generator=g)
    fx = VGG16FeatureExtractor(['relu3']).to(dev)
    crit = StyleCriterion(fx, SemanticStyleLoss(['relu3'], clusters=seg), content_lambda=0.001, style_lambda=1.0)
    crit.init_style(style, num_classes=5)
    pose = torch.tensor(poses[4], device=dev)
    opt = FusedAdam(m, lr=0.1, keywords=['x_color_embedder'])
    SCALE = 65536.0

    def image_loss(rgb, classes):
        return crit(rgb, target, classes, frame_key=4)[0]

    loss, rgb = deferred_backprop_step(r, pose, image_loss, patch_size=200, loss_scale=SCALE, with_classes=True)
    assert crit.style_loss.matching is not None and sorted(crit.style_loss.matching) == [0, 1, 2, 3, 4]
    assert np.isfinite(float(loss)) and rgb.shape == (H, W, 3)
    g_def = m.arena.grad.clone()
    m.arena.grad.zero_()
    out = r.render(pose, None, training=True)
    (image_loss(out['rgb_map'].view(H, W, 3), out['classes'].detach().view(H, W, 5)) * SCALE).backward()
    g_dir = m.arena.grad.clone()
    assert float(g_dir.abs().sum()) > 0
    assert rel_l2(g_def.cpu().numpy(), g_dir.cpu().numpy()) < 2e-3
    gt = g_def[:m.table_elems].view(m.rows, 2, 2)
    assert float(gt[:, 0].abs().max()) == 0.0 and float(gt[:, 1].abs().max()) > 0.0
    before = m.arena.detach().clone()
    opt.step(grad_scale=SCALE)
    after = m.arena.detach()
    assert torch.equal(before[m.table_elems:], after[m.table_elems:])
    assert not torch.equal(before[:m.table_elems], after[:m.table_elems])
    # a second iteration with the updated colour table gives a different, finite loss
    loss2, _ = deferred_backprop_step(r, pose, image_loss, patch_size=200, loss_scale=SCALE, with_classes=True)
    assert np.isfinite(float(loss2)) and float(loss2) != float(loss)


def test_reconstruction_training_learns_with_occupancy_updates(O, dev):
    """The reference's whole training flow on one camera: Renderer.render(training=True) with
    update_occ ON (update_state every 16 calls on the model's own densities, renderer.py:206-207),
    MSE + 0.001*CE loss (trainers/base.py:251-304), dynamic loss scaling, fused Adam + EMA.  The
    fit of a smooth target image must improve by > 8 dB in 120 steps of 8192 rays."""
    from nerfstyle_amd.common import BBox
    from nerfstyle_amd.config import NetworkConfig, RendererConfig
    from nerfstyle_amd.optim import FusedAdam, LossScaler, exp_lr
    from nerfstyle_amd.renderer import Renderer
    from nerfstyle_amd.scene import load_room_cameras
    from nerfstyle_amd.style_nerf import StyleTCNerf
    torch.manual_seed(0)
    poses, intr, _ = load_room_cameras()
    m = StyleTCNerf(NetworkConfig(), BBox.from_radius(2.0), 4, enc_dtype=None, use_dir=False)
    r = Renderer(m, RendererConfig.llff(), intr, 2.0, raymarch_channels=7, samples_per_ray_cap=1024).to(dev)
    W, H = intr.size()
    yy, xx = torch.meshgrid(torch.arange(H, device=dev), torch.arange(W, device=dev), indexing='ij')
    target = torch.stack([xx / W, yy / H, 0.5 + 0.5 * torch.sin(xx / 40.0) * torch.cos(yy / 30.0)], -1).reshape(-1, 3).float()
    tcls = ((xx >= W // 2).long() + 2 * (yy >= H // 2).long()).reshape(-1)
    pose = torch.tensor(poses[0], device=dev)
    opt = FusedAdam(m, lr=1e-2, betas=(0.9, 0.999), eps=1e-15, ema_decay=0.95)
    scaler = LossScaler()
    eval_pix = torch.arange(0, W * H, 23, device=dev)

    def eval_psnr():
        with torch.no_grad():
            out = r.render(pose, None, training=False, pix_subset=eval_pix)
        return float(-10 * torch.log10(torch.mean((out['rgb_map'] - target[eval_pix]) ** 2)))

    g = torch.Generator(device=dev)
    g.manual_seed(1)
    psnr0 = None
    for it in range(120):
        pix = torch.randperm(W * H, device=dev, generator=g)[:8192]
        out = r.render(pose, None, training=True, pix_subset=pix)
        loss = torch.mean((out['rgb_map'] - target[pix]) ** 2) + 1e-3 * torch.nn.functional.cross_entropy(out['classes'], tcls[pix])
        scaler.scale(loss).backward()
        scaler.step(opt, lr_decay_steps=30000)
        if it == 0:
            psnr0 = eval_psnr()
            assert r.local_step == 1 and int(r.density_bitfield.count_nonzero()) > 0      # update_state ran
    assert r.local_step == 120 and r.mean_density > 0
    psnr1 = eval_psnr()
    assert np.isfinite(psnr1) and psnr1 > psnr0 + 8.0, (psnr0, psnr1)
    with torch.no_grad():
        out = r.render(pose, None, training=False, pix_subset=eval_pix)
    acc = float((out['classes'].argmax(1) == tcls[eval_pix]).float().mean())
    assert acc > 0.5, acc                                # 4 quadrant classes, chance = 0.25


def test_graph_captured_step_equals_eager_step(O, dev):
    """hipGraph capture of render + loss + backward (nerfstyle_amd/graph.py): replaying the graph with
    new pose / pixel contents gives the eager step's loss and gradients (tolerance: the table-gradient
    atomics are order-free, fp32 sums differ in the last bits)."""
    from nerfstyle_amd.graph import GraphedRenderStep
    r, ref, poses, intr, bits = _setup(dev, cap=192)
    m = r.model
    n = 2048
    g = torch.Generator(device=dev)
    g.manual_seed(3)
    target = torch.rand(intr.w * intr.h, 3, device=dev, generator=g)

    def loss_fn(out, pix):
        return torch.mean((out['rgb_map'] - target[pix]) ** 2) + 1e-3 * out['classes'].square().mean()

    step = GraphedRenderStep(r, n, loss_fn)
    pose_t = torch.tensor(poses, device=dev)
    pix0 = torch.randperm(intr.w * intr.h, device=dev, generator=g)[:n]
    step.capture(pose_t[0], pix0)
    for k in (1, 5):                                  # replays with contents that differ from the capture
        pix = torch.randperm(intr.w * intr.h, device=dev, generator=g)[:n]
        m.arena.grad.zero_()
        loss_g = step(pose_t[k], pix).clone()
        grad_g = m.arena.grad.clone()
        cnt_g = step.counter.clone()
        m.arena.grad.zero_()
        out = r.render(pose_t[k], None, training=True, pix_subset=pix)
        loss_e = loss_fn(out, pix)
        loss_e.backward()
        assert int(cnt_g[0]) == int(r._last_counter[0]) > n
        assert abs(float(loss_g) - float(loss_e.detach())) <= 1e-6 * abs(float(loss_e.detach()))
        assert float(grad_g.abs().max()) > 0
        assert rel_l2(grad_g.cpu().numpy(), m.arena.grad.cpu().numpy()) < 1e-4


def test_bf16_graph_replayed_training_step(O, dev):
    """BASELINE config 5's render step on one GPU: bf16 MLP operands + fp32 composite, render + loss + backward replayed as
    one captured hipGraph, occupancy updates (device-side, no host read) between replays, fused Adam.  A replayed step
    equals the eager bf16 step, and the bf16 render clears the 40 dB bar against the fp32 oracle pipeline."""
    from nerfstyle_amd.common import BBox
    from nerfstyle_amd.config import NetworkConfig, RendererConfig
    from nerfstyle_amd.graph import GraphedRenderStep
    from nerfstyle_amd.optim import FusedAdam
    from nerfstyle_amd.renderer import Renderer
    from nerfstyle_amd.scene import load_room_cameras
    from nerfstyle_amd.style_nerf import StyleTCNerf
    from oracle import torch_port as TP
    nc = 5
    ref = TP.Field(num_classes=nc, table_scale=0.5)
    m = StyleTCNerf(NetworkConfig(), BBox.from_radius(2.0), nc, enc_dtype=torch.float32, use_dir=False, compute_dtype=torch.bfloat16)
    sd = m.state_dict()
    sd.update({'x_density_embedder.embeddings': ref.emb_density.detach(), 'x_color_embedder.embeddings': ref.emb_color.detach(),
               'density_net.params': ref.p_density.detach(), 'color1_net.params': ref.p_color1.detach(),
               'color2_net.params': ref.p_color2.detach(), 'class_net.params': ref.p_class.detach()})
    m.load_state_dict(sd)
    poses, intr, _ = load_room_cameras()
    r = Renderer(m, RendererConfig.llff(), intr, 2.0, raymarch_channels=3 + nc, samples_per_ray_cap=192).to(dev)
    grid, bits = small_scene()
    r.density_grid = torch.tensor(grid, device=dev)
    r.density_bitfield = torch.tensor(bits, device=dev)
    r.update_occ = False
    # ---- PSNR of the bf16 render vs the fp32 oracle pipeline -------------------------------------
    np.random.seed(1)
    pix = np.random.choice(intr.w * intr.h, 4096, replace=False)
    ro, rd = O.generate_rays(poses[0], intr.w, intr.h, intr.fx, intr.fy, intr.cx, intr.cy, 3, pix_indices=pix)
    rgb_o, _, _, cnt = _oracle_render(O, ref, bits, ro, rd)
    with torch.no_grad():
        out = r.render(torch.tensor(poses[0], device=dev), None, training=True, pix_subset=torch.tensor(pix, device=dev))
    psnr = O.compute_psnr(float(np.mean((out['rgb_map'].cpu().numpy() - rgb_o) ** 2)))
    assert cnt > 4096 * 8 and psnr > 40.0, psnr
    # ---- graph replay == eager, bf16 ----------------------------------------------------------------
    n = 4096
    g = torch.Generator(device=dev)
    g.manual_seed(3)
    target = torch.rand(intr.w * intr.h, 3, device=dev, generator=g)

    def loss_fn(o, p):
        return torch.mean((o['rgb_map'] - target[p]) ** 2) + 1e-3 * o['classes'].square().mean()

    step = GraphedRenderStep(r, n, loss_fn)
    pose_t = torch.tensor(poses, device=dev)
    step.capture(pose_t[0], torch.randperm(intr.w * intr.h, device=dev, generator=g)[:n])
    for k in (2, 7):
        p = torch.randperm(intr.w * intr.h, device=dev, generator=g)[:n]
        m.arena.grad.zero_()
        loss_g = step(pose_t[k], p).clone()
        grad_g = m.arena.grad.clone()
        m.arena.grad.zero_()
        o = r.render(pose_t[k], None, training=True, pix_subset=p)
        loss_e = loss_fn(o, p)
        loss_e.backward()
        assert abs(float(loss_g) - float(loss_e.detach())) <= 1e-6 * abs(float(loss_e.detach()))
        assert float(grad_g.abs().max()) > 0 and rel_l2(grad_g.cpu().numpy(), m.arena.grad.cpu().numpy()) < 1e-4
    # ---- with occupancy updates on: they run between replays on the device, the graph keeps replaying ---------------
    r.update_occ = True
    r.local_step = 0
    step2 = GraphedRenderStep(r, n, loss_fn)
    opt = FusedAdam(m, lr=1e-3)
    m.arena.grad.zero_()
    losses = []
    for it in range(18):                      # updates at steps 0 and 16
        p = torch.randperm(intr.w * intr.h, device=dev, generator=g)[:n]
        losses.append(float(step2(pose_t[it % 8], p)))
        opt.step()
    assert r.local_step == 18 and int(r._occ_state[0]) == 2 and np.all(np.isfinite(losses))
    assert int(r.density_bitfield.count_nonzero()) > 0


def test_graph_with_optimiser_inside_equals_eager_steps(O, dev):
    """GraphedRenderStep(optimizer=..., scaler=...): the optimiser step -- inf/nan check, GradScaler policy, LambdaLR value, Adam
    bias corrections and torch_ema's decay schedule, all device-side scalars -- is part of the replayed graph.  Five replayed
    steps on a dense 160 x 128 patch (spatial sample order + lattice scatter inside the graph, f16 tables and MFMA, loss scale
    with growth_interval 2 so the scale changes between replays, one step with an inf in the gradient path via an inf target)
    leave the same parameters, EMA shadow, scale and step counts as the same five steps run eagerly."""
    from nerfstyle_amd.common import BBox
    from nerfstyle_amd.config import NetworkConfig, RendererConfig
    from nerfstyle_amd.graph import GraphedRenderStep
    from nerfstyle_amd.optim import FusedAdam, LossScaler
    from nerfstyle_amd.recon_loss import recon_loss
    from nerfstyle_amd.renderer import Renderer
    from nerfstyle_amd.scene import load_room_cameras
    from nerfstyle_amd.style_nerf import StyleTCNerf
    poses, intr, _ = load_room_cameras()
    nc = 5
    W = intr.w
    rows, cols = torch.arange(100, 228, device=dev), torch.arange(50, 210, device=dev)
    pix = (rows[:, None] * W + cols[None, :]).reshape(-1)                  # 20 480 dense pixels
    g = torch.Generator(device=dev)
    g.manual_seed(12)
    t_rgb = torch.rand(intr.w * intr.h, 3, device=dev, generator=g)
    t_cls = torch.randint(0, nc, (intr.w * intr.h,), device=dev, generator=g)
    pose_t = torch.tensor(poses, device=dev)
    grid, bits = small_scene()

    def build():
        m = StyleTCNerf(NetworkConfig(), BBox.from_radius(2.0), nc, enc_dtype=None, use_dir=False)
        with torch.no_grad():
            m.arena[:m.table_elems].uniform_(-0.3, 0.3, generator=torch.Generator().manual_seed(1))
            m.arena.add_(0)
        r = Renderer(m, RendererConfig.llff(), intr, 2.0, raymarch_channels=3 + nc, samples_per_ray_cap=192).to(dev)
        r.density_grid = torch.tensor(grid, device=dev)
        r.density_bitfield = torch.tensor(bits, device=dev)
        r.update_occ = False
        return m, r, FusedAdam(m, lr=1e-2, ema_decay=0.95), LossScaler(init_scale=256.0, growth_interval=2)
    targets = [t_rgb, t_rgb, t_rgb.clone(), t_rgb, t_rgb]
    targets[2][pix[::5]] = float('inf')                                     # step 2: inf loss gradients on every 5th ray (about a
                                                                            # third of the rays carry samples) -> skipped, back-off
    cur = {'t': t_rgb.clone()}

    def run(graphed):
        m, r, opt, sc = build()
        assert r._use_spatial_order(pix.numel(), True)

        def loss_fn(o, p):
            return recon_loss(o['rgb_map'], o['classes'], cur['t'], t_cls, p, scale=sc.scale_tensor(dev))
        step = GraphedRenderStep(r, pix.numel(), loss_fn, dense=True, optimizer=opt, scaler=sc, lr_decay_steps=50.0,
                                 prefetch=graphed == 'prefetch') if graphed else None
        P = [pose_t[i] for i in range(6)]
        for it in range(5):
            cur['t'].copy_(targets[it])                                    # a static buffer: replays read the new contents
            if graphed == 'prefetch':
                # the next step's pose and pixels are announced; step 3 arrives as a DIFFERENT tensor object than announced
                # (same values): the staged samples are not trusted, the step marches eagerly first
                step(P[it] if it != 3 else pose_t[3], pix, P[it + 1], pix)
            elif graphed:
                step(pose_t[it], pix)
            else:
                out = r.render(pose_t[it], None, training=True, pix_subset=pix, dense=True)
                loss_fn(out, pix).backward()
                opt.step(scaler=sc, lr_decay_steps=50.0)
        return m.arena.detach().clone(), opt.ema.clone(), sc.state_dict(), opt.steps_taken, opt.ema_updates_made
    a_e, ema_e, sc_e, steps_e, n_e = run(False)
    a_g, ema_g, sc_g, steps_g, n_g = run(True)
    assert sc_e == sc_g and sc_e['skipped'] == 1 and steps_e == steps_g == 4 and n_e == n_g == 5
    assert sc_e['scale'] == 256.0 * 2 * 0.5 * 2                             # grew after 2 clean steps, backed off, grew again
    d = (a_g - a_e).abs()
    assert float((d > 1e-5).float().mean()) < 2e-3                          # fp32 atomics order: a few near-zero gradients flip sign
    assert float((ema_g - ema_e).abs().max()) < 0.05 and float((ema_g - a_g).abs().max()) > 1e-4
    # prefetch: the next step's ray generation + march + sample order on a second branch of the graph, beside the optimiser
    a_p, ema_p, sc_p, steps_p, n_p = run('prefetch')
    assert sc_p == sc_e and steps_p == 4 and n_p == 5
    assert float(((a_p - a_e).abs() > 1e-5).float().mean()) < 2e-3 and float((ema_p - ema_e).abs().max()) < 0.05


def test_graphed_patch_backward_equals_eager(O, dev):
    """stylize.deferred_backprop_step with patch_graphs (graph.GraphedPatchBackward: one hipGraph per patch shape) leaves
    the same colour-table gradient as the eager patch loop (trainers/style.py:189-198), on two frames -- the second
    iteration only replays.  504x378 frame, 200x200 patches: 6 patches of 4 different shapes."""
    from nerfstyle_amd.optim import FusedAdam
    from nerfstyle_amd.stylize import deferred_backprop_step
    r, ref, poses, intr, bits = _setup(dev, cap=256)
    r.cfg.max_steps = 512
    m = r.model
    FusedAdam(m, lr=0.1, keywords=['x_color_embedder'])         # colour table only, as in the stylisation stage
    W, H = intr.size()
    tgt = torch.rand(H, W, 3, generator=torch.Generator().manual_seed(5)).to(dev)

    def image_loss(rgb):
        return torch.mean((rgb - tgt) ** 2)

    def run(graphs, pose):
        m._ensure_grad()
        m.arena.grad.zero_()
        deferred_backprop_step(r, pose, image_loss, patch_size=200, loss_scale=1024.0, patch_graphs=graphs)
        return m.arena.grad.clone()

    from nerfstyle_amd.graph import GraphedPatchBackward
    graphs, serial = {}, {'concurrent': False}
    for it in (0, 3):
        pose = torch.tensor(poses[it], device=dev)
        g_eager = run(None, pose)
        g_graph = run(graphs, pose)            # replays alternate between the caller's stream and a side stream
        g_serial = run(serial, pose)           # one stream, one graph per shape
        assert float(g_eager.abs().sum()) > 0
        assert rel_l2(g_graph.cpu().numpy(), g_eager.cpu().numpy()) < 1e-5, it
        assert rel_l2(g_serial.cpu().numpy(), g_eager.cpu().numpy()) < 1e-5, it
        gt = g_graph[:m.table_elems].view(m.rows, 2, 2)
        assert float(gt[:, 0].abs().max()) == 0.0 and float(gt[:, 1].abs().max()) > 0.0
    # the whole frame rendered ONCE with autograd (activations resident), no deferral: the same gradient
    from nerfstyle_amd.stylize import resident_backprop_step
    pose = torch.tensor(poses[3], device=dev)
    m.arena.grad.zero_()
    l_res, rgb_res = resident_backprop_step(r, pose, image_loss, loss_scale=1024.0)
    g_res = m.arena.grad.clone()
    l_def, rgb_def = (lambda: (m.arena.grad.zero_(), deferred_backprop_step(r, pose, image_loss, patch_size=200, loss_scale=1024.0))[1])()
    assert torch.equal(rgb_res, rgb_def) and float(l_res) == float(l_def)
    assert rel_l2(g_res.cpu().numpy(), g_eager.cpu().numpy()) < 1e-5
    # 6 patches of 4 shapes: one graph per (shape, stream slot) they fell on / per shape
    assert sum(isinstance(v, GraphedPatchBackward) for v in graphs.values()) == 6
    assert sum(isinstance(v, GraphedPatchBackward) for v in serial.values()) == 4


def test_style_criterion_autocast_close_to_fp32(dev):
    """StyleCriterion(amp_dtype=f16 / bf16) -- the reference's enable_amp (style.py:182-184) -- against the fp32 criterion on
    the same 504x378 frame.  Whole loss: value within 1 % (bf16: 3 %), gradient finite and of the same size (on random images and random
    VGG weights the style features are near-equidistant, so most nearest-neighbour choices flip in half precision and the
    gradient DIRECTION of the style term is not comparable).  Content term alone (no matching): gradient cosine > 0.99."""
    from nerfstyle_amd.losses import SemanticStyleLoss
    from nerfstyle_amd.stylize import StyleCriterion
    from nerfstyle_amd.vgg import VGG16FeatureExtractor
    g = torch.Generator().manual_seed(11)
    H, W = 378, 504
    target = torch.rand(3, H, W, generator=g).to(dev)
    style = torch.rand(3, H, W, generator=g).to(dev)
    seg = torch.randint(0, 5, (H, W), generator=g)
    classes = torch.rand(H, W, 5, generator=g).to(dev)
    rgb0 = torch.rand(H, W, 3, generator=g).to(dev)
    fx = VGG16FeatureExtractor(['relu3']).to(dev)

    def run(amp, style_lambda):
        crit = StyleCriterion(fx, SemanticStyleLoss(['relu3'], clusters=seg, matching=[0, 1, 2, 3, 4]), content_lambda=0.001,
                              style_lambda=style_lambda, amp_dtype=amp)
        crit.init_style(style, num_classes=5)
        rgb = rgb0.clone().requires_grad_(True)
        loss = crit(rgb, target, classes)[0]
        (loss * 65536.0).backward()             # the GradScaler's initial scale (style.py:186): f16 activations' gradients underflow without it
        assert loss.dtype == torch.float32 and rgb.grad.dtype == torch.float32
        return float(loss.detach()), rgb.grad.flatten().double() / 65536.0

    for style_lambda in (1.0, 0.0):
        l32, g32 = run(None, style_lambda)
        for amp, tol_cos, tol_loss in ((torch.float16, 0.99, 1e-2), (torch.bfloat16, 0.95, 3e-2)):
            l, gv = run(amp, style_lambda)
            assert abs(l - l32) < tol_loss * abs(l32), (amp, style_lambda, l, l32)
            assert bool(torch.isfinite(gv).all()) and 0.7 < float(gv.norm() / g32.norm()) < 1.4, (amp, style_lambda)
            if style_lambda == 0.0:
                cos = float(torch.dot(gv, g32) / (gv.norm() * g32.norm()))
                assert cos > tol_cos, (amp, cos)


def test_sorts_on_two_streams_do_not_disturb_each_other(O, dev):
    """Two training renders with the spatial sample order issued back to back on two streams (each sort has its own
    stream-ordered workspace: no shared scratch, no null-stream memset as with round 2's library sort) give the results of the
    same renders run one after the other."""
    r, ref, poses, intr, bits = _setup(dev, cap=256)
    m = r.model
    r.sort_samples = True
    g = torch.Generator().manual_seed(9)
    pix = torch.randperm(intr.w * intr.h, generator=g)[:30000].to(dev)
    pa, pb = torch.tensor(poses[2], device=dev), torch.tensor(poses[3], device=dev)
    with torch.no_grad():
        ref_a = r.render(pa, None, training=True, pix_subset=pix)['rgb_map'].clone()
        ref_b = r.render(pb, None, training=True, pix_subset=pix)['rgb_map'].clone()
    torch.cuda.synchronize()
    side = torch.cuda.Stream(device=dev)
    side.wait_stream(torch.cuda.current_stream())
    with torch.enable_grad():
        with torch.cuda.stream(side):
            out_b = r.render(pb, None, training=True, pix_subset=pix)['rgb_map']
        out_a = r.render(pa, None, training=True, pix_subset=pix)['rgb_map']
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    assert torch.equal(out_a.detach(), ref_a) and torch.equal(out_b.detach(), ref_b)


def test_begin_train_on_a_side_stream_equals_plain_render(O, dev):
    """Renderer.begin_train_on (ray generation + march + compaction + sample order on a side stream) + finish_train ==
    render(training=True): bit-identical outputs and sample counts, also while the main stream is busy and with the side
    stream's buffers recycled over several rounds (they are allocated by one stream and released by another)."""
    r, ref, poses, intr, bits = _setup(dev, cap=192)
    r.sort_samples = True
    side = torch.cuda.Stream(device=dev)
    g = torch.Generator(device=dev)
    g.manual_seed(3)
    busy = torch.rand(4096, 4096, device=dev, generator=g)
    for it in range(4):
        pix = torch.randperm(intr.w * intr.h, device=dev, generator=g)[:30000]
        pose = torch.tensor(poses[it], device=dev)
        want = r.render(pose, None, training=True, pix_subset=pix)
        want_cnt = r._last_counter.clone()
        ev = torch.cuda.Event()
        ev.record()
        for _ in range(3):
            busy = busy @ busy * 1e-3                                         # main-stream work the side stream runs beside
        ctx = r.begin_train_on(side, pose, pix, after=ev)
        out = r.finish_train(ctx)
        for k in ('rgb_map', 'trans_map', 'classes'):
            assert torch.equal(out[k], want[k]), (it, k)
        assert torch.equal(r._last_counter, want_cnt)
        out['rgb_map'].sum().backward()                                      # the backward walks the side stream's permutation
        del ctx, out, want
    # two contexts rewritten in place in turn (no allocation per step), with the renderer's own step bookkeeping running:
    # the occupancy schedule's step counter and the count ring advance exactly as on the allocating path
    r.update_occ = True
    r.local_step = 1                                                         # (not a multiple of update_iter: no update is due)
    stage = [None, None]
    pix = torch.randperm(intr.w * intr.h, device=dev, generator=g)[:30000]
    for it in range(4):
        pose = torch.tensor(poses[it + 4], device=dev)
        step0 = r.local_step
        want = r.render(pose, None, training=True, pix_subset=pix)
        want_cnt = r._last_counter.clone()
        assert r.local_step == step0 + 1
        ev = torch.cuda.Event()
        ev.record()
        stage[it % 2] = r.begin_train_on(side, pose, pix, after=ev, into=stage[it % 2])
        assert r.local_step == step0 + 2
        out = r.finish_train(stage[it % 2])
        for k in ('rgb_map', 'trans_map', 'classes'):
            assert torch.equal(out[k], want[k]), (it, k)
        assert torch.equal(r._last_counter, want_cnt)
        assert torch.equal(r.step_counter[(step0 + 1) % r.step_counter.shape[0]], want_cnt)
        out['rgb_map'].sum().backward()
    r.update_occ = False
