"""CPU tests of the oracle: golden vectors captured from the importable reference Python
(tests/golden/make_goldens.py) and the known-answer properties the reference code implies
(SURVEY.md section 4).  No GPU, no product kernels."""
import numpy as np
import pytest
import torch

from helpers import rel_l2, room_cameras, room_rays, small_scene


# ---- goldens from the reference's importable Python ----------------------------------------------
def test_generate_rays_matches_reference(O, golden):
    c = room_cameras()
    pose = golden['pose0']
    args = (pose, c['w'], c['h'], c['fl_x'], c['fl_y'], c['cx'], c['cy'], 3)
    o, d = O.generate_rays(*args)
    sel = golden['rays_full_sel']
    assert np.array_equal(o[sel], golden['rays_full_o'])
    assert np.abs(d[sel] - golden['rays_full_d']).max() <= 1.2e-7      # 1 ulp of a unit vector (fp32 FMA order)
    o, d = O.generate_rays(*args, patch=(200, 0, 200, 200))
    assert np.abs(d[golden['rays_patch_sel']] - golden['rays_patch_d']).max() <= 1.2e-7
    o, d = O.generate_rays(*args, pix_indices=golden['rays_rand_idx'])
    assert np.abs(d[:512] - golden['rays_rand_d']).max() <= 1.2e-7
    assert np.array_equal(o[:512], golden['rays_rand_o'])


def test_integrate_points_port_matches_reference(golden):
    from oracle import torch_port as TP
    N = golden['ip_dists'].shape[0]
    rgb, acc, trans = TP.integrate_points(
        torch.tensor(golden['ip_dists']), torch.tensor(golden['ip_rgbs']), torch.tensor(golden['ip_dens']),
        torch.zeros(N, 3), torch.zeros(N, 1), torch.ones(N, 1))
    assert np.allclose(rgb.numpy(), golden['ip_rgb_map'], atol=1e-6)
    assert np.allclose(acc.numpy(), golden['ip_acc_map'], atol=1e-6)
    assert np.allclose(trans.numpy(), golden['ip_trans_map'], atol=1e-6)
    # acc + trans == 1
    assert np.abs(golden['ip_acc_map'] + golden['ip_trans_map'] - 1).max() < 1e-6


def test_composite_oracle_equals_reference_integrate_points(O, golden):
    """raymarching.cu's composite on K samples per ray == nerf_lib.integrate_points (the reference's
    own pure-PyTorch integrator) when nothing is early-stopped."""
    dists, rgbs, dens = golden['ip_dists'], golden['ip_rgbs'], golden['ip_dens']
    N, K = dists.shape
    deltas = np.zeros((N * K + 1, 4), np.float32)
    deltas[:N * K, 0] = dists.reshape(-1)
    deltas[:N * K, 1] = dists.reshape(-1)
    rays = np.stack([np.arange(N), np.arange(N) * K, np.full(N, K)], 1).astype(np.int32)
    sig = np.concatenate([dens.reshape(-1), [0]]).astype(np.float32)
    rgb = np.concatenate([rgbs.reshape(-1, 3), np.zeros((1, 3))]).astype(np.float32)
    ws, depth, image = O.composite_rays_train_forward(sig, rgb, deltas, rays, T_thresh=0.0)
    assert np.allclose(image, golden['ip_rgb_map'], atol=2e-6)
    assert np.allclose(ws[:, None], golden['ip_acc_map'], atol=2e-6)


def test_bbox_truncexp_psnr(O, golden):
    from oracle import torch_port as TP
    assert np.array_equal(O.bbox_normalize(golden['bbox_pts'], [-2] * 3, [2] * 3), golden['bbox_norm'])
    x = torch.tensor(golden['texp_x'], requires_grad=True)
    y = TP.TruncExp.apply(x)
    y.backward(torch.ones_like(y) * 0.5)
    assert np.array_equal(y.detach().numpy(), golden['texp_y'])
    assert np.array_equal(x.grad.numpy(), golden['texp_gx'])
    assert abs(O.compute_psnr(float(golden['psnr_in'])) - float(golden['psnr_out'])) < 1e-4


# ---- known-answer properties -----------------------------------------------------------------------
def test_morton_known_answers_and_roundtrip(O):
    assert list(O.morton3D(np.array([[1, 0, 0], [0, 1, 0], [0, 0, 1], [1, 1, 1], [2, 0, 0]]))) == [1, 2, 4, 7, 8]
    rng = np.random.default_rng(0)
    c = rng.integers(0, 1024, size=(5000, 3)).astype(np.int32)
    assert np.array_equal(O.morton3D_invert(O.morton3D(c)), c)
    # independent bit-interleave
    ref = np.zeros(len(c), np.int64)
    for b in range(10):
        ref |= ((c[:, 0] >> b) & 1).astype(np.int64) << (3 * b)
        ref |= ((c[:, 1] >> b) & 1).astype(np.int64) << (3 * b + 1)
        ref |= ((c[:, 2] >> b) & 1).astype(np.int64) << (3 * b + 2)
    assert np.array_equal(O.morton3D(c).astype(np.int64), ref)


def test_packbits_bit_order_and_strictness(O):
    g = np.zeros(16, np.float32)
    g[0] = 1.0; g[3] = 0.5; g[9] = 2.0
    assert list(O.packbits(g, 0.5)) == [0b00000001, 0b00000010]     # strict '>' drops the 0.5 cell
    rng = np.random.default_rng(1)
    g = rng.random(8 * 1000).astype(np.float32)
    bits = O.packbits(g, 0.3)
    assert np.array_equal(bits, np.packbits((g.reshape(-1, 8) > 0.3)[:, ::-1], axis=1).reshape(-1))


def test_aabb_miss_and_min_near(O):
    aabb = np.array([-2, -2, -2, 2, 2, 2], np.float32)
    o = np.array([[5, 5, 5], [0, 0, 0], [0, 0, -5]], np.float32)
    d = np.array([[1, 0.1, 0.1], [0, 0, 1], [0.001, 0.002, 1]], np.float32)
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    near, far = O.near_far_from_aabb(o, d, aabb, 0.2)
    fmax = np.finfo(np.float32).max
    assert near[0] == fmax and far[0] == fmax
    assert near[1] == np.float32(0.2) and abs(far[1] - 2) < 1e-6
    assert abs(near[2] - 3) < 1e-4 and abs(far[2] - 7) < 1e-3


def test_march_all_ones_all_zero(O):
    ro, rd = room_rays(O, 256, seed=3)
    aabb = np.array([-2, -2, -2, 2, 2, 2], np.float32)
    near, far = O.near_far_from_aabb(ro, rd, aabb, 0.2)
    ones = np.full(2 * 128 ** 3 // 8, 255, np.uint8)
    xyzs, dirs, deltas, rays, counter = O.march_rays_train(ro, rd, 2.0, ones, 2, 128, near, far, 1024)
    dt_min = np.float32(2 * np.sqrt(3) / 1024)
    hit = near < np.finfo(np.float32).max
    expect = np.minimum(1024, np.ceil((far - near) / dt_min)).astype(np.int64) * hit
    assert np.abs(rays[:, 2] - expect).max() <= 1            # fp32 accumulation of t can add/drop one step
    m = int(counter[0])
    assert np.allclose(deltas[:m, 0], dt_min)
    assert np.allclose(deltas[:m, 1], dt_min, atol=1e-6)
    assert counter[1] == 256 and np.array_equal(rays[:, 0], np.arange(256))
    assert np.array_equal(rays[:, 1], np.concatenate([[0], np.cumsum(rays[:-1, 2])]))
    zeros = np.zeros_like(ones)
    xyzs, dirs, deltas, rays, counter = O.march_rays_train(ro, rd, 2.0, zeros, 2, 128, near, far, 1024, align=128)
    assert counter[0] == 0 and np.all(rays[:, 2] == 0)
    ws, depth, image = O.composite_rays_train_forward(np.zeros(len(xyzs), np.float32), np.zeros((len(xyzs), 8), np.float32),
                                                      deltas, rays)
    rgb, _, _ = O.render_epilogue(ws, depth, image, near, far)
    assert np.all(rgb == 1.0)                                 # white pixel


def test_composite_properties_and_backward(O):
    rng = np.random.default_rng(5)
    N, K, C = 64, 40, 8
    counts = rng.integers(0, K, size=N)
    offs = np.concatenate([[0], np.cumsum(counts[:-1])])
    M = int(counts.sum()) + 128
    rays = np.stack([rng.permutation(N), offs, counts], 1).astype(np.int32)
    sig = (rng.random(M) * 30).astype(np.float32)
    rgb = rng.random((M, C)).astype(np.float32)
    deltas = np.zeros((M, 4), np.float32)
    deltas[:, 0] = 0.0034
    deltas[:, 1] = 0.0034
    ws, depth, image = O.composite_rays_train_forward(sig, rgb, deltas, rays, T_thresh=0.0)
    for n in range(N):
        idx, off, cnt = rays[n]
        alpha = 1 - np.exp(-sig[off:off + cnt] * 0.0034)
        assert abs(ws[idx] + np.prod(1 - alpha) - 1) < 1e-5    # weights_sum + prod(1 - alpha) == 1
    # backward == finite differences of the forward (no early stop)
    gws = rng.standard_normal(N).astype(np.float32)
    gim = rng.standard_normal((N, C)).astype(np.float32)
    gs, gr = O.composite_rays_train_backward(gws, gim, sig, rgb, deltas, rays, ws, image, T_thresh=0.0)

    def loss(s, r):
        w, _, im = O.composite_rays_train_forward(s, r, deltas, rays, T_thresh=0.0)
        return float((w.astype(np.float64) * gws).sum() + (im.astype(np.float64) * gim).sum())
    for m in rng.choice(int(counts.sum()), 12, replace=False):
        e = np.zeros_like(sig); e[m] = 1e-2
        fd = (loss(sig + e, rgb) - loss(sig - e, rgb)) / 2e-2
        assert abs(fd - gs[m]) <= 2e-2 * max(1.0, abs(gs[m]))
    # last-sample asymmetry (:862 vs :961): with an early stop the crossing sample is accumulated
    # forward but receives no gradient
    sig2 = np.full(M, 2000.0, np.float32)
    ws2, _, im2 = O.composite_rays_train_forward(sig2, rgb, deltas, rays, T_thresh=1e-4)
    gs2, gr2 = O.composite_rays_train_backward(gws, gim, sig2, rgb, deltas, rays, ws2, im2, T_thresh=1e-4)
    n0 = int(np.argmax(counts > 3))
    off = rays[n0, 1]
    assert np.all(gr2[off + 1:off + rays[n0, 2]] == 0) and gs2[off + 1] == 0


def test_inference_composite_matches_train_composite(O):
    rng = np.random.default_rng(7)
    N, K, C = 32, 8, 8
    sig = (rng.random(N * K) * 20).astype(np.float32)
    rgb = rng.random((N * K, C)).astype(np.float32)
    deltas = np.zeros((N * K + 1, 4), np.float32)
    deltas[:, 0] = 0.0034; deltas[:, 1] = 0.0034
    rays = np.stack([np.arange(N), np.arange(N) * K, np.full(N, K)], 1).astype(np.int32)
    ws_t, d_t, im_t = O.composite_rays_train_forward(np.append(sig, 0).astype(np.float32), np.vstack([rgb, np.zeros((1, C), np.float32)]),
                                                     deltas, rays, T_thresh=0.0)
    alive = np.arange(N, dtype=np.int32)
    rays_t = np.zeros((N, 1), np.float32)
    ws = np.zeros(N, np.float32); depth = np.zeros(N, np.float32); image = np.zeros((N, C), np.float32)
    O.composite_rays(N, K, alive, rays_t, sig, rgb, deltas[:N * K], ws, depth, image, T_thresh=0.0)
    assert np.allclose(ws, ws_t, atol=1e-5) and np.allclose(image, im_t, atol=1e-5) and np.allclose(depth, d_t, atol=1e-5)
    assert np.all(alive >= 0) and np.allclose(rays_t[:, 0], K * 0.0034, atol=1e-5)


# ---- hash grid ---------------------------------------------------------------------------------------
def _grid_setup(O, rng, B=500):
    pls = O.per_level_scale_from_cfg()
    off = O.grid_offsets(16, pls, 16, 19, True)
    emb = ((rng.random((int(off[-1]), 2)) * 2 - 1)).astype(np.float32)
    x = (0.5 + 0.5 * rng.random((B, 3))).astype(np.float32)
    return pls, off, emb, x


def test_grid_offsets_and_resolutions(O):
    pls = O.per_level_scale_from_cfg()
    off = O.grid_offsets(16, pls, 16, 19, True)
    sizes = np.diff(off)
    assert list(sizes[:5]) == [4096, 13824, 39304, 117656, 357912] and np.all(sizes[5:] == 2 ** 19)
    assert int(off[-1]) == 6299960
    res = O.grid_resolutions(16, O.grid_S(pls), 16)
    assert res[0] == 16 and res[15] in (4095, 4096)
    assert np.all(np.diff(res.astype(np.int64)) > 0)


def test_grid_c_oracle_equals_torch_port(O):
    from oracle import torch_port as TP
    rng = np.random.default_rng(11)
    pls, off, emb, x = _grid_setup(O, rng, 300)
    a = O.grid_encode_forward(x, emb, off, pls, 16, 0, True, 0)
    b = TP.grid_encode(torch.tensor(x), torch.tensor(emb), off, pls, 16, True).numpy()
    assert np.abs(a - b).max() < 2e-6
    # align_corners=False path too
    off2 = O.grid_offsets(16, pls, 16, 19, False)
    emb2 = rng.random((int(off2[-1]), 2)).astype(np.float32)
    x2 = rng.random((200, 3)).astype(np.float32)
    a = O.grid_encode_forward(x2, emb2, off2, pls, 16, 0, False, 0)
    b = TP.grid_encode(torch.tensor(x2), torch.tensor(emb2), off2, pls, 16, False).numpy()
    assert np.abs(a - b).max() < 2e-6


def test_grid_level_locality_oob_and_transpose(O):
    rng = np.random.default_rng(13)
    pls, off, emb, x = _grid_setup(O, rng, 200)
    base = O.grid_encode_forward(x, emb, off, pls, 16, 0, True, 0)
    # level l output depends only on rows offsets[l]..offsets[l+1]
    emb2 = emb.copy()
    emb2[off[3]:off[4]] += 1.0
    out2 = O.grid_encode_forward(x, emb2, off, pls, 16, 0, True, 0)
    changed = np.abs(out2 - base).reshape(len(x), 16, 2).max(axis=(0, 2)) > 0
    assert list(np.nonzero(changed)[0]) == [3]
    # OOB -> zeros and zero gradient
    xo = x.copy(); xo[0, 1] = 1.5; xo[1, 2] = -0.1
    out = O.grid_encode_forward(xo, emb, off, pls, 16, 0, True, 0)
    assert np.all(out[:2] == 0) and np.any(out[2:] != 0)
    g = rng.standard_normal(base.shape).astype(np.float32)
    ge = O.grid_encode_backward(g, xo, off, len(emb), 2, pls, 16, 0, True, 0)
    ge_ref = O.grid_encode_backward(g[2:], xo[2:], off, len(emb), 2, pls, 16, 0, True, 0)
    assert np.array_equal(ge, ge_ref)
    # backward is the transpose of forward: <g, F(e)> == <B(g), e>
    lhs = float((g.astype(np.float64) * base).sum())
    ge = O.grid_encode_backward(g, x, off, len(emb), 2, pls, 16, 0, True, 0)
    rhs = float((ge.astype(np.float64) * emb).sum())
    assert abs(lhs - rhs) < 1e-3 * max(1.0, abs(lhs))
    # every level is hashed on this config (SURVEY section 7 quirk ii): rows beyond the dense range occur
    rows = O.grid_corner_rows(x, off, pls, 16, 0, True, 0)
    assert rows.max() < 2 ** 19 and rows[0].max() < 4096


def test_f16_rounding_helper(O):
    rng = np.random.default_rng(17)
    a = np.concatenate([rng.standard_normal(4000) * 10.0 ** rng.integers(-8, 5, 4000), [0, 65504, 65520, 1e-8, -1e-8, 6e-8]]).astype(np.float32)
    assert np.array_equal(O.round_f16(a), a.astype(np.float16).astype(np.float32))


def test_field_oracle_equals_torch_port(O):
    from oracle import torch_port as TP
    f = TP.Field(num_classes=5, table_scale=0.5)
    rng = np.random.default_rng(19)
    pts = (rng.random((300, 3)) * 4 - 2).astype(np.float32)
    with torch.no_grad():
        out_t, sig_t = f(torch.tensor(pts))
    fp = O.FieldParams(f.emb_density.detach().numpy(), f.emb_color.detach().numpy(), f.p_density.detach().numpy(),
                       f.p_color1.detach().numpy(), f.p_color2.detach().numpy(), f.p_class.detach().numpy(), f.offsets, f.pls)
    out_o, sig_o, _ = O.field_forward(fp, pts)
    assert rel_l2(out_o, out_t.numpy()) < 1e-5 and rel_l2(sig_o, sig_t.numpy()[:, 0]) < 1e-5
