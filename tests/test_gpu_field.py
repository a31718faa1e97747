"""GPU parity of the encode / MLP / fused-field kernels against the CPU oracle.

Tolerances (floating point, stated per test):
  * hash-grid encode, fp32 tables: fp32 interpolation with FMA vs the oracle's non-contracted
    fp32 -> max abs error <= 4e-6 * max|table|; corner ROWS are checked bit-exactly through a table
    whose entries encode their own row index.
  * fp16 tables: inputs identical halves; kernel accumulates in fp32 and rounds once, the reference
    (and the oracle in half_accum mode) rounds after every corner -> <= 2 half ulps of the result.
  * MLP / fused field (f16 or bf16 MFMA, fp32 accumulate): compared with the oracle emulating the
    same roundings (rel L2 <= 2e-3 f16 / 1e-2 bf16) and with the pure fp32 restatement
    (rel L2 <= 5e-3 f16 / 3e-2 bf16).  tinycudann itself is absent: "parity unpinned" there.
  * backward: fp32 autograd of the PyTorch restatement; rel L2 <= 2e-2 (f16) / 5e-2 (bf16) per
    parameter block, atomics make the summation order run-dependent.
"""
import numpy as np
import pytest
import torch

from helpers import rel_l2

pytestmark = pytest.mark.gpu


def T(a, dev):
    return torch.as_tensor(np.ascontiguousarray(a), device=dev)


# ---------------------------------------------------------------------------------------------
# stand-alone GridEncoder
# ---------------------------------------------------------------------------------------------
def _enc(O, align=True, gridtype='hash'):
    from nerfstyle_amd.gridencoder import GridEncoder
    pls = O.per_level_scale_from_cfg()
    return GridEncoder(3, 16, 2, pls, 16, 19, gridtype=gridtype, align_corners=align), pls


@pytest.mark.parametrize('align,gridtype', [(True, 'hash'), (False, 'hash'), (True, 'tiled')])
def test_grid_encode_forward_fp32_and_rows(O, dev, align, gridtype):
    """fp32 encode vs the C oracle, and corner-row bit-exactness at EVERY level (hashed ones and, with
    gridtype='tiled', the dense-index branch of gridencoder.cu:62-77): one-hot probes on the rows the oracle
    says a sample touches must reproduce weight sums of exactly 1 at that level and 0 at all others."""
    gt = 0 if gridtype == 'hash' else 1
    enc, pls = _enc(O, align, gridtype)
    rng = np.random.default_rng(5)
    R = enc.embeddings.shape[0]
    emb = (rng.random((R, 2)) * 2 - 1).astype(np.float32)
    B = 20011
    x = rng.random((B, 3)).astype(np.float32)
    x[:5] = [[1.2, 0.5, 0.5], [0.5, -0.1, 0.5], [0, 0, 0], [1, 1, 1], [0.5, 0.5, 1.0]]
    off = enc.offsets.numpy()
    enc = enc.to(dev)
    with torch.no_grad():
        enc.embeddings.copy_(T(emb, dev))
        out = enc(T(x, dev) * 2 - 1).cpu().numpy()              # GridEncoder.forward maps [-1,1] -> [0,1]
    xin = ((x * 2 - 1) + np.float32(1)) / np.float32(2)
    ref = O.grid_encode_forward(xin.astype(np.float32), emb, off, pls, 16, gt, align, 0)
    assert np.abs(out - ref).max() <= 4e-6
    assert np.all(out[:2] == 0)                                 # OOB rows
    rows = O.grid_corner_rows(xin.astype(np.float32), off, pls, 16, gt, align, 0)      # [L, B, 8]
    lo, hi = 5, 400
    for lvl in range(16):
        probe = np.zeros((R, 2), np.float32)
        sel = np.unique(rows[lvl, lo:hi].reshape(-1).astype(np.int64))
        probe[off[lvl] + sel, 0] = 1.0                         # all corners of those samples -> weights sum to 1
        with torch.no_grad():
            enc.embeddings.copy_(T(probe, dev))
            o2 = enc(T(x, dev) * 2 - 1).cpu().numpy().reshape(B, 16, 2)
        # every corner row the kernel touched at this level is in the oracle's set (else the weight sum drops below 1)
        assert np.abs(o2[lo:hi, lvl, 0] - 1.0).max() < 1e-6, lvl
        other = np.delete(np.arange(16), lvl)
        assert np.all(o2[lo:hi][:, other, 0] == 0), lvl


def test_grid_encode_half_tables_and_backward(O, dev):
    enc, pls = _enc(O, True)
    rng = np.random.default_rng(6)
    R = enc.embeddings.shape[0]
    emb = O.round_f16((rng.random((R, 2)) * 2 - 1).astype(np.float32))
    B = 8191
    x = (0.5 + 0.5 * rng.random((B, 3))).astype(np.float32)
    off = enc.offsets.numpy()
    enc = enc.to(dev)
    with torch.no_grad():
        enc.embeddings.copy_(T(emb, dev))
    xt = T(x * 2 - 1, dev)
    with torch.autocast('cuda', dtype=torch.float16):
        out = enc(xt)
    assert out.dtype == torch.float16
    ref = O.grid_encode_forward(x, emb, off, pls, 16, 0, True, 0, half_accum=True)
    ref32 = O.grid_encode_forward(x, emb, off, pls, 16, 0, True, 0, half_accum=False)
    o = out.detach().float().cpu().numpy()
    # one rounding of the fp32 sum: equal to round_f16(oracle fp32 sum) except where the two fp32 sums (different
    # summation order) straddle a rounding boundary -- then exactly one f16 ulp apart, and rare
    r16 = O.round_f16(ref32)
    d = np.abs(o - r16)
    ulp = np.spacing(np.maximum(np.abs(r16), 2.0 ** -14).astype(np.float16)).astype(np.float32)
    assert np.all(d <= ulp) and float((d > 0).mean()) < 2e-3
    assert np.abs(o - ref).max() <= 2 * 2.0 ** -10                                          # vs per-corner rounding
    # backward, fp32 path
    out = enc(xt)
    g = rng.standard_normal(out.shape).astype(np.float32)
    out.backward(T(g, dev))
    ge = enc.embeddings.grad.cpu().numpy()
    ge_ref = O.grid_encode_backward(g, x, off, R, 2, pls, 16, 0, True, 0)
    assert rel_l2(ge, ge_ref) < 1e-5
    assert np.array_equal(ge == 0, ge_ref == 0)


# ---------------------------------------------------------------------------------------------
# stand-alone Network (tcnn.Network replacement)
# ---------------------------------------------------------------------------------------------
@pytest.mark.parametrize('cfg', [(32, 1, 1, 'None'), (32, 16, 1, 'None'), (32, 5, 1, 'None'), (16, 3, 2, 'Sigmoid')])
@pytest.mark.parametrize('dt', ['f16', 'bf16'])
def test_network_forward_backward(O, dev, cfg, dt):
    from nerfstyle_amd.network import Network
    from oracle import torch_port as TP
    n_in, n_out, nh, act = cfg
    tdt = torch.float16 if dt == 'f16' else torch.bfloat16
    net = Network(n_in, n_out, {'otype': 'FullyFusedMLP', 'activation': 'ReLU', 'output_activation': act, 'n_neurons': 64,
                                'n_hidden_layers': nh}, seed=7, dtype=tdt).to(dev)
    rng = np.random.default_rng(8)
    M = 10007                                                   # ragged: not a multiple of 16
    x = rng.standard_normal((M, n_in)).astype(np.float32)
    p = net.params.detach().cpu().numpy()
    xt = T(x, dev).requires_grad_()
    y = net(xt)
    assert y.shape == (M, n_out)
    oa = 'sigmoid' if act == 'Sigmoid' else 'none'
    ref_emul = O.mlp_forward(x, p, n_in, n_out, 64, nh, oa, half=dt)
    ref_f32 = O.mlp_forward(x, p, n_in, n_out, 64, nh, oa, half=None)
    yn = y.detach().cpu().numpy()
    assert rel_l2(yn, ref_emul) < (2e-3 if dt == 'f16' else 1e-2)
    assert rel_l2(yn, ref_f32) < (5e-3 if dt == 'f16' else 3e-2)
    # backward: (a) autograd through a forward that emulates the operand rounding (same ReLU masks):
    # tight; (b) pure fp32 autograd: loose -- rounded pre-activations flip a few ReLU masks per
    # thousand samples, each flip moves that sample's gradient by O(1/sqrt(width))
    g = rng.standard_normal(yn.shape).astype(np.float32)
    y.backward(T(g, dev))
    gp = net.params.grad.cpu().numpy()
    for half, tol in ((dt, 4e-3 if dt == 'f16' else 2.5e-2), (None, 5e-2 if dt == 'f16' else 1.5e-1)):
        xc = torch.tensor(x, requires_grad=True)
        pc = torch.tensor(p, requires_grad=True)
        TP.mlp(xc, pc, n_in, n_out, nh, 64, oa, half=half).backward(torch.tensor(g))
        assert rel_l2(xt.grad.cpu().numpy(), xc.grad.numpy()) < tol, (half, 'dx')
        assert rel_l2(gp, pc.grad.numpy()) < tol, (half, 'dparams')
    # padded output rows receive no gradient
    assert np.all(gp.reshape(-1)[-(16 - n_out) * 64:] == 0) if n_out < 16 else True


# ---------------------------------------------------------------------------------------------
# fused field
# ---------------------------------------------------------------------------------------------
def _field_pair(dev, dt, table_dtype, nc=5, table_scale=0.5, min_res=16):
    from nerfstyle_amd.common import BBox
    from nerfstyle_amd.config import NetworkConfig, PosEncConfig
    from nerfstyle_amd.style_nerf import StyleTCNerf
    from oracle import torch_port as TP
    ref = TP.Field(num_classes=nc, table_scale=table_scale, min_res=min_res)
    tdt = torch.float16 if dt == 'f16' else torch.bfloat16
    m = StyleTCNerf(NetworkConfig(pos_enc=PosEncConfig(min_res=min_res)), BBox.from_radius(2.0), nc, enc_dtype=table_dtype,
                    use_dir=False, compute_dtype=tdt)
    sd = m.state_dict()
    sd['x_density_embedder.embeddings'] = ref.emb_density.detach()
    sd['x_color_embedder.embeddings'] = ref.emb_color.detach()
    sd['density_net.params'] = ref.p_density.detach()
    sd['color1_net.params'] = ref.p_color1.detach()
    sd['color2_net.params'] = ref.p_color2.detach()
    sd['class_net.params'] = ref.p_class.detach()
    m.load_state_dict(sd)
    return m.to(dev), ref


def _oracle_params(O, ref, table_half=False):
    return O.FieldParams(ref.emb_density.detach().numpy(), ref.emb_color.detach().numpy(), ref.p_density.detach().numpy(),
                         ref.p_color1.detach().numpy(), ref.p_color2.detach().numpy(), ref.p_class.detach().numpy(),
                         ref.offsets, ref.pls, num_classes=ref.nc)


@pytest.mark.parametrize('dt,table_dtype', [('f16', torch.float32), ('f16', None), ('bf16', torch.float32)])
def test_field_forward(O, dev, dt, table_dtype):
    m, ref = _field_pair(dev, dt, table_dtype)
    rng = np.random.default_rng(10)
    M = 12345
    pts = (rng.random((M, 3)) * 4 - 2).astype(np.float32)
    pts[:3] = [[2, 2, 2], [-2, -2, -2], [0, 0, 0]]
    rgbs, sig = m(T(pts, dev), dirs=T(pts, dev))
    assert rgbs.shape == (M, 8) and sig.shape == (M, 1)
    fp = _oracle_params(O, ref)
    half = table_dtype is None
    out_e, sig_e, logit_e = O.field_forward(fp, pts, half=dt, table_half=half)
    out_f, sig_f, _ = O.field_forward(fp, pts, half=None, table_half=half)
    rn, sn = rgbs.detach().cpu().numpy(), sig.detach().cpu().numpy()[:, 0]
    tol_e, tol_f = (2e-3, 5e-3) if dt == 'f16' else (1e-2, 3e-2)
    assert rel_l2(rn, out_e) < tol_e and rel_l2(rn, out_f) < tol_f
    # sigma = exp(logit): compare in log space (logit abs error), it spans orders of magnitude
    assert np.abs(np.log(sn) - logit_e).max() < (2e-2 if dt == 'f16' else 1e-1)
    # sigma-only branch (style_nerf.py:125-126) gives the same sigma
    s_only = m(T(pts, dev))
    assert torch.equal(s_only, sig)
    # device-side sample count: outputs past the count are untouched
    cnt = torch.tensor([1000, 0], dtype=torch.int32, device=dev)
    s2, r2 = m.field(T(pts, dev), False, cnt)
    assert torch.equal(s2[:1000], sig[:1000, 0]) and torch.equal(r2[:1000], rgbs[:1000])


def test_field_forward_psnr_vs_fp32_restatement(O, dev):
    """The metric's PSNR definition (utils/__init__.py:323-325) applied to per-sample colours:
    f16 MFMA vs the fp32 restatement must clear the 40 dB bar with margin."""
    m, ref = _field_pair(dev, 'f16', torch.float32)
    rng = np.random.default_rng(12)
    pts = (rng.random((50000, 3)) * 4 - 2).astype(np.float32)
    rgbs, _ = m(T(pts, dev), dirs=T(pts, dev))
    out_f, _, _ = O.field_forward(_oracle_params(O, ref), pts)
    mse = float(np.mean((rgbs.detach().cpu().numpy()[:, :3] - out_f[:, :3]) ** 2))
    assert O.compute_psnr(mse) > 55.0


@pytest.mark.parametrize('dt,table_dtype', [('f16', torch.float32), ('bf16', torch.float32), ('f16', None)])
def test_field_backward(O, dev, dt, table_dtype):
    m, ref = _field_pair(dev, dt, table_dtype)
    rng = np.random.default_rng(14)
    M = 4099
    pts = (rng.random((M, 3)) * 4 - 2).astype(np.float32)
    gs = (rng.standard_normal(M) * 1e-2).astype(np.float32)
    gr = rng.standard_normal((M, 8)).astype(np.float32)
    sig, rgbs = m.field(T(pts, dev), False)
    ((sig * T(gs, dev)).sum() + (rgbs * T(gr, dev)).sum()).backward()
    # reference: autograd through the PyTorch restatement with the same operand roundings emulated
    # (straight-through), so the ReLU masks agree; see test_network_forward_backward for why
    out_r, sig_r = ref(torch.tensor(pts), half=dt, table_half=table_dtype is None)
    ((sig_r[:, 0] * torch.tensor(gs)).sum() + (out_r * torch.tensor(gr)).sum()).backward()
    ga = m.arena.grad.cpu().numpy()
    gt = ga[:m.table_elems].reshape(m.rows, 2, 2)
    tol = 5e-3 if dt == 'f16' else 3e-2
    assert rel_l2(gt[:, 0, :], ref.emb_density.grad.numpy()) < tol
    assert rel_l2(gt[:, 1, :], ref.emb_color.grad.numpy()) < tol
    gm = ga[m.table_elems:]
    for name, off, n, p in (('density', 0, 3072, ref.p_density), ('color1', 3072, 3072, ref.p_color1),
                            ('color2', 6144, 6144, ref.p_color2), ('class', 12288, 3072, ref.p_class)):
        assert rel_l2(gm[off:off + n], p.grad.numpy()) < tol, name
    # rows nobody touched stay exactly zero (scatter goes only where the gather went)
    assert not np.any((gt[:, 0, :] != 0) & (ref.emb_density.grad.numpy() == 0))
    assert not np.any((gt[:, 1, :] != 0) & (ref.emb_color.grad.numpy() == 0))
    # accumulation semantics: a second backward doubles the gradient
    sig, rgbs = m.field(T(pts, dev), False)
    ((sig * T(gs, dev)).sum() + (rgbs * T(gr, dev)).sum()).backward()
    assert rel_l2(m.arena.grad.cpu().numpy(), 2 * ga) < 1e-3
    # the backward with saved encoder features (default) == the backward that re-gathers
    m.arena.grad.zero_()
    m.save_features = False
    sig, rgbs = m.field(T(pts, dev), False)
    ((sig * T(gs, dev)).sum() + (rgbs * T(gr, dev)).sum()).backward()
    assert rel_l2(m.arena.grad.cpu().numpy(), ga) < 1e-3
    m.save_features = True
    # stylisation mode: only the colour table is trained (trainers/style.py:25)
    m.arena.grad.zero_()
    m.train_density_table = False
    sig, rgbs = m.field(T(pts, dev), False)
    ((sig * T(gs, dev)).sum() + (rgbs * T(gr, dev)).sum()).backward()
    gt2 = m.arena.grad.cpu().numpy()[:m.table_elems].reshape(m.rows, 2, 2)
    assert np.all(gt2[:, 0, :] == 0) and rel_l2(gt2[:, 1, :], gt[:, 1, :]) < 1e-3


@pytest.mark.parametrize('min_res,sorted_walk', [(16, False), (2, False), (16, True), (2, True)])
def test_field_backward_run_tracker_on_ray_structured_samples(O, dev, min_res, sorted_walk):
    """The backward's scatter keeps the open run of every (level, corner) in registers along CONSECUTIVE
    samples and hands runs over when a sample moves one cell along one axis (field_bwd.hip,
    field_scatter_seq).  Random points never exercise that, so: samples marching along rays with steps from
    far below a fine cell to several coarse cells, axis-aligned rays in both directions (pure x / y / z
    hand-overs), samples that stand still, samples outside the box in between (dead lanes), ray ends in
    the middle of 16-sample tiles.  min_res = 2 makes the coarsest levels dense (use_hash = 0) and not a
    power of two.  sorted_walk: the same samples walked in the spatial order of nsr_sample_order, scattered by the
    LDS lattice accumulator instead (field_scatter_lattice): tile changes on every level, dead samples in between."""
    m, ref = _field_pair(dev, 'f16', torch.float32, min_res=min_res)
    assert int(m.x_density_embedder.offsets[-1]) == int(ref.offsets[-1])
    rng = np.random.default_rng(7 + min_res)
    chunks = []
    for k in range(220):
        n = int(rng.integers(1, 90))
        o = (rng.random(3) * 3.6 - 1.8).astype(np.float32)
        kind = k % 5
        if kind == 0:                                     # generic direction, march-like step
            d = rng.standard_normal(3).astype(np.float32)
            d /= np.linalg.norm(d)
            step = 2 * np.sqrt(3) / 1024
        elif kind in (1, 2, 3):                           # axis-aligned, both signs
            d = np.zeros(3, np.float32)
            d[kind - 1] = 1.0 if (k // 5) % 2 == 0 else -1.0
            step = float(rng.choice([1e-4, 2e-3, 3e-2]))
        else:                                             # coarse jumps
            d = rng.standard_normal(3).astype(np.float32)
            d /= np.linalg.norm(d)
            step = float(rng.choice([0.05, 0.3]))
        t = np.arange(n, dtype=np.float32)[:, None] * np.float32(step)
        p = o[None, :] + t * d[None, :]
        if k % 7 == 0:
            p[n // 2:] = p[n // 2]                        # the ray stops: repeated identical samples
        if k % 11 == 0:
            p[::3] += 10.0                                # every third sample far outside the box (dead)
        chunks.append(p.astype(np.float32))
    pts = np.concatenate(chunks)
    M = pts.shape[0]
    assert M % 16 != 0
    gs = (rng.standard_normal(M) * 1e-2).astype(np.float32)
    gr = rng.standard_normal((M, 8)).astype(np.float32)
    gr[rng.random(M) < 0.2] = 0.0                         # zero upstream gradients on a fifth of the samples
    gs[rng.random(M) < 0.2] = 0.0
    perm = m.sample_order(T(pts, dev)) if sorted_walk else None
    sig, rgbs = m.field(T(pts, dev), False, perm=perm)
    ((sig * T(gs, dev)).sum() + (rgbs * T(gr, dev)).sum()).backward()
    # the restatement applies the reference's own out-of-range rule (zero features, gridencoder.cu:118-131): the
    # encoder input is (x_hat + 1) / 2, so points down to -6 on an axis are still encoded, points above +2 are not
    out_r, sig_r = ref(torch.tensor(pts), half='f16')
    ((sig_r[:, 0] * torch.tensor(gs)).sum() + (out_r * torch.tensor(gr)).sum()).backward()
    gt = m.arena.grad.cpu().numpy()[:m.table_elems].reshape(m.rows, 2, 2)
    assert rel_l2(gt[:, 0, :], ref.emb_density.grad.numpy()) < 5e-3
    assert rel_l2(gt[:, 1, :], ref.emb_color.grad.numpy()) < 5e-3
    assert not np.any((gt[:, 0, :] != 0) & (ref.emb_density.grad.numpy() == 0))
    # per level, so that a hand-over bug on a small (coarse) level cannot hide behind the fine levels' mass
    off = np.asarray(ref.offsets)
    for l in range(16):
        a, b = gt[off[l]:off[l + 1], 1, :], ref.emb_color.grad.numpy()[off[l]:off[l + 1]]
        assert rel_l2(a, b) < 1e-2, l


def _march_samples(O, dev, n_rays, cap):
    """ray-marched samples of the small scene in a capacity-sized buffer + device-side count"""
    from nerfstyle_amd import raymarching as R
    from helpers import room_rays, small_scene
    grid, bits = small_scene()
    ro, rd = room_rays(O, n_rays, seed=3)
    aabb = T(np.array([-2, -2, -2, 2, 2, 2], np.float32), dev)
    near, far = R.near_far_from_aabb(T(ro, dev), T(rd, dev), aabb, 0.2)
    counter = torch.zeros(2, dtype=torch.int32, device=dev)
    xyzs, _, deltas, rays = R.march_rays_train_nosync(T(ro, dev), T(rd, dev), 2.0, T(bits, dev), 2, 128, near, far, n_rays * cap,
                                                      counter, 0., 1024)
    return xyzs, counter


@pytest.mark.parametrize('dt,table_dtype', [('f16', None), ('bf16', torch.float32)])
def test_sample_order_and_sorted_walk_equal_buffer_order(O, dev, dt, table_dtype):
    """nsr_sample_order + the `perm` argument of the fused field: the permutation is a bijection of the valid slots in
    non-decreasing Morton key order (stable), whatever sort_prefix is; the forward through it is BIT-identical per
    sample; the backward (lattice accumulator) equals the buffer-order backward (run tracker) up to fp32 summation
    order, with saved features and with the re-gather path, and with only the colour table trained."""
    m, ref = _field_pair(dev, dt, table_dtype)
    xyzs, counter = _march_samples(O, dev, 3000, 160)
    M, cnt = xyzs.shape[0], int(counter[0])
    assert 16 * 3000 < cnt < M
    xyzs[cnt:] = float('nan')                                   # capacity tail: never read
    x = xyzs[:cnt].cpu().numpy()
    u = ((x + 2.0) / 4.0 + 1.0) * 0.5
    q = np.clip(u * 1024.0, 0, 1023).astype(np.uint32)

    def spread(v):
        v = v.astype(np.uint64) & 0x3FF
        v = (v | (v << 16)) & 0x030000FF
        v = (v | (v << 8)) & 0x0300F00F
        v = (v | (v << 4)) & 0x030C30C3
        v = (v | (v << 2)) & 0x09249249
        return v
    key = spread(q[:, 0]) | (spread(q[:, 1]) << 1) | (spread(q[:, 2]) << 2)
    want = np.argsort(key, kind='stable')
    for prefix in (None, cnt + 1000, cnt // 2, 0):
        perm = m.sample_order(xyzs, m_dev=counter, sort_prefix=prefix)
        ph = perm.cpu().numpy().astype(np.int64)
        assert np.array_equal(np.sort(ph), np.arange(M))                         # a permutation of every slot
        assert np.array_equal(np.sort(ph[:cnt]), np.arange(cnt))                # valid samples first
        if prefix is None or prefix >= cnt:
            assert np.array_equal(ph[:cnt], want)                                # stable Morton order
        elif prefix > 0:
            assert np.all(np.diff(key[ph[:prefix]].astype(np.int64)) >= 0) and np.array_equal(ph[prefix:], np.arange(prefix, M))
        else:
            assert np.array_equal(ph, np.arange(M))
    perm = m.sample_order(xyzs, m_dev=counter, sort_prefix=cnt + 4096)
    g = torch.Generator(device=dev)
    g.manual_seed(1)
    gs = torch.randn(M, device=dev, generator=g) * 1e-2
    gr = torch.randn(M, 8, device=dev, generator=g)

    def run(p, feats=True, density=True):
        m.arena.grad = None
        m.grad_arena = None
        m.save_features, m.train_density_table = feats, density
        sig, rgb = m.field(xyzs, False, counter, perm=p)
        torch.autograd.backward([sig, rgb], [gs, gr])
        m.save_features, m.train_density_table = True, True
        return sig.detach()[:cnt].clone(), rgb.detach()[:cnt].clone(), m.arena.grad.detach().clone()
    s0, r0, g0 = run(None)
    s1, r1, g1 = run(perm)
    assert torch.equal(s0, s1) and torch.equal(r0, r1)                           # per-sample results do not depend on the walk
    assert float(g0.abs().sum()) > 0
    assert rel_l2(g1.cpu().numpy(), g0.cpu().numpy()) < 2e-5
    assert int(((g1 == 0) != (g0 == 0)).sum()) < 10                              # the scatter goes exactly where the gather went
    _, _, g2 = run(perm, feats=False)
    assert rel_l2(g2.cpu().numpy(), g0.cpu().numpy()) < 2e-5
    _, _, g3 = run(perm, density=False)
    t3, t0 = g3[:m.table_elems].view(m.rows, 2, 2), g0[:m.table_elems].view(m.rows, 2, 2)
    assert float(t3[:, 0].abs().max()) == 0.0 and rel_l2(t3[:, 1].cpu().numpy(), t0[:, 1].cpu().numpy()) < 2e-5
    # ... and no net trained either (the stylisation stage, trainers/style.py:25): the colour-only gradients-out kernel
    # (k_field_bwd_color: class + colour nets' input gradients, no weight gradients, nothing behind sigma)
    m.train_mlps = False
    try:
        _, _, g4 = run(perm, density=False)
    finally:
        m.train_mlps = True
    t4 = g4[:m.table_elems].view(m.rows, 2, 2)
    assert float(t4[:, 0].abs().max()) == 0.0 and float(g4[m.table_elems:].abs().max()) == 0.0
    assert rel_l2(t4[:, 1].cpu().numpy(), t3[:, 1].cpu().numpy()) < 2e-6         # the same arithmetic; atomics order only
    assert rel_l2(t4[:, 1].cpu().numpy(), t0[:, 1].cpu().numpy()) < 2e-5
    # sigma-only forward through the permutation
    with torch.no_grad():
        s_only = m.field(xyzs, True, counter, perm=perm)
    assert torch.equal(s_only[:cnt], s0)


def test_sample_order_captured_alone_replays_equal_eager(O, dev):
    """nsr_sample_order is capture-safe (round 3: its own LSD radix sort; round 2's rocPRIM call reset a tile counter with a
    stream-less hipMemset per pass and faulted on the SECOND replay): captured alone, four replays -- with a different
    device-side count before the last one -- equal the eager permutation."""
    m, _ = _field_pair(dev, 'f16', None)
    M = 40000 * 64
    g = torch.Generator(device=dev)
    g.manual_seed(11)
    xyzs = (torch.rand(M, 3, device=dev, generator=g) * 3.6 - 1.8).contiguous()
    cnt = torch.tensor([M - 1000, 0], dtype=torch.int32, device=dev)
    p_eager = m.sample_order(xyzs, cnt).clone()
    side = torch.cuda.Stream(device=dev)
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        m.sample_order(xyzs, cnt)
    torch.cuda.current_stream().wait_stream(side)
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        p_graph = m.sample_order(xyzs, cnt)
    for _ in range(3):
        p_graph.zero_()
        graph.replay()
        assert torch.equal(p_graph, p_eager)
    cnt.copy_(torch.tensor([M // 3, 0], dtype=torch.int32))
    p_small = m.sample_order(xyzs, cnt).clone()
    graph.replay()
    assert torch.equal(p_graph, p_small) and not torch.equal(p_small, p_eager)
    assert torch.equal(p_small[M // 3:], torch.arange(M // 3, M, dtype=torch.int32, device=dev))


@pytest.mark.parametrize('nc', [1, 13])
def test_field_other_class_counts(O, dev, nc):
    """C_ch = 3 + nc other than the vectorised 8-channel case: 4 channels (nc = 1) and the largest the
    16-wide class layer holds (nc = 13); forward and backward against the rounding-emulating restatement."""
    m, ref = _field_pair(dev, 'f16', torch.float32, nc=nc)
    rng = np.random.default_rng(nc)
    M = 3001
    pts = (rng.random((M, 3)) * 4 - 2).astype(np.float32)
    gs = (rng.standard_normal(M) * 1e-2).astype(np.float32)
    gr = rng.standard_normal((M, 3 + nc)).astype(np.float32)
    sig, rgbs = m.field(T(pts, dev), False)
    ((sig * T(gs, dev)).sum() + (rgbs * T(gr, dev)).sum()).backward()
    out_r, sig_r = ref(torch.tensor(pts), half='f16')
    ((sig_r[:, 0] * torch.tensor(gs)).sum() + (out_r * torch.tensor(gr)).sum()).backward()
    assert rgbs.shape == (M, 3 + nc)
    assert rel_l2(rgbs.detach().cpu().numpy(), out_r.detach().numpy()) < 2e-3
    ga = m.arena.grad.cpu().numpy()
    gt = ga[:m.table_elems].reshape(m.rows, 2, 2)
    assert rel_l2(gt[:, 0, :], ref.emb_density.grad.numpy()) < 5e-3
    assert rel_l2(gt[:, 1, :], ref.emb_color.grad.numpy()) < 5e-3
    assert rel_l2(ga[m.table_elems + 12288:m.table_elems + 15360], ref.p_class.grad.numpy()) < 5e-3


def test_spatial_scatter_falls_back_on_unsupported_grid(O, dev):
    """A grid whose finest level has more than ~5 cells per 1/1024 sample-order block does not fit the scatter kernel's LDS
    lattices: nsr_field_backward reports NSR_ERR_UNSUPPORTED for a `perm` call without launching anything, and the host layer
    re-issues the fused (run tracker) backward -- same gradient as a call without perm."""
    import ctypes
    from nerfstyle_amd import _lib as L
    from nerfstyle_amd.common import BBox
    from nerfstyle_amd.config import NetworkConfig, PosEncConfig
    from nerfstyle_amd.style_nerf import StyleTCNerf
    # max_res_coeff 4096: finest resolution 16384 -> 16 cells per block
    m = StyleTCNerf(NetworkConfig(pos_enc=PosEncConfig(max_res_coeff=4096)), BBox.from_radius(2.0), 5, enc_dtype=torch.float32,
                    use_dir=False).to(dev)
    with torch.no_grad():
        m.arena[:m.table_elems].uniform_(-0.5, 0.5)
        m.arena.add_(0)
    rng = np.random.default_rng(3)
    pts = T((rng.random((5000, 3)) * 4 - 2).astype(np.float32), dev)
    gs = T((rng.standard_normal(5000) * 1e-2).astype(np.float32), dev)
    gr = T(rng.standard_normal((5000, 8)).astype(np.float32), dev)

    def run(p):
        m.arena.grad = None
        m.grad_arena = None
        sig, rgb = m.field(pts, False, perm=p)
        torch.autograd.backward([sig, rgb], [gs, gr])
        return m.arena.grad.detach().clone()
    g0 = run(None)
    assert not getattr(m, '_spatial_scatter_unsupported', False)
    g1 = run(m.sample_order(pts))
    assert m._spatial_scatter_unsupported
    assert float(g0.abs().sum()) > 0 and rel_l2(g1.cpu().numpy(), g0.cpu().numpy()) < 2e-5


@pytest.mark.parametrize('min_res,coeff', [(24, 1280), (48, 1280)])
def test_spatial_scatter_near_lds_limit_counts_mlp_gradient_once(O, dev, min_res, coeff):
    """Grids whose lattices need 961..1024 float4 slots per wave: round 2 accepted them in nsr_table_scatter_supported and
    rejected them (64 KB of LDS) AFTER the MLP backward kernel had accumulated its weight gradient, and the host re-issued the
    whole backward -- MLP gradients counted twice.  The LDS bound is now part of the support test: whichever way such a grid
    goes (supported, or rejected before any launch), the gradient equals the call without perm."""
    from nerfstyle_amd.common import BBox
    from nerfstyle_amd.config import NetworkConfig, PosEncConfig
    from nerfstyle_amd.style_nerf import StyleTCNerf
    m = StyleTCNerf(NetworkConfig(pos_enc=PosEncConfig(min_res=min_res, max_res_coeff=coeff)), BBox.from_radius(2.0), 5,
                    enc_dtype=torch.float32, use_dir=False).to(dev)
    with torch.no_grad():
        m.arena[:m.table_elems].uniform_(-0.5, 0.5)
        m.arena.add_(0)
    rng = np.random.default_rng(5)
    pts = T((rng.random((6000, 3)) * 4 - 2).astype(np.float32), dev)
    gs = T((rng.standard_normal(6000) * 1e-2).astype(np.float32), dev)
    gr = T(rng.standard_normal((6000, 8)).astype(np.float32), dev)

    def run(p):
        m.arena.grad = None
        m.grad_arena = None
        sig, rgb = m.field(pts, False, perm=p)
        torch.autograd.backward([sig, rgb], [gs, gr])
        return m.arena.grad.detach().clone()
    g0 = run(None)
    g1 = run(m.sample_order(pts))
    mlp0, mlp1 = g0[m.table_elems:].cpu().numpy(), g1[m.table_elems:].cpu().numpy()
    assert float(np.abs(mlp0).sum()) > 0 and rel_l2(mlp1, mlp0) < 2e-5
    assert rel_l2(g1.cpu().numpy(), g0.cpu().numpy()) < 2e-5

