"""Host logic that needs no GPU: sizing, parameter layout, checkpoints, the synthetic scene."""
import numpy as np
import pytest
import torch

from helpers import small_scene


def test_gridencoder_sizing_matches_oracle(O):
    from nerfstyle_amd.gridencoder import GridEncoder
    pls = O.per_level_scale_from_cfg()
    enc = GridEncoder(3, 16, 2, pls, 16, 19, gridtype='hash', align_corners=True)
    assert np.array_equal(enc.offsets.numpy(), O.grid_offsets(16, pls, 16, 19, True))
    assert enc.embeddings.shape == (6299960, 2) and enc.n_output_dims == 32
    assert float(enc.embeddings.abs().max()) <= 1e-4
    enc2 = GridEncoder(3, 16, 2, pls, 16, 19, align_corners=False)
    assert np.array_equal(enc2.offsets.numpy(), O.grid_offsets(16, pls, 16, 19, False))


def test_network_param_layout(O):
    from nerfstyle_amd.network import Network, mlp_layer_shapes
    assert mlp_layer_shapes(32, 1) == O.mlp_layer_shapes(32, 1)
    assert mlp_layer_shapes(16, 3, 64, 2) == [(64, 16), (64, 64), (16, 64)]
    n = Network(32, 5, {'otype': 'FullyFusedMLP', 'activation': 'ReLU', 'output_activation': 'None', 'n_neurons': 64,
                        'n_hidden_layers': 1}, seed=80000)
    assert n.params.shape == (3072,) and n.n_output_dims == 5
    n2 = Network(32, 5, {'n_neurons': 64, 'n_hidden_layers': 1}, seed=80000)
    assert torch.equal(n.params, n2.params)      # seeded


def test_style_nerf_arena_and_checkpoint_roundtrip():
    from nerfstyle_amd.common import BBox
    from nerfstyle_amd.config import NetworkConfig
    from nerfstyle_amd.style_nerf import MLP_PARAMS, StyleTCNerf
    m = StyleTCNerf(NetworkConfig(), BBox.from_radius(2.0), 5, None, use_dir=False)
    assert m.rows == 6299960 and m.arena.numel() == m.rows * 4 + MLP_PARAMS == 25215200
    sd = m.state_dict()
    assert sorted(sd.keys()) == sorted([
        'x_density_embedder.embeddings', 'x_density_embedder.offsets', 'x_color_embedder.embeddings',
        'x_color_embedder.offsets', 'density_net.params', 'color1_net.params', 'color2_net.params', 'class_net.params'])
    assert sd['x_density_embedder.embeddings'].shape == (m.rows, 2) and sd['color2_net.params'].shape == (6144,)
    # interleaving: tables[row][enc][feat]
    t = m.tables_view()
    assert torch.equal(sd['x_color_embedder.embeddings'], t[:, 1, :])
    assert m.x_color_embedder.embeddings.data_ptr() == m.arena.data_ptr() + 8
    m2 = StyleTCNerf(NetworkConfig(network_seed=1), BBox.from_radius(2.0), 5)
    assert not torch.equal(m2.arena, m.arena)
    m2.load_state_dict(sd)
    assert torch.equal(m2.arena.detach(), m.arena.detach())
    try:
        StyleTCNerf(NetworkConfig(), BBox.from_radius(2.0), 5, use_dir=True)
        assert False
    except NotImplementedError:
        pass


def test_renderer_state_keys():
    from nerfstyle_amd.common import BBox, Intrinsics
    from nerfstyle_amd.config import NetworkConfig, RendererConfig
    from nerfstyle_amd.renderer import Renderer
    from nerfstyle_amd.style_nerf import StyleTCNerf
    m = StyleTCNerf(NetworkConfig(), BBox.from_radius(2.0), 5)
    r = Renderer(m, RendererConfig.llff(), Intrinsics(378, 504, 383.8, 383.8, 252., 189.), 2.0, raymarch_channels=8)
    assert r.cascade == 2 and r.density_bitfield.numel() == 524288 and r.density_grid.shape == (2, 128 ** 3)
    sd = r.state_dict()
    assert sorted(sd.keys()) == sorted(['model', 'intr', 'precrop_frac', 'raymarch_channels', 'bound', 'density_grid',
                                        'density_bitfield', 'step_counter', 'local_step', 'mean_count', 'mean_density'])
    r.load_state_dict(sd)
    assert r.sample_capacity(4096) == 4096 * 1024


def test_synthetic_scene_and_cameras(O):
    from nerfstyle_amd.scene import load_room_cameras, morton3d_np
    grid, bits = small_scene()
    assert grid.shape == (2, 128 ** 3) and bits.shape == (524288,)
    assert np.array_equal(bits, O.packbits(grid, 0.5))
    frac = grid.mean(axis=1)
    assert 0.005 < frac[0] < 0.6
    c = np.random.default_rng(0).integers(0, 128, (1000, 3))
    assert np.array_equal(morton3d_np(c[:, 0], c[:, 1], c[:, 2]).astype(np.int32), O.morton3D(c))
    poses, intr, meta = load_room_cameras()
    assert poses.shape == (35, 4, 4) and intr.w == 504 and intr.h == 378
    _, intr2, _ = load_room_cameras(2)
    assert intr2.w == 1008 and abs(intr2.fx - 2 * intr.fx) < 1e-9


def test_checkpoint_layout_roundtrip_and_refusal(tmp_path):
    """nerfstyle_amd/checkpoint.py: the reference's top-level keys (trainers/base.py:26-28), loadable with
    weights_only=True; a file holding pickled objects (what the reference writes) is refused with a message,
    never unpickled."""
    import dataclasses
    import pytest
    from nerfstyle_amd import checkpoint as C
    from nerfstyle_amd.common import BBox, Intrinsics
    from nerfstyle_amd.config import NetworkConfig, RendererConfig
    from nerfstyle_amd.optim import LossScaler
    from nerfstyle_amd.renderer import Renderer
    from nerfstyle_amd.style_nerf import StyleTCNerf
    intr = Intrinsics(378, 504, 383.8, 383.8, 252., 189.)
    m = StyleTCNerf(NetworkConfig(), BBox.from_radius(2.0), 5)
    r = Renderer(m, RendererConfig.llff(), intr, 2.0, raymarch_channels=8)
    with torch.no_grad():
        m.arena.uniform_(-1, 1)
    r.density_grid[0, :100] = 3.0
    r.local_step, r.mean_count = 17, 123
    sc = LossScaler()
    sc.load_state_dict(dict(sc.state_dict(), scale=1024.0, _growth_tracker=5, steps=17))
    p = tmp_path / 'iter_0017.pth'
    C.save_checkpoint(p, r, scaler=sc, iter_ctr=17, log_dir=tmp_path, train_cfg={'num_rays_per_batch': 4096})
    sd = C.load_checkpoint(p)
    assert sorted(sd.keys()) == sorted(C.SAVE_KEYS + C.SD_SAVE_KEYS)
    assert sd['iter_ctr'] == 17 and sd['render_cfg']['__dataclass__'] == 'RendererConfig'
    assert sd['render_cfg']['max_steps'] == dataclasses.asdict(RendererConfig.llff())['max_steps']
    assert sd['renderer']['intr'] == {'__dataclass__': 'Intrinsics', 'h': 378, 'w': 504, 'fx': 383.8, 'fy': 383.8,
                                       'cx': 252., 'cy': 189.}
    m2 = StyleTCNerf(NetworkConfig(), BBox.from_radius(2.0), 5)
    r2 = Renderer(m2, RendererConfig.llff(), intr, 2.0, raymarch_channels=8)
    sc2 = LossScaler()
    assert C.restore(sd, r2, scaler=sc2) == 17
    assert torch.equal(m2.arena.detach(), m.arena.detach())
    assert torch.equal(r2.density_grid, r.density_grid) and r2.local_step == 17 and r2.mean_count == 123
    assert sc2.get_scale() == 1024.0 and sc2.state_dict()['_growth_tracker'] == 5 and sc2.state_dict()['steps'] == 17
    # a checkpoint with pickled class instances, as the reference writes them: refused, not executed
    bad = tmp_path / 'reference_style.pth'
    torch.save({'renderer': {'intr': intr}, 'render_cfg': RendererConfig.llff()}, bad)
    with pytest.raises(RuntimeError, match='nothing from a checkpoint file is executed'):
        C.load_checkpoint(bad)


def test_reference_written_checkpoint_is_ingested_without_executing_it(tmp_path):
    """A file in the REFERENCE's own layout (trainers/base.py:231-249: pickled config.* dataclasses incl. the nested
    HashGridConfig / TrainIntervalConfig, common.Intrinsics inside `renderer`, pathlib.PosixPath, torch.optim.Adam /
    GradScaler / torch_ema state dicts) loads through nerfstyle_amd.reference_schema's inert stand-ins on the weights-only
    allow-list: the file below is pickled from classes registered as `config.*` / `common.Intrinsics` exactly as the reference's
    are, then read back with those modules GONE from sys.modules (nothing is imported, nothing from the file runs).
    A file that names any other global is still refused."""
    import pathlib
    import sys
    import types
    import pytest
    from nerfstyle_amd import checkpoint as C
    from nerfstyle_amd import reference_schema as RS
    from nerfstyle_amd.common import BBox, Intrinsics
    from nerfstyle_amd.config import NetworkConfig, PosEncConfig, RendererConfig
    from nerfstyle_amd.optim import FusedAdam, LossScaler
    from nerfstyle_amd.renderer import Renderer
    from nerfstyle_amd.style_nerf import MLP_LAYOUT, StyleTCNerf
    intr = Intrinsics(378, 504, 383.8, 383.8, 252., 189.)
    ncfg = NetworkConfig(pos_enc=PosEncConfig(hashmap_size=14))          # small tables: the layout is what is tested
    m = StyleTCNerf(ncfg, BBox.from_radius(2.0), 5)
    r = Renderer(m, RendererConfig.llff(), intr, 2.0, raymarch_channels=8)
    with torch.no_grad():
        m.arena.uniform_(-1, 1)
    r.density_grid[1, 5:50] = 2.0
    r.local_step, r.mean_count = 33, 77

    def mk(path, **kw):
        o = RS.STANDINS[path].__new__(RS.STANDINS[path])
        o.__dict__.update(kw)
        return o
    names = [n for n, _ in m.named_views()]
    g = torch.Generator().manual_seed(3)
    adam_state = {i: {'step': torch.tensor(41.), 'exp_avg': torch.randn(v.shape, generator=g), 'exp_avg_sq': torch.rand(v.shape, generator=g)}
                  for i, (n, v) in enumerate(m.named_views())}
    shadow = [torch.randn(v.shape, generator=g) for _, v in m.named_views()]
    rs = r.state_dict()
    rs['intr'] = mk('common.Intrinsics', h=378, w=504, fx=383.8, fy=383.8, cx=252., cy=189.)
    rs['model'] = dict(rs['model'], **{'x_density_embedder.offsets': torch.zeros(17, dtype=torch.int32)})   # a buffer the reference saves
    ref_sd = {
        'version': 'deadbeef', 'log_dir': pathlib.Path('/runs/room'), 'iter_ctr': 41,
        'cfg': mk('config.BaseConfig', log_dir=pathlib.Path('runs/room'), data_cfg=pathlib.Path('cfgs/dataset/llff_room.yaml'), ckpt=None),
        'dataset_cfg': mk('config.DatasetConfig', root_path=pathlib.Path('datasets/nerf_llff_data/room'), type='llff', bound=2.0, scale=0.33,
                          replica_cfg=mk('config.DatasetConfig.ReplicaConfig', name='room_0', focal_ratio=0.5, traj_ids=[1, 2], black2white=True)),
        'train_cfg': mk('config.TrainConfig', num_rays_per_batch=4096, enable_amp=True,
                        intervals=mk('config.TrainConfig.TrainIntervalConfig', print=100, log=100, ckpt=5000, test=5000)),
        'net_cfg': mk('config.NetworkConfig', network_seed=80000, density_out_dims=16,
                      pos_enc=mk('config.NetworkConfig.HashGridConfig', n_lvls=16, n_feats_per_lvl=2, hashmap_size=19, min_res=16, max_res_coeff=1024.0)),
        'render_cfg': mk('config.RendererConfig', grid_size=128, max_steps=1024, t_thresh=1e-4),
        'renderer': rs,
        'optim': {'state': adam_state, 'param_groups': [{'lr': 0.0097, 'initial_lr': 0.01, 'betas': (0.9, 0.999), 'eps': 1e-15,
                                                        'params': list(range(len(names)))}]},
        'scheduler': {'base_lrs': [0.01], 'last_epoch': 41, 'lr_lambdas': [None]},
        'scaler': {'scale': 32768.0, 'growth_factor': 2.0, 'backoff_factor': 0.5, 'growth_interval': 2000, '_growth_tracker': 12},
        'ema': {'decay': 0.95, 'num_updates': 41, 'shadow_params': shadow, 'collected_params': None},
    }
    cfgm, comm = types.ModuleType('config'), types.ModuleType('common')
    for path, cls in RS.STANDINS.items():
        mod, _, q = path.partition('.')
        if '.' not in q:
            setattr(cfgm if mod == 'config' else comm, q, cls)
    keep = {k: sys.modules.get(k) for k in ('config', 'common')}
    sys.modules['config'], sys.modules['common'] = cfgm, comm
    f = tmp_path / 'iter_00041.pth'
    try:
        torch.save(ref_sd, f)                         # pickles `config.NetworkConfig`, `common.Intrinsics`, `pathlib.PosixPath`, getattr(...)
    finally:
        for k, v in keep.items():
            if v is None:
                sys.modules.pop(k, None)
            else:
                sys.modules[k] = v
    raw = open(f, 'rb').read()
    assert b'config' in raw and b'HashGridConfig' in raw and b'PosixPath' in raw and b'Intrinsics' in raw
    with pytest.raises(Exception):
        torch.load(f, weights_only=True)               # refused without the allow-list
    sd = C.load_checkpoint(f)
    assert 'config' not in sys.modules or sys.modules['config'] is keep['config']
    assert sd['log_dir'] == '/runs/room' and sd['cfg']['data_cfg'] == 'cfgs/dataset/llff_room.yaml'
    assert sd['net_cfg']['pos_enc'] == {'__dataclass__': 'HashGridConfig', 'n_lvls': 16, 'n_feats_per_lvl': 2, 'hashmap_size': 19,
                                        'min_res': 16, 'max_res_coeff': 1024.0}
    assert sd['train_cfg']['intervals']['ckpt'] == 5000 and sd['dataset_cfg']['replica_cfg']['traj_ids'] == [1, 2]
    assert sd['renderer']['intr']['__dataclass__'] == 'Intrinsics'
    m2 = StyleTCNerf(ncfg, BBox.from_radius(2.0), 5)
    r2 = Renderer(m2, RendererConfig.llff(), intr, 2.0, raymarch_channels=8)
    opt2 = FusedAdam(m2, lr=1e-2, ema_decay=0.95)
    sc2 = LossScaler()
    assert C.restore(sd, r2, optim=opt2, scaler=sc2) == 41
    assert torch.equal(m2.arena.detach(), m.arena.detach()) and torch.equal(r2.density_grid, r.density_grid)
    assert r2.local_step == 33 and r2.mean_count == 77
    va = opt2.exp_avg[:m2.table_elems].view(m2.rows, 2, 2)
    assert torch.equal(va[:, 0, :], adam_state[0]['exp_avg']) and torch.equal(va[:, 1, :], adam_state[1]['exp_avg'])
    name, off, n = MLP_LAYOUT[2]
    assert torch.equal(opt2.exp_avg_sq[m2.table_elems + off: m2.table_elems + off + n], adam_state[4]['exp_avg_sq'].reshape(-1))
    assert torch.equal(opt2.ema[:m2.table_elems].view(m2.rows, 2, 2)[:, 1, :], shadow[1])
    assert opt2.step_count == 41 and opt2.ema_updates == 41 and abs(opt2.param_groups[0]['lr'] - 0.0097) < 1e-12
    st = sc2.state_dict()
    assert st['scale'] == 32768.0 and st['_growth_tracker'] == 12 and st['steps'] == 41
    # anything outside the schema is still refused
    bad = tmp_path / 'evil.pth'
    torch.save({'x': types.SimpleNamespace(a=1)}, bad)
    with pytest.raises(RuntimeError, match='nothing from a checkpoint file is executed'):
        C.load_checkpoint(bad)


def test_resident_dataset_targets():
    """nerfstyle_amd/dataset.py: per-pixel target rows (RGB + segment id) and poses resident on the device,
    gathered by pixel id -- the rows nerf_lib.generate_rays hands to calc_loss (nerf_lib.py:126-141)."""
    from nerfstyle_amd.common import Intrinsics
    from nerfstyle_amd.dataset import ResidentDataset
    from nerfstyle_amd.scene import load_room_cameras
    poses, _, _ = load_room_cameras()
    rng = np.random.default_rng(0)
    h, w, n = 12, 16, 5
    images = rng.random((n, 3, h, w), dtype=np.float32)
    segs = rng.integers(-1, 3, (n, h, w))
    intr = Intrinsics(h, w, 11.5, 11.5, w / 2, h / 2)
    res = ResidentDataset(images, poses[:n], intr, 2.0, 'cpu', seg_maps=segs, num_classes=3)
    assert len(res) == n and res.targets.shape == (n, h * w, 4)
    pix = torch.tensor([0, 5, w * h - 1])
    pose_t, tgt = res.sample(2, pix)
    assert tgt.shape == (3, 4) and torch.equal(pose_t, torch.from_numpy(poses[2]))
    want = np.concatenate((images[2], segs[2][None].astype(np.float32)), 0).reshape(4, -1)[:, pix.numpy()].T
    assert np.array_equal(tgt.numpy(), want)
    rgb_only = ResidentDataset(images, poses[:n], intr, 2.0, 'cpu')
    assert rgb_only.targets.shape == (n, h * w, 3)
    with pytest.raises(ValueError):
        ResidentDataset(images[:, :, :-1], poses[:n], intr, 2.0, 'cpu')


def test_tile_order_is_a_permutation_of_the_frame_in_tiles():
    from nerfstyle_amd.rays import tile_order
    for w, h, tw, th in ((1008, 756, 8, 8), (37, 21, 8, 8), (16, 16, 4, 2)):
        p = tile_order(w, h, tw, th).numpy()
        assert np.array_equal(np.sort(p), np.arange(w * h))
        y, x = np.divmod(p, w)
        tid = (y // th) * ((w + tw - 1) // tw) + x // tw
        assert np.all(np.diff(tid) >= 0)                                    # tile after tile
        first = p[:min(tw, w) * min(th, h)]
        assert np.array_equal(first, (np.arange(min(th, h))[:, None] * w + np.arange(min(tw, w))[None, :]).reshape(-1))
