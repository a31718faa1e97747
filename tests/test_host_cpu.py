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
