"""Host logic that needs no GPU: sizing, parameter layout, checkpoints, the synthetic scene."""
import numpy as np
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
