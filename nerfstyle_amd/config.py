"""Hot-path constants with the field names and defaults of the reference's config dataclasses
(config.py: NetworkConfig, RendererConfig; defaults from cfgs/network/default.yaml,
cfgs/renderer/default.yaml, cfgs/renderer/llff.yaml).  Only what the renderer / model read."""
from dataclasses import dataclass, field
from typing import Optional


@dataclass
class PosEncConfig:
    n_lvls: int = 16
    n_feats_per_lvl: int = 2
    hashmap_size: int = 19
    min_res: int = 16
    max_res_coeff: int = 1024


@dataclass
class NetworkConfig:
    network_seed: Optional[int] = 80000
    density_out_dims: int = 16
    density_hidden_dims: int = 64
    density_hidden_layers: int = 1
    rgb_hidden_dims: int = 64
    rgb_hidden_layers: int = 2
    pos_enc: PosEncConfig = field(default_factory=PosEncConfig)
    dir_enc_sh_deg: int = 4


@dataclass
class RendererConfig:
    grid_size: int = 128
    grid_bsize: Optional[int] = None
    update_iter: int = 16
    min_near: float = 0.2
    t_thresh: float = 1e-4
    use_ndc: bool = False
    flip_camera: int = 0
    max_steps: int = 1024
    update_thres: int = 256
    density_scale: float = 1
    density_thresh: float = 10
    density_decay: float = 0.95

    @classmethod
    def llff(cls):
        """cfgs/renderer/llff.yaml"""
        return cls(use_ndc=False, flip_camera=3)
