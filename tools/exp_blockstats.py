"""Experiment (GPU): population statistics of the nsr_sample_order blocks on the bench frame."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from tools.exp_sorted import spread10


def main():
    from nerfstyle_amd import raymarching
    from nerfstyle_amd.common import BBox
    from nerfstyle_amd.config import NetworkConfig, RendererConfig
    from nerfstyle_amd.rays import generate_rays
    from nerfstyle_amd.renderer import Renderer
    from nerfstyle_amd.scene import load_room_cameras, synthetic_density_grid
    from nerfstyle_amd.style_nerf import StyleTCNerf
    dev = torch.device('cuda:0')
    m = StyleTCNerf(NetworkConfig(), BBox.from_radius(2.0), 5, enc_dtype=None, use_dir=False)
    poses, intr, _ = load_room_cameras(2)
    cfg = RendererConfig.llff()
    r = Renderer(m, cfg, intr, 2.0, raymarch_channels=8, samples_per_ray_cap=160).to(dev)
    r.density_grid = torch.tensor(synthetic_density_grid(2.0, 128, n_boxes=28, seed=0), device=dev)
    r.density_bitfield = raymarching.packbits(r.density_grid, 0.5)
    rays, _ = generate_rays(torch.tensor(poses[0], device=dev), intr, camera_flip=3, device=dev)
    N = rays.origins.shape[0]
    nears, fars = raymarching.near_far_from_aabb(rays.origins, rays.dirs, r.aabb, cfg.min_near)
    counter = torch.zeros(2, dtype=torch.int32, device=dev)
    xyzs, _, deltas, rays_info = raymarching.march_rays_train_nosync(rays.origins, rays.dirs, 2.0, r.density_bitfield, 2, 128,
                                                                     nears, fars, N * 160, counter, 0., 1024)
    total = int(counter[0])
    xyz = xyzs[:total]
    t = (xyz - rays.origins[0]).norm(dim=1)
    u = ((xyz + 2.0) / 4.0 + 1.0) / 2.0
    for bits in (10, 9, 8):
        q = (u * (1 << bits)).clamp(0, (1 << bits) - 1).to(torch.int64)
        key = q[:, 0] | (q[:, 1] << bits) | (q[:, 2] << (2 * bits))
        uk, inv, cnt = torch.unique(key, return_inverse=True, return_counts=True)
        pop = cnt[inv].float()                       # population of each sample's block
        print('bits', bits, 'blocks', uk.numel(), 'mean samples/block', total / uk.numel())
        for lo, hi in ((1, 4), (4, 8), (8, 16), (16, 32), (32, 64), (64, 128), (128, 1 << 30)):
            f = ((pop >= lo) & (pop < hi)).float().mean().item()
            print('   samples in blocks of population [%d,%d): %.3f' % (lo, hi, f))
    print('t quantiles', torch.quantile(t[::97], torch.tensor([0.1, 0.25, 0.5, 0.75, 0.9], device=dev)).tolist())


if __name__ == '__main__':
    main()
