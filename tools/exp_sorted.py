"""Experiment (GPU): field forward / backward time of one full 1008x756 frame of the bench scene with the samples in ray
order vs physically permuted into Morton order of their position (10 bits per axis), and what a radix sort of that
many (key, index) pairs costs with torch.sort.  Decides whether a sorted-sample path is worth building."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch


def spread10(v):
    v = v & 0x3FF
    v = (v | (v << 16)) & 0x030000FF
    v = (v | (v << 8)) & 0x0300F00F
    v = (v | (v << 4)) & 0x030C30C3
    v = (v | (v << 2)) & 0x09249249
    return v


def main():
    from nerfstyle_amd import raymarching
    from nerfstyle_amd.common import BBox
    from nerfstyle_amd.config import NetworkConfig, RendererConfig
    from nerfstyle_amd.rays import generate_rays
    from nerfstyle_amd.renderer import Renderer
    from nerfstyle_amd.scene import load_room_cameras, synthetic_density_grid
    from nerfstyle_amd.style_nerf import StyleTCNerf
    dev = torch.device('cuda:0')
    nc = 5
    m = StyleTCNerf(NetworkConfig(), BBox.from_radius(2.0), nc, enc_dtype=None, use_dir=False)
    poses, intr, _ = load_room_cameras(2)
    cfg = RendererConfig.llff()
    r = Renderer(m, cfg, intr, 2.0, raymarch_channels=3 + nc, samples_per_ray_cap=160).to(dev)
    r.density_grid = torch.tensor(synthetic_density_grid(2.0, 128, n_boxes=28, seed=0), device=dev)
    r.density_bitfield = raymarching.packbits(r.density_grid, 0.5)
    rays, _ = generate_rays(torch.tensor(poses[0], device=dev), intr, camera_flip=3, device=dev)
    N = rays.origins.shape[0]
    nears, fars = raymarching.near_far_from_aabb(rays.origins, rays.dirs, r.aabb, cfg.min_near)
    counter = torch.zeros(2, dtype=torch.int32, device=dev)
    xyzs, _, deltas, rays_info = raymarching.march_rays_train_nosync(rays.origins, rays.dirs, 2.0, r.density_bitfield, 2, 128,
                                                                     nears, fars, N * 160, counter, 0., 1024)
    total = int(counter[0])
    xyz = xyzs[:total].contiguous()
    print('samples', total)
    u = ((xyz + 2.0) / 4.0 + 1.0) / 2.0
    q = (u * 1024.0).clamp(0, 1023).to(torch.int32)
    key = spread10(q[:, 0]) | (spread10(q[:, 1]) << 1) | (spread10(q[:, 2]) << 2)
    torch.cuda.synchronize()
    for _ in range(3):
        t0 = time.perf_counter()
        ks, perm = torch.sort(key)
        torch.cuda.synchronize()
        print('torch.sort of {} int32 keys: {:.2f} ms'.format(total, (time.perf_counter() - t0) * 1e3))
    m._ensure_grad()
    g = torch.Generator(device=dev)
    g.manual_seed(0)
    M = xyzs.shape[0]
    gs = torch.randn(M, device=dev, generator=g) * 1e-3
    gr = torch.randn(M, 3 + nc, device=dev, generator=g) * 1e-3

    def run(p, tag):
        for it in range(4):
            e = [torch.cuda.Event(enable_timing=True) for _ in range(4)]
            e[0].record()
            pp = m.sample_order(xyzs, m_dev=counter, sort_prefix=(0 if os.environ.get('EXP_IDENTITY_PERM') else total + 65536)) if p else None
            e[1].record()
            sig, rgb = m.field(xyzs, False, counter, perm=pp)
            e[2].record()
            torch.autograd.backward([sig, rgb], [gs, gr])
            e[3].record()
            torch.cuda.synchronize()
            print('{}: order {:.2f} ms  fwd {:.2f} ms  bwd {:.2f} ms'.format(tag, e[0].elapsed_time(e[1]), e[1].elapsed_time(e[2]),
                                                                            e[2].elapsed_time(e[3])), flush=True)
        gsum = float(m.arena.grad.double().abs().sum())
        m.arena.grad.zero_()
        return gsum

    a = run(False, 'buffer (ray) order, run tracker ')
    b = run(True, 'nsr_sample_order, lattice tiles ')
    print('grad abs-sum ray {:.6e} sorted {:.6e}'.format(a, b))


if __name__ == '__main__':
    main()
