"""Times a full-frame inference render (render.py:97 path: Renderer.render(pose) -> render_test) and
the same frame through the training path without gradients; prints ms and PSNR between the two."""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from nerfstyle_amd import raymarching
from nerfstyle_amd.common import BBox
from nerfstyle_amd.config import NetworkConfig, RendererConfig
from nerfstyle_amd.renderer import Renderer
from nerfstyle_amd.scene import load_room_cameras, synthetic_density_grid
from nerfstyle_amd.style_nerf import StyleTCNerf

dev = torch.device('cuda:0')
scale = int(sys.argv[1]) if len(sys.argv) > 1 else 2
model = StyleTCNerf(NetworkConfig(), BBox.from_radius(2.0), 5, enc_dtype=None, use_dir=False)
with torch.no_grad():
    model.arena[:model.table_elems].uniform_(-0.5, 0.5)
poses, intr, _ = load_room_cameras(scale)
r = Renderer(model, RendererConfig.llff(), intr, 2.0, raymarch_channels=8, samples_per_ray_cap=192).to(dev)
r.density_grid = torch.tensor(synthetic_density_grid(2.0, 128, 28, 0), device=dev)
r.density_bitfield = raymarching.packbits(r.density_grid, 0.5)
r.update_occ = False
pose = torch.tensor(poses[0], device=dev)
for name, training in (('render_test', False), ('render_train(no_grad)', True)):
    with torch.no_grad():
        out = r.render(pose, None, training=training)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(3):
            out = r.render(pose, None, training=training)
        torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / 3 * 1e3
    print('{:24s} {}x{}: {:8.2f} ms/frame  ({:.2f} Mrays/s)'.format(name, intr.w, intr.h, ms, intr.w * intr.h / ms / 1e3))
    if training:
        b = out['rgb_map']
    else:
        a = out['rgb_map']
mse = float(((a - b) ** 2).mean())
print('PSNR(render_test vs render_train) = {:.1f} dB'.format(-10 * np.log10(max(mse, 1e-12))))
