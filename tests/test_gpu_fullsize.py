"""GPU: the bench workload's own shape (BASELINE config 2: one full 1008x756 frame = 762 048 rays, sample capacity
160 per ray = 121.9 M slots) checked through size-independent properties and oracle spot checks -- the oracle finishes
only a few thousand rays in seconds, so:

  * march: `rays` offsets are the exclusive scan of the counts, the total is counter[0], nothing is written past it;
    counts and bit-exact positions / deltas of a 2 000-ray subset equal the sequential oracle run on that subset alone;
  * field: 20 000 random samples of the 48 M equal the rounding-emulating oracle field (f16 tables, f16 MFMA);
  * composite: weights_sum + prod(1 - alpha) = 1 on every ray that did not stop early; dropped / empty rays are background.
"""
import numpy as np
import pytest
import torch

from helpers import rel_l2

pytestmark = pytest.mark.gpu


def test_full_frame_762048_rays(O, dev):
    from nerfstyle_amd import _lib as L
    from nerfstyle_amd import raymarching
    from nerfstyle_amd.common import BBox
    from nerfstyle_amd.config import NetworkConfig, RendererConfig
    from nerfstyle_amd.rays import generate_rays
    from nerfstyle_amd.renderer import Renderer, _render_train
    from nerfstyle_amd.scene import load_room_cameras, synthetic_density_grid
    from nerfstyle_amd.style_nerf import StyleTCNerf
    from oracle import torch_port as TP
    nc, cap = 5, 160
    ref = TP.Field(num_classes=nc, table_scale=0.5)
    m = StyleTCNerf(NetworkConfig(), BBox.from_radius(2.0), nc, enc_dtype=None, use_dir=False)       # f16 gather tables
    sd = m.state_dict()
    sd.update({'x_density_embedder.embeddings': ref.emb_density.detach(), 'x_color_embedder.embeddings': ref.emb_color.detach(),
               'density_net.params': ref.p_density.detach(), 'color1_net.params': ref.p_color1.detach(),
               'color2_net.params': ref.p_color2.detach(), 'class_net.params': ref.p_class.detach()})
    m.load_state_dict(sd)
    poses, intr, _ = load_room_cameras(2)
    assert (intr.w, intr.h) == (1008, 756)
    cfg = RendererConfig.llff()
    r = Renderer(m, cfg, intr, 2.0, raymarch_channels=3 + nc, samples_per_ray_cap=cap).to(dev)
    grid = synthetic_density_grid(2.0, 128, n_boxes=28, seed=0)                  # the bench scene
    r.density_grid = torch.tensor(grid, device=dev)
    r.density_bitfield = raymarching.packbits(r.density_grid, 0.5)
    bits = r.density_bitfield.cpu().numpy()

    rays, _ = generate_rays(torch.tensor(poses[0], device=dev), intr, camera_flip=3, device=dev)
    N = rays.origins.shape[0]
    assert N == 762048
    M = r.sample_capacity(N)
    nears, fars = raymarching.near_far_from_aabb(rays.origins, rays.dirs, r.aabb, cfg.min_near)
    # ---- march into sentinel-filled buffers --------------------------------------------------
    sent = float('nan')
    xyzs = torch.full((M, 3), sent, dtype=torch.float32, device=dev)
    deltas = torch.full((M, 4), sent, dtype=torch.float32, device=dev)
    rays_info = torch.empty(N, 3, dtype=torch.int32, device=dev)
    counter = torch.zeros(2, dtype=torch.int32, device=dev)
    ws_bytes = int(L.lib().nsr_march_rays_train_workspace_bytes(N, 2.0, cfg.max_steps))
    wsb = torch.empty((ws_bytes + 3) // 4, dtype=torch.int32, device=dev)
    L.check(L.lib().nsr_march_rays_train(L.p(rays.origins), L.p(rays.dirs), None, L.p(r.density_bitfield), 2.0, 0.0, cfg.max_steps,
                                         0, N, r.cascade, cfg.grid_size, M, L.p(nears), L.p(fars), L.p(xyzs), None, L.p(deltas),
                                         L.p(rays_info), L.p(counter), None, L.p(wsb), L.stream()))
    total = int(counter[0])
    assert int(counter[1]) == N and 40 * N < total < M
    cnt = rays_info[:, 2].long()
    off = rays_info[:, 1].long()
    assert torch.equal(rays_info[:, 0].long(), torch.arange(N, device=dev))
    assert torch.equal(off, torch.cumsum(cnt, 0) - cnt) and int(cnt.sum()) == total
    assert int(cnt.max()) <= cfg.max_steps
    assert torch.isfinite(xyzs[:total]).all() and torch.isfinite(deltas[:total, :2]).all()
    assert torch.isnan(xyzs[total:]).all() and torch.isnan(deltas[total:]).all()          # nothing written past counter[0]
    # ---- 2 000-ray subset vs the sequential oracle --------------------------------------------
    rng = np.random.default_rng(0)
    sub = np.sort(rng.choice(N, 2000, replace=False))
    ro, rd = rays.origins[sub].cpu().numpy(), rays.dirs[sub].cpu().numpy()
    ro_o, rd_o = O.generate_rays(poses[0], intr.w, intr.h, intr.fx, intr.fy, intr.cx, intr.cy, 3, pix_indices=sub)
    assert np.abs(ro - ro_o).max() == 0 and np.abs(rd - rd_o).max() <= 2.4e-7           # <= 2 ulp, as in the golden test
    near_o, far_o = O.near_far_from_aabb(ro, rd, np.array([-2, -2, -2, 2, 2, 2], np.float32), 0.2)
    assert np.array_equal(nears[sub].cpu().numpy(), near_o) and np.array_equal(fars[sub].cpu().numpy(), far_o)
    xo, _, dlo, rays_o, cnt_o = O.march_rays_train(ro, rd, 2.0, bits, 2, 128, near_o, far_o, 1024)
    cnt_h, off_h = cnt[sub].cpu().numpy(), off[sub].cpu().numpy()
    assert np.array_equal(cnt_h, rays_o[:, 2])
    xyz_h = torch.cat([xyzs[o:o + c] for o, c in zip(off_h.tolist(), cnt_h.tolist())]).cpu().numpy()
    dl_h = torch.cat([deltas[o:o + c, :2] for o, c in zip(off_h.tolist(), cnt_h.tolist())]).cpu().numpy()
    assert np.array_equal(xyz_h, xo[:int(cnt_o[0])]) and np.array_equal(dl_h, dlo[:int(cnt_o[0]), :2])
    # ---- field at 48 M samples: random sample vs the oracle ----------------------------------
    with torch.no_grad():
        sigmas, rgbs = m.field(xyzs, sigma_only=False, m_dev=counter, density_scale=1.0)
    pick = torch.tensor(rng.choice(total, 20000, replace=False), device=dev)
    fp = O.FieldParams(ref.emb_density.detach().numpy(), ref.emb_color.detach().numpy(), ref.p_density.detach().numpy(),
                       ref.p_color1.detach().numpy(), ref.p_color2.detach().numpy(), ref.p_class.detach().numpy(), ref.offsets,
                       ref.pls, num_classes=nc)
    out_o, sig_o, _ = O.field_forward(fp, xyzs[pick].cpu().numpy(), half='f16', table_half=True)
    assert rel_l2(sigmas[pick].cpu().numpy(), sig_o) < 3e-3
    assert rel_l2(rgbs[pick].cpu().numpy(), out_o) < 3e-3
    # ---- composite: ws + prod(1 - alpha) = 1 where the ray did not stop early ------------------
    with torch.no_grad():
        rgb_map, depth, classes, ws = _render_train(sigmas, rgbs, deltas, rays_info, nears, fars, cfg.t_thresh)
        image = torch.cat((rgb_map - (1 - ws).unsqueeze(-1), classes), 1)
    tau = torch.zeros(M + 1, dtype=torch.float64, device=dev)
    tau[1:total + 1] = torch.cumsum((sigmas[:total].double() * deltas[:total, 0].double()), 0)
    T_final = torch.exp(-(tau[off + cnt] - tau[off]))
    live = (cnt > 0) & (T_final > 4 * cfg.t_thresh)
    assert int(live.sum()) > N // 4
    assert float((ws[live].double() + T_final[live] - 1.0).abs().max()) < 2e-4
    empty = cnt == 0
    assert int(empty.sum()) > 0 and float(ws[empty].abs().max()) == 0.0 and float(image[empty].abs().max()) == 0.0
    assert float((rgb_map[empty] - 1.0).abs().max()) == 0.0                # white background
    assert torch.isfinite(image).all() and torch.isfinite(depth).all()
