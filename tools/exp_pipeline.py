"""Experiment: march + sample sort of step i+1 on a side stream while the backward of step i runs (they do not depend on the
parameters).  python tools/exp_pipeline.py  -> ms/step sequential vs pipelined on the bench workload (no occupancy updates)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from nerfstyle_amd.optim import FusedAdam
from nerfstyle_amd.rays import generate_rays

sys.argv = ['bench.py', '--no-occ-update']
args = bench.parse()
dev = torch.device('cuda:0')
torch.cuda.set_device(0)
model, r, rcfg, poses, intr = bench.build(args, dev, 0)
opt = FusedAdam(model, lr=1e-2, betas=(0.9, 0.999), eps=1e-15, ema_decay=0.95)
npix = intr.w * intr.h
gen = torch.Generator(device=dev); gen.manual_seed(1)
target = torch.rand(npix, 3, device=dev, generator=gen)
main = torch.cuda.current_stream()
side = torch.cuda.Stream()


def prepare(it):
    pix = torch.randperm(npix, device=dev, generator=gen)
    rays, _ = generate_rays(poses[(it * 7) % poses.shape[0]], r.intr, None, camera_flip=r.cfg.flip_camera, pix_subset=pix, device=dev)
    mt = r.march_train(rays)
    perm = r.model.sample_order(mt['xyzs'], mt['counter'], r._sort_prefix(mt['M'], mt['counter']))
    return pix, mt, perm


def finish(prep):
    pix, mt, perm = prep
    image, _, _ = r.shade_train(mt, perm)
    loss = torch.mean((image - target[pix]) ** 2) * 65536.0
    return loss


def run_early(steps=20, warm=5):
    """prefetch of step i+1 issued BEFORE the forward of step i"""
    nxt = prepare(0)
    for it in range(warm + steps):
        if it == warm:
            torch.cuda.synchronize(); t0 = time.perf_counter()
        cur = nxt
        side.wait_stream(main)
        with torch.cuda.stream(side):
            nxt = prepare(it + 1)
        loss = finish(cur)
        loss.backward()
        main.wait_stream(side)
        for t in (nxt[0], nxt[2], *[v for v in nxt[1].values() if torch.is_tensor(v)]):
            t.record_stream(main)
        opt.step(grad_scale=65536.0)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / steps * 1e3


def run(pipelined, steps=20, warm=5):
    nxt = prepare(0)
    ts = []
    for it in range(warm + steps):
        if it == warm:
            torch.cuda.synchronize(); t0 = time.perf_counter()
        cur = nxt
        loss = finish(cur)
        if pipelined:
            side.wait_stream(main)              # (inputs of the prefetch are ready; nothing of this step is needed)
            with torch.cuda.stream(side):
                nxt = prepare(it + 1)
            loss.backward()
            main.wait_stream(side)
            for t in (nxt[0], nxt[2], *[v for v in nxt[1].values() if torch.is_tensor(v)]):
                t.record_stream(main)
        else:
            loss.backward()
            nxt = prepare(it + 1)
        opt.step(grad_scale=65536.0)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / steps * 1e3


print('sequential %.2f ms/step' % run(False))
print('pipelined (under the backward)  %.2f ms/step' % run(True))
print('pipelined (from the forward on) %.2f ms/step' % run_early())
print('sequential %.2f ms/step' % run(False))
