#!/usr/bin/env python3
"""bench.py -- training throughput of the MI355X-native stylized-NeRF hot path.

    python bench.py --gpus N --steps K --warmup W            (N == 1)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...  (N > 1)

One "step" = one reconstruction-stage training step on one batch of synthetic LLFF 'room'-shaped
rays per GPU (BASELINE.json configs[1]: LLFF 'room' reconstruction, full 1008x756 frames, HIP
hash-encode + raymarch + fused MLP): device ray generation for `rays_per_gpu` pixels of a random
training pose -> near/far -> occupancy-grid march + scan compaction -> fused field forward ->
composite -> MSE + 0.001*CE loss (trainers/base.py:251-304) -> backward (composite bwd, fused field
bwd with table scatter) -> [RCCL all-reduce of the gradient arena when N > 1] -> fused Adam+EMA.
Nothing is skipped or cached inside the timed region; inputs (poses, bitfield, targets) are
resident in HBM before it starts.

Prints ONE JSON line on rank 0 (contract in the task statement) including `roofline` for the
dominant kernel (HIP-event timed on the launch stream inside the timed region) and `cpu_baseline`
(the pure-PyTorch CPU port of the same render step, oracle/torch_port.py, on a bounded sample).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import numpy as np
import torch


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=20)
    ap.add_argument('--warmup', type=int, default=3)
    ap.add_argument('--rays-per-gpu', type=int, default=1008 * 756,
                    help='rays per step per GPU (weak scaling); default = every pixel of one 1008x756 frame. The reference '
                         'trains on 4096-ray batches (cfgs/training/default.yaml:1); large batches are the documented '
                         'deviation that amortises the batch-independent optimiser / launch costs (SURVEY.md section 7)')
    ap.add_argument('--res-scale', type=int, default=2, help='2 = 1008x756 frames (configs[1]); 1 = 504x378')
    ap.add_argument('--num-classes', type=int, default=5)
    ap.add_argument('--table-dtype', choices=['f16', 'f32'], default='f16')
    ap.add_argument('--compute-dtype', choices=['f16', 'bf16'], default='f16')
    ap.add_argument('--samples-cap', type=int, default=160, help='sample-buffer capacity in samples per ray')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--graph', action='store_true',
                    help='replay the render+loss+backward part of the step as one captured hipGraph (no per-kernel '
                         'event timing, so no roofline object): for the small-batch series')
    ap.add_argument('--cpu-budget-s', type=float, default=20.0)
    ap.add_argument('--seed', type=int, default=69420)
    return ap.parse_args()


def cpu_baseline(budget_s, nc):
    """The reference's pure-PyTorch path restated (oracle/torch_port.py): BASELINE config 1 shape
    -- 200x200 patch rays of LLFF room frame 0, 64 samples/ray, fp32, forward + backward + SGD-free
    gradient (no optimiser) -- on the host cores.  Bounded: runs whole 200x200 passes in 10k-ray
    chunks until the budget is used; reports rays/s of the forward+backward pass."""
    from oracle import oracle as O
    from oracle import torch_port as TP
    from nerfstyle_amd.scene import load_room_cameras
    poses, intr, _ = load_room_cameras(1)
    ro, rd = O.generate_rays(poses[0], intr.w, intr.h, intr.fx, intr.fy, intr.cx, intr.cy, 3, patch=(200, 0, 200, 200))
    field = TP.Field(num_classes=nc, sparse_grad=True)

    def one(sl):
        o, d = torch.tensor(ro[sl]), torch.tensor(rd[sl])
        t0 = time.perf_counter()
        image, classes = TP.render_fixed_k(field, o, d, 0.2, 4.0, 64)
        t1 = time.perf_counter()
        loss = ((image - 0.5) ** 2).mean() + 1e-3 * classes.pow(2).mean()
        loss.backward()
        t2 = time.perf_counter()
        for p in field.parameters():
            p.grad = None
        return t1 - t0, t2 - t0

    # thread count: all host cores unless the (cgroup-limited) box runs faster on one -- calibrated on
    # a 250-ray chunk, the count actually used is what `cores` reports
    best = None
    try:
        ncpu = len(os.sched_getaffinity(0))
    except AttributeError:
        ncpu = os.cpu_count() or 1
    ncpu = max(1, min(ncpu, 16))      # the 1-GPU box's CPU share is 16 cores
    for nt in sorted({ncpu, 1}, reverse=True):
        torch.set_num_threads(nt)
        one(slice(0, 250))
        _, t = one(slice(250, 500))
        if best is None or t < best[1]:
            best = (nt, t)
    torch.set_num_threads(best[0])
    chunk = 500
    done, t_used, t_fwd_only, i = 0, 0.0, None, 0
    while t_used < budget_s and i < 200:
        lo = (i * chunk) % 40000
        tf, tt = one(slice(lo, lo + chunk))
        done += chunk
        t_used += tt
        t_fwd_only = tf if t_fwd_only is None else min(t_fwd_only, tf)
        i += 1
    return {
        'value': round(done / t_used / 1e6, 9), 'unit': 'Mrays/s', 'cores': int(torch.get_num_threads()), 'kind': 'port',
        'sample': '{} rays of the 200x200 patch of LLFF room frame 0 x 64 samples/ray, fp32 pure-PyTorch port '
                  '(oracle/torch_port.py), forward+backward, {:.1f} s; forward-only best {:.4f} Mrays/s'.format(
                      done, t_used, chunk / t_fwd_only / 1e6),
    }


def main():
    args = parse()
    from nerfstyle_amd import parallel as P
    # rehearsal knobs (NOT used by the driver): run N ranks on ONE GPU over gloo to exercise the N > 1
    # code path on a 1-GPU box -- NSR_BENCH_BACKEND=gloo NSR_BENCH_DEVICE=0
    backend = os.environ.get('NSR_BENCH_BACKEND')
    rank, local_rank, world = P.env_world()
    assert torch.cuda.is_available(), 'bench.py needs a HIP device (no CPU fallback for the product path)'
    dev_index = int(os.environ.get('NSR_BENCH_DEVICE', local_rank))
    torch.cuda.set_device(dev_index)
    rank, local_rank, world = P.init(backend)
    assert world == args.gpus, 'launch with torchrun --nproc-per-node {} (WORLD_SIZE={})'.format(args.gpus, world)
    dev = torch.device('cuda', dev_index)

    from nerfstyle_amd import profiling, raymarching
    from nerfstyle_amd.common import BBox
    from nerfstyle_amd.config import NetworkConfig, RendererConfig
    from nerfstyle_amd.optim import FusedAdam, exp_lr
    from nerfstyle_amd.renderer import Renderer
    from nerfstyle_amd.scene import load_room_cameras, synthetic_density_grid
    from nerfstyle_amd.style_nerf import StyleTCNerf

    nc = args.num_classes
    tdt = None if args.table_dtype == 'f16' else torch.float32
    cdt = torch.float16 if args.compute_dtype == 'f16' else torch.bfloat16
    # identical replicas on every rank: same seed for parameters and occupancy
    model = StyleTCNerf(NetworkConfig(), BBox.from_radius(2.0), nc, enc_dtype=tdt, use_dir=False, compute_dtype=cdt)
    poses_np, intr, _ = load_room_cameras(args.res_scale)
    rcfg = RendererConfig.llff()
    r = Renderer(model, rcfg, intr, 2.0, raymarch_channels=3 + nc, samples_per_ray_cap=args.samples_cap).to(dev)
    # seeded synthetic occupancy: 28 boxes (seed 0) give 64.1 emitted samples per ray on the room cameras
    grid = synthetic_density_grid(2.0, 128, n_boxes=28, seed=0)
    r.density_grid = torch.tensor(grid, device=dev)
    r.density_bitfield = raymarching.packbits(r.density_grid, 0.5)
    r.update_occ = False
    # loss scaling as the reference's GradScaler (init scale 65536) when the MFMA chain is f16
    loss_scale = 65536.0 if args.compute_dtype == 'f16' else 1.0
    opt = FusedAdam(model, lr=1e-2, betas=(0.9, 0.999), eps=1e-15, ema_decay=0.95)
    poses = torch.tensor(poses_np, device=dev)
    npix = intr.w * intr.h
    n_rays = min(args.rays_per_gpu, npix)
    gen = torch.Generator(device=dev)
    gen.manual_seed(args.seed * 1000003 + rank)            # rank-distinct pixel streams
    # synthetic targets resident in HBM: seeded uniform RGB + integer class ids per pixel
    tg = torch.Generator(device=dev)
    tg.manual_seed(args.seed)
    target_rgb = torch.rand(npix, 3, device=dev, generator=tg)
    target_cls = torch.randint(0, nc, (npix,), device=dev, generator=tg)
    total_samples = torch.zeros(1, dtype=torch.int64, device=dev)
    overflow = torch.zeros(1, dtype=torch.int64, device=dev)

    def loss_fn(out, pix):
        mse = torch.mean((out['rgb_map'] - target_rgb[pix]) ** 2)
        # cross entropy (trainers/base.py:281, nn.CrossEntropyLoss) as logsumexp - picked logit: same value and
        # gradient, without torch's one-block nll_loss reduction kernels (0.9 ms per 762 048-ray step)
        logits = out['classes']
        ce = (torch.logsumexp(logits, dim=1) - logits.gather(1, target_cls[pix][:, None])[:, 0]).mean() * 1e-3
        return (mse + ce) * (loss_scale / world)

    graphed = None
    if args.graph:
        from nerfstyle_amd.graph import GraphedRenderStep
        graphed = GraphedRenderStep(r, n_rays, loss_fn)

    def step(it):
        frame = (it * 7 + rank) % poses.shape[0]
        pix = torch.randperm(npix, device=dev, generator=gen)[:n_rays]
        if graphed is not None:
            loss = graphed(poses[frame], pix)
        else:
            out = r.render(poses[frame], None, training=True, pix_subset=pix)
            loss = loss_fn(out, pix)
            loss.backward()
        if world > 1:
            P.sync_gradients(model)
        opt.param_groups[0]['lr'] = exp_lr(1e-2, it, 30000)
        opt.step(grad_scale=loss_scale)
        cnt = r._last_counter
        total_samples.add_(cnt[0].to(torch.int64))
        overflow.add_((cnt[0] > r.sample_capacity(n_rays)).to(torch.int64))
        return loss

    for it in range(args.warmup):
        step(it)
    total_samples.zero_()
    overflow.zero_()
    profiling.reset()
    profiling.enabled = not args.graph      # event records cannot be captured into a graph
    P.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for it in range(args.steps):
        loss = step(args.warmup + it)
    torch.cuda.synchronize()
    P.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    profiling.enabled = False
    elapsed = P.max_over_ranks(elapsed, dev)
    prof = profiling.summary()

    if rank == 0:
        samples = int(total_samples.item())
        spr = samples / max(args.steps * n_rays, 1)
        value = world * n_rays * args.steps / elapsed / 1e6
        # ---- roofline of the dominant kernel ------------------------------------------------
        dom = max(prof.items(), key=lambda kv: kv[1][1])[0] if prof else None
        tb = 2 if args.table_dtype == 'f16' else 4
        # algorithmic bytes per sample (SURVEY 8d / DESIGN.md): forward gather = 2 enc x 16 lvl x 8 corners x
        # 2 feat x tb = 512 x tb; backward = read-modify-write of the same 512 fp32 gradient elements
        # (2 x 2048 B) + either the 128 B of encoded features the forward saved or the gather again
        saved = bool(getattr(model, 'save_features', False))
        bytes_per_sample = {'field_fwd': 512 * tb, 'field_bwd': 2 * 512 * 4 + (128 if saved else 512 * tb)}
        roofline = None
        # HBM traffic per launch from the PMC passes of profiles/ (separate rocprofv3 --pmc FETCH_SIZE /
        # --pmc WRITE_SIZE runs of this same command, corrected as MI355X_MICROARCH.md prescribes);
        # scaled by the sample count of this run, null when the profile does not match the configuration
        traffic_per_sample = {}
        atomic_req_per_sample = None
        try:
            with open(os.path.join(ROOT, 'profiles', 'r01_pmc_traffic.json')) as f:
                pm = json.load(f)
            if args.table_dtype == 'f16' and args.compute_dtype == 'f16':
                traffic_per_sample = {k[2:]: v['traffic_bytes_per_sample'] for k, v in pm['kernels'].items()}
                atomic_req_per_sample = pm['kernels'].get('k_field_bwd', {}).get('atomic_requests_per_sample')
        except (OSError, KeyError, ValueError):
            pass

        def roof(name):
            launches, tot_ms, avg_ms = prof[name]
            per_launch = bytes_per_sample[name] * samples / max(launches, 1)
            ach = per_launch / (avg_ms * 1e-3) / 1e9
            tr = traffic_per_sample.get(name)
            return {'bound': 'hbm', 'kernel': 'k_' + name, 'achieved': round(ach, 1), 'peak': 8000.0, 'unit': 'GB/s',
                    'frac': round(ach / 8000.0, 4),
                    'traffic': None if tr is None else int(tr * samples / max(launches, 1)),
                    'avg_launch_ms': round(avg_ms, 4), 'algorithmic_bytes_per_sample': bytes_per_sample[name]}
        if dom in bytes_per_sample:
            roofline = roof(dom)
        extra = {}
        if 'field_fwd' in prof:
            extra['hash_gather_fwd'] = roof('field_fwd')
            launches, tot_ms, avg_ms = prof['field_fwd']
            flops = 2.0 * (12544 + 64 * nc) * samples / max(launches, 1)      # SURVEY 8d, no padding counted
            tf = flops / (avg_ms * 1e-3) / 1e12
            extra['mlp_mfma_fwd'] = {'bound': 'mfma', 'kernel': 'k_field_fwd', 'achieved': round(tf, 1), 'peak': 2500.0,
                                     'unit': 'TFLOP/s', 'frac': round(tf / 2500.0, 4),
                                     'note': 'MLP FLOPs of the fused kernel over its whole duration (gather-bound kernel)'}
        if 'field_bwd' in prof and atomic_req_per_sample:
            # the resource k_field_bwd actually saturates: memory-side float-atomic requests (TCC_EA0_ATOMIC per sample
            # from the PMC pass in profiles/) against the chip-wide rate tools/atomic_footprint_bench.hip measures
            launches, tot_ms, avg_ms = prof['field_bwd']
            rate = atomic_req_per_sample * samples / max(launches, 1) / (avg_ms * 1e-3) / 1e9
            extra['atomic_requests_bwd'] = {'bound': 'memory-side atomic unit', 'kernel': 'k_field_bwd', 'achieved': round(rate, 2),
                                            'peak': 21.06, 'unit': 'G requests/s', 'frac': round(rate / 21.06, 4),
                                            'requests_per_sample': atomic_req_per_sample}
        result = {
            'metric': 'train Mrays/sec', 'value': round(value, 4), 'unit': 'Mrays/s', 'n_gpus': world, 'steps': args.steps,
            'warmup': args.warmup, 'ms_per_step': round(elapsed / args.steps * 1e3, 3), 'higher_is_better': True,
            'scaling': 'weak', 'vs_baseline': None, 'dtype': args.compute_dtype, 'data': 'synthetic',
            'config': {
                'workload': "LLFF 'room' reconstruction stage, {}x{} frames, {} rays/step/GPU, synthetic occupancy "
                            '(28 seeded boxes), {:.1f} samples/ray'.format(intr.w, intr.h, n_rays, spr),
                'rays_per_step_per_gpu': n_rays, 'samples_per_ray': round(spr, 2), 'max_steps': rcfg.max_steps,
                'num_classes': nc, 'table_dtype': args.table_dtype, 'mfma_dtype': args.compute_dtype,
                'params': int(model.arena.numel()), 'parallelism': 'rays sharded x{} + RCCL all-reduce'.format(world),
                'sample_capacity_overflows': int(overflow.item()), 'final_loss': float(loss.detach()) / loss_scale * world,
            },
            'kernel_ms_per_step': {k: round(v[1] / args.steps, 4) for k, v in sorted(prof.items())},
            'roofline': roofline,
            'rooflines_other': extra,
        }
        print('[bench] gpu leg done: ' + json.dumps(result), file=sys.stderr, flush=True)
        if world == 1 and not args.no_cpu_baseline:
            result['cpu_baseline'] = cpu_baseline(args.cpu_budget_s, nc)
        print(json.dumps(result), flush=True)
    P.barrier()


if __name__ == '__main__':
    main()
