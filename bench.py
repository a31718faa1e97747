#!/usr/bin/env python3
"""bench.py -- training throughput of the MI355X-native stylized-NeRF hot path.

    python bench.py --gpus N --steps K --warmup W            (N == 1)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...  (N > 1)

Default workload = BASELINE.json configs[1]: LLFF 'room' reconstruction stage, full 1008x756 frames, one frame
(762 048 rays) per step per GPU.  One "step" = device ray generation for the step's pixels of a training pose ->
near/far -> [every 16 steps: device-side occupancy update, 4.19 M sigma queries] -> occupancy-grid march + scan
compaction -> fused field forward -> composite -> MSE + 0.001*CE loss (trainers/base.py:251-304) [+ sparsity term,
--sparsity-lambda, trainers/base.py:285-291,409-413] -> backward (composite bwd, fused field bwd with table scatter) ->
[RCCL all-reduce of the gradient arena when N > 1] -> fused Adam + EMA.  Nothing is skipped or cached inside the timed
region; inputs (poses, bitfield, targets) are resident in HBM before it starts.

Other workloads (BASELINE.json configs[2..4]):
    --stage style                       configs[2]: one stylisation iteration per step (full-frame pass + VGG16/semantic-NNFM
                                        loss in PyTorch + 24 deferred-backprop patches, colour table only), 1008x756
    --scene fern --sparsity-lambda 0.01 configs[3]: fern cameras + the sparsity term (per-GPU part; the driver scales it out)
    --compute-dtype bf16 --graph --rays-per-gpu 4096 | 40000
                                        configs[4]: bf16 MLP + fp32 composite, render+loss+backward replayed as one hipGraph

Prints ONE JSON line on rank 0 (contract in the task statement) including `roofline` for the dominant kernel and
`cpu_baseline` (the pure-PyTorch CPU port of the same render step, oracle/torch_port.py, on a bounded sample).
The GPU leg is printed to stderr as soon as it is done; the CPU leg runs after it, inside a hard time budget.
"""
import argparse
import gc
import json
import os
import statistics
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import numpy as np
import torch

PROFILE_JSON = os.path.join(ROOT, 'profiles', 'r03_pmc_traffic.json')


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=50)
    ap.add_argument('--warmup', type=int, default=5)
    ap.add_argument('--stage', choices=['recon', 'style'], default='recon')
    ap.add_argument('--scene', choices=['room', 'fern'], default='room')
    ap.add_argument('--rays-per-gpu', type=int, default=1008 * 756,
                    help='rays per step per GPU (weak scaling); default = every pixel of one 1008x756 frame. The reference '
                         'trains on 4096-ray batches (cfgs/training/default.yaml:1); large batches are the documented '
                         'deviation that amortises the batch-independent optimiser / launch costs (SURVEY.md section 7)')
    ap.add_argument('--res-scale', type=int, default=2, help='2 = 1008x756 frames (configs[1]); 1 = 504x378')
    ap.add_argument('--num-classes', type=int, default=5)
    ap.add_argument('--table-dtype', choices=['f16', 'f32'], default='f16')
    ap.add_argument('--compute-dtype', choices=['f16', 'bf16'], default='f16')
    ap.add_argument('--samples-cap', type=int, default=160, help='sample-buffer capacity in samples per ray')
    ap.add_argument('--sparsity-lambda', type=float, default=0.0, help='cfgs: --sparsity_lambda (0.01 in BASELINE configs[3])')
    ap.add_argument('--no-patch-graphs', action='store_true', help='style stage: launch the patch kernels eagerly (host-bound) instead of replaying graphs')
    ap.add_argument('--fp32-loss', action='store_true', help='style stage: VGG + style loss in fp32 instead of autocast')
    ap.add_argument('--no-occ-update', action='store_true', help='leave the periodic occupancy update out of the step')
    ap.add_argument('--occ-phase', choices=['steady', 'warmup'], default='steady',
                    help="which phase of the reference's update schedule the timed steps are in: 'steady' = after the first "
                         "update_thres (256) steps, partial updates (2 x H^3/4 sigma queries per cascade every 16 steps: what a training "
                         "run spends all but its first 256 steps in; SURVEY 8d: 'steady state, after occupancy warm-up'); 'warmup' = "
                         "the first 256 steps, full updates (C x H^3 queries)")
    ap.add_argument('--sort-samples', choices=['auto', 'on', 'off'], default='auto',
                    help="spatially ordered table scatter in the backward (nsr_sample_order): 'auto' = batches of >= 90 000 rays, dense pixel sets from 16 384")
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--psnr-rays', type=int, default=4096, help='rays of the PSNR-vs-oracle check after the timed region (0: skip)')
    ap.add_argument('--no-loss-scaler', action='store_true', help='constant loss scale, no inf/nan check (round 2 behaviour)')
    ap.add_argument('--style-backprop', choices=['resident', 'deferred'], default='resident',
                    help="style stage: 'resident' = ONE render of the frame with autograd, activations kept in HBM, d loss / d rgb "
                         "back through it (stylize.resident_backprop_step); 'deferred' = the reference's structure: no-grad pass, then "
                         "24 patch re-renders with autograd (trainers/style.py:162-204) -- same gradient")
    ap.add_argument('--side-march', action='store_true',
                    help="N == 1, eager steps: ray generation + march + compaction + sample sort of step i+1 on a side stream beside "
                         "the backward of step i (Renderer.begin_train_on): 34.9 -> 33.5 ms per full-frame step")
    ap.add_argument('--graph', action='store_true',
                    help='replay the render+loss+backward part of the step as one captured hipGraph (small-batch series)')
    ap.add_argument('--cpu-budget-s', type=float, default=12.0)
    ap.add_argument('--max-steps', type=int, default=None, help='march steps per ray (1024; README stylisation command: 512)')
    ap.add_argument('--seed', type=int, default=69420)
    return ap.parse_args()


# ---------------------------------------------------------------------------------------------------------------------
# CPU baseline
# ---------------------------------------------------------------------------------------------------------------------
def cpu_baseline(budget_s, nc):
    """The reference's pure-PyTorch path restated (oracle/torch_port.py) on BASELINE configs[0]'s shape -- rays of the
    200x200 patch of LLFF room frame 0, 64 samples/ray, fp32 -- forward + backward, on the host cores.
    SURVEY 8d: all host cores, 1 warm-up, median of >= 5 repetitions.  Bounded: a repetition is a slice of the 40 000
    rays sized from a calibration chunk so that warm-up + 5 repetitions fit `budget_s`; the thread count is the one
    that ran the calibration chunk faster (the 1-GPU box's cgroup gives 16 logical CPUs a fractional share, where one
    thread can beat sixteen), and the line says which and why."""
    from oracle import oracle as O
    from oracle import torch_port as TP
    from nerfstyle_amd.scene import load_cameras
    t_start = time.perf_counter()
    poses, intr, _ = load_cameras('room', 1)
    ro, rd = O.generate_rays(poses[0], intr.w, intr.h, intr.fx, intr.fy, intr.cx, intr.cy, 3, patch=(200, 0, 200, 200))
    field = TP.Field(num_classes=nc, sparse_grad=True)

    def one(lo, n):
        o, d = torch.tensor(ro[lo:lo + n]), torch.tensor(rd[lo:lo + n])
        t0 = time.perf_counter()
        image, classes = TP.render_fixed_k(field, o, d, 0.2, 4.0, 64)
        t1 = time.perf_counter()
        loss = ((image - 0.5) ** 2).mean() + 1e-3 * classes.pow(2).mean()
        loss.backward()
        t2 = time.perf_counter()
        for p in field.parameters():
            p.grad = None
        return t1 - t0, t2 - t0

    try:
        affinity = len(os.sched_getaffinity(0))
    except AttributeError:
        affinity = os.cpu_count() or 1
    cand = sorted({min(affinity, 16), 1}, reverse=True)
    calib = {}
    for nt in cand:
        torch.set_num_threads(nt)
        one(0, 50)
        calib[nt] = one(50, 100)[1] / 100.0                  # seconds per ray, forward + backward
    nt = min(calib, key=calib.get)
    torch.set_num_threads(nt)
    left = budget_s - (time.perf_counter() - t_start)
    rep_rays = int(max(50, min(40000, left / 6.5 / calib[nt])))
    reps, fwd = [], []
    one(0, rep_rays)                                          # warm-up at the repetition size
    for i in range(5):
        lo = ((i + 1) * rep_rays) % max(1, 40000 - rep_rays)
        tf, tt = one(lo, rep_rays)
        reps.append(rep_rays / tt)
        fwd.append(rep_rays / tf)
        if time.perf_counter() - t_start > 2.0 * budget_s and len(reps) >= 3:     # hard stop: never run away
            break
    why = ('{} threads beat 1 thread on the calibration chunk ({:.0f} vs {:.0f} rays/s)'.format(nt, 1 / calib[nt], 1 / calib[1])
           if nt != 1 else '1 thread beat {} threads on the calibration chunk ({:.0f} vs {:.0f} rays/s): the box gives its {} '
           'logical CPUs a fractional cgroup share'.format(cand[0], 1 / calib[1], 1 / calib[cand[0]], affinity)) if len(cand) > 1 else 'one CPU'
    return {
        'value': round(statistics.median(reps) / 1e6, 9), 'unit': 'Mrays/s', 'cores': int(nt), 'kind': 'port',
        'sample': 'median of {} repetitions (1 warm-up) of {} rays of the 200x200 patch of LLFF room frame 0 x 64 samples/ray, '
                  'fp32 pure-PyTorch port (oracle/torch_port.py), forward+backward; forward-only median {:.6f} Mrays/s; '
                  'os.cpu_count()={}, affinity={}, threads used={} ({}); {:.1f} s'.format(
                      len(reps), rep_rays, statistics.median(fwd) / 1e6, os.cpu_count(), affinity, nt, why,
                      time.perf_counter() - t_start),
    }


# ---------------------------------------------------------------------------------------------------------------------
# common set-up
# ---------------------------------------------------------------------------------------------------------------------
def build(args, dev, rank):
    from nerfstyle_amd import raymarching
    from nerfstyle_amd.common import BBox
    from nerfstyle_amd.config import NetworkConfig, RendererConfig
    from nerfstyle_amd.renderer import Renderer
    from nerfstyle_amd.scene import load_cameras, synthetic_density_grid
    from nerfstyle_amd.style_nerf import StyleTCNerf
    nc = args.num_classes
    tdt = None if args.table_dtype == 'f16' else torch.float32
    cdt = torch.float16 if args.compute_dtype == 'f16' else torch.bfloat16
    # identical replicas on every rank: same seed for parameters and occupancy
    model = StyleTCNerf(NetworkConfig(), BBox.from_radius(2.0), nc, enc_dtype=tdt, use_dir=False, compute_dtype=cdt)
    poses_np, intr, _ = load_cameras(args.scene, args.res_scale)
    rcfg = RendererConfig.llff()
    if args.max_steps:
        rcfg.max_steps = args.max_steps
    r = Renderer(model, rcfg, intr, 2.0, raymarch_channels=3 + nc, samples_per_ray_cap=args.samples_cap).to(dev)
    r.manual_seed(args.seed)
    # Seeded synthetic occupancy: 28 boxes (seed 0) give ~64 emitted samples per ray on the room cameras.  The
    # march reads THIS bitfield (a random-initialised model has no scene); the periodic occupancy update still runs at full
    # cost from the model itself and maintains density_grid / density_bitfield / mean_density (Renderer.pin_march_bitfield).
    grid = torch.tensor(synthetic_density_grid(2.0, 128, n_boxes=28, seed=0), device=dev)
    scene_bits = raymarching.packbits(grid, 0.5)
    if args.no_occ_update:
        r.density_grid = grid
        r.density_bitfield = scene_bits
        r.update_occ = False
    else:
        r.pin_march_bitfield(scene_bits)
        r.update_occ = True
    r.sort_samples = {'auto': 'auto', 'on': True, 'off': False}[args.sort_samples]
    if r.update_occ and args.occ_phase == 'steady':
        # one full update (the grid the partial updates refine), then continue the schedule from step update_thres on
        r.update_state()
        r.local_step = rcfg.update_thres
    return model, r, rcfg, torch.tensor(poses_np, device=dev), intr


def profile_traffic(args):
    """HBM traffic / atomic requests per sample from the committed PMC passes (profiles/): NOT measured in this run --
    tagged as such, with the profile's commit and sample count so that a stale file is detectable."""
    try:
        with open(PROFILE_JSON) as f:
            pm = json.load(f)
    except (OSError, ValueError):
        return None
    if not (args.table_dtype == 'f16' and args.compute_dtype == 'f16' and args.stage == 'recon' and args.sort_samples == 'auto'
            and args.rays_per_gpu >= 1008 * 756 and args.scene == 'room' and args.sparsity_lambda == 0):
        return None
    return pm


def profile_kernel_share(name, group):
    """TotalDurationNs of kernel `name` over that of the kernels in `group`, from the committed rocprofv3 summary (None if absent)."""
    import csv
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'profiles', 'r03_bench_kernel_stats.csv')
    try:
        tot = {}
        with open(path) as f:
            for r in csv.DictReader(f):
                for g in group:
                    if g in r['Name']:
                        tot[g] = tot.get(g, 0.0) + float(r['TotalDurationNs'])
        return tot[name] / sum(tot.values()) if name in tot and sum(tot.values()) > 0 else None
    except (OSError, ValueError, KeyError):
        return None


def self_launch(args):
    """`python bench.py --gpus N` without torchrun: start N fresh rank processes (decided from the environment BEFORE anything
    touches the GPU; this parent never does), one per GPU, rendezvous on 127.0.0.1; rank 0 prints the JSON line to the inherited
    stdout.  Exit code = the worst child's."""
    import socket
    import subprocess
    with socket.socket() as sk:
        sk.bind(('127.0.0.1', 0))
        port = sk.getsockname()[1]
    procs = []
    for rank in range(args.gpus):
        env = dict(os.environ, RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(args.gpus), MASTER_ADDR='127.0.0.1',
                   MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get('HSA_ENABLE_IPC_MODE_LEGACY', '0'))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env))
    rc = 0
    for p in procs:
        rc = max(rc, abs(p.wait()))
    return rc


def psnr_vs_oracle(r, poses_np, intr, nc, n_rays):
    """BASELINE.json's second metric: PSNR of the build's render against the reference render on the identical checkpoint and
    pose.  The reference cannot run here, so the checker is the fp32 CPU oracle pipeline (oracle/: march -> hash encode ->
    MLPs -> composite -> epilogue) on the model's CURRENT parameters, `n_rays` fixed pixels of frame 0, outside the timed
    region.  The product side is Renderer.render(training=False) -- the path that produces test renders."""
    from oracle import oracle as O
    m = r.model
    rng = np.random.default_rng(12345)
    pix = np.sort(rng.choice(intr.w * intr.h, n_rays, replace=False))
    dev = r.device
    with torch.no_grad():
        out = r.render(torch.tensor(poses_np[0], device=dev), None, training=False, pix_subset=torch.tensor(pix, device=dev))
    rgb = out['rgb_map'].float().cpu().numpy()
    sd = {k: v.detach().float().cpu().numpy() for k, v in m.state_dict().items()}
    ro, rd = O.generate_rays(poses_np[0], intr.w, intr.h, intr.fx, intr.fy, intr.cx, intr.cy, 3, pix_indices=pix)
    near, far = O.near_far_from_aabb(ro, rd, np.array([-2, -2, -2, 2, 2, 2], np.float32), r.cfg.min_near)
    xyzs, _, deltas, rays, cnt = O.march_rays_train(ro, rd, 2.0, r.march_bitfield.cpu().numpy(), r.cascade, r.cfg.grid_size, near, far,
                                                    r.cfg.max_steps, align=128)
    fp = O.FieldParams(sd['x_density_embedder.embeddings'], sd['x_color_embedder.embeddings'], sd['density_net.params'],
                       sd['color1_net.params'], sd['color2_net.params'], sd['class_net.params'],
                       m._offsets_np, m.per_level_scale, num_classes=nc)
    o, sig, _ = O.field_forward(fp, xyzs)
    ws, depth, image = O.composite_rays_train_forward(sig * np.float32(r.cfg.density_scale), o, deltas, rays, r.cfg.t_thresh)
    rgb_o, _, _ = O.render_epilogue(ws, depth, image, near, far)
    mse = float(np.mean((rgb - rgb_o) ** 2))
    return {'psnr_vs_oracle_db': round(float(O.compute_psnr(max(mse, 1e-14))), 2), 'psnr_rays': int(n_rays),
            'psnr_samples': int(cnt[0]), 'psnr_mean_opacity': round(float(ws.mean()), 4),
            'psnr_note': 'inference render of the final parameters (f16/bf16 MFMA, table dtype as benchmarked) vs the fp32 CPU '
                         'oracle pipeline on the same parameters, pose 0, fixed pixels; computed after the timed region'}


def main():
    args = parse()
    if args.gpus > 1 and 'RANK' not in os.environ:
        sys.exit(self_launch(args))
    from nerfstyle_amd import parallel as P
    # rehearsal knobs (NOT used by the driver): run N ranks on ONE GPU over gloo to exercise the N > 1
    # code path on a 1-GPU box -- NSR_BENCH_BACKEND=gloo NSR_BENCH_DEVICE=0
    backend = os.environ.get('NSR_BENCH_BACKEND')
    rank, local_rank, world = P.env_world()
    assert torch.cuda.is_available(), 'bench.py needs a HIP device (no CPU fallback for the product path)'
    dev_index = int(os.environ.get('NSR_BENCH_DEVICE', local_rank))
    torch.cuda.set_device(dev_index)
    rank, local_rank, world = P.init(backend, seed=args.seed)
    assert world == args.gpus, '--gpus {} but WORLD_SIZE={} (launch plainly, or with torchrun --nproc-per-node {})'.format(
        args.gpus, world, args.gpus)
    dev = torch.device('cuda', dev_index)
    if args.stage == 'style':
        result = run_style(args, dev, rank, world)
    else:
        result = run_recon(args, dev, rank, world)
    if rank == 0:
        import torch.distributed as dist
        result['rccl_ranks'] = dist.get_world_size() if dist.is_initialized() else 1
        result['dist_backend'] = dist.get_backend() if dist.is_initialized() else 'none (single process)'
        print('[bench] gpu leg done: ' + json.dumps(result), file=sys.stderr, flush=True)
        if world == 1 and not args.no_cpu_baseline:
            result['cpu_baseline'] = cpu_baseline(args.cpu_budget_s, args.num_classes)
        print(json.dumps(result), flush=True)
    P.barrier()


# ---------------------------------------------------------------------------------------------------------------------
# reconstruction stage (configs[1], [3], [4])
# ---------------------------------------------------------------------------------------------------------------------
def run_recon(args, dev, rank, world):
    from nerfstyle_amd import parallel as P
    from nerfstyle_amd import profiling
    from nerfstyle_amd.optim import FusedAdam, LossScaler, exp_lr
    model, r, rcfg, poses, intr = build(args, dev, rank)
    nc = args.num_classes
    # The reference's GradScaler (trainers/base.py:228,420-425: init scale 65536, inf/nan check + skip + back-off / growth every
    # step), device-side: optim.LossScaler.  Enabled when the MFMA chain is f16 (bf16 needs no scaling: GradScaler(enabled=False)).
    # The check (one streaming pass over the gradient arena) and the policy kernel run INSIDE the timed step.
    loss_scale = 65536.0 if args.compute_dtype == 'f16' else 1.0
    scaler = None if args.no_loss_scaler else LossScaler(init_scale=65536.0, enabled=(args.compute_dtype == 'f16'))
    scale_t = scaler.scale_tensor(dev) if scaler is not None else None
    loss_scale_t = torch.tensor(loss_scale, dtype=torch.float32, device=dev)
    opt = FusedAdam(model, lr=1e-2, betas=(0.9, 0.999), eps=1e-15, ema_decay=0.95)
    npix = intr.w * intr.h
    n_rays = min(args.rays_per_gpu, npix)
    gen = torch.Generator(device=dev)
    gen.manual_seed(args.seed * 1000003 + rank)            # rank-distinct pixel streams
    # synthetic targets resident in HBM: seeded uniform RGB + integer class ids per pixel
    tg = torch.Generator(device=dev)
    tg.manual_seed(args.seed)
    target_rgb = torch.rand(npix, 3, device=dev, generator=tg)
    target_cls = torch.randint(0, nc, (npix,), device=dev, generator=tg)
    # per-step emitted-sample counts (device side): ONE 4-byte device copy per step; totals and overflows are worked out after
    # the timed region (five tiny torch kernels per step for running sums were 2 % of a 4 096-ray step)
    count_log = torch.zeros(args.warmup + args.steps, dtype=torch.int32, device=dev)
    sp_lambda, sp_coeff, sp_n = args.sparsity_lambda, 0.05, 50000          # cfgs/training/default.yaml:17-19

    from nerfstyle_amd.recon_loss import recon_loss

    def loss_fn(out, pix):
        # trainers/base.py:251-304: MSE + 0.001 * cross-entropy on the class logits -- one fused kernel for value and gradient
        # (nsr_recon_loss: target gather, loss scale and 1 / world folded in)
        total = recon_loss(out['rgb_map'], out['classes'], target_rgb, target_cls, pix, ce_lambda=1e-3, factor=1.0 / world,
                           scale=scale_t if scale_t is not None else loss_scale_t, backward=os.environ.get('NSR_BENCH_LOSS_BACKWARD', '1') != '0')
        if sp_lambda > 0:
            # trainers/base.py:409-413: sigma of 50 000 uniform points of the bbox, WITH autograd; :285-291: the loss
            pts = torch.rand(sp_n, 3, device=dev, generator=gen) * 4.0 - 2.0
            sig = model(pts)
            sp = torch.mean(torch.abs(1 - torch.exp(-sp_coeff * sig))) * sp_lambda
            total = total + sp * ((scale_t if scale_t is not None else loss_scale_t) / world)
        return total

    graphed = None
    if args.graph:
        assert sp_lambda == 0, 'the sparsity term draws fresh points from a torch generator every step: not part of the captured step'
        from nerfstyle_amd.graph import GraphedRenderStep
        # one process: the optimiser step (device-side scaler, lr schedule, EMA decay) is part of the captured step
        in_graph_opt = world == 1 and scaler is not None
        # ... and the next step's ray generation + march run beside it (GraphedRenderStep(prefetch=True)); batches that fill the
        # chip with their march gain nothing from it
        graph_prefetch = in_graph_opt and n_rays <= 65536 and os.environ.get('NSR_BENCH_GRAPH_PREFETCH', '1') != '0'
        graphed = GraphedRenderStep(r, n_rays, loss_fn, optimizer=opt if in_graph_opt else None,
                                    scaler=scaler if in_graph_opt else None, lr_decay_steps=30000, prefetch=graph_prefetch)

    # Pixels of a step: n_rays distinct pixels, uniformly at random (np.random.choice(..., replace=False), nerf_lib.py:134).
    # Consecutive chunks of ONE random permutation of the frame are exactly such draws, so a permutation (a device sort: 22
    # kernels, 0.2 ms -- an eighth of a 4 096-ray step) is drawn only when the previous one is used up.
    perm_state = {'perm': None, 'pos': 0}

    tile_order = None
    if n_rays == npix and os.environ.get('NSR_BENCH_PIXEL_ORDER', 'tiles') != 'random':
        # a draw of EVERY pixel without replacement is the whole frame, in an order nothing depends on (the loss is a sum over
        # the rays): the step visits the frame in 8x8-pixel tiles (nerfstyle_amd.rays.tile_order)
        from nerfstyle_amd.rays import tile_order as _tile_order
        tile_order = _tile_order(intr.w, intr.h, 8, 8, device=dev)

    def draw_pixels():
        if tile_order is not None:
            return tile_order
        if perm_state['perm'] is None or perm_state['pos'] + n_rays > npix:
            perm_state['perm'] = torch.randperm(npix, device=dev, generator=gen)
            perm_state['pos'] = 0
        p0 = perm_state['pos']
        perm_state['pos'] = p0 + n_rays
        return perm_state['perm'][p0:p0 + n_rays]

    # N > 1: the parameter-independent front of step i+1 (ray generation, march, compaction, sample sort: Renderer.begin_train)
    # is issued while the gradient all-reduce of step i is in flight (parallel.sync_gradients_async)
    nxt, pending = {}, {}
    # N == 1, eager steps: the parameter-independent front of step i+1 runs on a side stream beside the backward of step i
    # (--side-march; off by default: the backward pair's HIP-event time -- the roofline figure of this line -- then includes the
    # march and sort kernels that run beside it, 23.5 instead of 21.8 ms, while the step drops from 34.9 to 33.5 ms)
    side_on = args.side_march or os.environ.get('NSR_BENCH_SIDE_MARCH', '0') == '1'
    side_stream = torch.cuda.Stream(device=dev) if (world == 1 and not args.graph and side_on) else None
    last_it = args.warmup + args.steps - 1
    stage = [None, None]

    def inputs(it):
        # (two entries are kept: the captured step with prefetch is told the NEXT step's pose and pixels, and recognises them
        # by identity when they come back as the current ones)
        if it not in nxt:
            for k in [k for k in nxt if k < it - 1]:
                del nxt[k]
            nxt[it] = (poses[(it * 7 + rank) % poses.shape[0]], draw_pixels())
        return nxt[it]

    def step(it):
        pose, pix = inputs(it)
        if graphed is not None:
            if graphed.prefetch:
                loss = graphed(pose, pix, *inputs(it + 1))
            else:
                loss = graphed(pose, pix)
            cnt = r._last_counter
        else:
            ev_next = None
            if side_stream is not None and it < last_it:
                inputs(it + 1)                 # (drawn now: whatever produces them is enqueued ahead of this step's kernels)
            ctx = pending.pop(it, None)
            if ctx is None:
                ctx = r.begin_train(pose, pix)
            out = r.finish_train(ctx)
            if side_stream is not None:
                # the side stream's march starts when this step's forward is done: beside the BACKWARD (the table scatter waits on
                # the atomic unit with idle issue slots; beside the forward the march's bitfield loads fight the hash gather for
                # the L1).  Everything the march depends on -- its inputs, the last occupancy update -- is older than this event
                ev_next = torch.cuda.Event()
                ev_next.record()
            cnt = r._last_counter           # this render's device-side sample count
            loss = loss_fn(out, pix)
            if loss.requires_grad:         # the sparsity term; the reconstruction loss has back-propagated itself
                loss.backward()
            if side_stream is not None and it < last_it and not r.occupancy_update_due():
                # the next step's ray generation + march + compaction + sample sort, on the side stream, beside this step's
                # backward (Renderer.begin_train_on)
                # two sets of sample buffers, rewritten in place in turn: set k is read by step i (forward + backward) while the
                # side stream fills the other one, and is refilled only after a later forward's event, i.e. after that backward
                slot = (it + 1) % 2
                stage[slot] = r.begin_train_on(side_stream, *inputs(it + 1), after=ev_next, into=stage[slot])
                pending[it + 1] = stage[slot]
        if world > 1:
            sync = P.sync_gradients_async(model, optimizer=opt)
            if graphed is None and it < last_it and not r.occupancy_update_due():
                pending[it + 1] = r.begin_train(*inputs(it + 1))
            sync.wait()
        if graphed is not None and graphed.optimizer is not None:
            pass                                               # stepped inside the replayed graph
        elif scaler is not None:
            opt.step(scaler=scaler, lr_decay_steps=30000)      # lr = 1e-2 * 0.1^(steps / 30000), on the device
        else:
            opt.param_groups[0]['lr'] = exp_lr(1e-2, it, 30000)
            opt.step(grad_scale=loss_scale)
        count_log[it].copy_(cnt[0])
        return loss.detach()           # not the graph: whatever its nodes still hold would stay allocated over the next step

    for it in range(args.warmup):
        step(it)
    occ_before = int(r._occ_state[0]) if getattr(r, '_occ_state', None) is not None else 0
    profiling.reset()
    profiling.enabled = not args.graph      # event records cannot be captured into a graph
    graph_ev = None
    if args.graph:
        graph_ev = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
    # the cyclic collector stays out of the timed region (a generation-2 pass over torch's module graph is a 30-60 ms host
    # pause; the step itself creates no reference cycles): collect now, switch it back on afterwards
    gc.collect()
    gc.disable()
    P.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    if graph_ev:
        graph_ev[0].record()
    _tr = []
    for it in range(args.steps):
        loss = step(args.warmup + it)
        if os.environ.get('NSR_BENCH_TRACE'):
            torch.cuda.synchronize()
            ms = torch.cuda.memory_stats(dev)
            _tr.append((time.perf_counter(), ms.get('num_device_alloc', 0), ms.get('num_device_free', 0), ms.get('reserved_bytes.all.current', 0) >> 20))
    if _tr:
        print('[trace] per-step ms:', [round((b[0] - a) * 1e3, 1) for a, b in zip([t0] + [x[0] for x in _tr[:-1]], _tr)], file=sys.stderr)
        print('[trace] device allocs / frees / reserved MiB:', [(x[1], x[2], x[3]) for x in _tr], file=sys.stderr)
        for name, evs in profiling._events.items():
            print('[trace] {:>14s} ms:'.format(name), [round(a.elapsed_time(b), 1) for a, b in evs], file=sys.stderr)
    if graph_ev:
        graph_ev[1].record()
    torch.cuda.synchronize()
    P.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    gc.enable()
    profiling.enabled = False
    elapsed = P.max_over_ranks(elapsed, dev)
    prof = profiling.summary()
    if rank != 0:
        return None

    timed_counts = count_log[args.warmup:].to(torch.int64)
    samples = int(timed_counts.sum().item())
    overflows = int((timed_counts >= r.sample_capacity(n_rays)).sum().item())
    spr = samples / max(args.steps * n_rays, 1)
    value = world * n_rays * args.steps / elapsed / 1e6
    occ_updates = (int(r._occ_state[0]) - occ_before) if getattr(r, '_occ_state', None) is not None else 0
    tb = 2 if args.table_dtype == 'f16' else 4
    # Algorithmic bytes per sample, SURVEY 8d: forward gather = 2 enc x 16 lvl x 8 corners x 2 feat x tb = 512 x tb;
    # backward = the gather-side read-modify-write of the same 256 entries = 2 x the forward bytes.
    bytes_per_sample = {'field_fwd': 512 * tb, 'field_bwd': 2 * 512 * tb}
    pm = profile_traffic(args)

    sorted_bwd = r._use_spatial_order(n_rays, False)
    kernel_names = {'field_fwd': ['k_field_fwd'],
                    'field_bwd': ['k_field_bwd', 'k_table_scatter'] if sorted_bwd else ['k_field_bwd']}

    def roof(name, launches, avg_ms):
        per_launch = bytes_per_sample[name] * samples / max(launches, 1)
        ach = per_launch / (avg_ms * 1e-3) / 1e9
        d = {'bound': 'hbm', 'kernel': ' + '.join(kernel_names[name]), 'achieved': round(ach, 1), 'peak': 8000.0, 'unit': 'GB/s',
             'frac': round(ach / 8000.0, 4), 'traffic': None, 'avg_launch_ms': round(avg_ms, 4),
             'algorithmic_bytes_per_sample': bytes_per_sample[name], 'samples_per_launch': int(samples / max(launches, 1))}
        ks = [pm['kernels'].get(kn) for kn in kernel_names[name]] if pm else []
        if ks and all(ks):
            d['traffic'] = int(sum(k['traffic_bytes_per_sample'] for k in ks) * samples / max(launches, 1))
            d['traffic_source'] = ('NOT measured in this run: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this command in '
                                   'profiles/{} (per-sample figure x this run\'s samples per launch)'.format(os.path.basename(PROFILE_JSON)))
            d['profile_commit'] = pm.get('commit')
            d['profile_samples_per_launch'] = pm.get('samples_per_launch')
        return d

    roofline, extra = None, {}
    if prof:
        dom = max(prof.items(), key=lambda kv: kv[1][1])[0]
        if dom in bytes_per_sample:
            roofline = roof(dom, prof[dom][0], prof[dom][2])
        if 'field_fwd' in prof:
            launches, tot_ms, avg_ms = prof['field_fwd']
            extra['hash_gather_fwd'] = roof('field_fwd', launches, avg_ms)
            flops = 2.0 * (12544 + 64 * nc) * samples / max(launches, 1)      # SURVEY 8d, no padding counted
            tf = flops / (avg_ms * 1e-3) / 1e12
            extra['mlp_mfma_fwd'] = {'bound': 'mfma', 'kernel': 'k_field_fwd', 'achieved': round(tf, 1), 'peak': 2500.0,
                                     'unit': 'TFLOP/s', 'frac': round(tf / 2500.0, 4),
                                     'note': 'MLP FLOPs of the fused kernel over its whole duration (gather-bound kernel)'}
        k = pm['kernels'].get('k_table_scatter' if sorted_bwd else 'k_field_bwd') if pm else None
        if 'field_bwd' in prof and k and k.get('atomic_requests_per_sample'):
            # what k_field_bwd actually queues on: memory-side float-atomic requests (TCC_EA0_ATOMIC per sample from the PMC
            # pass in profiles/) against the chip-wide rate tools/atomic_footprint_bench.hip measures
            launches, tot_ms, avg_ms = prof['field_bwd']
            share, share_src = 1.0, None
            if sorted_bwd:
                # the HIP events bracket the PAIR (one C call launches both kernels): the scatter kernel's share of it comes
                # from the committed rocprofv3 kernel summary of this command
                share = profile_kernel_share('k_table_scatter', ('k_table_scatter', 'k_field_bwd'))
                share_src = 'profiles/r03_bench_kernel_stats.csv' if share else None
                share = share or 1.0
            rate = k['atomic_requests_per_sample'] * samples / max(launches, 1) / (avg_ms * share * 1e-3) / 1e9
            extra['atomic_requests_bwd'] = {'bound': 'memory-side atomic unit', 'kernel': 'k_table_scatter' if sorted_bwd else 'k_field_bwd',
                                            'achieved': round(rate, 2),
                                            'peak': 21.06, 'unit': 'G requests/s', 'frac': round(rate / 21.06, 4),
                                            'requests_per_sample': k['atomic_requests_per_sample'],
                                            'kernel_ms': round(avg_ms * share, 3), 'kernel_share_of_pair': round(share, 4),
                                            'kernel_share_source': share_src,
                                            'source': 'requests/sample from profiles/ (not this run); peak = measured microbenchmark'}
    elif graph_ev:
        # whole-replay timing: the kernels cannot be event-timed inside a graph; the per-kernel split of this
        # configuration is in profiles/ (rocprofv3 --kernel-trace of the same command)
        ms = graph_ev[0].elapsed_time(graph_ev[1]) / args.steps
        per_step = (bytes_per_sample['field_fwd'] + bytes_per_sample['field_bwd']) * samples / args.steps
        ach = per_step / (ms * 1e-3) / 1e9
        roofline = {'bound': 'hbm', 'kernel': 'whole captured step (graph replay + fused Adam), k_field_fwd + k_field_bwd bytes',
                    'achieved': round(ach, 1), 'peak': 8000.0, 'unit': 'GB/s', 'frac': round(ach / 8000.0, 4), 'traffic': None,
                    'avg_launch_ms': round(ms, 4), 'algorithmic_bytes_per_sample': bytes_per_sample['field_fwd'] + bytes_per_sample['field_bwd'],
                    'samples_per_launch': int(samples / args.steps),
                    'note': 'small-batch steps are launch/latency-bound, not bandwidth-bound; per-kernel split in profiles/'}
    psnr = {}
    if args.psnr_rays > 0 and world == 1:
        psnr = psnr_vs_oracle(r, poses.cpu().numpy(), intr, nc, args.psnr_rays)
    wl = "LLFF '{}' reconstruction stage, {}x{} frames, {} rays/step/GPU, synthetic occupancy (28 seeded boxes), {:.1f} samples/ray".format(
        args.scene, intr.w, intr.h, n_rays, spr)
    if sp_lambda > 0:
        wl += ', --sparsity_lambda {} (50 000 sigma-only points per step, with gradient)'.format(sp_lambda)
    if args.graph:
        wl += ', render+loss+backward{} replayed as one hipGraph{}'.format(
            '+optimiser' if graphed.optimizer is not None else '',
            " (the next step's ray generation + march on a second branch beside the optimiser)" if graphed.prefetch else '')
    return {
        'metric': 'train Mrays/sec', 'value': round(value, 4), 'unit': 'Mrays/s', 'n_gpus': world, 'steps': args.steps,
        'warmup': args.warmup, 'ms_per_step': round(elapsed / args.steps * 1e3, 3), 'higher_is_better': True,
        'scaling': 'weak', 'vs_baseline': None, 'dtype': args.compute_dtype, 'data': 'synthetic',
        'config': {
            'workload': wl, 'rays_per_step_per_gpu': n_rays, 'samples_per_ray': round(spr, 2), 'max_steps': rcfg.max_steps,
            'num_classes': nc, 'table_dtype': args.table_dtype, 'mfma_dtype': args.compute_dtype,
            'params': int(model.arena.numel()), 'parallelism': 'rays sharded x{} + RCCL all-reduce'.format(world) + (
                " (async, overlapped with the next step's march + sample sort)" if world > 1 and graphed is None else ''),
            'next_step_front': ('ray generation + march + compaction + sample sort of step i+1 on a side stream beside the backward of '
                                'step i (Renderer.begin_train_on; never across an occupancy update)' if side_stream is not None else
                                ('second branch of the captured step, beside the optimiser' if (graphed is not None and graphed.prefetch)
                                 else 'in line')),
            'pixel_order': ('every pixel of the frame once per step, visited in 8x8 tiles (a full draw without replacement: order-free)'
                            if tile_order is not None else 'uniform draws without replacement (chunks of one device randperm of the frame)'),
            'occupancy_updates_in_timed_region': occ_updates,
            'occupancy': ('device-side update every {} steps inside the step ({}); the march reads the seeded '
                          'synthetic bitfield (random-init model has no scene)'.format(
                              rcfg.update_iter, 'steady state of the schedule, local_step >= update_thres: partial updates, 2 x {} sigma queries per cascade'.format(
                                  rcfg.grid_size ** 3 // 4) if args.occ_phase == 'steady' else 'first {} steps of the schedule: full updates, {} sigma queries'.format(
                                  rcfg.update_thres, r.cascade * rcfg.grid_size ** 3))
                          if not args.no_occ_update else 'fixed synthetic bitfield, no update'),
            'table_scatter': ('spatial order (nsr_sample_order + stand-alone lattice scatter kernel)'
                              if r._use_spatial_order(n_rays, False) else 'ray order (run tracker, fused)'),
            'sample_capacity_overflows': overflows,
            'final_loss': float(loss.detach()) / (scaler.get_scale() if scaler is not None else loss_scale) * world,
            'grad_scaler': ({'device_side': True, 'enabled': scaler.enabled, 'scale': scaler.get_scale(), 'steps_skipped': scaler.steps_skipped(),
                             'steps_taken': opt.steps_taken} if scaler is not None else 'constant scale, no inf/nan check'),
        },
        'kernel_ms_per_step': {k: round(v[1] / args.steps, 4) for k, v in sorted(prof.items())},
        'roofline': roofline,
        'rooflines_other': extra,
        **psnr,
    }


# ---------------------------------------------------------------------------------------------------------------------
# stylisation stage (configs[2])
# ---------------------------------------------------------------------------------------------------------------------
def run_style(args, dev, rank, world):
    """One step = one StyleTrainer.run_iter (trainers/style.py:162-204) on a 1008x756 frame: full-frame pass without autograd,
    VGG16 relu3 content + SemanticStyleLoss (PyTorch, MIOpen convolutions), d loss / d pixels, 24 patch re-renders of
    200x200 with autograd (sharded over ranks), colour-table-only fused Adam (style.py:25), lr 0.1 (cfgs/training/style.yaml)."""
    from nerfstyle_amd import parallel as P
    from nerfstyle_amd import profiling
    from nerfstyle_amd.losses import SemanticStyleLoss
    from nerfstyle_amd.optim import FusedAdam, LossScaler
    from nerfstyle_amd.stylize import StyleCriterion, deferred_backprop_step, patch_list, resident_backprop_step
    from nerfstyle_amd.vgg import VGG16FeatureExtractor
    if args.max_steps is None:
        args.max_steps = 512                                  # README stylisation command: --max_steps 512
    args.no_occ_update = True                                 # StyleTrainer loads a trained model; its run_iter never calls update_state
    model, r, rcfg, poses, intr = build(args, dev, rank)
    nc = args.num_classes
    W, H = intr.size()
    g = torch.Generator(device=dev)
    g.manual_seed(args.seed)
    targets = {}
    style = torch.rand(3, H, W, device=dev, generator=g)                       # SingleImage(longer edge = max(W, H)), style.py:60-61
    seg = torch.randint(0, nc, (H, W), device=dev, generator=g)
    fx = VGG16FeatureExtractor(['relu3']).to(dev)
    # the reference evaluates the loss under autocast (enable_amp: true, style.py:182-184)
    amp = None if args.fp32_loss else (torch.float16 if args.compute_dtype == 'f16' else torch.bfloat16)
    crit = StyleCriterion(fx, SemanticStyleLoss(['relu3'], clusters=seg), content_lambda=0.001, style_lambda=1.0, amp_dtype=amp)
    crit.init_style(style, num_classes=nc)
    opt = FusedAdam(model, lr=0.1, keywords=['x_color_embedder'])
    # trainers/style.py:200-204: scaler.step / scaler.update every iteration -- device-side here (optim.LossScaler)
    scaler = LossScaler(init_scale=65536.0, enabled=(args.compute_dtype == 'f16'))
    loss_scale = scaler.scale_tensor(dev)
    n_patches = len(patch_list(W, H, 200))
    patch_graphs = None if args.no_patch_graphs else {'streams': int(os.environ.get('NSR_PATCH_STREAMS', '4'))}                        # pass 2: one hipGraph per patch shape (graph.GraphedPatchBackward)
    total_samples = torch.zeros(1, dtype=torch.int64, device=dev)

    def step(it):
        frame = (it * 7) % poses.shape[0]                                       # every rank works on the SAME frame
        if frame not in targets:
            tg = torch.Generator(device=dev)
            tg.manual_seed(args.seed + frame)
            targets[frame] = torch.rand(3, H, W, device=dev, generator=tg)

        def image_loss(rgb, classes):
            return crit(rgb, targets[frame], classes, frame_key=frame, it=it)[0]
        if args.style_backprop == 'resident':
            loss, _ = resident_backprop_step(r, poses[frame], image_loss, loss_scale=loss_scale, rank=rank, world=world,
                                             optimizer=opt, with_classes=True)
        else:
            loss, _ = deferred_backprop_step(r, poses[frame], image_loss, patch_size=200, loss_scale=loss_scale, rank=rank, world=world,
                                             optimizer=opt, with_classes=True, patch_graphs=patch_graphs)
        opt.step(scaler=scaler)
        return loss

    for it in range(args.warmup):
        step(it)
    profiling.reset()
    profiling.enabled = True
    gc.collect()
    gc.disable()                            # as in run_recon: no generation-2 collection pause inside the timed region
    P.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for it in range(args.steps):
        loss = step(args.warmup + it)
    torch.cuda.synchronize()
    P.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    gc.enable()
    profiling.enabled = False
    elapsed = P.max_over_ranks(elapsed, dev)
    prof = profiling.summary()
    if rank != 0:
        return None
    # rays whose forward+backward completed: the frame's pixels once per iteration (pass 2 covers every pixel once with
    # autograd; the no-grad pass 1 is overhead of the method, not counted as extra rays)
    value = W * H * args.steps / elapsed / 1e6
    tb = 2 if args.table_dtype == 'f16' else 4
    roofline = None
    if 'field_bwd' in prof:
        launches, tot_ms, avg_ms = prof['field_bwd']
        roofline = {'bound': 'hbm', 'kernel': 'k_field_bwd (colour table only)', 'achieved': None, 'peak': 8000.0, 'unit': 'GB/s',
                    'frac': None, 'traffic': None, 'avg_launch_ms': round(avg_ms, 4), 'launches_per_step': launches / args.steps,
                    'algorithmic_bytes_per_sample': 512 * tb,
                    'note': 'per-launch sample counts stay on the device in this mode; see kernel_ms_per_step and profiles/'}
    return {
        'metric': 'train Mrays/sec', 'value': round(value, 4), 'unit': 'Mrays/s', 'n_gpus': world, 'steps': args.steps,
        'warmup': args.warmup, 'ms_per_step': round(elapsed / args.steps * 1e3, 3), 'higher_is_better': True,
        'scaling': 'strong', 'vs_baseline': None, 'dtype': args.compute_dtype, 'data': 'synthetic',
        'config': {
            'workload': (("LLFF '{}' stylisation stage, {}x{} frame per iteration: 1 full-frame no-grad pass + VGG16-relu3 content / "
                          "semantic-NNFM loss (PyTorch, {}) + {} {} deferred-backprop patches of 200x200, colour table only, max_steps {}; "
                          "random seeded VGG weights, style image and segment maps (none exist offline)").format(
                              args.scene, W, H, 'fp32' if args.fp32_loss else 'autocast ' + args.compute_dtype, n_patches,
                              'eager' if args.no_patch_graphs else 'graph-replayed ({} streams)'.format(patch_graphs.get('streams', 4)), rcfg.max_steps)
                         if args.style_backprop == 'deferred' else
                         ("LLFF '{}' stylisation stage, {}x{} frame per iteration: ONE render of the frame with autograd (activations "
                          "resident in HBM: the reference's no-grad pass + 24 deferred-backprop patch re-renders compute the same gradient "
                          "twice over) + VGG16-relu3 content / semantic-NNFM loss (PyTorch, {}) + backward through that render, colour table "
                          "only, max_steps {}; random seeded VGG weights, style image and segment maps (none exist offline)").format(
                              args.scene, W, H, 'fp32' if args.fp32_loss else 'autocast ' + args.compute_dtype, rcfg.max_steps)),
            'style_backprop': args.style_backprop,
            'rays_per_step': W * H, 'patches': n_patches, 'max_steps': rcfg.max_steps, 'num_classes': nc,
            'table_dtype': args.table_dtype, 'mfma_dtype': args.compute_dtype,
            'parallelism': ('bands of pixel rows sharded x{} (rendered with autograd and back-propagated by their rank; the image loss is '
                            'evaluated on the all-gathered frame), packed colour-table gradient all-reduce'.format(world)
                            if args.style_backprop == 'resident' else
                            'patches + pass-1 pixel rows sharded x{}, packed colour-table gradient all-reduce'.format(world)),
            'final_loss': float(loss) , 'matching': crit.style_loss.matching,
        },
        'kernel_ms_per_step': {k: round(v[1] / args.steps, 4) for k, v in sorted(prof.items())},
        'roofline': roofline,
        'rooflines_other': {},
    }


if __name__ == '__main__':
    main()
