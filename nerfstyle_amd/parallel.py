"""Multi-GPU data parallelism over rays: one process per GPU, every rank
holds a full replica of the parameter arena, the occupancy bitfield and the optimiser state, the
ray batch is split across ranks, and ONE RCCL all-reduce (sum) of the flat gradient arena per
optimiser step keeps the replicas identical.  The reference has no distributed code at all
(device hard-coded to cuda:0, trainers/base.py:119); this is the only collective the path needs.

xGMI is point-to-point (7 links per GPU): a single large message lets RCCL drive all links, so
the gradient is reduced as one flat bucket (25.2 M fp32 = 100.9 MB; stylisation: the packed colour-table
gradient, 50.4 MB), not per
tensor.  Loss terms are divided by the world size before backward so the summed gradient equals
the single-GPU gradient of the global batch.
"""
import os

import torch
import torch.distributed as dist


def env_world():
    return int(os.environ.get('RANK', 0)), int(os.environ.get('LOCAL_RANK', 0)), int(os.environ.get('WORLD_SIZE', 1))


def init(backend=None, seed=None):
    """Initialises torch.distributed from the torchrun environment (no-op for one process).
    backend: 'nccl' (= RCCL on ROCm) when CUDA/HIP is available, else 'gloo'.
    seed: when given, seeds torch's default CPU and device generators IDENTICALLY on every rank, so that anything
    replicated that still draws from them (parameter init, torch.rand in user loss code) stays identical across
    ranks.  (The occupancy jitter does not depend on it: Renderer.manual_seed keys a counter-based generator.)"""
    rank, local_rank, world = env_world()
    if world > 1 and not dist.is_initialized():
        if backend is None:
            backend = 'nccl' if torch.cuda.is_available() else 'gloo'
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        os.environ.setdefault('MASTER_PORT', '29500')
        if backend == 'nccl' and 'NSR_BENCH_DEVICE' not in os.environ:
            torch.cuda.set_device(local_rank)
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    if seed is not None:
        torch.manual_seed(int(seed))          # seeds the CPU generator and every HIP device generator
    return rank, local_rank, world


def world_size():
    return dist.get_world_size() if dist.is_initialized() else 1


def shard_bounds(n, rank, world):
    """Contiguous, balanced [begin, end) split of n units (rays / pixel rows / patches)."""
    base, rem = divmod(n, world)
    begin = rank * base + min(rank, rem)
    return begin, begin + base + (1 if rank < rem else 0)


def rank_generator(seed, rank, device='cpu'):
    """Rank-distinct RNG stream for pixel sampling (rank-identical streams, e.g. for the occupancy
    jitter, simply use `seed` on every rank)."""
    g = torch.Generator(device=device)
    g.manual_seed(int(seed) * 1000003 + int(rank))
    return g


def all_reduce_sum_(flat: torch.Tensor, max_bucket_bytes=None):
    """In-place sum over ranks of one flat tensor; optionally in buckets of max_bucket_bytes so
    that an optimiser pass over an already-reduced bucket can overlap the next bucket."""
    if world_size() == 1:
        return flat
    if max_bucket_bytes is None or flat.numel() * flat.element_size() <= max_bucket_bytes:
        dist.all_reduce(flat, op=dist.ReduceOp.SUM)
        return flat
    per = max(1, max_bucket_bytes // flat.element_size())
    works = [dist.all_reduce(flat[i:i + per], op=dist.ReduceOp.SUM, async_op=True) for i in range(0, flat.numel(), per)]
    for w in works:
        w.wait()
    return flat


def broadcast_(flat: torch.Tensor, src=0):
    if world_size() > 1:
        dist.broadcast(flat, src=src)
    return flat


def sync_gradients(model, optimizer=None, only_color_table=None):
    """All-reduce (sum) of the gradient arena: ONE flat 100.9 MB message -- or, when `optimizer` (a FusedAdam) trains one
    hash table only and no MLP (the stylisation stage: `x_color_embedder` alone, trainers/style.py:25), that table's
    gradient alone: every second float2 of the interleaved rows, packed into a contiguous 50.4 MB buffer, reduced and
    unpacked -- half the bytes on the links for two extra streaming passes.  What is reduced is derived from what the
    optimiser trains (table_mask / nets), so every region it steps -- and every region its inf/nan check looks at -- is
    identical on all ranks; untrained regions stay rank-local (the optimiser ignores and zeroes them).
    only_color_table (legacy switch) forces the packed colour path when True."""
    g = model._ensure_grad()
    if world_size() == 1:
        return g
    half = None
    if only_color_table:
        half = 1
    elif only_color_table is None and optimizer is not None and not optimizer.nets and optimizer.table_mask in (0x3, 0xC):
        half = 0 if optimizer.table_mask == 0x3 else 1
    if half is None:
        return all_reduce_sum_(g)
    part = g[:model.table_elems].view(model.rows, 2, 2)[:, half, :]
    packed = part.contiguous()
    all_reduce_sum_(packed)
    part.copy_(packed)
    return g


class _GradSync:
    """Handle of an all-reduce in flight (sync_gradients_async): wait() makes the CURRENT stream wait for it (NCCL / RCCL:
    a stream dependency, no host block; gloo: a host wait) and, for the packed single-table form, unpacks the result."""

    def __init__(self, works, unpack=None):
        self.works, self.unpack = works, unpack

    def wait(self):
        for w in self.works:
            w.wait()
        if self.unpack is not None:
            dst, src = self.unpack
            dst.copy_(src)
        self.works, self.unpack = [], None


def sync_gradients_async(model, optimizer=None, buckets: int = 1) -> _GradSync:
    """sync_gradients without waiting: the all-reduce (RCCL's own stream) is in flight when this returns, and whatever the
    caller enqueues next on the compute stream overlaps it -- in the data-parallel step that is the parameter-independent
    front of the NEXT step (Renderer.begin_train: ray generation, occupancy march, compaction, sample sort: ~5 ms of a 37 ms
    full-frame step, ~0.25 ms of a 4 096-ray step), several times the ~1.2 ms a 100.9 MB ring all-reduce needs on one xGMI
    link.  `buckets` > 1 splits the message (the inf/nan check of bucket k can then run while bucket k+1 is on the links;
    the optimiser step itself has to wait for the last one: the GradScaler decision needs every bucket).
    The 15 360 MLP gradients are final a whole scatter kernel earlier than the table gradients, but they are 61 KB --
    a separate early message would cost a collective launch to hide ~1 us of transfer; they travel with the arena."""
    g = model._ensure_grad()
    if world_size() == 1:
        return _GradSync([])
    half = None
    if optimizer is not None and not optimizer.nets and optimizer.table_mask in (0x3, 0xC):
        half = 0 if optimizer.table_mask == 0x3 else 1
    if half is None:
        flat, unpack = g, None
    else:
        part = g[:model.table_elems].view(model.rows, 2, 2)[:, half, :]
        flat = part.contiguous().view(-1)
        unpack = (part, flat.view(model.rows, 2))
    n = flat.numel()
    per = (n + buckets - 1) // max(buckets, 1)
    works = [dist.all_reduce(flat[i:i + per], op=dist.ReduceOp.SUM, async_op=True) for i in range(0, n, per)]
    return _GradSync(works, unpack)


def barrier():
    if world_size() > 1:
        dist.barrier()


def max_over_ranks(value: float, device) -> float:
    if world_size() == 1:
        return value
    t = torch.tensor([value], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())
