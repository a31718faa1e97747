"""N > 1 path on CPU: world_size-2 gloo processes exercise the ray sharding and the flat
gradient all-reduce of nerfstyle_amd.parallel (the kernels themselves need a GPU)."""
import os
import socket
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

WORKER = r'''
import os, sys
sys.path.insert(0, os.environ["NSR_ROOT"])
import torch
from nerfstyle_amd import parallel as P
rank, local_rank, world = P.init(backend="gloo")
assert world == 2 and P.world_size() == 2
# ray sharding: disjoint, balanced, covering
n = 40001
b, e = P.shard_bounds(n, rank, world)
sizes = [P.shard_bounds(n, r, world) for r in range(world)]
assert sizes[0][0] == 0 and sizes[-1][1] == n and sizes[0][1] == sizes[1][0]
assert abs((sizes[0][1] - sizes[0][0]) - (sizes[1][1] - sizes[1][0])) <= 1
# rank-distinct pixel streams, rank-identical occupancy streams
g = P.rank_generator(69420, rank)
pix = torch.randperm(1000, generator=g)[:16]
gathered = [torch.zeros_like(pix) for _ in range(world)]
torch.distributed.all_gather(gathered, pix)
assert not torch.equal(gathered[0], gathered[1])
# gradient all-reduce: each rank's "gradient" of its ray shard, loss pre-divided by world
torch.manual_seed(0)
x = torch.randn(n, 8)
w = torch.randn(8, requires_grad=True)
full = ((x @ w) ** 2).mean()
gfull, = torch.autograd.grad(full, w)
local = ((x[b:e] @ w) ** 2).sum() / n          # global normalisation
glocal, = torch.autograd.grad(local, w)
flat = torch.zeros(100); flat[:8] = glocal
P.all_reduce_sum_(flat, max_bucket_bytes=64)    # bucketed path
assert torch.allclose(flat[:8], gfull, atol=1e-5), (flat[:8], gfull)
flat2 = torch.full((10,), float(rank + 1)); P.all_reduce_sum_(flat2)
assert torch.equal(flat2, torch.full((10,), 3.0))
t = torch.arange(5.) * (rank + 1); P.broadcast_(t, 0)
assert torch.equal(t, torch.arange(5.))
assert P.max_over_ranks(float(rank), "cpu") == 1.0
# colour-table-only reduction on the interleaved rows: [row][enc][feat], enc 1 = colour
class M:
    rows, table_elems = 6, 24
    def __init__(self): self.g = torch.arange(24 + 5, dtype=torch.float32) * (rank + 1)
    def _ensure_grad(self): return self.g
m = M(); before = m.g.clone()
P.sync_gradients(m, only_color_table=True)
t4 = m.g[:24].view(6, 2, 2); b4 = before[:24].view(6, 2, 2)
assert torch.equal(t4[:, 1], torch.arange(24.).view(6, 2, 2)[:, 1] * 3)      # summed over ranks 1x + 2x
assert torch.equal(t4[:, 0], b4[:, 0]) and torch.equal(m.g[24:], before[24:]) # everything else untouched
m2 = M(); P.sync_gradients(m2)
assert torch.equal(m2.g, torch.arange(29.) * 3)
# what is reduced follows from what the optimiser trains; the async form returns a handle, other work can be issued, wait() ends it
class Opt:
    def __init__(self, mask, nets): self.table_mask, self.nets = mask, nets
m3 = M(); before = m3.g.clone()
h = P.sync_gradients_async(m3, optimizer=Opt(0xC, []), buckets=2)      # colour table only: packed, two buckets
busy = torch.ones(1000).sum()                                             # (stands for the next step's march)
h.wait()
t3 = m3.g[:24].view(6, 2, 2)
assert torch.equal(t3[:, 1], torch.arange(24.).view(6, 2, 2)[:, 1] * 3) and torch.equal(t3[:, 0], before[:24].view(6, 2, 2)[:, 0])
assert torch.equal(m3.g[24:], before[24:])
m4 = M(); h = P.sync_gradients_async(m4, optimizer=Opt(0xF, [(0, 5)]), buckets=3); h.wait()
assert torch.equal(m4.g, torch.arange(29.) * 3)                          # tables + a net trained: the whole arena
m5 = M(); P.sync_gradients(m5, optimizer=Opt(0x3, []))                   # density table only
assert torch.equal(m5.g[:24].view(6, 2, 2)[:, 0], torch.arange(24.).view(6, 2, 2)[:, 0] * 3)
# identical default-generator streams after init(seed=...)
P.init(backend="gloo", seed=1234)
r = torch.rand(4); rs = [torch.zeros(4) for _ in range(world)]
torch.distributed.all_gather(rs, r)
assert torch.equal(rs[0], rs[1])
P.barrier()
print("RANK_OK", rank)
'''


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    p = s.getsockname()[1]
    s.close()
    return p


def test_world_size_2_gloo(tmp_path):
    script = tmp_path / 'worker.py'
    script.write_text(WORKER)
    port = _free_port()
    procs = []
    for rank in range(2):
        env = dict(os.environ, RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE='2', MASTER_ADDR='127.0.0.1',
                   MASTER_PORT=str(port), NSR_ROOT=ROOT, OMP_NUM_THREADS='2')
        procs.append(subprocess.Popen([sys.executable, str(script)], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT,
                                      text=True))
    outs = []
    for p in procs:
        try:
            out, _ = p.communicate(timeout=240)
        except subprocess.TimeoutExpired:
            p.kill()
            out, _ = p.communicate()
        outs.append(out)
    for rank, (p, out) in enumerate(zip(procs, outs)):
        assert p.returncode == 0, out
        assert 'RANK_OK {}'.format(rank) in out


def test_single_process_is_noop():
    import torch
    from nerfstyle_amd import parallel as P
    t = torch.ones(4)
    assert P.all_reduce_sum_(t) is t and P.world_size() == 1
    assert P.shard_bounds(10, 0, 1) == (0, 10)
    assert P.shard_bounds(10, 2, 3) == (7, 10) and P.shard_bounds(10, 0, 3) == (0, 4)
