"""Experiment (GPU): nsr_sample_order (hand-written LSD radix sort) -- correctness against a stable argsort of the Morton keys
at several sizes, timing at the bench frame's size, and capture + 4 replays.
    python tools/exp_sort.py [M_big]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch


def say(*a):
    print(*a, file=sys.stderr, flush=True)


dev = torch.device('cuda:0')
from nerfstyle_amd.common import BBox
from nerfstyle_amd.config import NetworkConfig
from nerfstyle_amd.style_nerf import StyleTCNerf

model = StyleTCNerf(NetworkConfig(), BBox.from_radius(2.0), 5, enc_dtype=torch.float32, use_dir=False).to(dev)


def keys_of(xyzs):
    u = ((xyzs + 2.0) / 4.0 + 1.0) * 0.5
    q = torch.clamp(u * 1024.0, 0, 1023).to(torch.int64)

    def spread(v):
        v = v & 0x3FF
        v = (v | (v << 16)) & 0x030000FF
        v = (v | (v << 8)) & 0x0300F00F
        v = (v | (v << 4)) & 0x030C30C3
        v = (v | (v << 2)) & 0x09249249
        return v
    return spread(q[:, 0]) | (spread(q[:, 1]) << 1) | (spread(q[:, 2]) << 2)


def check(M, cnt, prefix, clustered=False):
    g = torch.Generator(device=dev)
    g.manual_seed(M)
    if clustered:
        # few distinct keys: long runs of one digit in every pass
        xyzs = (torch.randint(0, 7, (M, 3), device=dev, generator=g).float() * 0.5 - 1.7).contiguous()
    else:
        xyzs = (torch.rand(M, 3, device=dev, generator=g) * 3.6 - 1.8).contiguous()
    counter = torch.tensor([cnt, 0], dtype=torch.int32, device=dev)
    perm = model.sample_order(xyzs, counter, prefix).to(torch.int64)
    n = min(cnt, M if prefix is None else prefix)
    key = keys_of(xyzs[:n])
    want = torch.sort(key, stable=True)[1]
    ok = bool(torch.equal(perm[:n], want)) and bool(torch.equal(perm[n:], torch.arange(n, M, device=dev)))
    say('M', M, 'cnt', cnt, 'prefix', prefix, 'clustered', clustered, 'ok', ok)
    return ok


allok = True
for M, cnt, prefix in [(1, 1, None), (63, 63, None), (4096, 4096, None), (4097, 4097, None), (5000, 3000, None), (5000, 0, None),
                       (100000, 99999, 65000), (1 << 20, (1 << 20) - 17, None), (3000001, 2999999, None),
                       (4096 * 1024 + 5, 4096 * 1024 + 5, None), (9000000, 8500000, None)]:
    allok &= check(M, cnt, prefix)
allok &= check(300000, 299000, None, clustered=True)
allok &= check(5000000, 4999000, None, clustered=True)
say('ALL OK' if allok else 'FAILURES')

Mb = int(sys.argv[1]) if len(sys.argv) > 1 else 56000000
xyzs = (torch.rand(Mb, 3, device=dev) * 3.6 - 1.8).contiguous()
counter = torch.tensor([Mb - 12345, 0], dtype=torch.int32, device=dev)
out = torch.empty(Mb, dtype=torch.int32, device=dev)
for _ in range(2):
    model.sample_order(xyzs, counter, Mb, out=out)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(10):
    model.sample_order(xyzs, counter, Mb, out=out)
e1.record()
torch.cuda.synchronize()
say('sort of', Mb, 'pairs: %.3f ms per call' % (e0.elapsed_time(e1) / 10))
key = keys_of(xyzs[:Mb - 12345])
p = out[:Mb - 12345].to(torch.int64)
sk = key[p]
say('big sorted:', bool((sk[1:] >= sk[:-1]).all()), 'stable:', bool(((sk[1:] > sk[:-1]) | (p[1:] > p[:-1])).all()))
del key, p, sk

# capture + replays
M = 40000 * 64
xyzs = (torch.rand(M, 3, device=dev) * 3.6 - 1.8).contiguous()
cnt = torch.tensor([M - 1000, 0], dtype=torch.int32, device=dev)
p_eager = model.sample_order(xyzs, cnt, M).clone()
torch.cuda.synchronize()
s = torch.cuda.Stream()
s.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(s):
    model.sample_order(xyzs, cnt, M)
torch.cuda.current_stream().wait_stream(s)
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g):
    p_graph = model.sample_order(xyzs, cnt, M)
torch.cuda.synchronize()
for i in range(4):
    p_graph.zero_()
    g.replay()
    torch.cuda.synchronize()
    say('replay', i, 'equal to eager:', bool(torch.equal(p_graph, p_eager)))
# another count through the same graph
cnt.copy_(torch.tensor([M // 3, 0], dtype=torch.int32))
p2 = model.sample_order(xyzs, cnt, M).clone()
g.replay()
torch.cuda.synchronize()
say('replay with a new device count equal to eager:', bool(torch.equal(p_graph, p2)))
