"""Per-kernel-family device timing with HIP events on the launch stream (torch.cuda.Event records
on torch's current stream, which is the stream every nsr_* call is given).  Off by default; bench.py
switches it on for the timed region to compute the roofline figure of the dominant kernel."""
import contextlib
from collections import defaultdict

import torch

enabled = False
_events = defaultdict(list)


@contextlib.contextmanager
def timed(name):
    if not enabled:
        yield
        return
    a = torch.cuda.Event(enable_timing=True)
    b = torch.cuda.Event(enable_timing=True)
    a.record()
    try:
        yield
    finally:
        b.record()
        _events[name].append((a, b))


def reset():
    _events.clear()


def summary():
    """-> {name: (launches, total_ms, avg_ms)}; call after torch.cuda.synchronize()."""
    out = {}
    for name, evs in _events.items():
        tot = sum(a.elapsed_time(b) for a, b in evs)
        out[name] = (len(evs), tot, tot / max(len(evs), 1))
    return out
