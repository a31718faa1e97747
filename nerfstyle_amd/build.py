"""Builds nerfstyle_amd/libnsr_hip.so: every HIP source under csrc/ compiled for gfx950 with
hipcc and linked into ONE plain-C-ABI shared library (no torch, no pybind11).

`python -m nerfstyle_amd.build` or `__graft_entry__.build()`.  Cross-compiles without a GPU.
"""
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, 'csrc')
OBJ = os.path.join(HERE, 'csrc', '_obj')
LIB = os.path.join(HERE, 'libnsr_hip.so')
ARCH = 'gfx950'
SOURCES = ['raymarch.hip', 'composite.hip', 'occupancy.hip', 'sample_order.hip', 'gridenc.hip', 'field.hip', 'field_bwd.hip', 'field_bwd_gout.hip', 'table_scatter.hip', 'mlp.hip',
           'optim.hip']
# MFMA destinations in VGPRs (no AGPR round trip for results that VALU code consumes next): the forward, and the GOUT backward,
# whose 240 weight-gradient accumulators are pinned to AGPRs by inline assembly instead (field_bwd.hip)
EXTRA_FLAGS = {'field.hip': ['-mllvm', '--amdgpu-mfma-vgpr-form'],
               'field_bwd_gout.hip': ['-DNSR_BWD_ASM_WGRAD=1', '-mllvm', '--amdgpu-mfma-vgpr-form']}
HEADERS = ['nsr_common.h', 'rm_util.h', 'table_scatter.h', 'mfma_tiles.h', 'field_common.h', os.path.join('..', '..', 'include', 'nsr.h')]
FLAGS = ['-O3', '-fPIC', '-std=c++17', '--offload-arch=' + ARCH, '-Wall', '-Wno-unused-function']


def _hipcc():
    for c in (os.environ.get('HIPCC'), '/opt/rocm/bin/hipcc', 'hipcc'):
        if c and (os.path.isabs(c) and os.path.exists(c) or not os.path.isabs(c)):
            return c
    raise RuntimeError('hipcc not found')


def _stale(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.exists(d) and os.path.getmtime(d) > t for d in deps)


def build(force=False, verbose=False):
    os.makedirs(OBJ, exist_ok=True)
    hipcc = _hipcc()
    hdrs = [os.path.join(CSRC, h) for h in HEADERS] + [os.path.abspath(__file__)]
    srcs = [s for s in SOURCES if os.path.exists(os.path.join(CSRC, s))]
    jobs = []
    for s in srcs:
        src = os.path.join(CSRC, s)
        obj = os.path.join(OBJ, s.replace('.hip', '.o'))
        if force or _stale(obj, [src] + hdrs + ([os.path.join(CSRC, 'field_bwd.hip')] if s == 'field_bwd_gout.hip' else [])):
            jobs.append([hipcc] + FLAGS + EXTRA_FLAGS.get(s, []) + ['-c', src, '-o', obj])

    def run(cmd):
        if verbose:
            print(' '.join(cmd), flush=True)
        r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
        if r.returncode != 0:
            raise RuntimeError('hipcc failed:\n' + ' '.join(cmd) + '\n' + r.stdout)
        return r.stdout

    if jobs:
        with ThreadPoolExecutor(max_workers=min(4, len(jobs))) as ex:
            for out in ex.map(run, jobs):
                if verbose and out.strip():
                    print(out)
    objs = [os.path.join(OBJ, s.replace('.hip', '.o')) for s in srcs]
    if force or jobs or _stale(LIB, objs):
        run([hipcc, '-shared', '-fPIC', '--offload-arch=' + ARCH, '-o', LIB] + objs)
    return LIB


if __name__ == '__main__':
    print(build(force='--force' in sys.argv, verbose=True))
