"""The C-ABI library loads and exports every symbol include/nsr.h declares; the ctypes table in
nerfstyle_amd/_lib.py covers the same set; error paths that need no GPU behave."""
import ctypes
import os
import re

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    src = open(os.path.join(ROOT, 'include', 'nsr.h')).read()
    src = re.sub(r'/\*.*?\*/', '', src, flags=re.S)
    return sorted(set(re.findall(r'\b(nsr_[a-z0-9_]+)\s*\(', src)))


@pytest.fixture(scope='module')
def built():
    from nerfstyle_amd import build
    return build.build()


def test_library_exports_every_declared_symbol(built):
    lib = ctypes.CDLL(built)
    syms = declared_symbols()
    assert len(syms) >= 26
    for s in syms:
        assert hasattr(lib, s), 'libnsr_hip.so does not export ' + s


def test_ctypes_table_matches_header(built):
    from nerfstyle_amd import _lib
    assert sorted(_lib.SIGNATURES.keys()) == declared_symbols()
    L = _lib.lib()
    assert L.nsr_abi_version() == _lib.ABI_VERSION
    assert L.nsr_target_arch() == b'gfx950'
    assert b'invalid' in L.nsr_status_string(-1) and L.nsr_status_string(0) == b'ok'


def test_host_only_entry_points(built):
    """nsr_grid_resolutions / workspace queries / param counts run on the host."""
    from nerfstyle_amd import _lib
    from oracle import oracle as O
    L = _lib.lib()
    pls = O.per_level_scale_from_cfg()
    res = (ctypes.c_uint32 * 16)()
    assert L.nsr_grid_resolutions(16, O.grid_S(pls), 16, res) == 0
    assert list(res) == list(O.grid_resolutions(16, O.grid_S(pls), 16))
    assert L.nsr_mlp_param_count(32, 1, 64, 1) == 3072 and L.nsr_mlp_param_count(16, 3, 64, 2) == 6144
    assert L.nsr_mlp_param_count(32, 5, 64, 1) == 3072
    assert L.nsr_march_rays_train_workspace_bytes(4096, 2.0, 1024) >= 4096 * 4
    assert L.nsr_march_rays_train_workspace_bytes(100000, 2.0, 1024) >= 100000 * 4 * (1 + 2048 // 32)
    assert L.nsr_compact_alive_workspace_bytes(1000) > 0


def test_invalid_arguments_return_status_not_crash(built):
    from nerfstyle_amd import _lib
    L = _lib.lib()
    # null pointers -> NSR_ERR_INVALID_ARG before anything touches a device
    assert L.nsr_near_far_from_aabb(None, None, None, 8, 0.2, None, None, None) == -1
    assert L.nsr_packbits(None, 8, 0.5, None, None) == -1
    assert L.nsr_mlp_forward(None, None, 8, 32, 1, 64, 1, 0, 1, None, None) == -1
    # empty work is a no-op success
    assert L.nsr_morton3d(None, 0, None, None) == 0
    with pytest.raises(RuntimeError):
        _lib.check(-2, 'x')


def test_product_has_no_cpu_fallback(built):
    from nerfstyle_amd import raymarching
    with pytest.raises(RuntimeError, match='no CPU fallback'):
        raymarching.morton3D(torch.zeros(4, 3, dtype=torch.int32))


def test_product_never_imports_oracle():
    pkg = os.path.join(ROOT, 'nerfstyle_amd')
    for dp, _, fns in os.walk(pkg):
        for fn in fns:
            if fn.endswith(('.py', '.hip', '.h')):
                txt = open(os.path.join(dp, fn)).read()
                assert 'import oracle' not in txt and 'from oracle' not in txt and 'liboracle' not in txt, fn
