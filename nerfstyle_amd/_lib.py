"""ctypes binding of libnsr_hip.so (the C ABI declared in include/nsr.h).

There is no CPU fallback: if the library is missing, or a kernel is asked to run on a tensor that
is not on a HIP device, the call raises.
"""
import ctypes
import os

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get('NSR_LIB_PATH') or os.path.join(_HERE, 'libnsr_hip.so')   # override: ablation builds (tools/)

NSR_F32, NSR_F16, NSR_BF16 = 0, 1, 2
NSR_ACT_NONE, NSR_ACT_SIGMOID = 0, 1
ABI_VERSION = 4

_DT = {torch.float32: NSR_F32, torch.float16: NSR_F16, torch.bfloat16: NSR_BF16}

vp = ctypes.c_void_p
u32 = ctypes.c_uint32
u64 = ctypes.c_uint64
i32 = ctypes.c_int
f32 = ctypes.c_float


class FieldDesc(ctypes.Structure):
    """struct nsr_field_desc (include/nsr.h)"""
    _fields_ = [
        ('L', u32), ('H', u32), ('S', f32), ('num_classes', u32),
        ('table_dtype', i32), ('compute_dtype', i32),
        ('bbox_min', f32 * 3), ('bbox_size', f32 * 3),
        ('density_scale', f32),
        ('offsets', ctypes.POINTER(ctypes.c_int32)),
    ]


# name -> (restype, argtypes); must list every symbol include/nsr.h declares
SIGNATURES = {
    'nsr_status_string': (ctypes.c_char_p, [i32]),
    'nsr_abi_version': (i32, []),
    'nsr_target_arch': (ctypes.c_char_p, []),
    'nsr_near_far_from_aabb': (i32, [vp, vp, vp, u32, f32, vp, vp, vp]),
    'nsr_morton3d': (i32, [vp, u32, vp, vp]),
    'nsr_morton3d_invert': (i32, [vp, u32, vp, vp]),
    'nsr_packbits': (i32, [vp, u32, f32, vp, vp]),
    'nsr_march_rays_train_workspace_bytes': (u64, [u32, f32, u32]),
    'nsr_march_rays_train': (i32, [vp, vp, vp, vp, f32, f32, u32, i32, u32, u32, u32, u32, vp, vp, vp, vp, vp, vp,
                                   vp, vp, vp, vp]),
    'nsr_composite_rays_train_forward': (i32, [vp, vp, vp, vp, u32, u32, u32, f32, i32, vp, vp, vp, vp]),
    'nsr_composite_rays_train_backward': (i32, [vp, vp, vp, vp, vp, vp, i32, vp, vp, u32, u32, u32, f32, vp, vp, vp]),
    'nsr_render_train_forward': (i32, [vp, vp, vp, vp, vp, vp, u32, u32, u32, f32, vp, vp, vp, vp, vp, vp, vp]),
    'nsr_render_train_backward': (i32, [vp, vp, vp, vp, vp, vp, vp, vp, vp, u32, u32, u32, f32, vp, vp, vp]),
    'nsr_recon_loss_workspace_bytes': (u64, [u32]),
    'nsr_recon_loss': (i32, [vp, vp, u32, u32, vp, vp, vp, f32, f32, vp, vp, vp, vp, vp, vp]),
    'nsr_march_rays': (i32, [u32, u32, vp, vp, vp, vp, vp, f32, f32, u32, i32, u32, u32, vp, vp, vp, vp, vp, vp, vp,
                             vp]),
    'nsr_composite_rays': (i32, [u32, u32, f32, vp, vp, vp, vp, vp, u32, i32, vp, vp, vp, vp]),
    'nsr_composite_rays_infer': (i32, [vp, vp, vp, vp, vp, u32, u32, u32, f32, vp, vp, vp, vp]),
    'nsr_compact_alive_workspace_bytes': (u64, [u32]),
    'nsr_compact_alive': (i32, [vp, u32, vp, vp, vp, vp]),
    'nsr_grid_resolutions': (i32, [u32, f32, u32, ctypes.POINTER(ctypes.c_uint32)]),
    'nsr_grid_encode_forward': (i32, [vp, vp, i32, ctypes.POINTER(ctypes.c_int32), vp, u32, u32, u32, u32, f32, u32,
                                      i32, u32, i32, u32, i32, vp]),
    'nsr_grid_encode_backward': (i32, [vp, i32, vp, ctypes.POINTER(ctypes.c_int32), vp, u32, u32, u32, u32, f32, u32,
                                       u32, i32, u32, i32, vp]),
    'nsr_mlp_param_count': (u32, [u32, u32, u32, u32]),
    'nsr_mlp_forward': (i32, [vp, vp, u32, u32, u32, u32, u32, i32, i32, vp, vp]),
    'nsr_mlp_backward': (i32, [vp, vp, vp, vp, u32, u32, u32, u32, u32, i32, i32, vp, vp, vp]),
    'nsr_field_forward': (i32, [ctypes.POINTER(FieldDesc), vp, vp, vp, u32, vp, vp, vp, vp, vp, vp]),
    'nsr_field_backward': (i32, [ctypes.POINTER(FieldDesc), vp, vp, vp, u32, vp, vp, vp, vp, vp, i32, i32, vp, vp, vp, vp]),
    'nsr_field_backward_workspace_bytes': (u64, [u32, i32]),
    'nsr_sample_order_workspace_bytes': (u64, [u32]),
    'nsr_sample_order': (i32, [vp, u32, vp, u32, ctypes.POINTER(f32), ctypes.POINTER(f32), vp, vp, vp]),
    'nsr_cast_f32_to_f16': (i32, [vp, vp, u64, vp]),
    'nsr_adam_step': (i32, [vp, vp, vp, vp, vp, vp, u64, f32, f32, f32, f32, f32, f32, u32, u32, vp]),
    'nsr_grad_check': (i32, [vp, u64, u32, vp, vp]),
    'nsr_scaler_update': (i32, [vp, f32, f32, f32, f32, f32, f32, u32, i32, f32, vp]),
    'nsr_adam_step_scaled': (i32, [vp, vp, vp, vp, vp, vp, u64, u64, f32, f32, f32, u32, vp, vp]),
    'nsr_occ_workspace_bytes': (u64, [u32, u32]),
    'nsr_occ_num_points': (u32, [u32, u32, i32]),
    'nsr_occ_sample_points': (i32, [vp, u32, u32, f32, i32, u64, u32, vp, vp, vp, vp, vp, vp]),
    'nsr_occ_update': (i32, [vp, vp, vp, u32, u32, u32, i32, f32, f32, vp, vp, vp, vp, vp]),
    'nsr_generate_rays': (i32, [vp, u32, u32, f32, f32, f32, f32, i32, vp, u32, vp, vp, vp]),
}

_lib = None


def lib():
    """Loads the shared library once; raises if it is missing (no fallback)."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                'libnsr_hip.so not found at {}: run `python -m nerfstyle_amd.build` '
                '(hipcc --offload-arch=gfx950).  There is no CPU fallback.'.format(LIB_PATH))
        L = ctypes.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(L, name)   # AttributeError if the library does not export it
            fn.restype = res
            fn.argtypes = args
        if L.nsr_abi_version() != ABI_VERSION:
            raise RuntimeError('libnsr_hip.so ABI {} != expected {}'.format(L.nsr_abi_version(), ABI_VERSION))
        _lib = L
    return _lib


def check(status, what=''):
    if status != 0:
        msg = lib().nsr_status_string(status).decode()
        raise RuntimeError('{} failed: {} (status {})'.format(what or 'nsr call', msg, status))


def p(t):
    """Device pointer of a tensor (None -> NULL).  The tensor must live on a HIP device."""
    if t is None:
        return None
    if not t.is_cuda:
        raise RuntimeError('nerfstyle_amd kernels need HIP device tensors; got a {} tensor (there is no CPU '
                           'fallback)'.format(t.device))
    if not t.is_contiguous():
        raise RuntimeError('nerfstyle_amd kernels need contiguous tensors')
    return t.data_ptr()


def stream():
    return torch.cuda.current_stream().cuda_stream


def dt(dtype):
    return _DT[dtype]
