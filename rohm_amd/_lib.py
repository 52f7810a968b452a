"""ctypes binding of librohm_hip.so (the C ABI in include/rohm_hip.h).

There is no CPU fallback: if the library is missing, or a tensor is not on a HIP device,
the call raises.  Build the library with `python -m rohm_amd.build`.
"""
from __future__ import annotations

import ctypes as C
import os

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get('ROHM_HIP_LIB') or os.path.join(_HERE, 'librohm_hip.so')

c_float_p = C.POINTER(C.c_float)
c_int64_p = C.POINTER(C.c_int64)


class RohmHipError(RuntimeError):
    pass


ROHM_ERR_EXCHANGE = -5      # include/rohm_hip.h


class LayerWeights(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in (
        'in_proj_w', 'in_proj_b', 'out_proj_w', 'out_proj_b', 'lin1_w', 'lin1_b', 'lin2_w', 'lin2_b',
        'norm1_w', 'norm1_b', 'norm2_w', 'norm2_b')]


class PoseNetWeights(C.Structure):
    _fields_ = [('in_x_w', C.c_void_p), ('in_x_b', C.c_void_p), ('in_c_w', C.c_void_p), ('in_c_b', C.c_void_p),
                ('pe', C.c_void_p), ('pe_len', C.c_int),
                ('t_w0', C.c_void_p), ('t_b0', C.c_void_p), ('t_w2', C.c_void_p), ('t_b2', C.c_void_p),
                ('out_w', C.c_void_p), ('out_b', C.c_void_p),
                ('layers', C.POINTER(LayerWeights))]


class TensorRef(C.Structure):
    _fields_ = [('data', C.c_void_p), ('numel', C.c_size_t)]


class TrajNetWeights(C.Structure):
    _fields_ = [('tensors', C.POINTER(TensorRef)), ('n_tensors', C.c_int)]


class ProfileRow(C.Structure):
    _fields_ = [('name', C.c_char * 48), ('launches', C.c_uint64), ('total_ms', C.c_double),
                ('flops', C.c_double), ('bytes', C.c_double)]


_lib = None

# name -> (restype, argtypes); must list every symbol declared in include/rohm_hip.h
SIGNATURES = {
    'rohm_last_error': (C.c_char_p, []),
    'rohm_version': (C.c_int, []),
    'rohm_profile_start': (C.c_int, [C.c_int]),
    'rohm_profile_stop': (C.c_int, [C.POINTER(ProfileRow), C.c_int, C.POINTER(C.c_int)]),
    'rohm_profile_detail': (C.c_int, [C.c_int]),
    'rohm_gemm_f32': (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_int,
                                C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p]),
    'rohm_exchange_probe': (C.c_int, [C.c_int, C.POINTER(C.c_char_p)]),
    'rohm_gemm_res_layernorm_scratch_bytes': (C.c_size_t, [C.c_int, C.c_int]),
    'rohm_gemm_res_layernorm_f32': (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_int,
                                              C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_float,
                                              C.c_void_p, C.c_size_t, C.c_void_p]),
    'rohm_layernorm_f32': (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p]),
    'rohm_attention_f32': (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p]),
    'rohm_planes_bytes': (C.c_size_t, [C.c_int, C.c_int, C.c_int]),
    'rohm_planes_split': (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_float, C.c_void_p, C.c_void_p]),
    'rohm_gemm_planes': (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_int,
                                   C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_float, C.c_float, C.c_int, C.c_int, C.c_int,
                                   C.c_void_p]),
    'rohm_gemm_planes_ln': (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_int,
                                      C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_float, C.c_float, C.c_int, C.c_int,
                                      C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_float,
                                      C.c_void_p]),
    'rohm_layernorm_planes_f32': (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p,
                                            C.c_void_p]),
    'rohm_attention_planes_f32': (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p]),
    'rohm_posenet_precision': (C.c_int, [C.c_void_p]),
    'rohm_ddpm_step': (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_float, C.c_float, C.c_float,
                                 C.c_float, C.c_void_p, C.c_size_t, C.c_void_p]),
    'rohm_ddpm_step_table': (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_float, C.c_void_p,
                                       C.c_float, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_int,
                                       C.c_size_t, C.c_void_p]),
    'rohm_posenet_create': (C.c_int, [C.POINTER(C.c_void_p), C.POINTER(PoseNetWeights), C.c_int, C.c_int, C.c_int,
                                      C.c_int, C.c_int, C.c_int, C.c_int, C.c_int]),
    'rohm_posenet_destroy': (None, [C.c_void_p]),
    'rohm_posenet_workspace_bytes': (C.c_size_t, [C.c_void_p, C.c_int, C.c_int]),
    'rohm_posenet_exchange_status': (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_size_t, C.c_void_p]),
    'rohm_posenet_status_offset': (C.c_size_t, [C.c_void_p, C.c_int, C.c_int]),
    'rohm_posenet_exchange_mode': (C.c_int, [C.c_void_p]),
    'rohm_posenet_exchange_guard': (C.c_char_p, [C.c_void_p]),
    'rohm_posenet_set_exchange': (C.c_int, [C.c_void_p, C.c_int]),
    'rohm_posenet_inject_exchange_fault': (C.c_int, [C.c_void_p, C.c_int]),
    'rohm_posenet_stack_timeline_bytes': (C.c_size_t, [C.c_int]),
    'rohm_posenet_set_stack_timeline': (C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]),
    'rohm_output_process_scratch_bytes': (C.c_size_t, []),
    'rohm_output_process_plan': (C.c_int, [C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int)]),
    'rohm_output_process_f32': (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int,
                                          C.c_int, C.c_int, C.c_void_p, C.c_size_t, C.c_void_p]),
    'rohm_posenet_forward': (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int,
                                       C.c_int, C.c_void_p, C.c_size_t, C.c_void_p]),
    'rohm_posenet_sample_loop': (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, c_int64_p, c_float_p, C.c_void_p,
                                           C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_size_t,
                                           C.c_void_p]),
    'rohm_trajnet_create': (C.c_int, [C.POINTER(C.c_void_p), C.POINTER(TrajNetWeights), C.c_int, C.c_int, C.c_int,
                                      C.c_int, C.c_int, C.c_int]),
    'rohm_trajnet_destroy': (None, [C.c_void_p]),
    'rohm_trajnet_tune': (C.c_int, [C.c_int, C.c_int, C.c_int]),
    'rohm_trajnet_loop_mode': (C.c_int, []),
    'rohm_trajnet_workspace_bytes': (C.c_size_t, [C.c_void_p, C.c_int, C.c_int]),
    'rohm_trajnet_forward': (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                       C.c_int, C.c_int, C.c_void_p, C.c_size_t, C.c_void_p]),
    'rohm_trajnet_sample_loop': (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, c_int64_p, c_float_p,
                                           C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p,
                                           C.c_size_t, C.c_void_p]),
    'rohm_smplx_create': (C.c_int, [C.POINTER(C.c_void_p), C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p,
                                    C.c_int, C.c_int, C.c_int]),
    'rohm_smplx_destroy': (None, [C.c_void_p]),
    'rohm_smplx_frames_to_world': (C.c_int, [C.c_void_p] * 6 + [C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]),
    'rohm_smplx_joints': (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p,
                                    C.c_int, C.c_void_p]),
    'rohm_guidance_workspace_bytes': (C.c_size_t, [C.c_int, C.c_int]),
    'rohm_guidance_skating_grad': (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int,
                                             C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]),
    'rohm_guidance_skating_prepare': (C.c_int, [C.c_void_p] * 4 + [C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_size_t,
                                                C.c_void_p]),
    'rohm_guidance_skating_apply': (C.c_int, [C.c_void_p] * 4 + [C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p,
                                              C.c_size_t, C.c_void_p]),
    'rohm_guidance_proj2d_grad': (C.c_int, [C.c_void_p] * 10 + [C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p,
                                            C.c_size_t, C.c_void_p]),
    'rohm_smplx_set_skinning': (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p]),
    'rohm_smplx_skinning_mode': (C.c_int, [C.c_void_p]),
    'rohm_smplx_lbs_workspace_bytes': (C.c_size_t, [C.c_void_p, C.c_int]),
    'rohm_smplx_forward': (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_int,
                                     C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]),
    'rohm_repr_joints': (C.c_int, [C.c_void_p, C.c_void_p, C.c_longlong, C.c_longlong, C.c_longlong, C.c_void_p,
                                   C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p]),
    'rohm_amass_metrics': (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_longlong, C.c_void_p, C.c_longlong, C.c_uint,
                                     C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p]),
    'rohm_traj_rederive': (C.c_int, [C.c_void_p, C.c_void_p, C.c_longlong, C.c_longlong, C.c_longlong] +
                           [C.c_void_p] * 4 + [C.c_int, C.c_int, C.c_void_p, C.c_longlong, C.c_longlong, C.c_longlong,
                                               C.c_void_p]),
}


def lib():
    """Load (once) and return the ctypes library; raises if it has not been built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RohmHipError(
                f'{LIB_PATH} not found: the HIP extension is required (no CPU fallback). '
                f'Build it with `python -m rohm_amd.build` (needs hipcc, --offload-arch=gfx950).')
        handle = C.CDLL(LIB_PATH)
        # the shipped library must export every declared symbol; an experiment library selected through ROHM_HIP_LIB (same-box A/B
        # runs against an older tree, scripts/build_variant.py) may lack the newest ones -- they then fail where they are used
        strict = not os.environ.get('ROHM_HIP_LIB')
        for name, (res, args) in SIGNATURES.items():
            if not strict and not hasattr(handle, name):
                continue
            fn = getattr(handle, name)
            fn.restype = res
            fn.argtypes = args
        _lib = handle
    return _lib


def check(rc, what=''):
    if rc != 0:
        msg = lib().rohm_last_error()
        raise RohmHipError(f'{what} failed (code {rc}): {msg.decode() if msg else "?"}')


def ptr(t):
    """Device pointer of a contiguous fp32/int64 HIP tensor (None -> NULL)."""
    if t is None:
        return None
    if not t.is_cuda:
        raise RohmHipError('rohm_amd kernels need tensors on a HIP device (got CPU tensor); '
                           'there is no CPU fallback')
    if not t.is_contiguous():
        raise RohmHipError('rohm_amd kernels need contiguous tensors')
    return C.c_void_p(t.data_ptr())


def stream_ptr(device=None):
    return C.c_void_p(torch.cuda.current_stream(device).cuda_stream)


def require_hip(*tensors):
    for t in tensors:
        if t is not None and not t.is_cuda:
            raise RohmHipError('this operator only runs on an AMD GPU through librohm_hip.so; '
                               'got a CPU tensor and there is deliberately no CPU fallback')


def profile_start(step_stride=1):
    check(lib().rohm_profile_start(step_stride), 'rohm_profile_start')


def profile_stop():
    """-> {label: dict(launches, total_ms, flops, bytes)} measured with HIP events on the launch stream."""
    rows = (ProfileRow * 256)()
    n = C.c_int(0)
    check(lib().rohm_profile_stop(rows, 256, C.byref(n)), 'rohm_profile_stop')
    return {rows[i].name.decode(): dict(launches=int(rows[i].launches), total_ms=rows[i].total_ms,
                                        flops=rows[i].flops, bytes=rows[i].bytes) for i in range(n.value)}
