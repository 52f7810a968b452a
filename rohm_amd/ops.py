"""Thin Python wrappers over the building-block entry points of librohm_hip.so.

These exist so each kernel can be parity-tested / profiled in isolation and so the diffusion
engine can issue the DDPM update.  Every function needs HIP tensors and raises otherwise.
"""
from __future__ import annotations

import ctypes as C

import torch

from . import _lib
from ._lib import check, lib, ptr, stream_ptr

EPI_BIAS, EPI_BIAS_GELU, EPI_BIAS_RES = 0, 1, 2


def gemm(a, w, bias=None, residual=None, epi=EPI_BIAS, out=None):
    """out[M,N] = epi(a[M,K] @ w[N,K]^T); K must be a multiple of 32."""
    _lib.require_hip(a, w)
    M, K = a.shape
    N = w.shape[0]
    if out is None:
        out = torch.empty(M, N, device=a.device, dtype=torch.float32)
    check(lib().rohm_gemm_f32(ptr(a), a.stride(0), ptr(w), w.stride(0), ptr(out), out.stride(0), M, N, K,
                              ptr(bias), ptr(residual), residual.stride(0) if residual is not None else 0, epi,
                              stream_ptr(a.device)), 'rohm_gemm_f32')
    return out


def gemm_res_layernorm(a, w, bias, residual, gamma, beta, eps=1e-5, out=None, return_scratch=False):
    """out[M,N] = LayerNorm(a @ w^T + bias + residual) * gamma + beta in one launch (include/rohm_hip.h
    rohm_gemm_res_layernorm_f32); raises RohmHipError(ROHM_ERR_UNSUPPORTED) for shapes without the in-kernel form."""
    _lib.require_hip(a, w, bias, residual, gamma, beta)
    M, K = a.shape
    N = w.shape[0]
    if out is None:
        out = torch.empty(M, N, device=a.device, dtype=torch.float32)
    nbytes = lib().rohm_gemm_res_layernorm_scratch_bytes(M, N)
    scratch = torch.empty(max(nbytes, 64) // 4 + 16, device=a.device, dtype=torch.int32)
    off = (-scratch.data_ptr() % 64) // 4
    scratch = scratch[off:]
    check(lib().rohm_gemm_res_layernorm_f32(ptr(a), a.stride(0), ptr(w), w.stride(0), ptr(out), out.stride(0), M, N, K,
                                            ptr(bias), ptr(residual), residual.stride(0), ptr(gamma), ptr(beta), eps,
                                            ptr(scratch), nbytes, stream_ptr(a.device)), 'rohm_gemm_res_layernorm_f32')
    return (out, scratch) if return_scratch else out


def output_process(h, w, b, B, T, ch_off=0, c_total=None, out=None, stream_k=True, return_scratch=False):
    """OutputProcess.forward (model/heads.py:171-176) on token-major h [B * (T + 1), D]: out[b, ch_off + c, 0, tok - 1] of a
    [B, c_total, 1, T] tensor (include/rohm_hip.h rohm_output_process_f32).  `stream_k=False` keeps plain tiling."""
    _lib.require_hip(h, w, b)
    D = h.shape[1]
    c_out = w.shape[0]
    if h.shape[0] != B * (T + 1) or w.shape[1] != D:
        raise ValueError(f'h must be [{B * (T + 1)}, D] and w [C_out, D]; got {tuple(h.shape)} / {tuple(w.shape)}')
    c_total = c_out + ch_off if c_total is None else c_total
    if out is None:
        out = torch.zeros(B, c_total, 1, T, device=h.device, dtype=torch.float32)
    scratch, nbytes = None, 0
    if stream_k:
        nbytes = lib().rohm_output_process_scratch_bytes()
        scratch = torch.empty(nbytes // 4 + 64, device=h.device, dtype=torch.int32)
        scratch = scratch[(-scratch.data_ptr() % 256) // 4:]
    check(lib().rohm_output_process_f32(ptr(h), ptr(w), ptr(b), ptr(out), B, T, D, c_out, ch_off, c_total, ptr(scratch), nbytes,
                                        stream_ptr(h.device)), 'rohm_output_process_f32')
    return (out, scratch) if return_scratch else out


def exchange_probe(device=None):
    """rohm_exchange_probe: (re-)run the layout probe of `device` -> (ok, reason).  A set-up call (it synchronises the device): run it
    before recording a graph that contains gemm_res_layernorm / output_process launches, or after a tenant has left the device."""
    import ctypes as C
    dev = torch.device('cuda', torch.cuda.current_device()) if device is None else torch.device(device)
    why = C.c_char_p()
    ok = lib().rohm_exchange_probe(dev.index or 0, C.byref(why))
    return bool(ok), (why.value.decode() if why.value else '')


def layernorm_(x, gamma, beta):
    _lib.require_hip(x)
    M, D = x.shape
    check(lib().rohm_layernorm_f32(ptr(x), ptr(gamma), ptr(beta), M, D, stream_ptr(x.device)), 'rohm_layernorm_f32')
    return x


def attention(qkv, n_seq, n_head, n_tok=144, head_dim=128):
    """qkv [n_seq*n_tok, 3*n_head*head_dim] (q pre-scaled) -> ctx [n_seq*n_tok, n_head*head_dim]."""
    _lib.require_hip(qkv)
    if qkv.shape != (n_seq * n_tok, 3 * n_head * head_dim):
        raise ValueError(f'qkv must be [{n_seq * n_tok}, {3 * n_head * head_dim}], got {tuple(qkv.shape)}')
    ctx = torch.empty(qkv.shape[0], n_head * head_dim, device=qkv.device, dtype=torch.float32)
    check(lib().rohm_attention_f32(ptr(qkv), ptr(ctx), n_seq, n_head, n_tok, head_dim, stream_ptr(qkv.device)),
          'rohm_attention_f32')
    return ctx


# ---- opt-in precision ladder: bf16 planes of fp32 matrices and the split-bf16 GEMM on them (rohm_hip.h) ----
EPI_QKV = 3


def planes_empty(rows, k, nplane, device):
    """Uninitialised plane tensor of an fp32 matrix [rows, k] (fragment-major layout of csrc/planes.h), as int16."""
    n = lib().rohm_planes_bytes(rows, k, nplane)
    return torch.empty(n // 2, dtype=torch.int16, device=device)


def planes_split(x, nplane=3, scale=1.0):
    """fp32 [rows, K] (rows % 16 == 0, K % 32 == 0) -> the planes of scale * x (nplane = mode: 3 / 2 bf16 planes, 16 = two fp16)."""
    _lib.require_hip(x)
    rows, k = x.shape
    out = planes_empty(rows, k, nplane, x.device)
    check(lib().rohm_planes_split(ptr(x), x.stride(0), rows, k, nplane, scale, ptr(out), stream_ptr(x.device)), 'rohm_planes_split')
    return out


def gemm_planes(a_planes, w_planes, m, n, k, nplane=3, bias=None, residual=None, epi=EPI_BIAS, qcols=0, qscale=1.0,
                out_f32=True, out_planes=False, flags=0, acc_scale=0.0):
    """epi(A @ W^T) from the planes of A [m, k] and W [n, k] -> (fp32 [m, n] or None, planes of it or None)."""
    _lib.require_hip(a_planes, w_planes)
    c = torch.empty(m, n, device=a_planes.device, dtype=torch.float32) if out_f32 else None
    cp = planes_empty(m, n, nplane, a_planes.device) if out_planes else None
    check(lib().rohm_gemm_planes(ptr(a_planes), ptr(w_planes), ptr(c), n, ptr(cp), m, n, k, ptr(bias), ptr(residual),
                                 residual.stride(0) if residual is not None else 0, qcols, qscale, acc_scale, epi, nplane, flags,
                                 stream_ptr(a_planes.device)), 'rohm_gemm_planes')
    return c, cp


def gemm_planes_ln(a_planes, w_planes, m, n, k, nplane, bias, residual=None, epi=EPI_BIAS_RES, qcols=0, qscale=1.0, acc_scale=0.0,
                   ln_stats=None, ln_c=None, r_stats=None, r_gamma=None, r_beta=None, want_stats=False, ln_dim=0, ln_eps=1e-5,
                   out_planes=False):
    """gemm_planes with LayerNorm folded in (include/rohm_hip.h rohm_gemm_planes_ln) -> (fp32 [m, n], planes or None, row
    statistics [m / 16, n / 16, 16, 2] or None)."""
    _lib.require_hip(a_planes, w_planes)
    c = torch.empty(m, n, device=a_planes.device, dtype=torch.float32)
    cp = planes_empty(m, n, nplane, a_planes.device) if out_planes else None
    st = torch.empty(m // 16, n // 16, 16, 2, device=a_planes.device, dtype=torch.float32) if want_stats else None
    check(lib().rohm_gemm_planes_ln(ptr(a_planes), ptr(w_planes), ptr(c), n, ptr(cp), m, n, k, ptr(bias), ptr(residual),
                                    residual.stride(0) if residual is not None else 0, qcols, qscale, acc_scale, epi, nplane,
                                    ptr(ln_stats), ptr(ln_c), ptr(r_stats), ptr(r_gamma), ptr(r_beta), ptr(st), ln_dim, ln_eps,
                                    stream_ptr(a_planes.device)), 'rohm_gemm_planes_ln')
    return c, cp, st


def layernorm_planes_(x, gamma, beta, nplane=3):
    """In-place LayerNorm that also returns the planes of its result."""
    _lib.require_hip(x)
    m, d = x.shape
    out = planes_empty(m, d, nplane, x.device)
    check(lib().rohm_layernorm_planes_f32(ptr(x), ptr(gamma), ptr(beta), m, d, nplane, ptr(out), stream_ptr(x.device)),
          'rohm_layernorm_planes_f32')
    return out


def attention_planes(qkv, n_seq, n_head, nplane=3):
    """attention() for n_tok = 144, head_dim = 128 returning the PLANES of ctx [n_seq * 144, n_head * 128]."""
    _lib.require_hip(qkv)
    if qkv.shape != (n_seq * 144, 3 * n_head * 128):
        raise ValueError(f'qkv must be [{n_seq * 144}, {3 * n_head * 128}], got {tuple(qkv.shape)}')
    out = planes_empty(n_seq * 144, n_head * 128, nplane, qkv.device)
    check(lib().rohm_attention_planes_f32(ptr(qkv), ptr(out), n_seq, n_head, nplane, stream_ptr(qkv.device)),
          'rohm_attention_planes_f32')
    return out


def ddpm_step(x_t, x0, noise, c1, c2, sigma, grad=None, grad_scale=0.0, out=None):
    _lib.require_hip(x_t, x0)
    if out is None:
        out = torch.empty_like(x_t)
    check(lib().rohm_ddpm_step(ptr(x_t), ptr(x0), ptr(noise), ptr(grad), c1, c2, sigma, grad_scale, ptr(out),
                               x_t.numel(), stream_ptr(x_t.device)), 'rohm_ddpm_step')
    return out


def ddpm_step_table(x_t, x0, noise, tables, t, grad_a=None, w_a=0.0, grad_b=None, w_b=0.0, out=None):
    """Per-sample-timestep DDPM update with device-resident schedule tables [n_steps, 4]."""
    _lib.require_hip(x_t, x0, tables, t)
    if out is None:
        out = torch.empty_like(x_t)
    B = x_t.shape[0]
    check(lib().rohm_ddpm_step_table(ptr(x_t), ptr(x0), ptr(noise), ptr(grad_a), w_a, ptr(grad_b), w_b,
                                     ptr(tables), ptr(t), tables.shape[0], ptr(out), B, x_t.numel() // B,
                                     stream_ptr(x_t.device)), 'rohm_ddpm_step_table')
    return out
