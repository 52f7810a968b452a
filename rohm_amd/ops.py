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
