"""PoseNet with the reference's constructor / state_dict / call contract, computed by librohm_hip.so.

Mirror of `model/posenet.py` in RoHM: same class name, constructor arguments, sub-module names
(so released checkpoints load with `strict=True`), same `forward(batch, timesteps)` semantics and
the same guidance hooks.  The arithmetic does NOT run in PyTorch: `forward` hands raw device
pointers to `rohm_posenet_forward` (hand-written gfx950 kernels) on the current HIP stream.
There is no CPU path: calling the module with CPU tensors raises.
"""
from __future__ import annotations

import ctypes as C
import warnings

import torch
import torch.nn as nn

from .. import _lib
from .._lib import LayerWeights, PoseNetWeights, check, lib, ptr, stream_ptr
from .heads import InputProcess, OutputProcess, PositionalEncoding, TimestepEmbedder


def _make_body_model(body_model_path, device):
    """`smplx.create(...)` as in model/posenet.py:57-58, or an injected nn.Module body model."""
    if isinstance(body_model_path, nn.Module):
        return body_model_path.to(device) if device is not None else body_model_path
    try:
        import smplx  # noqa: WPS433  (third-party; optional in this environment)
    except ImportError:
        from ..body_model import SMPLXLayer
        if isinstance(body_model_path, str) and body_model_path.endswith('.npz'):
            m = SMPLXLayer.from_npz(body_model_path)
            return m.to(device) if device is not None else m
        warnings.warn('smplx is not installed and no body model was injected: PoseNet.smplx_model is None; '
                      'test-time guidance will raise until a body model is attached')
        return None
    m = smplx.create(model_path=body_model_path, model_type='smplx', gender='neutral', flat_hand_mean=True,
                     use_pca=False)
    return m.to(device) if device is not None else m


class _NativePoseNet:
    """Owns a `rohm_posenet_t*` built from a snapshot of the module's parameters."""

    def __init__(self, module, device):
        sd = {k: v.detach() for k, v in module.state_dict().items() if not k.startswith('smplx_model.')}
        f = lambda k: sd[k].to(device=device, dtype=torch.float32).contiguous()
        keep = []

        def p(k):
            t = f(k)
            keep.append(t)
            return C.c_void_p(t.data_ptr())

        L = module.num_layers
        layers = (LayerWeights * L)()
        for i in range(L):
            pre = f'seqTransEncoder.layers.{i}.'
            lw = layers[i]
            lw.in_proj_w, lw.in_proj_b = p(pre + 'self_attn.in_proj_weight'), p(pre + 'self_attn.in_proj_bias')
            lw.out_proj_w, lw.out_proj_b = p(pre + 'self_attn.out_proj.weight'), p(pre + 'self_attn.out_proj.bias')
            lw.lin1_w, lw.lin1_b = p(pre + 'linear1.weight'), p(pre + 'linear1.bias')
            lw.lin2_w, lw.lin2_b = p(pre + 'linear2.weight'), p(pre + 'linear2.bias')
            lw.norm1_w, lw.norm1_b = p(pre + 'norm1.weight'), p(pre + 'norm1.bias')
            lw.norm2_w, lw.norm2_b = p(pre + 'norm2.weight'), p(pre + 'norm2.bias')
        w = PoseNetWeights()
        w.in_x_w, w.in_x_b = p('input_process.poseEmbedding.weight'), p('input_process.poseEmbedding.bias')
        w.in_c_w, w.in_c_b = p('input_process_cond.poseEmbedding.weight'), p('input_process_cond.poseEmbedding.bias')
        pe = sd['sequence_pos_encoder.pe'][:, 0].to(device=device, dtype=torch.float32).contiguous()
        keep.append(pe)
        w.pe, w.pe_len = C.c_void_p(pe.data_ptr()), pe.shape[0]
        w.t_w0, w.t_b0 = p('embed_timestep.time_embed.0.weight'), p('embed_timestep.time_embed.0.bias')
        w.t_w2, w.t_b2 = p('embed_timestep.time_embed.2.weight'), p('embed_timestep.time_embed.2.bias')
        w.out_w, w.out_b = p('output_process.poseFinal.weight'), p('output_process.poseFinal.bias')
        w.layers = layers
        self.handle = C.c_void_p()
        self.device = device
        torch.cuda.synchronize(device)
        with torch.cuda.device(device):
            check(lib().rohm_posenet_create(C.byref(self.handle), C.byref(w), module.latent_dim, module.num_heads,
                                            module.ff_size, L, module.input_feats, module.dataset_pose_feat_dim,
                                            module.input_feats - module.dataset_pose_feat_dim, device.index or 0),
                  'rohm_posenet_create')
        del keep
        self._ws = {}
        # 0: exact fp32 MFMA GEMMs (default); 3 / 2: created under ROHM_GEMM_PRECISION=bf16x6 / bf16x3 (labelled second line)
        self.gemm_planes = int(lib().rohm_posenet_precision(self.handle))

    def workspace(self, B, T):
        """Caller-owned workspace of the C ABI, one per (shape, HIP stream): forwards issued on different streams get
        distinct workspaces (the library's calls are re-entrant across streams only under that condition)."""
        key = (B, T, torch.cuda.current_stream(self.device).cuda_stream)
        ws = self._ws.get(key)
        if ws is None:
            nbytes = lib().rohm_posenet_workspace_bytes(self.handle, B, T)
            if nbytes == 0:
                raise _lib.RohmHipError('rohm_posenet_workspace_bytes returned 0 (bad shape)')
            for k in [k for k in self._ws if k[2] == key[2]]:      # one live shape per stream
                del self._ws[k]
            ws = torch.empty(nbytes, dtype=torch.uint8, device=self.device)
            self._ws[key] = ws
        return ws

    def poll_exchange(self):
        """rohm_posenet_exchange_status on the workspace(s) of the current stream: synchronises it; returns None, or the library's
        message if an in-kernel exchange failed in a forward / loop since the last check (the word is cleared by the read)."""
        stream = torch.cuda.current_stream(self.device).cuda_stream
        msg = None
        for (B, T, st), ws in list(self._ws.items()):
            if st == stream:
                rc = lib().rohm_posenet_exchange_status(self.handle, B, T, ptr(ws), ws.numel(), stream_ptr(self.device))
                if rc == _lib.ROHM_ERR_EXCHANGE:
                    m = lib().rohm_last_error()
                    msg = f'rohm_posenet_exchange_status failed (code {rc}): {m.decode() if m else "?"}'
                else:
                    check(rc, 'rohm_posenet_exchange_status')
        return msg

    def check_exchange(self):
        """poll_exchange that raises RohmHipError (ROHM_ERR_EXCHANGE)."""
        msg = self.poll_exchange()
        if msg is not None:
            raise _lib.RohmHipError(msg)

    @property
    def exchange_mode(self):
        """include/rohm_hip.h rohm_posenet_exchange_mode: bit 0 LayerNorm inside the GEMMs, bit 1 stream-K head, bit 2 refused by
        the layout guard at create, bit 3 switched off after a failed exchange."""
        return int(lib().rohm_posenet_exchange_mode(self.handle))

    @property
    def exchange_guard(self):
        m = lib().rohm_posenet_exchange_guard(self.handle)
        return m.decode() if m else ''

    def set_exchange(self, on):
        check(lib().rohm_posenet_set_exchange(self.handle, 1 if on else 0), 'rohm_posenet_set_exchange')

    def inject_exchange_fault(self, n_launches):
        check(lib().rohm_posenet_inject_exchange_fault(self.handle, int(n_launches)), 'rohm_posenet_inject_exchange_fault')

    def __del__(self):
        try:
            if self.handle:
                lib().rohm_posenet_destroy(self.handle)
                self.handle = None
        except Exception:
            pass


class PoseNet(nn.Module):
    """Drop-in for `model.posenet.PoseNet` (model/posenet.py:11-72)."""

    def __init__(self, dataset, body_feat_dim, nfeats=1,
                 latent_dim=256, ff_size=1024, num_layers=8, num_heads=4, dropout=0.1, activation="gelu",
                 body_model_path='', device=None, traj_feat_dim=4,
                 weight_loss_rec_repr_full_body=0.0, weight_loss_repr_foot_contact_mse=0.0,
                 weight_loss_joint_pos_global=0.0, weight_loss_joint_vel_global=0.0,
                 weight_loss_joint_smooth=0.0, weight_loss_foot_skating=0.0, start_skating_loss_epoch=0):
        super().__init__()
        if activation != 'gelu':
            raise ValueError('the HIP PoseNet implements the erf-form ("gelu") feed-forward only')
        self.dataset = dataset
        self.body_feat_dim, self.nfeats, self.traj_feat_dim = body_feat_dim, nfeats, traj_feat_dim
        self.foot_joint_index_list = [7, 10, 8, 11]   # l-ankle, l-toe, r-ankle, r-toe (posenet.py:30-31)
        self.foot_skating_vel_thres = 0.1
        self.fps = 30
        self.latent_dim, self.ff_size = latent_dim, ff_size
        self.num_layers, self.num_heads = num_layers, num_heads
        self.dropout, self.activation = dropout, activation
        self.input_feats = body_feat_dim * nfeats
        self.dataset_pose_feat_dim = dataset.pose_feat_dim
        self.device = device
        self.weight_loss_rec_repr_full_body = weight_loss_rec_repr_full_body
        self.weight_loss_repr_foot_contact_mse = weight_loss_repr_foot_contact_mse
        self.weight_loss_joint_pos_global = weight_loss_joint_pos_global
        self.weight_loss_joint_vel_global = weight_loss_joint_vel_global
        self.weight_loss_joint_smooth = weight_loss_joint_smooth
        self.weight_loss_foot_skating = weight_loss_foot_skating
        self.start_skating_loss_epoch = start_skating_loss_epoch

        body = _make_body_model(body_model_path, device)
        if body is not None:
            self.smplx_model = body
        else:
            self.smplx_model = None
        self.input_process = InputProcess(self.input_feats, latent_dim)
        self.input_process_cond = InputProcess(self.input_feats, latent_dim)
        self.sequence_pos_encoder = PositionalEncoding(latent_dim, dropout)
        with warnings.catch_warnings():
            warnings.simplefilter('ignore')
            layer = nn.TransformerEncoderLayer(d_model=latent_dim, nhead=num_heads, dim_feedforward=ff_size,
                                               dropout=dropout, activation=activation)
            self.seqTransEncoder = nn.TransformerEncoder(layer, num_layers=num_layers)
        self.embed_timestep = TimestepEmbedder(latent_dim, self.sequence_pos_encoder)
        self.output_process = OutputProcess(dataset.pose_feat_dim, latent_dim, nfeats)
        self._native = None
        self._native_key = None

    # ------------------------------------------------------------------ native handle cache
    def _apply(self, fn, *args, **kwargs):          # .to() / .cuda() / .float(): parameters may be re-created
        self._plist = None
        return super()._apply(fn, *args, **kwargs)

    def load_state_dict(self, *args, **kwargs):     # assign=True re-binds Parameter objects
        self._plist = None
        return super().load_state_dict(*args, **kwargs)

    def _fingerprint(self, device):
        """(device, storage address and version counter of every parameter).  The parameter list is collected once (the
        module-tree walk is the expensive part of a per-forward check on the guided, step-wise path); in-place updates
        (load_state_dict, optimiser steps) move the version counters, re-allocation moves the addresses."""
        pl = getattr(self, '_plist', None)
        # A re-bound Parameter object (`layer.weight = nn.Parameter(...)`, parametrizations) does not pass through _apply /
        # load_state_dict: every 64th call the cached list is checked against a fresh walk (object identity), so a stale list
        # lives for at most 64 forwards of the step-wise path instead of forever.
        self._pl_age = getattr(self, '_pl_age', 0) + 1
        if pl is not None and self._pl_age >= 64:
            self._pl_age = 0
            fresh = list(self.parameters(recurse=True))
            if len(fresh) != len(pl) or any(a is not b for a, b in zip(fresh, pl)):
                pl = None
        if pl is None:
            pl = self._plist = list(self.parameters(recurse=True))
        return (str(device), hash(tuple([(p.data_ptr(), p._version) for p in pl])))

    @property
    def gemm_precision(self):
        """'fp32' (default) or the opt-in 'bf16x6' / 'bf16x3' / 'fp16x3' the native handle was created under (ROHM_GEMM_PRECISION)."""
        n = self._native.gemm_planes if self._native is not None else 0
        return {0: 'fp32', 3: 'bf16x6', 2: 'bf16x3', 16: 'fp16x3'}[n]

    def native(self, device=None):
        """The `rohm_posenet_t` for the current weights on `device` (rebuilt if they changed)."""
        if device is None:
            device = next(self.parameters()).device
        device = torch.device(device)
        if device.type != 'cuda':
            raise _lib.RohmHipError('PoseNet runs only on an AMD GPU via librohm_hip.so (no CPU fallback); '
                                    'move the module and its inputs to a HIP device')
        if device.index is None:
            device = torch.device('cuda', torch.cuda.current_device())
        key = self._fingerprint(device)
        if self._native is None or self._native_key != key:
            self._native = _NativePoseNet(self, device)
            self._native_key = key
        return self._native

    def check_exchange(self):
        """Raise if a forward / sampling loop since the last check saw one of its in-kernel exchanges fail (the LayerNorm inside
        the out-projection / FF2 GEMMs, the stream-K output head: include/rohm_hip.h rohm_posenet_exchange_status).  Synchronises
        the current stream.  MANDATORY after direct `forward` calls whose output matters (the status is not polled per forward: that
        would be a host synchronisation on the hot path); the diffusion loops use `recover_exchange` instead."""
        if self._native is not None:
            msg = self._native.poll_exchange()
            if msg is not None:
                if self._native.exchange_mode & 3 == 0:
                    msg += ' -- and the handle was not using the exchanging launches'
                raise _lib.RohmHipError(msg)

    def uses_exchange(self):
        """True while the native handle launches kernels whose workgroups exchange data (rohm_posenet_exchange_mode bits 0 / 1)."""
        return self._native is not None and self._native.exchange_mode & 3 != 0

    def recover_exchange(self):
        """What the diffusion loops call after every fused chunk of steps and after every step-wise forward: False if every
        in-kernel exchange since the last check went through.  Otherwise the handle is switched -- for good -- to the exchange-free
        launches (GEMM + LayerNorm kernel pair, plain output-head tiles: rohm_posenet_set_exchange), a warning says so, and True
        tells the caller to RE-RUN what it computed since the last check (noise is injected or pre-drawn, so the re-run is exact).
        A failure while the exchange-free launches were already in use cannot come from them and raises."""
        nat = self._native
        if nat is None or nat.exchange_mode & 3 == 0:
            # nothing exchanges: no D2H copy, no stream synchronisation on the step-wise paths.  The once-per-run
            # `check_exchange` at the end of a sampling run still reads the word.
            return False
        msg = nat.poll_exchange()
        if msg is None:
            return False
        nat.set_exchange(False)
        warnings.warn('PoseNet: ' + msg + '.  The device is shared, partitioned or masked in a way the layout guard did not see at '
                      'create; this handle now runs the GEMM + LayerNorm kernel pair and plain output-head tiles (a few percent '
                      'slower), and the affected steps are re-run.')
        return True

    # ------------------------------------------------------------------ forward
    def forward(self, batch, timesteps):
        """batch['x_t'], batch['cond']: [B, body_feat_dim, 1, T]; timesteps int64 [B] -> [B, body_feat_dim, 1, T]
        (model/posenet.py:75-96)."""
        x_t, cond = batch['x_t'], batch['cond']
        _lib.require_hip(x_t, cond, timesteps)
        nat = self.native(x_t.device)
        B, Cc, nf, T = x_t.shape
        if Cc != self.input_feats or nf != 1:
            raise ValueError(f'x_t must be [B, {self.input_feats}, 1, T]; got {tuple(x_t.shape)}')
        if tuple(cond.shape) != tuple(x_t.shape):
            raise ValueError(f'cond must have the shape of x_t {tuple(x_t.shape)}, got {tuple(cond.shape)}')
        if tuple(timesteps.shape) != (B,):
            raise ValueError(f'timesteps must be [{B}], got {tuple(timesteps.shape)}')
        x_c = x_t.detach().to(torch.float32).contiguous()
        c_c = cond.detach().to(torch.float32).contiguous()
        t_c = timesteps.to(torch.int64).contiguous()
        out = torch.empty_like(x_c)
        if B == 0 or T == 0:            # an empty batch passes through like the reference's modules (nothing to launch)
            return out
        ws = nat.workspace(B, T)
        check(lib().rohm_posenet_forward(nat.handle, ptr(x_c), ptr(c_c), ptr(t_c), ptr(out), B, T, ptr(ws),
                                         ws.numel(), stream_ptr(x_c.device)), 'rohm_posenet_forward')
        return out

    # ------------------------------------------------------------------ fused sampling loop
    def sample_loop_native(self, x, cond, t_model, coef, noise, want_x0_last=False, batch=None, x_in_last=None):
        """Run `n = len(t_model)` DDPM steps on the device (rohm_posenet_sample_loop).

        x [B,C,1,T] is updated in place; noise [n,B,C,1,T]; coef float32 host array [n,3] of
        (coef1, coef2, sigma); t_model int64 host array [n].  Returns pred_xstart of the last step
        when `want_x0_last`; `x_in_last` (optional, shaped like x) receives the input of the last step."""
        import numpy as np
        _lib.require_hip(x, cond, noise)
        nat = self.native(x.device)
        B, Cc, _, T = x.shape
        if Cc != self.input_feats or tuple(cond.shape) != tuple(x.shape):
            raise ValueError(f'x and cond must both be [B, {self.input_feats}, 1, T]; got {tuple(x.shape)} / '
                             f'{tuple(cond.shape)}')
        if not (x.is_contiguous() and cond.is_contiguous()):
            raise ValueError('x and cond must be contiguous')
        n = len(t_model)
        t_arr = np.ascontiguousarray(t_model, dtype=np.int64)
        c_arr = np.ascontiguousarray(coef, dtype=np.float32).reshape(-1)
        assert c_arr.size == 3 * n
        if noise is not None and not (noise.is_contiguous() and noise.shape[0] >= n and
                                      tuple(noise.shape[1:]) == tuple(x.shape)):
            raise ValueError(f'noise must be contiguous [>= {n}, {B}, {Cc}, 1, {T}], got {tuple(noise.shape)}')
        x0_last = torch.empty_like(x) if want_x0_last else None
        if x_in_last is not None:        # the C side copies B * C * T floats into it
            _lib.require_hip(x_in_last)
            if not (tuple(x_in_last.shape) == tuple(x.shape) and x_in_last.dtype == torch.float32 and x_in_last.is_contiguous()
                    and x_in_last.device == x.device):
                raise ValueError(f'x_in_last must be a contiguous float32 tensor shaped like x {tuple(x.shape)} on {x.device}')
        if B == 0 or T == 0 or n == 0:
            return x0_last
        ws = nat.workspace(B, T)
        check(lib().rohm_posenet_sample_loop(nat.handle, ptr(x), ptr(cond),
                                             t_arr.ctypes.data_as(_lib.c_int64_p),
                                             c_arr.ctypes.data_as(_lib.c_float_p), ptr(noise), ptr(x0_last), ptr(x_in_last),
                                             n, B, T, ptr(ws), ws.numel(), stream_ptr(x.device)),
              'rohm_posenet_sample_loop')
        return x0_last

    # ------------------------------------------------------------------ guidance hooks (posenet.py:196-317)
    def guide_skating_with_smpl(self, batch, out, denoise_t, compute_grad='x_t'):
        from ..guidance import guide_skating
        return guide_skating(self, batch, out, denoise_t, compute_grad)

    def guide_2d_projection_with_smpl(self, batch, out, denoise_t, compute_grad='x_t'):
        from ..guidance import guide_2d_projection
        return guide_2d_projection(self, batch, out, denoise_t, compute_grad)

    def compute_losses_with_smpl(self, batch, model_output, smplx_model=None, epoch=0):
        """Evaluation loss report (model/posenet.py:98-194), forward only."""
        from .eval_losses import posenet_losses
        return posenet_losses(self, batch, model_output, smplx_model, epoch)
