"""TrajNet / TrajControl with the reference's constructor, state_dict and call contract, computed by
librohm_hip.so (mirror of RoHM's `model/trajnet.py`).

`TrajNet(...)(batch, time)` reads `batch['x_t']`, `batch['cond']` [B, T, 13] (and `batch['control_cond']`
[B, T, 272] with `trajcontrol=True`) and returns the x0 prediction [B, T, 13].  The modules below only hold
parameters under the reference's key names (186 keys, +84 `controlnet.*`); the convolutions, GroupNorm, Mish
and the time embedding run in hand-written gfx950 kernels.  No CPU path.
"""
from __future__ import annotations

import ctypes as C

import torch
import torch.nn as nn

from .. import _lib
from .._lib import TensorRef, TrajNetWeights, check, lib, ptr, stream_ptr
from .heads import Conv1dBlock, Downsample1d, ResidualTemporalBlock, Upsample1d, zero_module


class ControlNet(nn.Module):
    """Parameter container of the TrajControl branch (model/trajnet.py:10-41)."""

    def __init__(self, time_dim=32, control_cond_dim=272, traj_feat_dim=4, mid_dim=256):
        super().__init__()
        self.control_cond_dim, self.traj_feat_dim = control_cond_dim, traj_feat_dim
        m = mid_dim
        self.control_zero_conv_0 = zero_module(nn.Conv1d(control_cond_dim, traj_feat_dim, 1))
        widths = [m // 8, m // 4, m // 2, m]
        zero_out = [32, m // 8, m // 4, m // 2]
        c_in = traj_feat_dim
        for i, (w, z) in enumerate(zip(widths, zero_out), start=1):
            setattr(self, f'control_enc{i}', ResidualTemporalBlock(c_in, w, input_t=True, t_embed_dim=time_dim))
            setattr(self, f'control_zero_conv_{i}', zero_module(nn.Conv1d(w, z, 1)))
            setattr(self, f'control_downsample{i}', Downsample1d(2 * w))
            c_in = 2 * w
        self.control_mid_block1 = ResidualTemporalBlock(2 * m, m, input_t=True, t_embed_dim=time_dim)
        self.control_mid_block2 = ResidualTemporalBlock(m, m, input_t=True, t_embed_dim=time_dim)
        self.control_zero_conv_mid = zero_module(nn.Conv1d(m, m, 1))


def _res_keys(prefix, has_time, has_res):
    keys = []
    for b in (0, 1):
        for part in ('0', '2'):
            keys += [f'{prefix}.blocks.{b}.block.{part}.weight', f'{prefix}.blocks.{b}.block.{part}.bias']
    if has_time:
        keys += [f'{prefix}.time_mlp.1.weight', f'{prefix}.time_mlp.1.bias']
    if has_res:
        keys += [f'{prefix}.residual_conv.weight', f'{prefix}.residual_conv.bias']
    return keys


def weight_order(mid_dim, traj_feat_dim, trajcontrol):
    """Parameter names in the order `rohm_trajnet_create` consumes them (= the reference's state_dict order)."""
    m = mid_dim
    wb = lambda k: [k + '.weight', k + '.bias']
    keys = []
    if trajcontrol:
        c = 'controlnet.'
        keys += wb(c + 'control_zero_conv_0')
        c_in = traj_feat_dim
        for i, w in enumerate([m // 8, m // 4, m // 2, m], start=1):
            keys += _res_keys(c + f'control_enc{i}', True, c_in != w)
            keys += wb(c + f'control_zero_conv_{i}') + wb(c + f'control_downsample{i}.conv')
            c_in = 2 * w
        keys += _res_keys(c + 'control_mid_block1', True, True) + _res_keys(c + 'control_mid_block2', True, False)
        keys += wb(c + 'control_zero_conv_mid')
    keys += wb('time_mlp.1') + wb('time_mlp.3')
    c_in = traj_feat_dim
    for i, w in enumerate([m // 8, m // 4, m // 2, m], start=1):
        keys += _res_keys(f'diff_enc{i}', True, c_in != w) + wb(f'diff_downsample{i}.conv')
        c_in = 2 * w
    keys += _res_keys('diff_mid_block1', True, True) + _res_keys('diff_mid_block2', True, False)
    for i, w in zip((4, 3, 2, 1), (m, m // 2, m // 4, m // 8)):
        keys += wb(f'diff_upsample{i}.conv') + _res_keys(f'diff_dec{i}', True, True)
    keys += ['diff_final_conv.0.block.0.weight', 'diff_final_conv.0.block.0.bias',
             'diff_final_conv.0.block.2.weight', 'diff_final_conv.0.block.2.bias'] + wb('diff_final_conv.1')
    c_in = traj_feat_dim
    for i, w in enumerate([m // 8, m // 4, m // 2, m], start=1):
        keys += _res_keys(f'cond_enc{i}', False, c_in != w) + wb(f'cond_downsample{i}.conv')
        c_in = w
    return keys


class _NativeTrajNet:
    def __init__(self, module, device):
        sd = {k: v.detach() for k, v in module.state_dict().items()}
        order = weight_order(module.mid_dim, module.traj_feat_dim, module.trajcontrol)
        if set(order) != set(sd):
            raise _lib.RohmHipError(f'TrajNet state_dict mismatch: {sorted(set(order) ^ set(sd))[:6]}')
        keep = [sd[k].to(device=device, dtype=torch.float32).contiguous() for k in order]
        refs = (TensorRef * len(keep))()
        for r, t in zip(refs, keep):
            r.data, r.numel = t.data_ptr(), t.numel()
        w = TrajNetWeights(refs, len(keep))
        self.handle = C.c_void_p()
        self.device = device
        torch.cuda.synchronize(device)
        with torch.cuda.device(device):
            check(lib().rohm_trajnet_create(C.byref(self.handle), C.byref(w), module.mid_dim, module.time_dim,
                                            module.traj_feat_dim, module.control_cond_dim, int(module.trajcontrol),
                                            device.index or 0), 'rohm_trajnet_create')
        del keep
        self._ws = {}

    def workspace(self, B, T):
        """One workspace per (shape, HIP stream), see `_NativePoseNet.workspace`."""
        key = (B, T, torch.cuda.current_stream(self.device).cuda_stream)
        ws = self._ws.get(key)
        if ws is None:
            n = lib().rohm_trajnet_workspace_bytes(self.handle, B, T)
            if n == 0:
                raise _lib.RohmHipError(f'TrajNet: unsupported shape B={B}, T={T} (T must be a multiple of 16)')
            for k in [k for k in self._ws if k[2] == key[2]]:
                del self._ws[k]
            ws = torch.empty(n, dtype=torch.uint8, device=self.device)
            self._ws[key] = ws
        return ws

    def __del__(self):
        try:
            if self.handle:
                lib().rohm_trajnet_destroy(self.handle)
                self.handle = None
        except Exception:
            pass


class TrajNet(nn.Module):
    """Drop-in for `model.trajnet.TrajNet` (model/trajnet.py:80-174)."""

    def __init__(self, time_dim=32, cond_dim=4, mid_dim=256, traj_feat_dim=4, device=None, dataset=None,
                 repr_abs_only=False, trajcontrol=False, control_cond_dim=272,
                 weight_loss_root_rec_repr=0.0, weight_loss_root_pos_global=0.0, weight_loss_root_vel_global=0.0,
                 weight_loss_root_rot_vel_from_abs_traj=0.0, weight_loss_root_smplx_transl_vel=0.0,
                 weight_loss_root_smplx_rot_vel=0.0, weight_loss_root_smooth=0.0,
                 weight_loss_root_rot_cos_smooth_from_abs_traj=0.0):
        super().__init__()
        if cond_dim != traj_feat_dim:
            raise ValueError('the RoHM drivers always use cond_dim == traj_feat_dim; other shapes are unsupported')
        self.time_dim, self.mid_dim = time_dim, mid_dim
        self.traj_feat_dim, self.repr_abs_only = traj_feat_dim, repr_abs_only
        self.trajcontrol, self.control_cond_dim = trajcontrol, control_cond_dim
        self.dataset, self.device = dataset, device
        self.weight_loss_root_rec_repr = weight_loss_root_rec_repr
        self.weight_loss_root_pos_global = weight_loss_root_pos_global
        self.weight_loss_root_vel_global = weight_loss_root_vel_global
        self.weight_loss_root_rot_vel_from_abs_traj = weight_loss_root_rot_vel_from_abs_traj
        self.weight_loss_root_smplx_transl_vel = weight_loss_root_smplx_transl_vel
        self.weight_loss_root_smplx_rot_vel = weight_loss_root_smplx_rot_vel
        self.weight_loss_root_smooth = weight_loss_root_smooth
        self.weight_loss_root_rot_cos_smooth_from_abs_traj = weight_loss_root_rot_cos_smooth_from_abs_traj
        m, td = mid_dim, time_dim
        if trajcontrol:
            self.controlnet = ControlNet(time_dim=td, control_cond_dim=control_cond_dim, traj_feat_dim=traj_feat_dim,
                                         mid_dim=m)
        self.time_mlp = nn.ModuleList([nn.Identity(), nn.Linear(td, 4 * td), nn.Identity(), nn.Linear(4 * td, td)])
        widths = [m // 8, m // 4, m // 2, m]
        c_in = traj_feat_dim
        for i, w in enumerate(widths, start=1):
            setattr(self, f'diff_enc{i}', ResidualTemporalBlock(c_in, w, input_t=True, t_embed_dim=td))
            setattr(self, f'diff_downsample{i}', Downsample1d(2 * w))
            c_in = 2 * w
        self.diff_mid_block1 = ResidualTemporalBlock(2 * m, m, input_t=True, t_embed_dim=td)
        self.diff_mid_block2 = ResidualTemporalBlock(m, m, input_t=True, t_embed_dim=td)
        for i, w, out in ((4, m, m // 2), (3, m // 2, m // 4), (2, m // 4, m // 8), (1, m // 8, 32)):
            setattr(self, f'diff_upsample{i}', Upsample1d(w))
            setattr(self, f'diff_dec{i}', ResidualTemporalBlock(2 * w, out, input_t=True, t_embed_dim=td))
        self.diff_final_conv = nn.ModuleList([Conv1dBlock(32, 32, kernel_size=5), nn.Conv1d(32, traj_feat_dim, 1)])
        c_in = cond_dim
        for i, w in enumerate(widths, start=1):
            setattr(self, f'cond_enc{i}', ResidualTemporalBlock(c_in, w, input_t=False))
            setattr(self, f'cond_downsample{i}', Downsample1d(w))
            c_in = w
        self._native = None
        self._native_key = None

    def native(self, device=None):
        if device is None:
            device = next(self.parameters()).device
        device = torch.device(device)
        if device.type != 'cuda':
            raise _lib.RohmHipError('TrajNet runs only on an AMD GPU via librohm_hip.so (no CPU fallback)')
        if device.index is None:
            device = torch.device('cuda', torch.cuda.current_device())
        key = (str(device), hash(tuple((p.data_ptr(), p._version) for p in self.parameters())))
        if self._native is None or self._native_key != key:
            self._native = _NativeTrajNet(self, device)
            self._native_key = key
        return self._native

    def _inputs(self, batch):
        x_t, cond = batch['x_t'], batch['cond']
        _lib.require_hip(x_t, cond)
        B, T, Cc = x_t.shape
        if Cc != self.traj_feat_dim:
            raise ValueError(f'x_t must be [B, T, {self.traj_feat_dim}], got {tuple(x_t.shape)}')
        ctrl = self._check_cond(x_t, cond, batch)
        return x_t.detach().float().contiguous(), cond.detach().float().contiguous(), ctrl, B, T

    def _check_cond(self, x_t, cond, batch):
        """The C ABI takes raw pointers: a cond / control_cond of another shape would be read with the wrong stride
        (the reference raises inside its first conv instead), so shapes are checked here."""
        if tuple(cond.shape) != tuple(x_t.shape):
            raise ValueError(f'cond must have the shape of x_t {tuple(x_t.shape)}, got {tuple(cond.shape)}')
        if not self.trajcontrol:
            return None
        if batch is None or 'control_cond' not in batch:
            raise KeyError("TrajNet(trajcontrol=True) needs batch['control_cond']")
        ctrl = batch['control_cond']
        _lib.require_hip(ctrl)
        want = (x_t.shape[0], x_t.shape[1], self.control_cond_dim)
        if tuple(ctrl.shape) != want:
            raise ValueError(f'control_cond must be {want}, got {tuple(ctrl.shape)}')
        return ctrl.detach().float().contiguous()

    def forward(self, batch, time):
        """model/trajnet.py:177-275."""
        x, c, ctrl, B, T = self._inputs(batch)
        nat = self.native(x.device)
        t = time.to(torch.int64).contiguous()
        out = torch.empty_like(x)
        if B == 0:                      # an empty batch passes through like the reference's modules (nothing to launch)
            return out
        ws = nat.workspace(B, T)
        check(lib().rohm_trajnet_forward(nat.handle, ptr(x), ptr(c), ptr(ctrl), ptr(t), ptr(out), B, T, ptr(ws),
                                         ws.numel(), stream_ptr(x.device)), 'rohm_trajnet_forward')
        return out

    def sample_loop_native(self, x, cond, t_model, coef, noise, want_x0_last=False, batch=None, x_in_last=None):
        """`n` DDPM steps on the device (rohm_trajnet_sample_loop); x [B, T, 13] is updated in place.  `x_in_last` (optional,
        shaped like x) receives the input of the last step."""
        import numpy as np
        _lib.require_hip(x, cond, noise)
        nat = self.native(x.device)
        B, T, Cc = x.shape
        if Cc != self.traj_feat_dim:
            raise ValueError(f'x must be [B, T, {self.traj_feat_dim}], got {tuple(x.shape)}')
        ctrl = self._check_cond(x, cond, batch)
        n = len(t_model)
        if noise is not None and (noise.shape[0] < n or tuple(noise.shape[1:]) != tuple(x.shape)):
            raise ValueError(f'noise must be [>= {n}, {B}, {T}, {Cc}], got {tuple(noise.shape)}')
        if not (x.is_contiguous() and cond.is_contiguous() and (noise is None or noise.is_contiguous())):
            raise ValueError('x, cond and noise must be contiguous')
        t_arr = np.ascontiguousarray(t_model, dtype=np.int64)
        c_arr = np.ascontiguousarray(coef, dtype=np.float32).reshape(-1)
        x0_last = torch.empty_like(x) if want_x0_last else None
        if x_in_last is not None:        # the C side copies B * T * C floats into it
            _lib.require_hip(x_in_last)
            if not (tuple(x_in_last.shape) == tuple(x.shape) and x_in_last.dtype == torch.float32 and x_in_last.is_contiguous()
                    and x_in_last.device == x.device):
                raise ValueError(f'x_in_last must be a contiguous float32 tensor shaped like x {tuple(x.shape)} on {x.device}')
        if B == 0 or n == 0:
            return x0_last
        ws = nat.workspace(B, T)
        check(lib().rohm_trajnet_sample_loop(nat.handle, ptr(x), ptr(cond), ptr(ctrl),
                                             t_arr.ctypes.data_as(_lib.c_int64_p),
                                             c_arr.ctypes.data_as(_lib.c_float_p), ptr(noise), ptr(x0_last), ptr(x_in_last),
                                             n, B, T, ptr(ws), ws.numel(), stream_ptr(x.device)), 'rohm_trajnet_sample_loop')
        return x0_last

    def compute_losses_with_smpl(self, batch, model_output, smplx_model=None):
        """Evaluation loss report (model/trajnet.py:277-400), forward only."""
        from .eval_losses import trajnet_losses
        return trajnet_losses(self, batch, model_output, smplx_model)
