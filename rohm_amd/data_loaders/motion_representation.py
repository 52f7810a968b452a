"""Device versions of the motion-representation helpers the reference drivers call between the diffusion
stages (data_loaders/motion_representation.py), bound to the HIP kernels in csrc/rederive.hip.

* `recover_from_repr_smpl(data_dict, recover_mode, smplx_model, ...)`  -- motion_representation.py:332-398
* `rederive_traj(...)`  -- the host round trip of test_amass_full.py:262-311 / test_prox_egobody.py:238-287
  (`recover_from_repr_smpl` + per-sequence `get_repr_smplx` :187-282 + re-normalisation) as one kernel.

GPU only: tensors must live on an AMD GPU (no CPU fallback; `rohm_amd._lib.RohmHipError` otherwise).
"""
from __future__ import annotations

import numpy as np
import torch

from .. import _lib
from .._lib import check, lib, ptr, stream_ptr
from ..body_model import native_for

REPR_LIST = ['root_rot_angle', 'root_rot_angle_vel', 'root_l_pos', 'root_l_vel', 'root_height',
             'smplx_rot_6d', 'smplx_rot_vel', 'smplx_trans', 'smplx_trans_vel',
             'local_positions', 'local_vel', 'smplx_body_pose_6d', 'smplx_betas', 'foot_contact']
REPR_DIM_DICT = {'root_rot_angle': 1, 'root_rot_angle_vel': 1, 'root_l_pos': 2, 'root_l_vel': 2, 'root_height': 1,
                 'smplx_rot_6d': 6, 'smplx_rot_vel': 3, 'smplx_trans': 3, 'smplx_trans_vel': 3,
                 'local_positions': 66, 'local_vel': 66, 'smplx_body_pose_6d': 126, 'smplx_betas': 10,
                 'foot_contact': 4}          # utils/other_utils.py:17-37
_MODES = {'smplx_params': 0, 'joint_abs_traj': 1, 'joint_rel_traj': 2}


def _stats(ds, device):
    """(Mean, Std) of a dataset object (or a (mean, std) pair) as float32 device tensors, cached on the object."""
    if isinstance(ds, (tuple, list)):
        mean, std = ds
        holder = None
    else:
        mean, std, holder = ds.Mean, ds.Std, ds
    key = '_rohm_stats_' + str(device)
    if holder is not None and key in getattr(holder, '__dict__', {}):
        return holder.__dict__[key]
    m = torch.as_tensor(np.asarray(mean, dtype=np.float32)).to(device).contiguous()
    s = torch.as_tensor(np.asarray(std, dtype=np.float32)).to(device).contiguous()
    if m.numel() != 294 or s.numel() != 294:
        raise ValueError('Mean / Std must have 294 entries')
    if holder is not None and hasattr(holder, '__dict__'):
        holder.__dict__[key] = (m, s)
    return m, s


def _strides(x, layout):
    """(B, T, stride_b, stride_t, stride_c) in floats for a [B,T,294] ('btc') or [B,294,1,T] ('bc1t') tensor."""
    if layout == 'btc':
        if x.dim() != 3 or x.shape[2] != 294:
            raise ValueError(f'expected [B, T, 294], got {tuple(x.shape)}')
        return x.shape[0], x.shape[1], x.stride(0), x.stride(1), x.stride(2)
    if layout == 'bc1t':
        if x.dim() != 4 or x.shape[1] != 294 or x.shape[2] != 1:
            raise ValueError(f'expected [B, 294, 1, T], got {tuple(x.shape)}')
        return x.shape[0], x.shape[3], x.stride(0), x.stride(3), x.stride(1)
    raise ValueError(f'unknown layout {layout!r}')


def joints_from_repr(repr_full, recover_mode='smplx_params', smplx_model=None, stats=None, layout='btc'):
    """[B,T,22,3] joints from the full 294-channel representation (any strides, no copy).  `stats` = dataset or
    (mean, std) to de-normalise on the fly, None if `repr_full` is already de-normalised."""
    if recover_mode not in _MODES:
        raise ValueError(f'recover_mode {recover_mode!r} is not supported (joint_abs_traj | joint_rel_traj | smplx_params)')
    _lib.require_hip(repr_full)
    x = repr_full.detach()
    if x.dtype != torch.float32:
        x = x.float()
    B, T, sb, st, sc = _strides(x, layout)
    dev = x.device
    handle = None
    if recover_mode == 'smplx_params':
        if smplx_model is None:
            raise ValueError("recover_mode='smplx_params' needs smplx_model")
        handle = native_for(smplx_model, dev).handle
    mean = std = None
    if stats is not None:
        mean, std = _stats(stats, dev)
    out = torch.empty(B, T, 22, 3, device=dev, dtype=torch.float32)
    check(lib().rohm_repr_joints(handle, x.data_ptr(), sb, st, sc, ptr(mean) if mean is not None else None,
                                 ptr(std) if std is not None else None, B, T, _MODES[recover_mode], ptr(out),
                                 stream_ptr(dev)), 'rohm_repr_joints')
    return out


def recover_from_repr_smpl(data_dict, recover_mode='joint_abs_traj', smplx_model=None, return_verts=False,
                           return_full_joints=False):
    """Drop-in for motion_representation.py:332-398 with the reference's dict-of-slices argument
    ([bs, T, dim] tensors, de-normalised).  `return_verts` / `return_full_joints` run the full-LBS path
    (rohm_smplx_forward; the 72 landmark joints beyond the 55 kinematic ones stay zero)."""
    if recover_mode not in _MODES:
        print('[ERROR] recover_mode incorrect! in func recover_from_repr_smpl()')   # as the reference (:347-348)
        raise ValueError(recover_mode)
    if (return_verts or return_full_joints) and recover_mode == 'smplx_params':      # :389-396, full LBS
        from ..body_model import lbs_forward
        r6, b6 = data_dict['smplx_rot_6d'], data_dict['smplx_body_pose_6d']
        _lib.require_hip(r6)
        bs = len(r6)
        pose = torch.cat([r6.reshape(-1, 1, 6), b6.reshape(-1, 21, 6)], dim=1).float().contiguous()
        nat = native_for(smplx_model, r6.device)
        jall, verts = lbs_forward(nat, pose, 1, data_dict['smplx_betas'].reshape(-1, 10).float().contiguous(),
                                  data_dict['smplx_trans'].reshape(-1, 3).float().contiguous(), want_verts=return_verts)
        if return_full_joints:
            joints = torch.zeros(pose.shape[0], 127, 3, device=pose.device)
            joints[:, :jall.shape[1]] = jall
            joints = joints.reshape(bs, -1, 127, 3)
        else:
            joints = jall[:, 0:22].reshape(bs, -1, 22, 3)
        return (joints, verts.reshape(bs, -1, verts.shape[1], 3)) if return_verts else joints
    need = {'joint_abs_traj': ['root_rot_angle', 'root_l_pos', 'root_height', 'local_positions'],
            'joint_rel_traj': ['root_rot_angle_vel', 'root_l_vel', 'root_height', 'local_positions'],
            'smplx_params': ['smplx_rot_6d', 'smplx_trans', 'smplx_body_pose_6d', 'smplx_betas']}[recover_mode]
    ref = data_dict[need[0]]
    lead = ref.shape[:-1]
    full = torch.zeros(lead + (294,), device=ref.device, dtype=torch.float32)
    o = 0
    for name in REPR_LIST:
        if name in need:
            full[..., o:o + REPR_DIM_DICT[name]] = data_dict[name]
        o += REPR_DIM_DICT[name]
    full3 = full.reshape(-1, lead[-1], 294) if len(lead) >= 2 else full.reshape(1, -1, 294)
    j = joints_from_repr(full3, recover_mode, smplx_model)
    return j.reshape(lead + (22, 3))


def rederive_traj(motion_repr, traj_stats, pose_stats, smplx_model, out=None, layout='btc', out_layout='btc'):
    """Trajectory channels re-derived from TrajNet's output (test_amass_full.py:262-311).

    motion_repr: the full representation with the denoised trajectory written in (`motion_repr_clean_root_rec`,
      [B,T,294] or [B,294,1,T]), normalised with `traj_stats` (dataset object with .Mean/.Std or (mean, std)).
    Returns `traj_rec_full` [B,T-1,22] normalised with `pose_stats`; or, with `out` = PoseNet's cond tensor
    ([B,T-1,294] 'btc' or [B,294,1,T-1] 'bc1t'), writes channels 0..21 of it in place (the assignment of
    test_amass_full.py:336) and returns `out`."""
    _lib.require_hip(motion_repr)
    x = motion_repr.detach()
    if x.dtype != torch.float32:
        x = x.float()
    B, T, sb, st, sc = _strides(x, layout)
    dev = x.device
    nat = native_for(smplx_model, dev)
    m_in, s_in = _stats(traj_stats, dev)
    m_out, s_out = _stats(pose_stats, dev)
    if out is None:
        res = torch.empty(B, T - 1, 22, device=dev, dtype=torch.float32)
        osb, ost, osc = res.stride()
        target = res
    else:
        _lib.require_hip(out)
        if out.dtype != torch.float32:
            raise ValueError('out must be float32')
        Bo, To, osb, ost, osc = _strides(out, out_layout)
        if Bo != B or To != T - 1:
            raise ValueError(f'out must hold {B} clips of {T - 1} frames, got {Bo} x {To}')
        res, target = out, out
    check(lib().rohm_traj_rederive(nat.handle, x.data_ptr(), sb, st, sc, ptr(m_in), ptr(s_in), ptr(m_out), ptr(s_out), B, T,
                                   target.data_ptr(), osb, ost, osc, stream_ptr(dev)), 'rohm_traj_rederive')
    return res
