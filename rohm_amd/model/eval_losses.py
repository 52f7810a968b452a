"""Evaluation loss reports of the single-stage drivers (`eval_losses(..., compute_loss=True)`, the default of
test_posenet.py:178 / test_trajnet.py:154): `PoseNet.compute_losses_with_smpl` (model/posenet.py:98-194) and
`TrajNet.compute_losses_with_smpl` (model/trajnet.py:277-400), forward only.

The joints come from the HIP kernels (`rohm_repr_joints`, all three recover modes, de-normalising on the fly); what
remains are means of squared differences over small tensors, done with device tensor reductions.  No gradients: these
are reports, training is outside the path.  Pinned to the reference's own methods by tests/golden/eval_losses.npz.
"""
from __future__ import annotations

import torch

from ..data_loaders.motion_representation import joints_from_repr

FOOT = [7, 10, 8, 11]


def _mse(a, b):
    return ((a - b) ** 2).mean()


def _diff(x):
    return x[:, 1:] - x[:, :-1]


def _skating(joints, contact, fps, thr):
    """posenet.py:159-166: mean speed of the foot joints that move while labelled in contact."""
    v = torch.norm(_diff(joints[:, :, FOOT]) * fps, dim=-1)
    mask = (v - thr).gt(0) * contact[:, 0:-1]
    return (v * mask).sum() / mask.sum()


def _three_recoveries(repr_norm, stats, smplx_model, layout):
    return [joints_from_repr(repr_norm, mode, smplx_model, stats=stats, layout=layout)
            for mode in ('joint_abs_traj', 'joint_rel_traj', 'smplx_params')]


def posenet_losses(net, batch, model_output, smplx_model=None, epoch=0):
    """model/posenet.py:98-194.  batch['motion_repr_clean'] and model_output: [bs, 294, 1, T] (normalised)."""
    clean, out = batch['motion_repr_clean'].float(), model_output.float()
    smplx_model = smplx_model if smplx_model is not None else net.smplx_model
    ds = net.dataset
    d = {}
    sq = (clean - out) ** 2
    d['loss_repr_full_body'] = sq[:, net.traj_feat_dim:-4].mean()
    j_clean = joints_from_repr(clean, 'joint_abs_traj', stats=ds, layout='bc1t')
    recs = dict(zip(('abs_traj', 'rel_traj', 'smpl'), _three_recoveries(out, ds, smplx_model, 'bc1t')))
    v_clean = _diff(j_clean)
    for name, j in recs.items():
        d[f'loss_joint_pos_global_from_{name}'] = _mse(j, j_clean)
    for name, j in recs.items():
        d[f'loss_joint_vel_global_from_{name}'] = _mse(_diff(j), v_clean)
    for name, j in recs.items():
        d[f'loss_joint_smooth_from_{name}'] = (_diff(_diff(j)) ** 2).mean()
    d['loss_repr_foot_contact_mse'] = sq[:, -4:].mean()
    std = torch.as_tensor(ds.Std, device=clean.device, dtype=torch.float32)
    mean = torch.as_tensor(ds.Mean, device=clean.device, dtype=torch.float32)
    contact = clean[:, -4:, 0].permute(0, 2, 1) * std[-4:] + mean[-4:]                  # [bs, T, 4]
    for name, j in recs.items():
        d[f'loss_foot_skating_from_{name}'] = _skating(j, contact, net.fps, net.foot_skating_vel_thres)
    w_skate = net.weight_loss_foot_skating if epoch >= net.start_skating_loss_epoch else 0.0
    s3 = lambda stem: sum(d[f'{stem}_from_{n}'] for n in recs)
    d['loss'] = (net.weight_loss_rec_repr_full_body * d['loss_repr_full_body'] +
                 net.weight_loss_repr_foot_contact_mse * d['loss_repr_foot_contact_mse'] +
                 net.weight_loss_joint_pos_global * s3('loss_joint_pos_global') +
                 net.weight_loss_joint_vel_global * s3('loss_joint_vel_global') +
                 net.weight_loss_joint_smooth * s3('loss_joint_smooth') + w_skate * s3('loss_foot_skating'))
    return d


def _angular_velocity(rot, d_rot):
    """utils/other_utils.py estimate_angular_velocity: vee of dR R^T, symmetric entries averaged."""
    w = torch.matmul(d_rot, rot.transpose(-1, -2))
    return torch.stack([(-w[..., 1, 2] + w[..., 2, 1]) / 2.0, (w[..., 0, 2] - w[..., 2, 0]) / 2.0,
                        (-w[..., 0, 1] + w[..., 1, 0]) / 2.0], dim=-1)


def _rot6d_to_rotmat(x):
    """data_loaders/common/quaternion.py:482-501 (interleaved columns, cross without dim)."""
    x = x.reshape(-1, 3, 2)
    a1, a2 = x[:, :, 0], x[:, :, 1]
    b1 = torch.nn.functional.normalize(a1)
    b2 = torch.nn.functional.normalize(a2 - torch.einsum('bi,bi->b', b1, a2).unsqueeze(-1) * b1)
    return torch.stack((b1, b2, torch.cross(b1, b2, dim=-1)), dim=-1)


def trajnet_losses(net, batch, model_output, smplx_model=None):
    """model/trajnet.py:277-400.  batch['motion_repr_clean'] [bs, T, 294], model_output [bs, T, traj_feat_dim]."""
    from ..inference import merge_traj_into_repr
    clean, out = batch['motion_repr_clean'].float(), model_output.float()
    ds = net.dataset
    rec = merge_traj_into_repr(clean, out, net.repr_abs_only, net.traj_feat_dim)
    sq = (clean - rec) ** 2
    d = {'loss_repr_traj_root_rot_angle': sq[:, :, 0].mean(), 'loss_repr_traj_root_l_pos': sq[:, :, 2:4].mean(),
         'loss_repr_traj_root_height': sq[:, :, 6].mean(), 'loss_repr_traj_smplx_rot_6d': sq[:, :, 7:13].mean(),
         'loss_repr_traj_smplx_trans': sq[:, :, 16:19].mean()}
    if not net.repr_abs_only:
        d.update({'loss_repr_traj_root_rot_angle_vel': sq[:, :, 1].mean(), 'loss_repr_traj_root_l_vel': sq[:, :, 4:6].mean(),
                  'loss_repr_traj_smplx_rot_vel': sq[:, :, 13:16].mean(),
                  'loss_repr_traj_smplx_trans_vel': sq[:, :, 19:22].mean(),
                  'loss_repr_traj': sq[..., 0:net.traj_feat_dim].mean()})
    else:
        d['loss_repr_traj'] = sq[..., [0, 2, 3, 6, 7, 8, 9, 10, 11, 12, 16, 17, 18]].mean()
    root_clean = joints_from_repr(clean, 'joint_abs_traj', stats=ds)[:, :, 0]
    roots = dict(zip(('abs_traj', 'rel_traj', 'smpl'), (j[:, :, 0] for j in _three_recoveries(rec, ds, smplx_model, 'btc'))))
    for name, r in roots.items():
        d[f'loss_root_pos_global_from_{name}'] = _mse(r, root_clean)
    for name, r in roots.items():
        d[f'loss_root_vel_global_from_{name}'] = _mse(_diff(r), _diff(root_clean))
    std = torch.as_tensor(ds.Std, device=clean.device, dtype=torch.float32)
    mean = torch.as_tensor(ds.Mean, device=clean.device, dtype=torch.float32)
    den = lambda x, lo, hi: x[..., lo:hi] * std[lo:hi] + mean[lo:hi]
    bs = clean.shape[0]
    R = _rot6d_to_rotmat(den(rec, 7, 13)).reshape(bs, -1, 3, 3)
    rot_vel = _angular_velocity(R[:, 0:-1], R[:, 1:] - R[:, 0:-1])
    d['loss_root_smplx_rot_vel'] = _mse(rot_vel, den(clean, 13, 16)[:, 0:-1])
    d['loss_root_smplx_transl_vel'] = _mse(_diff(den(rec, 16, 19)), den(clean, 19, 22)[:, 0:-1])
    for name, r in roots.items():
        d[f'loss_root_smooth_from_{name}'] = (_diff(_diff(r)) ** 2).mean()
    cos_vel = lambda x: torch.cos(den(x, 0, 1)[:, 1:] * 2) - torch.cos(den(x, 0, 1)[:, 0:-1] * 2)
    cv_rec = cos_vel(rec)
    d['loss_root_rot_cos_vel_from_abs_traj'] = _mse(cos_vel(clean), cv_rec)
    d['loss_root_rot_cos_smooth_from_abs_traj'] = (_diff(cv_rec) ** 2).mean()
    if net.repr_abs_only:          # the relative channels of `rec` are the ground truth then (trajnet.py:381-384)
        for k in ('loss_root_pos_global_from_rel_traj', 'loss_root_vel_global_from_rel_traj', 'loss_root_smooth_from_rel_traj'):
            d[k] = torch.tensor(0.0, device=clean.device)
    s3 = lambda stem: sum(d[f'{stem}_from_{n}'] for n in roots)
    d['loss'] = (net.weight_loss_root_rec_repr * d['loss_repr_traj'] +
                 net.weight_loss_root_pos_global * s3('loss_root_pos_global') +
                 net.weight_loss_root_vel_global * s3('loss_root_vel_global') +
                 net.weight_loss_root_rot_vel_from_abs_traj * d['loss_root_rot_cos_vel_from_abs_traj'] +
                 net.weight_loss_root_smplx_transl_vel * d['loss_root_smplx_transl_vel'] +
                 net.weight_loss_root_smplx_rot_vel * d['loss_root_smplx_rot_vel'] +
                 net.weight_loss_root_smooth * s3('loss_root_smooth') +
                 net.weight_loss_root_rot_cos_smooth_from_abs_traj * d['loss_root_rot_cos_smooth_from_abs_traj'])
    return d
