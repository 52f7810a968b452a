"""The drivers' inference-iteration loop as a callable, device-resident end to end (SURVEY.md §8(f) N1 + N2).

Reference: test_amass_full.py:218-386 (AMASS) and test_prox_egobody.py:213-313 (PROX / EgoBody).  There the loop
lives inline in the scripts and leaves the GPU between the stages (a numpy `get_repr_smplx` loop over the batch,
Python loops over the batch for the masks).  Here the same statements are tensor operations on the device plus
one kernel (`rohm_traj_rederive`), and the networks / samplers are the HIP ones behind `eval_losses`.

The functions mutate `test_batch_traj` / `test_batch_pose` exactly where the scripts do, so a driver can replace
its loop body by one call and keep using the dicts afterwards (e.g. `test_batch_pose['motion_repr_clean']` in the
[bs, 294, 1, T] layout for the metrics, test_amass_full.py:388-392).
"""
from __future__ import annotations

import numpy as np
import torch

from .data_loaders.motion_representation import rederive_traj

LOWER_JOINTS = (1, 2, 4, 5, 7, 8, 10, 11)                              # test_amass_full.py:343
UPPER_JOINTS = (3, 6, 9, 12, 13, 14, 15, 16, 17, 18, 19, 20)           # :353
ABS_TRAJ_CH = (0, 2, 3, 6, 7, 8, 9, 10, 11, 12, 16, 17, 18)            # repr_abs_only channels (:275-280)


def traj_infill_mask(batch_size, clip_len, ratio, traj_feat_dim, device):
    """test_amass_full.py:218-227: frames [65, 65 + int(ratio * 145)) of every clip hidden -> [bs, T, traj_feat_dim]."""
    m = torch.ones(batch_size, clip_len, device=device)
    m[:, 65:65 + int(ratio * 145)] = 0
    return m.unsqueeze(-1).repeat(1, 1, traj_feat_dim)


def merge_traj_into_repr(motion_repr, val_output_traj, repr_abs_only, traj_feat_dim):
    """test_amass_full.py:269-281 -> `motion_repr_clean_root_rec` [bs, T, 294]."""
    if not repr_abs_only:
        return torch.cat([val_output_traj, motion_repr[:, :, traj_feat_dim:]], dim=-1)
    out = motion_repr.clone()
    out[..., list(ABS_TRAJ_CH)] = val_output_traj
    return out


def build_control_cond(val_output_pose, clip_len, pose_feat_dim):
    """test_amass_full.py:254-257: PoseNet's local pose [bs, 294, 1, T-1] -> TrajControl condition [bs, T, 272]
    (last frame repeated)."""
    local = val_output_pose[:, -pose_feat_dim:, 0].permute(0, 2, 1)
    cc = torch.empty(local.shape[0], clip_len, pose_feat_dim, device=local.device, dtype=local.dtype)
    cc[:, 0:-1] = local
    cc[:, -1] = local[:, -1]
    return cc


def _joint_channels(joint_ids, traj_feat_dim, device):
    jid = np.asarray(joint_ids)
    ch = [traj_feat_dim + jid * 3 + k for k in range(3)]
    ch += [traj_feat_dim + 22 * 3 + jid * 3 + k for k in range(3)]
    ch += [traj_feat_dim + 22 * 3 + 22 * 3 + (jid - 1) * 6 + k for k in range(6)]
    return torch.as_tensor(np.sort(np.concatenate(ch)), device=device, dtype=torch.long)


def apply_occlusion_mask(cond, mask_scheme, traj_feat_dim=22, start=None, end=None):
    """test_amass_full.py:341-372 on `cond` [bs, T, 294], in place, without the Python loop over the batch.
    'full' needs `start` / `end` ([bs] frame indices)."""
    if mask_scheme in ('lower', 'upper'):
        ch = _joint_channels(LOWER_JOINTS if mask_scheme == 'lower' else UPPER_JOINTS, traj_feat_dim, cond.device)
        cond.index_fill_(2, ch, 0.)
        cond[:, :, -4:] = 0.
    elif mask_scheme == 'full':
        cond[:, :, -4:] = 0.
        frames = torch.arange(cond.shape[1], device=cond.device)[None]
        hide = (frames >= start.to(cond.device)[:, None]) & (frames < end.to(cond.device)[:, None])
        cond[:, :, 22:].masked_fill_(hide[..., None], 0.)
    else:
        raise ValueError(f'unknown mask_scheme {mask_scheme!r}')
    return cond


def apply_visibility_mask(cond, mask_vec_vis):
    """test_prox_egobody.py:291-294: per-channel visibility [bs, T+2, 294] applied to `cond` [bs, T, 294]."""
    cond = cond * mask_vec_vis[:, 0:-2, :]
    cond[:, :, -4:] = 0.
    return cond


def _traj_stage(args, it, models, diffusions, batch_traj, val_output_pose, shape, pose_feat_dim, smplx_model):
    if it == 0:
        model, diff = models['trajnet'], diffusions['trajnet']
    else:
        batch_traj['control_cond'] = build_control_cond(val_output_pose, shape[1], pose_feat_dim)
        model, diff = models['trajnet_control'], diffusions['trajnet_control']
    _, out = diff.eval_losses(model=model, batch=batch_traj, shape=shape, progress=False, clip_denoised=False,
                              timestep_respacing=args.timestep_respacing_eval,
                              cond_fn_with_grad=args.cond_fn_with_grad, compute_loss=False, smplx_model=smplx_model)
    return out


def run_amass_iterations(args, models, diffusions, test_batch_traj, test_batch_pose, test_traj_dataset,
                         test_pose_dataset, smplx_neutral, full_mask_start=None):
    """test_amass_full.py:218-386.  `models` / `diffusions`: dicts with keys 'trajnet', 'trajnet_control',
    'posenet' (the objects the script builds at :132-188).  `args`: the script's namespace (sample_iter,
    repr_abs_only, infill_traj, traj_mask_ratio, iter2_cond_noisy_traj, iter2_cond_noisy_pose, input_noise,
    mask_scheme, cond_fn_with_grad, early_stop, timestep_respacing_eval).  `full_mask_start`: the random start
    frames of the 'full' mask: None = drawn with the script's formula in every masked iteration, as the script does; a
    tensor [bs] = the same starts in every iteration; a list = one tensor per iteration.
    Returns (val_output_pose [bs,294,1,143], val_output_traj [bs,144,tfd], [traj_rec_full per iteration])."""
    dev = test_batch_traj['cond'].device
    tfd, pfd = test_traj_dataset.traj_feat_dim, test_traj_dataset.pose_feat_dim
    mask_traj = start = end = None
    if args.infill_traj:                                                                    # :218-229
        bs, clip_len = test_batch_traj['cond'].shape[:2]
        mask_traj = traj_infill_mask(bs, clip_len, args.traj_mask_ratio, tfd, dev)
        start = torch.full((bs,), 65, dtype=torch.long, device=dev)
        end = start + int(args.traj_mask_ratio * 145)
        test_batch_traj['cond'][:, :, 0:tfd] = test_batch_traj['cond'][:, :, 0:tfd] * mask_traj
    val_output_pose = val_output_traj = None
    recs = []
    for it in range(args.sample_iter):
        if args.iter2_cond_noisy_traj and args.infill_traj and it > 0:                      # :231-235
            vis = test_batch_traj['cond'][:, :, 0:tfd] * mask_traj
            test_batch_traj['cond'][:, :, 0:tfd] = vis + val_output_traj * (1 - mask_traj)
        shape = list(test_batch_traj['motion_repr_clean'][:, :, 0:tfd].shape)
        val_output_traj = _traj_stage(args, it, models, diffusions, test_batch_traj, val_output_pose, shape, pfd,
                                      smplx_neutral)
        rec = merge_traj_into_repr(test_batch_traj['motion_repr_clean'], val_output_traj, args.repr_abs_only, tfd)
        if it == 0:
            test_batch_traj['motion_repr_noisy'] = rec
        if it < args.sample_iter - 1 and not args.iter2_cond_noisy_traj:
            test_batch_traj['cond'] = val_output_traj

        if it == 0:                                                                         # :314-316
            test_batch_pose['motion_repr_noisy'] = test_batch_pose['motion_repr_noisy'][:, 0:-1]
            test_batch_pose['motion_repr_clean'] = test_batch_pose['motion_repr_clean'][:, 0:-1]
        if not args.input_noise:                                                            # :318-331
            cond = test_batch_pose['motion_repr_clean'].clone()
            if it > 0:
                cond = cond[:, :, 0].permute(0, 2, 1)
        elif args.iter2_cond_noisy_pose or it == 0:
            cond = test_batch_pose['motion_repr_noisy'].clone()
        else:
            cond = val_output_pose[:, :, 0].permute(0, 2, 1)
        cond = cond.contiguous()
        if not (args.mask_scheme == 'lower' and not args.input_noise):                      # :333-336, fused
            rederive_traj(rec, test_traj_dataset, test_pose_dataset, smplx_neutral, out=cond)
            recs.append(cond[:, :, 0:22].clone())
        else:
            recs.append(rederive_traj(rec, test_traj_dataset, test_pose_dataset, smplx_neutral))
        mask_iter_num = args.sample_iter if args.iter2_cond_noisy_pose else 1               # :338-339
        if it < mask_iter_num:
            if args.mask_scheme == 'full' and not args.infill_traj:                         # :361-368
                # The script draws a NEW start in every masked iteration with `clip_len = motion_repr_clean.shape[1]`
                # (:335, :363-367), which is T = 143 in iteration 0 and 294 afterwards (the tensor is [bs, 294, 1, T]
                # from :375 on): later draws range over [0, 293) and mostly fall outside the clip.  Reproduced as is.
                bs, clip_len = test_batch_pose['motion_repr_clean'].shape[:2]
                if full_mask_start is None:
                    start = torch.FloatTensor(bs).uniform_(0, clip_len - 1).long().to(dev)
                elif isinstance(full_mask_start, (list, tuple)):
                    start = full_mask_start[it].to(dev)
                else:
                    start = full_mask_start.to(dev)
                end = torch.clamp(start + 30, max=clip_len)
            apply_occlusion_mask(cond, args.mask_scheme, test_pose_dataset.traj_feat_dim, start, end)
        test_batch_pose['cond'] = cond.permute(0, 2, 1).unsqueeze(-2)                       # :374 (view, as the script)
        if it == 0:
            test_batch_pose['motion_repr_clean'] = test_batch_pose['motion_repr_clean'].permute(0, 2, 1).unsqueeze(-2)
        shape = list(test_batch_pose['motion_repr_clean'].shape)
        _, val_output_pose = diffusions['posenet'].eval_losses(
            model=models['posenet'], batch=test_batch_pose, shape=shape, progress=False, clip_denoised=False,
            timestep_respacing=args.timestep_respacing_eval, cond_fn_with_grad=args.cond_fn_with_grad,
            early_stop=args.early_stop, compute_loss=False, grad_type='amass', smplx_model=smplx_neutral)
    return val_output_pose, val_output_traj, recs


def run_prox_iterations(args, models, diffusions, test_batch_traj, test_batch_pose, test_traj_dataset,
                        test_pose_dataset, smplx_neutral):
    """test_prox_egobody.py:213-313: as the AMASS loop, with the noisy representation as the carrier, the
    visibility mask (`mask_vec_vis`, :291-294) instead of a mask scheme, and `grad_type='prox'`."""
    tfd, pfd = test_traj_dataset.traj_feat_dim, test_traj_dataset.pose_feat_dim
    val_output_joint = val_output_traj = None
    recs = []
    for it in range(args.sample_iter):
        shape = list(test_batch_traj['motion_repr_noisy'][:, :, 0:tfd].shape)
        val_output_traj = _traj_stage(args, it, models, diffusions, test_batch_traj, val_output_joint, shape, pfd,
                                      smplx_neutral)
        rec = merge_traj_into_repr(test_batch_traj['motion_repr_noisy'], val_output_traj, args.repr_abs_only, tfd)
        if it == 0:
            test_batch_traj['motion_repr_noisy'] = rec
        if it < args.sample_iter - 1 and not args.iter2_cond_noisy_traj:
            test_batch_traj['cond'] = val_output_traj
        if it == 0:                                                                         # :265-266
            test_batch_pose['motion_repr_noisy'] = test_batch_pose['motion_repr_noisy'][:, 0:-1]
        if args.iter2_cond_noisy_pose:                                                      # :267-275
            cond = test_batch_pose['motion_repr_noisy'].clone()
            if it > 0:
                cond = cond[:, :, 0].permute(0, 2, 1)
        elif it == 0:
            cond = test_batch_pose['motion_repr_noisy'].clone()
        else:
            cond = val_output_joint[:, :, 0].permute(0, 2, 1)
        cond = cond.contiguous()
        rederive_traj(rec, test_traj_dataset, test_pose_dataset, smplx_neutral, out=cond)   # :238-287 + :277
        recs.append(cond[:, :, 0:22].clone())
        mask_iter_num = args.sample_iter if args.iter2_cond_noisy_pose else 1
        if it < mask_iter_num:
            cond = apply_visibility_mask(cond, test_batch_pose['mask_vec_vis'])
        if it == 0:
            test_batch_pose['motion_repr_noisy'] = test_batch_pose['motion_repr_noisy'].permute(0, 2, 1).unsqueeze(-2)
        test_batch_pose['cond'] = cond.permute(0, 2, 1).unsqueeze(-2)
        shape = list(test_batch_pose['motion_repr_noisy'].shape)
        _, val_output_joint = diffusions['posenet'].eval_losses(
            model=models['posenet'], batch=test_batch_pose, shape=shape, progress=False, clip_denoised=False,
            timestep_respacing=args.timestep_respacing_eval, cond_fn_with_grad=args.cond_fn_with_grad,
            early_stop=args.early_stop, compute_loss=False, grad_type='prox', smplx_model=smplx_neutral)
    return val_output_joint, val_output_traj, recs
