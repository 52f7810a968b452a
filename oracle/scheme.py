"""TEST INFRASTRUCTURE ONLY: CPU restatement of the AMASS driver's inference-iteration loop
(test_amass_full.py:218-386): TrajNet -> trajectory re-derivation -> condition / occlusion-mask assembly ->
PoseNet -> (next iteration) TrajControl -> ... on top of the oracle networks, sampler and geometry.

The drivers are scripts (argparse + datasets + checkpoints at module level), not importable functions; their loop
bodies are pinned by EXECUTING the scripts' own text (test_amass_full.py:217-384, test_prox_egobody.py:214-324) with stub
samplers in oracle/make_golden.py::golden_scheme -> tests/golden/scheme.npz, which tests/test_scheme_oracle.py holds
this restatement to (9 configurations).  Noise is injected; the random starts of the 'full' mask (:363-367, one draw per
masked iteration) are an argument.
"""
from __future__ import annotations

import numpy as np
import torch

from . import diffusion as D
from . import geometry as G
from . import nets
from . import rederive as RD

LOWER_JOINTS = [1, 2, 4, 5, 7, 8, 10, 11]                              # test_amass_full.py:343
UPPER_JOINTS = [3, 6, 9, 12, 13, 14, 15, 16, 17, 18, 19, 20]           # :353
ABS_TRAJ_CH = [0, 2, 3, 6, 7, 8, 9, 10, 11, 12, 16, 17, 18]            # repr_abs_only channels (:275-280)


def infill_mask(B, T, ratio, traj_feat_dim):
    """:218-229 -- frames [65, 65 + int(ratio * 145)) of every clip are hidden."""
    m = torch.ones(B, T)
    n = int(ratio * 145)
    for b in range(B):
        m[b, 65:65 + n] = 0
    return m.unsqueeze(-1).repeat(1, 1, traj_feat_dim)


def merge_traj(motion_repr, traj, repr_abs_only, traj_feat_dim):
    """:272-281"""
    if not repr_abs_only:
        return torch.cat([traj, motion_repr[:, :, traj_feat_dim:]], dim=-1)
    out = motion_repr.clone()
    for k, c in enumerate(ABS_TRAJ_CH):
        out[..., c] = traj[..., k]
    return out


def occlusion_mask(cond, scheme, traj_feat_dim, start=None, end=None):
    """:341-372 on cond [B, T, 294] (in place)."""
    if scheme in ('lower', 'upper'):
        jid = np.asarray(LOWER_JOINTS if scheme == 'lower' else UPPER_JOINTS)
        for k in range(3):
            cond[:, :, traj_feat_dim + jid * 3 + k] = 0.
            cond[:, :, traj_feat_dim + 66 + jid * 3 + k] = 0.
        for k in range(6):
            cond[:, :, traj_feat_dim + 132 + (jid - 1) * 6 + k] = 0.
        cond[:, :, -4:] = 0.
    elif scheme == 'full':
        cond[:, :, -4:] = 0.
        for b in range(cond.shape[0]):
            cond[b, int(start[b]):int(end[b]), 22:] = 0
    return cond


def oracle_stages(sd_traj, sd_ctrl, sd_pose, tab_traj, tab_pose, idx_traj, idx_pose, stats_pose, body, args, noise,
                  grad_type='amass', camera=None):
    """Stage callables running the oracle networks + sampler: noise['traj'][it] / noise['pose'][it] =
    (x_T, [step noises]).  `grad_type` is what the driver passes to PoseNet's eval_losses ('amass':
    test_amass_full.py:383, 'prox': test_prox_egobody.py:323); 'prox' needs `camera` = the batch entries
    guide_2d_projection_with_smpl reads + cam_R / cam_t of the dataset.  `idx_pose` is the list of PoseNet timesteps as the
    sampler would visit them; `early_stop` cuts it to its first 980 entries (gaussian_diffusion_posenet.py:625-626)."""
    mean_p, std_p = (torch.as_tensor(v) for v in stats_pose)
    guidance = {'skating': lambda x0, i: G.guide_skating(x0, mean_p, std_p, body)}
    if camera is not None:
        guidance['2d'] = lambda x0, i: G.guide_2d_projection(
            x0, mean_p, std_p, body, camera['transf_matrix'], camera['focal_length'], camera['camera_center'],
            camera['keypoints_2d'], torch.as_tensor(camera['cam_R']), torch.as_tensor(camera['cam_t']))

    def traj_stage(it, batch):
        cond, B = batch['cond'], batch['cond'].shape[0]
        cc = batch.get('control_cond') if it > 0 else None
        sd = sd_traj if it == 0 else sd_ctrl
        fn = lambda x, i: nets.trajnet_forward(sd, x, cond, torch.full((B,), i, dtype=torch.int64), control_cond=cc)
        with torch.no_grad():
            return D.p_sample_loop(fn, noise['traj'][it][0], noise['traj'][it][1], tab_traj, idx_traj)

    def pose_stage(it, batch):
        cond, B = batch['cond'], batch['cond'].shape[0]
        fn = lambda x, i: nets.posenet_forward(sd_pose, x, cond, torch.full((B,), i, dtype=torch.int64))
        with torch.no_grad():
            return D.p_sample_loop(fn, noise['pose'][it][0], noise['pose'][it][1], tab_pose,
                                   idx_pose[:980] if args.early_stop else idx_pose, guidance=guidance,
                                   grad_type=grad_type if args.cond_fn_with_grad else None, early_stop=args.early_stop)
    return traj_stage, pose_stage


def amass_iterations(traj_stage, pose_stage, batch_traj, batch_pose, stats_traj, stats_pose, body, args):
    """The loop of test_amass_full.py:229-386 around two stage callables (`traj_stage(it, batch_traj)` ->
    [B,144,tfd], `pose_stage(it, batch_pose)` -> [B,294,1,143]; see `oracle_stages`).  batch_traj: 'cond'
    [B,144,tfd], 'motion_repr_clean' / 'motion_repr_noisy' [B,144,294]; batch_pose: 'motion_repr_clean' /
    'motion_repr_noisy' [B,144,294].  Returns (val_output_pose, val_output_traj, [traj_rec_full per iteration])."""
    tfd = 13 if args.repr_abs_only else 22
    B, T = batch_traj['cond'].shape[:2]
    mask_traj = None
    if args.infill_traj:
        mask_traj = infill_mask(B, T, args.traj_mask_ratio, tfd)
        batch_traj['cond'][:, :, 0:tfd] = batch_traj['cond'][:, :, 0:tfd] * mask_traj
    val_pose = val_traj = None
    recs = []
    for it in range(args.sample_iter):
        if args.iter2_cond_noisy_traj and args.infill_traj and it > 0:                      # :231-235
            batch_traj['cond'][:, :, 0:tfd] = batch_traj['cond'][:, :, 0:tfd] * mask_traj + val_traj * (1 - mask_traj)
        if it > 0:                                                                          # :252-257
            cc = torch.zeros(B, T, 272)
            cc[:, 0:-1] = val_pose[:, :, 0].permute(0, 2, 1)[:, :, -272:]
            cc[:, -1] = cc[:, -2].clone()
            batch_traj['control_cond'] = cc
        val_traj = traj_stage(it, batch_traj)                                               # :242-266
        rec = merge_traj(batch_traj['motion_repr_clean'], val_traj, args.repr_abs_only, tfd)   # :269-281
        if it == 0:
            batch_traj['motion_repr_noisy'] = rec
        if it < args.sample_iter - 1 and not args.iter2_cond_noisy_traj:
            batch_traj['cond'] = val_traj
        traj_rec_full = torch.from_numpy(RD.rederive_traj(rec, *stats_traj, *stats_pose, body))   # :283-311 (float64)
        recs.append(traj_rec_full)

        if it == 0:                                                                         # :314-316
            batch_pose['motion_repr_noisy'] = batch_pose['motion_repr_noisy'][:, 0:-1]
            batch_pose['motion_repr_clean'] = batch_pose['motion_repr_clean'][:, 0:-1]
        if not args.input_noise:                                                            # :318-331
            cond = batch_pose['motion_repr_clean'].clone() if it == 0 else \
                batch_pose['motion_repr_clean'].clone()[:, :, 0].permute(0, 2, 1)
        elif args.iter2_cond_noisy_pose or it == 0:
            cond = batch_pose['motion_repr_noisy'].clone()
        else:
            cond = val_pose[:, :, 0].permute(0, 2, 1)
        cond = cond.contiguous()
        if not (args.mask_scheme == 'lower' and not args.input_noise):                      # :333-334
            cond[:, :, 0:22] = traj_rec_full
        mask_iter_num = args.sample_iter if args.iter2_cond_noisy_pose else 1               # :338-339
        if it < mask_iter_num:
            if args.mask_scheme == 'full' and not args.infill_traj:
                # the script draws a NEW random start in every masked iteration (:363-367); `full_mask_start` is the
                # list of those draws (or one tensor reused for all iterations)
                # `clip_len` there is motion_repr_clean.shape[1] (:335), which is T = 143 in iteration 0 and 294 afterwards
                # (the tensor was permuted to [bs, 294, 1, T] at :375): later draws range over [0, 293) and mostly fall
                # outside the clip.  Reproduced as is.
                fms = args.full_mask_start
                start = (fms[it] if isinstance(fms, (list, tuple)) else fms).long()
                end = torch.clamp(start + 30, max=batch_pose['motion_repr_clean'].shape[1])
            elif args.mask_scheme == 'full':
                start = torch.full((B,), 65, dtype=torch.long)
                end = start + int(args.traj_mask_ratio * 145)
            else:
                start = end = None
            occlusion_mask(cond, args.mask_scheme, 22, start, end)
        batch_pose['cond'] = cond.permute(0, 2, 1).unsqueeze(-2)                            # :374
        if it == 0:
            batch_pose['motion_repr_clean'] = batch_pose['motion_repr_clean'].permute(0, 2, 1).unsqueeze(-2)
        val_pose = pose_stage(it, batch_pose)                                               # :377-386
    return val_pose, val_traj, recs


def prox_iterations(traj_stage, pose_stage, batch_traj, batch_pose, stats_traj, stats_pose, body, args):
    """The loop of test_prox_egobody.py:213-313: the noisy representation is the carrier, the visibility mask
    (`mask_vec_vis` [B, T+2, 294], :291-294) replaces the mask schemes.  Same stage callables as `amass_iterations`."""
    tfd = 13 if args.repr_abs_only else 22
    B, T = batch_traj['cond'].shape[:2]
    val_pose = val_traj = None
    recs = []
    for it in range(args.sample_iter):
        if it > 0:                                                                          # :230-234
            cc = torch.zeros(B, T, 272)
            cc[:, 0:-1] = val_pose[:, :, 0].permute(0, 2, 1)[:, :, -272:]
            cc[:, -1] = cc[:, -2].clone()
            batch_traj['control_cond'] = cc
        val_traj = traj_stage(it, batch_traj)
        rec = merge_traj(batch_traj['motion_repr_noisy'], val_traj, args.repr_abs_only, tfd)   # :245-254
        if it == 0:
            batch_traj['motion_repr_noisy'] = rec
        if it < args.sample_iter - 1 and not args.iter2_cond_noisy_traj:
            batch_traj['cond'] = val_traj
        traj_rec_full = torch.from_numpy(RD.rederive_traj(rec, *stats_traj, *stats_pose, body))   # :258-287
        recs.append(traj_rec_full)
        if it == 0:                                                                         # :290-291
            batch_pose['motion_repr_noisy'] = batch_pose['motion_repr_noisy'][:, 0:-1]
        if args.iter2_cond_noisy_pose:                                                      # :292-300
            cond = batch_pose['motion_repr_noisy'].clone()
            if it > 0:
                cond = cond[:, :, 0].permute(0, 2, 1)
        elif it == 0:
            cond = batch_pose['motion_repr_noisy'].clone()
        else:
            cond = val_pose[:, :, 0].permute(0, 2, 1)
        cond = cond.contiguous()
        cond[:, :, 0:22] = traj_rec_full                                                    # :302
        mask_iter_num = args.sample_iter if args.iter2_cond_noisy_pose else 1
        if it < mask_iter_num:                                                              # :305-309
            cond = cond * batch_pose['mask_vec_vis'][:, 0:-2, :]
            cond[:, :, -4:] = 0.
        if it == 0:
            batch_pose['motion_repr_noisy'] = batch_pose['motion_repr_noisy'].permute(0, 2, 1).unsqueeze(-2)
        batch_pose['cond'] = cond.permute(0, 2, 1).unsqueeze(-2)
        val_pose = pose_stage(it, batch_pose)
    return val_pose, val_traj, recs
