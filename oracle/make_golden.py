"""Generate tests/golden/*.npz by running the REFERENCE's own modules (build container only).

    python -m oracle.make_golden

The reference (/root/reference) is imported with the cv2/smplx stubs of oracle/refload.py; weights and
inputs come from rohm_amd.utils.synth (seeded, reproducible), so a fixture stores only seeds and the
reference's outputs.  The committed fixtures pin (a) the oracle restatement (CPU tests) and (b) the HIP
path (GPU tests) to what the reference computes.
"""
from __future__ import annotations

import os
import sys
import warnings

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import refload  # noqa: E402
from rohm_amd.utils import synth  # noqa: E402

OUT = os.path.join(ROOT, 'tests', 'golden')


class _DS:
    pose_feat_dim = 272
    traj_feat_dim = 22


def seeded(seed, *shape):
    g = np.random.Generator(np.random.PCG64(seed))
    return torch.from_numpy(g.standard_normal(size=shape).astype(np.float32))


class _Args:
    noise_schedule = 'cosine'
    sigma_small = True


def golden_rederive(ref):
    from oracle import geometry as G
    body = G.BodyModel(synth.synthetic_smplx_tensors(0))
    # ---- between-stage trajectory re-derivation: the reference's driver lines ----------
    mr = ref.motion_repr
    mean_in, std_in = synth.synthetic_stats(0)
    mean_out, std_out = synth.synthetic_stats(1)
    rn = synth.plausible_motion(5, 2, 144, mean_in, std_in)[:, :, 0].permute(0, 2, 1).contiguous()   # [2,144,294]
    den = rn.numpy() * std_in + mean_in
    dd = {k: torch.from_numpy(np.ascontiguousarray(v)) for k, v in G.split_repr(den).items()}
    jts, _ = mr.recover_from_repr_smpl(dd, recover_mode='smplx_params', smplx_model=body, return_verts=True)
    jts = jts.detach().cpu().numpy()
    full_list = []
    for i in range(len(jts)):
        g_aa = ref.konia.rotation_matrix_to_angle_axis(ref.quaternion.rot6d_to_rotmat(dd['smplx_rot_6d'][i]))
        b_aa = ref.konia.rotation_matrix_to_angle_axis(
            ref.quaternion.rot6d_to_rotmat(dd['smplx_body_pose_6d'][i].reshape(-1, 6))).reshape(-1, 21, 3)
        prm = {'transl': dd['smplx_trans'][i].numpy(), 'global_orient': g_aa.numpy(),
               'body_pose': b_aa.reshape(-1, 63).numpy(), 'betas': dd['smplx_betas'][i].numpy()}
        rd = mr.get_repr_smplx(positions=jts[i], smplx_params_dict=prm, feet_vel_thre=5e-5)
        full = np.concatenate([rd[k] for k in ref.other_utils.REPR_LIST], axis=-1)
        full_list.append((full - mean_out) / std_out)
    full_ref = np.asarray(full_list)                                   # float64 [2,143,294]
    # degenerate frames: hips/shoulders stacked along z -> zero forward -> NaN quaternion; the reference patches
    # only the first such frame (motion_representation.py:213-215)
    gq = np.random.Generator(np.random.PCG64(77))
    pos_nan = gq.standard_normal((12, 22, 3)).astype(np.float32)
    for k in (5, 9):
        pos_nan[k, [1, 16]] = pos_nan[k, [2, 17]] + np.array([0, 0, 0.3], np.float32)
    prm_nan = {'transl': gq.standard_normal((12, 3)).astype(np.float32),
               'global_orient': (gq.standard_normal((12, 3)) * 0.5).astype(np.float32),
               'body_pose': (gq.standard_normal((12, 63)) * 0.3).astype(np.float32),
               'betas': gq.standard_normal((12, 10)).astype(np.float32)}
    rd = mr.get_repr_smplx(positions=pos_nan, smplx_params_dict=prm_nan, feet_vel_thre=5e-5)
    full_nan = np.concatenate([rd[k] for k in ref.other_utils.REPR_LIST], axis=-1)
    np.savez_compressed(os.path.join(OUT, 'rederive.npz'), body_seed=0, stats_in_seed=0, stats_out_seed=1, motion_seed=5,
                        joints=jts, full_ref=full_ref, pos_nan=pos_nan, full_nan=full_nan,
                        **{'nan_' + k: v for k, v in prm_nan.items()})


def golden_ddim(ref):
    """The reference's `ddim_sample` cannot be called as written (it omits `batch` when it calls p_mean_variance,
    gaussian_diffusion_posenet.py:681-688, and no driver reaches it).  Its BODY -- the DDIM update -- runs fine once
    p_mean_variance is stubbed to return a given pred_xstart: that pins the update formula."""
    betas = ref.gd_posenet.get_named_beta_schedule('cosine', 1000)
    gd = ref.gd_posenet.GaussianDiffusionPoseNet(betas=betas, model_mean_type=ref.gd_posenet.ModelMeanType.START_X,
                                                 model_var_type=ref.gd_posenet.ModelVarType.FIXED_SMALL,
                                                 loss_type=ref.gd_posenet.LossType.MSE, device='cpu')
    x, x0 = seeded(501, 2, 16, 1, 9), seeded(502, 2, 16, 1, 9)      # elementwise formula: a small tensor pins it
    gd.p_mean_variance = lambda model, x_, t_, **kw: {'pred_xstart': x0}
    out = {}
    for k, (i, eta) in enumerate([(999, 0.0), (500, 0.0), (37, 0.7), (1, 1.0), (0, 1.0)]):
        torch.manual_seed(600 + k)
        r = gd.ddim_sample(None, x, torch.tensor([i, i]), eta=eta)
        out[f'case{k}'] = r['sample'].numpy()
        out[f'case{k}_i'], out[f'case{k}_eta'], out[f'case{k}_seed'] = i, eta, 600 + k
    np.savez_compressed(os.path.join(OUT, 'ddim.npz'), x_seed=501, x0_seed=502, n_cases=5, **out)
    print('ddim.npz', os.path.getsize(os.path.join(OUT, 'ddim.npz')))


def golden_rel(ref):
    """recover_from_repr_smpl(recover_mode='joint_rel_traj') (used by test_trajnet.py:204 and the training losses)."""
    from oracle import geometry as G
    mean, std = synth.synthetic_stats(0)
    x0 = synth.plausible_motion(3, 2, 143, mean, std)
    full = x0[:, :, 0].permute(0, 2, 1) * torch.from_numpy(std) + torch.from_numpy(mean)
    j_rel = ref.motion_repr.recover_from_repr_smpl(G.split_repr(full), recover_mode='joint_rel_traj', smplx_model=None)
    np.savez_compressed(os.path.join(OUT, 'recover_rel.npz'), stats_seed=0, motion_seed=3, j_rel=j_rel.numpy())
    print('recover_rel.npz', os.path.getsize(os.path.join(OUT, 'recover_rel.npz')))


def golden_eval_losses(ref):
    """The reference's own `compute_losses_with_smpl` of PoseNet and TrajNet (the eval report of test_posenet.py /
    test_trajnet.py) on synthetic clips, with the oracle body model standing in for smplx."""
    from oracle import geometry as G
    body = G.BodyModel(synth.synthetic_smplx_tensors(0))
    mean, std = synth.synthetic_stats(0)

    class DS:
        pose_feat_dim, traj_feat_dim, joints_num = 272, 22, 22
        Mean, Std = mean, std
    out = {}
    pw = dict(weight_loss_rec_repr_full_body=1.0, weight_loss_repr_foot_contact_mse=0.5, weight_loss_joint_pos_global=2.0,
              weight_loss_joint_vel_global=3.0, weight_loss_joint_smooth=0.7, weight_loss_foot_skating=0.3,
              start_skating_loss_epoch=0)
    pnet = ref.posenet.PoseNet(DS(), 294, latent_dim=64, ff_size=64, num_layers=1, num_heads=1, traj_feat_dim=22,
                               device='cpu', **pw).eval()
    clean = synth.plausible_motion(11, 3, 143, mean, std)
    rec = clean + 0.05 * seeded(12, 3, 294, 1, 143)
    with torch.no_grad():
        d = pnet.compute_losses_with_smpl({'motion_repr_clean': clean}, rec, smplx_model=body, epoch=0)
    for k, v in d.items():
        out['posenet_' + k] = np.float64(v)
    tw = dict(weight_loss_root_rec_repr=1.0, weight_loss_root_pos_global=2.0, weight_loss_root_vel_global=3.0,
              weight_loss_root_rot_vel_from_abs_traj=0.4, weight_loss_root_smplx_transl_vel=0.6,
              weight_loss_root_smplx_rot_vel=0.8, weight_loss_root_smooth=0.9,
              weight_loss_root_rot_cos_smooth_from_abs_traj=1.1)
    clean_t = clean[:, :, 0].permute(0, 2, 1).contiguous()[:, :128]          # [3, 128, 294]
    for abs_only, dim in ((True, 13), (False, 22)):
        DS.traj_feat_dim = dim
        tnet = ref.trajnet.TrajNet(time_dim=32, cond_dim=dim, mid_dim=64, traj_feat_dim=dim, device='cpu', dataset=DS(),
                                   repr_abs_only=abs_only, trajcontrol=False, **tw).eval()
        mo = seeded(13 + dim, 3, 128, dim) * 0.3
        with torch.no_grad():
            d = tnet.compute_losses_with_smpl({'motion_repr_clean': clean_t}, mo, smplx_model=body)
        for k, v in d.items():
            out[f'trajnet{dim}_' + k] = np.float64(v)
    np.savez_compressed(os.path.join(OUT, 'eval_losses.npz'), **out)
    print('eval_losses.npz', len(out), 'entries')


GUIDED_CASES = (('prox', 101), ('prox', 100), ('prox', 50), ('prox', 1), ('prox', 0),
                ('amass', 51), ('amass', 50), ('amass', 1), ('amass', 0))


def guided_step_inputs(g_or_seeds, B=2):
    """The seeded inputs of tests/golden/guided_step.npz (shared with the tests so both sides build the same bytes)."""
    s = {k: int(g_or_seeds[k]) for k in ('stats_seed', 'x_seed', 'xn_seed', 'cond_seed', 'cam_seed')}
    mean, std = synth.synthetic_stats(s['stats_seed'])
    x = synth.plausible_motion(s['x_seed'], B, 143, mean, std) + 0.05 * seeded(s['xn_seed'], B, 294, 1, 143)
    cond = synth.plausible_motion(s['cond_seed'], B, 143, mean, std)
    cam = synth.synthetic_camera_batch(s['cam_seed'], B)
    return mean, std, x, cond, cam


def golden_guided_step(ref):
    """The reference's OWN `p_sample_with_grad` (gaussian_diffusion_posenet.py:436-480) for grad_type 'prox' (2-D term
    3e5 then skating 1e5, t[0] <= 100) and 'amass' (skating 3e6, t[0] <= 50), one call per (grad_type, t) from seeded
    inputs: full-size reference PoseNet, the reference's guide_*_with_smpl hooks around the oracle body model, the
    reference's SpacedDiffusionPoseNet tables, `th.randn_like` noise from `torch.manual_seed(noise_seed)`."""
    from oracle import geometry as G
    body = G.BodyModel(synth.synthetic_smplx_tensors(0))
    refload.set_body_model(body)
    seeds = dict(stats_seed=0, x_seed=61, xn_seed=62, cond_seed=63, cam_seed=2, weight_seed=13, body_seed=0)
    mean, std, x, cond, cam = guided_step_inputs(seeds)

    class GDS:
        pose_feat_dim, traj_feat_dim, joints_num = 272, 22, 22
        Mean, Std = mean, std
        cam_R = torch.tensor(synth.SYNTH_CAM_R)
        cam_t = torch.tensor(synth.SYNTH_CAM_T)
    net = ref.posenet.PoseNet(GDS(), 294, latent_dim=512, ff_size=1024, num_layers=8, num_heads=4, traj_feat_dim=22,
                              device='cpu').eval()
    net.smplx_model = body
    sd = synth.posenet_state_dict(seeds['weight_seed'])
    net.load_state_dict(sd, strict=False)
    diff = ref.model_util.create_gaussian_diffusion(_Args, ref.gd_posenet, ref.respace.SpacedDiffusionPoseNet,
                                                    1000, '', device='cpu')
    out = {}
    for k, (gt, i) in enumerate(GUIDED_CASES):
        batch = dict(cam)
        batch['cond'] = cond
        t = torch.tensor([i] * 2)
        torch.manual_seed(700 + k)
        with torch.no_grad():
            r = diff.p_sample_with_grad(net, batch, x.clone(), t, clip_denoised=False, grad_type=gt)
        # which hooks were live (a 0-d return = "no active constraint"), recorded for the tests' sanity asserts
        with torch.no_grad():
            o = {'pred_xstart': r['pred_xstart']}
            live = [float(net.guide_skating_with_smpl(batch, o, t, compute_grad='x_0').dim() != 0)]
        out[f'case{k}_sample'] = r['sample'].numpy()
        out[f'case{k}_t'], out[f'case{k}_noise_seed'], out[f'case{k}_skating_live'] = i, 700 + k, live[0]
        out[f'case{k}_grad_type'] = gt
        if gt == 'prox' and i in (100, 0):
            out[f'case{k}_pred_xstart'] = r['pred_xstart'].numpy()
        print('guided_step', gt, i, 'max|sample|', float(r['sample'].abs().max()), 'skating live', live[0])
    np.savez_compressed(os.path.join(OUT, 'guided_step.npz'), n_cases=len(GUIDED_CASES), **seeds, **out)
    print('guided_step.npz', os.path.getsize(os.path.join(OUT, 'guided_step.npz')))


GUIDED_HEAD_T = (103, 102, 101, 100, 99, 98, 97)


def golden_guided_head(ref):
    """A free-running GUIDED run at the reference's own weights over the stretch where it is still well conditioned: the
    reference's `p_sample_with_grad(grad_type='prox')` called step after step on ITS OWN samples for t = 103 .. 97 (three
    un-guided steps, then the first four guided ones: 2-D term x 3e5 + skating term x 1e5), noise from one seed per step.
    scripts/guided_chaos.py / profiles/r3_guided_chaos.txt show why it stops there: after t = 96 the reference and its
    restatement drift apart by more than 1e-3 and the samples leave |x| ~ 50."""
    from oracle import geometry as G
    body = G.BodyModel(synth.synthetic_smplx_tensors(0))
    refload.set_body_model(body)
    seeds = dict(stats_seed=0, x_seed=61, xn_seed=62, cond_seed=63, cam_seed=2, weight_seed=13, body_seed=0)
    mean, std, x, cond, cam = guided_step_inputs(seeds)

    class GDS:
        pose_feat_dim, traj_feat_dim, joints_num = 272, 22, 22
        Mean, Std = mean, std
        cam_R = torch.tensor(synth.SYNTH_CAM_R)
        cam_t = torch.tensor(synth.SYNTH_CAM_T)
    net = ref.posenet.PoseNet(GDS(), 294, latent_dim=512, ff_size=1024, num_layers=8, num_heads=4, traj_feat_dim=22,
                              device='cpu').eval()
    net.smplx_model = body
    net.load_state_dict(synth.posenet_state_dict(seeds['weight_seed']), strict=False)
    diff = ref.model_util.create_gaussian_diffusion(_Args, ref.gd_posenet, ref.respace.SpacedDiffusionPoseNet,
                                                    1000, '', device='cpu')
    xx = x.clone()
    out = {}
    for k, i in enumerate(GUIDED_HEAD_T):
        batch = dict(cam)
        batch['cond'] = cond
        torch.manual_seed(900 + k)
        with torch.no_grad():
            r = diff.p_sample_with_grad(net, batch, xx.clone(), torch.tensor([i] * 2), clip_denoised=False, grad_type='prox')
        xx = r['sample']
        out[f'max_abs_{k}'] = float(xx.abs().max())
        print('guided_head t', i, 'max|sample|', out[f'max_abs_{k}'])
    np.savez_compressed(os.path.join(OUT, 'guided_head.npz'), n_steps=len(GUIDED_HEAD_T), t=np.asarray(GUIDED_HEAD_T), noise_seed0=900,
                        sample=xx.numpy(), **seeds, **out)
    print('guided_head.npz', os.path.getsize(os.path.join(OUT, 'guided_head.npz')))


SCHEME_CASES = (
    ('amass', dict(mask_scheme='lower')),
    ('amass', dict(mask_scheme='upper', iter2_cond_noisy_pose=False, iter2_cond_noisy_traj=False)),
    ('amass', dict(mask_scheme='full', full_seed=5)),
    ('amass', dict(mask_scheme='full', infill_traj=True)),
    ('amass', dict(mask_scheme='lower', input_noise=False)),
    ('amass', dict(mask_scheme='full', input_noise=False, infill_traj=True, sample_iter=3)),
    ('amass', dict(mask_scheme='lower', repr_abs_only=False)),
    ('prox', dict(sample_iter=3)),
    ('prox', dict(sample_iter=2, iter2_cond_noisy_pose=True, iter2_cond_noisy_traj=True)),
)


def scheme_case(kind, kw, B=2):
    """Seeded inputs, pre-made stage outputs and the args namespace of one driver-loop case (shared by the generator
    and the tests).  Stage outputs are pre-made tensors: the glue under test is everything BETWEEN the samplers."""
    import types
    a = dict(sample_iter=2, repr_abs_only=True, infill_traj=False, traj_mask_ratio=0.1, mask_scheme='lower',
             input_noise=True, iter2_cond_noisy_traj=(kind == 'amass'), iter2_cond_noisy_pose=(kind == 'amass'),
             early_stop=(kind == 'prox'), cond_fn_with_grad=True, timestep_respacing_eval='', full_seed=None)
    a.update(kw)
    args = types.SimpleNamespace(**a)
    tfd = 13 if args.repr_abs_only else 22
    abs_ch = [0, 2, 3, 6, 7, 8, 9, 10, 11, 12, 16, 17, 18]
    body_t = synth.synthetic_smplx_tensors(0)
    s_traj, s_pose = synth.synthetic_stats(0), synth.synthetic_stats(1)
    f32 = lambda seed, shape: torch.from_numpy(np.random.Generator(np.random.PCG64(seed)).standard_normal(shape).astype(np.float32))
    clean_t = synth.walking_motion(50, B, 144, *s_traj, body_t)
    noisy_t = clean_t + 0.05 * f32(51, clean_t.shape)
    clean_p = synth.walking_motion(50, B, 144, *s_pose, body_t)
    noisy_p = clean_p + 0.05 * f32(52, clean_p.shape)
    sel = abs_ch if args.repr_abs_only else list(range(22))
    bt = {'cond': noisy_t[:, :, sel].contiguous(), 'motion_repr_clean': clean_t, 'motion_repr_noisy': noisy_t}
    bp = {'motion_repr_clean': clean_p, 'motion_repr_noisy': noisy_p}
    if kind == 'prox':
        g = np.random.Generator(np.random.PCG64(53))
        jv = (g.uniform(size=(B, 145, 22)) < 0.8).astype(np.float32)
        vec = np.ones((B, 145, 294), np.float32)
        for j in range(22):
            vec[:, :, 22 + 3 * j:25 + 3 * j] = jv[:, :, j:j + 1]
            vec[:, :, 88 + 3 * j:91 + 3 * j] = jv[:, :, j:j + 1]
            if j > 0:
                vec[:, :, 154 + 6 * (j - 1):160 + 6 * (j - 1)] = jv[:, :, j:j + 1]
        bp['mask_joint_vis'], bp['mask_vec_vis'] = torch.from_numpy(jv), torch.from_numpy(vec)
    n = args.sample_iter
    traj_out = [synth.walking_motion(60 + i, B, 144, *s_traj, body_t)[:, :, sel].contiguous() for i in range(n)]
    pose_out = [synth.walking_motion(70 + i, B, 143, *s_pose, body_t).permute(0, 2, 1).unsqueeze(2).contiguous() for i in range(n)]
    return args, tfd, body_t, s_traj, s_pose, bt, bp, traj_out, pose_out


def digest(t, n_proj=6):
    """Shape + seeded random projections of a tensor (float64): pins every element to ~1e-7 relative at 6 numbers."""
    x = np.asarray(t.detach().cpu().double().numpy() if torch.is_tensor(t) else t, dtype=np.float64).reshape(-1)
    g = np.random.Generator(np.random.PCG64(x.size % 9973 + 17))
    return np.concatenate([[x.size, x.sum(), np.abs(x).sum()], g.standard_normal((n_proj, x.size)) @ x])


def exact_hash(t):
    """sha256 of the float32 bytes (with -0.0 folded onto +0.0): for regions that are bit-exact COPIES / masks of the inputs
    on every platform (channels 22.. of PoseNet's cond, control_cond, the un-touched channels of motion_repr_noisy)."""
    import hashlib
    x = np.ascontiguousarray((t.detach().cpu().float() + 0.0).numpy() if torch.is_tensor(t) else np.asarray(t, np.float32) + 0.0)
    return hashlib.sha256(x.tobytes()).hexdigest()


def scheme_full_entries(prefix, key, v):
    """What scheme.npz stores per tensor besides its digest (a digest is blind to one slightly wrong
    element): COMPUTED parts in full (TrajNet's cond; channels 0..21 of PoseNet's cond = the re-derived trajectory), COPIED
    parts as an exact hash."""
    out = {}
    v = v.detach().cpu()
    if key == 'control_cond':
        out[prefix + '_sha'] = exact_hash(v)
    elif v.dim() == 4:                                   # PoseNet cond [bs, 294, 1, T]
        out[prefix + '_full_traj'] = v[:, 0:22].numpy().astype(np.float32)
        out[prefix + '_sha_rest'] = exact_hash(v[:, 22:])
    else:                                                # TrajNet cond [bs, T, tfd]
        out[prefix + '_full'] = v.numpy().astype(np.float32)
    return out


def golden_scheme(ref):
    """The drivers' inference-iteration loops, EXECUTED from the reference scripts' own text (test_amass_full.py:217-384,
    test_prox_egobody.py:214-324; the scripts cannot be imported -- configargparse / datasets / checkpoints at module
    level) with stub samplers that return pre-made outputs and log what each stage was handed.  Stored: digests of
    every stage input, of traj_rec_full per iteration and of the dict entries the scripts rely on afterwards."""
    import textwrap
    import types
    from oracle import geometry as G
    body = G.BodyModel(synth.synthetic_smplx_tensors(0))
    src = {'amass': (open(os.path.join(refload.REF_ROOT, 'test_amass_full.py')).read().split('\n'), 216, 384),
           'prox': (open(os.path.join(refload.REF_ROOT, 'test_prox_egobody.py')).read().split('\n'), 213, 324)}
    out = {}
    for ci, (kind, kw) in enumerate(SCHEME_CASES):
        args, tfd, body_t, s_traj, s_pose, bt, bp, traj_out, pose_out = scheme_case(kind, kw)
        lines, lo, hi = src[kind]
        block = textwrap.dedent('\n'.join(lines[lo:hi]))
        log = []

        class Stub:
            def __init__(self, outs, name):
                self.outs, self.name = list(outs), name

            def eval_losses(self, model=None, batch=None, shape=None, **kw2):
                o = self.outs.pop(0)
                assert list(o.shape) == list(shape), (self.name, o.shape, shape)
                log.append((self.name, {k: batch[k].detach().clone() for k in ('cond', 'control_cond') if k in batch},
                            {k: kw2.get(k) for k in ('grad_type', 'early_stop', 'cond_fn_with_grad')}))
                return None, o
        tds = types.SimpleNamespace(traj_feat_dim=tfd, pose_feat_dim=272, Mean=s_traj[0], Std=s_traj[1])
        pds = types.SimpleNamespace(traj_feat_dim=22, pose_feat_dim=272, Mean=s_pose[0], Std=s_pose[1])
        ns = {'np': np, 'torch': torch, 'print': lambda *a, **k: None, 'args': args,
              'dist_util': types.SimpleNamespace(dev=lambda: torch.device('cpu')),
              'test_batch_traj': {k: v.clone() for k, v in bt.items()}, 'test_batch_pose': {k: v.clone() for k, v in bp.items()},
              'test_traj_dataset': tds, 'test_pose_dataset': pds,
              'diffusion_trajnet_eval': Stub(traj_out[:1], 'traj'), 'diffusion_trajnet_control_eval': Stub(traj_out[1:], 'traj'),
              'diffusion_posenet_eval': Stub(pose_out, 'pose'),
              'model_trajnet': None, 'model_trajnet_control': None, 'model_posenet': None, 'smplx_neutral': body,
              'REPR_LIST': ref.other_utils.REPR_LIST, 'REPR_DIM_DICT': ref.other_utils.REPR_DIM_DICT,
              'recover_from_repr_smpl': ref.motion_repr.recover_from_repr_smpl, 'get_repr_smplx': ref.motion_repr.get_repr_smplx,
              'rot6d_to_rotmat': ref.quaternion.rot6d_to_rotmat, 'rotation_matrix_to_angle_axis': ref.konia.rotation_matrix_to_angle_axis}
        if args.full_seed is not None:
            torch.manual_seed(args.full_seed)
        exec(compile(block, f'{kind}[{lo + 1}:{hi}]', 'exec'), ns)
        pre = f'case{ci}_'
        out[pre + 'n_calls'] = len(log)
        for k, (name, tens, kws) in enumerate(log):
            out[pre + f'call{k}_name'] = name
            out[pre + f'call{k}_kw'] = repr(sorted(kws.items()))
            for kk, v in tens.items():
                out[pre + f'call{k}_{kk}'] = digest(v)
                out[pre + f'call{k}_{kk}_shape'] = np.asarray(v.shape)
                out.update(scheme_full_entries(pre + f'call{k}_{kk}', kk, v))
        out[pre + 'traj_rec_full'] = digest(ns['traj_rec_full'])
        out[pre + 'traj_rec_full_full'] = ns['traj_rec_full'].numpy()            # float64, as the script holds it
        if args.full_seed is not None:
            # the script draws `torch.FloatTensor(bs).uniform_(0, clip_len - 1).long()` once per masked iteration and
            # nothing else touches the generator: replay the draws
            torch.manual_seed(args.full_seed)
            n_draw = args.sample_iter if args.iter2_cond_noisy_pose else 1
            # clip_len = motion_repr_clean.shape[1]: 143 in iteration 0, 294 once the tensor is [bs, 294, 1, T] (:335, :375)
            starts = np.stack([torch.FloatTensor(2).uniform_(0, (143 if k == 0 else 294) - 1).long().numpy() for k in range(n_draw)])
            assert np.array_equal(starts[-1], ns['start'].numpy()), (starts, ns['start'])
            out[pre + 'full_mask_start'] = starts
        tb, pb = ns['test_batch_traj'], ns['test_batch_pose']
        out[pre + 'after_traj_noisy'] = digest(tb['motion_repr_noisy'])
        out[pre + 'after_traj_cond'] = digest(tb['cond'])
        out[pre + 'after_traj_cond_full'] = tb['cond'].numpy().astype(np.float32)
        out[pre + 'after_traj_noisy_sha'] = exact_hash(tb['motion_repr_noisy'])
        out[pre + 'after_pose_noisy_shape'] = np.asarray(pb['motion_repr_noisy'].shape)
        out[pre + 'after_pose_clean_shape'] = np.asarray(pb['motion_repr_clean'].shape)
        print('scheme case', ci, kind, kw, [n for n, _, _ in log])
    np.savez_compressed(os.path.join(OUT, 'scheme.npz'), n_cases=len(SCHEME_CASES), **out)
    print('scheme.npz', os.path.getsize(os.path.join(OUT, 'scheme.npz')))


SCHEME_REAL_HEAD_T = (103, 102, 101, 100, 99)
# (driver, args overrides, PoseNet stage): an int = the reference's own sampler on its own N-step cosine schedule
# (free-running, un-guided: chaos is no excuse there); 'head' = the reference's own p_sample_with_grad over the guided head
# t = 103 .. 99 of the 1000-step schedule at the reference's weights (3e5 / 1e5): three un-guided steps, then the first two
# guided ones (2-D term + skating term each).  The stage starts from th.randn like every PoseNet stage; from that start the reference and its fp32
# restatement are 3e-4 / 4e-3 / 6e-2 / 5e-1 apart in x after t = 100 / 99 / 98 / 97 (|x| jumps from 4 to 43 at the first
# guided step with B = 2: the 2-D term is a batch MEAN of L1 terms, its per-clip gradient scales as 1 / B) and the returned
# pred_xstart 7e-6 apart at t = 99, 1e-4 at t = 98, 4e-3 at t = 97 -- measured in the build container
# (profiles/r4_scheme_head_chaos.txt); inside the scheme the second head stage multiplies what the first one left.
SCHEME_REAL_CASES = (
    ('amass', dict(mask_scheme='lower', cond_fn_with_grad=False), 50),                 # BASELINE configs[2] shape
    ('prox', dict(sample_iter=3, cond_fn_with_grad=False, early_stop=True), 40),       # configs[4] shape, un-guided
    ('prox', dict(sample_iter=2, cond_fn_with_grad=True, early_stop=True), 'head'),    # configs[3]/[4] guidance
    # the REAL step counts of the drivers (cfg_files/test_cfg/*.yaml: diffusion_steps_posenet 1000, diffusion_steps_trajnet 100), un-guided:
    ('amass', dict(mask_scheme='lower', cond_fn_with_grad=False), 1000),               # configs[2]: 100 + 1000 + 100 + 1000 steps
    ('prox', dict(sample_iter=3, cond_fn_with_grad=False, early_stop=True), 1000),     # configs[4]: 3 x (100 + 980) steps
)
SCHEME_REAL_SEEDS = dict(trajnet=71, control=72, posenet=73, noise=3100)
# dataset.cam_t of the guided case: the camera 12 m behind the canonical origin along its axis -- the trajectory this PoseNet
# stage is conditioned on is the output of a random-weight TrajNet, so the body may stand anywhere within a few metres of the
# origin; with synth.SYNTH_CAM_T it crosses the camera plane and the pinhole division sends the first guided step to |x| ~ 1e8
SCHEME_REAL_CAM_T = [[0.1, -0.2, -12.0]]


def scheme_real_case(ci, B=2):
    """Inputs of one free-running real-network case (shared by the generator and the tests)."""
    kind, kw, pose_steps = SCHEME_REAL_CASES[ci]
    args, tfd, body_t, s_traj, s_pose, bt, bp, _, _ = scheme_case(kind, kw, B=B)
    cam = synth.synthetic_camera_batch(4, B) if pose_steps == 'head' else {}
    # order and length of the reference's draws from the global generator: per stage one randn(*shape) + one randn_like
    # per step (gaussian_diffusion_posenet.py:613,458)
    n_pose = len(SCHEME_REAL_HEAD_T) if pose_steps == 'head' else (min(pose_steps, 980) if args.early_stop else pose_steps)      # early_stop: indices[0:980]
    plan = []
    for it in range(args.sample_iter):
        plan.append(('traj', (B, 144, tfd), 100))
        plan.append(('pose', (B, 294, 1, 143), n_pose))
    return args, tfd, body_t, s_traj, s_pose, bt, bp, cam, pose_steps, plan


def golden_scheme_real(ref):
    """The drivers' inference-iteration loops, EXECUTED from the reference scripts' own text (test_amass_full.py:217-384,
    test_prox_egobody.py:214-324) with the REFERENCE'S OWN networks and samplers, free-running end to end on CPU:
    TrajNet (100 steps) -> the script's host re-derivation -> PoseNet -> TrajControl (100 steps) -> PoseNet [-> ...], B = 2,
    noise from one `torch.manual_seed`.  Stored: the final tensors, traj_rec_full and every stage's output."""
    import textwrap
    import types
    from oracle import geometry as G
    body = G.BodyModel(synth.synthetic_smplx_tensors(0))
    refload.set_body_model(body)
    src = {'amass': (open(os.path.join(refload.REF_ROOT, 'test_amass_full.py')).read().split('\n'), 216, 384),
           'prox': (open(os.path.join(refload.REF_ROOT, 'test_prox_egobody.py')).read().split('\n'), 213, 324)}
    sd_t = synth.trajnet_state_dict(SCHEME_REAL_SEEDS['trajnet'], trajcontrol=False)
    sd_c = synth.trajnet_state_dict(SCHEME_REAL_SEEDS['control'], trajcontrol=True)
    sd_p = synth.posenet_state_dict(SCHEME_REAL_SEEDS['posenet'])
    out = {}
    for ci, (kind, kw, pose_steps) in enumerate(SCHEME_REAL_CASES):
        args, tfd, body_t, s_traj, s_pose, bt, bp, cam, _, plan = scheme_real_case(ci)
        tds = types.SimpleNamespace(traj_feat_dim=tfd, pose_feat_dim=272, Mean=s_traj[0], Std=s_traj[1])
        pds = types.SimpleNamespace(traj_feat_dim=22, pose_feat_dim=272, joints_num=22, Mean=s_pose[0], Std=s_pose[1],
                                    cam_R=torch.tensor(synth.SYNTH_CAM_R), cam_t=torch.tensor(SCHEME_REAL_CAM_T))
        tn = ref.trajnet.TrajNet(time_dim=32, mid_dim=512, cond_dim=tfd, traj_feat_dim=tfd, trajcontrol=False).eval()
        tn.load_state_dict(sd_t, strict=True)
        cn = ref.trajnet.TrajNet(time_dim=32, mid_dim=512, cond_dim=tfd, traj_feat_dim=tfd, trajcontrol=True).eval()
        cn.load_state_dict(sd_c, strict=True)
        pn = ref.posenet.PoseNet(pds, 294, latent_dim=512, ff_size=1024, num_layers=8, num_heads=4, traj_feat_dim=22,
                                 device='cpu').eval()
        pn.smplx_model = body
        pn.load_state_dict(sd_p, strict=False)
        mk = ref.model_util.create_gaussian_diffusion
        d_t = mk(_Args, ref.gd_trajnet, ref.respace.SpacedDiffusionTrajNet, 100, '', device='cpu')
        d_c = mk(_Args, ref.gd_trajnet, ref.respace.SpacedDiffusionTrajNet, 100, '', device='cpu')
        stage_out = []

        class Rec:           # the reference's sampler, with its result logged
            def __init__(self, d, name):
                self.d, self.name = d, name

            def eval_losses(self, **kw2):
                kw2['progress'] = False
                r = self.d.eval_losses(**kw2)
                stage_out.append((self.name, r[1].detach().clone()))
                return r

        class Head:          # the reference's own guided step, called step after step over the stable head
            def __init__(self, d):
                self.d = d

            def eval_losses(self, model=None, batch=None, shape=None, clip_denoised=False, cond_fn_with_grad=False,
                            grad_type=None, early_stop=False, **kw2):
                assert cond_fn_with_grad and grad_type == 'prox'
                img = torch.randn(*shape)
                for i in SCHEME_REAL_HEAD_T:
                    with torch.no_grad():
                        o = self.d.p_sample_with_grad(model, batch, img, torch.tensor([i] * shape[0]),
                                                      clip_denoised=clip_denoised, grad_type=grad_type)
                    img = o['sample']
                res = o['pred_xstart'] if early_stop else o['sample']      # p_sample_loop's return (:568-571)
                stage_out.append(('pose', res.detach().clone()))
                return None, res
        if pose_steps == 'head':
            d_p = Head(mk(_Args, ref.gd_posenet, ref.respace.SpacedDiffusionPoseNet, 1000, '', device='cpu'))
        else:
            d_p = Rec(mk(_Args, ref.gd_posenet, ref.respace.SpacedDiffusionPoseNet, pose_steps, '', device='cpu'), 'pose')
        tbp = {k: v.clone() for k, v in bp.items()}
        tbp.update({k: v.clone() for k, v in cam.items()})
        lines, lo, hi = src[kind]
        block = textwrap.dedent('\n'.join(lines[lo:hi]))
        ns = {'np': np, 'torch': torch, 'print': lambda *a, **k: None, 'args': args,
              'dist_util': types.SimpleNamespace(dev=lambda: torch.device('cpu')),
              'test_batch_traj': {k: v.clone() for k, v in bt.items()}, 'test_batch_pose': tbp,
              'test_traj_dataset': tds, 'test_pose_dataset': pds,
              'diffusion_trajnet_eval': Rec(d_t, 'traj'), 'diffusion_trajnet_control_eval': Rec(d_c, 'traj'),
              'diffusion_posenet_eval': d_p,
              'model_trajnet': tn, 'model_trajnet_control': cn, 'model_posenet': pn, 'smplx_neutral': body,
              'REPR_LIST': ref.other_utils.REPR_LIST, 'REPR_DIM_DICT': ref.other_utils.REPR_DIM_DICT,
              'recover_from_repr_smpl': ref.motion_repr.recover_from_repr_smpl, 'get_repr_smplx': ref.motion_repr.get_repr_smplx,
              'rot6d_to_rotmat': ref.quaternion.rot6d_to_rotmat, 'rotation_matrix_to_angle_axis': ref.konia.rotation_matrix_to_angle_axis}
        import time
        t0 = time.time()
        torch.manual_seed(SCHEME_REAL_SEEDS['noise'] + ci)
        exec(compile(block, f'{kind}[{lo + 1}:{hi}]', 'exec'), ns)
        pre = f'case{ci}_'
        final_pose = ns['val_output_pose' if kind == 'amass' else 'val_output_joint']
        out[pre + 'pose'] = final_pose.numpy()
        out[pre + 'traj'] = ns['val_output_traj'].numpy()
        out[pre + 'traj_rec_full'] = ns['traj_rec_full'].numpy()
        out[pre + 'n_stages'] = len(stage_out)
        for k, (name, v) in enumerate(stage_out):
            out[pre + f'stage{k}_name'] = name
            if pose_steps != 1000 or name == 'traj':      # the long cases keep the (small) trajectory outputs only
                out[pre + f'stage{k}_out'] = v.numpy()
        assert [n for n, _ in stage_out] == [p[0] for p in plan]
        print('scheme_real case', ci, kind, kw, pose_steps, [n for n, _ in stage_out], 'max|pose|', float(final_pose.abs().max()),
              'max|traj|', float(ns['val_output_traj'].abs().max()), f'{time.time() - t0:.1f}s')
    np.savez_compressed(os.path.join(OUT, 'scheme_real.npz'), n_cases=len(SCHEME_REAL_CASES),
                        **{k + '_seed': v for k, v in SCHEME_REAL_SEEDS.items()}, **out)
    print('scheme_real.npz', os.path.getsize(os.path.join(OUT, 'scheme_real.npz')))


def frames_inputs(seed=0, N=40):
    """Seeded per-frame SMPL-X parameters + a rigid cam2world (float32, as the pickles / json of the datasets hold)."""
    g = np.random.Generator(np.random.PCG64(900 + seed))
    f = lambda *sh, s=1.0: (g.standard_normal(sh) * s).astype(np.float32)
    params = {'transl': f(N, 3) + np.array([0, 0, 3], np.float32), 'global_orient': f(N, 3, s=1.2),
              'betas': f(N, 10), 'body_pose': f(N, 63, s=0.4)}
    params['global_orient'][0] = 0.0                         # identity and tiny rotations (scipy's small-angle branches)
    params['global_orient'][1] = np.array([1e-5, -2e-5, 3e-5], np.float32)
    params['global_orient'][2] = np.array([3.1, 0.2, -0.1], np.float32)          # close to pi
    c2w = np.eye(4, dtype=np.float32)
    c2w[:3, :3] = synth._rodrigues_np(np.array([[0.9, -1.7, 0.4]]))[0].astype(np.float32)
    c2w[:3, 3] = np.array([0.3, -1.2, 2.5], np.float32)
    return params, c2w


def golden_frames(ref):
    """The per-frame dataset statements of data_loaders/dataloader_video.py:121-142, one frame at a time as the loader
    runs them, through the reference's OWN `update_globalRT_for_smplx` (utils/other_utils.py:189-240) and the oracle body
    model in place of smplx."""
    from oracle import geometry as G
    body = G.BodyModel(synth.synthetic_smplx_tensors(0))
    params, cam2world_np = frames_inputs()
    cam2world = torch.from_numpy(cam2world_np).float()
    cam_R, cam_t = cam2world[:3, :3].reshape([3, 3]), cam2world[:3, 3].reshape([1, 3])
    joints_world, smplx_world = [], []
    for i in range(len(params['transl'])):
        param = {k: v[i:i + 1] for k, v in params.items()}
        torch_param = {k: torch.tensor(v) for k, v in param.items()}
        smpl_output = body(return_verts=True, **torch_param)
        joints_cam = smpl_output.joints[:, 0:22, :]
        joints = torch.matmul(cam_R, joints_cam.permute(0, 2, 1)).permute(0, 2, 1) + cam_t
        joints_world.append(joints[0].detach().numpy())
        d = ref.other_utils.update_globalRT_for_smplx(dict(param), cam2world.detach().cpu().numpy(),
                                                      delta_T=joints_cam[:, 0].detach().cpu().numpy() - param['transl'])
        smplx_world.append(np.concatenate([d['global_orient'], d['transl'], d['betas'], d['body_pose']], axis=-1)[0])
    np.savez_compressed(os.path.join(OUT, 'frames.npz'), seed=0, body_seed=0, joints_world=np.asarray(joints_world),
                        smplx_world=np.asarray(smplx_world))
    print('frames.npz', os.path.getsize(os.path.join(OUT, 'frames.npz')), np.asarray(smplx_world).dtype)


def golden_control_loop(ref):
    """TrajControl (TrajNet + ControlNet, randomised zero-convs) through the reference's own 100-step sampler: the stage
    every inference iteration >= 1 runs (test_amass_full.py:252-266), B = 2 clips."""
    sdt = synth.trajnet_state_dict(seed=23, trajcontrol=True)
    tn = ref.trajnet.TrajNet(time_dim=32, mid_dim=512, cond_dim=13, traj_feat_dim=13, trajcontrol=True).eval()
    tn.load_state_dict(sdt, strict=True)
    d = ref.model_util.create_gaussian_diffusion(_Args, ref.gd_trajnet, ref.respace.SpacedDiffusionTrajNet, 100, '', device='cpu')
    batch = {'cond': seeded(214, 2, 144, 13), 'control_cond': seeded(215, 2, 144, 272)}
    torch.manual_seed(4322)
    with torch.no_grad():
        _, y = d.eval_losses(model=tn, batch=batch, shape=[2, 144, 13], progress=False, clip_denoised=False,
                             timestep_respacing='', cond_fn_with_grad=True, compute_loss=False)
    np.savez_compressed(os.path.join(OUT, 'trajnet_control_loop100.npz'), weight_seed=23, cond_seed=214, control_seed=215,
                        torch_seed=4322, steps=100, y=y.numpy())
    print('trajnet_control_loop100.npz', os.path.getsize(os.path.join(OUT, 'trajnet_control_loop100.npz')))


def golden_posenet_loop1000(ref):
    """The headline configuration end to end through the REFERENCE's own code: full-size PoseNet, the 1000-step cosine
    schedule, `eval_losses` (p_sample_loop, no guidance), one clip, CPU generator noise from a seed."""
    sd = synth.posenet_state_dict(seed=17)
    net = ref.posenet.PoseNet(_DS(), 294, latent_dim=512, ff_size=1024, num_layers=8, num_heads=4, traj_feat_dim=22,
                              device='cpu').eval()
    net.load_state_dict(sd, strict=True)
    diff = ref.model_util.create_gaussian_diffusion(_Args, ref.gd_posenet, ref.respace.SpacedDiffusionPoseNet, 1000, '',
                                                    device='cpu')
    mean, std = synth.synthetic_stats(1)
    cond = synth.plausible_motion(8, 1, 143, mean, std)
    torch.manual_seed(2024)
    with torch.no_grad():
        _, y = diff.eval_losses(model=net, batch={'cond': cond}, shape=[1, 294, 1, 143], progress=False, clip_denoised=False,
                                timestep_respacing='', cond_fn_with_grad=False, compute_loss=False)
    np.savez_compressed(os.path.join(OUT, 'posenet_loop1000.npz'), weight_seed=17, stats_seed=1, cond_seed=8, torch_seed=2024,
                        steps=1000, y=y.numpy())
    print('posenet_loop1000.npz', os.path.getsize(os.path.join(OUT, 'posenet_loop1000.npz')), float(y.abs().max()))


def golden_metrics():
    """Run the reference's own metric statements (eval_amass_full.py:67-148, read from its file) on synthetic
    results.  The script cannot be imported (argparse / smplx / open3d at module level), the block can be executed."""
    import textwrap
    import types
    from oracle import metrics as M
    src = open(os.path.join(refload.REF_ROOT, 'eval_amass_full.py')).read().split('\n')
    block = textwrap.dedent('\n'.join(src[66:148]))
    out = {}
    for scheme, ratio in (('lower', 0.0), ('full', 0.1)):
        clean, rec, r_clean, r_rec = M.synthetic_results(5)
        ns = {'np': np, 'args': types.SimpleNamespace(mask_scheme=scheme, traj_mask_ratio=ratio),
              'rec_ric_data_clean_list': clean, 'rec_ric_data_rec_list_from_smpl': rec,
              'motion_repr_rec_list': r_rec.copy(), 'motion_repr_clean_list': r_clean.copy(), 'print': lambda *a, **k: None}
        exec(compile(block, 'eval_amass_full.py[67:148]', 'exec'), ns)
        vals = {'mpjpe_global': np.mean(ns['joints_mpjpe_global']), 'mpjpe_global_vis': np.mean(ns['joints_mpjpe_global_vis']),
                'mpjpe_global_occ': np.mean(ns['joints_mpjpe_global_invis']), 'contact_lbl_acc': np.mean(ns['contact_lbl_acc']),
                'skating_gt_ratio': ns['skating_gt_ratio'], 'skating_rec_ratio': ns['skating_rec_ratio'],
                'accel_error': ns['acc_error'], 'ground_pene_freq': ns['pene_freq'], 'ground_pene_dist': ns['pene_dist']}
        for k, v in vals.items():
            out[f'{scheme}_{k}'] = np.float64(v)
    np.savez_compressed(os.path.join(OUT, 'metrics.npz'), results_seed=5, **out)
    print({k: float(v) for k, v in out.items()})


def main():
    if sys.argv[1:] == ['metrics']:
        return golden_metrics()
    if sys.argv[1:] == ['eval_losses']:
        warnings.filterwarnings('ignore')
        return golden_eval_losses(refload.load())
    if sys.argv[1:] == ['rel']:
        warnings.filterwarnings('ignore')
        return golden_rel(refload.load())
    if sys.argv[1:] == ['control_loop']:
        warnings.filterwarnings('ignore')
        return golden_control_loop(refload.load())
    if sys.argv[1:] == ['posenet_loop1000']:
        warnings.filterwarnings('ignore')
        torch.set_num_threads(8)
        return golden_posenet_loop1000(refload.load())
    if sys.argv[1:] == ['frames']:
        warnings.filterwarnings('ignore')
        return golden_frames(refload.load())
    if sys.argv[1:] == ['scheme']:
        warnings.filterwarnings('ignore')
        return golden_scheme(refload.load())
    if sys.argv[1:] == ['scheme_real']:
        warnings.filterwarnings('ignore')
        torch.set_num_threads(8)
        return golden_scheme_real(refload.load())
    if sys.argv[1:] == ['guided_step']:
        warnings.filterwarnings('ignore')
        return golden_guided_step(refload.load())
    if sys.argv[1:] == ['guided_head']:
        warnings.filterwarnings('ignore')
        return golden_guided_head(refload.load())
    if sys.argv[1:] == ['ddim']:
        warnings.filterwarnings('ignore')
        return golden_ddim(refload.load())
    if sys.argv[1:] == ['rederive']:
        warnings.filterwarnings('ignore')
        return golden_rederive(refload.load())
    warnings.filterwarnings('ignore')
    os.makedirs(OUT, exist_ok=True)
    ref = refload.load()
    torch.set_num_threads(8)

    # ---- PoseNet forward, full config, B=2 -------------------------------------------------------
    sd = synth.posenet_state_dict(seed=11)
    net = ref.posenet.PoseNet(_DS(), 294, latent_dim=512, ff_size=1024, num_layers=8, num_heads=4,
                              traj_feat_dim=22, device='cpu').eval()
    net.load_state_dict(sd, strict=True)
    x, c = seeded(101, 2, 294, 1, 143), seeded(102, 2, 294, 1, 143)
    t = torch.tensor([999, 7])
    with torch.no_grad():
        y = net({'x_t': x, 'cond': c}, t)
    np.savez_compressed(os.path.join(OUT, 'posenet_forward.npz'), weight_seed=11, x_seed=101, cond_seed=102,
                        t=t.numpy(), y=y.numpy())

    # ---- PoseNet 8-step DDPM loop (reference SpacedDiffusionPoseNet, CPU generator noise) ----------
    diff = ref.model_util.create_gaussian_diffusion(_Args, ref.gd_posenet, ref.respace.SpacedDiffusionPoseNet,
                                                    8, '', device='cpu')
    batch = {'cond': seeded(103, 2, 294, 1, 143)}
    torch.manual_seed(1234)
    with torch.no_grad():
        _, y = diff.eval_losses(model=net, batch=batch, shape=[2, 294, 1, 143], progress=False,
                                clip_denoised=False, timestep_respacing='', cond_fn_with_grad=False,
                                compute_loss=False)
    np.savez_compressed(os.path.join(OUT, 'posenet_loop8.npz'), weight_seed=11, cond_seed=103, torch_seed=1234,
                        steps=8, y=y.numpy())

    # ---- TrajNet forward, vanilla and TrajControl ---------------------------------------------------
    for ctrl in (False, True):
        sdt = synth.trajnet_state_dict(seed=21 + ctrl, trajcontrol=ctrl)
        tn = ref.trajnet.TrajNet(time_dim=32, mid_dim=512, cond_dim=13, traj_feat_dim=13, trajcontrol=ctrl).eval()
        tn.load_state_dict(sdt, strict=True)
        x, c, cc = seeded(201, 2, 144, 13), seeded(202, 2, 144, 13), seeded(203, 2, 144, 272)
        t = torch.tensor([99, 3])
        with torch.no_grad():
            y = tn({'x_t': x, 'cond': c, 'control_cond': cc}, t)
        name = 'trajnet_control_forward.npz' if ctrl else 'trajnet_forward.npz'
        np.savez_compressed(os.path.join(OUT, name), weight_seed=21 + ctrl, x_seed=201, cond_seed=202,
                            control_seed=203, t=t.numpy(), y=y.numpy())
        # 100-step loop, B=1 (BASELINE config 1 shape)
        if not ctrl:
            d = ref.model_util.create_gaussian_diffusion(_Args, ref.gd_trajnet, ref.respace.SpacedDiffusionTrajNet,
                                                         100, '', device='cpu')
            batch = {'cond': seeded(204, 1, 144, 13)}
            torch.manual_seed(4321)
            with torch.no_grad():
                _, y = d.eval_losses(model=tn, batch=batch, shape=[1, 144, 13], progress=False,
                                     clip_denoised=False, timestep_respacing='', cond_fn_with_grad=True,
                                     compute_loss=False)
            np.savez_compressed(os.path.join(OUT, 'trajnet_loop100.npz'), weight_seed=21, cond_seed=204,
                                torch_seed=4321, steps=100, y=y.numpy())
    # ---- geometry + guidance: the reference's own functions around the oracle body model ----------------
    from oracle import geometry as G
    body = G.BodyModel(synth.synthetic_smplx_tensors(0))
    refload.set_body_model(body)
    mean, std = synth.synthetic_stats(0)

    class GDS:
        pose_feat_dim, traj_feat_dim, joints_num = 272, 22, 22
        Mean, Std = mean, std
        cam_R = torch.tensor(synth.SYNTH_CAM_R)
        cam_t = torch.tensor(synth.SYNTH_CAM_T)
    gnet = ref.posenet.PoseNet(GDS(), 294, latent_dim=64, ff_size=64, num_layers=1, num_heads=1, traj_feat_dim=22,
                               device='cpu').eval()
    gnet.smplx_model = body
    x0 = synth.plausible_motion(3, 2, 143, mean, std)
    full = x0[:, :, 0].permute(0, 2, 1) * torch.from_numpy(std) + torch.from_numpy(mean)
    d = G.split_repr(full)
    j_abs = ref.motion_repr.recover_from_repr_smpl(d, recover_mode='joint_abs_traj', smplx_model=body)
    j_smpl = ref.motion_repr.recover_from_repr_smpl(d, recover_mode='smplx_params', smplx_model=body)
    g_sk = gnet.guide_skating_with_smpl({}, {'pred_xstart': x0}, None, compute_grad='x_0')
    cam = synth.synthetic_camera_batch(0, 2)
    g_2d = gnet.guide_2d_projection_with_smpl(cam, {'pred_xstart': x0}, None, compute_grad='x_0')
    r6 = seeded(301, 64, 6)
    Rm = ref.quaternion.rot6d_to_rotmat(r6)
    aa = ref.konia.rotation_matrix_to_angle_axis(Rm)
    np.savez_compressed(os.path.join(OUT, 'guidance.npz'), body_seed=0, stats_seed=0, motion_seed=3, cam_seed=0,
                        j_abs=j_abs.numpy(), j_smpl=j_smpl.numpy(), g_skating=g_sk.numpy(), g_2d=g_2d.numpy(),
                        r6_seed=301, rotmat=Rm.numpy(), angle_axis=aa.numpy())
    golden_rederive(ref)
    golden_metrics()
    golden_ddim(ref)
    golden_rel(ref)
    golden_eval_losses(ref)
    golden_guided_step(ref)
    golden_scheme(ref)
    golden_scheme_real(ref)
    golden_frames(ref)
    golden_control_loop(ref)
    golden_posenet_loop1000(ref)
    for f in sorted(os.listdir(OUT)):
        print(f, os.path.getsize(os.path.join(OUT, f)))


if __name__ == '__main__':
    main()
