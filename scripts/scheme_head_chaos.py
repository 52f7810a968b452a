#!/usr/bin/env python
"""Build container only (needs /root/reference): the reference's own p_sample_with_grad(grad_type='prox') against oracle/diffusion.py, both
fp32 on the CPU, free-running from the same start and noise over the guided head t = 106 .. 95 -- once from a th.randn start (what a PoseNet
stage of the drivers starts from) and once from a plausible motion.  Shows why the guided case of tests/golden/scheme_real.npz stops at
t = 99 (profiles/r4_scheme_head_chaos.txt)."""
import os, sys, types, warnings
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
warnings.filterwarnings('ignore')
import torch, numpy as np
from oracle import refload, geometry as G, diffusion as odiff, nets
from oracle.make_golden import *
from oracle.make_golden import _Args
from helpers import golden
ref = refload.load()
body = G.BodyModel(synth.synthetic_smplx_tensors(0)); refload.set_body_model(body)
ci=2
args, tfd, body_t, s_traj, s_pose, bt, bp, cam, _, plan = scheme_real_case(ci)
g = golden('scheme_real.npz')
# PoseNet cond of iteration 0 cannot be rebuilt trivially; use a plausible cond instead: the test is about the head dynamics
mean,std = s_pose
cond = synth.plausible_motion(63, 2, 143, mean, std)
pds = types.SimpleNamespace(traj_feat_dim=22, pose_feat_dim=272, joints_num=22, Mean=mean, Std=std, cam_R=torch.tensor(synth.SYNTH_CAM_R), cam_t=torch.tensor(synth.SYNTH_CAM_T))
pn = ref.posenet.PoseNet(pds, 294, latent_dim=512, ff_size=1024, num_layers=8, num_heads=4, traj_feat_dim=22, device='cpu').eval()
pn.smplx_model = body
sd_p = synth.posenet_state_dict(73); pn.load_state_dict(sd_p, strict=False)
d = ref.model_util.create_gaussian_diffusion(_Args, ref.gd_posenet, ref.respace.SpacedDiffusionPoseNet, 1000, '', device='cpu')
tab = odiff.tables(odiff.cosine_betas(1000))
mean_p, std_p = torch.as_tensor(mean), torch.as_tensor(std)
guid = {'skating': lambda x0, i: G.guide_skating(x0, mean_p, std_p, body),
        '2d': lambda x0,i: G.guide_2d_projection(x0, mean_p, std_p, body, cam['transf_matrix'], cam['focal_length'], cam['camera_center'], cam['keypoints_2d'], torch.tensor(synth.SYNTH_CAM_R), torch.tensor(synth.SYNTH_CAM_T))}
fn = lambda x, i: nets.posenet_forward(sd_p, x, cond, torch.full((2,), i, dtype=torch.int64))
for start in ('randn', 'plausible'):
    torch.manual_seed(5)
    x = torch.randn(2,294,1,143) if start=='randn' else synth.plausible_motion(61, 2, 143, mean, std) + 0.05*torch.randn(2,294,1,143)
    xr = x.clone(); xo = x.clone()
    for i in (106,105,104,103,102,101,100,99,98,97,96,95):
        batch = dict(cam); batch['cond']=cond
        torch.manual_seed(100+i); 
        with torch.no_grad():
            r = d.p_sample_with_grad(pn, batch, xr.clone(), torch.tensor([i]*2), clip_denoised=False, grad_type='prox')
        torch.manual_seed(100+i); nz = torch.randn(2,294,1,143)
        with torch.no_grad():
            tr = odiff.p_sample_loop(fn, xo, [nz], tab, [i], guidance=guid, grad_type='prox', return_all=True)
        xo, x0o = tr[0]
        xr = r['sample']
        print(start, i, 'max|x| %.2f' % float(xr.abs().max()), 'dx %.2e' % float((xr-xo).abs().max()), 'dx0 %.2e' % float((r['pred_xstart']-x0o).abs().max()))
