"""GPU: the evaluation loss reports behind `eval_losses(compute_loss=True)` (the default of test_posenet.py /
test_trajnet.py) against the values the reference's own `compute_losses_with_smpl` methods produced
(tests/golden/eval_losses.npz)."""
import numpy as np
import pytest
import torch

from helpers import PoseDataset, golden, seeded
from rohm_amd.utils import synth

pytestmark = pytest.mark.gpu
DEV = 'cuda:0'


def _check(d, g, prefix):
    keys = [k[len(prefix):] for k in g.files if k.startswith(prefix)]
    assert set(keys) == set(d.keys())
    for k in keys:
        ref = float(g[prefix + k])
        val = float(d[k])
        assert abs(val - ref) <= 2e-4 * max(abs(ref), 1e-6), (k, val, ref)


def test_posenet_eval_losses():
    from rohm_amd.body_model import SMPLXLayer
    from rohm_amd.model.posenet import PoseNet
    g = golden('eval_losses.npz')
    mean, std = synth.synthetic_stats(0)
    layer = SMPLXLayer.from_tensors(synth.synthetic_smplx_tensors(0)).to(DEV)
    net = PoseNet(PoseDataset(mean, std), 294, latent_dim=512, ff_size=1024, num_layers=8, num_heads=4, traj_feat_dim=22,
                  body_model_path=layer, device=DEV, weight_loss_rec_repr_full_body=1.0,
                  weight_loss_repr_foot_contact_mse=0.5, weight_loss_joint_pos_global=2.0, weight_loss_joint_vel_global=3.0,
                  weight_loss_joint_smooth=0.7, weight_loss_foot_skating=0.3, start_skating_loss_epoch=0)
    clean = synth.plausible_motion(11, 3, 143, mean, std)
    rec = clean + 0.05 * seeded(12, 3, 294, 1, 143)
    d = net.compute_losses_with_smpl({'motion_repr_clean': clean.to(DEV)}, rec.to(DEV), smplx_model=layer, epoch=0)
    _check(d, g, 'posenet_')


@pytest.mark.parametrize('abs_only,dim', [(True, 13), (False, 22)])
def test_trajnet_eval_losses(abs_only, dim):
    from rohm_amd.body_model import SMPLXLayer
    from rohm_amd.model.trajnet import TrajNet
    g = golden('eval_losses.npz')
    mean, std = synth.synthetic_stats(0)
    ds = PoseDataset(mean, std)
    ds.traj_feat_dim = dim
    layer = SMPLXLayer.from_tensors(synth.synthetic_smplx_tensors(0)).to(DEV)
    net = TrajNet(time_dim=32, cond_dim=dim, mid_dim=512, traj_feat_dim=dim, device=DEV, dataset=ds,
                  repr_abs_only=abs_only, trajcontrol=False, weight_loss_root_rec_repr=1.0, weight_loss_root_pos_global=2.0,
                  weight_loss_root_vel_global=3.0, weight_loss_root_rot_vel_from_abs_traj=0.4,
                  weight_loss_root_smplx_transl_vel=0.6, weight_loss_root_smplx_rot_vel=0.8, weight_loss_root_smooth=0.9,
                  weight_loss_root_rot_cos_smooth_from_abs_traj=1.1)
    clean = synth.plausible_motion(11, 3, 143, mean, std)[:, :, 0].permute(0, 2, 1).contiguous()[:, :128]
    mo = seeded(13 + dim, 3, 128, dim) * 0.3
    d = net.compute_losses_with_smpl({'motion_repr_clean': clean.to(DEV)}, mo.to(DEV), smplx_model=layer)
    _check(d, g, f'trajnet{dim}_')


def test_eval_losses_compute_loss_true_returns_report():
    """The single-stage drivers call eval_losses with its default compute_loss=True (test_trajnet.py:154)."""
    from test_gpu_trajnet import Args
    from rohm_amd.body_model import SMPLXLayer
    from rohm_amd.diffusion import gaussian_diffusion_trajnet as gdt
    from rohm_amd.diffusion.respace import SpacedDiffusionTrajNet
    from rohm_amd.model.trajnet import TrajNet
    from rohm_amd.utils.model_util import create_gaussian_diffusion
    mean, std = synth.synthetic_stats(0)
    ds = PoseDataset(mean, std)
    ds.traj_feat_dim = 13
    layer = SMPLXLayer.from_tensors(synth.synthetic_smplx_tensors(0)).to(DEV)
    net = TrajNet(time_dim=32, cond_dim=13, mid_dim=512, traj_feat_dim=13, device=DEV, dataset=ds, repr_abs_only=True,
                  weight_loss_root_rec_repr=1.0)
    net.load_state_dict(synth.trajnet_state_dict(5), strict=True)
    net = net.to(DEV).eval()
    diff = create_gaussian_diffusion(Args, gdt, SpacedDiffusionTrajNet, 4, '', device=DEV)
    clean = synth.plausible_motion(11, 2, 144, mean, std)[:, :, 0].permute(0, 2, 1).contiguous().to(DEV)
    batch = {'motion_repr_clean': clean, 'cond': clean[:, :, [0, 2, 3, 6, 7, 8, 9, 10, 11, 12, 16, 17, 18]].contiguous()}
    report, out = diff.eval_losses(model=net, batch=batch, shape=[2, 144, 13], progress=False, clip_denoised=False,
                                   timestep_respacing='', cond_fn_with_grad=False, smplx_model=layer)
    assert out.shape == (2, 144, 13) and 'loss' in report and torch.isfinite(report['loss'])
    assert float(report['loss_root_pos_global_from_rel_traj']) == 0.0          # repr_abs_only (trajnet.py:381-384)
