#!/usr/bin/env python
"""Phase timeline of the clip-resident TrajNet step (ROHM_TRAJ_RESIDENT_TIMELINE=1: the library prints the last step's per-layer
spans of XCD 0 to stderr).  usage (GPU box): python scripts/resident_timeline.py B [control]"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ['ROHM_TRAJ_RESIDENT_TIMELINE'] = '1'
os.environ['ROHM_TRAJ_RESIDENT_VERBOSE'] = '1'
from rohm_amd.diffusion import gaussian_diffusion_trajnet as gdt  # noqa: E402
from rohm_amd.diffusion.respace import SpacedDiffusionTrajNet  # noqa: E402
from rohm_amd.model.trajnet import TrajNet  # noqa: E402
from rohm_amd.utils import synth  # noqa: E402
from rohm_amd.utils.model_util import create_gaussian_diffusion  # noqa: E402


class Args:
    noise_schedule, sigma_small = 'cosine', True


def main():
    dev = torch.device('cuda', 0)
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 1
    ctrl = len(sys.argv) > 2 and sys.argv[2] == 'control'
    net = TrajNet(time_dim=32, mid_dim=512, cond_dim=13, traj_feat_dim=13, trajcontrol=ctrl, device=dev)
    net.load_state_dict(synth.trajnet_state_dict(1, trajcontrol=ctrl), strict=True)
    net = net.to(dev).eval()
    batch = {'cond': torch.randn(B, 144, 13, device=dev), 'control_cond': torch.randn(B, 144, 272, device=dev)}
    diff = create_gaussian_diffusion(Args, gdt, SpacedDiffusionTrajNet, 100, '', device=dev)
    for _ in range(2):
        diff.eval_losses(model=net, batch=batch, shape=[B, 144, 13], progress=False, clip_denoised=False, timestep_respacing='',
                         cond_fn_with_grad=False, compute_loss=False)
        torch.cuda.synchronize()


if __name__ == '__main__':
    main()
