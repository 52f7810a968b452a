#!/usr/bin/env python
"""TrajNet / TrajControl 100-step sampling loop: the clip-resident step (csrc/trajnet_resident.hip, one launch per step) against the
launch-per-layer loop (ROHM_TRAJ_RESIDENT=0) -- same inputs and noise: largest difference of the samples, wall time per loop.
usage (GPU box): python scripts/resident_ab.py [B ...]"""
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from rohm_amd.diffusion import gaussian_diffusion_trajnet as gdt  # noqa: E402
from rohm_amd.diffusion.respace import SpacedDiffusionTrajNet  # noqa: E402
from rohm_amd.model.trajnet import TrajNet  # noqa: E402
from rohm_amd.utils import synth  # noqa: E402
from rohm_amd.utils.model_util import create_gaussian_diffusion  # noqa: E402


class Args:
    noise_schedule, sigma_small = 'cosine', True


def main():
    dev = torch.device('cuda', 0)
    batches = tuple(int(a) for a in sys.argv[1:]) or (1, 2, 8, 9, 32, 64)
    res = {}
    for ctrl in (False, True):
        net = TrajNet(time_dim=32, mid_dim=512, cond_dim=13, traj_feat_dim=13, trajcontrol=ctrl, device=dev)
        net.load_state_dict(synth.trajnet_state_dict(1, trajcontrol=ctrl), strict=True)
        net = net.to(dev).eval()
        for B in batches:
            g = torch.Generator(device='cpu').manual_seed(100 + B)
            x_T = torch.randn(B, 144, 13, generator=g).to(dev)
            noises = torch.randn(100, B, 144, 13, generator=g).to(dev)
            batch = {'cond': torch.randn(B, 144, 13, generator=g).to(dev), 'control_cond': torch.randn(B, 144, 272, generator=g).to(dev)}
            out = {}
            for mode in ('1', '0'):
                os.environ['ROHM_TRAJ_RESIDENT'] = mode
                diff = create_gaussian_diffusion(Args, gdt, SpacedDiffusionTrajNet, 100, '', device=dev)
                diff.noise_source = lambda step, like: (x_T if step == -1 else noises[step])
                run = lambda: diff.eval_losses(model=net, batch=batch, shape=[B, 144, 13], progress=False, clip_denoised=False,
                                               timestep_respacing='', cond_fn_with_grad=False, compute_loss=False)[1]
                y = run()
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                for _ in range(3):
                    y2 = run()
                torch.cuda.synchronize()
                out[mode] = (y.clone(), (time.perf_counter() - t0) / 3 * 1e3, bool(torch.equal(y, y2)))
            d = (out['1'][0] - out['0'][0]).abs().max().item()
            key = f'{"control" if ctrl else "vanilla"}_B{B}'
            res[key] = {'resident_ms': round(out['1'][1], 2), 'launches_ms': round(out['0'][1], 2), 'max_abs_diff': d,
                        'finite': bool(torch.isfinite(out['1'][0]).all()), 'resident_repeatable': out['1'][2],
                        'sample_abs_max': out['0'][0].abs().max().item()}
            print(key, res[key], flush=True)
    os.environ.pop('ROHM_TRAJ_RESIDENT', None)
    print(json.dumps(res, indent=1))


if __name__ == '__main__':
    main()
