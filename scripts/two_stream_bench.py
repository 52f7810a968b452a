#!/usr/bin/env python
"""Experiment: one stream x B clips against S streams x B/S clips (independent clips, one host thread, handle
and workspace per stream).  Kernels of different streams run side by side, so one stream's launch gaps, prologues
and HBM-bound epilogues overlap the other stream's MFMA loops.
usage (GPU box): python scripts/two_stream_bench.py [ddpm_steps] [streams]"""
import os
import sys
import threading
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from rohm_amd.diffusion import gaussian_diffusion_posenet as gdp  # noqa: E402
from rohm_amd.diffusion.respace import SpacedDiffusionPoseNet  # noqa: E402
from rohm_amd.model.posenet import PoseNet  # noqa: E402
from rohm_amd.utils import synth  # noqa: E402
from rohm_amd.utils.model_util import create_gaussian_diffusion  # noqa: E402


def make(dev, S):
    net = PoseNet(bench._Dataset(), 294, latent_dim=512, ff_size=1024, num_layers=8, num_heads=4, traj_feat_dim=22,
                  body_model_path=torch.nn.Identity(), device=dev)
    net.load_state_dict(synth.posenet_state_dict(0), strict=True)
    net = net.to(dev).eval()
    diff = create_gaussian_diffusion(bench._Args, gdp, SpacedDiffusionPoseNet, S, '', device=dev)
    return net, diff


def run(nets, conds, streams):
    def work(k):
        net, diff = nets[k]
        with torch.cuda.stream(streams[k]):
            B = conds[k].shape[0]
            diff.eval_losses(model=net, batch={'cond': conds[k]}, shape=[B, 294, 1, 143], progress=False,
                             clip_denoised=False, timestep_respacing='', cond_fn_with_grad=False, compute_loss=False)
    th = [threading.Thread(target=work, args=(k,)) for k in range(len(nets))]
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for t in th:
        t.start()
    for t in th:
        t.join()
    torch.cuda.synchronize()
    return time.perf_counter() - t0


def main():
    S = int(sys.argv[1]) if len(sys.argv) > 1 else 200
    n = int(sys.argv[2]) if len(sys.argv) > 2 else 2
    B = 64
    dev = torch.device('cuda', 0)
    cond = bench.synthetic_cond(B, dev, 1000)
    nets = [make(dev, S) for _ in range(n)]
    streams = [torch.cuda.Stream(dev) for _ in range(n)]
    if n == 1:
        conds = [cond]
    else:
        conds = [c.contiguous() for c in cond.chunk(n, 0)]
    run(nets, conds, streams)
    best = min(run(nets, conds, streams) for _ in range(3))
    print(f'{n} stream(s) x {B // n} clips, {S} ddpm steps: {best * 1e3:.1f} ms -> '
          f'{B / best * S / 1000:.2f} clips/s @1000-step equivalent', flush=True)


if __name__ == '__main__':
    main()
