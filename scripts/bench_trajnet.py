#!/usr/bin/env python
"""TrajNet / TrajControl 100-step sampling loop: wall time per run against the sum of kernel times (in-library HIP-event
profiler), to see whether the loop is launch-bound.  usage (GPU box): python scripts/bench_trajnet.py"""
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from rohm_amd import _lib  # noqa: E402
from rohm_amd.diffusion import gaussian_diffusion_trajnet as gdt  # noqa: E402
from rohm_amd.diffusion.respace import SpacedDiffusionTrajNet  # noqa: E402
from rohm_amd.model.trajnet import TrajNet  # noqa: E402
from rohm_amd.utils import synth  # noqa: E402
from rohm_amd.utils.model_util import create_gaussian_diffusion  # noqa: E402


class Args:
    noise_schedule, sigma_small = 'cosine', True


DEFAULT_TUNE = (1, 2, 0)        # the library's defaults (rohm_trajnet_tune): workgroups per CU, min chunks per split, pow2 splits


def sweep():
    """Wall time of the 100-step loop over the conv launch shapes of rohm_trajnet_tune (workgroups per CU x fewest K
    chunks per split-K slice); usage: python scripts/bench_trajnet.py --sweep [B ...]"""
    dev = torch.device('cuda', 0)
    batches = tuple(int(a) for a in sys.argv[2:]) or (1, 32)
    res = {}
    for ctrl in (False, True):
        net = TrajNet(time_dim=32, mid_dim=512, cond_dim=13, traj_feat_dim=13, trajcontrol=ctrl, device=dev)
        net.load_state_dict(synth.trajnet_state_dict(1, trajcontrol=ctrl), strict=True)
        net = net.to(dev).eval()
        diff = create_gaussian_diffusion(Args, gdt, SpacedDiffusionTrajNet, 100, '', device=dev)
        for B in batches:
            batch = {'cond': torch.randn(B, 144, 13, device=dev), 'control_cond': torch.randn(B, 144, 272, device=dev)}
            run = lambda: diff.eval_losses(model=net, batch=batch, shape=[B, 144, 13], progress=False, clip_denoised=False,
                                           timestep_respacing='', cond_fn_with_grad=False, compute_loss=False)
            for wg, mc, p2 in ((1, 4, 0), (1, 2, 0), (1, 1, 0), (1, 4, 1), (1, 2, 1), (1, 1, 1), (2, 2, 1)):
                _lib.check(_lib.lib().rohm_trajnet_tune(wg, mc, p2), 'rohm_trajnet_tune')
                run()
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                for _ in range(4):
                    run()
                torch.cuda.synchronize()
                res[f'{"control" if ctrl else "vanilla"}_B{B}_wg{wg}_minchunks{mc}_pow2{p2}'] = round((time.perf_counter() - t0) / 4 * 1e3, 2)
    _lib.check(_lib.lib().rohm_trajnet_tune(*DEFAULT_TUNE), 'rohm_trajnet_tune')
    print(json.dumps(res, indent=1))


def detail():
    """Per-launch-shape event times of one vanilla 100-step loop (rohm_profile_detail): which convolutions / GroupNorms the
    step's time sits in, and their fixed cost against their K-chunk count.  usage: bench_trajnet.py --detail B [wg mc pow2]"""
    dev = torch.device('cuda', 0)
    B = int(sys.argv[2]) if len(sys.argv) > 2 else 32
    tune = tuple(int(a) for a in sys.argv[3:6]) if len(sys.argv) > 5 else DEFAULT_TUNE
    net = TrajNet(time_dim=32, mid_dim=512, cond_dim=13, traj_feat_dim=13, trajcontrol=False, device=dev)
    net.load_state_dict(synth.trajnet_state_dict(1, trajcontrol=False), strict=True)
    net = net.to(dev).eval()
    diff = create_gaussian_diffusion(Args, gdt, SpacedDiffusionTrajNet, 100, '', device=dev)
    batch = {'cond': torch.randn(B, 144, 13, device=dev)}
    run = lambda: diff.eval_losses(model=net, batch=batch, shape=[B, 144, 13], progress=False, clip_denoised=False,
                                   timestep_respacing='', cond_fn_with_grad=False, compute_loss=False)
    _lib.check(_lib.lib().rohm_trajnet_tune(*tune), 'rohm_trajnet_tune')
    run()
    torch.cuda.synchronize()
    _lib.check(_lib.lib().rohm_profile_detail(1), 'rohm_profile_detail')
    _lib.profile_start(4)
    run()
    torch.cuda.synchronize()
    prof = _lib.profile_stop()
    _lib.check(_lib.lib().rohm_profile_detail(0), 'rohm_profile_detail')
    _lib.check(_lib.lib().rohm_trajnet_tune(*DEFAULT_TUNE), 'rohm_trajnet_tune')
    tot = sum(v['total_ms'] for v in prof.values())
    print(f'# B={B} tune={tune}: per-shape HIP-event times (every 4th step), total {tot:.2f} ms of events')
    for k, v in sorted(prof.items(), key=lambda kv: -kv[1]['total_ms']):
        print(f"{k:48s} n={v['launches']:4d} avg_us={v['total_ms'] / v['launches'] * 1e3:7.2f} share={v['total_ms'] / tot:.3f} "
              f"gflop={v['flops'] / max(v['launches'], 1) / 1e9:.3f}")


def main():
    if len(sys.argv) > 1 and sys.argv[1] == '--sweep':
        return sweep()
    if len(sys.argv) > 1 and sys.argv[1] == '--detail':
        return detail()
    dev = torch.device('cuda', 0)
    res = {}
    for ctrl in (False, True):
        net = TrajNet(time_dim=32, mid_dim=512, cond_dim=13, traj_feat_dim=13, trajcontrol=ctrl, device=dev)
        net.load_state_dict(synth.trajnet_state_dict(1, trajcontrol=ctrl), strict=True)
        net = net.to(dev).eval()
        diff = create_gaussian_diffusion(Args, gdt, SpacedDiffusionTrajNet, 100, '', device=dev)
        for B in ((1, 32, 256) if len(sys.argv) < 2 else tuple(int(a) for a in sys.argv[1:])):
            batch = {'cond': torch.randn(B, 144, 13, device=dev), 'control_cond': torch.randn(B, 144, 272, device=dev)}
            run = lambda: diff.eval_losses(model=net, batch=batch, shape=[B, 144, 13], progress=False, clip_denoised=False,
                                           timestep_respacing='', cond_fn_with_grad=False, compute_loss=False)
            def timed():
                run()
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                for _ in range(3):
                    run()
                t_enq = (time.perf_counter() - t0) / 3      # host time to enqueue (the call returns before the GPU is done)
                torch.cuda.synchronize()
                return (time.perf_counter() - t0) / 3, t_enq
            keep = os.environ.get('ROHM_TRAJ_RESIDENT')
            wall_default, _ = timed()                       # the library's default form (TrajNet: one launch per step at B <= 64)
            os.environ['ROHM_TRAJ_RESIDENT'] = '0'          # the launch-per-layer loop: what the launch profiler below breaks down
            wall, t_enq = timed()
            _lib.profile_start(1)
            run()
            torch.cuda.synchronize()
            prof = _lib.profile_stop()
            ksum = sum(v['total_ms'] for v in prof.values())
            n = sum(v['launches'] for v in prof.values())
            if keep is None:
                os.environ.pop('ROHM_TRAJ_RESIDENT', None)
            else:
                os.environ['ROHM_TRAJ_RESIDENT'] = keep
            res[f'{"control" if ctrl else "vanilla"}_B{B}'] = {'wall_ms_default_form': round(wall_default * 1e3, 2), 'wall_ms': round(wall * 1e3, 2), 'host_enqueue_ms': round(t_enq * 1e3, 2), 'kernel_event_ms': round(ksum, 2),
                                                                'launches': n, 'clips_per_s': round(B / wall, 1),
                                                                'kernels': {k: {'launches': v['launches'], 'avg_us': round(v['total_ms'] / v['launches'] * 1e3, 2),
                                                                                'share': round(v['total_ms'] / ksum, 3)}
                                                                            for k, v in sorted(prof.items(), key=lambda kv: -kv[1]['total_ms'])}}
    print(json.dumps(res, indent=1))


if __name__ == '__main__':
    main()
