#!/usr/bin/env python
"""Measurement of the between-stage kernels: device time per call (HIP events on the launch
stream, mean of 50 calls) beside the CPU restatement of the reference's host path (oracle/rederive.py: LBS with
vertices + per-sequence get_repr_smplx loop) timed on the host cores on a bounded sample.
usage (GPU box): python scripts/bench_stages.py > gpurun_out/<tag>/stages.json"""
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from rohm_amd.body_model import SMPLXLayer  # noqa: E402
from rohm_amd.data_loaders.motion_representation import joints_from_repr, rederive_traj  # noqa: E402
from rohm_amd.evaluation import amass_metrics  # noqa: E402
from rohm_amd.utils import synth  # noqa: E402


def timed(fn, n=50):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


def main():
    dev = torch.device('cuda', 0)
    body_t = synth.synthetic_smplx_tensors(0)
    layer = SMPLXLayer.from_tensors(body_t).to(dev)
    s_in, s_out = synth.synthetic_stats(0), synth.synthetic_stats(1)
    res = {'device_us': {}, 'alg_bytes': {}}
    for B in (32, 64, 256):
        x = synth.walking_motion(1, B, 144, *s_in, body_t).to(dev)
        cond = torch.zeros(B, 143, 294, device=dev)
        us = timed(lambda: rederive_traj(x, s_in, s_out, layer, out=cond))
        byt = B * (144 * 155 + 143 * 22) * 4
        res['device_us'][f'traj_rederive_B{B}'] = round(us, 2)
        res['alg_bytes'][f'traj_rederive_B{B}'] = byt
        us = timed(lambda: joints_from_repr(x, 'smplx_params', layer, stats=s_in))
        res['device_us'][f'repr_joints_B{B}'] = round(us, 2)
        j = joints_from_repr(x, 'smplx_params', layer, stats=s_in)
        j2 = j + 0.01 * torch.randn_like(j)
        us = timed(lambda: amass_metrics(j, j2, x, x, 'lower'))
        res['device_us'][f'amass_metrics_B{B}_incl_d2h'] = round(us, 2)
    # CPU: the reference-shaped host path on a bounded sample
    from oracle import geometry as G
    from oracle import rederive as RD
    torch.set_num_threads(min(16, os.cpu_count() or 1))
    Bc = 4
    xc = synth.walking_motion(1, Bc, 144, *s_in, body_t)
    body = G.BodyModel(body_t)
    t0 = time.perf_counter()
    d = G.split_repr(torch.from_numpy(xc.numpy() * s_in[1] + s_in[0]))
    G.joints_from_smplx(d, body, return_verts=True)          # the reference builds the vertices too (return_verts=True)
    t_lbs = time.perf_counter() - t0
    t0 = time.perf_counter()
    RD.rederive_traj(xc, *s_in, *s_out, body)
    t_all = time.perf_counter() - t0
    res['cpu_port_ms_per_clip'] = {'lbs_with_vertices': round(t_lbs / Bc * 1e3, 2),
                                   'joints_fk_plus_get_repr_loop': round(t_all / Bc * 1e3, 2),
                                   'sample': f'{Bc} clips of 144 frames, {torch.get_num_threads()} threads'}
    print(json.dumps(res))


if __name__ == '__main__':
    main()
