#!/usr/bin/env python
"""Measurement of the full SMPL-X LBS path (`rohm_smplx_forward`, the device replacement of
smplx's lbs() as the reference calls it with return_verts=True, data_loaders/motion_representation.py:389-396 from
test_amass_full.py:405-425): per-kernel device time from the library's own HIP events on the launch stream, the
pose-blendshape GEMM against the fp32 MFMA peak, the skinning kernel against HBM.
usage (GPU box): python scripts/bench_lbs.py [n_clips] > gpurun_out/<tag>/lbs.json"""
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from rohm_amd import _lib  # noqa: E402
from rohm_amd.body_model import SMPLXLayer, lbs_forward, native_for  # noqa: E402
from rohm_amd.utils import synth  # noqa: E402

PEAK_F32_TF, PEAK_HBM_TBS = 157.3, 8.0


def main():
    n_clips = int(sys.argv[1]) if len(sys.argv) > 1 else 32
    # model: 'dense' (the synthetic row-softmax weights: MFMA skinning unless ROHM_LBS_SKIN says otherwise) or 'sparse' (4 non-zero
    # joints per vertex, like a released SMPLX_*.npz: ELL skinning)
    model = sys.argv[2] if len(sys.argv) > 2 else 'dense'
    dev = torch.device('cuda', 0)
    t = synth.synthetic_smplx_tensors(0)
    if model == 'sparse':
        w = t['lbs_weights']
        top = torch.topk(w, 4, dim=1)
        sw = torch.zeros_like(w).scatter_(1, top.indices, top.values)
        t['lbs_weights'] = sw / sw.sum(1, keepdim=True)
    layer = SMPLXLayer.from_tensors(t).to(dev)
    nat = native_for(layer, dev)
    N = n_clips * 143
    g = torch.Generator().manual_seed(0)
    pose = (torch.randn(N, 22, 3, generator=g) * 0.4).to(dev)
    betas, transl = (torch.randn(N, 10, generator=g) * 0.5).to(dev), torch.randn(N, 3, generator=g).to(dev)
    res = {'frames': N, 'vertices': nat.num_verts, 'joints': nat.num_joints, 'model': model,
           'skinning_mode': {0: 'mfma', 1: 'sparse-ell', 2: 'dense-ell (round-3 VALU form)'}[_lib.lib().rohm_smplx_skinning_mode(nat.handle)]}
    for want_verts in (True, False):
        for _ in range(3):
            lbs_forward(nat, pose, 0, betas, transl, want_verts)
        torch.cuda.synchronize()
        _lib.profile_start(1)
        reps = 10
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            lbs_forward(nat, pose, 0, betas, transl, want_verts)
        e1.record()
        torch.cuda.synchronize()
        rows = _lib.profile_stop()
        out = {'wall_us_per_call': round(e0.elapsed_time(e1) / reps * 1e3, 1), 'kernels': {}}
        for name, r in rows.items():
            us = r['total_ms'] / r['launches'] * 1e3
            k = {'us': round(us, 1)}
            if r['flops'] and (name.startswith('gemm') or name == 'lbs_skin_mfma'):
                tf = r['flops'] / r['launches'] / (us * 1e-6) / 1e12
                k.update(gflop=round(r['flops'] / r['launches'] / 1e9, 2), tflops=round(tf, 1), frac_f32_mfma=round(tf / PEAK_F32_TF, 3))
            if r['bytes']:
                tbs = r['bytes'] / r['launches'] / (us * 1e-6) / 1e12
                k.update(alg_mb=round(r['bytes'] / r['launches'] / 1e6, 1), alg_tb_s=round(tbs, 2), frac_hbm=round(tbs / PEAK_HBM_TBS, 3))
            out['kernels'][name] = k
        out['frames_per_s'] = round(N / (out['wall_us_per_call'] * 1e-6))
        res['with_vertices' if want_verts else 'joints_only'] = out
    print(json.dumps(res))


if __name__ == '__main__':
    main()
