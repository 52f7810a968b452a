#!/usr/bin/env python
"""Why no guided sampling RUN is compared free-running at the reference's guidance weights: measured, not argued.

Build container only (needs /root/reference).  The reference's own `p_sample_with_grad(grad_type='prox')`
(diffusion/gaussian_diffusion_posenet.py:436-480: 2-D re-projection term x 3e5, skating term x 1e5 on t <= 100) and the CPU
oracle (oracle/diffusion.py::p_sample_loop with oracle/geometry.py's guidance), both fp32 on the same weights, inputs and
noise, are run over the guided tail t = 110 .. 20 (the early-stop end of the 980-step PROX run) twice:

  teacher-forced   every step starts from the REFERENCE's x_t            -> per-step error of the restatement
  free-running     each side continues from its own x_t                  -> how that error is amplified along the run

and the same with both weights divided by 1000 (the setting of tests/test_gpu_guidance.py's free-running loop test).
Prints one line per step and a summary; the trace is committed under profiles/."""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))
from oracle import diffusion as odiff  # noqa: E402
from oracle import geometry as G  # noqa: E402
from oracle import nets, refload  # noqa: E402
from oracle.make_golden import _Args, guided_step_inputs  # noqa: E402
from rohm_amd.utils import synth  # noqa: E402


def main():
    if not refload.available():
        raise SystemExit('needs the reference tree (/root/reference): build container only')
    torch.set_num_threads(min(8, os.cpu_count() or 1))
    ref = refload.load()
    body = G.BodyModel(synth.synthetic_smplx_tensors(0))
    refload.set_body_model(body)
    seeds = dict(stats_seed=0, x_seed=61, xn_seed=62, cond_seed=63, cam_seed=2, weight_seed=13, body_seed=0)
    mean, std, x0_in, cond, cam = guided_step_inputs(seeds)

    class GDS:
        pose_feat_dim, traj_feat_dim, joints_num = 272, 22, 22
        Mean, Std = mean, std
        cam_R = torch.tensor(synth.SYNTH_CAM_R)
        cam_t = torch.tensor(synth.SYNTH_CAM_T)
    import contextlib
    with contextlib.redirect_stdout(sys.stderr):
        net = ref.posenet.PoseNet(GDS(), 294, latent_dim=512, ff_size=1024, num_layers=8, num_heads=4, traj_feat_dim=22,
                                  device='cpu').eval()
    net.smplx_model = body
    sd = synth.posenet_state_dict(seeds['weight_seed'])
    net.load_state_dict(sd, strict=False)
    diff = ref.model_util.create_gaussian_diffusion(_Args, ref.gd_posenet, ref.respace.SpacedDiffusionPoseNet, 1000, '', device='cpu')
    tab = odiff.tables(odiff.cosine_betas(1000))
    m, s = torch.from_numpy(mean), torch.from_numpy(std)
    fn = lambda xx, i: nets.posenet_forward(sd, xx, cond, torch.full((xx.shape[0],), i, dtype=torch.int64))
    guid = {'skating': lambda x0, i: G.guide_skating(x0, m, s, body),
            '2d': lambda x0, i: G.guide_2d_projection(x0, m, s, body, cam['transf_matrix'], cam['focal_length'], cam['camera_center'],
                                                      cam['keypoints_2d'], torch.tensor(synth.SYNTH_CAM_R), torch.tensor(synth.SYNTH_CAM_T))}
    idx = list(range(110, 19, -1))
    ref_hooks = ref.gd_posenet.GaussianDiffusionPoseNet.p_sample_with_grad
    src = open(os.path.join(refload.REF_ROOT, 'diffusion', 'gaussian_diffusion_posenet.py')).read()
    assert '3e5' in src and '1e5' in src, 'the reference no longer hard-codes the PROX weights this script scales'

    def run(scale):
        """scale 1: the reference's weights; 1e-3: both divided by 1000 (the reference's are literals in the source: the method
        is re-compiled from its own text with the two literals replaced, nothing else touched)."""
        step_ref = diff.p_sample_with_grad
        if scale != 1.0:
            import textwrap
            i0 = src.index('    def p_sample_with_grad(')
            i1 = src.index('\n    def ', i0 + 10)
            body_src = textwrap.dedent(src[i0:i1]).replace('3e5', repr(3e5 * scale)).replace('1e5', repr(1e5 * scale))
            ns = dict(vars(ref.gd_posenet))
            exec(body_src, ns)
            step_ref = lambda *a, **k: ns['p_sample_with_grad'](diff, *a, **k)
        old = odiff.GUIDANCE['prox']
        odiff.GUIDANCE['prox'] = (100, (('2d', 3e5 * scale), ('skating', 1e5 * scale)))
        rows = []
        x_ref, x_or = x0_in.clone(), x0_in.clone()
        try:
            for k, i in enumerate(idx):
                batch = dict(cam)
                batch['cond'] = cond
                t = torch.tensor([i] * 2)
                torch.manual_seed(1000 + k)
                with torch.no_grad():
                    r = step_ref(net, batch, x_ref.clone(), t, clip_denoised=False, grad_type='prox')
                torch.manual_seed(1000 + k)
                noise = torch.randn(2, 294, 1, 143)
                (tf, _), = odiff.p_sample_loop(fn, x_ref, [noise], tab, [i], guidance=guid, grad_type='prox', return_all=True)
                (fr, _), = odiff.p_sample_loop(fn, x_or, [noise], tab, [i], guidance=guid, grad_type='prox', return_all=True)
                x_ref, x_or = r['sample'], fr
                rows.append((i, float((tf - x_ref).abs().max()), float((x_or - x_ref).abs().max()), float(x_ref.abs().max())))
                print(f'scale {scale:g}  t {i:4d}  teacher-forced step error {rows[-1][1]:.3e}   free-running distance {rows[-1][2]:.3e}   '
                      f'max|x_ref| {rows[-1][3]:.2f}', flush=True)
        finally:
            odiff.GUIDANCE['prox'] = old
        return rows

    t0 = time.time()
    out = {}
    for scale in (1.0, 1e-3):
        rows = run(scale)
        out[scale] = rows
        tf = max(r[1] for r in rows)
        first = next((r[0] for r in rows if r[2] > 1e-3), None)
        print(f'== weights x {scale:g}: worst teacher-forced step error {tf:.3e}; free-running distance at t = 20: {rows[-1][2]:.3e} '
              f'(max|x| {rows[-1][3]:.2f}); first t with distance > 1e-3: {first}')
    print(f'({time.time() - t0:.0f} s on {torch.get_num_threads()} threads)')


if __name__ == '__main__':
    main()
