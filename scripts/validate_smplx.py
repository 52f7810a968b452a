#!/usr/bin/env python
"""Close the SMPL-X parity gap on a machine that HAS the real body model.

The SMPL-X arithmetic RoHM runs lives in third-party `smplx==0.1.28` (environment.yml:198; call sites
data_loaders/motion_representation.py:379-396, model/posenet.py:57-58), which is neither in the reference tree nor in
this image, and no SMPLX_NEUTRAL.npz is available here -- so `oracle.geometry.BodyModel` (the CPU restatement of
smplx.lbs) and `rohm_smplx_forward` / `rohm_smplx_joints` (HIP) are only checked against EACH OTHER in the test-suite:
"parity unpinned".  This script pins them as far as the machine it runs on allows:

  python scripts/validate_smplx.py --model-path body_models/smplx_model      # needs `pip install smplx==0.1.28`
  python scripts/validate_smplx.py --npz body_models/smplx_model/smplx/SMPLX_NEUTRAL.npz

  (a) `smplx` importable + model files: the REAL smplx.create(model_type='smplx', gender='neutral',
      flat_hand_mean=True, use_pca=False) forward (joints[:, :55], vertices) vs the oracle restatement on the same
      tensors -> pins the oracle;  and, with an AMD GPU, vs SMPLXLayer (HIP) -> pins the kernels.        tolerance 1e-5 m
  (b) only the .npz: oracle vs HIP on the REAL model tensors (real sparsity / magnitudes instead of synthetic ones).
  (c) neither: exits 3 and says so.
Exit code 0 = everything it could check passed; 1 = a mismatch; 3 = nothing to check against."""
from __future__ import annotations

import argparse
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
TOL = 1e-5


def tensors_from_npz(path, num_betas=10, num_expr=10):
    """The tensors smplx builds from an SMPLX_*.npz (body_models.SMPLX.__init__): first `num_betas` shape components +
    first `num_expr` expression components (stored from index 300), posedirs as [(J-1)*9, V*3]."""
    d = np.load(path, allow_pickle=True)
    sd = np.asarray(d['shapedirs'], dtype=np.float32)
    if sd.shape[2] >= 300 + num_expr:
        sd = np.concatenate([sd[:, :, :num_betas], sd[:, :, 300:300 + num_expr]], axis=2)
    parents = np.asarray(d['kintree_table'])[0].astype(np.int64)
    parents[0] = -1
    pd = np.asarray(d['posedirs'], dtype=np.float32)
    pd = pd.reshape(-1, pd.shape[-1]).T
    t = lambda a: torch.from_numpy(np.ascontiguousarray(np.asarray(a, dtype=np.float32)))
    return {'v_template': t(d['v_template']), 'shapedirs': t(sd), 'posedirs': t(pd), 'J_regressor': t(d['J_regressor']),
            'lbs_weights': t(d['weights']), 'parents': torch.from_numpy(parents)}


def random_params(N, seed=0):
    g = torch.Generator().manual_seed(seed)
    r = lambda *s, k=1.0: torch.randn(*s, generator=g) * k
    p = {'betas': r(N, 10), 'global_orient': r(N, 3, k=0.8), 'body_pose': r(N, 63, k=0.4), 'transl': r(N, 3)}
    p['body_pose'][:4] = 0.0          # exact-zero rotations: Rodrigues' 1e-8 guard
    return p


def report(name, a, b):
    err = float((a.double() - b.double()).abs().max())
    ok = err < TOL
    print(f'  {name:58s} max|diff| = {err:.3e}  {"ok" if ok else "MISMATCH"}')
    return ok


def main():
    ap = argparse.ArgumentParser(description=__doc__, formatter_class=argparse.RawDescriptionHelpFormatter)
    ap.add_argument('--model-path', default=None, help='directory smplx.create() takes (contains smplx/SMPLX_NEUTRAL.npz)')
    ap.add_argument('--npz', default=None, help='an SMPLX_*.npz model file')
    ap.add_argument('--frames', type=int, default=64)
    args = ap.parse_args()
    from oracle import geometry as G
    ok, checked = True, 0
    p = random_params(args.frames)
    zeros = {'jaw_pose': torch.zeros(args.frames, 3), 'leye_pose': torch.zeros(args.frames, 3),
             'reye_pose': torch.zeros(args.frames, 3), 'left_hand_pose': torch.zeros(args.frames, 45),
             'right_hand_pose': torch.zeros(args.frames, 45), 'expression': torch.zeros(args.frames, 10)}
    real = None
    try:
        import smplx
        if args.model_path:
            real = smplx.create(model_path=args.model_path, model_type='smplx', gender='neutral', flat_hand_mean=True,
                                use_pca=False, batch_size=args.frames)
            print(f'smplx {getattr(smplx, "__version__", "?")} model loaded from {args.model_path}')
    except ImportError:
        print('smplx is not installed')
    tensors = None
    if real is not None:
        tensors = {'v_template': real.v_template.detach(), 'shapedirs': real.shapedirs.detach(),
                   'posedirs': real.posedirs.detach(), 'J_regressor': real.J_regressor.detach(),
                   'lbs_weights': real.lbs_weights.detach(), 'parents': real.parents.detach().clone()}
        tensors['parents'][0] = -1
    elif args.npz:
        tensors = tensors_from_npz(args.npz)
        print(f'model tensors from {args.npz}: V = {tensors["v_template"].shape[0]}, J = {tensors["J_regressor"].shape[0]}')
    if tensors is None:
        print('nothing to validate against (no smplx model, no .npz): SMPL-X parity stays UNPINNED on this machine')
        return 3
    oracle = G.BodyModel(tensors)
    with torch.no_grad():
        o = oracle(**p, **zeros, return_verts=True)
    if real is not None:
        with torch.no_grad():
            r = real(**p, **zeros, return_verts=True)
        print('real smplx vs oracle restatement (CPU):')
        ok &= report('joints[:, :55]', o.joints[:, :55], r.joints[:, :55])
        ok &= report('vertices', o.vertices, r.vertices)
        checked += 2
    if torch.cuda.is_available():
        from rohm_amd.body_model import SMPLXLayer
        layer = SMPLXLayer.from_tensors(tensors).to('cuda:0')
        pd = {k: v.to('cuda:0') for k, v in p.items()}
        with torch.no_grad():
            hj = layer(**pd).joints.cpu()
            hv = layer(**pd, return_verts=True)
        ref = r if real is not None else o
        print(f'HIP (librohm_hip.so) vs {"real smplx" if real is not None else "oracle on the real tensors"}:')
        ok &= report('joints-only kernel, joints[:, :22]', hj[:, :22], ref.joints[:, :22])
        ok &= report('LBS path, joints[:, :55]', hv.joints.cpu()[:, :55], ref.joints[:, :55])
        ok &= report('LBS path, vertices', hv.vertices.cpu(), ref.vertices)
        checked += 3
        from rohm_amd import _lib
        from rohm_amd.body_model import native_for
        mode = _lib.lib().rohm_smplx_skinning_mode(native_for(layer, torch.device('cuda', 0)).handle)
        zeros_frac = float((tensors['lbs_weights'] == 0).float().mean())
        print(f'  skinning path taken: {({0: "dense (MFMA)", 1: "sparse (ELL rows)", 2: "ELL rows of all joints"}).get(mode, mode)}; '
              f'{zeros_frac:.1%} of lbs_weights are zero')
    else:
        print('no AMD GPU visible: the HIP kernels were not exercised')
    print(f'{checked} comparisons, {"all within" if ok else "NOT all within"} {TOL} m')
    return 0 if ok else 1


if __name__ == '__main__':
    sys.exit(main())
