"""GPU: dataset-side per-frame SMPL-X work through the C ABI, against the golden recorded from the
reference's own `update_globalRT_for_smplx` run frame by frame (tests/golden/frames.npz) and against the oracle."""
import numpy as np
import pytest
import torch

from helpers import golden
from oracle import frames as OF
from oracle import geometry as G
from oracle.make_golden import frames_inputs
from rohm_amd.utils import synth

pytestmark = pytest.mark.gpu
DEV = 'cuda:0'


def _layer(seed=0):
    from rohm_amd.body_model import SMPLXLayer
    return SMPLXLayer.from_tensors(synth.synthetic_smplx_tensors(seed)).to(DEV)


def test_frames_to_world_vs_reference_golden():
    from rohm_amd.data_loaders.frames import frames_to_world
    g = golden('frames.npz')
    params, c2w = frames_inputs(int(g['seed']))
    joints, world = frames_to_world(_layer(int(g['body_seed'])), params, c2w)
    assert joints.shape == (40, 22, 3) and world.shape == (40, 79) and world.dtype == torch.float64
    assert np.abs(joints.cpu().numpy() - g['joints_world']).max() < 5e-6            # metres (fp32 FK on both sides)
    # orientation / translation: the rigid part is float64 on both sides, delta_T comes from the fp32 FK
    assert np.abs(world.cpu().numpy() - g['smplx_world']).max() < 5e-6
    assert np.array_equal(world[:, 6:].cpu().numpy(), g['smplx_world'][:, 6:])      # betas / body_pose pass through


def test_frames_to_world_vs_oracle_large_and_edge_rotations():
    from rohm_amd.data_loaders.frames import frames_to_world
    params, c2w = frames_inputs(3, N=3000)
    rng = np.random.Generator(np.random.PCG64(4))
    # rotations whose product with cam2world lands near pi and near the identity (every quaternion-extraction branch)
    from scipy.spatial.transform import Rotation as R
    Rc = R.from_matrix(c2w[:3, :3].astype(np.float64))
    inv = (Rc.inv() * R.from_rotvec(rng.standard_normal((100, 3)) * 1e-6)).as_rotvec()
    params['global_orient'][100:200] = inv.astype(np.float32)
    flip = (Rc.inv() * R.from_rotvec(np.pi * np.eye(3)[rng.integers(0, 3, 100)] + rng.standard_normal((100, 3)) * 1e-4)).as_rotvec()
    params['global_orient'][200:300] = flip.astype(np.float32)
    body = G.BodyModel(synth.synthetic_smplx_tensors(0))
    ref_j, ref_w = OF.frames_to_world(body, params, c2w)
    joints, world = frames_to_world(_layer(0), params, c2w)
    assert np.abs(joints.cpu().numpy() - ref_j).max() < 1e-5
    got = world.cpu().numpy()
    assert np.abs(got[:, 3:6] - ref_w[:, 3:6]).max() < 1e-5
    # axis-angle is double-valued at pi: compare the rotations, not the vectors
    d = (R.from_rotvec(got[:, 0:3]).inv() * R.from_rotvec(ref_w[:, 0:3])).magnitude()
    assert d.max() < 1e-6


def test_noisy_clip_joints_vs_oracle():
    from rohm_amd.data_loaders.frames import noisy_clip_joints
    params, _ = frames_inputs(5, N=145)
    p = dict(params)
    p['body_pose'] = params['body_pose'].reshape(-1, 21, 3)
    ref = OF.noisy_clip_joints(G.BodyModel(synth.synthetic_smplx_tensors(0)), {k: v.reshape(len(v), -1) for k, v in p.items()})
    out = noisy_clip_joints(_layer(0), p)
    assert out.shape == (145, 22, 3) and np.abs(out.cpu().numpy() - ref).max() < 5e-6


def test_validate_smplx_script_on_an_npz_of_the_real_layout(tmp_path):
    """scripts/validate_smplx.py (the S1 pin for machines that have SMPL-X) runs end to end on a model file with the
    real SMPLX_NEUTRAL.npz layout (400 shape components, posedirs [V,3,486], kintree_table, weights)."""
    import os
    import subprocess
    import sys
    t = synth.synthetic_smplx_tensors(0)
    V = t['v_template'].shape[0]
    sd = np.zeros((V, 3, 400), np.float32)
    sd[:, :, :10], sd[:, :, 300:310] = t['shapedirs'][:, :, :10].numpy(), t['shapedirs'][:, :, 10:].numpy()
    kt = np.stack([np.array(synth.SMPLX_PARENTS), np.arange(55)]).astype(np.int64)
    kt[0, 0] = 2 ** 32 - 1
    path = str(tmp_path / 'SMPLX_NEUTRAL.npz')
    np.savez(path, v_template=t['v_template'].numpy(), shapedirs=sd, posedirs=t['posedirs'].numpy().T.reshape(V, 3, 486),
             J_regressor=t['J_regressor'].numpy(), kintree_table=kt, weights=t['lbs_weights'].numpy(),
             f=np.zeros((4, 3), np.int64))
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, 'scripts', 'validate_smplx.py'), '--npz', path, '--frames', '16'],
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout + r.stderr
    assert '3 comparisons, all within' in r.stdout and 'MISMATCH' not in r.stdout
