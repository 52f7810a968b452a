"""CPU: oracle/frames.py (dataset-side per-frame SMPL-X work) against the reference's own
`update_globalRT_for_smplx` run frame by frame (tests/golden/frames.npz)."""
import numpy as np

from helpers import golden
from oracle import frames as OF
from oracle import geometry as G
from oracle.make_golden import frames_inputs
from rohm_amd.utils import synth


def test_frames_to_world_matches_reference():
    g = golden('frames.npz')
    params, c2w = frames_inputs(int(g['seed']))
    body = G.BodyModel(synth.synthetic_smplx_tensors(int(g['body_seed'])))
    joints, world = OF.frames_to_world(body, params, c2w)
    assert world.dtype == np.float64 and world.shape == (40, 79)
    # batched vs per-frame body-model calls differ by fp32 summation order only
    assert np.abs(joints - g['joints_world']).max() < 2e-6
    assert np.abs(world - g['smplx_world']).max() < 2e-6
    # the rigid part alone (same delta_T): float64 scipy on both sides
    assert np.array_equal(world[:, 6:], g['smplx_world'][:, 6:])
