"""CPU: the oracle's networks + sampler + driver-loop restatement, FREE-RUNNING end to end, against the reference
scripts' own loop text executed with the reference's OWN networks and samplers (tests/golden/scheme_real.npz, written by
oracle/make_golden.py::golden_scheme_real: test_amass_full.py:217-384 / test_prox_egobody.py:214-324, B = 2).

Cases: AMASS two-iteration scheme (BASELINE configs[2] shape), PROX / EgoBody three-iteration scheme with visibility mask
and early_stop (configs[4] shape), and a PROX scheme whose PoseNet stage is the reference's guided step over the stable
head t = 103 .. 97 at the reference's guidance weights (configs[3] / [4] guidance)."""
import numpy as np
import pytest
import torch

from helpers import cpu_noise_stream, golden, max_abs
from oracle import diffusion as odiff
from oracle import geometry as G
from oracle import scheme as OS
from oracle.make_golden import SCHEME_REAL_CAM_T, SCHEME_REAL_CASES, SCHEME_REAL_HEAD_T, SCHEME_REAL_SEEDS, scheme_real_case
from rohm_amd.utils import synth


def oracle_free_run(ci):
    kind, kw, pose_steps = SCHEME_REAL_CASES[ci]
    args, tfd, body_t, s_traj, s_pose, bt, bp, cam, _, plan = scheme_real_case(ci)
    noise = cpu_noise_stream(SCHEME_REAL_SEEDS['noise'] + ci, plan)
    sd_t = synth.trajnet_state_dict(SCHEME_REAL_SEEDS['trajnet'], trajcontrol=False)
    sd_c = synth.trajnet_state_dict(SCHEME_REAL_SEEDS['control'], trajcontrol=True)
    sd_p = synth.posenet_state_dict(SCHEME_REAL_SEEDS['posenet'])
    body = G.BodyModel(body_t)
    if pose_steps == 'head':
        tab_p, idx_p = odiff.tables(odiff.cosine_betas(1000)), list(SCHEME_REAL_HEAD_T)
        camera = dict(cam, cam_R=synth.SYNTH_CAM_R, cam_t=SCHEME_REAL_CAM_T)
    else:
        tab_p, idx_p, camera = odiff.tables(odiff.cosine_betas(pose_steps)), list(range(pose_steps))[::-1], None
    stages = OS.oracle_stages(sd_t, sd_c, sd_p, odiff.tables(odiff.cosine_betas(100)), tab_p, list(range(100))[::-1],
                              idx_p, s_pose, body, args, noise, grad_type='amass' if kind == 'amass' else 'prox',
                              camera=camera)
    outs = []
    traj_stage = lambda it, b: (outs.append(stages[0](it, b).clone()), outs[-1].clone())[1]      # the loops write into views of these
    pose_stage = lambda it, b: (outs.append(stages[1](it, b).clone()), outs[-1].clone())[1]
    fn = OS.amass_iterations if kind == 'amass' else OS.prox_iterations
    pose, traj, recs = fn(traj_stage, pose_stage, {k: v.clone() for k, v in bt.items()},
                          {k: v.clone() for k, v in bp.items()}, s_traj, s_pose, body, args)
    return pose, traj, recs, outs


# cases 3, 4 (the drivers' real step counts, 2000-3000 PoseNet steps on the CPU oracle) are held by the GPU test only: minutes of CPU here
@pytest.mark.parametrize('ci', range(3))
def test_free_running_scheme_vs_reference(ci):
    g = golden('scheme_real.npz')
    assert int(g['n_cases']) == len(SCHEME_REAL_CASES)
    pre = f'case{ci}_'
    torch.set_num_threads(min(8, torch.get_num_threads()))
    pose, traj, recs, outs = oracle_free_run(ci)
    assert len(outs) == int(g[pre + 'n_stages'])
    errs = [max_abs(o, torch.from_numpy(g[pre + f'stage{k}_out'])) for k, o in enumerate(outs)]
    print('stage errors', ['%.2e' % e for e in errs])
    assert max_abs(traj, torch.from_numpy(g[pre + 'traj'])) < 1e-3
    assert max_abs(recs[-1], torch.from_numpy(g[pre + 'traj_rec_full'])) < 1e-3
    assert max_abs(pose, torch.from_numpy(g[pre + 'pose'])) < 1e-3, errs
