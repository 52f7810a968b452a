"""CPU: oracle/scheme.py (the restatement of the drivers' inference-iteration loops) against what the REFERENCE scripts'
own statements did -- tests/golden/scheme.npz holds digests recorded while oracle/make_golden.py::golden_scheme EXECUTED
test_amass_full.py:217-384 and test_prox_egobody.py:214-324 with stub samplers."""
import numpy as np
import pytest
import torch

from helpers import golden
from oracle import geometry as G
from oracle import scheme as OS
from oracle.make_golden import SCHEME_CASES, digest, exact_hash, scheme_case


def run_oracle_case(ci, g, iterations=None):
    kind, kw = SCHEME_CASES[ci]
    args, tfd, body_t, s_traj, s_pose, bt, bp, traj_out, pose_out = scheme_case(kind, kw)
    pre = f'case{ci}_'
    if pre + 'full_mask_start' in g:
        args.full_mask_start = [torch.from_numpy(r) for r in g[pre + 'full_mask_start']]      # one draw per iteration
    log = []

    def traj_stage(it, batch):
        log.append(('traj', {k: batch[k].clone() for k in ('cond', 'control_cond') if k in batch and (k == 'cond' or it > 0)}))
        return traj_out[it]

    def pose_stage(it, batch):
        log.append(('pose', {'cond': batch['cond'].clone()}))
        return pose_out[it]
    fn = iterations or (OS.amass_iterations if kind == 'amass' else OS.prox_iterations)
    tb, pb = {k: v.clone() for k, v in bt.items()}, {k: v.clone() for k, v in bp.items()}
    _, _, recs = fn(traj_stage, pose_stage, tb, pb, s_traj, s_pose, G.BodyModel(body_t), args)
    return log, recs, tb, pb


def close(a, b, tol=1e-5, report=None):
    """Digests agree: a projection of element-wise noise of relative size `tol` has magnitude ~ tol * mean|x| * sqrt(n)."""
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    assert a[0] == b[0], 'size'
    thr = tol * max(b[2] / b[0], 1e-3) * np.sqrt(b[0])
    if report is not None:
        report.append(float(np.abs(a[1:] - b[1:]).max() / thr * tol))
    return bool(np.all(np.abs(a[1:] - b[1:]) <= thr))


def check_full_entries(g, prefix, key, v, traj_check=None):
    """Element-wise comparison with what the reference scripts' statements produced (scheme.npz stores COMPUTED tensors in
    full and bit-exact COPIES as a hash next to the digests): TrajNet's cond exactly; PoseNet's cond = channels 0..21 (the
    re-derived trajectory, `traj_check(ours [B,T,22], ref [B,T,22])`, default max-abs 2e-5) + an exact hash of the rest."""
    v = v.detach().cpu().float()
    if key == 'control_cond':
        assert exact_hash(v) == str(g[prefix + '_sha']), prefix
    elif v.dim() == 4:
        ours, ref = v[:, 0:22, 0].permute(0, 2, 1), torch.from_numpy(g[prefix + '_full_traj'])[:, :, 0].permute(0, 2, 1)
        if traj_check is None:
            assert float((ours.double() - ref.double()).abs().max()) < 2e-5, prefix
        else:
            traj_check(ours, ref)
        assert exact_hash(v[:, 22:]) == str(g[prefix + '_sha_rest']), prefix
    else:
        assert float((v.double() - torch.from_numpy(g[prefix + '_full']).double()).abs().max()) == 0.0, prefix


@pytest.mark.parametrize('ci', range(len(SCHEME_CASES)))
def test_driver_loop_matches_reference_script(ci):
    g = golden('scheme.npz')
    pre = f'case{ci}_'
    log, recs, tb, pb = run_oracle_case(ci, g)
    assert len(log) == int(g[pre + 'n_calls'])
    for k, (name, tens) in enumerate(log):
        assert name == str(g[pre + f'call{k}_name'])
        ref_keys = {kk for kk in ('cond', 'control_cond') if pre + f'call{k}_{kk}' in g}
        assert set(tens) == ref_keys, (k, name, set(tens), ref_keys)
        for kk, v in tens.items():
            assert list(v.shape) == list(g[pre + f'call{k}_{kk}_shape']), (k, kk)
            assert close(digest(v), g[pre + f'call{k}_{kk}']), (k, name, kk)
            check_full_entries(g, pre + f'call{k}_{kk}', kk, v)
    assert close(digest(recs[-1]), g[pre + 'traj_rec_full'])
    assert float((recs[-1].double() - torch.from_numpy(g[pre + 'traj_rec_full_full']).double()).abs().max()) < 1e-6
    assert exact_hash(tb['motion_repr_noisy']) == str(g[pre + 'after_traj_noisy_sha'])
    assert float((tb['cond'] - torch.from_numpy(g[pre + 'after_traj_cond_full'])).abs().max()) == 0.0
    assert close(digest(tb['motion_repr_noisy']), g[pre + 'after_traj_noisy'])
    assert close(digest(tb['cond']), g[pre + 'after_traj_cond'])
    assert list(pb['motion_repr_noisy'].shape) == list(g[pre + 'after_pose_noisy_shape'])
    assert list(pb['motion_repr_clean'].shape) == list(g[pre + 'after_pose_clean_shape'])


def test_digest_detects_a_single_wrong_element():
    g = np.random.Generator(np.random.PCG64(1))
    x = torch.from_numpy(g.standard_normal((2, 294, 143)))
    y = x.clone()
    y[1, 100, 17] += 1e-2                                   # one element of 84 084 off by 1e-2
    assert not close(digest(y), digest(x))
    z = x.clone()
    z[0, 5, :] = 0.                                         # one masked channel row
    assert not close(digest(z), digest(x))
    assert close(digest(x * (1 + 1e-6 * torch.from_numpy(g.standard_normal(x.shape)))), digest(x))   # fp32-class rounding noise
