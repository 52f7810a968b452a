"""GPU parity of the between-stage trajectory re-derivation and the repr -> joints kernels (SURVEY.md §8(f) N1,
through the C ABI) against the reference's golden outputs (tests/golden/rederive.npz, guidance.npz) and the
oracle (oracle/rederive.py).  Outputs are normalised representation channels; tolerances are absolute."""
import numpy as np
import pytest
import torch

from helpers import golden, max_abs, seeded
from oracle import geometry as G
from oracle import rederive as RD
from rohm_amd.utils import synth

pytestmark = pytest.mark.gpu
DEV = 'cuda:0'


def _close(out, ref, joints, tol=2e-5):
    """Absolute tolerance `tol`, widened on the facing-direction channels (0, 1, 4, 5) in proportion to the
    conditioning of the reference's own float32 computation (oracle/rederive.py::facing_margin): a ~1e-7 m
    difference in a joint position moves the facing angle by ~1e-7 / |across_xy|, and qbetween's
    w = 1 + forward.y cancels when the body faces -y."""
    out, ref = np.asarray(out, np.float64), np.asarray(ref, np.float64)
    raw_xy, w = (np.nan_to_num(v, nan=0.0) for v in RD.facing_margin(joints))        # [B, T] each
    pair = lambda v: np.maximum(np.minimum(v[:, :-1], v[:, 1:]), 1e-12)              # frame t uses q[t] and q[t+1]
    raw_xy, w = pair(raw_xy), pair(w)
    err = np.abs(out - ref)
    lim = np.full(err.shape, tol)
    for c in (0, 1, 4, 5):
        lim[:, :, c] = tol + 3e-6 / raw_xy + 4e-6 / w
    assert (err <= lim).all(), f'max err {err.max():.3e}; outside tolerance at {np.argwhere(err > lim)[:5].tolist()}'
    return float(((raw_xy < 0.03) | (w < 1e-2)).mean())


def _layer(tensors=None):
    from rohm_amd.body_model import SMPLXLayer
    return SMPLXLayer.from_tensors(tensors or synth.synthetic_smplx_tensors(0)).to(DEV)


def test_rederive_vs_reference_golden():
    from rohm_amd.data_loaders.motion_representation import rederive_traj
    g = golden('rederive.npz')
    s_in, s_out = synth.synthetic_stats(int(g['stats_in_seed'])), synth.synthetic_stats(int(g['stats_out_seed']))
    rn = synth.plausible_motion(int(g['motion_seed']), 2, 144, *s_in)[:, :, 0].permute(0, 2, 1).contiguous()
    out = rederive_traj(rn.to(DEV), s_in, s_out, _layer())
    assert out.shape == (2, 143, 22) and out.dtype == torch.float32
    assert _close(out.cpu().numpy(), g['full_ref'][:, :, :22], g['joints']) < 0.2


@pytest.mark.parametrize('B,T', [(1, 144), (5, 144), (3, 31), (2, 300)])
def test_rederive_vs_oracle_shapes_and_layouts(B, T):
    from rohm_amd.data_loaders.motion_representation import rederive_traj
    t = synth.synthetic_smplx_tensors(0)
    s_in, s_out = synth.synthetic_stats(3), synth.synthetic_stats(4)
    x = synth.plausible_motion(100 + B + T, B, T, *s_in)                     # [B,294,1,T]
    rn = x[:, :, 0].permute(0, 2, 1).contiguous()
    ref, joints = RD.rederive_traj(rn, *s_in, *s_out, G.BodyModel(t), return_joints=True)
    layer = _layer(t)
    a = rederive_traj(rn.to(DEV), s_in, s_out, layer)
    _close(a.cpu().numpy(), ref, joints)
    # channel-major input, written straight into channels 0..21 of a PoseNet `cond` in both layouts
    cond = torch.full((B, 294, 1, T - 1), 7.0, device=DEV)
    r = rederive_traj(x.to(DEV), s_in, s_out, layer, out=cond, layout='bc1t', out_layout='bc1t')
    assert r is cond
    assert torch.equal(cond[:, :22, 0].permute(0, 2, 1), a)
    assert float((cond[:, 22:] - 7.0).abs().max()) == 0.0
    cond2 = torch.zeros(B, T - 1, 294, device=DEV)
    rederive_traj(rn.to(DEV), s_in, s_out, layer, out=cond2)
    assert torch.equal(cond2[:, :, :22], a) and float(cond2[:, :, 22:].abs().max()) == 0.0


def test_rederive_degenerate_facing_reproduces_reference_nan_patch():
    """Frames whose hips and shoulders are stacked along z have no facing direction: the reference produces a NaN
    quaternion and patches only the FIRST such frame (motion_representation.py:212-215)."""
    from rohm_amd.data_loaders.motion_representation import rederive_traj
    t = {k: v.clone() for k, v in synth.synthetic_smplx_tensors(0).items()}
    # pelvis -> spine -> collars -> shoulders and both hips sit ON the z axis (every link offset has x = y = 0 exactly)
    for j, z in ((0, 0.0), (1, -0.1), (2, -0.12), (3, 0.1), (6, 0.2), (9, 0.3), (13, 0.35), (14, 0.36), (16, 0.4),
                 (17, 0.45)):
        t['J_regressor'][j] = 0.0
        t['J_regressor'][j, 100 + j] = 1.0
        t['v_template'][100 + j] = torch.tensor([0.0, 0.0, z])
        t['shapedirs'][100 + j] = 0.0
    mean, std = np.zeros(294, np.float32), np.ones(294, np.float32)
    B, T = 2, 40
    x = synth.plausible_motion(9, B, T, mean, std)[:, :, 0].permute(0, 2, 1).contiguous()
    ident6 = torch.tensor([1., 0., 0., 1., 0., 0.])
    for (b, f) in ((0, 7), (0, 19), (1, 0), (1, 5)):                     # identity pose -> joints stay on the axis
        x[b, f, 7:13] = ident6
        x[b, f, 154:280] = ident6.repeat(21)
    ref, joints = RD.rederive_traj(x, mean, std, mean, std, G.BodyModel(t), return_joints=True)
    out = rederive_traj(x.to(DEV), (mean, std), (mean, std), _layer(t)).cpu().numpy()
    assert np.isnan(ref).any()
    assert np.array_equal(np.isnan(out), np.isnan(ref))
    ok = ~np.isnan(ref)
    with np.errstate(invalid='ignore'):
        _close(np.where(ok, out, 0.0), np.where(ok, ref, 0.0), np.nan_to_num(joints, nan=0.0))


def test_joints_from_repr_vs_golden_and_oracle():
    from rohm_amd.data_loaders.motion_representation import joints_from_repr, recover_from_repr_smpl
    g = golden('guidance.npz')
    mean, std = synth.synthetic_stats(int(g['stats_seed']))
    x0 = synth.plausible_motion(int(g['motion_seed']), 2, 143, mean, std).to(DEV)        # [2,294,1,143]
    layer = _layer()
    j_s = joints_from_repr(x0, 'smplx_params', layer, stats=(mean, std), layout='bc1t')
    j_a = joints_from_repr(x0, 'joint_abs_traj', stats=(mean, std), layout='bc1t')
    assert max_abs(j_s.cpu(), torch.from_numpy(g['j_smpl'])) < 1e-5
    assert max_abs(j_a.cpu(), torch.from_numpy(g['j_abs'])) < 1e-5
    j_r = joints_from_repr(x0, 'joint_rel_traj', stats=(mean, std), layout='bc1t')
    assert max_abs(j_r.cpu(), torch.from_numpy(golden('recover_rel.npz')['j_rel'])) < 2e-5      # 143-term running sums
    # the reference's dict-of-slices signature on de-normalised data
    full = x0[:, :, 0].permute(0, 2, 1) * torch.from_numpy(std).to(DEV) + torch.from_numpy(mean).to(DEV)
    d = G.split_repr(full)
    assert max_abs(recover_from_repr_smpl(d, 'smplx_params', layer), j_s) < 1e-6
    assert max_abs(recover_from_repr_smpl(d, 'joint_abs_traj'), j_a) < 1e-6
    assert max_abs(recover_from_repr_smpl(d, 'joint_rel_traj'), j_r) < 2e-5     # running sums amplify the 1-ulp de-normalisation difference
    # full LBS (return_verts=True, motion_representation.py:389-396) against the oracle body model
    jv, verts = recover_from_repr_smpl(d, 'smplx_params', layer, return_verts=True)
    dc = {k: v.cpu() for k, v in d.items()}
    jr, vr = G.joints_from_smplx(dc, G.BodyModel(synth.synthetic_smplx_tensors(0)), return_verts=True)
    assert verts.shape == (2, 143, 10475, 3)
    assert max_abs(jv.cpu(), jr) < 1e-5 and max_abs(verts.cpu(), vr) < 2e-5


def test_rederive_rejects_cpu_and_bad_shapes():
    from rohm_amd._lib import RohmHipError
    from rohm_amd.data_loaders.motion_representation import rederive_traj
    s = synth.synthetic_stats(0)
    with pytest.raises(RohmHipError):
        rederive_traj(torch.zeros(1, 144, 294), s, s, _layer())
    with pytest.raises(ValueError):
        rederive_traj(torch.zeros(1, 144, 100, device=DEV), s, s, _layer())
    with pytest.raises(ValueError):
        rederive_traj(torch.zeros(1, 144, 294, device=DEV), s, s, _layer(), out=torch.zeros(1, 144, 294, device=DEV))


@pytest.mark.parametrize('scheme,ratio', [('lower', 0.0), ('full', 0.1)])
def test_amass_metrics_vs_reference_golden(scheme, ratio):
    """Device evaluation metrics against what the reference's own statements computed (tests/golden/metrics.npz)."""
    from oracle import metrics as M
    from rohm_amd.evaluation import amass_metrics
    g = golden('metrics.npz')
    clean, rec, r_clean, r_rec = (torch.from_numpy(a).to(DEV) for a in M.synthetic_results(int(g['results_seed'])))
    out = amass_metrics(clean, rec, r_clean, r_rec, scheme, ratio)
    unit = {'mpjpe_global': 1000., 'mpjpe_global_vis': 1000., 'mpjpe_global_occ': 1000., 'ground_pene_freq': 100.,
            'ground_pene_dist': 1000.}
    for k, v in out.items():
        ref = float(g[f'{scheme}_{k}']) * unit.get(k, 1.0)
        assert abs(v - ref) <= 2e-6 * max(1.0, abs(ref)), (k, v, ref)


def test_lbs_vertices_vs_oracle_with_face_and_hands():
    """Body-model layer called like smplx (axis-angle, all 55 joints posed): `.vertices`, `.joints[:, :55]`."""
    t = synth.synthetic_smplx_tensors(0)
    layer, body = _layer(t), G.BodyModel(t)
    N = 37
    betas, go, bp, tr = seeded(1, N, 10), seeded(2, N, 3) * 0.8, seeded(3, N, 63) * 0.5, seeded(4, N, 3)
    jaw, le, re_ = seeded(5, N, 3) * 0.2, seeded(6, N, 3) * 0.1, seeded(7, N, 3) * 0.1
    lh, rh = seeded(8, N, 45) * 0.3, seeded(9, N, 45) * 0.3
    bp[:3] = 0.0
    ref = body(betas=betas, global_orient=go, body_pose=bp, transl=tr, jaw_pose=jaw, leye_pose=le, reye_pose=re_,
               left_hand_pose=lh, right_hand_pose=rh, return_verts=True)
    g = lambda a: a.to(DEV)
    out = layer(betas=g(betas), global_orient=g(go), body_pose=g(bp), transl=g(tr), jaw_pose=g(jaw), leye_pose=g(le),
                reye_pose=g(re_), left_hand_pose=g(lh), right_hand_pose=g(rh), expression=torch.zeros(N, 10, device=DEV),
                return_verts=True)
    assert out.vertices.shape == (N, 10475, 3) and out.joints.shape == (N, 127, 3)
    assert max_abs(out.vertices.cpu(), ref.vertices) < 2e-5
    assert max_abs(out.joints[:, :55].cpu(), ref.joints[:, :55]) < 1e-5
    # body-only call (the reference's call sites pass zeros for face and hands)
    ref2 = body(betas=betas, global_orient=go, body_pose=bp, transl=tr, return_verts=True)
    out2 = layer(betas=g(betas), global_orient=g(go), body_pose=g(bp), transl=g(tr), return_verts=True)
    assert max_abs(out2.vertices.cpu(), ref2.vertices) < 2e-5


def _sparse_model(t, keep=4):
    """The synthetic body model with released-SMPL-X-like skinning weights: the `keep` largest weights of every vertex,
    renormalised; everything else exactly zero (>= 75 % zeros -> the ELL path of rohm_smplx_set_skinning)."""
    t = {k: v.clone() for k, v in t.items()}
    w = t['lbs_weights']
    top = torch.topk(w, keep, dim=1)
    sw = torch.zeros_like(w).scatter_(1, top.indices, top.values)
    t['lbs_weights'] = sw / sw.sum(1, keepdim=True)
    return t


@pytest.mark.parametrize('mode', ['mfma', 'sparse', 'ell'])
@pytest.mark.parametrize('N', [1, 37, 143 * 2 + 5])
def test_lbs_skinning_paths_vs_oracle(mode, N, monkeypatch):
    """The three skinning kernels of rohm_smplx_forward against the oracle body model (smplx 0.1.28 `lbs` restated): dense
    weights on the matrix core (T = W . A as an fp32-MFMA GEMM with the 12-fma apply in its epilogue), sparse weights through
    per-vertex ELL rows (what a released SMPLX_*.npz looks like: chosen automatically at >= 75 % zeros), and the round-3 VALU form
    (ROHM_LBS_SKIN=ell).  Frame counts that are not multiples of 16 / 8 exercise the partial groups."""
    from rohm_amd._lib import lib
    from rohm_amd.body_model import native_for
    t = synth.synthetic_smplx_tensors(0)
    if mode == 'sparse':
        t = _sparse_model(t)
        assert float((t['lbs_weights'] == 0).float().mean()) >= 0.75
    if mode == 'ell':
        monkeypatch.setenv('ROHM_LBS_SKIN', 'ell')
    layer, body = _layer(t), G.BodyModel(t)
    assert lib().rohm_smplx_skinning_mode(native_for(layer, torch.device(DEV)).handle) == {'mfma': 0, 'sparse': 1, 'ell': 2}[mode]
    betas, go, bp, tr = seeded(11, N, 10), seeded(12, N, 3) * 0.8, seeded(13, N, 63) * 0.5, seeded(14, N, 3)
    ref = body(betas=betas, global_orient=go, body_pose=bp, transl=tr, return_verts=True)
    g = lambda a: a.to(DEV)
    out = layer(betas=g(betas), global_orient=g(go), body_pose=g(bp), transl=g(tr), return_verts=True)
    assert max_abs(out.vertices.cpu(), ref.vertices) < 2e-5
    assert max_abs(out.joints[:, :55].cpu(), ref.joints[:, :55]) < 1e-5


def test_lbs_zero_pose_invariants_on_the_sparse_path():
    """with zero pose the body model reduces to its linear part -- vertices = v_template + shapedirs . beta
    + transl and joints = J_regressor . v_shaped + transl -- for the sparse-weight path a real SMPLX_NEUTRAL.npz takes, so the
    first machine with the real package only needs scripts/validate_smplx.py."""
    t = _sparse_model(synth.synthetic_smplx_tensors(0))
    layer = _layer(t)
    N = 19
    betas, tr = seeded(21, N, 10), seeded(22, N, 3)
    z = torch.zeros(N, 3, device=DEV)
    out = layer(betas=betas.to(DEV), global_orient=z, body_pose=torch.zeros(N, 63, device=DEV), transl=tr.to(DEV), return_verts=True)
    v_shaped = t['v_template'][None].double() + torch.einsum('vck,nk->nvc', t['shapedirs'][:, :, :10].double(), betas.double())
    assert max_abs(out.vertices.cpu(), v_shaped + tr[:, None].double()) < 5e-6
    joints = torch.einsum('jv,nvc->njc', t['J_regressor'].double(), v_shaped) + tr[:, None].double()
    assert max_abs(out.joints[:, :55].cpu(), joints) < 5e-6


def test_repr_round_trip_on_the_device():
    """SURVEY §4 invariant (the author's debug note, dataloader_amass.py:230-236) through the HIP kernels: joints recovered
    by `rohm_repr_joints` from the representation that `get_repr_smplx` builds give the canonical joints back, for both
    recoveries; and re-deriving the trajectory from that representation (`rohm_traj_rederive`) reproduces its first 22
    channels."""
    from test_geometry_oracle import _round_trip_inputs
    from oracle import rederive as RD
    from rohm_amd.body_model import SMPLXLayer
    from rohm_amd.data_loaders.motion_representation import joints_from_repr
    params, joints, _ = _round_trip_inputs()
    full = torch.from_numpy(RD.full_repr(RD.get_repr_smplx(joints, params))).float()[None].to(DEV)      # [1, T-1, 294]
    layer = SMPLXLayer.from_tensors(synth.synthetic_smplx_tensors(0)).to(DEV)
    j_smpl = joints_from_repr(full, 'smplx_params', layer)[0].cpu().numpy()
    j_abs = joints_from_repr(full, 'joint_abs_traj')[0].cpu().numpy()
    assert abs(j_smpl - joints[:-1]).max() < 2e-5
    assert abs(j_abs - joints[:-1]).max() < 2e-5
