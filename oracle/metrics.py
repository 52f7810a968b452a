"""TEST INFRASTRUCTURE ONLY: numpy restatement of the AMASS evaluation metrics (eval_amass_full.py:67-147): global
MPJPE (all / visible / occluded), contact-label accuracy, foot-skating ratios, acceleration error, ground
penetration.  Pinned against the reference's own statements, executed from its file on synthetic results by
oracle/make_golden.py -> tests/golden/metrics.npz (bit-exact)."""
import numpy as np

LOWER_JOINTS = [1, 2, 4, 5, 7, 8, 10, 11]        # eval_amass_full.py:76
FOOT = [7, 10, 8, 11]                            # :103
FPS = 30


def _skating(feet, min_h, thresh_h=0.10, thresh_v=0.10):
    """:105-133 -- both joints of BOTH feet fast and low."""
    vel = np.linalg.norm(feet[:, 1:, :, [0, 1]] - feet[:, :-1, :, [0, 1]], axis=-1) * FPS
    h = feet[:, 0:-1, :, 2] - np.tile(min_h.reshape(len(min_h), 1, 1), (1, feet.shape[1] - 1, 4))
    left = (vel[:, :, 0] > thresh_v) * (vel[:, :, 1] > thresh_v) * (h[:, :, 0] < (thresh_h + 0.05)) * (h[:, :, 1] < thresh_h)
    right = (vel[:, :, 2] > thresh_v) * (vel[:, :, 3] > thresh_v) * (h[:, :, 2] < (thresh_h + 0.05)) * (h[:, :, 3] < thresh_h)
    return np.mean(left * right)


def amass_metrics(joints_clean, joints_rec, repr_clean, repr_rec, mask_scheme='lower', traj_mask_ratio=0.0):
    """joints_*: [n_seq, clip_len, 22, 3] float32; repr_*: [n_seq, clip_len, 294] de-normalised.  Returns a dict of
    the quantities the script prints (before its unit scaling / rounding)."""
    n_seq, clip_len = joints_clean.shape[:2]
    out = {}
    err = np.linalg.norm(joints_clean - joints_rec, axis=-1)
    out['mpjpe_global'] = np.mean(err)
    if mask_scheme == 'lower':
        vis = sorted(set(range(22)) - set(LOWER_JOINTS))
        out['mpjpe_global_vis'], out['mpjpe_global_occ'] = np.mean(err[:, :, vis]), np.mean(err[:, :, LOWER_JOINTS])
    elif mask_scheme == 'full':
        start = 65
        end = start + int(traj_mask_ratio * 145)
        out['mpjpe_global_vis'] = np.mean(np.concatenate([err[:, 0:start], err[:, end:]], axis=1))
        out['mpjpe_global_occ'] = np.mean(err[:, start:end])
    c_rec = np.where(repr_rec[:, :, -4:] > 0.5, 1.0, 0.0).astype(repr_rec.dtype)
    out['contact_lbl_acc'] = np.mean(repr_clean[:, :, -4:] == c_rec)
    min_h = joints_clean[:, :, :, 2].min(axis=-1).min(axis=-1)
    out['skating_gt_ratio'] = _skating(joints_clean[:, :, FOOT, :], min_h)
    out['skating_rec_ratio'] = _skating(joints_rec[:, :, FOOT, :], min_h)
    acc = lambda j: (j[:, 2:] - 2 * j[:, 1:-1] + j[:, :-2]) * (FPS ** 2)
    out['accel_error'] = np.linalg.norm(acc(joints_rec) - acc(joints_clean), axis=-1).mean()
    pene = joints_rec[:, :, [10, 11], -1] - np.tile(min_h.reshape(n_seq, 1, 1), (1, clip_len, 2))
    out['ground_pene_freq'] = (pene < -0.05).mean()
    pene = pene.copy()
    pene[pene >= 0] = 0
    out['ground_pene_dist'] = pene.mean()
    return out


def synthetic_results(seed, n_seq=6, clip_len=143):
    """Joint tracks / representations with enough slow-low and fast-low feet that every metric is exercised."""
    g = np.random.Generator(np.random.PCG64(seed))
    base = np.cumsum(g.standard_normal((n_seq, clip_len, 1, 3)) * 0.004, axis=1)
    clean = (base + g.standard_normal((n_seq, 1, 22, 3)) * 0.25 + g.standard_normal((n_seq, clip_len, 22, 3)) * 0.002)
    clean[..., 2] = np.abs(clean[..., 2]) + 0.02
    clean[:, :, FOOT, 2] = g.uniform(0.0, 0.2, size=(n_seq, clip_len, 4))
    rec = clean + g.standard_normal(clean.shape) * 0.02
    rec[:, :, [10, 11], 2] -= g.uniform(0.0, 0.12, size=(n_seq, clip_len, 2))
    repr_clean = g.standard_normal((n_seq, clip_len, 294))
    repr_clean[:, :, -4:] = (g.uniform(size=(n_seq, clip_len, 4)) > 0.5)
    repr_rec = repr_clean + g.standard_normal(repr_clean.shape) * 0.3
    f = lambda a: np.ascontiguousarray(a, dtype=np.float32)
    return f(clean), f(rec), f(repr_clean), f(repr_rec)
