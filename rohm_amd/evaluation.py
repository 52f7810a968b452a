"""Evaluation metrics of the AMASS driver on the device (eval_amass_full.py:67-147).

`amass_metrics` takes what test_amass_full.py:387-429 produces (recovered joints of the clean clips and of the
reconstruction, the de-normalised representations) as device tensors and returns the quantities the evaluation
script prints, in its units.  One kernel launch (`rohm_amass_metrics`), one small D2H copy of the per-clip sums.
"""
from __future__ import annotations

import torch

from . import _lib
from ._lib import check, lib, ptr, stream_ptr

LOWER_JOINTS = (1, 2, 4, 5, 7, 8, 10, 11)          # eval_amass_full.py:76


def amass_metrics(joints_clean, joints_rec, repr_clean, repr_rec, mask_scheme='lower', traj_mask_ratio=0.0):
    """joints_*: [n_seq, clip_len, 22, 3]; repr_*: [n_seq, clip_len, 294] de-normalised.  Returns a dict with the
    script's names and units: mpjpe_global[_vis|_occ] (mm), contact_lbl_acc, skating_gt_ratio, skating_rec_ratio,
    accel_error (m/s^2), ground_pene_freq (%), ground_pene_dist (mm)."""
    for t in (joints_clean, joints_rec, repr_clean, repr_rec):
        _lib.require_hip(t)
    jc, jr = joints_clean.float().contiguous(), joints_rec.float().contiguous()
    if jc.shape != jr.shape or jc.dim() != 4 or jc.shape[2:] != (22, 3):
        raise ValueError(f'joints must both be [n_seq, clip_len, 22, 3], got {tuple(jc.shape)} / {tuple(jr.shape)}')
    n, T = jc.shape[:2]
    rc, rr = repr_clean.float().contiguous(), repr_rec.float().contiguous()
    if rc.shape != (n, T, 294) or rr.shape != (n, T, 294):
        raise ValueError('representations must be [n_seq, clip_len, 294]')
    mask, start, end = 0, 0, 0
    if mask_scheme == 'lower':
        for j in LOWER_JOINTS:
            mask |= 1 << j
    elif mask_scheme == 'full':
        start = 65
        end = start + int(traj_mask_ratio * 145)
    else:
        raise ValueError(f'unknown mask_scheme {mask_scheme!r}')
    out = torch.empty(n, 10, device=jc.device, dtype=torch.float64)
    check(lib().rohm_amass_metrics(ptr(jc), ptr(jr), rc.data_ptr() + 290 * 4, 294, rr.data_ptr() + 290 * 4, 294, mask,
                                   start, end, n, T, ptr(out), stream_ptr(jc.device)), 'rohm_amass_metrics')
    s = out.sum(dim=0).cpu().tolist()
    tot = n * T * 22
    n_occ = s[2]
    res = {'mpjpe_global': s[0] / tot * 1000.0,
           'mpjpe_global_vis': (s[0] - s[1]) / max(tot - n_occ, 1.0) * 1000.0,
           'mpjpe_global_occ': s[1] / max(n_occ, 1.0) * 1000.0,
           'contact_lbl_acc': s[3] / (n * T * 4),
           'skating_gt_ratio': s[4] / (n * (T - 1)),
           'skating_rec_ratio': s[5] / (n * (T - 1)),
           'accel_error': s[6] / (n * (T - 2) * 22),
           'ground_pene_freq': s[7] / (n * T * 2) * 100.0,
           'ground_pene_dist': s[8] / (n * T * 2) * 1000.0}
    return res
