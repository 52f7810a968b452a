"""CPU: the numpy restatement of the AMASS evaluation metrics against the values the reference's own statements
(eval_amass_full.py:67-148, executed from the reference file by oracle/make_golden.py) produced."""
import numpy as np
import pytest

from helpers import golden
from oracle import metrics as M


@pytest.mark.parametrize('scheme,ratio', [('lower', 0.0), ('full', 0.1)])
def test_metrics_match_reference(scheme, ratio):
    g = golden('metrics.npz')
    clean, rec, r_clean, r_rec = M.synthetic_results(int(g['results_seed']))
    out = M.amass_metrics(clean, rec, r_clean, r_rec, scheme, ratio)
    for k, v in out.items():
        assert np.float64(v) == g[f'{scheme}_{k}'], k
    assert 0.0 < out['skating_gt_ratio'] < out['skating_rec_ratio'] < 1.0 and out['ground_pene_dist'] < 0.0
