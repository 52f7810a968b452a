"""CPU: the vectorised condition / mask assembly helpers of rohm_amd.inference against the
loop-for-loop restatement of the driver statements in oracle/scheme.py (test_amass_full.py:218-372)."""
import pytest
import torch

from helpers import seeded
from oracle import scheme as OS
from rohm_amd import inference as INF


def test_infill_mask_and_merge():
    m = INF.traj_infill_mask(3, 144, 0.1, 13, 'cpu')
    assert torch.equal(m, OS.infill_mask(3, 144, 0.1, 13))
    assert float(m[:, 65:79].abs().max()) == 0.0 and float(m[:, 79:].min()) == 1.0 and float(m[:, :65].min()) == 1.0
    rep, traj13, traj22 = seeded(1, 3, 144, 294), seeded(2, 3, 144, 13), seeded(3, 3, 144, 22)
    assert torch.equal(INF.merge_traj_into_repr(rep, traj13, True, 13), OS.merge_traj(rep, traj13, True, 13))
    assert torch.equal(INF.merge_traj_into_repr(rep, traj22, False, 22), OS.merge_traj(rep, traj22, False, 22))
    out = INF.merge_traj_into_repr(rep, traj13, True, 13)
    untouched = [c for c in range(294) if c not in OS.ABS_TRAJ_CH]
    assert torch.equal(out[..., untouched], rep[..., untouched])


def test_control_cond():
    pose = seeded(4, 2, 294, 1, 143)
    cc = INF.build_control_cond(pose, 144, 272)
    ref = torch.zeros(2, 144, 272)
    ref[:, 0:-1] = pose[:, :, 0].permute(0, 2, 1)[:, :, -272:]
    ref[:, -1] = ref[:, -2].clone()
    assert torch.equal(cc, ref)


@pytest.mark.parametrize('scheme', ['lower', 'upper', 'full'])
def test_occlusion_masks(scheme):
    cond = seeded(5, 4, 143, 294)
    start = torch.tensor([0, 17, 120, 142])
    end = torch.clamp(start + 30, max=143)
    a = INF.apply_occlusion_mask(cond.clone(), scheme, 22, start, end)
    b = OS.occlusion_mask(cond.clone(), scheme, 22, start, end)
    assert torch.equal(a, b)
    assert float(a[:, :, -4:].abs().max()) == 0.0
    assert torch.equal(a[:, :, :22], cond[:, :, :22])            # the trajectory channels are never masked
    with pytest.raises(ValueError):
        INF.apply_occlusion_mask(cond.clone(), 'left', 22)


def test_visibility_mask():
    cond, vis = seeded(6, 2, 143, 294), (seeded(7, 2, 145, 294) > 0).float()
    out = INF.apply_visibility_mask(cond.clone(), vis)
    ref = cond * vis[:, 0:-2, :]
    ref[:, :, -4:] = 0.
    assert torch.equal(out, ref)
