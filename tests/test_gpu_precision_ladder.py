"""GPU: the opt-in split-bf16 mode of PoseNet (ROHM_GEMM_PRECISION, read when the native handle is created -> exercised in a
subprocess) against the SAME tests and the SAME bars as the exact-fp32 default.

bf16x6 (three bf16 planes, six MFMA products per fp32 product) and fp16x3 (two fp16 planes, three products, cross terms in a
scaled second accumulator) must each pass the whole PoseNet file: the reference-golden forward, the fused and step-wise 8-step
loops, ALL 1000 ancestral steps of one clip against the reference's own run (posenet_loop1000.npz), and the headline batch
(64 clips x 1000 steps, clip 0 on the reference golden, bit-identical on re-run).  bf16x3 (two bf16 planes) is held to the
north-star tolerance on the forward, the 8-step loops and the 1000-step run.
The kernels of the mode are tested one by one in tests/test_gpu_planes.py."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PN = 'tests/test_gpu_posenet.py'


def _run(mode, args, timeout=1500, **extra):
    env = dict(os.environ, ROHM_GEMM_PRECISION=mode, ROHM_EXPECT_GEMM_PRECISION=mode, **extra)
    r = subprocess.run([sys.executable, '-m', 'pytest', '-q', '-x', '-m', 'gpu', '-p', 'no:cacheprovider'] + args, cwd=ROOT,
                       env=env, capture_output=True, text=True, timeout=timeout)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-1000:]
    return r.stdout


def test_bf16x6_passes_the_whole_posenet_suite_at_the_fp32_bars():
    out = _run('bf16x6', [PN, 'tests/test_gpu_precision_ladder.py::test_mode_is_active'])
    assert ' passed' in out and 'failed' not in out


def test_fp16x3_passes_the_whole_posenet_suite_at_the_fp32_bars():
    out = _run('fp16x3', [PN, 'tests/test_gpu_precision_ladder.py::test_mode_is_active'])
    assert ' passed' in out and 'failed' not in out


def test_bf16x3_meets_the_north_star_tolerance():
    out = _run('bf16x3', [PN + '::test_forward_vs_reference_golden', PN + '::test_loop8_vs_reference_golden_fused_and_stepwise',
                          PN + '::test_full_1000_step_loop_vs_reference_golden',
                          'tests/test_gpu_precision_ladder.py::test_mode_is_active'], ROHM_TEST_FORWARD_BAR='1e-3')
    assert ' passed' in out and 'failed' not in out


def test_mode_is_active():
    """Inside the subprocess: the handle really runs the split-bf16 GEMMs (a silently ignored variable would make the two
    tests above vacuous).  In the parent process (no variable) the default must be exact fp32."""
    import torch
    from helpers import PoseDataset
    from rohm_amd.model.posenet import PoseNet
    from rohm_amd.utils import synth
    net = PoseNet(PoseDataset(), 294, latent_dim=512, ff_size=1024, num_layers=8, num_heads=4, traj_feat_dim=22,
                  body_model_path=torch.nn.Identity(), device='cuda:0')
    net.load_state_dict(synth.posenet_state_dict(0), strict=True)
    net = net.to('cuda:0').eval()
    net.native()
    want = os.environ.get('ROHM_EXPECT_GEMM_PRECISION', 'fp32')
    assert net.gemm_precision == want, (net.gemm_precision, want)
    assert os.environ.get('ROHM_GEMM_PRECISION', 'fp32') == want


def test_unknown_precision_is_refused(monkeypatch):
    import torch
    from helpers import PoseDataset
    from rohm_amd import _lib
    from rohm_amd.model.posenet import PoseNet
    from rohm_amd.utils import synth
    monkeypatch.setenv('ROHM_GEMM_PRECISION', 'fp8')
    net = PoseNet(PoseDataset(), 294, latent_dim=512, ff_size=1024, num_layers=8, num_heads=4, traj_feat_dim=22,
                  body_model_path=torch.nn.Identity(), device='cuda:0')
    net.load_state_dict(synth.posenet_state_dict(0), strict=True)
    with pytest.raises(_lib.RohmHipError):
        net.to('cuda:0').native()
