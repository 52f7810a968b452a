"""GPU: the opt-in split-bf16 GEMM modes (ROHM_GEMM_PRECISION, read once per process -> exercised in a subprocess).
bf16x6 (six bf16 MFMA products per fp32 product) must meet the SAME bars as the exact-fp32 path on the GEMM unit tests
and the reference-golden PoseNet forward; bf16x3 must stay within the north-star tolerance on the forward."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(mode, args):
    env = dict(os.environ, ROHM_GEMM_PRECISION=mode)
    r = subprocess.run([sys.executable, '-m', 'pytest', '-q', '-x', '-m', 'gpu', '-p', 'no:cacheprovider'] + args, cwd=ROOT,
                       env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-1000:]
    return r.stdout


def test_bf16x6_meets_the_fp32_bars():
    out = _run('bf16x6', ['tests/test_gpu_kernels.py', '-k', 'gemm', 'tests/test_gpu_posenet.py::test_forward_vs_reference_golden',
                          'tests/test_gpu_posenet.py::test_loop8_vs_reference_golden_fused_and_stepwise'])
    assert ' passed' in out


def test_bf16x3_meets_the_north_star_tolerance_on_the_forward():
    out = _run('bf16x3', ['tests/test_gpu_posenet.py::test_forward_vs_reference_golden',
                          'tests/test_gpu_posenet.py::test_loop8_vs_reference_golden_fused_and_stepwise'])
    assert ' passed' in out
