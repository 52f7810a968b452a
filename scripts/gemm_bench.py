#!/usr/bin/env python
"""Microbenchmark of rohm_gemm_f32 over the PoseNet GEMM shapes (B = 64 -> M = 9216).
ROHM_GEMM_VARIANT selects diagnostic schedules (see gemm_f32.hip); run through scripts/gemm_sweep.sh."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rohm_amd import ops  # noqa: E402

SHAPES = [('outproj', 9216, 512, 512), ('qkv', 9216, 1536, 512), ('ff1', 9216, 1024, 512),
          ('ff2', 9216, 512, 1024), ('outproj_b32', 4608, 512, 512)]
if os.environ.get('GEMM_SHAPES') == 'n':
    SHAPES = [(f'n{n}', 9216, n, 512) for n in (512, 1024, 1536, 2048, 3072, 4096)]


def main():
    dev = torch.device('cuda', 0)
    var = os.environ.get('ROHM_GEMM_VARIANT', '0')
    res = []
    for name, M, N, K in SHAPES:
        a = torch.randn(M, K, device=dev)
        w = torch.randn(N, K, device=dev) / K ** 0.5
        if os.environ.get('GEMM_ZERO'):
            a.zero_(); w.zero_()      # DVFS probe: same instruction stream, no toggling data
        b = torch.randn(N, device=dev)
        out = torch.empty(M, N, device=dev)
        for _ in range(5):
            ops.gemm(a, w, b, out=out)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        n = 50
        e0.record()
        for _ in range(n):
            ops.gemm(a, w, b, out=out)
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) / n * 1e3
        res.append(f'{name}:{us:7.1f}us {2 * M * N * K / us / 1e6:6.1f}TF')
    print(f'variant {var:>4s} | ' + ' | '.join(res), flush=True)


if __name__ == '__main__':
    main()
