#!/usr/bin/env python
"""Per-chunk cost of the plane GEMM kernels: rohm_gemm_planes timed at M = 9216, N = 512 / 1536 for K = 512 .. 4096, two and
three planes; the slope over K is the steady-state time of one 32-wide K chunk (prologue, epilogue and launch drop out).
Run once per kernel selection: ROHM_PP_STREAM=0 (one workgroup per tile) and default (persistent stream kernel).  (Round 3 also
measured an eight-waves-along-N layout behind ROHM_PP_W8, since removed: profiles/r3_f_pp_w8.txt.)
usage (GPU box): python scripts/bench_pp.py"""
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from rohm_amd import ops  # noqa: E402


def timed(fn, n=30):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


def main():
    dev = torch.device('cuda', 0)
    res = {'env': {k: os.environ.get(k) for k in ('ROHM_PP_STREAM',)}, 'rows': []}
    M = 9216
    for N in (512, 1536):
        for nplane in (3, 2):
            ts = {}
            for K in (512, 1024, 2048, 4096):
                a = torch.randn(M, K, device=dev)
                w = torch.randn(N, K, device=dev) / K ** 0.5
                ap, wp = ops.planes_split(a, nplane), ops.planes_split(w, nplane)
                out = torch.empty(M, N, device=dev)
                bias = torch.zeros(N, device=dev)
                from rohm_amd._lib import check, lib, ptr, stream_ptr
                fn = lambda: check(lib().rohm_gemm_planes(ptr(ap), ptr(wp), ptr(out), N, None, M, N, K, ptr(bias), None, 0, 0, 1.0, 0.0, 0,
                                                          nplane, 0, stream_ptr(dev)), 'gemm')
                ts[K] = timed(fn)
            tiles_per_cu = (M // 144) * (N // 128) / 256.0
            slope = (ts[4096] - ts[1024]) / ((4096 - 1024) / 32) / tiles_per_cu
            res['rows'].append({'N': N, 'nplane': nplane, 'us': {str(k): round(v, 1) for k, v in ts.items()},
                                'us_per_chunk_per_tile': round(slope, 3), 'fixed_us': round(ts[1024] - slope * 32 * tiles_per_cu, 1)})
            print(res['rows'][-1], flush=True)
    print(json.dumps(res))


if __name__ == '__main__':
    main()
