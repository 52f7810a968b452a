#!/usr/bin/env python
"""(Diagnostic GEMM variants: build the library with `ROHM_DIAG=1 python -m rohm_amd.build --force` first.)
Per-workgroup phase timeline of the GEMM kernel (diagnostic build path ROHM_GEMM_VARIANT=7 / 127):
every workgroup stamps the 100 MHz wall clock at entry, after its prologue landed, after the main loop, after the
epilogue stores were issued and after they were acknowledged.  Prints where one launch's time goes.
usage (GPU box): ROHM_GEMM_VARIANT=7 python scripts/gemm_timeline.py"""
import ctypes
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rohm_amd import _lib  # noqa: E402

SHAPES = [('outproj', 9216, 512, 512), ('qkv', 9216, 1536, 512), ('ff1', 9216, 1024, 512), ('ff2', 9216, 512, 1024),
          ('n4096', 9216, 4096, 512)]


def main():
    dev = torch.device('cuda', 0)
    lib = _lib.lib()
    stream = torch.cuda.current_stream().cuda_stream
    for name, M, N, K in SHAPES:
        zero = os.environ.get('ROHM_TL_ZERO') == '1'      # constant operands: the same instruction stream at lower switching power
        a = torch.zeros(M, K, device=dev) if zero else torch.randn(M, K, device=dev)
        w = torch.zeros(N, K, device=dev) if zero else torch.randn(N, K, device=dev) / K ** 0.5
        b = torch.randn(N, device=dev)
        out = torch.empty(M, N, device=dev)
        ts = torch.zeros(8192, 8, dtype=torch.int64, device=dev)

        def launch():
            rc = lib.rohm_gemm_f32(a.data_ptr(), K, w.data_ptr(), K, out.data_ptr(), N, M, N, K, b.data_ptr(),
                                   ts.data_ptr(), 0, 0, stream)
            assert rc == 0, lib.rohm_last_error()
        for _ in range(5):
            launch()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        ts.zero_()
        torch.cuda.synchronize()
        launch()
        e0.record()
        launch()
        e1.record()
        torch.cuda.synchronize()
        t = ts.cpu().numpy()
        t = t[t[:, 0] != 0]
        w_ = t[:, :5].astype(np.float64) * 0.01          # us
        base = w_[:, 0].min()
        w_ -= base
        ph = np.diff(w_, axis=1)
        cyc = t[:, 5].astype(np.float64)
        ghz = cyc / (ph[:, 1] * 1e3)
        xcc = t[:, 6] & 15
        q = lambda v: f'min {v.min():6.2f} med {np.median(v):6.2f} p95 {np.percentile(v, 95):6.2f} max {v.max():6.2f}'
        print(f'== {name} M={M} N={N} K={K}: {len(t)} workgroups, event time {e0.elapsed_time(e1) * 1e3:.1f} us, '
              f'first entry -> last ack {w_[:, 4].max():.1f} us')
        print(f'   entry offset      {q(w_[:, 0])}')
        print(f'   prologue          {q(ph[:, 0])}')
        print(f'   main loop         {q(ph[:, 1])}   ({np.median(ghz):.2f} GHz median shader clock)')
        print(f'   epilogue issue    {q(ph[:, 2])}')
        print(f'   store ack         {q(ph[:, 3])}')
        print(f'   exit time         {q(w_[:, 4])}')
        for x in range(8):
            m = xcc == x
            if m.any():
                print(f'   xcc {x}: {m.sum():4d} wgs, loop med {np.median(ph[m, 1]):6.2f}, exit med '
                      f'{np.median(w_[m, 4]):6.2f} max {w_[m, 4].max():6.2f}')


if __name__ == '__main__':
    main()
