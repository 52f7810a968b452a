"""Does the denoising loop of B clips run faster as P part-batches on P HIP streams (clips are independent)?  Each part runs
rohm_posenet_sample_loop on its own stream and workspace from its own host thread (ctypes releases the GIL).
(gpurun: python scripts/split_batch_probe.py [B] [steps])"""
import sys
import threading
import time

import numpy as np
import torch

sys.path.insert(0, '.')
sys.path.insert(0, 'tests')
from test_gpu_posenet import make_posenet, DEV          # noqa: E402
from helpers import seeded                               # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
N = int(sys.argv[2]) if len(sys.argv) > 2 else 200
net, _ = make_posenet(5)
cond = seeded(2, B, 294, 1, 143).to(DEV)
x0 = seeded(1, B, 294, 1, 143).to(DEV)
noise = torch.randn(N, B, 294, 1, 143, device=DEV)
t_model = list(range(N))[::-1]
coef = np.tile(np.asarray([[0.02, 0.97, 0.1]], np.float32), (N, 1))


def run(parts):
    bounds = [B * p // parts for p in range(parts + 1)]
    streams = [torch.cuda.Stream() for _ in range(parts)]
    xs = [x0[bounds[p]:bounds[p + 1]].clone() for p in range(parts)]
    cs = [cond[bounds[p]:bounds[p + 1]].contiguous() for p in range(parts)]
    ns = [noise[:, bounds[p]:bounds[p + 1]].contiguous() for p in range(parts)]

    def work(p, n):
        with torch.cuda.stream(streams[p]):
            net.sample_loop_native(xs[p], cs[p], t_model[:n], coef[:n], ns[p][:n])

    def go(n):
        th = [threading.Thread(target=work, args=(p, n)) for p in range(parts)]
        for t in th:
            t.start()
        for t in th:
            t.join()
        torch.cuda.synchronize()

    go(4)                                  # workspaces, first launches
    t0 = time.time()
    go(N)
    dt = time.time() - t0
    return dt, torch.cat(xs)


base_dt, base = run(1)
print(f'B={B}, {N} steps: 1 stream {base_dt * 1e3 / N:.3f} ms/step = {B / (base_dt / N * 1000):.2f} clips/s (1000-step run)')
for parts in (2, 4, 1, 2):
    dt, out = run(parts)
    print(f'   {parts} part-batches on {parts} streams: {dt * 1e3 / N:.3f} ms/step = {B / (dt / N * 1000):.2f} clips/s; '
          f'max |diff to 1 stream| {float((out - base).abs().max()):.2e}')
