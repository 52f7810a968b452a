"""Two PoseNet forwards at the headline batch on two HIP streams at the same time: do the in-kernel exchanges (LayerNorm statistics
between partner tiles, stream-K partials) survive sharing the chip?  Prints per-run wall time, the exchange status of both
workspaces and the agreement with a serial run.  (gpurun: python scripts/two_stream_probe.py [B] [rounds])"""
import sys
import time

import torch

sys.path.insert(0, '.')
sys.path.insert(0, 'tests')
from test_gpu_posenet import make_posenet, DEV          # noqa: E402
from helpers import seeded                               # noqa: E402
from rohm_amd import _lib                                # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
R = int(sys.argv[2]) if len(sys.argv) > 2 else 3
net, _ = make_posenet(5)
x, c = seeded(1, B, 294, 1, 143).to(DEV), seeded(2, B, 294, 1, 143).to(DEV)
t = torch.tensor([(37 * i + 1) % 1000 for i in range(B)], device=DEV)
ref = net({'x_t': x, 'cond': c}, t)
torch.cuda.synchronize()
t0 = time.time()
for _ in range(2 * R):
    net({'x_t': x, 'cond': c}, t)
torch.cuda.synchronize()
serial = time.time() - t0
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
for st in (s1, s2):                    # workspaces of both streams
    with torch.cuda.stream(st):
        net({'x_t': x, 'cond': c}, t)
torch.cuda.synchronize()
outs = []
t0 = time.time()
for _ in range(R):
    for st in (s1, s2):
        with torch.cuda.stream(st):
            outs.append(net({'x_t': x, 'cond': c}, t))
torch.cuda.synchronize()
both = time.time() - t0
status = []
for st in (s1, s2):
    with torch.cuda.stream(st):
        try:
            net.check_exchange()
            status.append('ok')
        except _lib.RohmHipError as e:
            status.append(str(e)[:120])
print(f'B={B}: {2 * R} forwards serial {serial * 1e3:.1f} ms, on two streams {both * 1e3:.1f} ms; status {status}; '
      f'equal to the serial result: {[bool(torch.equal(o, ref)) for o in outs]}')
