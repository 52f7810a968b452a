"""Debug aid: the fused sampling loop recorded into a hipGraph and replayed, against eager calls (prints max differences)."""
import os, sys
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, ROOT + '/tests')
from test_gpu_posenet import DEV, make_posenet
from rohm_amd import _lib

B, T, n = int(os.environ.get('B', 32)), 143, 4
net, _ = make_posenet(5)
nat = net.native(torch.device(DEV))
g = torch.Generator(device=DEV).manual_seed(3)
x0 = torch.randn(B, 294, 1, T, device=DEV, generator=g)
cond = torch.randn(B, 294, 1, T, device=DEV, generator=g)
noise = torch.randn(n, B, 294, 1, T, device=DEV, generator=g)
coef = np.asarray([[0.05, 0.95, 0.1]] * n, np.float32)
ts = [400, 399, 398, 397]
eager = x0.clone(); net.sample_loop_native(eager, cond, ts, coef, noise)
eager_b = x0.clone(); net.sample_loop_native(eager_b, cond, ts, coef, noise)
print('eager repeat equal:', torch.equal(eager, eager_b))
eager2 = eager.clone(); net.sample_loop_native(eager2, cond, ts, coef, noise)
side = torch.cuda.Stream()
xs = x0.clone()
with torch.cuda.stream(side):
    warm = x0.clone(); net.sample_loop_native(warm, cond, ts, coef, noise)
    ws = nat.workspace(B, T)
torch.cuda.synchronize()
print('side-stream eager equal:', torch.equal(warm, eager))
off = _lib.lib().rohm_posenet_status_offset(nat.handle, B, T)
word = ws[off:off + 16].view(torch.int32)
print('header before capture', word.tolist())
graph = torch.cuda.CUDAGraph()
with torch.cuda.graph(graph, stream=side):
    net.sample_loop_native(xs, cond, ts, coef, noise)
torch.cuda.synchronize()
print('header after capture', word.tolist(), 'xs untouched by capture:', torch.equal(xs, x0))
xs.copy_(x0)
for k, want in enumerate((eager, eager2, None)):
    graph.replay(); torch.cuda.synchronize()
    print(f'replay {k}: header', word.tolist(), 'finite', bool(torch.isfinite(xs).all()),
          'max|xs - want|', None if want is None else float((xs - want).abs().max()), 'max|xs|', float(xs.abs().max()))
with torch.cuda.stream(side):
    try:
        net.check_exchange(); print('exchange clean')
    except Exception as e:
        print('exchange:', e)
