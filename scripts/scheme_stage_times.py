"""Where does a pass of the full scheme (bench.py --workload scheme, B = 32) spend its wall time?  Wraps the stages of
rohm_amd.inference.run_amass_iterations with synchronising timers (so the stages no longer overlap with the host: read the sums, not
the pass time) and splits the PoseNet stage into its fused un-guided part and its step-wise guided tail.
(gpurun: python scripts/scheme_stage_times.py [B])"""
import sys
import time
import types

import torch

sys.path.insert(0, '.')
import bench                                                     # noqa: E402
from rohm_amd import inference                                   # noqa: E402
from rohm_amd.diffusion import ddpm                              # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
times = {}


def timed(name, fn):
    def wrapper(*a, **k):
        torch.cuda.synchronize()
        t0 = time.time()
        out = fn(*a, **k)
        torch.cuda.synchronize()
        times.setdefault(name, []).append(time.time() - t0)
        return out
    return wrapper


inference.rederive_traj = timed('rederive_traj', inference.rederive_traj)
inference._traj_stage = timed('traj_stage', inference._traj_stage)
cls = [c for c in vars(ddpm).values() if isinstance(c, type) and hasattr(c, '_fused_loop')][0]
orig_step = cls._step


cls._fused_loop = timed('sampling run (fused part + guided tail)', cls._fused_loop)
guided = {'n': 0, 't': 0.0}


def timed_step(self, model, batch, x, t, step_i, grad_type=None, t_int=None):
    if grad_type is None:
        return orig_step(self, model, batch, x, t, step_i, grad_type=grad_type, t_int=t_int)
    torch.cuda.synchronize()
    t0 = time.time()
    out = orig_step(self, model, batch, x, t, step_i, grad_type=grad_type, t_int=t_int)
    torch.cuda.synchronize()
    guided['n'] += 1
    guided['t'] += time.time() - t0
    return out


args = types.SimpleNamespace(batch=B, ddpm_steps=1000, workload='scheme', guidance_semantics='replica', steps=1, warmup=1,
                             profile_stride=16, force_dist=False, gpus=1, backend='nccl')
real_timed_region = bench.timed_region


def my_region(one_pass, a, world, dev, dist, profile=True):
    one_pass()                                      # warm-up, un-instrumented statistics discarded below
    torch.cuda.synchronize()
    times.clear()
    guided.update(n=0, t=0.0)
    cls._step = timed_step
    t0 = time.time()
    out = one_pass()
    torch.cuda.synchronize()
    total = time.time() - t0
    cls._step = orig_step
    print(f'B={B}: instrumented pass {total * 1e3:.1f} ms')
    for k, v in times.items():
        print(f'   {k}: {len(v)} calls, {sum(v) * 1e3:.1f} ms  {[round(x * 1e3, 1) for x in v]}')
    print(f'   guided steps inside the sampling runs: {guided["n"]} steps, {guided["t"] * 1e3:.1f} ms '
          f'({guided["t"] / max(guided["n"], 1) * 1e3:.3f} ms each)')
    cls._step = orig_step
    torch.cuda.synchronize()
    t0 = time.time()
    out = one_pass()
    torch.cuda.synchronize()
    print(f'   un-instrumented pass: {(time.time() - t0) * 1e3:.1f} ms')
    return total, out, {}


bench.timed_region = my_region
bench.finish = lambda *a: None
try:
    bench.scheme_bench(args, 1, 0, torch.device('cuda:0'), None)
except ZeroDivisionError:
    pass
