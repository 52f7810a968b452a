"""Where does the egobody workload lose finiteness?  Wraps every eval_losses of bench.py's scheme_bench."""
import sys, types, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from rohm_amd.diffusion import ddpm

orig = ddpm.DDPMSampler._eval if hasattr(ddpm.DDPMSampler, '_eval') else None
def wrap(cls):
    f = cls.eval_losses
    def g(self, *a, **k):
        r = f(self, *a, **k)
        o = r[1]
        b = k.get('batch')
        print(cls.__name__, 'out max', float(o.abs().max()), 'finite', bool(torch.isfinite(o).all()),
              'cond max', float(b['cond'].abs().max()), 'cond finite', bool(torch.isfinite(b['cond']).all()), flush=True)
        return r
    cls.eval_losses = g
from rohm_amd.diffusion.respace import SpacedDiffusionPoseNet, SpacedDiffusionTrajNet
wrap(SpacedDiffusionPoseNet); wrap(SpacedDiffusionTrajNet)
sys.argv = ['bench.py', '--workload', 'egobody', '--batch', '32', '--steps', '1', '--warmup', '0']
bench.main()
