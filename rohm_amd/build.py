"""Build librohm_hip.so (gfx950) in-tree with hipcc.

`python -m rohm_amd.build` or `rohm_amd.build.build()`; `__graft_entry__.build()` calls this.
hipcc cross-compiles for gfx950 without a GPU.  The .so is git-ignored but travels to the GPU
box with the gpurun snapshot.
"""
from __future__ import annotations

import glob
import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, 'csrc')
LIB = os.path.join(HERE, 'librohm_hip.so')
ARCH = 'gfx950'


def sources():
    return sorted(glob.glob(os.path.join(CSRC, '*.hip')))


def _deps():
    return sources() + glob.glob(os.path.join(CSRC, '*.h')) + \
        glob.glob(os.path.join(HERE, '..', 'include', '*.h'))


def is_stale():
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    return any(os.path.getmtime(f) > t for f in _deps())


def find_hipcc():
    for cand in (os.environ.get('HIPCC'), shutil.which('hipcc'), '/opt/rocm/bin/hipcc'):
        if cand and os.path.exists(cand):
            return cand
    return None


def build(force=False, verbose=True, out=None, diag=None):
    """Build the library; `out` / `diag` build a second copy (scripts/gemm_timeline.py uses a ROHM_GEMM_DIAGNOSTICS build
    under ROHM_HIP_LIB without touching the shipped librohm_hip.so)."""
    target = out or LIB
    if diag is None:
        diag = os.environ.get('ROHM_DIAG') == '1'
    if not force and out is None and not is_stale():
        return LIB
    hipcc = find_hipcc()
    if hipcc is None:
        raise RuntimeError('hipcc not found: cannot build librohm_hip.so')
    # one object per source, compiled concurrently (gemm_f32.hip alone is ~50 s of template instantiations), then linked
    import concurrent.futures
    import tempfile
    flags = [f'--offload-arch={ARCH}', '-O3', '-std=c++17', '-fPIC', '-Wno-unused-result']
    if diag:      # diagnostic GEMM variants / knobs for scripts/gemm_*.py (never shipped)
        flags.append('-DROHM_GEMM_DIAGNOSTICS')
    with tempfile.TemporaryDirectory(prefix='rohm_build_') as tmp:
        objs = [os.path.join(tmp, os.path.basename(f) + '.o') for f in sources()]

        def compile_one(pair):
            src, obj = pair
            cmd = [hipcc] + flags + ['-c', src, '-o', obj]
            if verbose:
                print('[rohm_amd.build]', ' '.join(cmd), flush=True)
            subprocess.run(cmd, check=True, cwd=CSRC)
        with concurrent.futures.ThreadPoolExecutor(max_workers=min(8, os.cpu_count() or 1)) as ex:
            list(ex.map(compile_one, zip(sources(), objs)))
        link = [hipcc, f'--offload-arch={ARCH}', '-shared', '-fPIC', '-o', target + '.tmp'] + objs
        if verbose:
            print('[rohm_amd.build]', ' '.join(link), flush=True)
        subprocess.run(link, check=True, cwd=CSRC)
    os.replace(target + '.tmp', target)
    return target


if __name__ == '__main__':
    if '--diag' in sys.argv:      # second copy with the diagnostic GEMM variants, next to the shipped library
        print(build(force=True, out=os.path.join(HERE, 'librohm_hip_diag.so'), diag=True))
    else:
        build(force='--force' in sys.argv)
        print(LIB)
