"""Build an experiment copy of librohm_hip.so for same-box A/B runs:  python scripts/build_variant.py NAME [--src DIR] [-DFLAG ...]

Compiles DIR/*.hip (default: rohm_amd/csrc of this tree; `--src` may point into an export of another commit, e.g.
`git archive HEAD~1 rohm_amd/csrc include | tar -x -C /tmp/prev` -> `--src /tmp/prev/rohm_amd/csrc`) with the shipped flags plus the
given defines into rohm_amd/librohm_hip_NAME.so, which `ROHM_HIP_LIB=...` selects at import (rohm_amd/_lib.py).  The shipped library is
never touched.  Only sources that differ from the shipped tree (or everything, when a define is given) are recompiled when --objs DIR
holds objects of an earlier call."""
import concurrent.futures
import glob
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def main():
    args = sys.argv[1:]
    name = args.pop(0)
    src = os.path.join(ROOT, 'rohm_amd', 'csrc')
    defs = []
    while args:
        a = args.pop(0)
        if a == '--src':
            src = os.path.abspath(args.pop(0))
        elif a.startswith('-D'):
            defs.append(a)
        else:
            raise SystemExit(f'unknown argument {a}')
    out = os.path.join(ROOT, 'rohm_amd', f'librohm_hip_{name}.so')
    objdir = f'/tmp/rohm_variant_{name}'
    os.makedirs(objdir, exist_ok=True)
    flags = ['--offload-arch=gfx950', '-O3', '-std=c++17', '-fPIC', '-Wno-unused-result'] + defs
    srcs = sorted(glob.glob(os.path.join(src, '*.hip')))

    def one(f):
        o = os.path.join(objdir, os.path.basename(f) + '.o')
        subprocess.run(['hipcc'] + flags + ['-c', f, '-o', o], check=True, cwd=src)
        return o
    with concurrent.futures.ThreadPoolExecutor(max_workers=8) as ex:
        objs = list(ex.map(one, srcs))
    subprocess.run(['hipcc', '--offload-arch=gfx950', '-shared', '-fPIC', '-o', out] + objs, check=True, cwd=src)
    print(out)


if __name__ == '__main__':
    main()
