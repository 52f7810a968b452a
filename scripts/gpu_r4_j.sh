#!/bin/bash
# Round 4: TrajNet fused [conv5 | 1x1 residual] launches -- the residual columns walk the centre tap only, the freed workgroup slots go to the conv tiles
TAG=${1:-r4_j}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd $R
timeout 900 python -m pytest tests/test_gpu_trajnet.py -x -q -p no:cacheprovider 2>&1 | tail -6 | tee $OUT/pytest_trajnet.txt
timeout 300 python scripts/bench_trajnet.py 1 32 > $OUT/trajnet_loop_tap1.json 2> $OUT/trajnet_loop_tap1.err
ROHM_TRAJ_RES_TAP=0 timeout 300 python scripts/bench_trajnet.py 1 32 > $OUT/trajnet_loop_tap0.json 2> $OUT/trajnet_loop_tap0.err
python - <<PY
import json
for n in ('tap1', 'tap0'):
    try:
        d = json.load(open('$OUT/trajnet_loop_%s.json' % n))
        for k, v in d.items():
            print(n, k, {a: b for a, b in v.items() if a != 'kernels'})
    except Exception as e:
        print(n, 'failed', e, open('$OUT/trajnet_loop_%s.err' % n).read()[-800:])
PY
timeout 600 python -m pytest tests/test_gpu_scheme.py -x -q -p no:cacheprovider -k "free_running" 2>&1 | tail -3 | tee $OUT/pytest_scheme_free.txt
