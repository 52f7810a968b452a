#!/bin/bash
# Round-2, second session: conv gather with a branch-free address path -- parity of the TrajNet tests, loop times, per-shape times.
TAG=${1:-r2_o}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd $R
timeout 600 python -m pytest tests/test_gpu_trajnet.py -x -q 2>&1 | tail -4 | tee $OUT/pytest_trajnet.txt
timeout 300 python scripts/bench_trajnet.py 1 32 256 > $OUT/trajnet_loop.json 2> $OUT/loop.err; python - <<PY
import json
d = json.load(open('$OUT/trajnet_loop.json'))
for k, v in d.items():
    print(k, v['wall_ms'], 'ms', v['launches'], 'launches', {n: (x['launches'], x['avg_us']) for n, x in v['kernels'].items()})
PY
timeout 200 python scripts/bench_trajnet.py --detail 32 > $OUT/trajnet_detail_b32.txt 2> $OUT/detail.err; head -32 $OUT/trajnet_detail_b32.txt
