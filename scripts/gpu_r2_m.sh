#!/bin/bash
# Round-2, second session: TrajNet step work (single-exp Mish, GroupNorm with every load up front, fused loop tail,
# conv launch-shape sweep) checked and measured on an MI355X box.  Usage: bash scripts/gpu_r2_m.sh TAG
TAG=${1:-r2_m}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd $R
timeout 600 python -m pytest tests/test_gpu_trajnet.py -x -q 2>&1 | tail -6 | tee $OUT/pytest_trajnet.txt
timeout 300 python scripts/bench_trajnet.py --sweep 1 32 > $OUT/trajnet_sweep.json 2> $OUT/sweep.err; cat $OUT/trajnet_sweep.json; tail -3 $OUT/sweep.err
timeout 300 python scripts/bench_trajnet.py 1 32 > $OUT/trajnet_loop.json 2> $OUT/loop.err; python - <<PY
import json
d = json.load(open('$OUT/trajnet_loop.json'))
for k, v in d.items():
    print(k, v['wall_ms'], 'ms', v['launches'], 'launches', {n: (x['launches'], x['avg_us']) for n, x in v['kernels'].items()})
PY
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -6 | tee $OUT/pytest_gpu.txt
ls $OUT
