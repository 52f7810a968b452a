#!/bin/bash
# Round 6, call S: the TrajNet suite with the resident step under the launch profiler, and the scheme bench leg (the profiler's label of the one-launch step).
TAG=${1:-r6_yb}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd $R
timeout 900 python -m pytest tests/test_gpu_trajnet.py -m gpu -q -p no:cacheprovider 2>&1 | tail -4 | tee $OUT/pytest.txt
timeout 600 python bench.py --workload scheme --batch 32 --steps 2 --warmup 1 --no-cpu-baseline --no-extras > $OUT/scheme.json 2> $OUT/scheme.err
python - <<PY
import json
try:
    d = json.loads(open('$OUT/scheme.json').read().strip().splitlines()[-1])
    print(d['value'], d['ms_per_step'], {k: (v['launches'], v['avg_us']) for k, v in d['roofline']['kernels'].items()})
except Exception as e:
    print('failed', e); print(open('$OUT/scheme.err').read()[-1500:])
PY
