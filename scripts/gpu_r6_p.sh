#!/bin/bash
# Round 6, call P: the whole GPU suite on the tree with the opt-in clip-resident TrajNet step, smoke, and the bench record exactly as the driver runs it (timed).
TAG=${1:-r6_p}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd $R
( time timeout 2700 python -m pytest tests -m gpu -q -p no:cacheprovider --durations=8 2>&1 | tail -16 ) 2>&1 | tee $OUT/pytest_gpu.txt
timeout 600 python __graft_entry__.py smoke 2>&1 | tail -3 | tee $OUT/smoke.txt
( time timeout 1500 python bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/bench_full.json 2> $OUT/bench_full.err ) 2>&1 | tail -3 | tee $OUT/bench_time.txt
python - <<PY
import json
try:
    d = json.loads(open('$OUT/bench_full.json').read().strip().splitlines()[-1])
    print('headline', round(d['value'], 2), d['ms_per_step'], 'dominant', round(d['roofline']['dominant']['frac'], 4), d['roofline']['dominant']['avg_launch_us'], 'cpu', {k: d['cpu_baseline'].get(k) for k in ('value', 'samples', 'spread', 'kind', 'cores')})
    for k, v in d['configs'].items():
        c = v.get('cpu_baseline') or {}
        print('  ', k, v.get('value'), v.get('error'), v.get('child_wall_s'), 'cpu', c.get('value'), c.get('error'))
except Exception as e:
    print('full bench failed', e); print(open('$OUT/bench_full.err').read()[-2000:])
PY
