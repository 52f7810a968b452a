#!/bin/bash
# Round 6, last call: the whole GPU suite + smoke + the default bench record on the FINAL tree.
TAG=${1:-r6_final}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd $R
( time timeout 1200 python bench.py > $OUT/bench_full.json 2> $OUT/bench_full.err ) 2>&1 | tail -3
python - <<PY
import json
try:
    d = json.loads(open('$OUT/bench_full.json').read().strip().splitlines()[-1])
    print('headline', round(d['value'], 2), 'dominant', round(d['roofline']['dominant']['frac'], 4), d['roofline']['dominant']['avg_launch_us'], 'attention', round(d['roofline']['attention']['frac'], 3), 'cpu', d['cpu_baseline']['value'], d['cpu_baseline']['samples'])
    for k, v in d['configs'].items():
        print('  ', k, v.get('value'), v.get('error'), (v.get('cpu_baseline') or {}).get('value'))
except Exception as e:
    print('bench failed', e); print(open('$OUT/bench_full.err').read()[-2000:])
PY
( time timeout 2700 python -m pytest tests -m gpu -q -p no:cacheprovider --durations=8 2>&1 | tail -16 ) 2>&1 | tee $OUT/pytest_gpu.txt
timeout 600 python __graft_entry__.py smoke 2>&1 | tail -3 | tee $OUT/smoke.txt
