#!/bin/bash
# Round 3: validation of the tree -- default bench first (cold chip), then the whole GPU suite, smoke
TAG=${1:-r3_h}
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
    print('headline', round(d['value'], 2), 'frac', round(d['roofline']['frac'], 3), 'cpu', d.get('cpu_baseline', {}).get('value'), d.get('cpu_baseline', {}).get('kind'))
    print('second_line', {k: d['second_line'].get(k) for k in ('value', 'error')}, (d['second_line'].get('accuracy') or {}).get('max_abs_vs_reference'))
    for k, v in d['configs'].items():
        print('  ', k, v.get('value'), v.get('error'), v.get('child_wall_s'))
except Exception as e:
    print('full bench failed', e); print(open('$OUT/bench_full.err').read()[-2000:])
PY
( time timeout 2400 python -m pytest tests -m gpu -x -q -p no:cacheprovider 2>&1 | tail -6 ) 2>&1 | tee $OUT/pytest_gpu.txt
timeout 600 python __graft_entry__.py smoke 2>&1 | tail -3 | tee $OUT/smoke.txt
