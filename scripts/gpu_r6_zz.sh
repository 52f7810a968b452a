#!/bin/bash
# Round 6, call ZZ: the default bench record as the driver runs it (per-config CPU baselines inside), and the multi-tenant test with its report printed.
TAG=${1:-r6_zz}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd $R
( time timeout 1500 python bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/bench_full.json 2> $OUT/bench_full.err ) 2>&1 | tail -3
python - <<PY
import json
try:
    d = json.loads(open('$OUT/bench_full.json').read().strip().splitlines()[-1])
    print('headline', round(d['value'], 2), d['ms_per_step'], 'cpu', {k: d['cpu_baseline'].get(k) for k in ('value', 'samples', 'spread', 'kind', 'cores')})
    a = d['roofline']['attention']; print('attention', {k: a.get(k) for k in ('achieved', 'frac', 'share_of_launch')}, 'dominant', d['roofline']['dominant']['frac'], d['roofline']['dominant']['avg_launch_us'])
    for k, v in d['configs'].items():
        c = v.get('cpu_baseline') or {}
        print('  ', k, v.get('value'), v.get('error'), v.get('child_wall_s'), 'cpu', c.get('value'), c.get('step_ms'), c.get('error'), c.get('samples'))
except Exception as e:
    print('full bench failed', e); print(open('$OUT/bench_full.err').read()[-2000:])
PY
timeout 900 python -m pytest tests/test_gpu_exchange.py -m gpu -q -s -p no:cacheprovider -k "second_busy or capture" 2>&1 | grep -E "tenant|probe|passed|failed" | tee $OUT/pytest_multitenant.txt
