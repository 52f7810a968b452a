#!/bin/bash
# Round 6, call C: the stack's phase timeline at B = 64 / 32; the independent SMPL-X pin on the kernels; one default bench record (per-config CPU baselines,
# median-of-3 headline baseline, in-stack attention in the line).
TAG=${1:-r6_c}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd $R
timeout 600 python scripts/stack_timeline.py $OUT/stack_phase_timeline.json 64 32 2>&1 | tee $OUT/stack_phase_timeline.txt
( time timeout 900 python -m pytest tests/test_smplx_independent.py -m gpu -q -p no:cacheprovider 2>&1 | tail -8 ) 2>&1 | tee $OUT/pytest_smplx_independent.txt
( time timeout 1200 python bench.py > $OUT/bench_full.json 2> $OUT/bench_full.err ) 2>&1 | tail -3
python - <<PY
import json
try:
    d = json.loads(open('$OUT/bench_full.json').read().strip().splitlines()[-1])
    print('headline', round(d['value'], 2), 'cpu', {k: d['cpu_baseline'].get(k) for k in ('value', 'samples', 'spread', 'kind', 'cores')})
    a = d['roofline']['attention']; print('attention', {k: a.get(k) for k in ('achieved', 'frac', 'share_of_launch', 'frac_incl_meeting', 'in_stack_error')})
    print('standalone', {k: (d['roofline'].get('attention_standalone') or {}).get(k) for k in ('achieved', 'frac', 'avg_launch_us')})
    for k, v in d['configs'].items():
        print('  ', k, v.get('value'), v.get('error'), v.get('child_wall_s'), (v.get('cpu_baseline') or {}).get('value'), (v.get('cpu_baseline') or {}).get('step_ms'), (v.get('cpu_baseline') or {}).get('error'))
except Exception as e:
    print('full bench failed', e); print(open('$OUT/bench_full.err').read()[-2000:])
PY
