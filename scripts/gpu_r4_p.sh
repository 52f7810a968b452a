#!/bin/bash
# Round 4: the full default bench record again (in-stream profiler at every 100th step instead of every 16th), same binary as r4_zzzz_final
TAG=${1:-r4_p}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd $R
( time timeout 1200 python bench.py > $OUT/bench_full.json 2> $OUT/bench_full.err ) 2>&1 | tail -3
timeout 300 python bench.py --no-extras --no-cpu-baseline --profile-stride 16 > $OUT/bench_stride16.json 2> $OUT/bench_stride16.err
python - <<PY
import json
for n in ('bench_full', 'bench_stride16'):
    d = json.loads(open('$OUT/%s.json' % n).read().strip().splitlines()[-1])
    print(n, 'headline', round(d['value'], 3), 'frac', round(d['roofline']['frac'], 4), 'launches_timed', d['roofline']['launches_timed'])
    for k, v in list(d['roofline']['kernels'].items())[:6]:
        print('    ', k, v['launches'], v['avg_us'])
    if 'configs' in d:
        sl = d['second_line']
        print('second_line', sl.get('mode'), sl.get('value'), {k: v.get('value') for k, v in sl.get('also', {}).items()})
        for k, v in d['configs'].items():
            print('  ', k, v.get('value'), v.get('error'), v.get('child_wall_s'))
        print('cpu', d['cpu_baseline']['value'], 'acc', d['accuracy']['max_abs_vs_reference'])
PY
