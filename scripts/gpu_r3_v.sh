#!/bin/bash
# Round 3, final tree: full bench record first (cold chip), the whole GPU suite, smoke, rocprofv3 kernel stats of the headline leg and of the
# scheme / prox configs
TAG=${1:-r3_v}
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
    print('headline', round(d['value'], 2), 'frac', round(d['roofline']['frac'], 3), 'traffic', d['roofline'].get('traffic'), d['roofline'].get('traffic_source'), 'cpu', d.get('cpu_baseline', {}).get('value'), d.get('cpu_baseline', {}).get('kind'))
    sl = d['second_line']
    print('second_line', sl.get('mode'), sl.get('value'), sl.get('error'), (sl.get('accuracy') or {}).get('max_abs_vs_reference'), 'also', {k: v.get('value') for k, v in sl.get('also', {}).items()})
    for k, v in d['configs'].items():
        print('  ', k, v.get('value'), v.get('error'), v.get('child_wall_s'))
except Exception as e:
    print('full bench failed', e); print(open('$OUT/bench_full.err').read()[-2000:])
PY
( time timeout 2400 python -m pytest tests -m gpu -x -q -p no:cacheprovider 2>&1 | tail -6 ) 2>&1 | tee $OUT/pytest_gpu.txt
timeout 600 python __graft_entry__.py smoke 2>&1 | tail -3 | tee $OUT/smoke.txt
cd /tmp
stats() {   # name, env, command...
  local name=$1; local env=$2; shift; shift
  rm -rf /tmp/prof_$name
  env $env timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$name -o p -- "$@" > $OUT/rocprof_$name.log 2>&1
  find /tmp/prof_$name -name "*kernel_stats.csv" -exec cp {} $OUT/kernel_stats_$name.csv \;
  python $R/scripts/rocprof_summary.py $OUT/kernel_stats_$name.csv > $OUT/rocprof_kernel_stats_$name.txt 2>&1
  head -9 $OUT/rocprof_kernel_stats_$name.txt
}
stats bench_default ROHM_NOOP=1 python $R/bench.py --no-cpu-baseline --no-extras
stats scheme_b32 ROHM_NOOP=1 python $R/bench.py --workload scheme --batch 32 --steps 1 --warmup 0 --no-cpu-baseline --no-extras
stats prox_b32 ROHM_NOOP=1 python $R/bench.py --workload prox --batch 32 --steps 1 --warmup 0 --no-cpu-baseline --no-extras
rm -f $OUT/kernel_stats_*.csv
