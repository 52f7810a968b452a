#!/bin/bash
# Round 6, full evidence set of a tree: bench record with extras (cold chip first), the whole GPU suite, smoke, rocprofv3 kernel stats of the
# headline / B = 32 / scheme / prox legs, PMC passes (separate, --pmc with --kernel-trace only), LBS and TrajNet loop records.
TAG=${1:-r6_z}
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
    print('headline', round(d['value'], 2), 'frac', round(d['roofline']['frac'], 3), 'e2e', round(d['e2e_frac'], 3), round(d['e2e_frac_executed'], 3), 'cpu', d.get('cpu_baseline', {}).get('value'), d.get('cpu_baseline', {}).get('kind'), 'acc', (d.get('accuracy') or {}).get('max_abs_vs_reference'))
    a = d['roofline'].get('attention') or {}
    print('   attention (in stack)', {k: a.get(k) for k in ('achieved', 'frac', 'share_of_launch', 'frac_incl_meeting', 'in_stack_error')}, 'standalone', {k: (d['roofline'].get('attention_standalone') or {}).get(k) for k in ('frac', 'avg_launch_us')})
    c = d.get('cpu_baseline', {}); print('   cpu samples', c.get('samples'), c.get('spread'))
    print('   rank_report', (d.get('rank_report') or {}).get('skew'))
    for k, v in list(d['roofline']['kernels'].items())[:9]:
        print('    ', k, v['launches'], v['avg_us'], v.get('tflops'))
    sl = d['second_line']
    print('second_line', sl.get('mode'), sl.get('value'), sl.get('error'), (sl.get('accuracy') or {}).get('max_abs_vs_reference'), 'also', {k: v.get('value') for k, v in sl.get('also', {}).items()})
    for k, v in d['configs'].items():
        print('  ', k, v.get('value'), v.get('error'), v.get('child_wall_s'), 'cpu', (v.get('cpu_baseline') or {}).get('value'), (v.get('cpu_baseline') or {}).get('error'))
except Exception as e:
    print('full bench failed', e); print(open('$OUT/bench_full.err').read()[-2000:])
PY
( time timeout 2700 python -m pytest tests -m gpu -q -p no:cacheprovider --durations=12 2>&1 | tail -30 ) 2>&1 | tee $OUT/pytest_gpu.txt
timeout 600 python __graft_entry__.py smoke 2>&1 | tail -3 | tee $OUT/smoke.txt
timeout 600 python scripts/stack_timeline.py $OUT/stack_phase_timeline.json 64 32 2>&1 | tee $OUT/stack_phase_timeline.txt
timeout 300 python scripts/bench_lbs.py 32 dense > $OUT/lbs_b32_mfma.json 2> $OUT/lbs.err
timeout 300 python scripts/bench_lbs.py 32 sparse > $OUT/lbs_b32_sparse.json 2>> $OUT/lbs.err
timeout 300 python scripts/bench_trajnet.py 1 32 > $OUT/trajnet_loop.json 2> $OUT/trajnet_loop.err
python - <<PY
import json
try:
    for n in ('mfma', 'sparse'):
        d = json.load(open('$OUT/lbs_b32_%s.json' % n)); print('lbs', n, d['with_vertices']['wall_us_per_call'], {k: v['us'] for k, v in d['with_vertices']['kernels'].items()})
    d = json.load(open('$OUT/trajnet_loop.json'))
    for k, v in d.items(): print('trajnet', k, {a: b for a, b in v.items() if a != 'kernels'})
except Exception as e:
    print('lbs / trajnet records failed', e)
PY
cd /tmp
stats() {   # name, env, command...
  local name=$1; local env=$2; shift; shift
  rm -rf /tmp/prof_$name
  env $env timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$name -o p -- "$@" > $OUT/rocprof_$name.log 2>&1
  find /tmp/prof_$name -name "*kernel_stats.csv" -exec cp {} $OUT/kernel_stats_$name.csv \;
  python $R/scripts/rocprof_summary.py $OUT/kernel_stats_$name.csv > $OUT/rocprof_kernel_stats_$name.txt 2>&1
  head -10 $OUT/rocprof_kernel_stats_$name.txt
}
stats bench_default ROHM_NOOP=1 python $R/bench.py --no-cpu-baseline --no-extras
stats bench_b32 ROHM_NOOP=1 python $R/bench.py --no-cpu-baseline --no-extras --batch 32
stats scheme_b32 ROHM_NOOP=1 python $R/bench.py --workload scheme --batch 32 --steps 1 --warmup 0 --no-cpu-baseline --no-extras
stats prox_b32 ROHM_NOOP=1 python $R/bench.py --workload prox --batch 32 --steps 1 --warmup 0 --no-cpu-baseline --no-extras
rm -f $OUT/kernel_stats_*.csv
pmc() {     # name, counters (quoted), command...
  local name=$1; local ctr=$2; shift; shift
  rm -rf /tmp/pmc_$name
  timeout 600 rocprofv3 --pmc $ctr --kernel-trace --output-format csv -d /tmp/pmc_$name -o p -- "$@" > $OUT/rocprof_pmc_$name.log 2>&1
  find /tmp/pmc_$name -name "*counter_collection.csv" -exec cp {} $OUT/pmc_$name.csv \;
}
SHORT="--no-cpu-baseline --no-extras --steps 1 --warmup 0 --ddpm-steps 12"
SQC="SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_WAVE_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_MFMA GRBM_GUI_ACTIVE"
pmc SQ "$SQC" python $R/bench.py $SHORT
python $R/scripts/sq_summary.py $OUT/pmc_SQ.csv $OUT/pmc_sq.json | tee $OUT/pmc_sq.txt | head -14
pmc FETCH_SIZE FETCH_SIZE python $R/bench.py $SHORT
pmc WRITE_SIZE WRITE_SIZE python $R/bench.py $SHORT
python $R/scripts/pmc_summary.py $OUT $OUT/pmc_traffic.json | tee $OUT/pmc_traffic.txt | head -24
pmc SQ_b32 "$SQC" python $R/bench.py $SHORT --batch 32
python $R/scripts/sq_summary.py $OUT/pmc_SQ_b32.csv $OUT/pmc_sq_b32.json | tee $OUT/pmc_sq_b32.txt | head -12
rm -f $OUT/pmc_SQ.csv $OUT/pmc_SQ_b32.csv $OUT/pmc_FETCH_SIZE.csv $OUT/pmc_WRITE_SIZE.csv
ls $OUT
