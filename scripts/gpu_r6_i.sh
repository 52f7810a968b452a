#!/bin/bash
# Round 6, call I: TIMING experiments (wrong results by design, experiment libraries only): what do (a) the meetings, (b) the LayerNorm part of the
# out-projection / linear2 epilogues, (c) the erf-form GELU of linear1's epilogue cost inside the stack?  Same box, two rounds, candidate first.
TAG=${1:-r6_i}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd $R
leg() {   # name lib batch
  if [ $2 = default ]; then unset ROHM_HIP_LIB; else export ROHM_HIP_LIB=$R/rohm_amd/librohm_hip_$2.so; fi
  timeout 400 python bench.py --no-extras --no-cpu-baseline --batch $3 --ddpm-steps 300 > $OUT/bench_$1.json 2> $OUT/bench_$1.err
  python - <<PY
import json
try:
    d = json.loads(open('$OUT/bench_$1.json').read().strip().splitlines()[-1])
    ph = (d['roofline'].get('attention') or {}).get('stack_phases') or {}
    print('$1', round(d['value'], 3), 'ms/pass', round(d['ms_per_step'], 1), {k: v['avg_us'] for k, v in list(d['roofline']['kernels'].items())[:1]}, {k: ph[k]['us_per_launch'] for k in ('out_proj_norm1', 'linear1_gelu', 'linear2_norm2') if k in ph}, 'meet', (d['roofline'].get('attention') or {}).get('meetings_share_of_launch'))
except Exception as e:
    print('$1 failed', e); print(open('$OUT/bench_$1.err').read()[-800:])
PY
  unset ROHM_HIP_LIB
}
for round in 1 2; do
  for lib in nomeet nolnepi nogelu default; do
    leg ${lib}64_$round $lib 64
  done
  for lib in nomeet nolnepi nogelu default; do
    leg ${lib}32_$round $lib 32
  done
done
