#!/bin/bash
# Round 6, call J: TIMING experiment at B = 32 -- the 64-wide tiles (out-projection, linear2) WITHOUT the LDS-DMA of their weight chunks (stale weights: wrong
# results; and 'wdead': the same chunks as ordinary global loads into registers nobody uses -- the ISSUE cost of a register path): the ceiling of any scheme that takes W off the LDS-DMA path there (e.g. weight fragments through registers).  Same box, two rounds.
TAG=${1:-r6_j}
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
    print('$1', round(d['value'], 3), 'ms/pass', round(d['ms_per_step'], 1), {k: v['avg_us'] for k, v in list(d['roofline']['kernels'].items())[:1]}, {k: ph[k]['us_per_launch'] for k in ('out_proj_norm1', 'linear1_gelu', 'linear2_norm2', 'embed') if k in ph})
except Exception as e:
    print('$1 failed', e); print(open('$OUT/bench_$1.err').read()[-800:])
PY
  unset ROHM_HIP_LIB
}
for round in 1 2; do
  leg wdead32_$round wdead 32
  leg nowdma32_$round nowdma 32
  leg default32_$round default 32
done
