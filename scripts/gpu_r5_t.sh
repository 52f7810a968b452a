#!/bin/bash
# Round 5, call T: HEAD tree, same box: the shipped path (encoder stack) against the round-4 launch plan (ROHM_POSENET_CHAIN=0: one launch per
# GEMM with the LayerNorm inside, separate embed / finish / pack) at B = 64 and 32, two rounds, candidate first.
TAG=${1:-r5_t}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd $R
for round in 1 2; do
for leg in "stack 64" "r4plan 64" "stack 32" "r4plan 32"; do
  set -- $leg
  if [ $1 = r4plan ]; then export ROHM_POSENET_CHAIN=0 ROHM_POSENET_FINISH_PACK=0; else unset ROHM_POSENET_CHAIN ROHM_POSENET_FINISH_PACK; fi
  timeout 400 python bench.py --no-extras --no-cpu-baseline --batch $2 > $OUT/bench_$1_b$2_$round.json 2> $OUT/bench_$1_b$2_$round.err
  python - <<PY
import json
try:
    d = json.loads(open('$OUT/bench_$1_b$2_$round.json').read().strip().splitlines()[-1])
    print('$1 b$2 $round', round(d['value'], 2), 'e2e', round(d['e2e_frac'], 3), {k: v['avg_us'] for k, v in list(d['roofline']['kernels'].items())[:3]})
except Exception as e:
    print('$1 failed', e); print(open('$OUT/bench_$1_b$2_$round.err').read()[-800:])
PY
done
done
