#!/bin/bash
# Round 5, call M: the loop tail without the redundant re-fetch DMA (default) against the re-fetching form (experiment library), B = 64 and 32,
# same box, two rounds; chain tests on the default.
TAG=${1:-r5_m}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd $R
for round in 1 2; do
for leg in "default 64" "refetch 64" "default 32" "refetch 32"; do
  set -- $leg
  if [ $1 = default ]; then unset ROHM_HIP_LIB; else export ROHM_HIP_LIB=$R/rohm_amd/librohm_hip_$1.so; fi
  timeout 400 python bench.py --no-extras --no-cpu-baseline --batch $2 > $OUT/bench_$1_b$2_$round.json 2> $OUT/bench_$1_b$2_$round.err
  python - <<PY
import json
try:
    d = json.loads(open('$OUT/bench_$1_b$2_$round.json').read().strip().splitlines()[-1])
    print('$1 b$2 $round', round(d['value'], 2), {k: v['avg_us'] for k, v in list(d['roofline']['kernels'].items())[:2]})
except Exception as e:
    print('$1 failed', e); print(open('$OUT/bench_$1_b$2_$round.err').read()[-800:])
PY
done
done
unset ROHM_HIP_LIB
timeout 900 python -m pytest tests/test_gpu_chain.py -m gpu -q -p no:cacheprovider 2>&1 | tail -3
