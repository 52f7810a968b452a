#!/bin/bash
# Round 5, call J: what the 32x32x2 MFMA form buys inside the stack (experiment library without it), same box; + the launcher GPU tests.
TAG=${1:-r5_j}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd $R
for leg in "nom32" "default" "nom32" "default"; do
  if [ $leg = nom32 ]; then export ROHM_HIP_LIB=$R/rohm_amd/librohm_hip_nom32.so; else unset ROHM_HIP_LIB; fi
  timeout 400 python bench.py --no-extras --no-cpu-baseline > $OUT/bench_$leg.json 2> $OUT/bench_$leg.err
  python - <<PY
import json
try:
    d = json.loads(open('$OUT/bench_$leg.json').read().strip().splitlines()[-1])
    print('$leg', round(d['value'], 2), {k: v['avg_us'] for k, v in list(d['roofline']['kernels'].items())[:3]})
except Exception as e:
    print('$leg failed', e); print(open('$OUT/bench_$leg.err').read()[-800:])
PY
done
unset ROHM_HIP_LIB
timeout 900 python -m pytest tests/test_bench_launcher.py -m gpu -q -p no:cacheprovider 2>&1 | tail -4
