#!/bin/bash
# Round 5, call Q: three more in-place variants of the stack (experiment libraries, same box): nosc1 = operand loads without the device-scope
# policy (TIMING only), nosleep = meeting-point polls without s_sleep, kfirst0 = the in-stack attention with the round-2 DMA order.
TAG=${1:-r5_q}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd $R
for B in 64 32; do
for leg in nosc1 nosleep kfirst0 default; do
  if [ $leg = default ]; then unset ROHM_HIP_LIB; else export ROHM_HIP_LIB=$R/rohm_amd/librohm_hip_$leg.so; fi
  timeout 400 python bench.py --no-extras --no-cpu-baseline --batch $B > $OUT/bench_${leg}_b$B.json 2> $OUT/bench_${leg}_b$B.err
  python - <<PY
import json
try:
    d = json.loads(open('$OUT/bench_${leg}_b$B.json').read().strip().splitlines()[-1])
    print('$leg b$B', round(d['value'], 2), {k: v['avg_us'] for k, v in list(d['roofline']['kernels'].items())[:1]})
except Exception as e:
    print('$leg failed', e); print(open('$OUT/bench_${leg}_b$B.err').read()[-600:])
PY
done
done
