#!/bin/bash
# Round 5, call K: variants of the stack's GEMM phases at B = 64 (experiment libraries, same box, two rounds): default = all-16x16 MFMA,
# m32 = mixed 32x32x2 form, peel = last chunk peeled for the 384-wide tile too, peelnocol = + its bias row from registers; then the chain tests
# on the default library.
TAG=${1:-r5_k}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd $R
for round in 1 2; do
for leg in peelnocol peel default m32; do
  if [ $leg = default ]; then unset ROHM_HIP_LIB; else export ROHM_HIP_LIB=$R/rohm_amd/librohm_hip_$leg.so; fi
  timeout 400 python bench.py --no-extras --no-cpu-baseline > $OUT/bench_${leg}_$round.json 2> $OUT/bench_${leg}_$round.err
  python - <<PY
import json
try:
    d = json.loads(open('$OUT/bench_${leg}_$round.json').read().strip().splitlines()[-1])
    print('$leg $round', round(d['value'], 2), {k: v['avg_us'] for k, v in list(d['roofline']['kernels'].items())[:2]})
except Exception as e:
    print('$leg failed', e); print(open('$OUT/bench_${leg}_$round.err').read()[-800:])
PY
done
done
unset ROHM_HIP_LIB
timeout 900 python -m pytest tests/test_gpu_chain.py -m gpu -q -p no:cacheprovider -s 2>&1 | grep -E "max\||passed|failed|Error" | tail -16
