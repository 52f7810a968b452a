#!/bin/bash
# Round 5, call L: more variants, same box, two rounds: peel128 / peel0 = the last-chunk peel only for tiles <= 128 columns / never (stack);
# nomixed = gemm_f32.hip's own launches all on 16x16x4 MFMAs (ROHM_GEMM_MIXED=0), measured on the launch-per-GEMM path (ROHM_POSENET_CHAIN=0).
TAG=${1:-r5_l}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd $R
for round in 1 2; do
for leg in peel128 peel0 default nomixed_chain0 default_chain0; do
  lib=${leg%%_chain0}
  if [ $lib = default ]; then unset ROHM_HIP_LIB; else export ROHM_HIP_LIB=$R/rohm_amd/librohm_hip_$lib.so; fi
  if [ $leg != $lib ]; then export ROHM_POSENET_CHAIN=0; else unset ROHM_POSENET_CHAIN; fi
  timeout 400 python bench.py --no-extras --no-cpu-baseline > $OUT/bench_${leg}_$round.json 2> $OUT/bench_${leg}_$round.err
  python - <<PY
import json
try:
    d = json.loads(open('$OUT/bench_${leg}_$round.json').read().strip().splitlines()[-1])
    print('$leg $round', round(d['value'], 2), {k: v['avg_us'] for k, v in list(d['roofline']['kernels'].items())[:4]})
except Exception as e:
    print('$leg failed', e); print(open('$OUT/bench_${leg}_$round.err').read()[-800:])
PY
done
done
