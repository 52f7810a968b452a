#!/bin/bash
# Round 4: LayerNorm inside the producer GEMMs (gemm_f32.hip EPI_BIAS_RES_LN): kernel + PoseNet parity, A/B benches at B = 64 / 32
TAG=${1:-r4_b}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd $R
timeout 900 python -m pytest tests/test_gpu_kernels.py -x -q -p no:cacheprovider -k "gemm" 2>&1 | tail -6 | tee $OUT/pytest_gemm.txt
timeout 1200 python -m pytest tests/test_gpu_posenet.py -x -q -p no:cacheprovider --durations=5 2>&1 | tail -12 | tee $OUT/pytest_posenet.txt
for cfg in "64 1" "64 0" "32 1" "32 0"; do
  set -- $cfg
  ROHM_POSENET_LN_FUSED=$2 timeout 300 python bench.py --no-cpu-baseline --no-extras --steps 3 --warmup 1 --batch $1 > $OUT/bench_b$1_lnfused$2.json 2> $OUT/bench_b$1_lnfused$2.err
  python - <<PY
import json
d = json.loads(open('$OUT/bench_b$1_lnfused$2.json').read().strip().splitlines()[-1])
print('fp32 b$1 ln_fused=$2', round(d['value'], 2), d['unit'], 'frac', round(d['roofline']['frac'], 4))
for k, v in list(d['roofline']['kernels'].items())[:9]:
    print('   ', k, v['avg_us'])
PY
done
