#!/bin/bash
# Round 3: LayerNorm folded into the plane GEMMs (two-plane modes) -- A/B bench, whole PoseNet suite under fp16x3 / bf16x3 with the fold
TAG=${1:-r3_l}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd $R
for f in 1 0; do
  ROHM_PP_LNFOLD=$f ROHM_GEMM_PRECISION=fp16x3 timeout 300 python bench.py --no-cpu-baseline --no-extras --with-accuracy --steps 2 --warmup 1 > $OUT/bench_fp16x3_fold$f.json 2> $OUT/bench_fp16x3_fold$f.err
  python - <<PY
import json
try:
    d = json.loads(open('$OUT/bench_fp16x3_fold$f.json').read().strip().splitlines()[-1])
    print('fp16x3 fold=$f', round(d['value'], 2), d['unit'], 'accuracy', d.get('accuracy', {}).get('max_abs_vs_reference'))
    for k, v in list(d['roofline']['kernels'].items())[:7]:
        print('   ', k, v)
except Exception as e:
    print('fold=$f failed', e); print(open('$OUT/bench_fp16x3_fold$f.err').read()[-1500:])
PY
done
timeout 1800 python -m pytest tests/test_gpu_precision_ladder.py -x -q -p no:cacheprovider 2>&1 | tail -25 | tee $OUT/pytest_ladder.txt
ROHM_GEMM_PRECISION=fp16x3 timeout 300 python bench.py --no-cpu-baseline --no-extras --steps 2 --warmup 1 --batch 32 > $OUT/bench_fp16x3_b32.json 2> $OUT/bench_fp16x3_b32.err; python -c "
import json; d=json.loads(open('$OUT/bench_fp16x3_b32.json').read().strip().splitlines()[-1]); print('fp16x3 b32 (fold)', round(d['value'],2))"
ROHM_GEMM_PRECISION=bf16x3 timeout 300 python bench.py --no-cpu-baseline --no-extras --steps 2 --warmup 1 > $OUT/bench_bf16x3.json 2> $OUT/bench_bf16x3.err; python -c "
import json; d=json.loads(open('$OUT/bench_bf16x3.json').read().strip().splitlines()[-1]); print('bf16x3 (fold)', round(d['value'],2))"
