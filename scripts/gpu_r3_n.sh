#!/bin/bash
# Round 3: LayerNorm fold A/B, short form: kernel-level fold tests + the two fp16x3 bench legs (fold on / off), twice each
TAG=${1:-r3_n}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd $R
timeout 1200 python -m pytest tests/test_gpu_planes.py -x -q -p no:cacheprovider -k "ln_fold or shape_errors" 2>&1 | tail -5 | tee $OUT/pytest_planes_fold.txt
for rep in a b; do
for f in 1 0; do
  ROHM_PP_LNFOLD=$f ROHM_GEMM_PRECISION=fp16x3 timeout 300 python bench.py --no-cpu-baseline --no-extras --with-accuracy --steps 2 --warmup 1 > $OUT/bench_fp16x3_fold$f$rep.json 2> $OUT/bench_fp16x3_fold$f$rep.err
  python - <<PY
import json
try:
    d = json.loads(open('$OUT/bench_fp16x3_fold$f$rep.json').read().strip().splitlines()[-1])
    print('fp16x3 fold=$f', round(d['value'], 2), d['unit'], 'accuracy', d.get('accuracy', {}).get('max_abs_vs_reference'))
    for k, v in list(d['roofline']['kernels'].items())[:4]:
        print('   ', k, v['avg_us'])
except Exception as e:
    print('fold=$f failed', e); print(open('$OUT/bench_fp16x3_fold$f$rep.err').read()[-1500:])
PY
done
done
