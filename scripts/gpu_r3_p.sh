#!/bin/bash
# Round 3: residual prefetch in the flush of the plane stream GEMM + ControlNet side stream: parity (planes, trajnet, scheme tests),
# fold A/B, bf16x6
TAG=${1:-r3_p}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd $R
timeout 1500 python -m pytest tests/test_gpu_planes.py tests/test_gpu_trajnet.py tests/test_gpu_scheme.py -x -q -p no:cacheprovider 2>&1 | tail -6 | tee $OUT/pytest_planes_trajnet_scheme.txt
for f in 1 0; do
  ROHM_PP_LNFOLD=$f ROHM_GEMM_PRECISION=fp16x3 timeout 300 python bench.py --no-cpu-baseline --no-extras --with-accuracy --steps 2 --warmup 1 > $OUT/bench_fp16x3_fold$f.json 2> $OUT/bench_fp16x3_fold$f.err
  python - <<PY
import json
try:
    d = json.loads(open('$OUT/bench_fp16x3_fold$f.json').read().strip().splitlines()[-1])
    print('fp16x3 fold=$f', round(d['value'], 2), d['unit'], 'accuracy', d.get('accuracy', {}).get('max_abs_vs_reference'))
    for k, v in list(d['roofline']['kernels'].items())[:4]:
        print('   ', k, v['avg_us'])
except Exception as e:
    print('fold=$f failed', e); print(open('$OUT/bench_fp16x3_fold$f.err').read()[-1500:])
PY
done
ROHM_GEMM_PRECISION=bf16x6 timeout 300 python bench.py --no-cpu-baseline --no-extras --steps 2 --warmup 1 > $OUT/bench_bf16x6.json 2> $OUT/bench_bf16x6.err; python -c "
import json; d=json.loads(open('$OUT/bench_bf16x6.json').read().strip().splitlines()[-1]); print('bf16x6', round(d['value'],2))"
ROHM_GEMM_PRECISION=fp16x3 timeout 300 python bench.py --no-cpu-baseline --no-extras --steps 2 --warmup 1 --batch 32 > $OUT/bench_fp16x3_b32.json 2> $OUT/bench_fp16x3_b32.err; python -c "
import json; d=json.loads(open('$OUT/bench_fp16x3_b32.json').read().strip().splitlines()[-1]); print('fp16x3 b32', round(d['value'],2))"
