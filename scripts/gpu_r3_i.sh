#!/bin/bash
# Round 3: the fp16x3 mode (two fp16 planes, three MFMA products) -- bench first, kernel tests, whole PoseNet suite under the mode
TAG=${1:-r3_i}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd $R
for m in fp16x3 bf16x6; do
  ROHM_GEMM_PRECISION=$m timeout 300 python bench.py --no-cpu-baseline --no-extras --with-accuracy --steps 2 --warmup 1 > $OUT/bench_$m.json 2> $OUT/bench_$m.err
  python - <<PY
import json
try:
    d = json.loads(open('$OUT/bench_$m.json').read().strip().splitlines()[-1])
    print('$m', round(d['value'], 2), d['unit'], 'gemm frac', d['roofline']['frac'], 'accuracy', d.get('accuracy', {}).get('max_abs_vs_reference'))
    for k, v in list(d['roofline']['kernels'].items())[:6]:
        print('   ', k, v)
except Exception as e:
    print('$m failed', e); print(open('$OUT/bench_$m.err').read()[-1500:])
PY
done
timeout 1500 python -m pytest tests/test_gpu_planes.py -x -q -p no:cacheprovider 2>&1 | tail -12 | tee $OUT/pytest_planes.txt
timeout 1800 python -m pytest tests/test_gpu_precision_ladder.py -x -q -p no:cacheprovider 2>&1 | tail -25 | tee $OUT/pytest_ladder.txt
ROHM_GEMM_PRECISION=fp16x3 timeout 300 python bench.py --no-cpu-baseline --no-extras --steps 2 --warmup 1 --batch 32 > $OUT/bench_fp16x3_b32.json 2> $OUT/bench_fp16x3_b32.err; python -c "
import json; d=json.loads(open('$OUT/bench_fp16x3_b32.json').read().strip().splitlines()[-1]); print('fp16x3 b32', round(d['value'],2))"
