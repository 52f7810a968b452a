#!/bin/bash
# attention rewrite: unit test, PoseNet parity, B=64 / B=32 benches
TAG=${1:-r2_b}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd $R
timeout 600 python -m pytest tests/test_gpu_kernels.py -m gpu -x -q -k attention 2>&1 | tail -8 | tee $OUT/pytest_attn.txt
python bench.py --no-cpu-baseline > $OUT/bench.json 2> $OUT/bench.err; tail -2 $OUT/bench.err; python - <<PY
import json
for f in ['bench']:
    d=json.loads(open('$OUT/'+f+'.json').read().strip().splitlines()[-1])
    print(f, d['value'], {k:(v['avg_us'],v.get('tflops')) for k,v in list(d['roofline']['kernels'].items())[:6]})
PY
python bench.py --no-cpu-baseline --batch 32 > $OUT/bench_b32.json 2>> $OUT/bench.err; python - <<PY
import json
d=json.loads(open('$OUT/bench_b32.json').read().strip().splitlines()[-1])
print('b32', d['value'], {k:(v['avg_us'],v.get('tflops')) for k,v in list(d['roofline']['kernels'].items())[:6]})
PY
timeout 900 python -m pytest tests/test_gpu_posenet.py -m gpu -x -q 2>&1 | tail -4 | tee $OUT/pytest_posenet.txt
