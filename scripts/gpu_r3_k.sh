#!/bin/bash
# Round 3: check of the output-head tile choice (144 x 128 at B = 64): headline bench, PoseNet suite, planes child tests
TAG=${1:-r3_k}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd $R
timeout 300 python bench.py --no-cpu-baseline --no-extras > $OUT/bench_default.json 2> $OUT/bench_default.err
python -c "
import json; d=json.loads(open('$OUT/bench_default.json').read().strip().splitlines()[-1]); print('headline', round(d['value'],2), 'frac', round(d['roofline']['frac'],3)); [print('   ', k, v) for k, v in d['roofline']['kernels'].items() if 'out_t' in k or 'embed' in k]"
ROHM_GEMM_PRECISION=fp16x3 timeout 300 python bench.py --no-cpu-baseline --no-extras > $OUT/bench_fp16x3.json 2> $OUT/bench_fp16x3.err
python -c "
import json; d=json.loads(open('$OUT/bench_fp16x3.json').read().strip().splitlines()[-1]); print('fp16x3', round(d['value'],2))"
( time timeout 1500 python -m pytest tests/test_gpu_posenet.py tests/test_gpu_planes.py tests/test_gpu_kernels.py -x -q -p no:cacheprovider 2>&1 | tail -4 ) 2>&1 | tee $OUT/pytest.txt
