#!/bin/bash
# Round 3: QKV tile's bias row through LDS (gemm_f32.hip COL_LDS): kernel + PoseNet parity, headline bench twice
TAG=${1:-r3_t}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd $R
timeout 1500 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_posenet.py -x -q -p no:cacheprovider 2>&1 | tail -4 | tee $OUT/pytest_subset.txt
for rep in a b; do
  timeout 300 python bench.py --no-cpu-baseline --no-extras --steps 3 --warmup 1 > $OUT/bench_default_$rep.json 2> $OUT/bench_default_$rep.err
  python - <<PY
import json
d = json.loads(open('$OUT/bench_default_$rep.json').read().strip().splitlines()[-1])
print('fp32 b64', round(d['value'], 2), d['unit'], 'frac', round(d['roofline']['frac'], 4))
for k, v in list(d['roofline']['kernels'].items())[:4]:
    print('   ', k, v['avg_us'])
PY
done
