#!/bin/bash
# Round 3: fp32 GEMM epilogue with the operands of all units read before the first store (gemm_f32.hip) + ControlNet side stream:
# kernel / PoseNet / TrajNet / scheme parity, headline bench (no extras) and B = 32
TAG=${1:-r3_q}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd $R
timeout 1500 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_posenet.py tests/test_gpu_trajnet.py tests/test_gpu_scheme.py -x -q -p no:cacheprovider 2>&1 | tail -6 | tee $OUT/pytest_subset.txt
timeout 300 python bench.py --no-cpu-baseline --no-extras --steps 3 --warmup 1 > $OUT/bench_default.json 2> $OUT/bench_default.err
python - <<PY
import json
d = json.loads(open('$OUT/bench_default.json').read().strip().splitlines()[-1])
print('fp32 b64', round(d['value'], 2), d['unit'], 'frac', d['roofline']['frac'])
for k, v in list(d['roofline']['kernels'].items())[:8]:
    print('   ', k, v['avg_us'])
PY
timeout 300 python bench.py --no-cpu-baseline --no-extras --steps 3 --warmup 1 --batch 32 > $OUT/bench_b32.json 2> $OUT/bench_b32.err
python - <<PY
import json
d = json.loads(open('$OUT/bench_b32.json').read().strip().splitlines()[-1])
print('fp32 b32', round(d['value'], 2), d['unit'], 'frac', d['roofline']['frac'])
for k, v in list(d['roofline']['kernels'].items())[:8]:
    print('   ', k, v['avg_us'])
PY
timeout 300 python scripts/bench_trajnet.py 1 32 > $OUT/trajnet_loop.json 2> $OUT/trajnet_loop.err; python -c "
import json; d=json.load(open('$OUT/trajnet_loop.json')); [print(k, {a: b for a, b in v.items() if a != 'kernels'}) for k, v in d.items()]"
