#!/bin/bash
# Round 3: ControlNet branch of the TrajControl sample loop on a second stream -- parity (tests/test_gpu_trajnet.py), loop times
# with the side stream on / off, and the LayerNorm-fold A/B of the plane GEMMs once more (two threads per row in the reduce)
TAG=${1:-r3_o}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd $R
timeout 900 python -m pytest tests/test_gpu_trajnet.py tests/test_gpu_inference.py -x -q -p no:cacheprovider 2>&1 | tail -6 | tee $OUT/pytest_trajnet.txt
for on in 1 0; do
  ROHM_TRAJ_CTRL_STREAM=$on timeout 300 python scripts/bench_trajnet.py 1 32 > $OUT/trajnet_loop_side$on.json 2> $OUT/trajnet_loop_side$on.err
  python - <<PY
import json
try:
    d = json.load(open('$OUT/trajnet_loop_side$on.json'))
    for k, v in d.items():
        print('side=$on', k, {a: b for a, b in v.items() if a != 'kernels'})
except Exception as e:
    print('side=$on failed', e); print(open('$OUT/trajnet_loop_side$on.err').read()[-1500:])
PY
done
timeout 600 python -m pytest tests/test_gpu_planes.py -x -q -p no:cacheprovider -k "ln_fold" 2>&1 | tail -3 | tee $OUT/pytest_planes_fold.txt
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
