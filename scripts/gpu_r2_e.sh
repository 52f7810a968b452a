#!/bin/bash
TAG=${1:-r2_e}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd $R
timeout 900 python -m pytest tests/test_gpu_frames.py tests/test_gpu_scheme.py tests/test_gpu_kernels.py -m gpu -x -q 2>&1 | tail -6 | tee $OUT/pytest_new.txt
python scripts/bench_trajnet.py 1 32 > $OUT/trajnet_loop.json 2> $OUT/err.txt; python - <<PY
import json
d=json.load(open('$OUT/trajnet_loop.json'))
for k,v in d.items():
    print(k, v['wall_ms'], v['launches'])
    for kk,vv in list(v['kernels'].items())[:12]: print('    ',kk,vv)
PY
