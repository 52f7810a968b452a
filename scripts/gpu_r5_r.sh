#!/bin/bash
# Round 5, call R: the dense LBS skinning kernel with the previous tile's epilogue under the next tile's MFMAs -- parity tests + timing.
TAG=${1:-r5_r}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd $R
timeout 600 python -m pytest tests/test_gpu_rederive.py tests/test_gpu_frames.py -m gpu -q -p no:cacheprovider -k "lbs or smplx or vert or skin" 2>&1 | tail -4
timeout 300 python scripts/bench_lbs.py 32 dense > $OUT/lbs_b32_mfma.json 2> $OUT/lbs.err
timeout 300 python scripts/bench_lbs.py 32 dense > $OUT/lbs_b32_mfma_2.json 2>> $OUT/lbs.err
python - <<PY
import json
for n in ('', '_2'):
    try:
        d = json.load(open('$OUT/lbs_b32_mfma%s.json' % n)); print('lbs dense', d['with_vertices']['wall_us_per_call'], {k: v['us'] for k, v in d['with_vertices']['kernels'].items()})
    except Exception as e:
        print('lbs failed', e); print(open('$OUT/lbs.err').read()[-800:])
PY
