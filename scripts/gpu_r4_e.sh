#!/bin/bash
# Round 4: LBS skinning retuned (workgroup count, posed vertices prefetched one tile ahead, component-major LDS image of the ELL kernel)
TAG=${1:-r4_e}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd $R
timeout 900 python -m pytest tests/test_gpu_rederive.py -x -q -p no:cacheprovider -k "lbs" 2>&1 | tail -5 | tee $OUT/pytest_lbs.txt
timeout 300 python scripts/bench_lbs.py 32 dense > $OUT/lbs_b32_mfma.json 2> $OUT/lbs_b32_mfma.err
ROHM_LBS_SKIN=ell timeout 300 python scripts/bench_lbs.py 32 dense > $OUT/lbs_b32_dense_ell.json 2> $OUT/lbs_b32_dense_ell.err
timeout 300 python scripts/bench_lbs.py 32 sparse > $OUT/lbs_b32_sparse.json 2> $OUT/lbs_b32_sparse.err
python - <<PY
import json
for n in ('mfma', 'dense_ell', 'sparse'):
    try:
        d = json.load(open('$OUT/lbs_b32_%s.json' % n))
        print(n, d['skinning_mode'], 'with verts', d['with_vertices']['wall_us_per_call'], 'us', {k: v['us'] for k, v in d['with_vertices']['kernels'].items()})
    except Exception as e:
        print(n, 'failed', e, open('$OUT/lbs_b32_%s.err' % n).read()[-600:])
PY
timeout 900 python -m pytest "tests/test_bench_launcher.py" -x -q -p no:cacheprovider -k "scheme_workload" 2>&1 | tail -30 | tee $OUT/pytest_launcher.txt
