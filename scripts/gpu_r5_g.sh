#!/bin/bash
# Round 5, call G: stack with leading embed / QKV phases + finish_pack -- parity tests (chain, posenet, exchange), LBS with the pinned skinning
# schedule, then same-box A/B legs (candidate first).
TAG=${1:-r5_g}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd $R
( time timeout 1500 python -m pytest tests/test_gpu_chain.py tests/test_gpu_posenet.py tests/test_gpu_exchange.py tests/test_gpu_rederive.py -m gpu -q -p no:cacheprovider --durations=6 2>&1 | grep -v "^$" | tail -30 ) 2>&1 | tee $OUT/pytest_subset.txt
timeout 300 python scripts/bench_lbs.py 32 dense > $OUT/lbs_b32_mfma.json 2> $OUT/lbs.err
python - <<PY
import json
try:
    d = json.load(open('$OUT/lbs_b32_mfma.json')); print('lbs dense', d['with_vertices']['wall_us_per_call'], {k: v['us'] for k, v in d['with_vertices']['kernels'].items()})
except Exception as e:
    print('lbs failed', e)
PY
for leg in "1 1 64" "0 0 64" "1 1 32" "0 0 32" "1 0 64" "0 1 64"; do
  set -- $leg
  ROHM_POSENET_STACK_FRONT=$1 ROHM_POSENET_FINISH_PACK=$2 timeout 400 python bench.py --no-extras --no-cpu-baseline --batch $3 > $OUT/bench_front$1_fp$2_b$3.json 2> $OUT/bench_front$1_fp$2_b$3.err
done
python - <<PY
import json
for n in ('front1_fp1_b64', 'front0_fp0_b64', 'front1_fp0_b64', 'front0_fp1_b64', 'front1_fp1_b32', 'front0_fp0_b32'):
    try:
        d = json.loads(open('$OUT/bench_%s.json' % n).read().strip().splitlines()[-1])
        print(n, round(d['value'], 2), 'frac', round(d['roofline']['frac'], 3), 'e2e', round(d.get('e2e_frac_executed') or 0, 3))
        for k, v in list(d['roofline']['kernels'].items())[:6]:
            print('    ', k, v['launches'], v['avg_us'], v.get('tflops'))
    except Exception as e:
        print(n, 'failed', e); print(open('$OUT/bench_%s.err' % n).read()[-1500:])
PY
