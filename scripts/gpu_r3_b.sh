#!/bin/bash
# Round 3, second GPU call: persistent stream kernel of the plane GEMM -- parity, A/B against one workgroup per tile, full bench
# record with second_line + configs, zero-operand timeline (DVFS), force-dist test.
TAG=${1:-r3_b}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd $R
echo "== kernel tests of the plane path (stream kernel default + the per-tile kernel in a child)"
timeout 1200 python -m pytest tests/test_gpu_planes.py -x -q -p no:cacheprovider 2>&1 | tail -15 | tee $OUT/pytest_planes.txt
echo "== bf16x6 A/B: stream vs per-tile"
for st in 1 0; do
  ROHM_PP_STREAM=$st ROHM_GEMM_PRECISION=bf16x6 timeout 300 python bench.py --no-cpu-baseline --no-extras --steps 2 --warmup 1 > $OUT/bench_bf16x6_stream$st.json 2> $OUT/bench_bf16x6_stream$st.err
  python - <<PY
import json
try:
    d = json.loads(open('$OUT/bench_bf16x6_stream$st.json').read().strip().splitlines()[-1])
    print('bf16x6 stream=$st', round(d['value'], 2), d['unit'], 'gemm frac', d['roofline']['frac'])
    for k, v in list(d['roofline']['kernels'].items())[:8]:
        print('   ', k, v)
except Exception as e:
    print('stream=$st failed', e); print(open('$OUT/bench_bf16x6_stream$st.err').read()[-1500:])
PY
done
ROHM_GEMM_PRECISION=bf16x3 timeout 300 python bench.py --no-cpu-baseline --no-extras --steps 2 --warmup 1 > $OUT/bench_bf16x3.json 2> $OUT/bench_bf16x3.err; python -c "
import json; d=json.loads(open('$OUT/bench_bf16x3.json').read().strip().splitlines()[-1]); print('bf16x3', round(d['value'],2))"
ROHM_GEMM_PRECISION=bf16x6 timeout 300 python bench.py --no-cpu-baseline --no-extras --steps 2 --warmup 1 --batch 32 > $OUT/bench_bf16x6_b32.json 2> $OUT/bench_bf16x6_b32.err; python -c "
import json; d=json.loads(open('$OUT/bench_bf16x6_b32.json').read().strip().splitlines()[-1]); print('bf16x6 b32', round(d['value'],2))"
echo "== ladder tests"
timeout 1500 python -m pytest tests/test_gpu_precision_ladder.py -x -q -p no:cacheprovider 2>&1 | tail -25 | tee $OUT/pytest_ladder.txt
echo "== force-dist"
timeout 900 python -m pytest tests/test_bench_launcher.py -x -q -m gpu -p no:cacheprovider 2>&1 | tail -8 | tee $OUT/pytest_forcedist.txt
echo "== full default bench record (headline + cpu leg + second_line + configs)"
( time timeout 1200 python bench.py > $OUT/bench_full.json 2> $OUT/bench_full.err ) 2>&1 | tail -3
python - <<PY
import json
try:
    d = json.loads(open('$OUT/bench_full.json').read().strip().splitlines()[-1])
    print('headline', round(d['value'], 2), 'frac', round(d['roofline']['frac'], 3), 'cpu', d.get('cpu_baseline', {}).get('value'), d.get('cpu_baseline', {}).get('kind'))
    print('second_line', {k: d['second_line'].get(k) for k in ('value', 'dtype', 'error')}, d['second_line'].get('accuracy'))
    for k, v in d['configs'].items():
        print('  ', k, v.get('value'), v.get('error'), v.get('child_wall_s'))
except Exception as e:
    print('full bench failed', e); print(open('$OUT/bench_full.err').read()[-2000:])
PY
echo "== GEMM timeline, zero operands (diag build)"
ROHM_TL_ZERO=1 ROHM_HIP_LIB=$R/rohm_amd/librohm_hip_diag.so ROHM_GEMM_VARIANT=7 timeout 300 python scripts/gemm_timeline.py > $OUT/gemm_timeline_zero.txt 2>&1; grep -E "^==|main loop" $OUT/gemm_timeline_zero.txt
