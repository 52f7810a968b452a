#!/bin/bash
# Round 3, first GPU call: headline bench under a power / sclk trace, the new split-bf16 path (kernel tests, ladder tests,
# bench second lines), GEMM phase timeline on the current kernels (diag build), LBS measurement.
TAG=${1:-r3_a}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd $R
echo "== power-traced headline bench (fp32, no CPU leg)"
timeout 300 python scripts/power_trace.py --out $OUT/power_sclk_bench.csv --hz 20 -- python bench.py --no-cpu-baseline --steps 3 --warmup 1 > $OUT/power_sclk_bench.txt 2> $OUT/bench_default.err
tail -c 1500 $OUT/power_sclk_bench.txt
echo "== kernel tests of the plane path"
timeout 900 python -m pytest tests/test_gpu_planes.py -x -q -p no:cacheprovider 2>&1 | tail -15 | tee $OUT/pytest_planes.txt
echo "== bf16x6 / bf16x3 bench"
for m in bf16x6 bf16x3; do
  ROHM_GEMM_PRECISION=$m timeout 300 python bench.py --no-cpu-baseline --steps 2 --warmup 1 > $OUT/bench_$m.json 2> $OUT/bench_$m.err
  python - <<PY
import json
try:
    d = json.loads(open('$OUT/bench_$m.json').read().strip().splitlines()[-1])
    print('$m', round(d['value'], 2), d['unit'], 'gemm frac', d['roofline']['frac'])
    for k, v in list(d['roofline']['kernels'].items())[:9]:
        print('   ', k, v)
except Exception as e:
    print('$m failed', e); print(open('$OUT/bench_$m.err').read()[-1500:])
PY
done
echo "== ladder tests (whole PoseNet suite under bf16x6)"
timeout 1500 python -m pytest tests/test_gpu_precision_ladder.py -x -q -p no:cacheprovider 2>&1 | tail -25 | tee $OUT/pytest_ladder.txt
echo "== fp32 default: kernels + posenet"
timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_posenet.py -x -q -p no:cacheprovider 2>&1 | tail -8 | tee $OUT/pytest_fp32.txt
echo "== GEMM timeline (diag build)"
ROHM_HIP_LIB=$R/rohm_amd/librohm_hip_diag.so ROHM_GEMM_VARIANT=7 timeout 300 python scripts/gemm_timeline.py > $OUT/gemm_timeline.txt 2>&1; tail -40 $OUT/gemm_timeline.txt
echo "== LBS"
timeout 300 python scripts/bench_lbs.py 32 > $OUT/lbs_b32.json 2> $OUT/lbs.err; cat $OUT/lbs_b32.json | head -c 3000; tail -3 $OUT/lbs.err
cd /tmp && rm -rf /tmp/prof_lbs && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_lbs -o p -- python $R/scripts/bench_lbs.py 32 > /dev/null 2> $OUT/rocprof_lbs.log
find /tmp/prof_lbs -name "*kernel_stats.csv" -exec cp {} $OUT/kernel_stats_lbs_b32.csv \;
python $R/scripts/rocprof_summary.py $OUT/kernel_stats_lbs_b32.csv > $OUT/rocprof_kernel_stats_lbs_b32.txt 2>&1; head -12 $OUT/rocprof_kernel_stats_lbs_b32.txt
