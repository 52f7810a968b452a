#!/bin/bash
# Round 3, fourth GPU call: W8 stream kernel (eight waves along N, A by LDS-DMA) -- parity, bench, rocprofv3 kernel stats of the bf16x6 bench
TAG=${1:-r3_d}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd $R
for m in bf16x6 bf16x3; do
  ROHM_GEMM_PRECISION=$m timeout 300 python bench.py --no-cpu-baseline --no-extras --steps 2 --warmup 1 > $OUT/bench_$m.json 2> $OUT/bench_$m.err
  python - <<PY
import json
try:
    d = json.loads(open('$OUT/bench_$m.json').read().strip().splitlines()[-1])
    print('$m', round(d['value'], 2), d['unit'], 'gemm frac', d['roofline']['frac'])
    for k, v in list(d['roofline']['kernels'].items())[:8]:
        print('   ', k, v)
except Exception as e:
    print('$m failed', e); print(open('$OUT/bench_$m.err').read()[-1500:])
PY
done
timeout 1200 python -m pytest tests/test_gpu_planes.py -x -q -p no:cacheprovider  2>&1 | tail -6 | tee $OUT/pytest_planes.txt
ROHM_GEMM_PRECISION=bf16x6 timeout 300 python bench.py --no-cpu-baseline --no-extras --steps 2 --warmup 1 --batch 32 > $OUT/bench_bf16x6_b32.json 2> $OUT/bench_bf16x6_b32.err; python -c "
import json; d=json.loads(open('$OUT/bench_bf16x6_b32.json').read().strip().splitlines()[-1]); print('bf16x6 b32', round(d['value'],2)); [print('   ', k, v) for k, v in list(d['roofline']['kernels'].items())[:6]]"
cd /tmp && rm -rf /tmp/prof_x6 && ROHM_GEMM_PRECISION=bf16x6 timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_x6 -o p -- python $R/bench.py --no-cpu-baseline --no-extras --steps 1 --warmup 1 > $OUT/bench_bf16x6_rocprof.json 2> $OUT/rocprof_x6.log
find /tmp/prof_x6 -name "*kernel_stats.csv" -exec cp {} $OUT/kernel_stats_bench_bf16x6.csv \;
python $R/scripts/rocprof_summary.py $OUT/kernel_stats_bench_bf16x6.csv > $OUT/rocprof_kernel_stats_bench_bf16x6.txt 2>&1; head -16 $OUT/rocprof_kernel_stats_bench_bf16x6.txt
cd $R
echo "== LBS with the 8-frames-per-block skinning kernel"
timeout 300 python scripts/bench_lbs.py 32 > $OUT/lbs_b32.json 2> $OUT/lbs.err; python -c "
import json; d=json.load(open('$OUT/lbs_b32.json')); print(d['with_vertices'])"
timeout 600 python -m pytest tests/test_gpu_rederive.py tests/test_gpu_frames.py -x -q -p no:cacheprovider 2>&1 | tail -3
