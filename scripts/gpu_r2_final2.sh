#!/bin/bash
# Round-2 (second session) evidence run on an MI355X box.  Usage: bash scripts/gpu_r2_final2.sh TAG
TAG=${1:-r2_s}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd $R
# the headline first, on a cool chip (measurement hygiene)
python bench.py > $OUT/bench.json 2> $OUT/bench.err; tail -1 $OUT/bench.err; cut -c1-200 $OUT/bench.json
# rocprofv3 kernel stats of the same command (without the host-side CPU baseline)
export TMPDIR=/tmp
( cd /tmp && rm -rf /tmp/prof_bench && rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_bench -o p -- python $R/bench.py --no-cpu-baseline > $OUT/rocprof_bench_default.log 2>&1 )
find /tmp/prof_bench -name "*kernel_stats.csv" -exec cp {} $OUT/kernel_stats_bench_default.csv \;
python scripts/rocprof_summary.py $OUT/kernel_stats_bench_default.csv > $OUT/rocprof_kernel_stats_bench_default.txt 2>&1; head -8 $OUT/rocprof_kernel_stats_bench_default.txt
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | grep graft | tee $OUT/smoke.txt
ROHM_GEMM_PRECISION=bf16x6 python bench.py --steps 2 --warmup 1 --no-cpu-baseline > $OUT/bench_bf16x6.json 2>> $OUT/bench.err
ROHM_GEMM_PRECISION=bf16x3 python bench.py --steps 2 --warmup 1 --no-cpu-baseline > $OUT/bench_bf16x3.json 2>> $OUT/bench.err
python bench.py --batch 32 --steps 2 --warmup 1 --no-cpu-baseline > $OUT/bench_b32.json 2>> $OUT/bench.err
python bench.py --workload scheme --batch 32 --steps 1 --warmup 1 > $OUT/bench_scheme_b32.json 2>> $OUT/bench.err
for f in bench_bf16x6 bench_bf16x3 bench_b32 bench_scheme_b32; do python -c "
import json; d=json.loads(open('$OUT/$f.json').read().strip().splitlines()[-1]); print('$f', round(d['value'],2), d['unit'])"; done
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -4 | tee $OUT/pytest_gpu.txt
ls $OUT | head -40
