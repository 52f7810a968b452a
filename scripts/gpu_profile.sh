#!/bin/bash
# Round profile run on an MI355X box: GPU tests, default bench, rocprofv3 kernel stats of the same bench
# command, and two PMC passes (FETCH_SIZE / WRITE_SIZE) on a shortened run.  Usage: bash scripts/gpu_profile.sh TAG
TAG=${1:-r1}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd $R
python -m pytest tests -m gpu -x -q 2>&1 | tail -3 | tee $OUT/pytest_gpu.txt
python bench.py > $OUT/bench.json 2> $OUT/bench.err; tail -2 $OUT/bench.err; cat $OUT/bench.json
export TMPDIR=/tmp
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_stats -o p -- python $R/bench.py --no-cpu-baseline > $OUT/rocprof_stats.log 2>&1
find /tmp/prof_stats -name "*kernel_stats.csv" -exec cp {} $OUT/kernel_stats.csv \;
for C in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $C --kernel-trace --output-format csv -d /tmp/prof_$C -o p -- python $R/bench.py --no-cpu-baseline --steps 1 --warmup 0 --ddpm-steps 12 > $OUT/rocprof_$C.log 2>&1
  find /tmp/prof_$C -name "*counter_collection.csv" -exec cp {} $OUT/pmc_$C.csv \;
done
ls -la $OUT
