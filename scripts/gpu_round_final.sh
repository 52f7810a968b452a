#!/bin/bash
# Round-end evidence run on an MI355X box.  Usage: bash scripts/gpu_round_final.sh TAG
TAG=${1:-r1_f}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd $R
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | grep graft | tee $OUT/smoke.txt
python -m pytest tests -m gpu -x -q 2>&1 | tail -3 | tee $OUT/pytest_gpu.txt
python bench.py > $OUT/bench.json 2> $OUT/bench.err; tail -2 $OUT/bench.err; cat $OUT/bench.json
python bench.py --workload scheme --batch 32 --steps 1 --warmup 1 > $OUT/bench_scheme_b32.json 2>> $OUT/bench.err
python bench.py --batch 32 --steps 2 --warmup 1 --no-cpu-baseline > $OUT/bench_b32.json 2>> $OUT/bench.err
python bench.py --workload prox --batch 32 --steps 1 --warmup 1 > $OUT/bench_prox_b32.json 2>> $OUT/bench.err
python scripts/bench_stages.py > $OUT/stages.json 2>> $OUT/bench.err
export TMPDIR=/tmp
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_stats -o p -- python $R/bench.py --no-cpu-baseline > $OUT/rocprof_stats.log 2>&1
find /tmp/prof_stats -name "*kernel_stats.csv" -exec cp {} $OUT/kernel_stats.csv \;
for C in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $C --kernel-trace --output-format csv -d /tmp/prof_$C -o p -- python $R/bench.py --no-cpu-baseline --steps 1 --warmup 0 --ddpm-steps 12 > $OUT/rocprof_$C.log 2>&1
  find /tmp/prof_$C -name "*counter_collection.csv" -exec cp {} $OUT/pmc_$C.csv \;
done
ls $OUT
