#!/bin/bash
# Round-2 profile evidence on an MI355X box.  Usage: bash scripts/gpu_r2_profile.sh TAG [quick]
#   kernel-trace stats of the default bench, --batch 32, the scheme / prox workloads and the TrajNet loop;
#   PMC passes on a shortened default bench: SQ (MFMA busy, LDS conflicts), FETCH_SIZE, WRITE_SIZE (separate passes).
TAG=${1:-r2}
QUICK=${2:-}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
stats() {   # name, command...
  local name=$1; shift
  rm -rf /tmp/prof_$name
  rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$name -o p -- "$@" > $OUT/rocprof_$name.log 2>&1
  find /tmp/prof_$name -name "*kernel_stats.csv" -exec cp {} $OUT/kernel_stats_$name.csv \;
  python $R/scripts/rocprof_summary.py $OUT/kernel_stats_$name.csv > $OUT/rocprof_kernel_stats_$name.txt 2>&1
  head -12 $OUT/rocprof_kernel_stats_$name.txt
}
pmc() {     # name, counters (quoted), command...
  local name=$1; local ctr=$2; shift; shift
  rm -rf /tmp/pmc_$name
  rocprofv3 --pmc $ctr --kernel-trace --output-format csv -d /tmp/pmc_$name -o p -- "$@" > $OUT/rocprof_pmc_$name.log 2>&1
  find /tmp/pmc_$name -name "*counter_collection.csv" -exec cp {} $OUT/pmc_$name.csv \;
}
SHORT="--no-cpu-baseline --steps 1 --warmup 0 --ddpm-steps 12"
pmc SQ "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_WAVE_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_MFMA GRBM_GUI_ACTIVE" python $R/bench.py $SHORT
python $R/scripts/sq_summary.py $OUT/pmc_SQ.csv $OUT/pmc_sq.json | tee $OUT/pmc_sq.txt
if [ -z "$QUICK" ]; then
  pmc SQ_b32 "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_WAVE_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_MFMA GRBM_GUI_ACTIVE" python $R/bench.py $SHORT --batch 32
  python $R/scripts/sq_summary.py $OUT/pmc_SQ_b32.csv $OUT/pmc_sq_b32.json | tee $OUT/pmc_sq_b32.txt
  pmc FETCH_SIZE FETCH_SIZE python $R/bench.py $SHORT
  pmc WRITE_SIZE WRITE_SIZE python $R/bench.py $SHORT
  python $R/scripts/pmc_summary.py $OUT $OUT/pmc_traffic.json | tee $OUT/pmc_traffic.txt
  stats bench_default python $R/bench.py --no-cpu-baseline
  stats bench_b32 python $R/bench.py --no-cpu-baseline --batch 32 --steps 1 --warmup 1
  stats scheme_b32 python $R/bench.py --workload scheme --batch 32 --steps 1 --warmup 0
  stats prox_b32 python $R/bench.py --workload prox --batch 32 --steps 1 --warmup 0
  stats trajnet_loop python $R/scripts/bench_trajnet.py
fi
ls $OUT
