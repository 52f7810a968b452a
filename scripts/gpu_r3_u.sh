#!/bin/bash
# Round 3: PMC passes on the final binary (separate passes, --pmc with --kernel-trace only): SQ (MFMA busy, LDS conflicts) at B = 64 and 32,
# FETCH_SIZE, WRITE_SIZE -> pmc_sq.json / pmc_traffic.json (bench.py reads the committed copies as `archived` records)
TAG=${1:-r3_u}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
pmc() {     # name, counters (quoted), command...
  local name=$1; local ctr=$2; shift; shift
  rm -rf /tmp/pmc_$name
  timeout 600 rocprofv3 --pmc $ctr --kernel-trace --output-format csv -d /tmp/pmc_$name -o p -- "$@" > $OUT/rocprof_pmc_$name.log 2>&1
  find /tmp/pmc_$name -name "*counter_collection.csv" -exec cp {} $OUT/pmc_$name.csv \;
}
SHORT="--no-cpu-baseline --no-extras --steps 1 --warmup 0 --ddpm-steps 12"
SQC="SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_WAVE_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_MFMA GRBM_GUI_ACTIVE"
pmc SQ "$SQC" python $R/bench.py $SHORT
python $R/scripts/sq_summary.py $OUT/pmc_SQ.csv $OUT/pmc_sq.json | tee $OUT/pmc_sq.txt | head -20
pmc FETCH_SIZE FETCH_SIZE python $R/bench.py $SHORT
pmc WRITE_SIZE WRITE_SIZE python $R/bench.py $SHORT
python $R/scripts/pmc_summary.py $OUT $OUT/pmc_traffic.json | tee $OUT/pmc_traffic.txt | head -30
pmc SQ_b32 "$SQC" python $R/bench.py $SHORT --batch 32
python $R/scripts/sq_summary.py $OUT/pmc_SQ_b32.csv $OUT/pmc_sq_b32.json | tee $OUT/pmc_sq_b32.txt | head -12
rm -f $OUT/pmc_SQ.csv $OUT/pmc_SQ_b32.csv $OUT/pmc_FETCH_SIZE.csv $OUT/pmc_WRITE_SIZE.csv
ls $OUT
