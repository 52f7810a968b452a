#!/bin/bash
# Round-2, second session: where the TrajNet step's time sits -- launch-shape sweep with power-of-two split counts and
# per-launch-shape event times.  Usage: bash scripts/gpu_r2_n.sh TAG
TAG=${1:-r2_n}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd $R
timeout 300 python scripts/bench_trajnet.py --sweep 1 32 > $OUT/trajnet_sweep.json 2> $OUT/sweep.err; cat $OUT/trajnet_sweep.json; tail -3 $OUT/sweep.err
timeout 200 python scripts/bench_trajnet.py --detail 32 1 2 0 > $OUT/trajnet_detail_b32.txt 2> $OUT/detail.err; cat $OUT/trajnet_detail_b32.txt
timeout 200 python scripts/bench_trajnet.py --detail 32 1 2 1 > $OUT/trajnet_detail_b32_pow2.txt 2>> $OUT/detail.err; head -40 $OUT/trajnet_detail_b32_pow2.txt
timeout 200 python scripts/bench_trajnet.py --detail 1 1 2 0 > $OUT/trajnet_detail_b1.txt 2>> $OUT/detail.err; head -30 $OUT/trajnet_detail_b1.txt
tail -3 $OUT/detail.err
