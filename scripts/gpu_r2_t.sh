#!/bin/bash
# rocprofv3 kernel stats of the TrajNet / TrajControl 100-step loops at B = 32 after the second pass on the step.
TAG=${1:-r2_t}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp && rm -rf /tmp/prof_traj
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_traj -o p -- python $R/scripts/bench_trajnet.py 32 > $OUT/trajnet_loop_b32.json 2> $OUT/rocprof.log
find /tmp/prof_traj -name "*kernel_stats.csv" -exec cp {} $OUT/kernel_stats_trajnet_loop_b32.csv \;
python $R/scripts/rocprof_summary.py $OUT/kernel_stats_trajnet_loop_b32.csv > $OUT/rocprof_kernel_stats_trajnet_loop_b32.txt 2>&1; head -14 $OUT/rocprof_kernel_stats_trajnet_loop_b32.txt
cd $R && python bench.py --workload prox --batch 32 --steps 1 --warmup 1 > $OUT/bench_prox_b32.json 2>> $OUT/rocprof.log; python -c "
import json; d=json.loads(open('$OUT/bench_prox_b32.json').read().strip().splitlines()[-1]); print('prox', round(d['value'],2), d['unit'])"
