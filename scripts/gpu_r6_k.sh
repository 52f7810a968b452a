#!/bin/bash
# Round 6, call K: the closing phase with the software-pipelined loop: tests, phase timeline, A/B against ROHM_POSENET_STACK_TAIL=0.
TAG=${1:-r6_k}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd $R
( time timeout 1500 python -m pytest tests/test_gpu_chain.py tests/test_gpu_config_batches.py -m gpu -q -s -p no:cacheprovider -k "one_launch or every_clip" 2>&1 | grep -E "max\||passed|failed|Error|error|assert" | tail -20 ) 2>&1 | tee $OUT/pytest_tail.txt
timeout 600 python scripts/stack_timeline.py $OUT/stack_phase_timeline.json 64 32 2>&1 | grep -E "encoder_stack|head_update|finish_skew|attention  " | tee $OUT/stack_phase_timeline.txt
leg() {   # name tail batch
  ROHM_POSENET_STACK_TAIL=$2 timeout 400 python bench.py --no-extras --no-cpu-baseline --batch $3 > $OUT/bench_$1.json 2> $OUT/bench_$1.err
  python - <<PY
import json
try:
    d = json.loads(open('$OUT/bench_$1.json').read().strip().splitlines()[-1])
    print('$1', round(d['value'], 3), {k: v['avg_us'] for k, v in list(d['roofline']['kernels'].items())[:3]}, 'dominant', round(d['roofline']['dominant']['frac'], 4))
except Exception as e:
    print('$1 failed', e); print(open('$OUT/bench_$1.err').read()[-1500:])
PY
}
for round in 1 2; do
  leg tail64_$round 1 64
  leg notail64_$round 0 64
  leg tail32_$round 1 32
  leg notail32_$round 0 32
done
