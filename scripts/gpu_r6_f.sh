#!/bin/bash
# Round 6, call F: the closing phase with three staging buffers (tests, A/B vs ROHM_POSENET_STACK_TAIL=0, timeline), and the batch-size
# sweep ADVICE r5 asked for: the stack against one launch per GEMM at batch sizes whose workgroup count does not fill whole rounds.
TAG=${1:-r6_f}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd $R
( time timeout 1500 python -m pytest tests/test_gpu_chain.py tests/test_gpu_config_batches.py -m gpu -q -s -p no:cacheprovider -k "one_launch or every_clip" 2>&1 | grep -E "max\||passed|failed|Error|error|assert" | tail -20 ) 2>&1 | tee $OUT/pytest_tail.txt
leg() {   # name env batch extra-args
  env $2 timeout 400 python bench.py --no-extras --no-cpu-baseline --batch $3 $4 > $OUT/bench_$1.json 2> $OUT/bench_$1.err
  python - <<PY
import json
try:
    d = json.loads(open('$OUT/bench_$1.json').read().strip().splitlines()[-1])
    print('$1', round(d['value'], 3), 'ms/pass', round(d['ms_per_step'], 1), {k: v['avg_us'] for k, v in list(d['roofline']['kernels'].items())[:3]})
except Exception as e:
    print('$1 failed', e); print(open('$OUT/bench_$1.err').read()[-1500:])
PY
}
for round in 1 2; do
  leg tail64_$round ROHM_POSENET_STACK_TAIL=1 64
  leg notail64_$round ROHM_POSENET_STACK_TAIL=0 64
  leg tail32_$round ROHM_POSENET_STACK_TAIL=1 32
  leg notail32_$round ROHM_POSENET_STACK_TAIL=0 32
done
timeout 600 python scripts/stack_timeline.py $OUT/stack_phase_timeline.json 64 32 2>&1 | tee $OUT/stack_phase_timeline.txt
for B in 40 48 56 72 96 128; do
  leg stack_b$B ROHM_NOOP=1 $B "--ddpm-steps 100 --steps 2 --warmup 1"
  leg pergemm_b$B ROHM_POSENET_CHAIN=0 $B "--ddpm-steps 100 --steps 2 --warmup 1"
done
