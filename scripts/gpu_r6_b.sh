#!/bin/bash
# Round 6, call B: (1) same-box A/B of the library with the (dormant) phase stamps against the previous tree, B = 64 and 32, two rounds;
# (2) the phase timeline of the stack at B = 64 / 32; (3) the new exchange / drop-in tests + the chain / exchange / posenet suites.
TAG=${1:-r6_b}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd $R
leg() {   # name lib batch
  if [ $2 = default ]; then unset ROHM_HIP_LIB; else export ROHM_HIP_LIB=$R/rohm_amd/librohm_hip_$2.so; fi
  timeout 400 python bench.py --no-extras --no-cpu-baseline --batch $3 > $OUT/bench_$1.json 2> $OUT/bench_$1.err
  python - <<PY
import json
try:
    d = json.loads(open('$OUT/bench_$1.json').read().strip().splitlines()[-1])
    a = d['roofline'].get('attention') or {}
    print('$1', round(d['value'], 3), {k: v['avg_us'] for k, v in list(d['roofline']['kernels'].items())[:3]}, 'attention', a.get('frac'), a.get('share_of_launch'), a.get('in_stack_error'))
except Exception as e:
    print('$1 failed', e); print(open('$OUT/bench_$1.err').read()[-800:])
PY
  unset ROHM_HIP_LIB
}
for round in 1 2; do
  leg new64_$round default 64
  leg prev64_$round prev 64
  leg new32_$round default 32
  leg prev32_$round prev 32
done
timeout 600 python scripts/stack_timeline.py $OUT/stack_phase_timeline.json 64 32 2>&1 | tee $OUT/stack_phase_timeline.txt
( time timeout 2400 python -m pytest tests/test_gpu_exchange.py tests/test_dropin.py tests/test_gpu_chain.py tests/test_gpu_posenet.py tests/test_gpu_guidance.py tests/test_gpu_kernels.py -m gpu -q -p no:cacheprovider --durations=8 2>&1 | tail -30 ) 2>&1 | tee $OUT/pytest_subset.txt
