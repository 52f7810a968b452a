#!/bin/bash
# Round 6, call Q: the stack's meeting flags by plain stores (shipped) against device-scope stores (librohm_hip_flagsc1.so): chain / exchange tests, then same-box A/B.
TAG=${1:-r6_q2}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd $R
( time timeout 1500 python -m pytest tests/test_gpu_chain.py tests/test_gpu_exchange.py tests/test_gpu_config_batches.py -m gpu -q -p no:cacheprovider 2>&1 | tail -5 ) 2>&1 | tee $OUT/pytest_chain.txt
leg() {   # name lib batch
  ROHM_HIP_LIB=$2 timeout 400 python bench.py --no-extras --no-cpu-baseline --batch $3 --steps 3 > $OUT/bench_$1.json 2> $OUT/bench_$1.err
  python - <<PY
import json
try:
    d = json.loads(open('$OUT/bench_$1.json').read().strip().splitlines()[-1])
    print('$1', round(d['value'], 3), {k: v['avg_us'] for k, v in list(d['roofline']['kernels'].items())[:2]}, 'dominant', round(d['roofline']['dominant']['frac'], 4))
except Exception as e:
    print('$1 failed', e); print(open('$OUT/bench_$1.err').read()[-1500:])
PY
}
for round in 1 2; do
  leg plain64_$round $R/rohm_amd/librohm_hip.so 64
  leg sc1_64_$round $R/rohm_amd/librohm_hip_flagsc1.so 64
  leg plain32_$round $R/rohm_amd/librohm_hip.so 32
  leg sc1_32_$round $R/rohm_amd/librohm_hip_flagsc1.so 32
done 2>&1 | tee $OUT/ab.txt
