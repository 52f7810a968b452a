#!/bin/bash
# Round 5, call C: the encoder stack (attention inside, layers looped in the kernel) -- parity tests of both chain forms, then
# stack / layer / off legs of the headline and B = 32 workloads (candidate first).
TAG=${1:-r5_c}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd $R
( time timeout 900 python -m pytest tests/test_gpu_chain.py -m gpu -q -p no:cacheprovider -s --durations=8 2>&1 | grep -v "^$" | tail -60 ) 2>&1 | tee $OUT/pytest_chain.txt
for leg in "stack 64" "layer 64" "0 64" "stack 32" "layer 32" "0 32"; do
  set -- $leg
  ROHM_POSENET_CHAIN=$1 timeout 400 python bench.py --no-extras --no-cpu-baseline --batch $2 > $OUT/bench_chain_$1_b$2.json 2> $OUT/bench_chain_$1_b$2.err
done
python - <<PY
import json
for n in ('stack_b64', 'layer_b64', '0_b64', 'stack_b32', 'layer_b32', '0_b32'):
    try:
        d = json.loads(open('$OUT/bench_chain_%s.json' % n).read().strip().splitlines()[-1])
        print(n, round(d['value'], 2), 'frac', round(d['roofline']['frac'], 3), 'e2e', round(d.get('e2e_frac_executed') or 0, 3))
        for k, v in list(d['roofline']['kernels'].items())[:6]:
            print('    ', k, v['launches'], v['avg_us'], v.get('tflops'))
    except Exception as e:
        print(n, 'failed', e); print(open('$OUT/bench_chain_%s.err' % n).read()[-1500:])
PY
