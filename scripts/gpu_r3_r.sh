#!/bin/bash
# Round 3: embed epilogue in the prefetch scheme; fp32 LayerNorm fold (ROHM_POSENET_LNFOLD=1, general instantiation) A/B again now
# that epilogues no longer serialise on their operand loads
TAG=${1:-r3_r}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd $R
timeout 1500 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_posenet.py -x -q -p no:cacheprovider 2>&1 | tail -4 | tee $OUT/pytest_subset.txt
for cfg in "0 64" "1 64" "0 32" "1 32"; do
  set -- $cfg
  ROHM_POSENET_LNFOLD=$1 timeout 300 python bench.py --no-cpu-baseline --no-extras --steps 3 --warmup 1 --batch $2 > $OUT/bench_fold$1_b$2.json 2> $OUT/bench_fold$1_b$2.err
  python - <<PY
import json
d = json.loads(open('$OUT/bench_fold$1_b$2.json').read().strip().splitlines()[-1])
print('fp32 lnfold=$1 b$2', round(d['value'], 2), d['unit'], 'frac', round(d['roofline']['frac'], 4))
for k, v in list(d['roofline']['kernels'].items())[:8]:
    print('   ', k, v['avg_us'])
PY
done
