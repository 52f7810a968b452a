#!/bin/bash
# Round 4: stream-K output head generalised to tiles in up to three pieces (B = 32: 144 tiles on 256 CUs): parity, A/B at B = 32 / 64
TAG=${1:-r4_n}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd $R
timeout 600 python -m pytest tests/test_gpu_kernels.py -x -q -p no:cacheprovider -s -k "output_process" 2>&1 | grep -v "^$" | tail -14 | tee $OUT/pytest_kernels.txt
timeout 600 python -m pytest tests/test_gpu_posenet.py -x -q -p no:cacheprovider -k "variants or exchange or golden" 2>&1 | tail -4 | tee $OUT/pytest_posenet.txt
for cfg in "32 1 1" "32 0 1" "64 1 1" "32 1 1" "32 0 1"; do
  set -- $cfg
  F=$OUT/bench_b$1_sk$2_hoist$3
  ROHM_POSENET_HEAD_SK=$2 ROHM_POSENET_COND_HOIST=$3 timeout 300 python bench.py --no-cpu-baseline --no-extras --steps 3 --warmup 1 --batch $1 > $F.json 2> $F.err
  python - <<PY
import json
d = json.loads(open('$F.json').read().strip().splitlines()[-1])
print('fp32 b$1 head_sk=$2 hoist=$3', round(d['value'], 3), d['unit'], 'frac', round(d['roofline']['frac'], 4))
for k, v in list(d['roofline']['kernels'].items())[:9]:
    print('   ', k, v['avg_us'])
PY
done 2>&1 | tee $OUT/ab.txt
