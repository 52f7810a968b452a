#!/bin/bash
# Round 6, call G: the round-aware stack gate (chain tests incl. 48 / 56 clips as one part-filled round of 4-part workgroups) and the sweep again.
TAG=${1:-r6_g}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd $R
( time timeout 1500 python -m pytest tests/test_gpu_chain.py -m gpu -q -p no:cacheprovider 2>&1 | tail -6 ) 2>&1 | tee $OUT/pytest_chain.txt
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
for B in 40 48 56 60 72 96 120; do
  leg default_b$B ROHM_NOOP=1 $B "--ddpm-steps 100 --steps 2 --warmup 1"
done
leg pergemm_b60 ROHM_POSENET_CHAIN=0 60 "--ddpm-steps 100 --steps 2 --warmup 1"
leg pergemm_b120 ROHM_POSENET_CHAIN=0 120 "--ddpm-steps 100 --steps 2 --warmup 1"
leg stackany_b120 ROHM_POSENET_CHAIN_ANY=1 120 "--ddpm-steps 100 --steps 2 --warmup 1"
