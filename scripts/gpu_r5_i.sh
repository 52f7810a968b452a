#!/bin/bash
# Round 5, call I: the TrajNet loops once more under hipGraph replay (ROHM_TRAJNET_GRAPH=1) against plain launches, same box.
TAG=${1:-r5_i}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd $R
for G in 0 1 0 1; do
  ROHM_TRAJNET_GRAPH=$G timeout 300 python scripts/bench_trajnet.py 1 32 > $OUT/trajnet_graph$G.json 2> $OUT/trajnet_graph$G.err
  python - <<PY
import json
try:
    d = json.load(open('$OUT/trajnet_graph$G.json'))
    print('graph=$G', {k: (v['wall_ms'], v['host_enqueue_ms']) for k, v in d.items()})
except Exception as e:
    print('graph=$G failed', e); print(open('$OUT/trajnet_graph$G.err').read()[-800:])
PY
done
