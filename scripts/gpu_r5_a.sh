#!/bin/bash
# Round 5, call A: the exchange-safety work on hardware (fallback + re-run, CU mask, graph capture, recycled workspaces), the
# guidance tests at the configs' batch sizes, the attention timeline under both DMA issue orders, quick headline / B = 32 legs.
TAG=${1:-r5_a}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd $R
( time timeout 1500 python -m pytest tests/test_gpu_exchange.py tests/test_gpu_posenet.py tests/test_gpu_kernels.py tests/test_gpu_guidance.py \
    -m gpu -q -p no:cacheprovider --durations=12 -s 2>&1 | grep -v "^$" | tail -60 ) 2>&1 | tee $OUT/pytest_exchange_guidance.txt
for KF in 0 1; do
  hipcc --offload-arch=gfx950 -O3 -std=c++17 -DAT_TIMELINE -DAT_KFIRST=$KF scripts/probes/attn_timeline.hip -o /tmp/attn_tl_$KF 2>&1 | tail -3
  for B in 32 64; do
    echo "=== AT_KFIRST=$KF B=$B" | tee -a $OUT/attn_timeline.txt
    timeout 120 /tmp/attn_tl_$KF $B $([ $B = 32 ] && echo 1 || echo 0) 2>&1 | tee -a $OUT/attn_timeline.txt | head -3
  done
done
timeout 400 python bench.py --no-extras --no-cpu-baseline > $OUT/bench_b64.json 2> $OUT/bench_b64.err
timeout 400 python bench.py --no-extras --no-cpu-baseline --batch 32 > $OUT/bench_b32.json 2> $OUT/bench_b32.err
python - <<PY
import json
for n in ('b64', 'b32'):
    try:
        d = json.loads(open('$OUT/bench_%s.json' % n).read().strip().splitlines()[-1])
        print(n, round(d['value'], 2), 'frac', round(d['roofline']['frac'], 3), d['config'].get('exchange_mode'))
        for k, v in list(d['roofline']['kernels'].items())[:8]:
            print('    ', k, v['avg_us'], v.get('tflops'))
    except Exception as e:
        print(n, 'failed', e); print(open('$OUT/bench_%s.err' % n).read()[-1500:])
PY
