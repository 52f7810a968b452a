#!/bin/bash
# Round-2, second session: the pipelined split-bf16 GEMM (ROHM_GEMM_PRECISION=bf16x6 | bf16x3) -- parity and the labelled second bench line.
TAG=${1:-r2_q}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd $R
for m in bf16x6 bf16x3; do
ROHM_GEMM_PRECISION=$m timeout 600 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_posenet.py -x -q -k "gemm or forward_vs_reference_golden or loop8" 2>&1 | tail -5 | tee $OUT/pytest_$m.txt
ROHM_GEMM_PRECISION=$m python bench.py --steps 2 --warmup 1 --no-cpu-baseline > $OUT/bench_$m.json 2>> $OUT/bench.err
ROHM_GEMM_PRECISION=$m python bench.py --batch 32 --steps 2 --warmup 1 --no-cpu-baseline > $OUT/bench_b32_$m.json 2>> $OUT/bench.err
done
python - <<PY
import json
for m in ('bf16x6', 'bf16x3'):
    for f in ('bench', 'bench_b32'):
        try:
            d = json.loads(open('$OUT/%s_%s.json' % (f, m)).read().strip().splitlines()[-1])
            k = d['roofline']['kernels']
            print(f, m, round(d['value'], 3), 'clips/s', {n: v['avg_us'] for n, v in k.items()})
        except Exception as e:
            print(f, m, 'failed', e)
PY
tail -5 $OUT/bench.err
