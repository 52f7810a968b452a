#!/bin/bash
# Round 2, first GPU call: the new PROX / API tests, then the whole GPU suite, default bench, egobody + prox workloads.
TAG=${1:-r2_a}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd $R
python bench.py > $OUT/bench.json 2> $OUT/bench.err; tail -2 $OUT/bench.err; cat $OUT/bench.json | cut -c1-400
timeout 900 python -m pytest tests/test_gpu_guidance.py tests/test_gpu_posenet.py tests/test_gpu_trajnet.py -m gpu -x -q 2>&1 | tail -15 | tee $OUT/pytest_new.txt
timeout 900 python -m pytest tests -m gpu -x -q --deselect tests/test_gpu_guidance.py --deselect tests/test_gpu_posenet.py --deselect tests/test_gpu_trajnet.py 2>&1 | tail -5 | tee $OUT/pytest_rest.txt
python bench.py --workload egobody --batch 32 --steps 1 --warmup 0 > $OUT/bench_egobody_b32.json 2>> $OUT/bench.err; cut -c1-300 $OUT/bench_egobody_b32.json
python bench.py --workload prox --batch 32 --steps 1 --warmup 0 > $OUT/bench_prox_b32.json 2>> $OUT/bench.err; cut -c1-300 $OUT/bench_prox_b32.json
tail -5 $OUT/bench.err
