#!/bin/bash
# Round-2 evidence run on an MI355X box.  Usage: bash scripts/gpu_r2_final.sh TAG
TAG=${1:-r2_f}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd $R
# the headline first, on a cool chip (measurement hygiene)
python bench.py > $OUT/bench.json 2> $OUT/bench.err; tail -2 $OUT/bench.err; cut -c1-300 $OUT/bench.json
python bench.py --batch 32 --steps 2 --warmup 1 --no-cpu-baseline > $OUT/bench_b32.json 2>> $OUT/bench.err
python bench.py --workload scheme --batch 32 --steps 1 --warmup 1 > $OUT/bench_scheme_b32.json 2>> $OUT/bench.err
python bench.py --workload prox --batch 32 --steps 1 --warmup 1 > $OUT/bench_prox_b32.json 2>> $OUT/bench.err
python bench.py --workload egobody --batch 32 --steps 1 --warmup 0 > $OUT/bench_egobody_b32.json 2>> $OUT/bench.err
python scripts/bench_trajnet.py 1 32 256 > $OUT/trajnet_loop.json 2>> $OUT/bench.err
python scripts/bench_stages.py > $OUT/stages.json 2>> $OUT/bench.err
for f in bench_b32 bench_scheme_b32 bench_prox_b32 bench_egobody_b32; do python -c "
import json; d=json.loads(open('$OUT/$f.json').read().strip().splitlines()[-1]); print('$f', round(d['value'],2), d['unit'])"; done
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | grep graft | tee $OUT/smoke.txt
bash scripts/gpu_r2_profile.sh $TAG > $OUT/profile.log 2>&1; tail -3 $OUT/profile.log
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -4 | tee $OUT/pytest_gpu.txt
ls $OUT | head -60
