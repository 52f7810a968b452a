#!/bin/bash
# Round 3: wave_sum64 (VALU lane swaps + DPP instead of six ds_bpermute) in LayerNorm / GroupNorm: bit-identity probe, parity subset, benches
TAG=${1:-r3_y}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd $R
hipcc --offload-arch=gfx950 -O3 -std=c++17 -Irohm_amd/csrc -Iinclude -w -o /tmp/wsp scripts/probes/wave_sum_probe.hip && /tmp/wsp | tee $OUT/wave_sum_probe.txt
timeout 1500 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_posenet.py tests/test_gpu_trajnet.py tests/test_gpu_planes.py -x -q -p no:cacheprovider -k "not one_tile" 2>&1 | tail -4 | tee $OUT/pytest_subset.txt
for b in 64 32; do
  timeout 300 python bench.py --no-cpu-baseline --no-extras --steps 3 --warmup 1 --batch $b > $OUT/bench_b$b.json 2> $OUT/bench_b$b.err
  python - <<PY
import json
d = json.loads(open('$OUT/bench_b$b.json').read().strip().splitlines()[-1])
print('fp32 b$b', round(d['value'], 2), d['unit'], 'frac', round(d['roofline']['frac'], 4), 'layernorm', d['roofline']['kernels'].get('layernorm', {}).get('avg_us'))
PY
done
timeout 300 python scripts/bench_trajnet.py 1 32 > $OUT/trajnet_loop.json 2> $OUT/trajnet_loop.err; python -c "
import json; d=json.load(open('$OUT/trajnet_loop.json')); [print(k, {a: b for a, b in v.items() if a != 'kernels'}) for k, v in d.items()]"
