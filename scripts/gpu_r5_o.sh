#!/bin/bash
# Round 5, call O: epilogue store order of the stack's GELU / QKV / embed phases -- row-major over a lane's units (default) against gemm_f32.hip's
# column-major order (experiment library); B = 64 and 32, same box, two rounds; WRITE_SIZE pass of both at B = 64.
TAG=${1:-r5_o}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd $R
for round in 1 2; do
for leg in "default 64" "colmajor 64" "default 32" "colmajor 32"; do
  set -- $leg
  if [ $1 = default ]; then unset ROHM_HIP_LIB; else export ROHM_HIP_LIB=$R/rohm_amd/librohm_hip_$1.so; fi
  timeout 400 python bench.py --no-extras --no-cpu-baseline --batch $2 > $OUT/bench_$1_b$2_$round.json 2> $OUT/bench_$1_b$2_$round.err
  python - <<PY
import json
try:
    d = json.loads(open('$OUT/bench_$1_b$2_$round.json').read().strip().splitlines()[-1])
    print('$1 b$2 $round', round(d['value'], 2), {k: v['avg_us'] for k, v in list(d['roofline']['kernels'].items())[:2]})
except Exception as e:
    print('$1 failed', e); print(open('$OUT/bench_$1_b$2_$round.err').read()[-800:])
PY
done
done
cd /tmp
for leg in default colmajor; do
  if [ $leg = default ]; then unset ROHM_HIP_LIB; else export ROHM_HIP_LIB=$R/rohm_amd/librohm_hip_$leg.so; fi
  rm -rf /tmp/pmc_w_$leg
  timeout 600 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d /tmp/pmc_w_$leg -o p -- python $R/bench.py --no-cpu-baseline --no-extras --steps 1 --warmup 0 --ddpm-steps 12 > $OUT/rocprof_w_$leg.log 2>&1
  python - <<PY
import csv, glob
tot, n = 0.0, 0
for f in glob.glob('/tmp/pmc_w_$leg/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        if 'encoder_stack_kernel' in r['Kernel_Name'] and r['Counter_Name'] == 'WRITE_SIZE':
            tot += float(r['Counter_Value']); n += 1
print('$leg stack WRITE_SIZE per launch (MB):', round(tot / max(n, 1) * 1024 / 1e6, 1), 'launches', n)
PY
done
