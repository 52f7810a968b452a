#!/bin/bash
# Round 3, sixth GPU call: where does a chunk of the plane GEMM go?  per-chunk slopes of the three kernels; SQ counters + GUI clock
TAG=${1:-r3_f}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd $R
echo "== W8 (register-staged A)"; timeout 300 python scripts/bench_pp.py > $OUT/pp_w8.txt 2>&1; grep "^{'N'" $OUT/pp_w8.txt
echo "== 2x4 stream (LDS-DMA)"; ROHM_PP_W8=0 timeout 300 python scripts/bench_pp.py > $OUT/pp_stream24.txt 2>&1; grep "^{'N'" $OUT/pp_stream24.txt
echo "== per tile (LDS-DMA)"; ROHM_PP_STREAM=0 timeout 300 python scripts/bench_pp.py > $OUT/pp_pertile.txt 2>&1; grep "^{'N'" $OUT/pp_pertile.txt
cd /tmp
for mode in bf16x6 bf16x3 fp32; do
rm -rf /tmp/pmc_$mode
ROHM_GEMM_PRECISION=$mode timeout 600 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_INSTS_MFMA GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d /tmp/pmc_$mode -o p -- python $R/bench.py --no-cpu-baseline --no-extras --steps 1 --warmup 0 --ddpm-steps 12 > $OUT/rocprof_pmc_$mode.log 2>&1
find /tmp/pmc_$mode -name "*counter_collection.csv" -exec cp {} $OUT/pmc_$mode.csv \;
python - <<PY | tee $OUT/pmc_$mode.txt
import collections, csv, re
rows = collections.defaultdict(lambda: collections.defaultdict(float)); disp = collections.defaultdict(dict)
for r in csv.DictReader(open('$OUT/pmc_$mode.csv')):
    k = re.sub(r'\(.*$', '', r['Kernel_Name'].replace('void ', '')).replace('rohm::', '').replace('(anonymous namespace)::', '')
    if 'gemm' not in k and 'attention' not in k and 'layernorm' not in k: continue
    disp[k][r['Dispatch_Id']] = (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) * 1e-3
    rows[k][r['Counter_Name']] += float(r['Counter_Value'])
print('# $mode: rocprofv3 --pmc (SQ pass + GRBM_GUI_ACTIVE) of bench.py --ddpm-steps 12; PMC serialises kernels')
for k, c in sorted(rows.items(), key=lambda kv: -sum(disp[kv[0]].values())):
    n = len(disp[k]); us = sum(disp[k].values()) / n
    g = lambda x: c.get(x, 0.0) / n
    wc = g('SQ_WAVE_CYCLES') or 1.0
    print(f"{k[:52]:52s} n={n:4d} us={us:7.1f} mfma_busy@2.4={g('SQ_VALU_MFMA_BUSY_CYCLES') / (1024 * us * 2.4e3):.3f} mfma_cyc/inst={g('SQ_VALU_MFMA_BUSY_CYCLES') / max(g('SQ_INSTS_MFMA'), 1):.2f} "
          f"wait_any={g('SQ_WAIT_ANY') / wc:.3f} wait_inst_any={g('SQ_WAIT_INST_ANY') / wc:.3f} wait_inst_lds={g('SQ_WAIT_INST_LDS') / wc:.3f} active_inst={g('SQ_ACTIVE_INST_ANY') / wc:.3f} "
          f"gui_rate_GHz={g('GRBM_GUI_ACTIVE') / us / 1e3:.2f} busy_cyc/us={g('SQ_BUSY_CYCLES') / us:.0f}")
PY
done
