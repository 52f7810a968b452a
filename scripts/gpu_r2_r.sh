#!/bin/bash
# Round-2, second session: SQ counters of the split-bf16 GEMM (where a wave's cycles go).  Usage: bash scripts/gpu_r2_r.sh TAG
TAG=${1:-r2_r}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
rm -rf /tmp/pmc_planes
ROHM_GEMM_PRECISION=bf16x6 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d /tmp/pmc_planes -o p -- python $R/bench.py --no-cpu-baseline --steps 1 --warmup 0 --ddpm-steps 12 > $OUT/rocprof_pmc.log 2>&1
find /tmp/pmc_planes -name "*counter_collection.csv" -exec cp {} $OUT/pmc_planes.csv \;
python - <<PY
import collections, csv, re
rows = collections.defaultdict(lambda: collections.defaultdict(float)); disp = collections.defaultdict(dict)
for r in csv.DictReader(open('$OUT/pmc_planes.csv')):
    k = re.sub(r'\(.*$', '', r['Kernel_Name'].replace('void ', ''))
    if 'gemm' not in k and 'attention' not in k: continue
    disp[k][r['Dispatch_Id']] = (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) * 1e-3
    rows[k][r['Counter_Name']] += float(r['Counter_Value'])
for k, c in sorted(rows.items(), key=lambda kv: -sum(disp[kv[0]].values())):
    n = len(disp[k]); us = sum(disp[k].values()) / n
    g = lambda x: c.get(x, 0.0) / n
    wc = g('SQ_WAVE_CYCLES') or 1.0
    print(f"{k[:58]:58s} n={n:4d} us={us:7.1f} mfma_busy/(4*256*us*2.4e3)={g('SQ_VALU_MFMA_BUSY_CYCLES') / (1024 * us * 2.4e3):.3f} "
          f"wait_any={g('SQ_WAIT_ANY') / wc:.3f} wait_inst_any={g('SQ_WAIT_INST_ANY') / wc:.3f} wait_inst_lds={g('SQ_WAIT_INST_LDS') / wc:.3f} "
          f"lds_conf={g('SQ_LDS_BANK_CONFLICT') / max(g('SQ_LDS_IDX_ACTIVE'), 1):.3f} lds_active/busy={g('SQ_LDS_IDX_ACTIVE') / max(g('SQ_BUSY_CYCLES'), 1):.3f} "
          f"gui_GHz={g('GRBM_GUI_ACTIVE') / us / 1e3:.2f}")
PY
tail -2 $OUT/rocprof_pmc.log
