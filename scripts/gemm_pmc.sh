#!/bin/bash
# PMC probe of the GEMM microbenchmark (effective clock, MFMA busy, wait buckets)
R=${GRAFT_REPO_ROOT:-$(pwd)}
export TMPDIR=/tmp
cd /tmp
rocprofv3 -L 2>/dev/null | grep -E "SQ_VALU_MFMA_BUSY|SQ_WAIT_INST_ANY|SQ_WAIT_ANY|SQ_ACTIVE_INST_ANY|SQ_WAVE_CYCLES|GRBM_GUI_ACTIVE|SQ_BUSY_CYCLES|SQ_WAIT_INST_LDS|SQ_INSTS_VALU_MFMA|SQ_ACTIVE_INST_MISC|SQ_INST_CYCLES_VMEM|SQ_ACTIVE_INST_LDS|SQ_ACTIVE_INST_VMEM" | head -30
GEMM_SHAPES=n rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAIT_INST_LDS --kernel-trace --output-format csv -d /tmp/pmcg -o p -- python $R/scripts/gemm_bench.py > /tmp/pmcg.log 2>&1
tail -3 /tmp/pmcg.log
python - <<'PY'
import csv, glob, collections
f = glob.glob('/tmp/pmcg/**/*counter_collection.csv', recursive=True)[0]
rows = collections.defaultdict(lambda: collections.defaultdict(list))
dur = collections.defaultdict(list)
for r in csv.DictReader(open(f)):
    if 'gemm_f32' not in r['Kernel_Name']: continue
    key = (r['Kernel_Name'][:50], r['Grid_Size'])
    rows[key][r['Counter_Name']].append(float(r['Counter_Value']))
    dur[key].append(int(r['End_Timestamp']) - int(r['Start_Timestamp']))
for k, c in rows.items():
    n = len(c['SQ_WAVE_CYCLES'])
    avg = {a: sum(v) / len(v) for a, v in c.items()}
    d = sum(dur[k]) / len(dur[k])
    print(k, 'n', n, 'dur_us %.1f' % (d / 1e3), ' '.join(f'{a}={v:.3g}' for a, v in avg.items()), 'clk_GHz %.2f' % (avg.get('GRBM_GUI_ACTIVE', 0) / d))
PY
