#!/usr/bin/env python
"""Per-kernel MFMA utilisation / LDS conflict summary from ONE rocprofv3 --pmc pass (counter_collection CSV) holding
SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_WAVE_CYCLES SQ_LDS_BANK_CONFLICT
SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_MFMA (8 SQ slots) + GRBM_GUI_ACTIVE.

Derived per launch (averages over the launches of a kernel; the end/start timestamps of the same rows give the duration):
  mfma_busy     = SQ_VALU_MFMA_BUSY_CYCLES / (4 SIMDs x 256 CUs x duration x shader clock)    [rocprof's MfmaUtil numerator]
  mfma_tflops   = SQ_INSTS_VALU_MFMA_MOPS_F32 x 512 flop / duration     (fp32 MFMA only)
  of_peak       = mfma_tflops / 157.3
  lds_conflict  = SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE              (extra LDS cycles / all LDS cycles)
PMC collection serialises kernels and runs at a lower clock (MI355X_MICROARCH.md, DVFS note), so durations here are
NOT the bench's; ratios are what this table is for.  usage: sq_summary.py <counter_collection.csv> [out.json]"""
import collections
import csv
import json
import re
import sys

PEAK = 157.3e12
CUS, SIMDS = 256, 4


def short(n):
    n = n.replace('void ', '')
    return re.sub(r'\(.*$', '', n)


def main():
    rows = collections.defaultdict(lambda: collections.defaultdict(float))
    disp = collections.defaultdict(dict)
    for r in csv.DictReader(open(sys.argv[1])):
        k = short(r['Kernel_Name'])
        if not k.startswith('rohm::'):
            continue
        d = r['Dispatch_Id']
        disp[k][d] = (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) * 1e-9
        rows[k][r['Counter_Name']] += float(r['Counter_Value'])
    out = {}
    hdr = f'{"kernel":64s} {"n":>6s} {"us":>8s} {"mfmaTF":>7s} {"ofpeak":>6s} {"busy":>6s} {"GHz":>5s} {"ldsconf":>7s} {"waitLDS":>7s}'
    print(hdr)
    for k, c in sorted(rows.items(), key=lambda kv: -sum(disp[kv[0]].values())):
        n = len(disp[k])
        dur = sum(disp[k].values()) / n
        g = lambda name: c.get(name, 0.0) / n
        flops = g('SQ_INSTS_VALU_MFMA_MOPS_F32') * 512.0
        gui = g('GRBM_GUI_ACTIVE')
        ghz = gui / dur / 1e9 if dur > 0 else 0.0
        # GRBM_GUI_ACTIVE may be summed over XCDs: report the raw per-second rate and let the reader see it
        busy = g('SQ_VALU_MFMA_BUSY_CYCLES')
        busy_frac_wall = busy / (SIMDS * CUS * dur * 2.4e9) if dur > 0 else 0.0
        conf = g('SQ_LDS_BANK_CONFLICT') / g('SQ_LDS_IDX_ACTIVE') if g('SQ_LDS_IDX_ACTIVE') else 0.0
        wl = g('SQ_WAIT_INST_LDS') / g('SQ_WAVE_CYCLES') if g('SQ_WAVE_CYCLES') else 0.0
        out[k] = {'launches': n, 'avg_us_under_pmc': dur * 1e6, 'mfma_tflops': flops / dur / 1e12 if dur else 0.0,
                  'mfma_of_peak': flops / dur / PEAK if dur else 0.0, 'mfma_busy_cycles_per_launch': busy,
                  'mfma_busy_frac_at_2p4GHz': busy_frac_wall, 'grbm_gui_active_per_launch': gui,
                  'gui_active_rate_GHz': ghz, 'lds_bank_conflict_frac': conf, 'wait_inst_lds_frac_of_wave_cycles': wl,
                  'sq_busy_cycles': g('SQ_BUSY_CYCLES'), 'sq_wave_cycles': g('SQ_WAVE_CYCLES'), 'insts_mfma': g('SQ_INSTS_MFMA')}
        print(f'{k[:64]:64s} {n:6d} {dur * 1e6:8.2f} {out[k]["mfma_tflops"]:7.1f} {out[k]["mfma_of_peak"]:6.3f} '
              f'{busy_frac_wall:6.3f} {ghz:5.2f} {conf:7.4f} {wl:7.4f}')
    if len(sys.argv) > 2:
        json.dump({'source': 'rocprofv3 --pmc (SQ pass) + GRBM_GUI_ACTIVE, bench.py --ddpm-steps 12', 'kernels': out},
                  open(sys.argv[2], 'w'), indent=1)


if __name__ == '__main__':
    main()
