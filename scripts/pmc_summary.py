#!/usr/bin/env python
"""Aggregate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes (counter_collection CSVs) into per-kernel HBM-side
traffic per launch.  Units: the counters are in KiB; on gfx950 FETCH_SIZE reports 1/2 of a wide coalesced read
(MI355X_MICROARCH.md §HBM; confirmed here on layernorm: 9.2 MiB counted for an 18.0 MiB read), so the read side
is doubled.  WRITE_SIZE is used as reported (exact on the layernorm / attention outputs).
usage: pmc_summary.py <dir with pmc_FETCH_SIZE.csv pmc_WRITE_SIZE.csv> [out.json]"""
import collections
import csv
import json
import os
import re
import sys


def short(n):
    n = n.replace('void ', '')
    return re.sub(r'\(.*$', '', n)


def load(path, counter):
    agg = collections.defaultdict(lambda: [0, 0.0])
    for r in csv.DictReader(open(path)):
        if r['Counter_Name'] != counter:
            continue
        k = short(r['Kernel_Name'])
        agg[k][0] += 1
        agg[k][1] += float(r['Counter_Value'])
    return agg


def main():
    d = sys.argv[1]
    f = load(os.path.join(d, 'pmc_FETCH_SIZE.csv'), 'FETCH_SIZE')
    w = load(os.path.join(d, 'pmc_WRITE_SIZE.csv'), 'WRITE_SIZE')
    rows = {}
    for k in sorted(set(f) | set(w)):
        if not k.startswith('rohm::'):
            continue
        nf, vf = f.get(k, (0, 0.0))
        nw, vw = w.get(k, (0, 0.0))
        rd = 2.0 * vf / nf * 1024 if nf else 0.0
        wr = vw / nw * 1024 if nw else 0.0
        rows[k] = {'launches': max(nf, nw), 'read_bytes_per_launch': rd, 'write_bytes_per_launch': wr,
                   'hbm_bytes_per_launch': rd + wr}
    # the fp32-MFMA family: the GEMM launches and, from round 5, the encoder chain / stack launches that contain most of them
    gem = {k: v for k, v in rows.items() if 'gemm_f32_kernel' in k or 'encoder_stack_kernel' in k or 'encoder_chain_kernel' in k}
    n = sum(v['launches'] for v in gem.values())
    fam = sum(v['hbm_bytes_per_launch'] * v['launches'] for v in gem.values()) / max(n, 1)
    out = {'source': 'rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes), bench.py --ddpm-steps 12, B=64',
           'correction': 'FETCH_SIZE x2 (gfx950), KiB -> bytes', 'gemm_family_bytes_per_launch': fam, 'kernels': rows}
    print(f'{"kernel":70s} {"launches":>8s} {"read MB":>10s} {"write MB":>10s}')
    for k, v in sorted(rows.items(), key=lambda kv: -kv[1]['hbm_bytes_per_launch'] * kv[1]['launches']):
        print(f'{k:70s} {v["launches"]:8d} {v["read_bytes_per_launch"] / 1e6:10.2f} '
              f'{v["write_bytes_per_launch"] / 1e6:10.2f}')
    print(f'GEMM family: {fam / 1e6:.1f} MB per launch')
    if len(sys.argv) > 2:
        json.dump(out, open(sys.argv[2], 'w'), indent=1)


if __name__ == '__main__':
    main()
