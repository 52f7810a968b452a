#!/usr/bin/env python
"""Condense a rocprofv3 run (rocpd sqlite `*_results.db` or `*_kernel_stats.csv`) into a small text table
(name, calls, total us, avg us, %) suitable for committing under profiles/."""
import csv
import glob
import os
import sqlite3
import sys


def short(name, n=110):
    name = name.replace('void ', '')
    if 'distribution_elementwise_grid_stride_kernel' in name:
        return 'at::native::distribution_elementwise_grid_stride_kernel<normal> (torch.randn)'
    return name if len(name) <= n else name[:n - 3] + '...'


def from_db(path):
    con = sqlite3.connect(path)
    rows = list(con.execute('select name, total_calls, total_duration, average, percentage from top_kernels'))
    return [(short(r[0]), int(r[1]), float(r[2]), float(r[3]), float(r[4])) for r in rows]


def from_csv(path):
    out = []
    with open(path) as f:
        for r in csv.DictReader(f):
            out.append((short(r['Name']), int(r['Calls']), float(r['TotalDurationNs']) / 1e3,
                        float(r['AverageNs']) / 1e3, float(r['Percentage'])))
    return out


def main():
    src = sys.argv[1]
    if os.path.isdir(src):
        cands = glob.glob(os.path.join(src, '**', '*kernel_stats.csv'), recursive=True) + \
            glob.glob(os.path.join(src, '**', '*_results.db'), recursive=True)
        src = cands[0]
    rows = from_csv(src) if src.endswith('.csv') else from_db(src)
    rows.sort(key=lambda r: -r[2])
    print(f'# rocprofv3 --kernel-trace --stats summary ({os.path.basename(src)})')
    print(f'{"kernel":112s} {"calls":>7s} {"total_us":>12s} {"avg_us":>10s} {"pct":>6s}')
    for n, c, t, a, p in rows:
        if p < 0.005 and c < 50:
            continue
        print(f'{n:112s} {c:7d} {t:12.1f} {a:10.2f} {p:6.2f}')


if __name__ == '__main__':
    main()
