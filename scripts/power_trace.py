#!/usr/bin/env python
"""Sample the GPU's power, shader clock, temperature and busy percentage while a command runs (is the
fp32 GEMM power-limited, i.e. does the package sit at its cap while sclk sags below the 2.4 GHz nominal?).
Sources, first one that answers: the amdgpu hwmon / sysfs files (>= 20 Hz, no subprocess), else `amd-smi metric` / `rocm-smi`
(about 1-2 Hz).  Prints a summary (whole run and the busy part) and writes every sample as CSV.
usage (GPU box): python scripts/power_trace.py --out gpurun_out/x/power.csv --hz 20 -- python bench.py --no-cpu-baseline"""
import argparse
import glob
import json
import os
import subprocess
import sys
import time


def _read(path):
    try:
        with open(path) as f:
            return f.read().strip()
    except OSError:
        return None


class Sysfs:
    def __init__(self):
        self.dev = None
        for d in sorted(glob.glob('/sys/class/drm/card*/device')):
            if _read(os.path.join(d, 'vendor')) == '0x1002' and glob.glob(os.path.join(d, 'hwmon/hwmon*')):
                self.dev = d
                break
        if self.dev is None:
            raise RuntimeError('no amdgpu hwmon directory')
        self.hw = sorted(glob.glob(os.path.join(self.dev, 'hwmon/hwmon*')))[0]
        self.power = next((p for p in ('power1_average', 'power1_input') if _read(os.path.join(self.hw, p)) not in (None, '')), None)
        self.cap = _read(os.path.join(self.hw, 'power1_cap'))
        self.temps = [os.path.basename(p) for p in sorted(glob.glob(os.path.join(self.hw, 'temp*_input')))]
        if self.power is None and _read(os.path.join(self.hw, 'freq1_input')) is None:
            raise RuntimeError('hwmon has neither power nor frequency')

    def describe(self):
        return {'source': 'sysfs', 'device': self.dev, 'hwmon': self.hw, 'power_file': self.power,
                'power_cap_w': (int(self.cap) / 1e6 if self.cap and self.cap.isdigit() else None), 'temps': self.temps}

    def sample(self):
        p = _read(os.path.join(self.hw, self.power)) if self.power else None
        f = _read(os.path.join(self.hw, 'freq1_input'))
        sclk = int(f) / 1e6 if f and f.isdigit() else None
        if sclk is None:
            cur = _read(os.path.join(self.dev, 'pp_dpm_sclk'))
            if cur:
                for line in cur.splitlines():
                    if line.rstrip().endswith('*'):
                        sclk = float(line.split(':')[1].strip().split('M')[0])
        busy = _read(os.path.join(self.dev, 'gpu_busy_percent'))
        temps = [_read(os.path.join(self.hw, t)) for t in self.temps]
        return {'power_w': int(p) / 1e6 if p and p.isdigit() else None, 'sclk_mhz': sclk,
                'busy': int(busy) if busy and busy.isdigit() else None,
                'temp_c': max((int(t) / 1e3 for t in temps if t and t.lstrip('-').isdigit()), default=None)}


class Smi:
    def __init__(self):
        for cmd in (['amd-smi', 'metric', '-g', '0', '-p', '-c', '-u', '--json'], ['rocm-smi', '--showpower', '--showclocks', '--showuse', '--json']):
            try:
                out = subprocess.run(cmd, capture_output=True, text=True, timeout=20)
                if out.returncode == 0 and out.stdout.strip():
                    self.cmd = cmd
                    self.first = out.stdout
                    return
            except (OSError, subprocess.TimeoutExpired):
                continue
        raise RuntimeError('neither amd-smi nor rocm-smi answered')

    def describe(self):
        return {'source': ' '.join(self.cmd), 'first_reply': self.first[:1500]}

    @staticmethod
    def _find(obj, keys):
        """first numeric value under any key containing one of `keys` (case-insensitive), depth-first"""
        if isinstance(obj, dict):
            for k, v in obj.items():
                if any(s in k.lower() for s in keys):
                    if isinstance(v, dict) and 'value' in v:
                        v = v['value']
                    try:
                        return float(str(v).split()[0].strip('()MmHhZzWw'))
                    except ValueError:
                        pass
                r = Smi._find(v, keys)
                if r is not None:
                    return r
        elif isinstance(obj, list):
            for v in obj:
                r = Smi._find(v, keys)
                if r is not None:
                    return r
        return None

    def sample(self):
        out = subprocess.run(self.cmd, capture_output=True, text=True, timeout=20).stdout
        try:
            d = json.loads(out)
        except ValueError:
            return {'power_w': None, 'sclk_mhz': None, 'busy': None, 'temp_c': None}
        return {'power_w': self._find(d, ['socket_power', 'average graphics package power', 'current socket graphics package power', 'power (w)']),
                'sclk_mhz': self._find(d, ['gfx_0', 'sclk clock speed', 'sclk']), 'busy': self._find(d, ['gfx_activity', 'gpu use']),
                'temp_c': None}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--out', required=True)
    ap.add_argument('--hz', type=float, default=20.0)
    ap.add_argument('cmd', nargs=argparse.REMAINDER)
    a = ap.parse_args()
    cmd = a.cmd[1:] if a.cmd and a.cmd[0] == '--' else a.cmd
    src = None
    errs = []
    for cls in (Sysfs, Smi):
        try:
            src = cls()
            break
        except RuntimeError as e:
            errs.append(str(e))
    if src is None:
        print(json.dumps({'error': errs}))
        return subprocess.call(cmd)
    info = src.describe()
    rows = []
    t0 = time.time()
    proc = subprocess.Popen(cmd)
    dt = 1.0 / a.hz
    while proc.poll() is None:
        t = time.time()
        s = src.sample()
        s['t'] = round(t - t0, 3)
        rows.append(s)
        time.sleep(max(0.0, dt - (time.time() - t)))
    os.makedirs(os.path.dirname(os.path.abspath(a.out)), exist_ok=True)
    with open(a.out, 'w') as f:
        f.write('t_s,power_w,sclk_mhz,busy_pct,temp_c\n')
        for r in rows:
            f.write(','.join('' if r.get(k) is None else str(r[k]) for k in ('t', 'power_w', 'sclk_mhz', 'busy', 'temp_c')) + '\n')

    def stats(sel, key):
        v = sorted(r[key] for r in sel if r.get(key) is not None)
        if not v:
            return None
        return {'n': len(v), 'min': v[0], 'p10': v[len(v) // 10], 'median': v[len(v) // 2], 'p90': v[(len(v) * 9) // 10], 'max': v[-1],
                'mean': round(sum(v) / len(v), 1)}
    pw = [r['power_w'] for r in rows if r.get('power_w') is not None]
    hot = [r for r in rows if r.get('power_w') is not None and pw and r['power_w'] >= 0.7 * max(pw)]
    summ = {'info': info, 'samples': len(rows), 'seconds': round(time.time() - t0, 1), 'rc': proc.returncode,
            'all': {k: stats(rows, k) for k in ('power_w', 'sclk_mhz', 'busy', 'temp_c')},
            'under_load(power >= 0.7 max)': {k: stats(hot, k) for k in ('power_w', 'sclk_mhz', 'busy', 'temp_c')}}
    print(json.dumps(summ, indent=1))
    return proc.returncode


if __name__ == '__main__':
    sys.exit(main())
