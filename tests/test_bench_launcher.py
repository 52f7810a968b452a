"""CPU: `bench.py --gpus N` really starts N ranks.  The launcher, rank binding, barrier-bracketed
timing, MAX-over-ranks reduction and the result all-gather are the code paths the GPU run uses; only the sampler is
a stand-in and the backend is gloo (ROHM_BENCH_SELFTEST=1).  The record says so in every field."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(ROOT, 'bench.py')


def _run(argv, env_extra=None, drop=('WORLD_SIZE', 'RANK', 'LOCAL_RANK')):
    env = {k: v for k, v in os.environ.items() if k not in drop}
    env.update(env_extra or {})
    return subprocess.run([sys.executable, BENCH] + argv, env=env, capture_output=True, text=True, timeout=600)


def _json_line(out):
    lines = [ln for ln in out.splitlines() if ln.startswith('{')]
    assert len(lines) == 1, out
    return json.loads(lines[0])


def test_gpus_2_spawns_two_ranks():
    r = _run(['--gpus', '2', '--backend', 'gloo', '--steps', '2', '--warmup', '1', '--batch', '3'],
             {'ROHM_BENCH_SELFTEST': '1'})
    assert r.returncode == 0, r.stderr[-2000:]
    rec = _json_line(r.stdout)
    assert rec['n_gpus'] == 2 and rec['world_size'] == 2 and rec['ranks_seen'] == [0, 1]
    assert rec['gathered_clips'] == 6 and rec['steps'] == 2 and rec['warmup'] == 1
    # MAX over ranks: rank 1 sleeps 20 ms per pass, rank 0 10 ms
    assert rec['ms_per_step'] >= 19.0
    assert rec['data'] == 'selftest-stub' and rec['value'] == 0.0


def test_single_rank_selftest_and_gloo_needs_the_switch():
    r = _run(['--gpus', '1', '--backend', 'gloo', '--steps', '1', '--warmup', '0', '--batch', '2'],
             {'ROHM_BENCH_SELFTEST': '1'})
    assert r.returncode == 0, r.stderr[-2000:]
    assert _json_line(r.stdout)['n_gpus'] == 1
    r = _run(['--gpus', '1', '--backend', 'gloo'])
    assert r.returncode != 0 and 'no CPU fallback' in r.stderr


def test_world_size_mismatch_is_an_error_not_a_silent_single_rank():
    r = _run(['--gpus', '2', '--backend', 'gloo'], {'ROHM_BENCH_SELFTEST': '1', 'WORLD_SIZE': '1', 'RANK': '0'}, drop=())
    assert r.returncode != 0 and '--gpus 2 but WORLD_SIZE=1' in r.stderr


def test_more_ranks_than_gpus_is_refused():
    import torch
    if torch.cuda.is_available() and torch.cuda.device_count() >= 2:
        return
    r = _run(['--gpus', '2'])
    assert r.returncode != 0 and 'GPU(s) visible' in r.stderr


def test_driver_style_launch_under_torch_distributed_run():
    """The way the driver starts N > 1: `python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N`
    (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* come from the launcher, bench.py must not spawn again)."""
    import socket
    with socket.socket() as so:
        so.bind(('127.0.0.1', 0))
        port = so.getsockname()[1]
    env = {k: v for k, v in os.environ.items() if k not in ('WORLD_SIZE', 'RANK', 'LOCAL_RANK')}
    env['ROHM_BENCH_SELFTEST'] = '1'
    r = subprocess.run([sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', '2',
                        '--master-addr', '127.0.0.1', '--master-port', str(port), BENCH, '--gpus', '2', '--backend', 'gloo',
                        '--steps', '1', '--warmup', '0', '--batch', '2'], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    rec = _json_line(r.stdout)
    assert rec['n_gpus'] == 2 and rec['ranks_seen'] == [0, 1] and rec['gathered_clips'] == 4
