"""CPU: `bench.py --gpus N` really starts N ranks.  The launcher, rank binding, barrier-bracketed
timing, MAX-over-ranks reduction and the result all-gather are the code paths the GPU run uses; only the sampler is
a stand-in and the backend is gloo (ROHM_BENCH_SELFTEST=1).  The record says so in every field."""
import json
import os
import subprocess
import sys

import pytest

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


@pytest.mark.parametrize('world', [4, 8])
def test_gpus_4_and_8_spawn_that_many_ranks(world):
    """The world sizes of BASELINE configs[4] (4 x 32) and configs[2] (8 x 32) through the launcher: N ranks, N slices gathered."""
    r = _run(['--gpus', str(world), '--backend', 'gloo', '--steps', '1', '--warmup', '1', '--batch', '2'],
             {'ROHM_BENCH_SELFTEST': '1', 'OMP_NUM_THREADS': '1'})
    assert r.returncode == 0, r.stderr[-2000:]
    rec = _json_line(r.stdout)
    assert rec['n_gpus'] == world and rec['world_size'] == world and rec['ranks_seen'] == list(range(world))
    assert rec['gathered_clips'] == 2 * world
    assert rec['ms_per_step'] >= 10.0 * world - 1.0          # MAX over ranks: the last rank sleeps 10 ms x world per pass
    # round 6: the first multi-GPU run must be diagnosable from its one line -- every rank reports its OWN time per pass, the host
    # time to enqueue a pass, the all-gather drain, its wait for the slowest rank and its CPU binding; rank 0 adds the skew
    rep = rec['rank_report']
    assert [r['rank'] for r in rep['ranks']] == list(range(world))
    for r in rep['ranks']:
        assert {'own_ms_per_step', 'host_enqueue_ms_per_step', 'wait_for_slowest_ms', 'allgather_drain_ms_per_step', 'cpu_binding',
                'device'} <= set(r)
        assert r['own_ms_per_step'] >= 10.0 * (r['rank'] + 1) - 1.0          # rank r sleeps 10 ms x (r + 1) per pass
    sk = rep['skew']
    assert sk['slowest_rank'] == world - 1 and sk['max_own_ms_per_step'] >= sk['min_own_ms_per_step'] and 0.0 < sk['skew_frac'] < 1.0
    assert rep['ranks'][0]['wait_for_slowest_ms'] >= rep['ranks'][world - 1]['wait_for_slowest_ms'] - 2.0      # the fast rank waits longest
    assert rep['ranks'][0]['allgather_drain_ms_per_step'] >= rep['ranks'][world - 1]['allgather_drain_ms_per_step'] - 2.0


@pytest.mark.parametrize('workload,world,cfg', [('scheme', 8, 'configs[2]'), ('egobody', 4, 'configs[4]')])
def test_dry_run_of_the_sharded_baseline_configs(workload, world, cfg):
    """`--workload scheme --gpus 8 --batch 32` (BASELINE.json configs[2]: batch = 256 sharded 8 x) and `--workload egobody --gpus 4
    --batch 32` (configs[4]: 128 sharded 4 x) through the launcher over gloo: the record names the configuration the measuring path
    would name (same helper), every rank contributed its slice, every rank bound itself to its own CPUs."""
    r = _run(['--gpus', str(world), '--backend', 'gloo', '--workload', workload, '--batch', '32', '--steps', '1', '--warmup', '0',
              '--guidance-semantics', 'global'], {'ROHM_BENCH_SELFTEST': '1', 'OMP_NUM_THREADS': '1'})
    assert r.returncode == 0, r.stderr[-2000:]
    rec = _json_line(r.stdout)
    assert rec['n_gpus'] == world and rec['ranks_seen'] == list(range(world)) and rec['gathered_clips'] == 32 * world
    c = rec['config']
    assert cfg in c['stands_for_workload'] and 'batch=32 clips per GPU' in c['stands_for_workload'] and c['clips_per_gpu'] == 32
    assert f'{world} x 32 independent clips' in c['sharding'] and f'the reference at batch {32 * world}' in c['sharding']
    assert c['cpu_binding'].startswith('rank bound to') or c['cpu_binding'].startswith('not bound')


def test_rank_cpu_plan_splits_numa_nodes_between_the_ranks_that_share_them():
    """per-rank CPU / NUMA placement.  An 8-GPU host with two NUMA nodes of 64 cores (GPUs 0-3 on node 0, 4-7 on
    node 1): every rank gets 16 cores of ITS node; unknown topology: an even split of the mask; a cgroup that leaves fewer cores
    than ranks: ranks share, nobody gets an empty mask; a node whose cores are all outside the mask falls back to the even split."""
    sys.path.insert(0, ROOT)
    import bench
    assert bench.parse_cpulist('0-3,8,10-11\n') == [0, 1, 2, 3, 8, 10, 11]
    allowed = list(range(128))
    node_of = lambda r: 0 if r < 4 else 1
    cpus_of = lambda n: list(range(64 * n, 64 * n + 64))
    plans = [bench.plan_rank_cpus(r, 8, allowed, node_of, cpus_of)[0] for r in range(8)]
    assert all(len(p) == 16 for p in plans) and sorted(sum(plans, [])) == allowed
    assert all(set(plans[r]) <= set(cpus_of(node_of(r))) for r in range(8))
    even = [bench.plan_rank_cpus(r, 8, allowed)[0] for r in range(8)]
    assert all(len(p) == 16 for p in even) and sorted(sum(even, [])) == allowed
    tight = [bench.plan_rank_cpus(r, 8, [3, 5, 9])[0] for r in range(8)]
    assert all(len(p) == 1 and p[0] in (3, 5, 9) for p in tight)
    outside, how = bench.plan_rank_cpus(5, 8, list(range(16)), node_of, cpus_of)      # node 1's cores are not in the mask
    assert outside == [10, 11] and 'even split' in how
    lone, how = bench.plan_rank_cpus(0, 1, allowed, lambda r: 1, cpus_of)
    assert lone == cpus_of(1) and 'NUMA node 1' in how


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


@pytest.mark.gpu
def test_force_dist_runs_the_rccl_plumbing_on_one_gpu():
    """the multi-GPU path's process-group code (nccl = RCCL init with device_id, barrier, MAX all-reduce of
    the timing, all-gather of the results, one JSON line from rank 0) executed on real hardware at world size 1, launched the
    way the driver launches N > 1 (torch.distributed.run, 127.0.0.1 rendezvous)."""
    import socket
    with socket.socket() as so:
        so.bind(('127.0.0.1', 0))
        port = so.getsockname()[1]
    env = dict(os.environ)
    env.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
    for k in ('RANK', 'LOCAL_RANK', 'WORLD_SIZE', 'MASTER_ADDR', 'MASTER_PORT'):
        env.pop(k, None)
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node=1', '--master-addr', '127.0.0.1',
           '--master-port', str(port), os.path.join(ROOT, 'bench.py'), '--gpus', '1', '--force-dist', '--steps', '1', '--warmup', '1',
           '--batch', '8', '--ddpm-steps', '20', '--no-cpu-baseline', '--no-extras']
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert r.returncode == 0, r.stdout[-1500:] + r.stderr[-1500:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith('{')]
    assert len(lines) == 1
    d = json.loads(lines[0])
    assert d['n_gpus'] == 1 and d['process_group'] == {'backend': 'nccl', 'world_size': 1, 'forced': True}
    assert d['value'] > 0 and d['roofline']['frac'] > 0
    # the per-rank CPU / NUMA binding ran on real sysfs (PCI address of the GPU -> numa_node -> that node's cores, or the even split)
    assert d['config']['cpu_binding'].startswith('rank bound to'), d['config']['cpu_binding']
    # and started bare (no launcher): the script provides its own rendezvous
    r = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--force-dist', '--steps', '1', '--warmup', '0', '--batch', '4',
                        '--ddpm-steps', '10', '--no-cpu-baseline', '--no-extras'], env=env, capture_output=True, text=True,
                       timeout=600, cwd=ROOT)
    assert r.returncode == 0, r.stdout[-1500:] + r.stderr[-1500:]
    assert json.loads([ln for ln in r.stdout.splitlines() if ln.startswith('{')][-1])['process_group']['backend'] == 'nccl'


@pytest.mark.gpu
def test_scheme_workload_under_force_dist_with_global_batch_guidance():
    """BASELINE configs[2]'s workload (`--workload scheme`: TrajNet -> PoseNet + skating guidance -> TrajControl -> PoseNet) through
    the RCCL process group at world size 1, with `--guidance-semantics global`: the mask-count all-reduce of every guided step and
    the result all-gather run over RCCL on the MI355X; at one rank the global batch IS the local batch, so the result must equal the
    replica-semantics run bit for bit."""
    env = dict(os.environ)
    env.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
    for k in ('RANK', 'LOCAL_RANK', 'WORLD_SIZE', 'MASTER_ADDR', 'MASTER_PORT'):
        env.pop(k, None)
    recs = {}
    for sem in ('global', 'replica'):
        r = subprocess.run([sys.executable, BENCH, '--force-dist', '--workload', 'scheme', '--guidance-semantics', sem, '--steps', '1',
                            '--warmup', '0', '--batch', '32', '--no-cpu-baseline', '--no-extras'], env=env,      # configs[2]'s per-GPU slice
                           capture_output=True, text=True, timeout=900, cwd=ROOT)
        assert r.returncode == 0, r.stdout[-1500:] + r.stderr[-1500:]
        recs[sem] = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith('{')][-1])
    assert 'GLOBAL-batch semantics' in recs['global']['config']['sharding']
    assert 'replica semantics' in recs['replica']['config']['sharding']
    assert recs['global']['config']['finite_output'] and recs['global']['value'] > 0
    assert recs['global']['config']['output_digest'] == recs['replica']['config']['output_digest']


def test_force_dist_at_world_size_1_goes_through_the_process_group():
    """CPU twin of the GPU test above: with --force-dist a single rank still initialises the group and runs the barrier, the MAX
    all-reduce and the result all-gather through it (gloo here, RCCL on the GPU)."""
    r = _run(['--gpus', '1', '--backend', 'gloo', '--force-dist', '--steps', '2', '--warmup', '1', '--batch', '3'],
             {'ROHM_BENCH_SELFTEST': '1'}, drop=('WORLD_SIZE', 'RANK', 'LOCAL_RANK', 'MASTER_PORT', 'MASTER_ADDR'))
    assert r.returncode == 0, r.stderr[-2000:]
    rec = _json_line(r.stdout)
    assert rec['n_gpus'] == 1 and rec['gathered_clips'] == 3 and rec['ranks_seen'] == [0]


def test_extras_attach_second_line_and_configs_and_survive_a_failing_child(monkeypatch):
    """The N = 1 record's `second_line` / `configs` come from child processes; a child that fails must leave an {'error': ...}
    entry, never cost the headline line.  (Pure host logic: the children are stubbed.)"""
    import importlib.util
    import types
    spec = importlib.util.spec_from_file_location('bench_mod', BENCH)
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    calls = []

    def fake_child(argv, env_extra=None, timeout=600):
        calls.append((tuple(argv), dict(env_extra or {})))
        if 'egobody' in argv:
            return {'error': 'rc=1', 'argv': list(argv)}
        mode = (env_extra or {}).get('ROHM_GEMM_PRECISION', 'fp32')
        return {'metric': 'm', 'value': 30.0 if mode != 'fp32' else 18.0, 'unit': 'clips/s', 'ms_per_step': 1.0, 'steps': 1, 'warmup': 1,
                'dtype': mode, 'config': {'workload': ' '.join(argv)},
                'roofline': {'achieved': 1.0, 'peak': 2.0, 'frac': 0.5, 'unit': 'TFLOP/s', 'kernel': 'k', 'bound': 'mfma'},
                'accuracy': {'max_abs_vs_reference': 5e-6}, 'child_wall_s': 1.0}
    monkeypatch.setattr(bench, 'run_child', fake_child)
    second, cfg = bench.extras(types.SimpleNamespace(ddpm_steps=1000, batch=64, no_cpu_baseline=False))
    assert second['mode'] == 'fp16x3' and second['value'] == 30.0 and second['accuracy']['max_abs_vs_reference'] == 5e-6
    assert second['roofline']['frac'] == 0.5 and 'NEVER the headline' in second['label']
    assert second['also']['bf16x6']['value'] == 30.0
    assert set(cfg) == {'b32', 'scheme_b32', 'prox_b32', 'egobody_b32'}
    assert cfg['b32']['value'] == 18.0 and cfg['egobody_b32']['error'] == 'rc=1'
    # the precision variable reaches only the second-line children; every child runs without the extras and the CPU leg
    assert [c[1].get('ROHM_GEMM_PRECISION') for c in calls] == ['fp16x3', 'bf16x6', None, None, None, None]
    # each config child is asked for its own CPU baseline (every config carries one) unless the run has none at all
    assert all('--config-cpu-baseline' in c[0] for c in calls[2:]) and not any('--config-cpu-baseline' in c[0] for c in calls[:2])
    calls.clear()
    bench.extras(types.SimpleNamespace(ddpm_steps=1000, batch=64, no_cpu_baseline=True))
    assert not any('--config-cpu-baseline' in c[0] for c in calls)
