"""CPU, world_size 2 over gloo: the N > 1 path (slice -> local sampling -> all-gather) reassembles exactly what a
single process computes.  The per-rank "sampler" here is a deterministic stand-in supplied by the test (the
product code has no CPU sampler); what is under test is the sharding / gather logic the GPU ranks run."""
import os
import socket

import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from rohm_amd import sharding


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _fake_sampler(batch, shape):
    # clip-wise (no cross-batch coupling), like the real denoiser; element-wise only, so that a clip's result does not depend on
    # how many clips share its tensor (a torch CPU reduction picks its summation order by shape)
    c = batch['cond']
    return c * 2.0 + c[:, 0:1, 0:1, 0:1] - 0.5 * c[:, 7:8, 0:1, 100:101]


def _worker(rank, world, port, n_clips, q):
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        g = torch.Generator().manual_seed(0)
        cond = torch.randn(n_clips, 294, 1, 143, generator=g)
        batch = {'cond': cond, 'scalar': torch.tensor(3.0), 'meta': 'x'}
        out = sharding.sharded_sample(_fake_sampler, batch, [n_clips, 294, 1, 143])
        ref = _fake_sampler(batch, None)
        lo, hi = sharding.slice_bounds(n_clips, world, rank)
        q.put((rank, bool(torch.equal(out, ref)), hi - lo))
    finally:
        dist.destroy_process_group()


def _run(n_clips, world=2):
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, n_clips, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    return sorted(res)


def test_even_split_world2():
    res = _run(4)
    assert [r[1] for r in res] == [True, True] and [r[2] for r in res] == [2, 2]


def test_ragged_split_world2():
    res = _run(5)            # 3 + 2 clips: gather must pad and trim
    assert [r[1] for r in res] == [True, True] and [r[2] for r in res] == [3, 2]


def test_ragged_splits_world4_and_world8():
    """The splits of BASELINE configs[2] / [4] (8 x 32 and 4 x 32 clips) and ragged ones at the same world sizes: every rank's
    slice is sampled locally and the all-gather reassembles the single-process result."""
    res = _run(10, world=4)      # 3 + 3 + 2 + 2
    assert [r[1] for r in res] == [True] * 4 and [r[2] for r in res] == [3, 3, 2, 2]
    res = _run(13, world=8)      # 2 x 5 + 1 x 3
    assert [r[1] for r in res] == [True] * 8 and [r[2] for r in res] == [2] * 5 + [1] * 3
    res = _run(16, world=8)
    assert [r[1] for r in res] == [True] * 8 and [r[2] for r in res] == [2] * 8


def test_global_batch_is_reduced_once_per_run_or_given_by_the_caller():
    """The global batch size is constant over a run: ONE all-reduce at the first guided step of every run -- entered by every rank
    whatever its local batch size was in earlier runs (a cache keyed by the local batch size let one rank of a ragged split
    skip a collective its peer entered) -- or none when the caller states it; never one collective + host sync per guided step."""
    from rohm_amd import guidance
    from rohm_amd.diffusion.ddpm import DDPMSampler

    class M:
        pass
    calls = []

    def group(t):
        calls.append(int(t.numel()))
        return t.mul_(4)                      # four ranks with the same local batch
    m = sharding.use_global_batch_guidance(M(), group)
    for _ in range(5):
        assert guidance.global_batch(8, group, 'cpu', m) == 32.0
    assert calls == [1]
    DDPMSampler._new_run(m)                                                            # the next sampling run starts: reduce again
    assert guidance.global_batch(7, group, 'cpu', m) == 28.0 and calls == [1, 1]
    assert guidance.global_batch(7, group, 'cpu', m) == 28.0 and calls == [1, 1]
    sharding.use_global_batch_guidance(m, group)                                       # re-arming forgets the cache
    assert guidance.global_batch(8, group, 'cpu', m) == 32.0 and calls == [1, 1, 1]
    m = sharding.use_global_batch_guidance(M(), group, global_batch=256)
    DDPMSampler._new_run(m)                                                            # ... but never what the caller stated
    assert guidance.global_batch(32, group, 'cpu', m) == 256.0 and calls == [1, 1, 1]


def _worker_ragged_runs(rank, world, port, q):
    """Two sampling runs with different ragged splits -- (3, 2) then (3, 3): rank 0's local batch size repeats, rank 1's does not.
    Every rank must enter exactly one size-1 all-reduce per run, followed by the per-step mask-count all-reduce (size 2)."""
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        from rohm_amd import guidance
        from rohm_amd.diffusion.ddpm import DDPMSampler

        class M:
            pass
        m = sharding.use_global_batch_guidance(M())
        totals = []
        for split in ((3, 2), (3, 3)):
            DDPMSampler._new_run(m)
            for _step in range(3):                                   # guided steps of the run
                totals.append(guidance.global_batch(split[rank], True, 'cpu', m))
                counts = torch.tensor([1.0 + rank, 2.0])
                guidance._allreduce_sum(counts, True)                # a mis-paired collective (size 1 vs size 2) would raise or hang here
                assert counts.tolist() == [3.0, 4.0]
        q.put((rank, totals))
    finally:
        dist.destroy_process_group()


def test_global_batch_over_runs_with_changing_ragged_splits_world2():
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker_ragged_runs, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in procs)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for _, totals in res:
        assert totals == [5.0] * 3 + [6.0] * 3


def test_slice_bounds_cover_everything():
    for n in (1, 7, 64, 256):
        for w in (1, 2, 3, 8):
            spans = [sharding.slice_bounds(n, w, r) for r in range(w)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))


def _worker_guidance(rank, world, port, q):
    """The host side of global-batch guidance: the all-reduce helpers of rohm_amd.guidance over a real process
    group (gloo stands in for RCCL): mask counts are summed, the global batch size of ragged shards is recovered."""
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        from rohm_amd import guidance
        counts = torch.tensor([3.0 + rank, 10.0 * (rank + 1)])
        guidance._allreduce_sum(counts, True)
        b_local = 3 if rank == 0 else 2
        total = guidance.global_batch(b_local, dist.group.WORLD, 'cpu')

        class M:
            pass
        m = sharding.use_global_batch_guidance(M())
        q.put((rank, counts.tolist(), total, m.guidance_group is True))
    finally:
        dist.destroy_process_group()


def test_global_batch_guidance_allreduce_world2():
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker_guidance, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in procs)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for _, counts, total, flag in res:
        assert counts == [7.0, 30.0] and total == 5.0 and flag
