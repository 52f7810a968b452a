"""Clip-level data parallelism for the denoising loops.

Clips are independent in both networks (GroupNorm / attention / LayerNorm never cross the batch dimension),
so a node runs one process per GPU, every rank denoises a contiguous slice of the batch with its own replica of
the weights, and the only communication is ONE all-gather of the finished clips per sampling run
(`torch.distributed` backend "nccl" = RCCL over xGMI on MI355X; "gloo" in the CPU tests).  There is no
collective inside the networks.

Guidance semantics under sharding: the reference normalises the skating loss by a mask count summed over the
whole batch (model/posenet.py:231,243) and averages the 2-D loss over it (:309).  By default a sharded run has
REPLICA semantics -- each rank behaves exactly like the reference run with batch_size = its local slice --
which needs no communication.  `use_global_batch_guidance` switches a PoseNet to GLOBAL-batch semantics: the two
skating mask counts are all-reduced (8 bytes) per guided step and the 2-D term is scaled by B_local / B_global,
reproducing the reference at the full batch size (rohm_amd/guidance.py, tests/test_gpu_guidance.py).
"""
from __future__ import annotations

import torch
import torch.distributed as dist


def slice_bounds(n, world, rank):
    """Contiguous, near-equal split of n items: rank r gets [lo, hi)."""
    base, extra = divmod(n, world)
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


def shard_batch(batch, world=None, rank=None, batch_dim_keys=None):
    """Slice every tensor of `batch` whose leading dimension is the (global) batch size."""
    world = dist.get_world_size() if world is None else world
    rank = dist.get_rank() if rank is None else rank
    sizes = [v.shape[0] for v in batch.values() if torch.is_tensor(v) and v.dim() > 0]
    if not sizes:
        return dict(batch)
    n = max(set(sizes), key=sizes.count)
    lo, hi = slice_bounds(n, world, rank)
    out = {}
    for k, v in batch.items():
        take = torch.is_tensor(v) and v.dim() > 0 and v.shape[0] == n and \
            (batch_dim_keys is None or k in batch_dim_keys)
        out[k] = v[lo:hi] if take else v
    return out


class PendingGather:
    """An all-gather in flight (`gather_clips(..., async_op=True)`): the collective runs on the backend's own stream behind the work
    already queued on the caller's stream; `result()` makes the caller's stream wait for it and assembles the global batch.  Between
    the two the caller may queue the next sampling run -- the gather of run k overlaps the start of run k + 1 (or, at the end of a
    job, the tail of the slowest rank)."""

    def __init__(self, work, parts, counts, keep):
        self._work, self._parts, self._counts, self._keep = work, parts, counts, keep

    def result(self):
        if self._work is not None:
            self._work.wait()
            self._work = None
        return torch.cat([p[:c] for p, c in zip(self._parts, self._counts)], dim=0)


def gather_clips(local, n_total, group=None, force=False, async_op=False):
    """All-gather the per-rank results (possibly ragged along dim 0) back into the global batch order.  A single-rank group
    returns its input untouched unless `force` asks for the collective anyway (bench.py --force-dist: the RCCL path on a
    one-GPU box).  `async_op=True` returns a `PendingGather` instead of the tensor."""
    world = dist.get_world_size(group)
    if world == 1 and not force:
        return PendingGather(None, [local], [local.shape[0]], None) if async_op else local
    counts = [slice_bounds(n_total, world, r)[1] - slice_bounds(n_total, world, r)[0] for r in range(world)]
    width = max(counts)
    pad = local
    if local.shape[0] < width:
        pad = torch.cat([local, local.new_zeros((width - local.shape[0],) + tuple(local.shape[1:]))], dim=0)
    pad = pad.contiguous()
    parts = [torch.empty_like(pad) for _ in range(world)]
    work = dist.all_gather(parts, pad, group=group, async_op=async_op)
    if async_op:
        return PendingGather(work, parts, counts, pad)
    return torch.cat([p[:c] for p, c in zip(parts, counts)], dim=0)


def sharded_sample(sample_fn, batch, shape, group=None):
    """Run `sample_fn(local_batch, local_shape) -> x0 [B_local, ...]` on this rank's slice and gather.

    `sample_fn` is the single-GPU call, e.g.
        lambda b, s: diffusion.eval_losses(model=net, batch=b, shape=s, compute_loss=False, ...)[1]
    """
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    n = shape[0]
    lo, hi = slice_bounds(n, world, rank)
    local = shard_batch(batch, world, rank)
    out = sample_fn(local, [hi - lo] + list(shape[1:]))
    return gather_clips(out, n, group)


def use_global_batch_guidance(model, group=True, global_batch=None):
    """Make the test-time guidance of `model` (a PoseNet) reproduce the reference at the GLOBAL batch size when the
    clips are sharded over the ranks of `group` (default process group if True; None switches back to per-rank
    semantics).  Costs one 8-byte all-reduce per guided step (the two skating mask counts, model/posenet.py:231,243).
    The 2-D term needs the global batch size (its loss is a mean over the batch, :309): pass `global_batch` if the caller knows
    it (no collective at all); otherwise it is all-reduced once per sampling run, at the first guided step -- every rank enters
    that collective in every run, whatever its local batch size (ragged splits included)."""
    raw = getattr(model, 'model', model)
    raw.guidance_group = group
    raw.__dict__['_rohm_global_batch'] = {} if global_batch is None else {'fixed': float(global_batch)}
    return model
