"""Timestep respacing (drop-in for RoHM's `diffusion/respace.py`).

`SpacedDiffusion*` keep a subset of the base process' timesteps, recompute the betas over the
kept steps and remember `timestep_map` (kept index -> original timestep) so the network is fed
original timesteps.  With `timestep_respacing=''` (the only setting that reaches a sampler in
RoHM) the map is the identity.
"""
from __future__ import annotations

import numpy as np

from .gaussian_diffusion_posenet import GaussianDiffusionPoseNet
from .gaussian_diffusion_trajnet import GaussianDiffusionTrajNet


def space_timesteps(num_timesteps, section_counts):
    """Pick timesteps section by section (respace.py:10-63).

    `section_counts`: list / comma string of per-section counts, or "ddimN" for the fixed DDIM
    stride.  Example: 300 steps, [10, 15, 20] -> 10 of the first 100, 15 of the next, 20 of the last.
    """
    if isinstance(section_counts, str):
        if section_counts.startswith('ddim'):
            want = int(section_counts[4:])
            for stride in range(1, num_timesteps):
                if len(range(0, num_timesteps, stride)) == want:
                    return set(range(0, num_timesteps, stride))
            raise ValueError(f'cannot create exactly {num_timesteps} steps with an integer stride')
        section_counts = [int(x) for x in section_counts.split(',')]
    n_sec = len(section_counts)
    base, extra = divmod(num_timesteps, n_sec)
    start, picked = 0, []
    for k, count in enumerate(section_counts):
        size = base + (1 if k < extra else 0)
        if size < count:
            raise ValueError(f'cannot divide section of {size} steps into {count}')
        stride = 1 if count <= 1 else (size - 1) / (count - 1)
        pos = 0.0                      # accumulated (not multiplied): rounding must match the reference
        for _ in range(count):
            picked.append(start + round(pos))
            pos += stride
        start += size
    return set(picked)


def _respaced_betas(base_cls, use_timesteps, kwargs):
    base = base_cls(**kwargs)
    keep = set(use_timesteps)
    last, betas, tmap = 1.0, [], []
    for i, ac in enumerate(base.alphas_cumprod):
        if i in keep:
            betas.append(1 - ac / last)
            last = ac
            tmap.append(i)
    return np.array(betas), tmap


class _WrappedModel:
    """Kept for API parity (respace.py:183-195): `.model` is the raw module."""

    def __init__(self, model, timestep_map, rescale_timesteps, original_num_steps):
        self.model, self.timestep_map = model, timestep_map
        self.rescale_timesteps, self.original_num_steps = rescale_timesteps, original_num_steps


class _SpacedMixin:
    def _init_spaced(self, base_cls, use_timesteps, kwargs):
        self.use_timesteps = set(use_timesteps)
        n_orig = len(kwargs['betas'])
        betas, tmap = _respaced_betas(base_cls, use_timesteps, kwargs)
        kwargs = dict(kwargs, betas=betas)
        base_cls.__init__(self, **kwargs)
        self.timestep_map = tmap
        self.original_num_steps = n_orig

    def _wrap_model(self, model):
        if isinstance(model, _WrappedModel):
            return model
        return _WrappedModel(model, self.timestep_map, self.rescale_timesteps, self.original_num_steps)

    def _scale_timesteps(self, t):
        return t


class SpacedDiffusionPoseNet(_SpacedMixin, GaussianDiffusionPoseNet):
    def __init__(self, use_timesteps, **kwargs):
        self._init_spaced(GaussianDiffusionPoseNet, use_timesteps, kwargs)


class SpacedDiffusionTrajNet(_SpacedMixin, GaussianDiffusionTrajNet):
    def __init__(self, use_timesteps, **kwargs):
        self._init_spaced(GaussianDiffusionTrajNet, use_timesteps, kwargs)
