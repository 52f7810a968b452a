"""CPU restatement of the reference's DDPM ancestral sampler (test oracle).

Follows `diffusion/gaussian_diffusion_posenet.py`: schedule tables :114-173 (float64), posterior mean
:212-234, `p_sample[_with_grad]` :388-480, loop :578-662.  Noise is always INJECTED (a list of tensors:
x_T first, then one per step) so the oracle and the HIP path consume identical randomness.
"""
from __future__ import annotations

import math

import numpy as np
import torch


def cosine_betas(n, max_beta=0.999):
    ab = lambda t: math.cos((t + 0.008) / 1.008 * math.pi / 2) ** 2        # :32-36
    return np.array([min(1 - ab((i + 1) / n) / ab(i / n), max_beta) for i in range(n)], dtype=np.float64)


def respaced_betas(betas, use_timesteps=None):
    """`SpacedDiffusion*.__init__` (diffusion/respace.py:75-90): betas re-derived from the cumulative
    alphas of the kept timesteps.  The drivers always go through this, even when every step is kept."""
    keep = set(range(len(betas))) if use_timesteps is None else set(use_timesteps)
    ac = np.cumprod(1.0 - np.asarray(betas, dtype=np.float64))
    last, out = 1.0, []
    for i, a in enumerate(ac):
        if i in keep:
            out.append(1 - a / last)
            last = a
    return np.array(out)


def tables(betas, spaced=True):
    """Posterior tables in float64 (:132-168); `spaced` applies the identity respacing first."""
    betas = np.asarray(betas, dtype=np.float64)
    if spaced:
        betas = respaced_betas(betas)
    alphas = 1.0 - betas
    ac = np.cumprod(alphas)
    ac_prev = np.append(1.0, ac[:-1])
    var = betas * (1.0 - ac_prev) / (1.0 - ac)
    return {
        'alphas_cumprod': ac, 'alphas_cumprod_prev': ac_prev,
        'variance': var,
        'log_variance': np.log(np.append(var[1], var[1:])),
        'coef1': betas * np.sqrt(ac_prev) / (1.0 - ac),
        'coef2': (1.0 - ac_prev) * np.sqrt(alphas) / (1.0 - ac),
    }


def _f(arr, i, dtype):
    # the reference indexes the float64 table then casts with .float() (:977)
    return torch.tensor(arr[i], dtype=torch.float64).to(dtype)


GUIDANCE = {'prox': (100, (('2d', 3e5), ('skating', 1e5))), 'amass': (50, (('skating', 3e6),))}


def p_sample_loop(model_fn, x_T, step_noise, tab, indices, guidance=None, grad_type=None, dtype=torch.float32,
                  early_stop=False, return_all=False):
    """model_fn(x, t_int) -> pred_xstart.  `guidance` maps 'skating'/'2d' -> fn(x0, t_int) -> grad or None.
    Returns the final sample (or last pred_xstart if early_stop)."""
    x = x_T.to(dtype)
    x0 = None
    trace = []
    for step, i in enumerate(indices):
        x0 = model_fn(x, i)
        mean = _f(tab['coef1'], i, dtype) * x0 + _f(tab['coef2'], i, dtype) * x          # :219-222
        noise = step_noise[step].to(dtype)                                                  # drawn before guidance (:458)
        if grad_type is not None:
            thr, hooks = GUIDANCE[grad_type]
            if i <= thr:                                                                    # :464,469,475
                var = _f(tab['variance'], i, dtype)
                for name, w in hooks:
                    g = guidance[name](x0, i)
                    if g is not None:
                        mean = mean + w * var * g
        nonzero = 0.0 if i == 0 else 1.0                                                    # :430-433
        x = mean + nonzero * torch.exp(0.5 * _f(tab['log_variance'], i, dtype)) * noise     # :479
        if return_all:
            trace.append((x.clone(), x0.clone()))
    if return_all:
        return trace
    return x0 if early_stop else x


def ddim_step(x, pred_xstart, noise, tab, i, eta=0.0, dtype=torch.float32):
    """The body of `ddim_sample` (gaussian_diffusion_posenet.py:693-712): eps re-derived from the x0 prediction
    (:299-303), DDIM eq. 12, float32 arithmetic on table entries cast with .float() like `_extract_into_tensor`."""
    ab, ab_prev = _f(tab['alphas_cumprod'], i, dtype), _f(tab['alphas_cumprod_prev'], i, dtype)
    r = _f(np.sqrt(1.0 / tab['alphas_cumprod']), i, dtype)
    m = _f(np.sqrt(1.0 / tab['alphas_cumprod'] - 1), i, dtype)
    eps = (r * x - pred_xstart) / m
    sigma = eta * torch.sqrt((1 - ab_prev) / (1 - ab)) * torch.sqrt(1 - ab / ab_prev)
    mean_pred = pred_xstart * torch.sqrt(ab_prev) + torch.sqrt(1 - ab_prev - sigma ** 2) * eps
    nonzero = 0.0 if i == 0 else 1.0
    return mean_pred + nonzero * sigma * noise


def ddim_sample_loop(model_fn, x_T, step_noise, tab, indices, eta=0.0, dtype=torch.float32):
    """`ddim_sample_loop` (:775-822) with injected noise (step_noise may be None when eta == 0)."""
    x = x_T.to(dtype)
    for step, i in enumerate(indices):
        x0 = model_fn(x, i)
        nz = step_noise[step].to(dtype) if step_noise is not None else torch.zeros_like(x)
        x = ddim_step(x, x0, nz, tab, i, eta, dtype)
    return x
