"""DDPM ancestral sampling engine shared by the PoseNet and TrajNet diffusions.

Behavioural contract = RoHM's `diffusion/gaussian_diffusion_{posenet,trajnet}.py` restricted to what
is reachable at inference: x0-prediction, fixed-small variance, the
`p_sample[_with_grad]` update, the 999..0 (or 999..20 with `early_stop`) loop and `eval_losses`.
The implementation is not the reference's: schedule tables are built once in float64 on the host
(same formulas, gaussian_diffusion_posenet.py:114-173) and uploaded once as one fp32 [steps, 4]
device table; every per-step update is ONE HIP kernel that indexes the table on the device (no
per-step H2D uploads, no host syncs); and when the model is the native PoseNet and no guidance is
due, whole runs of steps execute inside `rohm_posenet_sample_loop` without returning to Python.
"""
from __future__ import annotations

import enum
import math

import numpy as np
import torch

from .. import ops


class ModelMeanType(enum.Enum):
    PREVIOUS_X = enum.auto()
    START_X = enum.auto()
    EPSILON = enum.auto()


class ModelVarType(enum.Enum):
    LEARNED = enum.auto()
    FIXED_SMALL = enum.auto()
    FIXED_LARGE = enum.auto()
    LEARNED_RANGE = enum.auto()


class LossType(enum.Enum):
    MSE = enum.auto()
    RESCALED_MSE = enum.auto()
    KL = enum.auto()
    RESCALED_KL = enum.auto()


def betas_for_alpha_bar(num_diffusion_timesteps, alpha_bar, max_beta=0.999):
    """beta_i = min(1 - abar((i+1)/n) / abar(i/n), max_beta) (gaussian_diffusion_posenet.py:41-58)."""
    n = num_diffusion_timesteps
    return np.array([min(1 - alpha_bar((i + 1) / n) / alpha_bar(i / n), max_beta) for i in range(n)])


def get_named_beta_schedule(schedule_name, num_diffusion_timesteps, scale_betas=1.):
    """'linear' / 'cosine' schedules in float64 (gaussian_diffusion_posenet.py:14-38)."""
    if schedule_name == 'linear':
        scale = scale_betas * 1000 / num_diffusion_timesteps
        return np.linspace(scale * 0.0001, scale * 0.02, num_diffusion_timesteps, dtype=np.float64)
    if schedule_name == 'cosine':
        return betas_for_alpha_bar(num_diffusion_timesteps,
                                   lambda t: math.cos((t + 0.008) / 1.008 * math.pi / 2) ** 2)
    raise NotImplementedError(f'unknown beta schedule: {schedule_name}')


# Guidance schedule hard-coded in the reference (gaussian_diffusion_posenet.py:461-477):
#   grad_type -> (t threshold, [(hook name, weight), ...]) applied in this order.
GUIDANCE = {
    'prox': (100, (('guide_2d_projection_with_smpl', 3e5), ('guide_skating_with_smpl', 1e5))),
    'amass': (50, (('guide_skating_with_smpl', 3e6),)),
}


class DDPMSampler:
    """Base of `GaussianDiffusionPoseNet` / `GaussianDiffusionTrajNet`."""

    supports_guidance = False

    def __init__(self, *, betas, model_mean_type, model_var_type, loss_type, rescale_timesteps=False,
                 dataset=None, device=''):
        if model_mean_type.name != 'START_X':
            raise NotImplementedError('only x0-prediction (ModelMeanType.START_X) is used by RoHM')
        self.model_mean_type, self.model_var_type, self.loss_type = model_mean_type, model_var_type, loss_type
        self.rescale_timesteps = rescale_timesteps
        self.dataset, self.device = dataset, device
        betas = np.array(betas, dtype=np.float64)
        if betas.ndim != 1 or not ((betas > 0).all() and (betas <= 1).all()):
            raise ValueError('betas must be a 1-D array in (0, 1]')
        self.betas = betas
        self.num_timesteps = int(betas.shape[0])
        alphas = 1.0 - betas
        ac = np.cumprod(alphas, axis=0)
        ac_prev = np.append(1.0, ac[:-1])
        self.alphas_cumprod, self.alphas_cumprod_prev = ac, ac_prev
        self.alphas_cumprod_next = np.append(ac[1:], 0.0)
        self.sqrt_alphas_cumprod = np.sqrt(ac)
        self.sqrt_one_minus_alphas_cumprod = np.sqrt(1.0 - ac)
        self.posterior_variance = betas * (1.0 - ac_prev) / (1.0 - ac)
        # index 0 borrows index 1: the posterior variance is 0 at t = 0 (gaussian_diffusion_posenet.py:155-160)
        self.posterior_log_variance_clipped = np.log(np.append(self.posterior_variance[1],
                                                               self.posterior_variance[1:]))
        self.posterior_mean_coef1 = betas * np.sqrt(ac_prev) / (1.0 - ac)
        self.posterior_mean_coef2 = (1.0 - ac_prev) * np.sqrt(alphas) / (1.0 - ac)
        self._tables_cache = {}
        self._map_cache = {}
        self.timestep_map = list(range(self.num_timesteps))
        self.original_num_steps = self.num_timesteps
        self.noise_source = None     # optional callable (loop_step, like_tensor) -> noise; tests inject here
        self.fused_chunk = 50        # loop steps per rohm_posenet_sample_loop call / noise chunk
        self.guided_poll = 10        # guided (step-wise) tail: steps between two polls of the in-kernel exchange status

    # ------------------------------------------------------------------ schedule tables
    def host_tables(self):
        """float32 [steps, 4] = coef1, coef2, variance, log_variance (cast once, like `.float()` at use)."""
        return np.stack([self.posterior_mean_coef1, self.posterior_mean_coef2, self.posterior_variance,
                         self.posterior_log_variance_clipped], axis=1).astype(np.float32)

    def device_tables(self, device):
        key = str(device)
        if key not in self._tables_cache:
            self._tables_cache[key] = torch.from_numpy(self.host_tables()).to(device).contiguous()
        return self._tables_cache[key]

    def _mapped(self, t):
        """Network-side timestep (`_WrappedModel`, respace.py:183-195).  The reference's eval_losses hands the raw
        module (`model.model`) to p_sample_loop and `SpacedDiffusion*.p_mean_variance` re-wraps it (respace.py:92-95),
        so the map IS applied on the sampling path there too; the fused loops apply the same map on the host."""
        if self.timestep_map == list(range(self.num_timesteps)) and not self.rescale_timesteps:
            return t
        key = str(t.device)
        if key not in self._map_cache:
            self._map_cache[key] = torch.tensor(self.timestep_map, device=t.device, dtype=torch.int64)
        new_t = self._map_cache[key][t]
        if self.rescale_timesteps:
            new_t = new_t.float() * (1000.0 / self.original_num_steps)
        return new_t

    def _noise(self, step, like):
        if self.noise_source is not None:
            return self.noise_source(step, like).to(device=like.device, dtype=like.dtype).contiguous()
        return torch.randn_like(like)

    # ------------------------------------------------------------------ single steps (API parity)
    def p_mean_variance(self, model, batch, x, t, clip_denoised=True, denoised_fn=None, model_kwargs=None):
        """Network call + posterior mean (gaussian_diffusion_posenet.py:236-280)."""
        raw = getattr(model, 'model', model)
        B = x.shape[0]
        assert t.shape == (B,)
        batch['x_t'] = x
        pred_xstart = raw(batch, self._mapped(t), **(model_kwargs or {}))
        if self._exchange_failed(raw):                     # as in _step: the forward is pure, run it again exchange-free
            pred_xstart = raw(batch, self._mapped(t), **(model_kwargs or {}))
            self._check_exchange(raw)
        tab = self.device_tables(x.device)
        mean = ops.ddpm_step_table(x.contiguous(), pred_xstart, None, tab, t.contiguous())
        shape = (B,) + (1,) * (x.dim() - 1)
        return {'mean': mean, 'variance': tab[t, 2].view(shape), 'log_variance': tab[t, 3].view(shape),
                'pred_xstart': pred_xstart}

    def _step(self, model, batch, x, t, step, grad_type=None, t_int=None, noise=None, poll=True):
        """One ancestral step: network -> (guidance) -> fused update kernel.  `poll=False`: the caller checks the in-kernel
        exchanges itself, every few steps, and re-runs from its checkpoint (the guided tail of `_fused_loop`); `noise`: pre-drawn."""
        raw = getattr(model, 'model', model)
        batch['x_t'] = x
        x0 = raw(batch, self._mapped(t))
        if poll and self._exchange_failed(raw):            # the forward is a pure function of (x, cond, t): run it again, exchange-free
            x0 = raw(batch, self._mapped(t))
            self._check_exchange(raw)
        if noise is None:
            noise = self._noise(step, x)                   # drawn BEFORE guidance (…posenet.py:458)
        grads = []
        if grad_type is not None:
            if not self.supports_guidance or grad_type not in GUIDANCE:
                raise ValueError(f'unknown grad_type {grad_type!r}')
            thr, hooks = GUIDANCE[grad_type]
            if t_int is None:
                t_int = int(t[0])
            if t_int <= thr:
                out = {'pred_xstart': x0}
                for name, w in hooks:
                    g = getattr(raw, name)(batch, out, t, compute_grad='x_0')
                    if g.dim() != 0:                        # 0-d zero = "no active constraint"
                        grads.append((g.float().contiguous(), w))
        ga, wa = grads[0] if len(grads) > 0 else (None, 0.0)
        gb, wb = grads[1] if len(grads) > 1 else (None, 0.0)
        sample = ops.ddpm_step_table(x.contiguous(), x0, noise, self.device_tables(x.device), t.contiguous(),
                                     ga, wa, gb, wb)
        return {'sample': sample, 'pred_xstart': x0, 'x_t': x}

    def p_sample(self, model, batch, x, t, clip_denoised=True, denoised_fn=None, cond_fn=None,
                 model_kwargs=None, const_noise=False):
        if cond_fn is not None or const_noise:
            raise NotImplementedError('cond_fn / const_noise are never used by the RoHM drivers')
        self._new_run(getattr(model, 'model', model))       # a directly driven step has no run boundary: nothing per-run is reused
        with torch.no_grad():
            return self._step(model, batch, x, t, step=None)

    def p_sample_with_grad(self, model, batch, x, t, clip_denoised=True, denoised_fn=None, cond_fn=None,
                           grad_type=None, model_kwargs=None, const_noise=False):
        # driven step by step (the reference API allows it) there is no run boundary to key the guidance's per-run cache on: drop it,
        # so the global batch size is all-reduced by EVERY rank at EVERY directly driven guided step (8 bytes) -- never stale, never
        # a collective that only some ranks enter
        self._new_run(getattr(model, 'model', model))
        with torch.no_grad():
            return self._step(model, batch, x, t, step=None, grad_type=grad_type)

    # ------------------------------------------------------------------ loops
    def _indices(self, skip_timesteps=0, early_stop=False):
        idx = list(range(self.num_timesteps - skip_timesteps))[::-1]
        return idx[0:980] if early_stop else idx          # hard-coded in the reference (…posenet.py:625-626)

    def p_sample_loop_progressive(self, model, batch, shape, noise=None, clip_denoised=True, denoised_fn=None,
                                  cond_fn=None, model_kwargs=None, device=None, progress=False, skip_timesteps=0,
                                  init_image=None, randomize_class=False, cond_fn_with_grad=False, grad_type=None,
                                  early_stop=False, const_noise=False):
        """Generator over per-step dicts, one kernel-chain per step (…posenet.py:578-662)."""
        if skip_timesteps or init_image is not None or const_noise or cond_fn is not None:
            raise NotImplementedError('skip_timesteps / init_image / const_noise / cond_fn are unused by RoHM')
        raw = getattr(model, 'model', model)
        # a sampling run starts here whoever drives the generator (p_sample_loop, or a caller of the public generator as in the
        # reference API): per-run guidance caches are dropped before step 0 so every rank enters the same collectives
        self._new_run(raw)
        if device is None:
            device = next(raw.parameters()).device
        img = noise if noise is not None else self._x_T(shape, device)
        indices = self._indices(0, early_stop)
        if progress:
            from tqdm.auto import tqdm
            indices = tqdm(indices)
        B = shape[0]
        for step, i in enumerate(indices):
            t = torch.full((B,), i, device=device, dtype=torch.int64)
            with torch.no_grad():
                out = self._step(model, batch, img, t, step,
                                 grad_type=grad_type if cond_fn_with_grad else None, t_int=i)
            yield out
            img = out['sample']

    def _x_T(self, shape, device):
        if self.noise_source is not None:
            like = torch.empty(*shape, device=device)
            return self.noise_source(-1, like).to(device=device, dtype=torch.float32).contiguous()
        return torch.randn(*shape, device=device)

    @staticmethod
    def _copy_into(buf, x):
        if buf is None:
            return x.clone()
        buf.copy_(x)
        return buf

    def _fused_ok(self, raw):
        return hasattr(raw, 'sample_loop_native') and not self.rescale_timesteps

    @staticmethod
    def _check_exchange(raw):
        """Ask the network whether one of its in-kernel exchanges failed since the last check (rohm_posenet_exchange_status: one
        stream synchronisation) and raise instead of returning wrong samples."""
        fn = getattr(raw, 'check_exchange', None)
        if fn is not None:
            fn()

    @staticmethod
    def _exchange_failed(raw):
        """After a fused chunk / a step-wise forward: True if an in-kernel exchange failed since the last check -- the network has
        then switched itself to its exchange-free launches (PoseNet.recover_exchange) and the caller re-runs what it just computed.
        Networks without such launches (TrajNet) answer False without touching the device."""
        fn = getattr(raw, 'recover_exchange', None)
        return bool(fn()) if fn is not None else False

    @staticmethod
    def _new_run(raw):
        """Start of a sampling run: per-run caches of the guidance (the all-reduced global batch size, rohm_amd/guidance.py) are
        dropped so that every rank enters the same collectives in every run, whatever its local batch sizes were before."""
        cache = getattr(raw, '__dict__', {}).get('_rohm_global_batch')
        if cache:
            for k in [k for k in cache if k != 'fixed']:
                del cache[k]

    def p_sample_loop(self, model, batch, shape, noise=None, clip_denoised=True, denoised_fn=None, cond_fn=None,
                      model_kwargs=None, device=None, progress=False, skip_timesteps=0, init_image=None,
                      randomize_class=False, cond_fn_with_grad=False, grad_type=None, early_stop=False,
                      dump_steps=None, const_noise=False, save_intermediate_result=False):
        """Full sampling run; returns the last `sample` (or last `pred_xstart` with `early_stop`)."""
        raw = getattr(model, 'model', model)
        self._new_run(raw)
        if dump_steps is not None or save_intermediate_result or not self._fused_ok(raw):
            final = None
            dump = []
            x0s, xts, tls = [], [], []
            every = max(1, self.num_timesteps // 5)          # `save_steps_all = 5` (gaussian_diffusion_posenet.py:532)
            k = -1
            for k, out in enumerate(self.p_sample_loop_progressive(
                    model, batch, shape, noise=noise, device=device, progress=progress,
                    cond_fn_with_grad=cond_fn_with_grad, grad_type=grad_type, early_stop=early_stop)):
                if dump_steps is not None and k in dump_steps:
                    dump.append(out['sample'].clone())
                final = out
                if save_intermediate_result and k % every == 0:                       # :556-560
                    x0s.append(out['pred_xstart'].clone())
                    xts.append(out['x_t'].clone())
                    tls.append(self.num_timesteps - k - 1)
            if dump_steps is not None:
                self._check_exchange(raw)
                return dump
            if save_intermediate_result:
                # the reference appends the last step once more and returns the 4-tuple, early_stop or not (:570-574)
                x0s.append(final['pred_xstart'].clone())
                xts.append(final['x_t'].clone())
                tls.append(self.num_timesteps - k - 1)
                self._check_exchange(raw)
                return final['sample'], x0s, xts, tls
            self._check_exchange(raw)
            return final['pred_xstart'] if early_stop else final['sample']
        out = self._fused_loop(raw, batch, shape, noise, device, cond_fn_with_grad, grad_type, early_stop)
        self._check_exchange(raw)
        return out

    def _fused_loop(self, raw, batch, shape, noise, device, cond_fn_with_grad, grad_type, early_stop):
        """Device-resident runs of un-guided steps + per-step execution of the guided tail."""
        self._new_run(raw)
        if device is None:
            device = next(raw.parameters()).device
        x = (noise if noise is not None else self._x_T(shape, device)).to(torch.float32).contiguous().clone()
        cond = batch['cond'].detach().to(torch.float32).contiguous()
        indices = self._indices(0, early_stop)
        thr = -1
        if cond_fn_with_grad and grad_type is not None:
            if not self.supports_guidance or grad_type not in GUIDANCE:
                raise ValueError(f'unknown grad_type {grad_type!r}')
            thr = GUIDANCE[grad_type][0]
        n_free = sum(1 for i in indices if i > thr)        # leading un-guided steps
        tabs = self.host_tables()
        x0_last = None
        with torch.no_grad():
            pos = 0
            x_in_last = None
            x_save = None
            while pos < n_free:
                n = min(self.fused_chunk, n_free - pos)
                ts = indices[pos:pos + n]
                coef = np.empty((n, 3), np.float32)
                for k, i in enumerate(ts):
                    coef[k, 0], coef[k, 1] = tabs[i, 0], tabs[i, 1]
                    coef[k, 2] = np.exp(np.float32(0.5) * tabs[i, 3]) if i != 0 else 0.0
                if self.noise_source is not None:
                    nz = torch.stack([self._noise(pos + k, x) for k in range(n)])
                else:
                    nz = torch.randn((n,) + tuple(x.shape), device=x.device, dtype=torch.float32)
                last = (pos + n == len(indices))
                if last:      # the native loop copies the input of its last step out (one device copy, no extra loop call)
                    x_in_last = torch.empty_like(x)
                x_save = self._copy_into(x_save, x) if hasattr(raw, 'recover_exchange') else None
                x0 = raw.sample_loop_native(x, cond, [self.timestep_map[i] for i in ts], coef, nz,
                                            want_x0_last=last, batch=batch, x_in_last=x_in_last if last else None)
                if x_save is not None and self._exchange_failed(raw):
                    # an in-kernel exchange of this chunk failed: the network now runs its exchange-free launches; the chunk is
                    # repeated from its saved input with the same noise -- exact, not approximately right
                    x.copy_(x_save)
                    x0 = raw.sample_loop_native(x, cond, [self.timestep_map[i] for i in ts], coef, nz,
                                                want_x0_last=last, batch=batch, x_in_last=x_in_last if last else None)
                    self._check_exchange(raw)
                if last:
                    x0_last = x0
                pos += n
            B = shape[0]
            # Guided tail, step by step.  The in-kernel exchanges are polled (one stream synchronisation) every `guided_poll`
            # steps instead of after every forward: the steps since the last clean poll are kept re-runnable -- their input and
            # their noise -- so a failed exchange still costs a warning and a repeat, never wrong samples.
            exchanging = getattr(raw, 'uses_exchange', lambda: False)()
            step, ck_step, ck_x, ck_in, ck_x0, saved = n_free, n_free, x, x_in_last, x0_last, []
            while step < len(indices):
                i = indices[step]
                t = torch.full((B,), i, device=x.device, dtype=torch.int64)
                k = step - ck_step
                if k == len(saved):
                    saved.append(self._noise(step, x))
                x_in_last = x
                out = self._step(raw, batch, x, t, step, grad_type=grad_type, t_int=i, noise=saved[k], poll=not exchanging)
                x, x0_last = out['sample'], out['pred_xstart']
                step += 1
                if exchanging and (step - ck_step >= self.guided_poll or step == len(indices)):
                    if self._exchange_failed(raw):          # the handle now runs its exchange-free launches: repeat from the checkpoint
                        exchanging = False
                        step, x, x_in_last, x0_last = ck_step, ck_x, ck_in, ck_x0
                        continue
                    ck_step, ck_x, ck_in, ck_x0, saved = step, x, x_in_last, x0_last, []
        # the reference leaves the INPUT of the last executed step in batch['x_t'] (p_mean_variance, :264)
        batch['x_t'] = x_in_last if x_in_last is not None else x
        return x0_last if early_stop else x

    # ------------------------------------------------------------------ DDIM
    def ddim_coefficients(self, i, eta=0.0):
        """DDIM eq. 12 with the x0 parameterisation (gaussian_diffusion_posenet.py:693-712): with
        eps = (sqrt(1/ab) x_t - x0) / sqrt(1/ab - 1) the update x_{t-1} = sqrt(ab_prev) x0 + sqrt(1 - ab_prev - s^2) eps
        + s z  is affine in (x0, x_t):  returns (c1, c2, s) with x_{t-1} = c1 x0 + c2 x_t + s z  (s = 0 at i = 0)."""
        # The reference evaluates sigma and sqrt(1 - ab_prev - sigma^2) in float32 on table entries cast with .float()
        # (`_extract_into_tensor`); near t = 0 the subtraction cancels and float32 rounding moves the result by ~1e-3, so
        # the same float32 steps are taken here (bit-level agreement with the reference body, tests/golden/ddim.npz).
        f = np.float32
        ab, ab_prev = f(self.alphas_cumprod[i]), f(self.alphas_cumprod_prev[i])
        r, m = f(np.sqrt(1.0 / self.alphas_cumprod[i])), f(np.sqrt(1.0 / self.alphas_cumprod[i] - 1))
        one = f(1.0)
        sigma = f(eta) * np.sqrt((one - ab_prev) / (one - ab)) * np.sqrt(one - ab / ab_prev)
        c = np.sqrt(np.maximum(one - ab_prev - sigma * sigma, f(0.0)))
        c1 = float(np.sqrt(ab_prev)) - float(c) / float(m)
        c2 = float(c) * float(r) / float(m)
        return f(c1), f(c2), f(sigma if i != 0 else 0.0)

    def ddim_sample(self, model, batch, x, t, clip_denoised=True, denoised_fn=None, cond_fn=None, model_kwargs=None,
                    eta=0.0):
        """One DDIM step.  (The reference's `ddim_sample` cannot run -- it calls p_mean_variance without `batch`,
        gaussian_diffusion_posenet.py:681-688 -- and no driver reaches it; the update formula is pinned to the body
        of that function, tests/golden/ddim.npz.)"""
        if cond_fn is not None:
            raise NotImplementedError('cond_fn is unused by RoHM')
        raw = getattr(model, 'model', model)
        t_int = int(t[0])
        if not bool((t == t_int).all()):
            raise NotImplementedError('ddim_sample expects one timestep for the whole batch, as the loops produce')
        with torch.no_grad():
            batch['x_t'] = x
            x0 = raw(batch, self._mapped(t))
            noise = self._noise(None, x)
            c1, c2, sg = self.ddim_coefficients(t_int, eta)
            sample = ops.ddpm_step(x.contiguous(), x0, noise, float(c1), float(c2), float(sg))
        return {'sample': sample, 'pred_xstart': x0}

    def ddim_sample_loop(self, model, batch, shape, noise=None, clip_denoised=True, denoised_fn=None, cond_fn=None,
                         model_kwargs=None, device=None, progress=False, eta=0.0, skip_timesteps=0, init_image=None,
                         randomize_class=False, cond_fn_with_grad=False, dump_steps=None, const_noise=False):
        """DDIM sampling run (gaussian_diffusion_posenet.py:775-822), device-resident: the affine DDIM update has the
        shape of the DDPM one, so the fused `rohm_*_sample_loop` kernels run it with `ddim_coefficients`.  No
        guidance (the reference's DDIM path never had any that worked)."""
        if cond_fn_with_grad or cond_fn is not None or skip_timesteps or init_image is not None or dump_steps:
            raise NotImplementedError('guidance / skip_timesteps / init_image / dump_steps are not supported with DDIM')
        raw = getattr(model, 'model', model)
        if device is None:
            device = next(raw.parameters()).device
        x = (noise if noise is not None else self._x_T(shape, device)).to(torch.float32).contiguous().clone()
        cond = batch['cond'].detach().to(torch.float32).contiguous()
        indices = self._indices(0, False)
        with torch.no_grad():
            pos = 0
            x_save = None
            while pos < len(indices):
                n = min(self.fused_chunk, len(indices) - pos)
                ts = indices[pos:pos + n]
                coef = np.asarray([self.ddim_coefficients(i, eta) for i in ts], dtype=np.float32)
                if eta == 0.0:
                    nz = None
                elif self.noise_source is not None:
                    nz = torch.stack([self._noise(pos + k, x) for k in range(n)])
                else:
                    nz = torch.randn((n,) + tuple(x.shape), device=x.device, dtype=torch.float32)
                x_save = self._copy_into(x_save, x) if hasattr(raw, 'recover_exchange') else None
                raw.sample_loop_native(x, cond, [self.timestep_map[i] for i in ts], coef, nz, want_x0_last=False,
                                       batch=batch)
                if x_save is not None and self._exchange_failed(raw):      # see _fused_loop
                    x.copy_(x_save)
                    raw.sample_loop_native(x, cond, [self.timestep_map[i] for i in ts], coef, nz, want_x0_last=False,
                                           batch=batch)
                    self._check_exchange(raw)
                pos += n
        batch['x_t'] = x
        self._check_exchange(raw)
        return x

    # ------------------------------------------------------------------ entry point
    def _eval(self, model, batch, shape, progress, clip_denoised, cond_fn_with_grad, grad_type, early_stop,
              timestep_respacing, compute_loss, smplx_model=None, epoch=0):
        loss_dict, out = None, self._sample(model, batch, shape, progress, clip_denoised, cond_fn_with_grad, grad_type,
                                            early_stop, timestep_respacing)[1]
        if compute_loss:       # the evaluation report of test_posenet.py / test_trajnet.py (…posenet.py:957-958)
            raw = getattr(model, 'model', model)
            with torch.no_grad():
                loss_dict = (raw.compute_losses_with_smpl(batch, out, smplx_model, epoch) if self.supports_guidance
                             else raw.compute_losses_with_smpl(batch, out, smplx_model))
        return loss_dict, out

    def _sample(self, model, batch, shape, progress, clip_denoised, cond_fn_with_grad, grad_type, early_stop,
                timestep_respacing):
        if timestep_respacing[0:4] == 'ddim':
            # the branch the reference left commented out (gaussian_diffusion_posenet.py:949-952): eta = 0 DDIM over
            # the (already respaced) schedule of this object
            if cond_fn_with_grad and grad_type is not None and self.supports_guidance:
                raise NotImplementedError('test-time guidance is not defined for DDIM sampling in RoHM')
            return None, self.ddim_sample_loop(model=getattr(model, 'model', model), batch=batch, shape=shape,
                                               progress=progress, clip_denoised=clip_denoised, eta=0.0)
        if timestep_respacing != '':
            raise NotImplementedError("timestep_respacing must be '' (ancestral sampling) or 'ddimN'")
        out = self.p_sample_loop(model=getattr(model, 'model', model), batch=batch, shape=shape, progress=progress,
                                 clip_denoised=clip_denoised, cond_fn_with_grad=cond_fn_with_grad,
                                 grad_type=grad_type, early_stop=early_stop)
        return None, out

    def training_losses(self, *a, **k):
        raise NotImplementedError('training is outside the inference hot path (SURVEY.md §8)')


def _extract_into_tensor(arr, timesteps, broadcast_shape):
    """Schedule lookup helper kept for API parity (gaussian_diffusion_posenet.py:967-980)."""
    res = torch.from_numpy(np.asarray(arr)).to(device=timesteps.device)[timesteps].float()
    while res.dim() < len(broadcast_shape):
        res = res[..., None]
    return res.expand(broadcast_shape)
