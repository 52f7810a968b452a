"""CPU: the DDIM update.  The reference's `ddim_sample` cannot be called as written (missing
`batch` argument, gaussian_diffusion_posenet.py:681-688); its body was executed with a stubbed p_mean_variance by
oracle/make_golden.py (tests/golden/ddim.npz).  Checked here: the oracle restatement of that body, and the host
coefficients the HIP loop consumes (x_prev = c1 x0 + c2 x_t + s z)."""
import numpy as np
import torch

from helpers import golden, max_abs, seeded
from oracle import diffusion as odiff


def _cases():
    g = golden('ddim.npz')
    x, x0 = seeded(int(g['x_seed']), 2, 16, 1, 9), seeded(int(g['x0_seed']), 2, 16, 1, 9)
    for k in range(int(g['n_cases'])):
        torch.manual_seed(int(g[f'case{k}_seed']))
        noise = torch.randn_like(x)
        yield int(g[f'case{k}_i']), float(g[f'case{k}_eta']), x, x0, noise, torch.from_numpy(g[f'case{k}'])


def test_oracle_ddim_step_matches_reference_body():
    tab = odiff.tables(odiff.cosine_betas(1000), spaced=False)
    for i, eta, x, x0, noise, ref in _cases():
        out = odiff.ddim_step(x, x0, noise, tab, i, eta)
        assert max_abs(out, ref) <= 2e-6 * max(1.0, float(ref.abs().max())), (i, eta)


def test_host_ddim_coefficients_match_reference_body():
    from rohm_amd.diffusion import gaussian_diffusion_posenet as gdp
    d = gdp.GaussianDiffusionPoseNet(betas=gdp.get_named_beta_schedule('cosine', 1000),
                                     model_mean_type=gdp.ModelMeanType.START_X,
                                     model_var_type=gdp.ModelVarType.FIXED_SMALL, loss_type=gdp.LossType.MSE)
    for i, eta, x, x0, noise, ref in _cases():
        c1, c2, s = d.ddim_coefficients(i, eta)
        out = float(c1) * x0 + float(c2) * x + float(s) * noise
        assert max_abs(out, ref) <= 5e-6 * max(1.0, float(ref.abs().max())), (i, eta)
    c1, c2, s = d.ddim_coefficients(0, 1.0)
    assert s == 0.0 and abs(float(c1) - 1.0) < 1e-6 and abs(float(c2)) < 1e-6       # last step returns x0
