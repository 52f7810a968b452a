"""CPU: host-side logic of the drop-in (schedules, respacing, state_dict contract, ABI surface)."""
import ctypes
import os
import re

import numpy as np
import pytest
import torch

from helpers import PoseDataset
from oracle import diffusion as odiff
from rohm_amd import _lib
from rohm_amd.diffusion import gaussian_diffusion_posenet as gdp
from rohm_amd.diffusion import gaussian_diffusion_trajnet as gdt
from rohm_amd.diffusion.respace import SpacedDiffusionPoseNet, SpacedDiffusionTrajNet, space_timesteps
from rohm_amd.utils import synth
from rohm_amd.utils.model_util import create_gaussian_diffusion

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


class Args:
    noise_schedule = 'cosine'
    sigma_small = True


def test_tables_match_oracle():
    for steps, gd, cls in ((1000, gdp, SpacedDiffusionPoseNet), (100, gdt, SpacedDiffusionTrajNet)):
        d = create_gaussian_diffusion(Args, gd, cls, steps, '', device='cpu')
        tab = odiff.tables(odiff.cosine_betas(steps))
        assert np.array_equal(d.posterior_mean_coef1, tab['coef1'])
        assert np.array_equal(d.posterior_mean_coef2, tab['coef2'])
        assert np.array_equal(d.posterior_variance, tab['variance'])
        assert np.array_equal(d.posterior_log_variance_clipped, tab['log_variance'])
        assert d.timestep_map == list(range(steps))
        assert d.host_tables().shape == (steps, 4) and d.host_tables().dtype == np.float32


def test_space_timesteps_docstring_example():
    # respace.py:16-18: 300 steps, [10, 15, 20]
    s = space_timesteps(300, [10, 15, 20])
    assert len(s) == 45 and len([i for i in s if i < 100]) == 10 and len([i for i in s if i >= 200]) == 20
    assert space_timesteps(1000, 'ddim50') == set(range(0, 1000, 20))
    with pytest.raises(ValueError):
        space_timesteps(10, [20])


def test_respaced_betas_reproduce_alphas():
    d = create_gaussian_diffusion(Args, gdp, SpacedDiffusionPoseNet, 1000, '100', device='cpu')
    base = create_gaussian_diffusion(Args, gdp, SpacedDiffusionPoseNet, 1000, '', device='cpu')
    assert d.num_timesteps == 100 and len(d.timestep_map) == 100
    assert np.allclose(d.alphas_cumprod, base.alphas_cumprod[d.timestep_map], rtol=1e-12)


def test_posenet_state_dict_contract():
    from rohm_amd.model.posenet import PoseNet
    net = PoseNet(PoseDataset(), 294, latent_dim=512, ff_size=1024, num_layers=8, num_heads=4, traj_feat_dim=22,
                  body_model_path=torch.nn.Identity(), device='cpu')
    sd = synth.posenet_state_dict(3)
    keys = [k for k in net.state_dict() if not k.startswith('smplx_model.')]
    assert sorted(keys) == sorted(sd.keys()) and len(keys) == 108
    net.load_state_dict(sd, strict=True)
    assert torch.equal(net.state_dict()['seqTransEncoder.layers.3.linear1.weight'],
                       sd['seqTransEncoder.layers.3.linear1.weight'])
    assert sum(p.numel() for n, p in net.named_parameters() if not n.startswith('smplx_model.')) == 17789200


def test_trajnet_state_dict_contract():
    from rohm_amd.model.trajnet import TrajNet, weight_order
    for ctrl, n_keys in ((False, 186), (True, 270)):
        net = TrajNet(time_dim=32, mid_dim=512, cond_dim=13, traj_feat_dim=13, trajcontrol=ctrl, device='cpu')
        sd = synth.trajnet_state_dict(4, trajcontrol=ctrl)
        assert list(net.state_dict().keys()) == list(sd.keys()) and len(sd) == n_keys
        net.load_state_dict(sd, strict=True)
        assert weight_order(512, 13, ctrl) == list(sd.keys())         # the order the C ABI consumes
    n_all = sum(p.numel() for p in net.parameters())
    n_ctrl = sum(p.numel() for n, p in net.named_parameters() if n.startswith('controlnet.'))
    assert n_ctrl == 15242557 and n_all - n_ctrl == 22582893
    # zero-initialised control convs (heads.py:12-18)
    fresh = TrajNet(time_dim=32, mid_dim=512, cond_dim=13, traj_feat_dim=13, trajcontrol=True)
    assert float(fresh.controlnet.control_zero_conv_mid.weight.abs().max()) == 0.0
    with pytest.raises(_lib.RohmHipError):
        fresh({'x_t': torch.zeros(1, 144, 13), 'cond': torch.zeros(1, 144, 13),
               'control_cond': torch.zeros(1, 144, 272)}, torch.zeros(1, dtype=torch.int64))


def test_no_cpu_fallback():
    """The product path must fail loudly on CPU tensors instead of silently computing elsewhere."""
    from rohm_amd.model.posenet import PoseNet
    net = PoseNet(PoseDataset(), 294, latent_dim=512, ff_size=1024, num_layers=8, num_heads=4, traj_feat_dim=22,
                  body_model_path=torch.nn.Identity(), device='cpu').eval()
    batch = {'x_t': torch.zeros(1, 294, 1, 143), 'cond': torch.zeros(1, 294, 1, 143)}
    with pytest.raises(_lib.RohmHipError):
        net(batch, torch.zeros(1, dtype=torch.int64))


def test_abi_exports_every_declared_symbol():
    """librohm_hip.so loads and exports exactly what include/rohm_hip.h declares (no compute calls)."""
    hdr = open(os.path.join(ROOT, 'include', 'rohm_hip.h')).read()
    declared = set(re.findall(r'\b(rohm_[a-z0-9_]+)\s*\(', hdr))
    assert declared, 'no declarations parsed'
    lib = ctypes.CDLL(_lib.LIB_PATH)
    for name in declared:
        assert hasattr(lib, name), f'{name} declared in rohm_hip.h but not exported'
    assert declared == set(_lib.SIGNATURES), (declared ^ set(_lib.SIGNATURES))
    lib.rohm_version.restype = ctypes.c_int
    assert lib.rohm_version() >= 100
