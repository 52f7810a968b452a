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


def _stream_k_walk(B, T=143, D=512, c_out=272):
    """The segment walk of gemm_f32.hip's stream-K workgroups (csrc/gemm_f32.hip, `do { ... } while (SK && sk_hi > sk_lo)`) restated on
    the host from the plan the library reports: yields (block, tile, chunk_lo, chunk_hi, role, sources) in processing order."""
    import ctypes as C
    from rohm_amd import _lib
    u, t8 = C.c_int(), C.c_int()
    if not _lib.lib().rohm_output_process_plan(B, T, D, c_out, C.byref(u), C.byref(t8)):
        assert u.value == 0 and t8.value == 0
        return None
    u, t8, nk = u.value, t8.value, D // 32
    out = []
    for block in range(256):
        x, j = block % 8, block // 8
        lo, hi, role = j * u, (j + 1) * u, -1
        while hi > lo:
            t = (hi - 1) // nk
            c_lo, c_hi = max(lo - t * nk, 0), hi - t * nk
            role = 1 if c_hi < nk else (2 if c_lo > 0 else 0)
            src = []
            if role == 2:                              # the blocks in front that hold the tile's earlier chunks, nearest first
                jb = j - 1
                while jb >= 0 and (jb + 1) * u > t * nk:
                    src.append(jb * 8 + x)
                    jb -= 1
            out.append((block, x * t8 + t, c_lo, c_hi, role, src))
            hi = t * nk + c_lo
    return u, t8, nk, out


def test_stream_k_schedule_covers_every_unit_once_and_waits_only_downwards():
    """Invariants the stream-K output head relies on, for every batch size the plan accepts: each (tile, K chunk) unit
    is contracted exactly once; a workgroup produces at most one partial tile and does so BEFORE it owns one; an owner's sources are
    lower-numbered blocks of its own XCD (dispatched earlier: no wait for a workgroup that may not have started) and together with its
    own chunks they tile the whole K range in order; the tiling it replaces is only ever left when that saves at least four chunks."""
    taken = 0
    for B in list(range(1, 130)) + [160, 192, 256, 320, 512]:
        plan = _stream_k_walk(B)
        tiles = 2 * -(-(B * 144) // 64)
        if plan is None:
            continue
        taken += 1
        u, t8, nk, segs = plan
        assert t8 * 8 == tiles and u * 32 == t8 * nk and 2 * u >= nk and -(-tiles // 256) * nk - u >= 4, B
        seen = {}
        produced = {}                                           # block -> (tile, lo, hi) of the partial it leaves
        order = {}
        for block, tile, lo, hi, role, src in segs:
            for c in range(lo, hi):
                assert (tile, c) not in seen, (B, tile, c)
                seen[(tile, c)] = block
            order.setdefault(block, []).append(role)
            if role == 1:
                assert block not in produced, 'one slot per workgroup'
                produced[block] = (tile, lo, hi)
        assert len(seen) == tiles * nk
        for block, roles in order.items():                      # partial first, whole tiles, the owned (cut) tile last
            assert roles == sorted(roles, key=lambda r: {1: 0, 0: 1, 2: 2}[r]), (B, block, roles)
            assert roles.count(1) <= 1 and roles.count(2) <= 1
        for block, tile, lo, hi, role, src in segs:
            if role != 2:
                continue
            assert src and all(s < block and s % 8 == block % 8 for s in src), (B, block, src)
            assert len(src) <= 2                                 # tiles in at most three pieces
            pieces = sorted([produced[s] for s in src], key=lambda p: p[1])
            assert all(p[0] == tile for p in pieces)
            edges = [0] + [p[2] for p in pieces]
            assert [p[1] for p in pieces] == edges[:-1] and edges[-1] == lo and hi == nk, (B, block, pieces, lo)
    assert taken >= 6                                            # B = 32, 64, 96, 128, 192, 256, ...
    assert _stream_k_walk(64)[0] == 18 and _stream_k_walk(32)[0] == 9


class _StubNet(torch.nn.Module):
    """A network with the fused-loop interface of rohm_amd.model.posenet.PoseNet on CPU tensors: `sample_loop_native` applies a known
    affine update per step (so the expected result is computable), `recover_exchange` reports a failed exchange on demand."""

    def __init__(self, fail_on_calls=()):
        super().__init__()
        self.w = torch.nn.Parameter(torch.zeros(1))
        self.fail_on_calls, self.calls, self.fallback, self.inputs = set(fail_on_calls), 0, False, []

    def sample_loop_native(self, x, cond, t_model, coef, noise, want_x0_last=False, batch=None, x_in_last=None):
        self.calls += 1
        self.inputs.append(x.clone())
        poisoned = self.calls in self.fail_on_calls and not self.fallback
        for k in range(len(t_model)):
            if x_in_last is not None and k == len(t_model) - 1:
                x_in_last.copy_(x)
            x0 = 0.5 * x + cond
            x.copy_(float(coef[k][0]) * x0 + float(coef[k][1]) * x + float(coef[k][2]) * noise[k])
            if poisoned:
                x.add_(1000.0)                     # what a failed in-kernel exchange leaves behind: garbage
        self._pending = poisoned
        return (0.5 * self.inputs[-1] + cond) if want_x0_last else None

    def recover_exchange(self):
        failed, self._pending = getattr(self, '_pending', False), False
        if failed:
            self.fallback = True
        return failed


def test_fused_loop_reruns_a_chunk_whose_exchange_failed():
    """rohm_amd/diffusion/ddpm.py::_fused_loop: after every fused chunk the network is asked whether an in-kernel exchange failed
    (PoseNet.recover_exchange: the handle has then switched to its exchange-free launches); the chunk is repeated from its saved input
    with the SAME noise, so the run equals one in which nothing failed -- also when the failing chunk is the last one (x0 / x_in_last
    outputs) and when two different chunks fail."""
    steps, shape = 10, [2, 3, 1, 5]
    cond = torch.arange(30, dtype=torch.float32).view(shape) * 0.01

    def run(fail_on_calls, chunk):
        diff = create_gaussian_diffusion(Args, gdp, SpacedDiffusionPoseNet, steps, '', device='cpu')
        g = torch.Generator().manual_seed(3)
        x_T = torch.randn(*shape, generator=g)
        noises = [torch.randn(*shape, generator=g) for _ in range(steps)]
        diff.noise_source = lambda step, like: (x_T if step == -1 else noises[step])
        diff.fused_chunk = chunk
        net = _StubNet(fail_on_calls)
        batch = {'cond': cond}
        out = diff.p_sample_loop(net, batch, shape, device='cpu')
        return out, batch['x_t'], net
    clean, xin_clean, net0 = run((), 4)
    assert net0.calls == 3                                    # chunks of 4, 4, 2 steps
    for fails in ((1,), (3,), (2, 4)):                        # call numbers count re-runs: (2, 4) = second chunk, then the last chunk
        out, xin, net = run(fails, 4)
        assert net.fallback and net.calls == 3 + 1            # a handle falls back once, then nothing fails any more
        assert torch.equal(out, clean) and torch.equal(xin, xin_clean), fails
        # the repeated call started from the very input of the failed one
        k = min(fails)
        assert torch.equal(net.inputs[k - 1], net.inputs[k])


def test_per_config_cpu_baseline_prices_a_pass_by_the_references_step_counts():
    """bench.py `configs.*.cpu_baseline`: every KIND of step of a multi-stage workload is timed with the oracle
    port and the pass is priced as sum(count x step time) with the reference's own step counts -- TrajNet 100, PoseNet 1000 (980 with
    early_stop), guidance on t <= 50 ('amass') / t <= 100 ('prox') (gaussian_diffusion_posenet.py:461-477,625-626)."""
    import sys
    sys.path.insert(0, ROOT)
    import bench
    rec = bench.cpu_baseline_config('scheme', 2, budget_s=4.0)
    assert rec['kind'] == 'port' and rec['unit'] == 'clips/s' and rec['value'] > 0
    assert rec['step_counts'] == {'trajnet_step': 100, 'trajcontrol_step': 100, 'posenet_step': 2 * 949, 'posenet_guided_step': 2 * 51}
    sec = sum(rec['step_ms'][k] * n for k, n in rec['step_counts'].items()) / 1e3
    assert abs(sec - rec['seconds_per_pass']) < 0.1 * sec + 0.1 and abs(rec['value'] - 2 / sec) < 0.05 * rec['value']
    ego = bench.cpu_baseline_config('egobody', 2, budget_s=4.0)
    assert ego['step_counts'] == {'trajnet_step': 100, 'trajcontrol_step': 200, 'posenet_step': 3 * 899, 'posenet_guided_step': 3 * 81}
    prox = bench.cpu_baseline_config('prox', 2, budget_s=3.0)
    assert prox['step_counts'] == {'posenet_step': 899, 'posenet_guided_step': 81}
