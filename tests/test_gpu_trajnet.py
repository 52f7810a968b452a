"""GPU parity of the TrajNet / TrajControl path (through the C ABI) vs the reference's golden outputs and the
CPU oracle.  Single forwards 2e-4 (51-71 stacked fp32 convolutions); full 100-step sampling 1e-3."""
import pytest
import torch

from helpers import cpu_noise_sequence, golden, max_abs, seeded
from oracle import diffusion as odiff
from oracle import nets
from rohm_amd.utils import synth

pytestmark = pytest.mark.gpu
DEV = 'cuda:0'


class Args:
    noise_schedule, sigma_small = 'cosine', True


def make_trajnet(seed, ctrl):
    from rohm_amd.model.trajnet import TrajNet
    net = TrajNet(time_dim=32, mid_dim=512, cond_dim=13, traj_feat_dim=13, trajcontrol=ctrl, device=DEV)
    sd = synth.trajnet_state_dict(seed, trajcontrol=ctrl)
    net.load_state_dict(sd, strict=True)
    return net.to(DEV).eval(), sd


def make_diffusion(steps=100):
    from rohm_amd.diffusion import gaussian_diffusion_trajnet as gdt
    from rohm_amd.diffusion.respace import SpacedDiffusionTrajNet
    from rohm_amd.utils.model_util import create_gaussian_diffusion
    return create_gaussian_diffusion(Args, gdt, SpacedDiffusionTrajNet, steps, '', device=DEV)


@pytest.mark.parametrize('name', ['trajnet_forward.npz', 'trajnet_control_forward.npz'])
def test_forward_vs_reference_golden(name):
    g = golden(name)
    ctrl = 'control' in name
    net, _ = make_trajnet(int(g['weight_seed']), ctrl)
    batch = {'x_t': seeded(int(g['x_seed']), 2, 144, 13).to(DEV), 'cond': seeded(int(g['cond_seed']), 2, 144, 13).to(DEV),
             'control_cond': seeded(int(g['control_seed']), 2, 144, 272).to(DEV)}
    y = net(batch, torch.from_numpy(g['t']).to(DEV)).cpu()
    assert max_abs(y, torch.from_numpy(g['y'])) < 2e-4


@pytest.mark.parametrize('B,T,ctrl', [(1, 144, False), (3, 144, True), (2, 48, True), (5, 16, False), (32, 144, True),
                                      (72, 144, True),      # B > 64: the un-fused conv forms (compute-bound batches)
                                      (2, 288, True), (1, 400, False)])   # long clips: GroupNorm groups of 2304 / 3200 values
def test_forward_vs_oracle(B, T, ctrl):
    net, sd = make_trajnet(40 + B, ctrl)
    x, c, cc = seeded(1, B, T, 13), seeded(2, B, T, 13), seeded(3, B, T, 272)
    t = torch.tensor([(37 * i + 3) % 100 for i in range(B)])
    ref = nets.trajnet_forward(sd, x, c, t, control_cond=cc if ctrl else None, dtype=torch.float64)
    y = net({'x_t': x.to(DEV), 'cond': c.to(DEV), 'control_cond': cc.to(DEV)}, t.to(DEV)).cpu()
    assert max_abs(y, ref) < 2e-4


def test_zero_control_is_identity_on_gpu():
    """with zero-initialised control convs TrajControl reproduces the vanilla net exactly."""
    from rohm_amd.model.trajnet import TrajNet
    sd_c = synth.trajnet_state_dict(5, trajcontrol=True, zero_convs_random=False)
    sd_v = {k: v for k, v in sd_c.items() if not k.startswith('controlnet.')}
    nc = TrajNet(time_dim=32, mid_dim=512, cond_dim=13, traj_feat_dim=13, trajcontrol=True).to(DEV)
    nv = TrajNet(time_dim=32, mid_dim=512, cond_dim=13, traj_feat_dim=13, trajcontrol=False).to(DEV)
    nc.load_state_dict(sd_c)
    nv.load_state_dict(sd_v)
    batch = {'x_t': seeded(1, 2, 144, 13).to(DEV), 'cond': seeded(2, 2, 144, 13).to(DEV),
             'control_cond': seeded(3, 2, 144, 272).to(DEV)}
    t = torch.tensor([42, 7], device=DEV)
    assert torch.equal(nc(batch, t), nv(batch, t))


def test_loop100_vs_reference_golden():
    """BASELINE.json configs[0]: TrajNet vanilla, one clip, the full 100-step loop; noise = the reference's CPU stream."""
    g = golden('trajnet_loop100.npz')
    net, _ = make_trajnet(int(g['weight_seed']), False)
    cond = seeded(int(g['cond_seed']), 1, 144, 13)
    x_T, noises = cpu_noise_sequence(int(g['torch_seed']), (1, 144, 13), 100, trajnet_layout=True)
    for fused in (True, False):
        diff = make_diffusion()
        diff.noise_source = lambda step, like: (x_T if step == -1 else noises[step])
        diff.fused_chunk = 33
        batch = {'cond': cond.to(DEV)}
        if fused:
            _, y = diff.eval_losses(model=net, batch=batch, shape=[1, 144, 13], progress=False, clip_denoised=False,
                                    timestep_respacing='', cond_fn_with_grad=True, compute_loss=False)
        else:
            y = list(diff.p_sample_loop_progressive(net, batch, [1, 144, 13]))[-1]['sample']
        assert max_abs(y.cpu(), torch.from_numpy(g['y'])) < 1e-3, fused


def test_control_loop100_vs_reference_golden():
    """TrajControl (ControlNet residuals on) through the full 100-step loop against the REFERENCE's own sampler
    (tests/golden/trajnet_control_loop100.npz): the stage every inference iteration >= 1 of configs 3-5 runs."""
    g = golden('trajnet_control_loop100.npz')
    net, _ = make_trajnet(int(g['weight_seed']), True)
    cond, cc = seeded(int(g['cond_seed']), 2, 144, 13), seeded(int(g['control_seed']), 2, 144, 272)
    x_T, noises = cpu_noise_sequence(int(g['torch_seed']), (2, 144, 13), 100, trajnet_layout=True)
    diff = make_diffusion()
    diff.noise_source = lambda step, like: (x_T if step == -1 else noises[step])
    _, y = diff.eval_losses(model=net, batch={'cond': cond.to(DEV), 'control_cond': cc.to(DEV)}, shape=[2, 144, 13],
                            progress=False, clip_denoised=False, timestep_respacing='', cond_fn_with_grad=True,
                            compute_loss=False)
    assert max_abs(y.cpu(), torch.from_numpy(g['y'])) < 1e-3


def test_control_loop_vs_oracle():
    net, sd = make_trajnet(61, True)
    B = 2
    cond, cc = seeded(5, B, 144, 13), seeded(6, B, 144, 272)
    x_T, noises = cpu_noise_sequence(8, (B, 144, 13), 100)
    diff = make_diffusion()
    diff.noise_source = lambda step, like: (x_T if step == -1 else noises[step])
    _, y = diff.eval_losses(model=net, batch={'cond': cond.to(DEV), 'control_cond': cc.to(DEV)}, shape=[B, 144, 13],
                            progress=False, clip_denoised=False, timestep_respacing='', cond_fn_with_grad=True,
                            compute_loss=False)
    fn = lambda x, i: nets.trajnet_forward(sd, x, cond, torch.full((B,), i, dtype=torch.int64), control_cond=cc)
    ref = odiff.p_sample_loop(fn, x_T, noises, odiff.tables(odiff.cosine_betas(100)), list(range(100))[::-1])
    assert max_abs(y.cpu(), ref) < 1e-3


def test_graph_replay_loop_is_bit_identical(monkeypatch):
    """The opt-in hipGraph replay of the sampling loop (ROHM_TRAJNET_GRAPH=1: one captured step, per-step values read
    from device tables through a step counter) must reproduce the plain loop bit for bit."""
    monkeypatch.setenv('ROHM_TRAJ_RESIDENT', '0')      # both legs are forms of the launch-per-layer loop (the clip-resident step is not recorded)
    net, _ = make_trajnet(81, True)
    B = 2
    cond, cc = seeded(5, B, 144, 13), seeded(6, B, 144, 272)
    x_T, noises = cpu_noise_sequence(8, (B, 144, 13), 100)
    outs = []
    for flag in ('0', '1'):
        monkeypatch.setenv('ROHM_TRAJNET_GRAPH', flag)
        diff = make_diffusion()
        diff.noise_source = lambda step, like: (x_T if step == -1 else noises[step])
        _, y = diff.eval_losses(model=net, batch={'cond': cond.to(DEV), 'control_cond': cc.to(DEV)}, shape=[B, 144, 13],
                                progress=False, clip_denoised=False, timestep_respacing='', cond_fn_with_grad=False,
                                compute_loss=False)
        outs.append(y.clone())
    assert torch.equal(outs[0], outs[1])


def test_control_side_stream_is_bit_identical(monkeypatch):
    """The ControlNet branch of the TrajControl loop runs on a second stream, one step ahead of the U-Net (two sets of residual
    buffers, events): ROHM_TRAJ_CTRL_STREAM=0 (everything on the caller's stream, in the reference's order) must give the same bits,
    run after run (a missing event would show up as a race)."""
    monkeypatch.setenv('ROHM_TRAJ_RESIDENT', '0')      # the side stream belongs to the launch-per-layer loop
    net, _ = make_trajnet(82, True)
    B = 3
    cond, cc = seeded(15, B, 144, 13), seeded(16, B, 144, 272)
    x_T, noises = cpu_noise_sequence(18, (B, 144, 13), 100)
    outs = []
    for flag in ('1', '0', '1', '1'):
        monkeypatch.setenv('ROHM_TRAJ_CTRL_STREAM', flag)
        diff = make_diffusion()
        diff.noise_source = lambda step, like: (x_T if step == -1 else noises[step])
        _, y = diff.eval_losses(model=net, batch={'cond': cond.to(DEV), 'control_cond': cc.to(DEV)}, shape=[B, 144, 13],
                                progress=False, clip_denoised=False, timestep_respacing='', cond_fn_with_grad=False,
                                compute_loss=False)
        outs.append(y.clone())
    assert all(torch.equal(outs[0], o) for o in outs[1:])
    assert torch.isfinite(outs[0]).all()


def test_shape_errors_are_raised_before_the_c_abi():
    """cond / control_cond of the wrong shape must raise (the C ABI takes raw pointers)."""
    from rohm_amd.model.trajnet import TrajNet
    net = TrajNet(time_dim=32, mid_dim=512, cond_dim=13, traj_feat_dim=13, trajcontrol=True, device=DEV)
    net.load_state_dict(synth.trajnet_state_dict(1, trajcontrol=True), strict=True)
    net = net.to(DEV).eval()
    x = torch.zeros(2, 144, 13, device=DEV)
    t = torch.zeros(2, dtype=torch.int64, device=DEV)
    with pytest.raises(ValueError):
        net({'x_t': x, 'cond': torch.zeros(2, 144, 22, device=DEV), 'control_cond': torch.zeros(2, 144, 272, device=DEV)}, t)
    with pytest.raises(ValueError):
        net({'x_t': x, 'cond': x, 'control_cond': torch.zeros(2, 143, 272, device=DEV)}, t)
    with pytest.raises(KeyError):
        net({'x_t': x, 'cond': x}, t)


def test_launch_shape_knobs_change_nothing_but_the_summation_order():
    """rohm_trajnet_tune (workgroups per CU, fewest K chunks per split-K slice, power-of-two split counts) only re-shapes
    the conv launches: every setting must agree with the default to fp32 rounding of a differently ordered K sum."""
    from rohm_amd import _lib
    net, _ = make_trajnet(7, True)
    batch = {'x_t': seeded(1, 3, 144, 13).to(DEV), 'cond': seeded(2, 3, 144, 13).to(DEV),
             'control_cond': seeded(3, 3, 144, 272).to(DEV)}
    t = torch.tensor([99, 42, 0], device=DEV)
    ref = net(batch, t).clone()
    try:
        for cfg in ((1, 4, 0), (1, 1, 1), (2, 2, 0), (2, 3, 1)):
            _lib.check(_lib.lib().rohm_trajnet_tune(*cfg), 'rohm_trajnet_tune')
            assert max_abs(net(batch, t), ref) < 2e-5, cfg
    finally:
        _lib.check(_lib.lib().rohm_trajnet_tune(1, 2, 0), 'rohm_trajnet_tune')
    assert torch.equal(net(batch, t), ref)          # back on the defaults: bit-identical again
    assert _lib.lib().rohm_trajnet_tune(3, 2, 0) != 0 and _lib.lib().rohm_trajnet_tune(1, 0, 0) != 0


def test_profile_detail_labels_carry_the_launch_shape():
    from rohm_amd import _lib
    net, _ = make_trajnet(7, False)
    batch = {'x_t': seeded(1, 2, 144, 13).to(DEV), 'cond': seeded(2, 2, 144, 13).to(DEV)}
    t = torch.tensor([5, 6], device=DEV)
    net(batch, t)
    _lib.check(_lib.lib().rohm_profile_detail(1), 'rohm_profile_detail')
    try:
        _lib.profile_start(1)
        net(batch, t)
        torch.cuda.synchronize()
        rows = _lib.profile_stop()
    finally:
        _lib.check(_lib.lib().rohm_profile_detail(0), 'rohm_profile_detail')
    convs = [k for k in rows if k.startswith('conv_gemm')]
    assert convs and all(' M' in k and ' K' in k and ' S' in k for k in convs), rows.keys()
    assert any(k.startswith('gn_mish C') for k in rows)
    _lib.profile_start(1)
    net(batch, t)
    torch.cuda.synchronize()
    plain = _lib.profile_stop()
    assert 'conv_gemm/64' in plain and 'gn_mish' in plain          # detail off: one row per kernel again


def test_large_batch_is_clip_independent():
    """B = 512 reaches the launch shapes small batches never do (144 x 128 conv tiles, no split-K, the separate residual /
    transposed-conv forms of compute-bound batches); clips are independent, so every 64-clip slice of the result must agree
    with a 64-clip forward of the same inputs to fp32 rounding of a differently ordered K sum."""
    net, _ = make_trajnet(9, True)
    B = 512
    x, c, cc = seeded(11, B, 144, 13).to(DEV), seeded(12, B, 144, 13).to(DEV), seeded(13, B, 144, 272).to(DEV)
    t = torch.arange(B, device=DEV) % 100
    y = net({'x_t': x, 'cond': c, 'control_cond': cc}, t)
    assert torch.isfinite(y).all()
    for lo in (0, 192, 448):
        sl = slice(lo, lo + 64)
        ys = net({'x_t': x[sl].contiguous(), 'cond': c[sl].contiguous(), 'control_cond': cc[sl].contiguous()}, t[sl].contiguous())
        assert max_abs(y[sl], ys) < 2e-5, lo


# ---- the clip-resident step (csrc/trajnet_resident.hip): one launch per denoising step; default for TrajNet, ROHM_TRAJ_RESIDENT=1 for TrajControl ----

def _loop(net, batch, shape, x_T, noises, fused_chunk=None):
    diff = make_diffusion()
    diff.noise_source = lambda step, like: (x_T if step == -1 else noises[step])
    if fused_chunk:
        diff.fused_chunk = fused_chunk
    _, y = diff.eval_losses(model=net, batch=batch, shape=list(shape), progress=False, clip_denoised=False, timestep_respacing='',
                            cond_fn_with_grad=False, compute_loss=False)
    return y


def test_resident_step_vs_reference_goldens(monkeypatch):
    """The reference's own 100-step runs (TrajNet, one clip: BASELINE.json configs[0]; TrajControl, two clips) through the
    clip-resident step: same bar as the launch-per-layer loop (1e-3), and the loop really ran in that form."""
    from rohm_amd import _lib
    monkeypatch.setenv('ROHM_TRAJ_RESIDENT', '1')
    g = golden('trajnet_loop100.npz')
    net, _ = make_trajnet(int(g['weight_seed']), False)
    cond = seeded(int(g['cond_seed']), 1, 144, 13)
    x_T, noises = cpu_noise_sequence(int(g['torch_seed']), (1, 144, 13), 100, trajnet_layout=True)
    y = _loop(net, {'cond': cond.to(DEV)}, (1, 144, 13), x_T, noises)
    assert _lib.lib().rohm_trajnet_loop_mode() == 1
    assert max_abs(y.cpu(), torch.from_numpy(g['y'])) < 1e-3
    g = golden('trajnet_control_loop100.npz')
    net, _ = make_trajnet(int(g['weight_seed']), True)
    cond, cc = seeded(int(g['cond_seed']), 2, 144, 13), seeded(int(g['control_seed']), 2, 144, 272)
    x_T, noises = cpu_noise_sequence(int(g['torch_seed']), (2, 144, 13), 100, trajnet_layout=True)
    y = _loop(net, {'cond': cond.to(DEV), 'control_cond': cc.to(DEV)}, (2, 144, 13), x_T, noises)
    assert _lib.lib().rohm_trajnet_loop_mode() == 1
    assert max_abs(y.cpu(), torch.from_numpy(g['y'])) < 1e-3


@pytest.mark.parametrize('B,ctrl', [(1, True), (8, False), (9, True), (13, False), (32, True), (64, False)])
def test_resident_step_vs_launch_per_layer(monkeypatch, B, ctrl):
    """Every way the clips fall onto the 8 XCDs -- one XCD busy, one clip each, a ragged last XCD (9 = 2 + 2 + 2 + 2 + 1, 13 = 6 x 2 + 1),
    4 and 8 clips per XCD (several clips per work item at the deep levels, GroupNorm statistics exchanged between 2 / 4 items) -- against
    the launch-per-layer loop on the same inputs and noise: fp32 summation order is all that differs (2e-5 after 100 steps), and the
    resident loop repeats itself bit for bit."""
    from rohm_amd import _lib
    net, _ = make_trajnet(90 + B, ctrl)
    cond, cc = seeded(5, B, 144, 13), seeded(6, B, 144, 272)
    x_T, noises = cpu_noise_sequence(8, (B, 144, 13), 100)
    batch = {'cond': cond.to(DEV), 'control_cond': cc.to(DEV)}
    monkeypatch.setenv('ROHM_TRAJ_RESIDENT', '0')
    ref = _loop(net, batch, (B, 144, 13), x_T, noises).clone()
    assert _lib.lib().rohm_trajnet_loop_mode() == 0
    monkeypatch.setenv('ROHM_TRAJ_RESIDENT', '1')
    y = _loop(net, batch, (B, 144, 13), x_T, noises, fused_chunk=37).clone()      # chunks of 37 + 37 + 26 steps: three calls
    assert _lib.lib().rohm_trajnet_loop_mode() == 1
    assert torch.isfinite(y).all() and max_abs(y, ref) < 2e-5
    assert torch.equal(y, _loop(net, batch, (B, 144, 13), x_T, noises, fused_chunk=37))


def test_resident_step_survives_a_missing_partner(monkeypatch):
    """Fault injection (ROHM_TRAJ_RESIDENT_FAULT=1: one workgroup stays away from a meeting): the partners' bounded waits expire, the
    error word is set, the host restores x_T and re-runs the loop launch per layer -- the caller gets that loop's exact result."""
    from rohm_amd import _lib
    net, _ = make_trajnet(97, True)
    B = 3
    cond, cc = seeded(5, B, 144, 13), seeded(6, B, 144, 272)
    x_T, noises = cpu_noise_sequence(8, (B, 144, 13), 100)
    batch = {'cond': cond.to(DEV), 'control_cond': cc.to(DEV)}
    monkeypatch.setenv('ROHM_TRAJ_RESIDENT', '0')
    ref = _loop(net, batch, (B, 144, 13), x_T, noises).clone()
    monkeypatch.setenv('ROHM_TRAJ_RESIDENT', '1')
    monkeypatch.setenv('ROHM_TRAJ_RESIDENT_FAULT', '1')
    y = _loop(net, batch, (B, 144, 13), x_T, noises)
    assert _lib.lib().rohm_trajnet_loop_mode() == 0
    assert torch.equal(y, ref)


def test_resident_step_is_the_default_for_trajnet_and_opt_in_for_trajcontrol(monkeypatch):
    """Without the environment variable the TrajNet loop runs clip-resident (mode 1), the TrajControl loop launch per layer (mode 0: its
    ControlNet branch hides on a second stream there); above 64 clips and under the launch profiler's per-shape detail mode the launch-per-layer
    loop serves both."""
    from rohm_amd import _lib
    monkeypatch.delenv('ROHM_TRAJ_RESIDENT', raising=False)
    for ctrl, B, want in ((False, 2, 1), (True, 2, 0), (False, 72, 0)):
        net, _ = make_trajnet(70, ctrl)
        cond, cc = seeded(5, B, 144, 13), seeded(6, B, 144, 272)
        x_T, noises = cpu_noise_sequence(8, (B, 144, 13), 100)
        y = _loop(net, {'cond': cond.to(DEV), 'control_cond': cc.to(DEV)}, (B, 144, 13), x_T, noises)
        assert _lib.lib().rohm_trajnet_loop_mode() == want, (ctrl, B)
        assert torch.isfinite(y).all()
    # the launch profiler brackets the step's one launch; only its per-shape detail mode needs the launch-per-layer loop
    net, _ = make_trajnet(70, False)
    cond = seeded(5, 2, 144, 13)
    x_T, noises = cpu_noise_sequence(8, (2, 144, 13), 100)
    _lib.profile_start(10)
    try:
        _loop(net, {'cond': cond.to(DEV)}, (2, 144, 13), x_T, noises)
        assert _lib.lib().rohm_trajnet_loop_mode() == 1
    finally:
        prof = _lib.profile_stop()
    assert prof['conv_gemm/resident_step']['launches'] == 10 and prof['conv_gemm/resident_step']['flops'] > 0
    _lib.check(_lib.lib().rohm_profile_detail(1), 'rohm_profile_detail')
    _lib.profile_start(10)
    try:
        _loop(net, {'cond': cond.to(DEV)}, (2, 144, 13), x_T, noises)
        assert _lib.lib().rohm_trajnet_loop_mode() == 0
    finally:
        _lib.profile_stop()
        _lib.check(_lib.lib().rohm_profile_detail(0), 'rohm_profile_detail')


@pytest.mark.parametrize('T', [16, 64, 128, 160])
def test_resident_step_at_other_clip_lengths(monkeypatch, T):
    """T is not tied to the drivers' 144 frames: the shortest the U-Net takes (16: one row at the deepest level), a power of two, and
    the longest the resident step stages (160) -- each against the launch-per-layer loop; longer clips fall back to that loop."""
    from rohm_amd import _lib
    net, _ = make_trajnet(33, False)
    B = 3
    cond = seeded(5, B, T, 13)
    x_T, noises = cpu_noise_sequence(8, (B, T, 13), 100)
    batch = {'cond': cond.to(DEV)}
    monkeypatch.setenv('ROHM_TRAJ_RESIDENT', '0')
    ref = _loop(net, batch, (B, T, 13), x_T, noises).clone()
    monkeypatch.delenv('ROHM_TRAJ_RESIDENT')
    y = _loop(net, batch, (B, T, 13), x_T, noises)
    assert _lib.lib().rohm_trajnet_loop_mode() == 1
    assert torch.isfinite(y).all() and max_abs(y, ref) < 2e-5


def test_resident_step_leaves_long_clips_to_the_launch_per_layer_loop(monkeypatch):
    from rohm_amd import _lib
    monkeypatch.delenv('ROHM_TRAJ_RESIDENT', raising=False)
    net, _ = make_trajnet(33, False)
    B, T = 2, 176
    x_T, noises = cpu_noise_sequence(8, (B, T, 13), 100)
    y = _loop(net, {'cond': seeded(5, B, T, 13).to(DEV)}, (B, T, 13), x_T, noises)
    assert _lib.lib().rohm_trajnet_loop_mode() == 0 and torch.isfinite(y).all()
