"""GPU parity of the PoseNet path: HIP (through the C ABI) vs the CPU oracle and vs the reference's own
outputs (tests/golden).  Tolerance: 1e-3 is the bar stated by BASELINE.json's north_star for outputs
after a full sampling run; single forwards are held to 1e-4."""
import os

import numpy as np
import pytest
import torch

from helpers import PoseDataset, cpu_noise_sequence, golden, max_abs, seeded
from oracle import diffusion as odiff
from oracle import nets
from rohm_amd.utils import synth

pytestmark = pytest.mark.gpu

DEV = 'cuda:0'


class Args:
    noise_schedule = 'cosine'
    sigma_small = True


def make_posenet(seed):
    from rohm_amd.model.posenet import PoseNet
    net = PoseNet(PoseDataset(), 294, latent_dim=512, ff_size=1024, num_layers=8, num_heads=4, traj_feat_dim=22,
                  body_model_path=torch.nn.Identity(), device=DEV)
    sd = synth.posenet_state_dict(seed)
    net.load_state_dict(sd, strict=True)
    return net.to(DEV).eval(), sd


def make_diffusion(steps):
    from rohm_amd.diffusion import gaussian_diffusion_posenet as gdp
    from rohm_amd.diffusion.respace import SpacedDiffusionPoseNet
    from rohm_amd.utils.model_util import create_gaussian_diffusion
    return create_gaussian_diffusion(Args, gdp, SpacedDiffusionPoseNet, steps, '', device=DEV)


@pytest.mark.parametrize('ln_fold', ['0', '1'])
def test_forward_vs_reference_golden(ln_fold, monkeypatch):
    """ln_fold = '1': the opt-in path with LayerNorm folded into the surrounding GEMMs (csrc/posenet.hip) must meet
    the same bar as the default path with the separate LayerNorm kernel."""
    monkeypatch.setenv('ROHM_POSENET_LNFOLD', ln_fold)
    g = golden('posenet_forward.npz')
    net, _ = make_posenet(int(g['weight_seed']))
    x, c = seeded(int(g['x_seed']), 2, 294, 1, 143), seeded(int(g['cond_seed']), 2, 294, 1, 143)
    y = net({'x_t': x.to(DEV), 'cond': c.to(DEV)}, torch.from_numpy(g['t']).to(DEV)).cpu()
    # 1e-4 for the default and for bf16x6; the two-plane mode (bf16x3) is held to the north-star tolerance instead and its
    # ladder test says so by setting this variable (tests/test_gpu_precision_ladder.py)
    assert max_abs(y, torch.from_numpy(g['y'])) < float(os.environ.get('ROHM_TEST_FORWARD_BAR', '1e-4'))
    assert torch.equal(y[:, :22], c[:, :22])              # trajectory channels are a bit-exact copy


@pytest.mark.parametrize('B', [1, 5])
def test_forward_vs_oracle(B):
    net, sd = make_posenet(77)
    x, c = seeded(1, B, 294, 1, 143), seeded(2, B, 294, 1, 143)
    t = torch.tensor([(131 * i + 5) % 1000 for i in range(B)])
    ref64 = nets.posenet_forward(sd, x, c, t, dtype=torch.float64)
    # non-contiguous cond (the drivers pass a permuted view, test_amass_full.py:370)
    c_view = c[:, :, 0].permute(0, 2, 1).contiguous().permute(0, 2, 1).unsqueeze(2)
    y = net({'x_t': x.to(DEV), 'cond': c_view.to(DEV)}, t.to(DEV)).cpu()
    assert max_abs(y, ref64) < 1e-4


def test_loop8_vs_reference_golden_fused_and_stepwise():
    g = golden('posenet_loop8.npz')
    net, _ = make_posenet(int(g['weight_seed']))
    steps = int(g['steps'])
    cond = seeded(int(g['cond_seed']), 2, 294, 1, 143)
    x_T, noises = cpu_noise_sequence(int(g['torch_seed']), (2, 294, 1, 143), steps)
    src = lambda step, like: (x_T if step == -1 else noises[step])
    for fused in (True, False):
        diff = make_diffusion(steps)
        diff.noise_source = src
        diff.fused_chunk = 3                            # exercise chunk boundaries
        batch = {'cond': cond.to(DEV)}
        if fused:
            _, y = diff.eval_losses(model=net, batch=batch, shape=[2, 294, 1, 143], progress=False,
                                    clip_denoised=False, timestep_respacing='', cond_fn_with_grad=False,
                                    compute_loss=False)
        else:
            outs = list(diff.p_sample_loop_progressive(net, batch, [2, 294, 1, 143]))
            y = outs[-1]['sample']
        assert max_abs(y.cpu(), torch.from_numpy(g['y'])) < 1e-3, f'fused={fused}'


def test_loop_early_stop_returns_pred_xstart():
    net, sd = make_posenet(5)
    cond = seeded(9, 1, 294, 1, 143)
    x_T, noises = cpu_noise_sequence(7, (1, 294, 1, 143), 1000)
    diff = make_diffusion(1000)
    diff.noise_source = lambda step, like: (x_T if step == -1 else noises[step])
    # monkey-patch the index list to 6 steps starting at t=999 so the test stays short
    diff._indices = lambda skip=0, early_stop=False: [999, 998, 997, 996, 995, 994]
    _, y = diff.eval_losses(model=net, batch={'cond': cond.to(DEV)}, shape=[1, 294, 1, 143], progress=False,
                            clip_denoised=False, timestep_respacing='', cond_fn_with_grad=False,
                            compute_loss=False, early_stop=True)
    tab = odiff.tables(odiff.cosine_betas(1000))
    fn = lambda x, i: nets.posenet_forward(sd, x, cond, torch.full((1,), i, dtype=torch.int64))
    ref = odiff.p_sample_loop(fn, x_T, noises, tab, [999, 998, 997, 996, 995, 994], early_stop=True)
    assert max_abs(y.cpu(), ref) < 1e-3


@pytest.mark.parametrize('eta', [0.0, 0.5])
def test_ddim_loop_vs_oracle(eta):
    """DDIM sampling on a 'ddim10' respaced schedule: fused HIP loop with the DDIM coefficients
    against the oracle loop built on the reference's update body (tests/test_ddim_oracle.py pins it)."""
    from rohm_amd.diffusion import gaussian_diffusion_posenet as gdp
    from rohm_amd.diffusion.respace import SpacedDiffusionPoseNet, space_timesteps
    from rohm_amd.utils.model_util import create_gaussian_diffusion
    net, sd = make_posenet(21)
    B, n = 2, 10
    diff = create_gaussian_diffusion(Args, gdp, SpacedDiffusionPoseNet, 1000, f'ddim{n}', device=DEV)
    assert diff.num_timesteps == n
    cond = seeded(5, B, 294, 1, 143)
    x_T, noises = cpu_noise_sequence(77, (B, 294, 1, 143), n)
    diff.noise_source = lambda step, like: (x_T if step == -1 else noises[step])
    _, y = diff.eval_losses(model=net, batch={'cond': cond.to(DEV)}, shape=[B, 294, 1, 143], progress=False,
                            clip_denoised=False, timestep_respacing=f'ddim{n}', cond_fn_with_grad=False,
                            compute_loss=False)
    if eta != 0.0:
        y = diff.ddim_sample_loop(net, {'cond': cond.to(DEV)}, [B, 294, 1, 143], eta=eta)
    keep = sorted(space_timesteps(1000, f'ddim{n}'))
    tab = odiff.tables(odiff.respaced_betas(odiff.cosine_betas(1000), keep), spaced=False)
    fn = lambda x, i: nets.posenet_forward(sd, x, cond, torch.full((B,), keep[i], dtype=torch.int64))
    ref = odiff.ddim_sample_loop(fn, x_T, noises, tab, list(range(n))[::-1], eta=eta)
    assert max_abs(y.cpu(), ref) < 1e-3


def test_full_1000_step_loop_vs_oracle():
    """The headline configuration end to end on one clip: 1000 ancestral steps through `eval_losses` (fused HIP loop)
    against the CPU oracle with the same injected noise -- the bar of BASELINE.json's north_star (1e-3 on the output
    representation, sub-millimetre MPJPE on the joints recovered from it)."""
    from rohm_amd.body_model import SMPLXLayer
    from rohm_amd.data_loaders.motion_representation import joints_from_repr
    net, sd = make_posenet(31)
    B, S = 4, 1000          # SURVEY.md §8(d) cfg 2: parity run at B = 4
    diff = make_diffusion(S)
    mean, std = synth.synthetic_stats(1)
    cond = synth.plausible_motion(7, B, 143, mean, std)
    x_T, noises = cpu_noise_sequence(99, (B, 294, 1, 143), S)
    diff.noise_source = lambda step, like: (x_T if step == -1 else noises[step])
    _, y = diff.eval_losses(model=net, batch={'cond': cond.to(DEV)}, shape=[B, 294, 1, 143], progress=False,
                            clip_denoised=False, timestep_respacing='', cond_fn_with_grad=False, compute_loss=False)
    torch.set_num_threads(16)
    fn = lambda x, i: nets.posenet_forward(sd, x, cond, torch.full((B,), i, dtype=torch.int64))
    with torch.no_grad():
        ref = odiff.p_sample_loop(fn, x_T, noises, odiff.tables(odiff.cosine_betas(S)), list(range(S))[::-1])
    err = max_abs(y.cpu(), ref)
    layer = SMPLXLayer.from_tensors(synth.synthetic_smplx_tensors(0)).to(DEV)
    j_hip = joints_from_repr(y, 'smplx_params', layer, stats=(mean, std), layout='bc1t')
    j_ref = joints_from_repr(ref.to(DEV), 'smplx_params', layer, stats=(mean, std), layout='bc1t')
    mpjpe_mm = float((j_hip - j_ref).norm(dim=-1).mean()) * 1000.0
    print(f'1000-step loop: max|HIP - oracle| = {err:.3e}, MPJPE = {mpjpe_mm:.4f} mm')
    assert err < 1e-3, err
    assert mpjpe_mm < 1.0, mpjpe_mm


def test_loop_api_details_of_the_reference():
    """Drop-in details of p_sample_loop (gaussian_diffusion_posenet.py:483-576): `save_intermediate_result` returns
    (sample, x0_list, xt_list, t_list) with every (num_timesteps // 5)-th step plus the last one; after a run
    batch['x_t'] is the INPUT of the last executed step (p_mean_variance writes it, :264) on the fused and the
    step-wise path alike."""
    net, sd = make_posenet(3)
    steps = 10
    cond = seeded(4, 1, 294, 1, 143)
    x_T, noises = cpu_noise_sequence(8, (1, 294, 1, 143), steps)
    src = lambda step, like: (x_T if step == -1 else noises[step])
    tab = odiff.tables(odiff.cosine_betas(steps))
    fn = lambda x, i: nets.posenet_forward(sd, x, cond, torch.full((1,), i, dtype=torch.int64))
    trace = odiff.p_sample_loop(fn, x_T, noises, tab, list(range(steps))[::-1], return_all=True)
    diff = make_diffusion(steps)
    diff.noise_source = src
    batch = {'cond': cond.to(DEV)}
    sample, x0s, xts, tls = diff.p_sample_loop(net, batch, [1, 294, 1, 143], save_intermediate_result=True)
    assert tls == [9, 7, 5, 3, 1, 0] and len(x0s) == len(xts) == 6
    assert max_abs(sample.cpu(), trace[-1][0]) < 1e-3
    for k, t_left in enumerate(tls):
        step = steps - 1 - t_left
        assert max_abs(x0s[k].cpu(), trace[step][1]) < 1e-3
        x_in = x_T if step == 0 else trace[step - 1][0]
        assert max_abs(xts[k].cpu(), x_in) < 1e-3
    for chunk in (50, 4):                      # one fused call / several chunks
        diff.fused_chunk = chunk
        batch = {'cond': cond.to(DEV)}
        y = diff.p_sample_loop(net, batch, [1, 294, 1, 143])
        assert max_abs(y.cpu(), trace[-1][0]) < 1e-3
        assert max_abs(batch['x_t'].cpu(), trace[-2][0]) < 1e-3        # input of the last step, not its output


def test_shape_errors_are_raised_before_the_c_abi():
    net, _ = make_posenet(3)
    x = torch.zeros(2, 294, 1, 143, device=DEV)
    with pytest.raises(ValueError):
        net({'x_t': x, 'cond': x[:1]}, torch.zeros(2, dtype=torch.int64, device=DEV))
    with pytest.raises(ValueError):
        net({'x_t': x, 'cond': x}, torch.zeros(3, dtype=torch.int64, device=DEV))


@pytest.mark.parametrize('latent,heads,ff,T,B', [(256, 4, 1024, 99, 3), (512, 4, 1024, 60, 2), (512, 8, 512, 143, 2),
                                                 (256, 2, 256, 195, 1)])
def test_forward_other_widths_and_clip_lengths(latent, heads, ff, T, B):
    """The reference class is not tied to the released configuration (model/posenet.py:12-20: latent_dim=256, 4 heads by
    default, any clip length): widths 256 / 512, head dims 64 / 128 and other T go through the general attention kernel."""
    from rohm_amd.model.posenet import PoseNet
    net = PoseNet(PoseDataset(), 294, latent_dim=latent, ff_size=ff, num_layers=2, num_heads=heads, traj_feat_dim=22,
                  body_model_path=torch.nn.Identity(), device=DEV)
    sd = synth.posenet_state_dict(41, latent_dim=latent, ff_size=ff, num_layers=2)
    net.load_state_dict(sd, strict=True)
    net = net.to(DEV).eval()
    x, c = seeded(3, B, 294, 1, T), seeded(4, B, 294, 1, T)
    t = torch.tensor([(337 * i + 11) % 1000 for i in range(B)])
    ref = nets.posenet_forward(sd, x, c, t, n_head=heads, dtype=torch.float64)
    y = net({'x_t': x.to(DEV), 'cond': c.to(DEV)}, t.to(DEV)).cpu()
    assert max_abs(y, ref) < 1e-4
    # and a short fused loop at this shape
    diff = make_diffusion(6)
    x_T, noises = cpu_noise_sequence(5, (B, 294, 1, T), 6)
    diff.noise_source = lambda step, like: (x_T if step == -1 else noises[step])
    _, out = diff.eval_losses(model=net, batch={'cond': c.to(DEV)}, shape=[B, 294, 1, T], progress=False,
                              clip_denoised=False, timestep_respacing='', cond_fn_with_grad=False, compute_loss=False)
    fn = lambda xx, i: nets.posenet_forward(sd, xx, c, torch.full((B,), i, dtype=torch.int64), n_head=heads)
    ref = odiff.p_sample_loop(fn, x_T, noises, odiff.tables(odiff.cosine_betas(6)), list(range(6))[::-1])
    assert max_abs(out.cpu(), ref) < 1e-3


def test_two_streams_get_distinct_workspaces_and_agree():
    """The C ABI is re-entrant across streams given distinct workspaces; the binding hands out one per stream."""
    net, sd = make_posenet(77)
    x, c = seeded(1, 2, 294, 1, 143).to(DEV), seeded(2, 2, 294, 1, 143).to(DEV)
    t = torch.tensor([5, 700], device=DEV)
    ref = net({'x_t': x, 'cond': c}, t)
    s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
    torch.cuda.synchronize()
    outs = []
    for st in (s1, s2):
        with torch.cuda.stream(st):
            for _ in range(3):
                outs.append(net({'x_t': x, 'cond': c}, t))
    torch.cuda.synchronize()
    assert len({k[2] for k in net.native()._ws}) >= 2
    for o in outs:
        assert torch.equal(o, ref)


def test_full_1000_step_loop_vs_reference_golden():
    """The headline configuration against the REFERENCE's own sampler, not only the oracle: 1000 ancestral steps of one clip
    through `eval_losses` (fused HIP loop) vs tests/golden/posenet_loop1000.npz (reference PoseNet + SpacedDiffusionPoseNet
    on CPU, generator noise replayed)."""
    g = golden('posenet_loop1000.npz')
    net, _ = make_posenet(int(g['weight_seed']))
    mean, std = synth.synthetic_stats(int(g['stats_seed']))
    cond = synth.plausible_motion(int(g['cond_seed']), 1, 143, mean, std)
    x_T, noises = cpu_noise_sequence(int(g['torch_seed']), (1, 294, 1, 143), 1000)
    diff = make_diffusion(1000)
    diff.noise_source = lambda step, like: (x_T if step == -1 else noises[step])
    _, y = diff.eval_losses(model=net, batch={'cond': cond.to(DEV)}, shape=[1, 294, 1, 143], progress=False,
                            clip_denoised=False, timestep_respacing='', cond_fn_with_grad=False, compute_loss=False)
    err = max_abs(y.cpu(), torch.from_numpy(g['y']))
    print(f'1000-step loop vs the reference: max|HIP - reference| = {err:.3e}')
    assert err < 1e-3, err


def test_headline_batch_is_clip_independent_and_meets_the_reference_golden():
    """The headline SIZE (BASELINE.json configs[1]: B = 64 -> the widest GEMM tiles, the full-occupancy attention shape) held
    to the reference through a size-independent property: clips are independent, so clip 0 of a 64-clip 1000-step run --
    its inputs and noise stream the reference's own (tests/golden/posenet_loop1000.npz), the other 63 clips arbitrary --
    must land on the reference's 1-clip result.  Also: a second identical run is bit-identical (no atomics, fixed orders)."""
    g = golden('posenet_loop1000.npz')
    net, _ = make_posenet(int(g['weight_seed']))
    mean, std = synth.synthetic_stats(int(g['stats_seed']))
    B = 64
    cond = torch.cat([synth.plausible_motion(int(g['cond_seed']), 1, 143, mean, std),
                      synth.plausible_motion(4242, B - 1, 143, mean, std)]).to(DEV)
    x_T, noises = cpu_noise_sequence(int(g['torch_seed']), (1, 294, 1, 143), 1000)
    gen = torch.Generator(device=DEV)

    def noise_source(step, like):          # clip 0: the reference's stream; clips 1..63: a device stream keyed by the step
        gen.manual_seed(1000 + step)
        rest = torch.randn(B - 1, 294, 1, 143, device=DEV, generator=gen)
        first = (x_T if step == -1 else noises[step]).to(DEV)
        return torch.cat([first, rest])
    outs = []
    for _ in range(2):
        diff = make_diffusion(1000)
        diff.noise_source = noise_source
        _, y = diff.eval_losses(model=net, batch={'cond': cond}, shape=[B, 294, 1, 143], progress=False,
                                clip_denoised=False, timestep_respacing='', cond_fn_with_grad=False, compute_loss=False)
        outs.append(y.clone())
    err = max_abs(outs[0][:1].cpu(), torch.from_numpy(g['y']))
    print(f'clip 0 of the 64-clip 1000-step run vs the reference: {err:.3e}')
    assert err < 1e-3, err
    assert torch.isfinite(outs[0]).all()
    assert torch.equal(outs[0], outs[1])


@pytest.mark.parametrize('hoist,head_sk', [('1', '1'), ('0', '1'), ('1', '0'), ('0', '0')])
def test_loop8_variants_of_the_launch_plan_vs_reference_golden(hoist, head_sk, monkeypatch):
    """Two launch-plan choices of the sampling loop, each against the reference's own 8-step run: the cond half of the input embedding
    computed once per call (on: K = 320 per step + a row table; off: K = 608 every step), and the output head as a stream-K launch.
    They change summation order only."""
    monkeypatch.setenv('ROHM_POSENET_COND_HOIST', hoist)
    monkeypatch.setenv('ROHM_POSENET_HEAD_SK', head_sk)
    g = golden('posenet_loop8.npz')
    net, _ = make_posenet(int(g['weight_seed']))
    steps = int(g['steps'])
    cond = seeded(int(g['cond_seed']), 2, 294, 1, 143)
    x_T, noises = cpu_noise_sequence(int(g['torch_seed']), (2, 294, 1, 143), steps)
    diff = make_diffusion(steps)
    diff.noise_source = lambda step, like: (x_T if step == -1 else noises[step])
    y = diff.p_sample_loop(net, {'cond': cond.to(DEV)}, [2, 294, 1, 143])          # one fused call of 8 steps
    err = max_abs(y.cpu(), torch.from_numpy(g['y']))
    print(f'hoist={hoist} head_sk={head_sk}: {err:.3e}')
    assert err < 1e-4, err


def test_forward_recorded_into_a_graph_keeps_the_in_kernel_exchanges():
    """The LayerNorm-carrying GEMMs and the stream-K head tag their exchange slots with (salt + launch index) + 64 x a pass counter
    that lives in the WORKSPACE and is advanced on the device by the first kernel of every pass: a forward recorded into a hipGraph
    keeps the fused launches, and every replay draws fresh tags -- bit-equal to the eager fused forward on new inputs."""
    from rohm_amd import _lib
    B, T = 32, 143
    x, c = seeded(1, B, 294, 1, T).to(DEV), seeded(2, B, 294, 1, T).to(DEV)
    t = torch.tensor([(37 * i + 1) % 1000 for i in range(B)], device=DEV)
    net, _ = make_posenet(5)
    refs = [net({'x_t': x * k, 'cond': c}, t) for k in (1.0, 0.5)]          # eager: fused LayerNorm, stream-K head
    nat = net.native(torch.device(DEV))
    if nat.exchange_mode & 3 != 3:
        pytest.skip(f'the layout guard refused the exchanging launches on this device: {nat.exchange_guard}')
    side = torch.cuda.Stream()
    xs = x.clone()
    with torch.cuda.stream(side):
        net({'x_t': xs, 'cond': c}, t)                                      # this stream's workspace exists (and is armed) before the capture
        ws = nat.workspace(B, T)
    torch.cuda.synchronize()
    off = _lib.lib().rohm_posenet_status_offset(nat.handle, B, T)
    word = ws[off:off + 12].view(torch.int32)
    pass0 = int(word[2])
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph, stream=side):
        y = net({'x_t': xs, 'cond': c}, t)
    for n, (k, r) in enumerate(zip((1.0, 0.5, 1.0, 0.5), (refs[0], refs[1], refs[0], refs[1]))):
        xs.copy_(x * k)
        graph.replay()
        torch.cuda.synchronize()
        assert torch.equal(y, r), k                                         # the recorded launches ARE the fused ones
        assert int(word[2]) == pass0 + n + 1                                # one pass per replay: new tags every time
    assert int(word[0]) == 0
    with torch.cuda.stream(side):
        net.check_exchange()


def test_exchange_status_is_sticky_and_the_loops_recover():
    """Two kernels of the forward hand data between workgroups of one launch (LayerNorm statistics, stream-K partials); their
    bounded waits report into a status word of the workspace.  The word survives later calls on that workspace until
    rohm_posenet_exchange_status reads it.  Direct forwards: check_exchange raises.  Sampling loops: they switch the handle to the
    exchange-free launches, warn, and re-run the chunk (the word is poked by hand here; tests/test_gpu_exchange.py provokes real
    failures)."""
    from rohm_amd import _lib
    lib = _lib.lib()
    net, _ = make_posenet(5)
    B, T = 2, 143
    x, c = seeded(1, B, 294, 1, T).to(DEV), seeded(2, B, 294, 1, T).to(DEV)
    t = torch.tensor([3, 700], device=DEV)
    nat = net.native(torch.device(DEV))
    ws = nat.workspace(B, T)
    ws.fill_(0xAB)                                  # a workspace nobody initialised: the first call arms the status words
    y0 = net({'x_t': x, 'cond': c}, t)
    net.check_exchange()                            # clean
    off = lib.rohm_posenet_status_offset(nat.handle, B, T)
    word = ws[off:off + 12].view(torch.int32)
    assert int(word[0]) == 0 and int(word[1]) == 0x524f484d and int(word[2]) == 1
    word[0] = 1                                     # "a wait ran into its bound"
    y1 = net({'x_t': x, 'cond': c}, t)              # later calls do not clear it
    assert torch.equal(y0, y1) and int(word[0]) == 1 and int(word[2]) == 2
    with pytest.raises(_lib.RohmHipError, match='code -5'):
        net.check_exchange()
    net.check_exchange()                            # reported once, cleared
    dif = make_diffusion(4)
    x_T, noises = cpu_noise_sequence(11, (B, 294, 1, T), 4)
    dif.noise_source = lambda step, like: (x_T if step == -1 else noises[step])
    clean = dif.p_sample_loop(net, {'cond': c}, [B, 294, 1, T])
    mode = nat.exchange_mode
    word[0] = 2
    if mode & 3:
        with pytest.warns(UserWarning, match='different XCDs'):                # the loops check after every chunk: fall back, re-run
            again = dif.p_sample_loop(net, {'cond': c}, [B, 294, 1, T])
        assert nat.exchange_mode & 3 == 0 and nat.exchange_mode & 8
        assert max_abs(again, clean) < 1e-4                                    # same run on the exchange-free launches
        word[0] = 1
    with pytest.raises(_lib.RohmHipError, match='not using the exchanging launches'):      # nothing left to fall back to: raise
        dif.p_sample_loop(net, {'cond': c}, [B, 294, 1, T])
    net.check_exchange()
    final = dif.p_sample_loop(net, {'cond': c}, [B, 294, 1, T])                # and pass when nothing happened
    assert max_abs(final, clean) < 1e-4
    nat.set_exchange(True)                                                     # back to what the guard allowed
    assert nat.exchange_mode == mode


def test_empty_batch_passes_through():
    """B = 0 (an empty shard of a ragged split): forward and the whole sampling run return empty tensors of the right shape,
    as the reference's torch modules do, without touching the C ABI (which rejects B <= 0)."""
    from test_gpu_trajnet import make_trajnet
    from test_gpu_trajnet import make_diffusion as make_traj_diffusion
    net, _ = make_posenet(3)
    x = torch.zeros(0, 294, 1, 143, device=DEV)
    y = net({'x_t': x, 'cond': x}, torch.zeros(0, dtype=torch.int64, device=DEV))
    assert tuple(y.shape) == (0, 294, 1, 143)
    diff = make_diffusion(1000)
    _, y = diff.eval_losses(model=net, batch={'cond': x}, shape=[0, 294, 1, 143], progress=False, clip_denoised=False,
                            timestep_respacing='', cond_fn_with_grad=False, compute_loss=False)
    assert tuple(y.shape) == (0, 294, 1, 143)
    tnet, _ = make_trajnet(3, False)
    xt = torch.zeros(0, 144, 13, device=DEV)
    assert tuple(tnet({'x_t': xt, 'cond': xt}, torch.zeros(0, dtype=torch.int64, device=DEV)).shape) == (0, 144, 13)
    _, yt = make_traj_diffusion().eval_losses(model=tnet, batch={'cond': xt}, shape=[0, 144, 13], progress=False,
                                             clip_denoised=False, timestep_respacing='', cond_fn_with_grad=False,
                                             compute_loss=False)
    assert tuple(yt.shape) == (0, 144, 13)


def test_large_batch_is_clip_independent():
    """B = 256 picks GEMM tile widths the smaller batches never reach (e.g. 144 x 256 for the N = 512 GEMMs, several rounds of
    144 x 384 for QKV); clips are independent, so every 64-clip slice must agree with a 64-clip forward of the same inputs
    (the K order of the fp32 MFMA chains does not depend on the tile width: agreement to the last few ulps)."""
    net, _ = make_posenet(5)
    B = 256
    x, c = seeded(21, B, 294, 1, 143).to(DEV), seeded(22, B, 294, 1, 143).to(DEV)
    t = (torch.arange(B, device=DEV) * 7) % 1000
    y = net({'x_t': x, 'cond': c}, t)
    assert torch.isfinite(y).all()
    for lo in (0, 64, 192):
        sl = slice(lo, lo + 64)
        ys = net({'x_t': x[sl].contiguous(), 'cond': c[sl].contiguous()}, t[sl].contiguous())
        assert max_abs(y[sl], ys) < 2e-5, lo
