"""GPU parity AT THE BATCH SIZES OF THE BASELINE CONFIGS, for the kernels those sizes actually launch.

From 32 clips on a PoseNet forward is one `encoder_stack_kernel<8>` (B = 32: configs[2-4] per GPU) or `<4>` (B = 64: the headline)
launch (csrc/encoder_chain.hip), a path the B = 2 reference fixtures never reach.  Here:

  * every clip of a B = 64 and of a B = 32 forward, and of an 8-step fused loop, against the CPU oracle (float64 / fp32), default
    environment, with `exchange_mode & 32` (stack in use) asserted;
  * the reference's own free-running runs (tests/golden/scheme_real.npz: the scripts' loop text of test_amass_full.py:217-384 /
    test_prox_egobody.py:214-324 with the reference's networks and samplers, B = 2; tests/golden/guided_head.npz: its own
    p_sample_with_grad over t = 103..97) EMBEDDED at slots 5 and 29 of a 32-clip batch: clips are independent in both networks
    (model/posenet.py:75-96 -- attention, LayerNorm and GroupNorm never cross the batch), so with the reference's inputs and noise
    stream in those two slots and filler clips elsewhere the two clips must land on the reference's B = 2 result.  A defect tied to
    a clip's slot (clip -> XCD map, part index, the cooperative query block) cannot hide behind clip 0.
    Guidance couples clips through batch-wide normalisers (model/posenet.py:231,243,309): the guided cases run with the global-batch
    hooks of rohm_amd.sharding restricted to the two golden clips (mask counts of those clips, batch size 2), i.e. the normalisers
    of the reference's B = 2 run, while the gradient kernels run on all 32 clips.
"""
import numpy as np
import pytest
import torch

from helpers import PoseDataset, cpu_noise_sequence, cpu_noise_stream, golden, max_abs, seeded
from oracle import diffusion as odiff
from oracle import geometry as G
from oracle import nets
from rohm_amd.utils import synth

pytestmark = pytest.mark.gpu
DEV = 'cuda:0'
SLOTS = (5, 29)


def _stack_bit(net):
    nat = net.native(torch.device(DEV))
    mode = nat.exchange_mode
    if mode & 4:
        pytest.skip(f'the layout guard refused the exchanging launches on this device: {nat.exchange_guard}')
    return mode & 32


def embed(gold, fill, B=32, slots=SLOTS):
    """[len(slots), ...] golden clips at `slots` of a B-clip batch, `fill` [B - len(slots), ...] elsewhere (device of `fill`)."""
    out = torch.empty((B,) + tuple(gold.shape[1:]), dtype=gold.dtype, device=fill.device)
    rest = torch.ones(B, dtype=torch.bool)
    rest[list(slots)] = False
    out[torch.tensor(slots, device=fill.device)] = gold.to(fill.device)
    out[rest.to(fill.device)] = fill
    return out


def embed_batch(gold, fill, B=32, slots=SLOTS):
    n = len(slots)
    out = {}
    for k, v in gold.items():
        assert torch.is_tensor(v) and v.shape[0] == n and fill[k].shape[0] == B - n, k
        out[k] = embed(v, fill[k].to(DEV), B, slots)
    return out


class EmbeddedNoise:
    """The reference's CPU noise stream (B = 2 draws) at SLOTS, a seeded device stream for the filler clips."""

    def __init__(self, runs, B=32, slots=SLOTS, seed=9000):
        self.runs, self.k, self.B, self.slots = runs, -1, B, slots
        self.gen = torch.Generator(device=DEV)
        self.seed = seed

    def __call__(self, step, like):
        if step == -1:
            self.k += 1
        gold = self.runs[self.k][0] if step == -1 else self.runs[self.k][1][step]
        self.gen.manual_seed(self.seed + 5000 * self.k + step + 1)
        fill = torch.randn((self.B - len(self.slots),) + tuple(gold.shape[1:]), device=DEV, generator=self.gen)
        return embed(gold, fill, self.B, self.slots)


def restrict_guidance_to(pnet, slots):
    """Global-batch guidance hooks (rohm_amd.sharding.use_global_batch_guidance) whose "all-reduce" returns the skating mask counts
    of the clips at `slots` only and whose global batch size is len(slots): the batch-wide normalisers of model/posenet.py:231,243,309
    then equal those of the reference run that holds just these clips."""
    from rohm_amd import guidance
    from rohm_amd.sharding import use_global_batch_guidance
    state = {}
    orig = pnet.guide_skating_with_smpl
    idx = torch.tensor(slots, device=DEV)

    def skating(batch, out, denoise_t, compute_grad='x_t'):
        x0 = out['pred_xstart'] if compute_grad == 'x_0' else batch['x_t']
        grp, pnet.guidance_group = pnet.guidance_group, None
        try:
            _, state['counts'] = guidance.guide_skating(pnet, {}, {'pred_xstart': x0[idx].contiguous()}, denoise_t, 'x_0',
                                                        return_counts=True)
        finally:
            pnet.guidance_group = grp
        return orig(batch, out, denoise_t, compute_grad)
    pnet.guide_skating_with_smpl = skating
    use_global_batch_guidance(pnet, group=lambda t: t.copy_(state['counts']), global_batch=len(slots))
    return pnet


# ----------------------------------------------------------------------------------------------- every clip vs the oracle
def _posenet(seed):
    from test_gpu_posenet import make_posenet
    return make_posenet(seed)


@pytest.mark.parametrize('B', [64, 32])
def test_every_clip_of_a_config_sized_forward_vs_float64_oracle(B):
    """`PoseNet.forward` (model/posenet.py:75-96) at B = 64 (`encoder_stack_kernel<4>`) and B = 32 (`<8>`): ALL clips against
    oracle.nets.posenet_forward in float64, distinct timesteps per clip, default environment."""
    net, sd = _posenet(77)
    x, c = seeded(11, B, 294, 1, 143), seeded(12, B, 294, 1, 143)
    t = torch.tensor([(131 * i + 5) % 1000 for i in range(B)])
    y = net({'x_t': x.to(DEV), 'cond': c.to(DEV)}, t.to(DEV))
    net.check_exchange()
    assert _stack_bit(net), 'the encoder stack is the default from 32 clips on'
    torch.set_num_threads(8)
    with torch.no_grad():
        ref = nets.posenet_forward(sd, x, c, t, dtype=torch.float64)
    per_clip = (y.cpu().double() - ref).abs().flatten(1).max(dim=1).values
    print(f'\nB = {B}: max over clips of max|HIP - float64 oracle| = {float(per_clip.max()):.3e} (clip {int(per_clip.argmax())})')
    assert float(per_clip.max()) < 1e-4, per_clip
    assert torch.equal(y[:, :22].cpu(), c[:, :22])


@pytest.mark.parametrize('B', [64, 32])
def test_every_clip_of_a_config_sized_fused_loop_vs_oracle(B):
    """Eight ancestral steps through `eval_losses` at B = 64 / 32 -- in the loop the stack carries the input embedding as its leading
    phase and `finish_pack` closes a step -- every clip against the oracle's loop (fp32 CPU, same injected noise).  Bar 1e-3."""
    from test_gpu_posenet import make_diffusion
    net, sd = _posenet(31)
    S = 8
    mean, std = synth.synthetic_stats(1)
    cond = synth.plausible_motion(7, B, 143, mean, std)
    x_T, noises = cpu_noise_sequence(99, (B, 294, 1, 143), S)
    diff = make_diffusion(S)
    diff.noise_source = lambda step, like: (x_T if step == -1 else noises[step])
    diff.fused_chunk = 5
    _, y = diff.eval_losses(model=net, batch={'cond': cond.to(DEV)}, shape=[B, 294, 1, 143], progress=False,
                            clip_denoised=False, timestep_respacing='', cond_fn_with_grad=False, compute_loss=False)
    assert _stack_bit(net)
    torch.set_num_threads(8)
    fn = lambda x, i: nets.posenet_forward(sd, x, cond, torch.full((B,), i, dtype=torch.int64))
    with torch.no_grad():
        ref = odiff.p_sample_loop(fn, x_T, noises, odiff.tables(odiff.cosine_betas(S)), list(range(S))[::-1])
    per_clip = (y.cpu() - ref).abs().flatten(1).max(dim=1).values
    print(f'\nB = {B}, 8 steps: max over clips of max|HIP - oracle| = {float(per_clip.max()):.3e}')
    assert float(per_clip.max()) < 1e-3, per_clip


# ----------------------------------------------------------------------------------------------- reference runs embedded in B = 32
def _filler_case(ci, n):
    """`n` filler clips with the structure of scheme case `ci` (other seeds: they must not mirror the golden clips)."""
    from oracle.make_golden import scheme_real_case
    args, tfd, body_t, s_traj, s_pose, bt, bp, cam, _, _ = scheme_real_case(ci, B=n)
    roll = lambda d: {k: torch.roll(v, 3, 0) + (0.01 * seeded(700 + j, *v.shape) if k.startswith('motion_repr') else 0)
                      for j, (k, v) in enumerate(d.items())}
    bt, bp = roll(bt), roll(bp)
    bt['cond'] = bt['motion_repr_noisy'][:, :, [0, 2, 3, 6, 7, 8, 9, 10, 11, 12, 16, 17, 18]].contiguous()
    if cam:
        cam = synth.synthetic_camera_batch(40, n)
    return bt, bp, cam


@pytest.mark.parametrize('ci', range(5))
def test_reference_scheme_runs_embedded_in_a_32_clip_batch(ci):
    """BASELINE configs[2] / [4] at the per-GPU batch of the configs (32 clips): the reference's own free-running runs of
    tests/golden/scheme_real.npz (see test_gpu_scheme.py::test_free_running_scheme_vs_reference_golden for the five cases) at slots
    5 and 29, filler clips elsewhere.  TrajNet / TrajControl run their B = 32 launch plans, PoseNet the encoder stack.  Case 2 is
    guided (reference weights, t = 103..99): its normalisers are restricted to the two golden clips."""
    from oracle.make_golden import (SCHEME_REAL_CAM_T, SCHEME_REAL_CASES, SCHEME_REAL_HEAD_T, SCHEME_REAL_SEEDS,
                                    scheme_real_case)
    from rohm_amd import inference as INF
    from test_gpu_scheme import Recording, TrajDataset, _diffusions, _real_models
    g = golden('scheme_real.npz')
    kind, kw, pose_steps = SCHEME_REAL_CASES[ci]
    args, tfd, body_t, s_traj, s_pose, bt, bp, cam, _, plan = scheme_real_case(ci)
    B = 32
    fbt, fbp, fcam = _filler_case(ci, B - 2)
    noise = cpu_noise_stream(SCHEME_REAL_SEEDS['noise'] + ci, plan)
    layer, models, _, pds = _real_models(B, s_pose, body_t, SCHEME_REAL_SEEDS, cam_t=SCHEME_REAL_CAM_T)
    head = pose_steps == 'head'
    log = []
    diffs, d_p = _diffusions(100, 1000 if head else pose_steps, noise, log)
    diffs['trajnet'].diff.noise_source = EmbeddedNoise(noise['traj'][:1], B, seed=9100)
    diffs['trajnet_control'].diff.noise_source = EmbeddedNoise(noise['traj'][1:], B, seed=9200)
    d_p.noise_source = EmbeddedNoise(noise['pose'], B, seed=9300)
    if head:
        d_p._indices = lambda skip=0, early_stop=False: list(SCHEME_REAL_HEAD_T)
        restrict_guidance_to(models['posenet'], SLOTS)
    gbt = embed_batch(bt, fbt, B)
    gbp = embed_batch(bp, fbp, B)
    if cam:
        gbp.update(embed_batch(cam, fcam, B))
    fn = INF.run_amass_iterations if kind == 'amass' else INF.run_prox_iterations
    pose, traj, recs = fn(args, models, diffs, gbt, gbp, TrajDataset(*s_traj), pds, layer)
    assert _stack_bit(models['posenet'])
    sl = list(SLOTS)
    pre = f'case{ci}_'
    assert len(log) == int(g[pre + 'n_stages'])
    errs = [max_abs(o[sl], torch.from_numpy(g[pre + f'stage{k}_out'])) if pre + f'stage{k}_out' in g else float('nan')
            for k, (_, _, o) in enumerate(log)]
    e_pose = max_abs(pose[sl].cpu(), torch.from_numpy(g[pre + 'pose']))
    e_traj = max_abs(traj[sl].cpu(), torch.from_numpy(g[pre + 'traj']))
    e_rec = max_abs(recs[-1][sl].cpu(), torch.from_numpy(g[pre + 'traj_rec_full']))
    print(f'\nscheme case {ci} ({kind}, PoseNet {pose_steps}) at slots {SLOTS} of 32: per-stage max|HIP - reference| =',
          ['%.2e' % e for e in errs], f'final pose {e_pose:.2e} traj {e_traj:.2e} traj_rec_full {e_rec:.2e}')
    assert torch.isfinite(pose[sl]).all() and torch.isfinite(traj[sl]).all()
    assert e_traj < 1e-3 and e_rec < 1e-3 and e_pose < 1e-3, errs
    den = lambda y: torch.from_numpy(y[:, :, 0].transpose(0, 2, 1) * s_pose[1] + s_pose[0])
    body = G.BodyModel(body_t)
    j_hip = G.joints_from_smplx(G.split_repr(den(pose[sl].cpu().numpy())), body)
    j_ref = G.joints_from_smplx(G.split_repr(den(g[pre + 'pose'])), body)
    mpjpe_mm = float((j_hip - j_ref).norm(dim=-1).mean()) * 1000
    print(f'MPJPE vs reference {mpjpe_mm:.5f} mm')
    assert mpjpe_mm < 1.0


def test_reference_guided_head_embedded_in_a_32_clip_batch():
    """BASELINE configs[3] (PROX guidance, B = 32): the reference's own free-running `p_sample_with_grad(grad_type='prox')` over
    t = 103..97 at its own weights (tests/golden/guided_head.npz, B = 2) at slots 5 and 29 of a 32-clip batch -- stack forward,
    both gradient kernels and the two-gradient update on all 32 clips, normalisers of the reference's run."""
    from oracle.make_golden import guided_step_inputs
    from test_gpu_guidance import _diffusion, _prox_net
    g = golden('guided_head.npz')
    net = restrict_guidance_to(_prox_net(g), SLOTS)
    B = 32
    mean, std, x, cond, cam = guided_step_inputs(g)
    seeds = {k: int(g[k]) + 100 for k in ('stats_seed', 'x_seed', 'xn_seed', 'cond_seed', 'cam_seed')}
    seeds['stats_seed'] = int(g['stats_seed'])
    _, _, fx, fcond, fcam = guided_step_inputs(seeds, B=B - 2)
    batch = embed_batch(cam, fcam, B)
    batch['cond'] = embed(cond, fcond.to(DEV), B)
    xx = embed(x, fx.to(DEV), B)
    diff = _diffusion()
    gen = torch.Generator(device=DEV)
    for k, i in enumerate(int(v) for v in g['t']):
        torch.manual_seed(int(g['noise_seed0']) + k)
        gold = torch.randn(2, 294, 1, 143)
        gen.manual_seed(77 + k)
        noise = embed(gold, torch.randn(B - 2, 294, 1, 143, device=DEV, generator=gen), B)
        diff.noise_source = lambda step, like, noise=noise: noise
        t = torch.full((B,), i, device=DEV, dtype=torch.int64)
        xx = diff.p_sample_with_grad(net, batch, xx, t, clip_denoised=False, grad_type='prox')['sample']
    assert _stack_bit(net)
    ref = torch.from_numpy(g['sample'])
    err = max_abs(xx[list(SLOTS)].cpu(), ref)
    print(f'\nguided head t = 103..97 at slots {SLOTS} of 32: max|HIP - reference| = {err:.3e} on max|x| = {float(ref.abs().max()):.1f}')
    assert err < 5e-3          # the B = 2 test's bar (test_gpu_guidance.py::test_free_running_guided_head_vs_reference_golden)
