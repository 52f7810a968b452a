#!/usr/bin/env python
"""Headline benchmark: denoised 145-frame clips/s at 1000 DDPM steps (BASELINE.json `metric`).

    python bench.py [--gpus N] [--steps K] [--warmup W] [--batch 64] [--ddpm-steps 1000]

One "step" = one complete pass of the hot path over one batch: a full PoseNet `eval_losses` run
(x_T -> x_0 through 1000 ancestral DDPM steps, on-device noise generation included) for `--batch`
synthetic clips per GPU (BASELINE.json configs[1]: B = 64 on one MI355X).  Clips are independent, so
with N > 1 every rank denoises its own B clips (weak scaling, no data-path collective) and the ranks
exchange only the finished results with one RCCL all-gather inside the timed region.

`--gpus N` with N > 1 and no WORLD_SIZE in the environment re-executes this script under
`python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1` (one rank per GPU over
RCCL); launched by the driver under torch.distributed.run it reads RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* as
given.  A world size that disagrees with --gpus, or fewer visible GPUs than ranks, is an error, never a silent
1-rank run.

Rank 0 prints ONE JSON line (contract in the task statement) carrying `roofline` (the fp32-MFMA GEMM
family: algorithmic flops / HIP-event-measured launch time, sampled every 16th denoising step of the
timed region on the launch stream) and, at N = 1, `cpu_baseline` (the reference's own p_sample on its own PoseNet where
/root/reference exists -- the build container -- else the CPU oracle port of the same step, timed on the host cores).

At N = 1 the default run then adds, each measured by a child process AFTER the headline leg (thermal history: the headline
always runs first on a cold chip) and none of them ever replacing `value` / `dtype` / `roofline`, which stay exact fp32:
  `second_line`  the same workload under ROHM_GEMM_PRECISION=fp16x3 (split GEMMs on fp16 planes) with its own dtype,
                 roofline (fp16 MFMA peak / 3) and 1000-step accuracy against the reference's own run (+ `also.bf16x6`);
  `configs`      one short pass each of the other BASELINE.json configurations on this GPU: `b32` (PoseNet at the per-GPU
                 batch of configs[2..4]), `scheme_b32` (configs[2]), `prox_b32` (configs[3]), `egobody_b32` (configs[4]).
`--no-extras` skips both (the children run with it).  `--force-dist` initialises the RCCL process group even at world size
1 and runs the barrier / MAX all-reduce / result all-gather of the multi-GPU path through it.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

# dmabuf IPC for RCCL on this driver: must be in the environment before the first HIP call of the process
os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

PEAK_F32_MFMA_TFLOPS = 157.3          # MI355X_MICROARCH.md: v_mfma_f32_16x16x4_f32, dense, 2.4 GHz
PEAK_BF16_MFMA_TFLOPS = 2500.0        # dense bf16 MFMA
# Opt-in precision ladder of the GEMMs (rohm_amd/csrc/gemm_f32.hip): the default and the headline are exact fp32 MFMA.
_PREC = os.environ.get('ROHM_GEMM_PRECISION', '')
_PRODUCTS = {'bf16x6': 6, 'bf16x3': 3, 'fp16x3': 3}.get(_PREC, 0)
_PLANE_TYPE = 'fp16' if _PREC == 'fp16x3' else 'bf16'
POSENET_GFLOP_PER_CLIP_STEP = 5.298   # SURVEY.md §8(d) / BASELINE.md §2


class _Dataset:
    pose_feat_dim, traj_feat_dim, body_feat_dim = 272, 22, 294
    Mean = np.zeros(294, np.float32)
    Std = np.ones(294, np.float32)


class _Args:
    noise_schedule = 'cosine'
    sigma_small = True


def synthetic_cond(B, device, seed):
    """cfg 2 of SURVEY.md §8(d): cond ~ N(0,1), contact channels zeroed, lower-body joints
    [1,2,4,5,7,8,10,11] occluded as in test_amass_full.py:340-348."""
    g = torch.Generator(device='cpu').manual_seed(seed)
    cond = torch.randn(B, 143, 294, generator=g)
    jid = np.asarray([1, 2, 4, 5, 7, 8, 10, 11])
    for k in range(3):
        cond[:, :, 22 + jid * 3 + k] = 0.
        cond[:, :, 22 + 66 + jid * 3 + k] = 0.
    for k in range(6):
        cond[:, :, 22 + 132 + (jid - 1) * 6 + k] = 0.
    cond[:, :, -4:] = 0.
    return cond.permute(0, 2, 1).unsqueeze(2).contiguous().to(device)


def pmc_traffic():
    """HBM-side bytes per GEMM launch from the committed rocprofv3 PMC passes (profiles/*pmc_traffic.json,
    produced by scripts/gpu_profile.sh + scripts/pmc_summary.py; FETCH_SIZE x2 on gfx950).  PMC collection
    serialises kernels, so it is a separate run of the same workload, never part of the timed region."""
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, 'profiles', '*pmc_traffic.json')))
    if not files:
        return None, None
    try:
        d = json.load(open(files[-1]))
        return float(d['gemm_family_bytes_per_launch']), os.path.basename(files[-1])
    except (OSError, ValueError, KeyError):
        return None, None


def pmc_traffic_of(kernel_substr):
    """ARCHIVED, like pmc_traffic(): HBM-side bytes per launch of ONE kernel from the most recent committed PMC passes."""
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, 'profiles', '*pmc_traffic.json')))
    if not files:
        return None
    try:
        k = json.load(open(files[-1]))['kernels']
        hit = [v for n, v in k.items() if kernel_substr in n]
        return float(hit[0]['hbm_bytes_per_launch']) if hit else None
    except (OSError, ValueError, KeyError):
        return None


def pmc_mfma(batch):
    """ARCHIVED counter figures, not part of this run: MFMA-pipe utilisation of the GEMM family as rocprofv3 counted it in the
    most recent committed PMC pass of THIS batch size (profiles/*pmc_sq[_b<B>].json from scripts/gpu_r2_profile.sh +
    scripts/sq_summary.py: SQ_VALU_MFMA_BUSY_CYCLES / (4 SIMDs x 256 CUs x kernel time x 2.4 GHz), launch-weighted).  PMC
    collection serialises and slows the kernels, so it is always a separate run of an earlier build; None when no pass of
    this batch size is committed."""
    import glob
    want = '' if batch == 64 else f'_b{batch}'
    files = sorted(f for f in glob.glob(os.path.join(ROOT, 'profiles', '*pmc_sq*.json'))
                   if os.path.basename(f).split('pmc_sq')[1] == want + '.json')
    if not files:
        return None
    try:
        k = json.load(open(files[-1]))['kernels']
        gem = {n: v for n, v in k.items() if 'gemm_f32_kernel' in n or 'encoder_stack_kernel' in n or 'encoder_chain_kernel' in n}
        t = sum(v['avg_us_under_pmc'] * v['launches'] for v in gem.values())
        busy = sum(v['mfma_busy_frac_at_2p4GHz'] * v['avg_us_under_pmc'] * v['launches'] for v in gem.values()) / t
        att = [v for n, v in k.items() if 'attention_f32_kernel' in n]
        conf = max(v['lds_bank_conflict_frac'] for v in k.values())
        return {'archived': True, 'note': 'separate rocprofv3 PMC pass of an earlier build at this batch size, not this run',
                'gemm_family_mfma_busy': busy, 'attention_mfma_busy': att[0]['mfma_busy_frac_at_2p4GHz'] if att else None,
                'max_lds_bank_conflict_frac': conf, 'source': os.path.basename(files[-1])}
    except (OSError, ValueError, KeyError, ZeroDivisionError):
        return None


def accuracy_vs_reference(dev):
    """The metric's second half ("MPJPE vs ref"): one clip through all 1000 DDPM steps of the drop-in API with the noise
    stream the REFERENCE drew on CPU, compared with what the reference's own code produced from it
    (tests/golden/posenet_loop1000.npz, generated by oracle/make_golden.py in the build container): max |diff| on the
    294-channel output and MPJPE of the SMPL-X joints recovered from both (synthetic body model: no SMPL-X files here).
    Outside the timed region."""
    path = os.path.join(ROOT, 'tests', 'golden', 'posenet_loop1000.npz')
    if not os.path.exists(path):
        return None
    from rohm_amd.body_model import SMPLXLayer
    from rohm_amd.data_loaders.motion_representation import joints_from_repr
    from rohm_amd.diffusion import gaussian_diffusion_posenet as gdp
    from rohm_amd.diffusion.respace import SpacedDiffusionPoseNet
    from rohm_amd.model.posenet import PoseNet
    from rohm_amd.utils import synth
    from rohm_amd.utils.model_util import create_gaussian_diffusion
    g = np.load(path)
    net = PoseNet(_Dataset(), 294, latent_dim=512, ff_size=1024, num_layers=8, num_heads=4, traj_feat_dim=22,
                  body_model_path=torch.nn.Identity(), device=dev)
    net.load_state_dict(synth.posenet_state_dict(int(g['weight_seed'])), strict=True)
    net = net.to(dev).eval()
    mean, std = synth.synthetic_stats(int(g['stats_seed']))
    cond = synth.plausible_motion(int(g['cond_seed']), 1, 143, mean, std).to(dev)
    state = torch.get_rng_state()
    torch.manual_seed(int(g['torch_seed']))                 # the reference: one randn(*shape), then one randn_like per step
    x_T = torch.randn(1, 294, 1, 143)
    noises = [torch.randn(1, 294, 1, 143) for _ in range(1000)]     # separate calls: one big randn is a different stream
    torch.set_rng_state(state)
    diff = create_gaussian_diffusion(_Args, gdp, SpacedDiffusionPoseNet, 1000, '', device=dev)
    diff.noise_source = lambda step, like: (x_T if step == -1 else noises[step])
    _, y = diff.eval_losses(model=net, batch={'cond': cond}, shape=[1, 294, 1, 143], progress=False, clip_denoised=False,
                            timestep_respacing='', cond_fn_with_grad=False, compute_loss=False)
    ref = torch.from_numpy(g['y']).to(dev)
    layer = SMPLXLayer.from_tensors(synth.synthetic_smplx_tensors(0)).to(dev)
    j_hip = joints_from_repr(y, 'smplx_params', layer, stats=(mean, std), layout='bc1t')
    j_ref = joints_from_repr(ref, 'smplx_params', layer, stats=(mean, std), layout='bc1t')
    return {'max_abs_vs_reference': float((y - ref).abs().max()),
            'mpjpe_mm_vs_reference': float((j_hip - j_ref).norm(dim=-1).mean()) * 1000.0,
            'sample': '1 clip, 1000 DDPM steps, reference CPU run of the same inputs / noise (tests/golden/posenet_loop1000.npz); '
                      'joints through a synthetic SMPL-X model', 'tolerance': '1e-3 / 1 mm (BASELINE.json north_star)'}


def usable_cores():
    """Cores this process may actually use: affinity mask capped by the cgroup CPU quota."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, 'sched_getaffinity') else (os.cpu_count() or 1)
    try:
        quota, period = open('/sys/fs/cgroup/cpu.max').read().split()
        if quota != 'max':
            n = max(1, min(n, int(float(quota) / float(period))))
    except (OSError, ValueError):
        pass
    return n


def cpu_model():
    try:
        for line in open('/proc/cpuinfo'):
            if line.startswith('model name'):
                return line.split(':', 1)[1].strip()
    except OSError:
        pass
    return None


def cpu_baseline(batch=8, budget_s=12.0):
    """One PoseNet p_sample step on the host cores, extrapolated to 1000 steps: the REFERENCE's own
    SpacedDiffusionPoseNet.p_sample on its own model.posenet.PoseNet (diffusion/gaussian_diffusion_posenet.py:388-434) where
    the reference tree exists (kind "reference": the build container), otherwise the oracle port of the same step (kind
    "port": the GPU boxes, where /root/reference does not exist).  Same synthetic weights, same batch."""
    from oracle import refload
    from rohm_amd.utils import synth
    cores = min(usable_cores(), 64)      # torch intra-op scaling flattens (and then degrades) past ~64 threads
    torch.set_num_threads(cores)
    sd = synth.posenet_state_dict(0)
    g = torch.Generator().manual_seed(0)
    x = torch.randn(batch, 294, 1, 143, generator=g)
    cond = torch.randn(batch, 294, 1, 143, generator=g)
    kind = 'port'
    if refload.available():
        try:
            import contextlib
            with contextlib.redirect_stdout(sys.stderr):      # the reference prints while it builds its modules
                ref = refload.load()
                net = ref.posenet.PoseNet(_Dataset(), 294, latent_dim=512, ff_size=1024, num_layers=8, num_heads=4,
                                          traj_feat_dim=22, device='cpu').eval()
            net.load_state_dict(sd, strict=True)
            diff = ref.model_util.create_gaussian_diffusion(_Args, ref.gd_posenet, ref.respace.SpacedDiffusionPoseNet, 1000, '',
                                                            device='cpu')

            def one(i):
                t = torch.full((batch,), i, dtype=torch.int64)
                return diff.p_sample(net, {'cond': cond}, x, t, clip_denoised=False)['sample']
            with torch.no_grad():
                one(999)
            kind = 'reference'
        except Exception as e:      # a reference tree that does not import here: say so and time the port
            print(f'bench.py: reference CPU baseline unavailable ({type(e).__name__}: {e}); timing the oracle port', file=sys.stderr)
    if kind == 'port':
        from oracle import diffusion as odiff
        from oracle import nets
        tab = odiff.tables(odiff.cosine_betas(1000))
        fn = lambda xx, i: nets.posenet_forward(sd, xx, cond, torch.full((batch,), i, dtype=torch.int64))

        def one(i):
            nz = [torch.randn(batch, 294, 1, 143, generator=g)]
            return odiff.p_sample_loop(fn, x, nz, tab, [i])
    # three samples of budget_s / 3 each, the MEDIAN reported with the spread: one draw of a shared host moved by +-25 % between
    # boxes and rounds (0.076-0.132 clips/s in rounds 2-5)
    samples, n_all, t_all, k = [], 0, 0.0, 0
    with torch.no_grad():
        one(999)
        for _ in range(3):
            n, t0 = 0, time.perf_counter()
            while True:
                one(998 - k)
                n, k = n + 1, k + 1
                el = time.perf_counter() - t0
                if el > budget_s / 3.0 or n >= 70:
                    break
            samples.append(el / n)
            n_all, t_all = n_all + n, t_all + el
    sec_per_step = sorted(samples)[1]
    vals = sorted(batch / (sps * 1000.0) for sps in samples)
    what = ("the reference's own SpacedDiffusionPoseNet.p_sample on model.posenet.PoseNet (torch CPU fp32)" if kind == 'reference'
            else 'oracle (torch-CPU fp32 restatement) PoseNet p_sample')
    return {'value': batch / (sec_per_step * 1000.0), 'unit': 'clips/s', 'cores': cores,
            'host_cpus': os.cpu_count(), 'cpu_model': cpu_model(), 'kind': kind,
            'samples': [round(v, 5) for v in vals], 'spread': round((vals[2] - vals[0]) / vals[1], 4),
            'sample': f'{what}, B={batch}, MEDIAN of 3 samples ({n_all} timed steps in {t_all:.1f} s; median {sec_per_step * 1e3:.1f} ms/step) '
                      f'extrapolated to 1000 steps',
            'torch_threads': torch.get_num_threads()}


def cpu_baseline_config(workload, B, budget_s=18.0):
    """CPU baseline of the multi-stage workloads (BASELINE.json configs[2] / [3] / [4]) on the host cores: every KIND of denoising
    step the workload executes is timed with the oracle port (kind "port" -- oracle/nets.py + oracle/diffusion.py + oracle/geometry.py,
    the torch-CPU fp32 restatement pinned to the reference by tests/golden), a few repetitions each, and the pass is priced as
    sum(count x median step time): the reference's step counts (diffusion_steps_trajnet 100, diffusion_steps_posenet 1000, 980 with
    early_stop; guidance on t <= 50 ('amass') / t <= 100 ('prox'), gaussian_diffusion_posenet.py:461-477,625-626).  The guided steps
    differentiate through the body model's shape blend (575 MB of shaped vertices at B = 32): they are timed on `Bg` clips and scaled
    by B / Bg, which the record says.  The host glue between the stages (recover_from_repr_smpl / get_repr_smplx per sequence) is not
    priced (it would only lower the figure)."""
    from oracle import diffusion as odiff
    from oracle import geometry as G
    from oracle import nets
    from rohm_amd.utils import synth
    cores = min(usable_cores(), 64)
    torch.set_num_threads(cores)
    g = torch.Generator().manual_seed(0)
    t_end = time.perf_counter() + budget_s

    def timed(fn, reps, share):
        """median seconds of fn() over up to `reps` runs within `share` of what is left of the budget (always >= 1 run)"""
        limit = time.perf_counter() + max(0.5, (t_end - time.perf_counter()) * share)
        ts = []
        fn()
        while len(ts) < reps:
            t0 = time.perf_counter()
            fn()
            ts.append(time.perf_counter() - t0)
            if time.perf_counter() > limit:
                break
        ts.sort()
        return ts[len(ts) // 2], len(ts)

    sd_p = synth.posenet_state_dict(0)
    mean, std = synth.synthetic_stats(1)
    m_t, s_t = torch.from_numpy(mean), torch.from_numpy(std)
    tab = odiff.tables(odiff.cosine_betas(1000))
    tab_t = odiff.tables(odiff.cosine_betas(100))
    parts, counts = {}, {}
    with torch.no_grad():
        # un-guided PoseNet step at the full batch
        x = torch.randn(B, 294, 1, 143, generator=g)
        cond = synth.plausible_motion(3, B, 143, mean, std)
        nz = [torch.randn(B, 294, 1, 143, generator=g)]
        fn_p = lambda xx, i: nets.posenet_forward(sd_p, xx, cond, torch.full((xx.shape[0],), i, dtype=torch.int64))
        parts['posenet_step'], counts['posenet_step'] = timed(lambda: odiff.p_sample_loop(fn_p, x, nz, tab, [500]), 5, 0.25)
        if workload in ('scheme', 'egobody'):
            sd_t, sd_c = synth.trajnet_state_dict(1, trajcontrol=False), synth.trajnet_state_dict(2, trajcontrol=True)
            xt, ct = torch.randn(B, 144, 13, generator=g), torch.randn(B, 144, 13, generator=g)
            cc = torch.randn(B, 144, 272, generator=g)
            nzt = [torch.randn(B, 144, 13, generator=g)]
            tt = lambda i: torch.full((B,), i, dtype=torch.int64)
            f_t = lambda xx, i: nets.trajnet_forward(sd_t, xx, ct, tt(i))
            f_c = lambda xx, i: nets.trajnet_forward(sd_c, xx, ct, tt(i), control_cond=cc)
            parts['trajnet_step'], counts['trajnet_step'] = timed(lambda: odiff.p_sample_loop(f_t, xt, nzt, tab_t, [50]), 7, 0.15)
            parts['trajcontrol_step'], counts['trajcontrol_step'] = timed(lambda: odiff.p_sample_loop(f_c, xt, nzt, tab_t, [50]), 7, 0.2)
    # guided PoseNet step (autograd through the body model) on Bg clips, scaled to B
    Bg = min(B, 4)
    body = G.BodyModel(synth.synthetic_smplx_tensors(0))
    xg = synth.plausible_motion(5, Bg, 143, mean, std) + 0.05 * torch.randn(Bg, 294, 1, 143, generator=g)
    cg = synth.plausible_motion(6, Bg, 143, mean, std)
    nzg = [torch.randn(Bg, 294, 1, 143, generator=g)]
    fn_g = lambda xx, i: nets.posenet_forward(sd_p, xx, cg, torch.full((Bg,), i, dtype=torch.int64))
    guid = {'skating': lambda x0, i: G.guide_skating(x0, m_t, s_t, body)}
    gt, thr = ('amass', 50) if workload == 'scheme' else ('prox', 100)
    if gt == 'prox':
        cam = synth.synthetic_camera_batch(0, Bg)
        guid['2d'] = lambda x0, i: G.guide_2d_projection(x0, m_t, s_t, body, cam['transf_matrix'], cam['focal_length'], cam['camera_center'],
                                                         cam['keypoints_2d'], torch.tensor(synth.SYNTH_CAM_R), torch.tensor(synth.SYNTH_CAM_T))
    tg, ng = timed(lambda: odiff.p_sample_loop(fn_g, xg, nzg, tab, [thr - 10], guidance=guid, grad_type=gt), 3, 0.9)
    parts['posenet_guided_step'] = tg * B / Bg
    counts['posenet_guided_step'] = ng
    p, pg = parts['posenet_step'], parts['posenet_guided_step']
    if workload == 'scheme':         # test_amass_full.py:217-384: TrajNet 100, PoseNet 1000 (t <= 50 guided), TrajControl 100, PoseNet 1000
        plan = {'trajnet_step': 100, 'trajcontrol_step': 100, 'posenet_step': 2 * 949, 'posenet_guided_step': 2 * 51}
    elif workload == 'egobody':      # test_prox_egobody.py:214-324, sample_iter 3, early stop: 3 x (100 + 980), t = 100 .. 20 guided
        plan = {'trajnet_step': 100, 'trajcontrol_step': 200, 'posenet_step': 3 * 899, 'posenet_guided_step': 3 * 81}
    else:                            # prox: PoseNet 980 steps, t = 100 .. 20 guided
        plan = {'posenet_step': 899, 'posenet_guided_step': 81}
    sec = sum(parts[k] * n for k, n in plan.items())
    return {'value': B / sec, 'unit': 'clips/s', 'cores': cores, 'host_cpus': os.cpu_count(), 'cpu_model': cpu_model(), 'kind': 'port',
            'seconds_per_pass': round(sec, 1), 'step_ms': {k: round(v * 1e3, 1) for k, v in parts.items()}, 'step_counts': plan,
            'sample': (f'oracle port, B={B}: median step time of each kind of step ' + ', '.join(f'{k} x{counts[k]}' for k in parts) +
                       f'; guided steps ({gt}: autograd through the oracle body model) timed on {Bg} clips and scaled x{B // Bg}; pass = sum(count x '
                       f'step time), host glue between the stages not priced'),
            'torch_threads': torch.get_num_threads()}


class _TrajDataset(_Dataset):
    traj_feat_dim = 13


def scheme_bench(args, world, rank, dev, dist):
    """SURVEY.md §8(d) cfg 3: the drivers' whole inference-iteration loop (rohm_amd.inference), per GPU batch B."""
    import types
    from rohm_amd import _lib, sharding
    from rohm_amd.body_model import SMPLXLayer
    from rohm_amd.diffusion import gaussian_diffusion_posenet as gdp
    from rohm_amd.diffusion import gaussian_diffusion_trajnet as gdt
    from rohm_amd.diffusion.respace import SpacedDiffusionPoseNet, SpacedDiffusionTrajNet
    from rohm_amd.inference import run_amass_iterations
    from rohm_amd.model.posenet import PoseNet
    from rohm_amd.model.trajnet import TrajNet
    from rohm_amd.utils import synth
    from rohm_amd.utils.model_util import create_gaussian_diffusion
    B, S = args.batch, args.ddpm_steps
    body_t = synth.synthetic_smplx_tensors(0)
    layer = SMPLXLayer.from_tensors(body_t).to(dev)
    s_traj, s_pose = synth.synthetic_stats(0), synth.synthetic_stats(1)
    if args.workload == 'egobody':
        # The 2-D re-projection term divides by the camera-space depth of joints recovered from the NETWORKS' outputs
        # (random weights here): keep the synthetic body ~4 m in front of the camera whatever they predict, as a real
        # recording does (SURVEY.md §8(d) cfg 4: z ~ 3 m) -- translation channels 16..18 get a small spread around z = 4.
        s_traj = (s_traj[0].copy(), s_traj[1].copy())
        s_pose = (s_pose[0].copy(), s_pose[1].copy())
        for m, sd in (s_traj, s_pose):
            m[16:19] = (0.0, 0.0, 4.0)
            sd[16:19] = 0.15
    tds, pds = _TrajDataset(), _Dataset()
    tds.Mean, tds.Std = s_traj
    pds.Mean, pds.Std = s_pose
    pnet = PoseNet(pds, 294, latent_dim=512, ff_size=1024, num_layers=8, num_heads=4, traj_feat_dim=22,
                   body_model_path=layer, device=dev)
    pnet.load_state_dict(synth.posenet_state_dict(0), strict=False)
    pnet = pnet.to(dev).eval()
    nets = {'posenet': pnet}
    for name, ctrl in (('trajnet', False), ('trajnet_control', True)):
        n = TrajNet(time_dim=32, mid_dim=512, cond_dim=13, traj_feat_dim=13, trajcontrol=ctrl, device=dev)
        n.load_state_dict(synth.trajnet_state_dict(1 + ctrl, trajcontrol=ctrl), strict=True)
        nets[name] = n.to(dev).eval()
    diffs = {'posenet': create_gaussian_diffusion(_Args, gdp, SpacedDiffusionPoseNet, S, '', device=dev),
             'trajnet': create_gaussian_diffusion(_Args, gdt, SpacedDiffusionTrajNet, 100, '', device=dev),
             'trajnet_control': create_gaussian_diffusion(_Args, gdt, SpacedDiffusionTrajNet, 100, '', device=dev)}
    ego = args.workload == 'egobody'
    if args.guidance_semantics == 'global' and dist is not None:
        # the reference at the GLOBAL batch (model/posenet.py:231,243,309 normalise over every clip of the batch): mask counts
        # all-reduced per guided step, the 2-D term scaled by B_local / B_global (known here: no collective for it)
        sharding.use_global_batch_guidance(pnet, True, global_batch=world * B)
    abs_ch = [0, 2, 3, 6, 7, 8, 9, 10, 11, 12, 16, 17, 18]
    clean_t = synth.walking_motion(1000 + rank, B, 144, *s_traj, body_t).to(dev)
    clean_p = synth.walking_motion(1000 + rank, B, 144, *s_pose, body_t).to(dev)
    torch.manual_seed(rank)
    noisy_t, noisy_p = clean_t + 0.05 * torch.randn_like(clean_t), clean_p + 0.05 * torch.randn_like(clean_p)
    if ego:
        # SURVEY.md §8(d) cfg 5 (test_prox_egobody.py): sample_iter=3 (--sample_iter, :56), iterations >= 1 through
        # TrajControl (:230-242), per-channel visibility mask ~80 % visible (:291-294), PROX guidance, early stop
        from rohm_amd.inference import run_prox_iterations
        sargs = types.SimpleNamespace(sample_iter=3, repr_abs_only=True, iter2_cond_noisy_traj=False,
                                      iter2_cond_noisy_pose=False, early_stop=True, cond_fn_with_grad=True,
                                      timestep_respacing_eval='')
        pds.cam_R, pds.cam_t = torch.tensor(synth.SYNTH_CAM_R), torch.tensor(synth.SYNTH_CAM_T)
        pds.joints_num = 22
        cam = {k: v.to(dev) for k, v in synth.synthetic_camera_batch(rank, B).items()}
        gvis = torch.Generator().manual_seed(77 + rank)
        joint_vis = (torch.rand(B, 145, 22, generator=gvis) < 0.8).float()
        vec = torch.ones(B, 145, 294)
        for j in range(22):                                  # joint visibility -> channel visibility (local pos / vel / 6-D)
            vec[:, :, 22 + 3 * j:25 + 3 * j] = joint_vis[:, :, j:j + 1]
            vec[:, :, 88 + 3 * j:91 + 3 * j] = joint_vis[:, :, j:j + 1]
            if j > 0:
                vec[:, :, 154 + 6 * (j - 1):160 + 6 * (j - 1)] = joint_vis[:, :, j:j + 1]
        mask_vec_vis = vec.to(dev)
    else:
        sargs = types.SimpleNamespace(sample_iter=2, repr_abs_only=True, infill_traj=False, traj_mask_ratio=0.1,
                                      mask_scheme='lower', input_noise=True, iter2_cond_noisy_traj=True,
                                      iter2_cond_noisy_pose=True, early_stop=False, cond_fn_with_grad=True,
                                      timestep_respacing_eval='')

    def batches():
        bt = {'cond': noisy_t[:, :, abs_ch].contiguous(), 'motion_repr_clean': clean_t.clone(),
              'motion_repr_noisy': noisy_t.clone()}
        bp = {'motion_repr_clean': clean_p.clone(), 'motion_repr_noisy': noisy_p.clone()}
        if ego:
            bp.update(cam)
            bp['mask_vec_vis'] = mask_vec_vis
        return bt, bp

    def one_pass():
        gather.drain()
        bt, bp = batches()
        run = run_prox_iterations if ego else run_amass_iterations
        pose, _, _ = run(sargs, nets, diffs, bt, bp, tds, pds, layer)
        gather.submit(pose)
        return pose

    gather = ResultGather(world * B, dist)
    elapsed, out, prof = timed_region(one_pass, args, world, dev, dist, drain=gather.drain)
    finite = bool(torch.isfinite(out).all())
    assert finite, 'non-finite samples'
    report = rank_report(args, world, rank, dev, dist, gather, pnet)
    if rank == 0:
        clips = world * B * args.steps
        all_ms = sum(v['total_ms'] for v in prof.values())
        gemm = {k: v for k, v in prof.items() if k.startswith('gemm_') or k.startswith('conv_gemm')}
        g_ms, g_fl = sum(v['total_ms'] for v in gemm.values()), sum(v['flops'] for v in gemm.values())
        g_n = sum(v['launches'] for v in gemm.values())
        achieved = g_fl / (g_ms * 1e-3) / 1e12 if g_ms > 0 else None
        kernels = {k: {'launches': v['launches'], 'avg_us': round(v['total_ms'] / v['launches'] * 1e3, 2),
                       'time_share': round(v['total_ms'] / all_ms, 4)}
                   for k, v in sorted(prof.items(), key=lambda kv: -kv[1]['total_ms'])}
        metric, wl = scheme_names(args.workload, B, S)
        rec = {'metric': metric,
               'value': clips / elapsed, 'unit': 'clips/s', 'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup,
               'ms_per_step': elapsed / args.steps * 1e3, 'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None,
               'dtype': 'f32', 'data': 'synthetic',
               'config': {'workload': wl, 'clips_per_gpu': B, 'ddpm_steps': S, 'finite_output': finite,
                          'output_digest': [float(out.double().sum()), float(out.double().abs().sum())],      # rank 0's clips
                          'sharding': sharding_note(args, world, B, dist), 'cpu_binding': args.cpu_binding},
               'roofline': {'kernel': 'gemm_f32_kernel (all fp32-MFMA GEMM / conv-GEMM launches, sampled)', 'bound': 'mfma',
                            'achieved': achieved, 'peak': PEAK_F32_MFMA_TFLOPS, 'unit': 'TFLOP/s',
                            'frac': achieved / PEAK_F32_MFMA_TFLOPS if achieved else None, 'traffic': None,
                            'launches_timed': g_n, 'gemm_time_share_of_kernels': g_ms / all_ms if all_ms else None,
                            'kernels': kernels}}
        rec['rank_report'] = report
        if world == 1 and (not args.no_cpu_baseline or args.config_cpu_baseline):
            try:
                rec['cpu_baseline'] = cpu_baseline_config(args.workload, B)
            except Exception as e:      # a side measurement never costs the record
                rec['cpu_baseline'] = {'error': f'{type(e).__name__}: {e}'}
        print(json.dumps(rec), flush=True)
    finish(world, dist)


def scheme_names(workload, B, S):
    """(metric, config.workload) of the scheme / egobody workloads -- one place, so that the launcher self-test names what the
    measuring path would name."""
    if workload == 'egobody':
        return ((f'denoised 145-frame clips/sec, 3-iteration PROX/EgoBody scheme (TrajNet 100 + 2 x TrajControl 100 + '
                 f'3 x PoseNet 980 of {S} steps, 2-D + skating guidance on t<=100)'),
                (f'EgoBody scheme [BASELINE.json configs[4] / SURVEY cfg 5], batch={B} clips per GPU, sample_iter=3, '
                 f'visibility mask 80 % visible, early stop, guidance weights as the reference (3e5 / 1e5)'))
    return ((f'denoised 145-frame clips/sec, full 2-iteration RoHM scheme (TrajNet 100 + PoseNet {S} '
             f'+ TrajControl 100 + PoseNet {S} steps, skating guidance on t<=50)'),
            (f'full scheme [BASELINE.json configs[2] / SURVEY cfg 3], batch={B} clips per GPU, '
             f'sample_iter=2, mask_scheme=lower, guidance weights as the reference (3e6)'))


def sharding_note(args, world, B, dist):
    """What `config.sharding` says: the split and WHICH guidance semantics the run had."""
    base = f'{world} x {B} independent clips, all-gather of results only' if world > 1 else 'single GPU'
    if args.workload == 'posenet':
        return base
    if args.guidance_semantics == 'global' and dist is not None:
        return base + (f'; guidance = GLOBAL-batch semantics (the reference at batch {world * B}: skating mask counts all-reduced '
                       'per guided step, 2-D term scaled by B_local / B_global)')
    return base + f'; guidance = replica semantics (each rank = the reference at batch {B}, no communication)'


def run_child(argv, env_extra=None, timeout=600):
    """One more measurement as a child process of this script (own process: own handles, own failure domain); returns its
    JSON record, or {'error': ...} -- a failing extra never costs the headline line."""
    import subprocess
    env = dict(os.environ)
    for k in ('RANK', 'LOCAL_RANK', 'WORLD_SIZE', 'MASTER_ADDR', 'MASTER_PORT', 'ROHM_GEMM_PRECISION'):
        env.pop(k, None)
    env.update(env_extra or {})
    cmd = [sys.executable, os.path.abspath(__file__), '--gpus', '1', '--no-extras', '--no-cpu-baseline'] + list(argv)
    t0 = time.perf_counter()
    try:
        r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=timeout)
    except subprocess.TimeoutExpired:
        return {'error': f'timeout after {timeout} s', 'argv': list(argv)}
    lines = [ln for ln in r.stdout.strip().splitlines() if ln.startswith('{')]
    if r.returncode != 0 or not lines:
        return {'error': f'rc={r.returncode}', 'stderr_tail': r.stderr[-400:], 'argv': list(argv)}
    try:
        d = json.loads(lines[-1])
    except ValueError as e:
        return {'error': f'bad JSON: {e}', 'argv': list(argv)}
    d['child_wall_s'] = round(time.perf_counter() - t0, 1)
    return d


def brief(d):
    """The fields of a child's record that the parent line carries."""
    if 'error' in d:
        return d
    rf = d.get('roofline') or {}
    out = {'value': d['value'], 'unit': d['unit'], 'ms_per_pass': d['ms_per_step'], 'passes': d['steps'], 'warmup': d['warmup'],
           'dtype': d['dtype'], 'metric': d['metric'], 'workload': d['config']['workload'],
           'finite_output': d['config'].get('finite_output', True),
           'gemm_roofline': {'achieved': rf.get('achieved'), 'peak': rf.get('peak'), 'frac': rf.get('frac'), 'unit': rf.get('unit'),
                             'time_share_of_kernels': rf.get('gemm_time_share_of_kernels')},
           'child_wall_s': d.get('child_wall_s')}
    if d.get('cpu_baseline'):
        out['cpu_baseline'] = d['cpu_baseline']
    if rf.get('attention'):
        out['attention_roofline'] = {k: rf['attention'].get(k) for k in ('achieved', 'frac', 'avg_launch_us', 'share_of_launch',
                                                                           'frac_incl_meeting')}
        out['attention_roofline']['measured'] = ('inside the encoder-stack launch (phase stamps)' if 'share_of_launch' in rf['attention']
                                                 else 'attention kernel launches')
    return out


def extras(args, budget_s=420.0):
    """`second_line` and `configs` of the N = 1 record (see the module docstring); headline first, these afterwards.  All children
    together get `budget_s` of wall time: one that does not fit any more is reported as skipped, never waited for."""
    S = str(args.ddpm_steps)
    t_start = time.perf_counter()

    def child(argv, env=None):
        left = budget_s - (time.perf_counter() - t_start)
        if left < 20.0:
            return {'error': f'skipped: the extras budget of {budget_s:.0f} s is spent', 'argv': list(argv)}
        return run_child(argv, env, timeout=min(300.0, left))
    labels = {
        'fp16x3': ('opt-in ROHM_GEMM_PRECISION=fp16x3: every fp32 product of the four encoder Linears emulated by three fp16 MFMA products of '
                   'two fp16 planes (h = fp16(x), l = fp16((x - h) 2^11); cross terms in a second accumulator of weight 2^-11), fp32 '
                   'accumulation; ~2^-22 per product; LayerNorm folded into the GEMMs around it (ROHM_PP_LNFOLD, default on in the two-plane modes); '
                   'held to the fp32 parity bars by tests/test_gpu_precision_ladder.py (whole PoseNet suite); NEVER the headline (narrower '
                   'arithmetic than the reference)'),
        'bf16x6': ('opt-in ROHM_GEMM_PRECISION=bf16x6: six bf16 MFMA products of three exact truncation planes per fp32 product, fp32 '
                   'accumulation; held to the fp32 parity bars by the same suite; NEVER the headline'),
    }

    def line(mode):
        sl = child(['--workload', 'posenet', '--batch', str(args.batch), '--ddpm-steps', S, '--steps', '2', '--warmup', '1',
                    '--with-accuracy'], {'ROHM_GEMM_PRECISION': mode})
        out = brief(sl)
        if 'error' not in sl:
            out['roofline'] = {k: sl['roofline'].get(k) for k in ('kernel', 'bound', 'achieved', 'peak', 'unit', 'frac', 'peak_note',
                                                                   'launches_timed', 'avg_launch_us')}
            out['accuracy'] = sl.get('accuracy')
            out['label'] = labels[mode]
        out['mode'] = mode
        return out
    second = line('fp16x3')
    second['also'] = {'bf16x6': {k: v for k, v in line('bf16x6').items() if k in ('value', 'unit', 'ms_per_pass', 'dtype', 'accuracy', 'error',
                                                                                  'label', 'gemm_roofline')}}
    cfg = {}
    for key, argv in (('b32', ['--workload', 'posenet', '--batch', '32']), ('scheme_b32', ['--workload', 'scheme', '--batch', '32']),
                      ('prox_b32', ['--workload', 'prox', '--batch', '32']), ('egobody_b32', ['--workload', 'egobody', '--batch', '32'])):
        cfg[key] = brief(child(argv + ['--ddpm-steps', S, '--steps', '2', '--warmup', '1'] +
                               ([] if getattr(args, 'no_cpu_baseline', False) else ['--config-cpu-baseline'])))
    return second, cfg


def standalone_attention(B, dev, reps=40):
    """attention_f32_kernel on its own (rohm_attention_f32: the launch shape of the launch-per-GEMM path, random q / k / v of the
    workload's shape), timed with events on torch's current stream, which is the stream the call launches on."""
    from rohm_amd import ops
    try:
        qkv = torch.randn(B * 144, 1536, device=dev)
        for _ in range(5):
            ops.attention(qkv, B, 4)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            ops.attention(qkv, B, 4)
        e1.record()
        torch.cuda.synchronize(dev)
        us = e0.elapsed_time(e1) / reps * 1e3
        tf = 4.0 * 144 * 144 * 128 * B * 4 / (us * 1e-6) / 1e12
        return {'achieved': tf, 'frac': tf / PEAK_F32_MFMA_TFLOPS, 'avg_launch_us': us, 'unit': 'TFLOP/s', 'peak': PEAK_F32_MFMA_TFLOPS,
                'note': 'inside the timed region attention runs within the encoder-stack launch (gemm_stack); this is the same work item as '
                        f'its own launch, {reps} back-to-back launches after the timed region'}
    except Exception as e:      # never lose the record over a side measurement
        return {'error': f'{type(e).__name__}: {e}'}


def in_stack_attention(net, B, dev):
    """The attention phase INSIDE the encoder-stack launch, measured by the kernel's own phase stamps right after the timed region
    (rohm_amd/stack_timeline.py: lane 0 of every workgroup writes the 100 MHz wall clock at every seam of a few stamped launches).
    Falls back to the stand-alone kernel where no stack launch exists (B < 32, plane modes, a refused layout)."""
    try:
        from rohm_amd import stack_timeline
        rec = stack_timeline.measure(net, B, reps=4, device=dev)
        if 'error' in rec:
            raise RuntimeError(rec['error'])
        out = dict(rec['attention_in_stack'])
        out['avg_launch_us'] = out['us_per_launch']
        out['stack_phases'] = rec['phases']
        out['stack_launch_span_us'] = rec['launch_span_us']
        out['meetings_share_of_launch'] = rec['meetings_share_of_launch']
        return out
    except Exception as e:      # diagnostics must never cost the record
        sa = standalone_attention(B, dev)
        sa['in_stack_error'] = repr(e)[:200]
        return sa


def exchange_note(net):
    """Which launch forms the PoseNet handle ended the run with (include/rohm_hip.h rohm_posenet_exchange_mode)."""
    nat = getattr(net, '_native', None)
    if nat is None:
        return None
    m = nat.exchange_mode
    return {'layernorm_in_gemm': bool(m & 1), 'stream_k_head': bool(m & 2), 'refused_by_layout_guard': bool(m & 4),
            'fell_back_after_failed_exchange': bool(m & 8), 'gemm_chain': bool(m & 16), 'encoder_stack': bool(m & 32),
            'guard': nat.exchange_guard}


def free_port():
    import socket
    with socket.socket() as so:
        so.bind(('127.0.0.1', 0))
        return so.getsockname()[1]


def spawn_ranks(n, argv, env=None):
    """Re-execute this script as `n` ranks of one node under torch.distributed.run (127.0.0.1 rendezvous) and return
    the launcher's exit code.  Used when `--gpus N` (N > 1) is given without a WORLD_SIZE in the environment."""
    import subprocess
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', f'--nproc-per-node={n}',
           '--master-addr', '127.0.0.1', '--master-port', str(free_port()), os.path.abspath(__file__)] + list(argv)
    e = dict(os.environ if env is None else env)
    e.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')      # dmabuf IPC: RCCL needs it on this driver
    e.setdefault('OMP_NUM_THREADS', '8')
    return subprocess.call(cmd, env=e)


def parse_cpulist(text):
    """'0-3,8,10-11' -> [0, 1, 2, 3, 8, 10, 11] (sysfs cpulist format)."""
    out = []
    for part in text.strip().split(','):
        if not part:
            continue
        lo, _, hi = part.partition('-')
        out.extend(range(int(lo), int(hi or lo) + 1))
    return out


def plan_rank_cpus(local_rank, local_world, allowed, node_of_rank=None, cpus_of_node=None):
    """Which CPUs rank `local_rank` of `local_world` ranks on this host should run on.  `allowed`: this process's affinity mask.
    `node_of_rank(r)`: NUMA node of rank r's GPU (None / negative = unknown); `cpus_of_node(n)`: that node's CPUs.  Ranks whose GPUs
    hang off the same node split that node's allowed CPUs evenly (contiguous blocks, in rank order); a rank whose node is unknown,
    or has no allowed CPU, takes its block of an even split of the whole mask.  Never returns an empty list: with more ranks than
    CPUs ranks share.  Pure function: tests/test_bench_launcher.py drives it with made-up topologies."""
    allowed = sorted(allowed)

    def block(cpus, idx, parts):
        if not cpus:
            return []
        if len(cpus) < parts:
            return [cpus[idx % len(cpus)]]
        base, extra = divmod(len(cpus), parts)
        lo = idx * base + min(idx, extra)
        return cpus[lo:lo + base + (1 if idx < extra else 0)]
    nodes = [None] * local_world
    if node_of_rank is not None and cpus_of_node is not None:
        for r in range(local_world):
            try:
                n = node_of_rank(r)
            except Exception:
                n = None
            nodes[r] = n if (n is not None and n >= 0) else None
    mine = nodes[local_rank]
    if mine is not None:
        try:
            node_cpus = sorted(set(cpus_of_node(mine)) & set(allowed))
        except Exception:
            node_cpus = []
        peers = [r for r in range(local_world) if nodes[r] == mine]
        got = block(node_cpus, peers.index(local_rank), len(peers))
        if got:
            return got, f'NUMA node {mine}: {len(got)} of its {len(node_cpus)} allowed CPUs ({len(peers)} rank(s) on the node)'
    got = block(allowed, local_rank, local_world)
    return got, f'even split of the {len(allowed)} allowed CPUs (GPU NUMA node unknown)'


def gpu_numa_node(index):
    """NUMA node of HIP device `index` from sysfs (PCI address of the device), or None."""
    try:
        p = torch.cuda.get_device_properties(index)
        addr = f'{p.pci_domain_id:04x}:{p.pci_bus_id:02x}:{p.pci_device_id:02x}.0'
        n = int(open(f'/sys/bus/pci/devices/{addr}/numa_node').read())
        return n if n >= 0 else None
    except Exception:
        return None


def node_cpus(node):
    return parse_cpulist(open(f'/sys/devices/system/node/node{node}/cpulist').read())


def bind_rank(local_rank, local_world, gpu=True):
    """Pin this rank to its share of the host's cores (the TrajNet loops are host-enqueue-bound, eight unpinned
    ranks would fight over whatever cores the OS hands out and wander between NUMA nodes).  Returns the note that goes into
    `config.cpu_binding`.  ROHM_BENCH_NO_BIND=1 leaves the affinity alone."""
    if os.environ.get('ROHM_BENCH_NO_BIND') == '1' or not hasattr(os, 'sched_setaffinity'):
        return 'not bound'
    allowed = sorted(os.sched_getaffinity(0))
    cpus, how = plan_rank_cpus(local_rank, local_world, allowed, gpu_numa_node if gpu else None, node_cpus if gpu else None)
    try:
        os.sched_setaffinity(0, cpus)
    except OSError as e:
        return f'not bound ({e})'
    torch.set_num_threads(max(1, min(len(cpus), 8)))
    return f'rank bound to {len(cpus)} CPU(s) [{cpus[0]}..{cpus[-1]}]: {how}'


def init_ranks(args):
    """(world, rank, device, dist): one process per GPU; `--gpus` must agree with the launched world size."""
    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    args.cpu_binding = 'single rank: not bound'
    if world > 1 or args.force_dist:      # (--force-dist: the multi-GPU plumbing at world size 1, the binding included)
        args.cpu_binding = bind_rank(local_rank, int(os.environ.get('LOCAL_WORLD_SIZE', world)), gpu=args.backend != 'gloo')
    if world != args.gpus:
        raise SystemExit(f'bench.py: --gpus {args.gpus} but WORLD_SIZE={world}: launch with '
                         f'--nproc-per-node {args.gpus} (or drop WORLD_SIZE and let --gpus spawn the ranks)')
    selftest = args.backend == 'gloo'
    if selftest:
        if os.environ.get('ROHM_BENCH_SELFTEST') != '1':
            raise SystemExit('bench.py: --backend gloo is the CPU self-test of the launcher / timing harness only '
                             '(set ROHM_BENCH_SELFTEST=1); the hot path has no CPU fallback')
        dev = torch.device('cpu')
    else:
        if not torch.cuda.is_available():
            raise SystemExit('bench.py needs an AMD GPU: the hot path has no CPU fallback')
        if torch.cuda.device_count() <= local_rank:
            raise SystemExit(f'bench.py: rank {rank} wants GPU {local_rank} but only {torch.cuda.device_count()} visible')
        torch.cuda.set_device(local_rank)
        dev = torch.device('cuda', local_rank)
    dist = None
    if world > 1 or args.force_dist:
        import torch.distributed as dist
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        if 'MASTER_PORT' not in os.environ:        # bare `--force-dist` start without a launcher
            os.environ['MASTER_PORT'] = str(free_port())
        os.environ.setdefault('RANK', '0')
        os.environ.setdefault('WORLD_SIZE', '1')
        os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
        if selftest:
            dist.init_process_group('gloo')
        else:
            dist.init_process_group('nccl', device_id=dev)       # RCCL over xGMI
    return world, rank, dev, dist


def timed_region(one_pass, args, world, dev, dist, profile=True, drain=None):
    """W untimed warm-up passes, then EXACTLY K passes bracketed by barrier + device synchronize on both sides; the
    elapsed time is the MAX over ranks.  `drain`: completes what `one_pass` left in flight (the asynchronous result all-gather of
    the last pass) -- called inside the bracket, so the timed region contains every collective it started.
    Returns (elapsed_s, last_output, profiler_dict)."""
    def sync():
        if drain is not None:
            drain()
        if dist is not None:
            dist.barrier()
        if dev.type == 'cuda':
            torch.cuda.synchronize(dev)
    out = None
    for _ in range(args.warmup):
        out = one_pass()
    sync()
    prof = {}
    if profile and dev.type == 'cuda':
        from rohm_amd import _lib
        _lib.profile_start(args.profile_stride)
    gobj = getattr(drain, '__self__', None)          # the ResultGather whose collective waits are NOT this rank's own time
    drained0 = getattr(gobj, 'drain_s', 0.0)
    t0 = time.perf_counter()
    enqueue = []
    for _ in range(args.steps):
        tp, d0 = time.perf_counter(), getattr(gobj, 'drain_s', 0.0)
        out = one_pass()
        enqueue.append(time.perf_counter() - tp - (getattr(gobj, 'drain_s', 0.0) - d0))
    t_loop = time.perf_counter() - t0
    if dev.type == 'cuda':      # this rank's OWN compute: an event on the compute stream (a device-wide synchronize would also wait
        ev = torch.cuda.Event()      # for the all-gather in flight on RCCL's stream, i.e. for the slowest peer)
        ev.record()
        ev.synchronize()
    own_done = time.perf_counter() - t0
    own = own_done - (getattr(gobj, 'drain_s', 0.0) - drained0)          # ... minus the time its passes spent inside collectives
    sync()
    elapsed = time.perf_counter() - t0
    if profile and dev.type == 'cuda':
        prof = _lib.profile_stop()
    # what this rank saw (rank_report gathers it): its own time for the K passes, how long the host needed to enqueue a pass (the
    # TrajNet loops are host-bound: eight ranks on shared cores show up here first) and how long it then waited for the slowest rank
    args.rank_timing = {'own_ms_per_step': own / args.steps * 1e3, 'host_enqueue_ms_per_step': sum(enqueue) / args.steps * 1e3,
                        'host_loop_ms_per_step': t_loop / args.steps * 1e3, 'wait_for_slowest_ms': (elapsed - own) * 1e3}
    if dist is not None:
        tt = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())
    return elapsed, out, prof


def rank_report(args, world, rank, dev, dist, gather=None, net=None):
    """Per-rank diagnostics of a multi-rank run, gathered to rank 0 (None on the others; a one-entry list at world size 1): own
    ms_per_step (before the MAX over ranks), host enqueue time per pass, time spent draining the result all-gather, the wait for the
    slowest rank, which launch forms the PoseNet handle ended with and what the layout guard said, the CPU binding, the device.  The
    first 8-GPU run of this code has to be diagnosable from its one JSON line: which rank was slow, whether a rank fell back to the
    exchange-free launches, whether the host or the device was the straggler."""
    me = {'rank': rank, 'device': str(dev), **getattr(args, 'rank_timing', {}), 'cpu_binding': getattr(args, 'cpu_binding', None)}
    if gather is not None:
        me['allgather_drain_ms_per_step'] = gather.drain_s / max(1, gather.drains) * 1e3
        me['allgather_drains'] = gather.drains
    if net is not None:
        me['exchange_mode'] = exchange_note(net)
    if dev.type == 'cuda':
        try:
            me['gpu'] = torch.cuda.get_device_name(dev)
        except Exception:
            pass
    if dist is None or world == 1:
        ranks = [me]
    else:
        ranks = [None] * world
        dist.all_gather_object(ranks, me)
    if rank != 0:
        return None
    own = [r['own_ms_per_step'] for r in ranks if r and 'own_ms_per_step' in r]
    skew = ({'max_own_ms_per_step': max(own), 'min_own_ms_per_step': min(own), 'skew_frac': (max(own) - min(own)) / max(own),
             'slowest_rank': int(max(range(len(own)), key=lambda i: own[i]))} if own else None)
    return {'ranks': ranks, 'skew': skew}


class ResultGather:
    """The path's only exchange -- the finished clips of a pass, one all-gather -- issued asynchronously on the backend's stream:
    the host goes on preparing the next pass while the collective waits for the slowest rank.  It is completed (`drain`) BEFORE the
    next pass enqueues compute -- an RCCL kernel that spins on its CUs while it waits for a late peer would otherwise take CUs away
    from the encoder-stack launch, whose 256 one-per-CU workgroups would then need two rounds -- and by `drain()` inside the timed
    bracket at the end."""

    def __init__(self, n_total, dist):
        self.n_total, self.dist, self.pending, self.last = n_total, dist, None, None
        self.drain_s, self.drains = 0.0, 0          # host time spent completing the collective (rank_report)

    def submit(self, x0):
        from rohm_amd import sharding
        if self.dist is None:
            return
        self.drain()
        self.pending = sharding.gather_clips(x0, self.n_total, force=True, async_op=True)

    def drain(self):
        if self.pending is not None:
            t0 = time.perf_counter()
            self.last = self.pending.result()
            self.pending = None
            self.drain_s += time.perf_counter() - t0
            self.drains += 1


def finish(world, dist):
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


def selftest_bench(args, world, rank, dev, dist):
    """CPU (gloo) self-test of THIS file's multi-rank plumbing -- spawn, rank binding, barrier-bracketed timing, MAX over
    ranks, result all-gather, one JSON line from rank 0 -- with a stand-in for the sampler.  It measures nothing and says
    so in every field; tests/test_bench_launcher.py drives it."""
    from rohm_amd import sharding
    B = args.batch

    gather = ResultGather(world * B, dist if world > 1 else None)

    def one_pass():
        gather.drain()
        x0 = torch.full((B, 4), float(rank))
        time.sleep(0.01 * (rank + 1))
        gather.submit(x0)
        return x0
    elapsed, out, _ = timed_region(one_pass, args, world, dev, dist, profile=False, drain=gather.drain)
    if world > 1:
        out = gather.last
    report = rank_report(args, world, rank, dev, dist if world > 1 else None, gather)
    if rank == 0:
        ranks_seen = sorted(set(int(v) for v in out[:, 0].tolist()))
        # what the measuring path would call this launch (same helper functions), so that a dry run of e.g.
        # `--workload scheme --gpus 8 --batch 32` shows the record naming BASELINE.json configs[2] before any hardware run does
        stands_for = (scheme_names(args.workload, B, args.ddpm_steps) if args.workload in ('scheme', 'egobody') else
                      (None, f'{args.workload} [see main()]'))
        print(json.dumps({'metric': 'bench launcher self-test (NOT a measurement: stub sampler on CPU/gloo)', 'value': 0.0,
                          'unit': 'none', 'n_gpus': world, 'world_size': world, 'backend': 'gloo', 'steps': args.steps,
                          'warmup': args.warmup, 'ms_per_step': elapsed / args.steps * 1e3, 'data': 'selftest-stub',
                          'gathered_clips': int(out.shape[0]), 'ranks_seen': ranks_seen, 'rank_report': report,
                          'config': {'workload': 'selftest', 'stands_for_metric': stands_for[0], 'stands_for_workload': stands_for[1],
                                     'clips_per_gpu': B, 'sharding': sharding_note(args, world, B, dist),
                                     'cpu_binding': args.cpu_binding}}), flush=True)
    finish(world, dist)


def main(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=2)
    ap.add_argument('--warmup', type=int, default=1)
    ap.add_argument('--batch', type=int, default=64, help='clips per GPU')
    ap.add_argument('--ddpm-steps', type=int, default=1000)
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--config-cpu-baseline', action='store_true',
                    help='attach the CPU baseline of THIS workload even under --no-cpu-baseline (what the parent record asks of its configs children)')
    ap.add_argument('--backend', choices=['nccl', 'gloo'], default='nccl',
                    help="'nccl' (= RCCL on ROCm) is the only measuring backend; 'gloo' runs the launcher self-test")
    ap.add_argument('--workload', choices=['posenet', 'scheme', 'prox', 'egobody'], default='posenet',
                    help="'posenet' = BASELINE.json configs[1] (the headline metric); 'scheme' = configs[2]: the full "
                         "two-iteration RoHM scheme per clip (TrajNet 100 -> PoseNet 1000 + skating guidance -> "
                         "TrajControl 100 -> PoseNet 1000 + skating guidance); 'prox' = configs[3]: PoseNet with the PROX "
                         "test-time guidance (2-D re-projection + skating on t <= 100, early stop at 980 steps); 'egobody' = "
                         "configs[4]: the PROX/EgoBody driver loop (run_prox_iterations) with sample_iter=3, a random 80 %% "
                         "visibility mask and PROX guidance; all but 'posenet' are extra measurements, not the headline")
    ap.add_argument('--profile-stride', type=int, default=100,
                    help='the in-stream HIP events of the roofline record bracket every launch of every N-th denoising step.  Two event '
                         'records around each of a step\'s 51 launches cost ~0.5 ms per sampled step: at N = 16 (rounds 1-4) that was 1 %% of '
                         'the headline value and 2.7 %% of the scheme workloads; at N = 100 it is 0.16 %% and still times 160 launches of the '
                         'dominant kernel per 1000-step pass')
    ap.add_argument('--no-extras', action='store_true', help='N = 1 only: skip the second_line / configs child measurements')
    ap.add_argument('--with-accuracy', action='store_true', help='attach the 1000-step accuracy record even without the CPU leg')
    ap.add_argument('--guidance-semantics', choices=['replica', 'global'], default='replica',
                    help="guided workloads under clip sharding (SURVEY.md §8(e)): 'replica' = every rank behaves like the reference at "
                         "its LOCAL batch (no communication; default), 'global' = the reference at the GLOBAL batch (model/posenet.py:"
                         "231,243,309): one 8-byte all-reduce of the skating mask counts per guided step")
    ap.add_argument('--force-dist', action='store_true',
                    help='initialise the RCCL process group even at world size 1 and run barrier / all-reduce / all-gather '
                         'through it (the multi-GPU plumbing on a 1-GPU box)')
    argv = sys.argv[1:] if argv is None else list(argv)
    args = ap.parse_args(argv)
    if args.gpus < 1:
        raise SystemExit('bench.py: --gpus must be >= 1')
    if args.gpus > 1 and 'WORLD_SIZE' not in os.environ:
        # not launched by torch.distributed.run: become the launcher (one rank per GPU)
        if args.backend == 'nccl' and torch.cuda.device_count() < args.gpus:
            raise SystemExit(f'bench.py: --gpus {args.gpus} but only {torch.cuda.device_count()} GPU(s) visible')
        raise SystemExit(spawn_ranks(args.gpus, argv))
    world, rank, dev, dist = init_ranks(args)
    if args.backend == 'gloo':
        return selftest_bench(args, world, rank, dev, dist)

    from rohm_amd import _lib
    from rohm_amd.diffusion import gaussian_diffusion_posenet as gdp
    from rohm_amd.diffusion.respace import SpacedDiffusionPoseNet
    from rohm_amd.model.posenet import PoseNet
    from rohm_amd.utils import synth
    from rohm_amd.utils.model_util import create_gaussian_diffusion

    B, S = args.batch, args.ddpm_steps
    if args.workload in ('scheme', 'egobody'):
        return scheme_bench(args, world, rank, dev, dist)
    prox = args.workload == 'prox'
    ds = _Dataset()
    body = torch.nn.Identity()
    if prox:       # SURVEY.md §8(d) cfg 4: guidance needs a body model, dataset statistics and the camera
        from rohm_amd.body_model import SMPLXLayer
        body = SMPLXLayer.from_tensors(synth.synthetic_smplx_tensors(0)).to(dev)
        ds.Mean, ds.Std = synth.synthetic_stats(1)
        ds.cam_R, ds.cam_t = torch.tensor(synth.SYNTH_CAM_R), torch.tensor(synth.SYNTH_CAM_T)
    net = PoseNet(ds, 294, latent_dim=512, ff_size=1024, num_layers=8, num_heads=4, traj_feat_dim=22,
                  body_model_path=body, device=dev)
    net.load_state_dict(synth.posenet_state_dict(0), strict=not prox)
    net = net.to(dev).eval()
    from rohm_amd import sharding
    if prox and args.guidance_semantics == 'global' and dist is not None:
        sharding.use_global_batch_guidance(net, True, global_batch=world * B)
    diffusion = create_gaussian_diffusion(_Args, gdp, SpacedDiffusionPoseNet, S, '', device=dev)
    cond = synthetic_cond(B, dev, seed=1000 + rank)
    extra = {}
    if prox:
        cond = synth.plausible_motion(1000 + rank, B, 143, ds.Mean, ds.Std).to(dev)
        extra = {k: v.to(dev) for k, v in synth.synthetic_camera_batch(rank, B).items()}
    torch.manual_seed(rank)

    def one_pass():
        gather.drain()
        batch = {'cond': cond, **extra}
        if prox:
            _, x0 = diffusion.eval_losses(model=net, batch=batch, shape=[B, 294, 1, 143], progress=False,
                                          clip_denoised=False, timestep_respacing='', cond_fn_with_grad=True,
                                          grad_type='prox', early_stop=True, compute_loss=False)
        else:
            _, x0 = diffusion.eval_losses(model=net, batch=batch, shape=[B, 294, 1, 143], progress=False,
                                          clip_denoised=False, timestep_respacing='', cond_fn_with_grad=False,
                                          compute_loss=False)
        gather.submit(x0)      # the path's only exchange: finished clips, one RCCL all-gather, overlapping the next pass
        return x0

    gather = ResultGather(world * B, dist)
    elapsed, out, prof = timed_region(one_pass, args, world, dev, dist, drain=gather.drain)
    finite = bool(torch.isfinite(out).all())
    assert finite, 'non-finite samples'
    report = rank_report(args, world, rank, dev, dist, gather, net)

    if rank == 0:
        clips = world * B * args.steps
        gemm = {k: v for k, v in prof.items() if k.startswith('gemm_')}
        g_ms = sum(v['total_ms'] for v in gemm.values())
        g_fl = sum(v['flops'] for v in gemm.values())
        g_n = sum(v['launches'] for v in gemm.values())
        all_ms = sum(v['total_ms'] for v in prof.values())
        achieved = g_fl / (g_ms * 1e-3) / 1e12 if g_ms > 0 else None
        dom = max((k for k in ('gemm_stack_tail', 'gemm_stack') if prof.get(k, {}).get('total_ms')), key=lambda k: prof[k]['total_ms'],
                  default=None)
        kernels = {k: {'launches': v['launches'], 'avg_us': round(v['total_ms'] / v['launches'] * 1e3, 2),
                       'tflops': round(v['flops'] / (v['total_ms'] * 1e-3) / 1e12, 2) if v['flops'] else None,
                       'gbps': round(v['bytes'] / (v['total_ms'] * 1e-3) / 1e9, 1),
                       'time_share': round(v['total_ms'] / all_ms, 4)}
                   for k, v in sorted(prof.items(), key=lambda kv: -kv[1]['total_ms'])}
        rec = {
            'metric': ('denoised 145-frame clips/sec, PROX guidance (2-D re-projection + skating on t <= 100), 980 of '
                       f'{S} DDPM steps (early stop)') if prox else
                      ('denoised 145-frame clips/sec @1000 DDPM steps' if S == 1000 else
                       f'denoised 145-frame clips/sec @{S} DDPM steps'),
            'value': clips / elapsed, 'unit': 'clips/s', 'n_gpus': world, 'world_size': world, 'steps': args.steps,
            'warmup': args.warmup, 'ms_per_step': elapsed / args.steps * 1e3, 'higher_is_better': True,
            'scaling': 'weak', 'vs_baseline': None,
            'dtype': 'f32' if not _PRODUCTS else f'f32 emulated with {_PRODUCTS} {_PLANE_TYPE} MFMA products per product, f32 accumulate '
                                                  f'(ROHM_GEMM_PRECISION={_PREC}, opt-in)',
            'data': 'synthetic',
            'config': {'workload': (f'PoseNet {S}-step DDPM with PROX test-time guidance, early stop at 980 steps, batch={B} '
                                    f'synthetic 145-frame clips per GPU [BASELINE.json configs[3]]') if prox else
                                   (f'PoseNet {S}-step DDPM (x0-pred, fixed-small var), batch={B} synthetic '
                                    f'145-frame clips per GPU (T=143 -> 144 tokens, d=512, 8 layers), no guidance '
                                    f'[BASELINE.json configs[1]]'),
                       'guidance': 'prox [BASELINE.json configs[3]]: synthetic camera + OpenPose-style keypoints, weights as '
                                   f'the reference (3e5 / 1e5); finite_output={finite}' if prox else 'none',
                       'clips_per_gpu': B, 'ddpm_steps': S, 'sharding': sharding_note(args, world, B, dist), 'cpu_binding': args.cpu_binding,
                       'weights': 'random (seed 0)', 'exchange_mode': exchange_note(net)},
            'model_tflops': clips * S * POSENET_GFLOP_PER_CLIP_STEP * 1e-3 / elapsed,
            # end to end against the fp32-MFMA peak: on the reference's flops, and on the flops the device EXECUTES (the cond half of
            # the input embedding is contracted once per sampling loop instead of once per step: 2 x 144 x 512 x 288 flop per clip-step less)
            'e2e_frac': clips * S * POSENET_GFLOP_PER_CLIP_STEP * 1e-3 / elapsed / PEAK_F32_MFMA_TFLOPS if not _PRODUCTS else None,
            'e2e_frac_executed': (clips * S * (POSENET_GFLOP_PER_CLIP_STEP - 2e-9 * 144 * 512 * 288) * 1e-3 / elapsed / PEAK_F32_MFMA_TFLOPS
                                  if not _PRODUCTS else None),
            'roofline': {
                'kernel': ('encoder_stack_kernel / gemm_f32_kernel<BN,EPI> (all fp32-MFMA GEMM launches of the timed region -- from round 5 the '
                           'whole encoder is ONE launch per step, attention inside --, ' if not _PRODUCTS else
                           f'gemm_pp_stream_kernel / gemm_pp_kernel on {_PLANE_TYPE} planes + the fp32 embed / output-head GEMMs (all GEMM launches '
                           'of the timed region, ') + f'sampled every {args.profile_stride}th denoising step)',
                'bound': 'mfma', 'achieved': achieved,
                'peak': PEAK_F32_MFMA_TFLOPS if not _PRODUCTS else PEAK_BF16_MFMA_TFLOPS / _PRODUCTS, 'unit': 'TFLOP/s',
                'frac': (achieved / (PEAK_F32_MFMA_TFLOPS if not _PRODUCTS else PEAK_BF16_MFMA_TFLOPS / _PRODUCTS))
                if achieved else None,
                'peak_note': ('fp32 MFMA at the nominal 2.4 GHz.  Under this kernel the chip does not hold 2.4 GHz: in-kernel cycle counter vs '
                              'wall clock gives 2.05-2.13 GHz in the main loop on random operands and 2.27-2.34 GHz on all-zero operands (same '
                              'binary; profiles/r3_gemm_timeline.txt, r3_gemm_timeline_zero.txt; host telemetry is static in this VF, '
                              'profiles/r3_power_sclk_bench.txt) -- a power-limited (DVFS) ceiling of ~135-140 TFLOP/s, of which the loops reach '
                              '~0.9') if not _PRODUCTS else
                             (f'{_PLANE_TYPE} MFMA peak (2.5 PFLOP/s at 2.4 GHz) / {_PRODUCTS} products = fp32-equivalent flops.  A bare stream of these MFMAs '
                              'on random operands sustains 1.63-2.0 PFLOP/s (1.7-2.0 GHz under load, profiles/r3_g_mfma_bf16_chain_probe.txt)'),
                'traffic': pmc_traffic()[0] if not _PRODUCTS else None, 'traffic_unit': 'bytes per launch (HBM side, PMC)',
                'traffic_source': pmc_traffic()[1],
                'traffic_note': 'ARCHIVED: from the most recent committed rocprofv3 PMC passes of this workload (separate, serialised runs of '
                                'the build named by traffic_source), not collected by this run',
                'mfma_busy_pmc': pmc_mfma(B),
                # north_star: "... as fraction of the attention/GEMM roofline": the attention kernel by the same event timing -- or,
                # when attention runs INSIDE the encoder-stack launch (no launch of its own to time), the same kernel launched on its
                # own right after the timed region
                'attention': ({'achieved': prof['attention']['flops'] / (prof['attention']['total_ms'] * 1e-3) / 1e12,
                               'frac': prof['attention']['flops'] / (prof['attention']['total_ms'] * 1e-3) / 1e12 / PEAK_F32_MFMA_TFLOPS,
                               'avg_launch_us': prof['attention']['total_ms'] / prof['attention']['launches'] * 1e3,
                               'unit': 'TFLOP/s', 'peak': PEAK_F32_MFMA_TFLOPS}
                              if prof.get('attention', {}).get('total_ms') else in_stack_attention(net, B, dev)),
                # ... and the stand-alone attention kernel (the launch shape of the launch-per-GEMM path) for comparison
                'attention_standalone': (None if prof.get('attention', {}).get('total_ms') or _PRODUCTS else standalone_attention(B, dev)),
                # the dominant kernel on its own (VERDICT convention: algorithmic flops per launch / average launch duration)
                'dominant': ({'kernel': ('encoder_stack_kernel (label gemm_stack_tail: the WHOLE denoising step -- input embedding, the eight encoder '
                                         'layers, output head, DDPM update and the next step\'s pack -- as one launch)' if dom == 'gemm_stack_tail' else
                                         'encoder_stack_kernel (label gemm_stack: the whole encoder of one denoising step)'),
                              'launches_timed': prof[dom]['launches'],
                              'alg_gflop_per_launch': prof[dom]['flops'] / prof[dom]['launches'] / 1e9,
                              'avg_launch_us': prof[dom]['total_ms'] / prof[dom]['launches'] * 1e3,
                              'achieved': prof[dom]['flops'] / (prof[dom]['total_ms'] * 1e-3) / 1e12,
                              'frac': prof[dom]['flops'] / (prof[dom]['total_ms'] * 1e-3) / 1e12 / PEAK_F32_MFMA_TFLOPS,
                              'time_share_of_kernels': prof[dom]['total_ms'] / all_ms,
                              'traffic': pmc_traffic_of('encoder_stack_kernel') if B == 64 else None,
                              'traffic_note': 'ARCHIVED HBM-side bytes per launch (FETCH_SIZE x2 + WRITE_SIZE, separate rocprofv3 PMC passes of the '
                                              'B = 64 workload).  Two yardsticks: what a fused stack MUST move -- the weights once (8 x 101 MB), the '
                                              'packed input and the output: ~0.85 GB -- and what THIS design moves by construction -- every '
                                              'inter-phase activation written and read once through L2 / HBM, the weights once per XCD: 3.3 GB.  The '
                                              'measured figure is ~4x the first and ~1.02x the second: no re-reads beyond the design, but the design '
                                              'itself is far from traffic-minimal; at 1.2 TB/s (15 % of HBM peak) on an MFMA-bound kernel that is '
                                              'not what limits it'}
                             if dom else None),
                'launches_timed': g_n, 'avg_launch_us': g_ms / g_n * 1e3 if g_n else None,
                'alg_gflop_per_launch': g_fl / g_n / 1e9 if g_n else None,
                # round 4: the out-projection / FF2 launches (`gemm_bias_res_ln`) carry the LayerNorm that used to be 16 separate
                # launches per step; their time counts here against the GEMM's 2 M N K flops only
                'note': ('gemm_stack = ONE launch per denoising step for the whole encoder: per layer attention, out-projection + norm1, linear1 + '
                         'GELU, linear2 + norm2 and the next layer\'s in-projection (csrc/encoder_chain.hip); its flops are the executed 2 M N K '
                         'of its GEMMs plus 4 S^2 d_h per (clip, head) of attention, so the family figure is all fp32-MFMA work of the step.  '
                         if any(k.startswith('gemm_stack') for k in gemm) else
                         'gemm_chain = one launch for out-projection + norm1, linear1 + GELU, linear2 + norm2 and the next in-projection.  '
                         if any(k.startswith('gemm_chain') for k in gemm) else '') +
                        ('GEMM time includes the LayerNorm work fused into the out-projection / FF2 launches (gemm_bias_res_ln); rounds 1-3 '
                         'timed LayerNorm as its own kernel outside this family.  Flops are the EXECUTED 2 M N K of each launch: the embed '
                         'GEMM contracts the x_t half only (K = 320; the cond half is computed once per sampling loop), 0.9 % fewer flops per '
                         'step than model_tflops credits (the reference\'s 5.298 GFLOP per clip-step)'),
                'gemm_time_share_of_kernels': g_ms / all_ms if all_ms else None,
                'kernels': kernels,
            },
        }
        rec['rank_report'] = report
        if dist is not None:
            rec['process_group'] = {'backend': dist.get_backend(), 'world_size': dist.get_world_size(), 'forced': bool(args.force_dist)}
        if world == 1 and (not args.no_cpu_baseline or args.config_cpu_baseline) and not prox:
            rec['cpu_baseline'] = cpu_baseline(batch=B)
        if world == 1 and (not args.no_cpu_baseline or args.config_cpu_baseline) and prox:
            try:
                rec['cpu_baseline'] = cpu_baseline_config('prox', B)
            except Exception as e:
                rec['cpu_baseline'] = {'error': f'{type(e).__name__}: {e}'}
        if world == 1 and not prox and (args.with_accuracy or not args.no_cpu_baseline):
            rec['accuracy'] = accuracy_vs_reference(dev)
        if world == 1 and not args.no_extras and not prox and not _PRODUCTS:
            torch.cuda.synchronize(dev)
            # safety copy of the headline on stderr before the child legs start (a slow or hung child can then cost the extras, never
            # the measurement); stdout still carries exactly ONE JSON line, printed below
            print('bench.py headline (extras follow): ' + json.dumps({k: rec[k] for k in ('metric', 'value', 'unit', 'ms_per_step')}),
                  file=sys.stderr, flush=True)
            rec['second_line'], rec['configs'] = extras(args)
        print(json.dumps(rec), flush=True)
    finish(world, dist)


if __name__ == '__main__':
    main()
