"""Phase timeline of the encoder-stack launch (diagnostics; `rohm_posenet_set_stack_timeline`, include/rohm_hip.h).

From 32 clips on, `nn.TransformerEncoder` of a PoseNet forward (model/posenet.py:63-69,92) is ONE launch in which every workgroup
walks, per layer, attention -> out-projection + norm1 -> linear1 + GELU -> linear2 + norm2 -> the next in-projection, meeting the
other workgroups of its clip between phases (csrc/encoder_chain.hip).  A launch-level profiler sees one kernel; this module asks the
kernel itself: lane 0 of every workgroup stamps the 100 MHz wall clock at every seam, and `measure()` turns the stamps of a few
launches into per-phase spans, waits, and the rate of the attention phase INSIDE the stack (bench.py `roofline.attention`).
"""
from __future__ import annotations

import numpy as np
import torch

from . import _lib
from ._lib import check, lib, ptr

LAYERS, STAMPS = 9, 12          # csrc/common.h kStackTimelineLayers / kStackTimelineStamps
TICK_US = 0.01                  # s_memrealtime: 100 MHz
PEAK_F32_MFMA_TFLOPS = 157.3    # MI355X_MICROARCH.md: v_mfma_f32_16x16x4_f32 at 2.4 GHz on 256 CUs

# (name, first stamp, last stamp) of a layer's spans: phases and the meetings between them
SPANS = (('wait_qkv', 0, 1), ('attention', 1, 2), ('wait_ctx', 2, 3), ('out_proj_norm1', 3, 4), ('wait_y', 4, 5),
         ('linear1_gelu', 5, 6), ('wait_ff', 6, 7), ('linear2_norm2', 7, 8), ('wait_h', 8, 9), ('in_proj_next', 9, 10))
FRONT_SPANS = (('embed', 0, 1), ('wait_embed', 1, 2), ('in_proj_0', 2, 3), ('wait_qkv0', 3, 4))


def _flops_per_layer(B, D=512, F=1024, S=144, H=4):
    M = B * S
    return {'attention': 4.0 * S * S * (D // H) * H * B, 'out_proj_norm1': 2.0 * M * D * D, 'linear1_gelu': 2.0 * M * D * F,
            'linear2_norm2': 2.0 * M * F * D, 'in_proj_next': 2.0 * M * 3 * D * D}


def measure(net, B, T=143, reps=5, device='cuda:0', seed=0):
    """Stamp `reps` one-step launches of the fused sampling loop (the shipped path: the stack with its leading embed / in-projection
    phases) at batch size B and return the per-phase record.  Needs a handle that runs the stack (exchange_mode bit 5) at this B."""
    dev = torch.device(device)
    nat = net.native(dev)
    if not nat.exchange_mode & 32 or B < 32:
        return {'error': f'no encoder-stack launch at B = {B} on this handle (exchange_mode {nat.exchange_mode}: {nat.exchange_guard})'}
    G = 4 if B >= 48 else 8          # csrc/encoder_chain.hip encoder_chain_parts
    groups8 = (B + 7) // 8 * 8
    nbytes = lib().rohm_posenet_stack_timeline_bytes(B)
    buf = torch.zeros(nbytes // 8, dtype=torch.int64, device=dev)
    g = torch.Generator(device=dev).manual_seed(seed)
    x = torch.randn(B, 294, 1, T, device=dev, generator=g)
    cond = torch.randn(B, 294, 1, T, device=dev, generator=g)
    noise = torch.randn(2, B, 294, 1, T, device=dev, generator=g)
    coef = np.asarray([[0.02, 0.98, 0.05]] * 2, np.float32)
    net.sample_loop_native(x, cond, [500, 499], coef, noise)          # warm: weights in L2 / MALL, clocks up
    check(lib().rohm_posenet_set_stack_timeline(nat.handle, ptr(buf), nbytes, B), 'rohm_posenet_set_stack_timeline')
    recs, wall = [], []
    try:
        for r in range(reps):
            # two steps per call; the second launch's stamps are the ones left in the buffer (steady state: operands warm)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            net.sample_loop_native(x, cond, [498 - 2 * r, 497 - 2 * r], coef, noise)
            e1.record()
            torch.cuda.synchronize(dev)
            wall.append(e0.elapsed_time(e1) * 1e3 / 2)
            recs.append(buf.cpu().numpy().reshape(-1, LAYERS, STAMPS)[:groups8 * G].astype(np.float64) * TICK_US)      # sized for 8 parts per clip
    finally:
        check(lib().rohm_posenet_set_stack_timeline(nat.handle, None, 0, 0), 'rohm_posenet_set_stack_timeline')
    net.check_exchange()
    blocks = np.arange(groups8 * G)
    clip = (blocks // 8 // G) * 8 + blocks % 8
    valid = clip < B
    fl = _flops_per_layer(B)
    out = {'batch': B, 'parts_per_clip': G, 'workgroups': int(valid.sum()), 'launches_stamped': reps,
           'step_wall_us_by_events': float(np.median(wall)),
           'clock': 's_memrealtime (100 MHz wall clock), lane 0 of every workgroup; spans are means over workgroups, summed over the 8 layers'}
    spans, launch, finish = {}, [], []
    for Tm in recs:
        Tm = Tm[valid]
        front = Tm[:, 8, 0].min() > 0
        tail = Tm[:, 7, 10].min() > 0          # the launch closes with head + DDPM update + pack (StackParams::tail)
        t0 = Tm[:, 8, 0].min() if front else Tm[:, 0, 0].min()
        last = Tm[:, 7, 10] if tail else Tm[:, 7, 8]
        t1 = last.max()
        launch.append(t1 - t0)
        for name, a, b in SPANS:
            tot = 0.0
            for l in range(8):
                if (name == 'wait_qkv' and l == 0) or (name == 'in_proj_next' and l == 7) or (name == 'wait_h' and l == 7 and not tail):
                    continue
                tot += float((Tm[:, l, b] - Tm[:, l, a]).mean())
            spans.setdefault(name, []).append(tot)
        if tail:
            spans.setdefault('head_update_pack', []).append(float((Tm[:, 7, 10] - Tm[:, 7, 9]).mean()))
        if front:
            for name, a, b in FRONT_SPANS:
                spans.setdefault(name, []).append(float((Tm[:, 8, b] - Tm[:, 8, a]).mean()))
        # skew: how far apart the workgroups of the launch finish
        spans.setdefault('finish_skew', []).append(float(last.max() - last.min()))
        finish.append(last - last.min())
    span = float(np.median(launch))          # medians over the stamped launches: one disturbed launch must not move the record
    out['launch_span_us'] = span
    out['phases'] = {}
    for name, vals in spans.items():
        us = float(np.median(vals))
        rec = {'us_per_launch': round(us, 2), 'share_of_launch': round(us / span, 4)}
        n_l = 7 if name == 'in_proj_next' else 8
        if name == 'head_update_pack':
            tf = 2.0 * B * 144 * 512 * 272 / (us * 1e-6) / 1e12
            rec.update(tflops=round(tf, 2), frac_of_fp32_mfma_peak=round(tf / PEAK_F32_MFMA_TFLOPS, 4))
        if name in fl:
            tf = fl[name] * n_l / (us * 1e-6) / 1e12
            rec.update(tflops=round(tf, 2), frac_of_fp32_mfma_peak=round(tf / PEAK_F32_MFMA_TFLOPS, 4))
        out['phases'][name] = rec
    att, wctx = out['phases']['attention']['us_per_launch'], out['phases']['wait_ctx']['us_per_launch']
    f_att = fl['attention'] * 8
    out['attention_in_stack'] = {
        'achieved': f_att / (att * 1e-6) / 1e12, 'frac': f_att / (att * 1e-6) / 1e12 / PEAK_F32_MFMA_TFLOPS, 'unit': 'TFLOP/s',
        'peak': PEAK_F32_MFMA_TFLOPS, 'us_per_launch': att, 'share_of_launch': att / span,
        'achieved_incl_meeting': f_att / ((att + wctx) * 1e-6) / 1e12,
        'frac_incl_meeting': f_att / ((att + wctx) * 1e-6) / 1e12 / PEAK_F32_MFMA_TFLOPS,
        'alg_gflop_per_launch': f_att / 1e9,
        'note': 'attention phase of encoder_stack_kernel, measured inside the launch: algorithmic 4 S^2 d_h flop per (clip, head) x 8 layers / '
                'the mean over workgroups of the phase span (stamp 1 -> 2), summed over the layers; "_incl_meeting" adds the wait for the '
                'clip\'s other workgroups behind it (stamp 2 -> 3: the imbalance between the halves of a split item lands there)'}
    # Is the finish skew systematic (the same workgroups / XCDs late in every launch) or random?  Systematic skew cannot be recovered
    # by letting clips run ahead across step boundaries; random skew could.
    F = np.stack(finish)                                       # [reps, workgroups]: us behind the first finisher
    xcd = (blocks % 8)[valid]
    per_xcd = np.array([[F[r][xcd == k].mean() for k in range(8)] for r in range(len(F))])
    mean_wg = F.mean(axis=0)
    cors = [float(np.corrcoef(F[a], F[b])[0, 1]) for a in range(len(F)) for b in range(a + 1, len(F))]
    out['finish_skew_analysis'] = {
        'per_xcd_mean_us_behind_first': [round(float(v), 2) for v in per_xcd.mean(axis=0)],
        'per_xcd_std_over_launches_us': [round(float(v), 2) for v in per_xcd.std(axis=0)],
        'mean_over_workgroups_us_behind_first': round(float(F.mean()), 2),
        'correlation_of_workgroup_lateness_between_launches': round(float(np.mean(cors)), 3) if cors else None,
        'max_of_mean_lateness_us': round(float(mean_wg.max()), 2),
        'note': 'lateness = a workgroup\'s last stamp minus the launch\'s first finisher; correlation near 1 = the same workgroups are late in '
                'every launch (systematic: placement, XCD), near 0 = random'}
    waits = sum(v['us_per_launch'] for k, v in out['phases'].items() if k.startswith('wait_'))
    out['meetings_us_per_launch'] = round(waits, 2)
    out['meetings_share_of_launch'] = round(waits / span, 4)
    return out


def text(rec):
    if 'error' in rec:
        return rec['error']
    lines = [f"encoder_stack_kernel<{rec['parts_per_clip']}> at B = {rec['batch']}: {rec['workgroups']} workgroups, {rec['launches_stamped']} launches stamped; "
             f"launch span {rec['launch_span_us']:.1f} us (denoising step by events: {rec['step_wall_us_by_events']:.1f} us)",
             f"{'phase':<18}{'us / launch':>12}{'share':>9}{'TFLOP/s':>10}{'of 157.3':>10}"]
    for name, v in rec['phases'].items():
        lines.append(f"{name:<18}{v['us_per_launch']:>12.1f}{v['share_of_launch']:>9.3f}"
                     f"{v.get('tflops', float('nan')):>10.1f}{v.get('frac_of_fp32_mfma_peak', float('nan')):>10.3f}")
    a = rec['attention_in_stack']
    lines.append(f"meetings (all waits): {rec['meetings_us_per_launch']:.1f} us = {rec['meetings_share_of_launch']:.3f} of the launch")
    fa = rec.get('finish_skew_analysis')
    if fa:
        lines.append(f"finish skew: mean lateness {fa['mean_over_workgroups_us_behind_first']} us, per XCD {fa['per_xcd_mean_us_behind_first']} "
                     f"(std over launches {fa['per_xcd_std_over_launches_us']}), lateness correlation between launches "
                     f"{fa['correlation_of_workgroup_lateness_between_launches']}")
    lines.append(f"attention inside the stack: {a['achieved']:.1f} TFLOP/s = {a['frac']:.3f} of the fp32-MFMA peak "
                 f"({a['share_of_launch']:.3f} of the launch); with the meeting behind it {a['achieved_incl_meeting']:.1f} = {a['frac_incl_meeting']:.3f}")
    return '\n'.join(lines)
