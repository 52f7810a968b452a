"""The GEMM chain of an encoder layer as one launch (csrc/encoder_chain.hip: out-projection + norm1, linear1 + GELU, linear2 + norm2
and the next layer's in-projection -- nn.TransformerEncoderLayer, model/posenet.py:63-69) against the launch-per-GEMM path it
replaces, and against the reference's goldens.  At the batch sizes it is used by default (B >= 32: 4 or 8 column tiles per clip in
every phase, the very tiles the launch-per-GEMM path picks) the two paths must agree BIT FOR BIT: same fragments, same k order, same
LayerNorm statistics tree."""
import pytest
import torch

from helpers import cpu_noise_sequence, golden, max_abs, seeded
from test_gpu_posenet import DEV, make_diffusion, make_posenet

pytestmark = pytest.mark.gpu


def _pair(monkeypatch, seed=5, chain='stack', any_batch=False):
    """(handle with one launch per GEMM, handle with the chain form `chain`: 'layer' = the four GEMMs between two attention launches
    as one launch, 'stack' = the whole encoder, attention included, as one launch)."""
    with monkeypatch.context() as m:
        m.setenv('ROHM_POSENET_CHAIN', '0')
        plain, _ = make_posenet(seed)
        plain.native(torch.device(DEV))
    with monkeypatch.context() as m:
        m.setenv('ROHM_POSENET_CHAIN', chain)
        if any_batch:
            m.setenv('ROHM_POSENET_CHAIN_ANY', '1')
        net, _ = make_posenet(seed)
        nat = net.native(torch.device(DEV))
    if nat.exchange_mode & 16 == 0:
        pytest.skip(f'the layout guard refused the exchanging launches on this device: {nat.exchange_guard}')
    assert bool(nat.exchange_mode & 32) == (chain == 'stack')
    assert plain.native(torch.device(DEV)).exchange_mode & 48 == 0
    return plain, net


def _inputs(B, T=143):
    x, c = seeded(1, B, 294, 1, T).to(DEV), seeded(2, B, 294, 1, T).to(DEV)
    t = torch.tensor([(37 * i + 1) % 1000 for i in range(B)], device=DEV)
    return x, c, t


@pytest.mark.parametrize('chain', ['layer', 'stack'])
@pytest.mark.parametrize('B', [64, 32, 128])
def test_chain_forward_matches_one_launch_per_gemm(B, chain, monkeypatch):
    """Same tiles, same fragments, same k order, same LayerNorm statistics tree as the launch-per-GEMM path: the two agree to the last
    bit or two (hipcc contracts `a * b + c` of the epilogues into an fma per instantiation, so an occasional last-place difference
    is allowed: 4e-6 on |y| <= 4; B = 64 has been seen bit-identical), and the chain itself is bit-reproducible run after run.
    'stack' at B >= 64 also runs attention as a whole (clip, head) item on four waves instead of eight (other cooperative split of the
    ninth query block: summation order)."""
    plain, net = _pair(monkeypatch, chain=chain)
    x, c, t = _inputs(B)
    want = plain({'x_t': x, 'cond': c}, t)
    got = net({'x_t': x, 'cond': c}, t)
    net.check_exchange()
    diff = (got - want).abs()
    print(f'B={B} {chain}: max|chain - launches| = {float(diff.max()):.3e}, {int((diff > 0).sum())} of {diff.numel()} elements differ')
    assert float(diff.max()) < 2e-5
    for _ in range(20):      # race screen: the clip's workgroups meet five times per layer; every run the same bits
        assert torch.equal(net({'x_t': x, 'cond': c}, t), got)
    net.check_exchange()


@pytest.mark.parametrize('chain', ['layer', 'stack'])
@pytest.mark.parametrize('B', [56, 48, 40, 33, 3, 1])
def test_chain_at_other_batch_sizes(B, chain, monkeypatch):
    """Row tiles that are not a multiple of 8 (surplus workgroups leave), more than one round of workgroups (B = 40: 320), tiny
    batches (forced: by default they keep the launch-per-GEMM path, whose narrower tiles fill more CUs), and -- round 6 -- 48 / 56
    clips as ONE part-filled round of 4-part workgroups (192 / 224 of 256).  The launch-per-GEMM path picks other tile widths here,
    so agreement is to summation order."""
    plain, net = _pair(monkeypatch, chain=chain, any_batch=True)
    x, c, t = _inputs(B)
    want = plain({'x_t': x, 'cond': c}, t)
    got = net({'x_t': x, 'cond': c}, t)
    net.check_exchange()
    assert max_abs(got, want) < 2e-5
    assert torch.equal(net({'x_t': x, 'cond': c}, t), got)


@pytest.mark.parametrize('chain', ['layer', 'stack'])
def test_chain_vs_reference_goldens(chain, monkeypatch):
    """Forward < 1e-4 and the 8-step loop < 1e-4 against the reference's own outputs, with the chain forced at B = 2."""
    monkeypatch.setenv('ROHM_POSENET_CHAIN', chain)
    monkeypatch.setenv('ROHM_POSENET_CHAIN_ANY', '1')
    g = golden('posenet_forward.npz')
    net, _ = make_posenet(int(g['weight_seed']))
    x, c = seeded(int(g['x_seed']), 2, 294, 1, 143), seeded(int(g['cond_seed']), 2, 294, 1, 143)
    y = net({'x_t': x.to(DEV), 'cond': c.to(DEV)}, torch.from_numpy(g['t']).to(DEV)).cpu()
    if net.native(torch.device(DEV)).exchange_mode & 16 == 0:
        pytest.skip('layout guard refused the exchanging launches')
    assert max_abs(y, torch.from_numpy(g['y'])) < 1e-4
    g = golden('posenet_loop8.npz')
    net, _ = make_posenet(int(g['weight_seed']))
    steps = int(g['steps'])
    cond = seeded(int(g['cond_seed']), 2, 294, 1, 143)
    x_T, noises = cpu_noise_sequence(int(g['torch_seed']), (2, 294, 1, 143), steps)
    diff = make_diffusion(steps)
    diff.noise_source = lambda step, like: (x_T if step == -1 else noises[step])
    y = diff.p_sample_loop(net, {'cond': cond.to(DEV)}, [2, 294, 1, 143])
    assert max_abs(y.cpu(), torch.from_numpy(g['y'])) < 1e-4


@pytest.mark.parametrize('chain', ['layer', 'stack'])
def test_chain_loop_at_the_headline_batch(chain, monkeypatch):
    """8 denoising steps of 64 clips through the fused loop: chain vs launch-per-GEMM; recorded into a hipGraph the chain replays
    correctly (its tags come from the workspace's pass counter)."""
    plain, net = _pair(monkeypatch, chain=chain)
    B = 64
    cond = seeded(4, B, 294, 1, 143).to(DEV)
    x_T, noises = cpu_noise_sequence(9, (B, 294, 1, 143), 8)
    outs = []
    for n in (plain, net):
        diff = make_diffusion(8)
        diff.noise_source = lambda step, like: (x_T if step == -1 else noises[step])
        outs.append(diff.p_sample_loop(n, {'cond': cond}, [B, 294, 1, 143]))
    d = float((outs[0] - outs[1]).abs().max())
    print(f'8-step loop, B = 64, {chain}: max|chain - launches| = {d:.3e}')
    assert d < 2e-5
    x, c, t = _inputs(B)
    ref = net({'x_t': x, 'cond': c}, t)
    side = torch.cuda.Stream()
    xs = x.clone()
    with torch.cuda.stream(side):
        net({'x_t': xs, 'cond': c}, t)
    torch.cuda.synchronize()
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph, stream=side):
        y = net({'x_t': xs, 'cond': c}, t)
    for _ in range(3):
        graph.replay()
        torch.cuda.synchronize()
        assert torch.equal(y, ref)
    with torch.cuda.stream(side):
        net.check_exchange()


@pytest.mark.parametrize('chain', ['layer', 'stack'])
def test_a_failed_exchange_inside_the_chain_is_survived(chain, monkeypatch):
    """The chain's first LayerNorm exchange sabotaged (rohm_posenet_inject_exchange_fault): waits expire, the loop falls back to one
    launch per GEMM without the in-kernel LayerNorm and repeats the chunk -- the result of a handle that never used them."""
    with monkeypatch.context() as m:
        m.setenv('ROHM_POSENET_LN_FUSED', '0')
        m.setenv('ROHM_POSENET_HEAD_SK', '0')
        plain, _ = make_posenet(5)
        assert plain.native(torch.device(DEV)).exchange_mode == 0      # the handle reads the environment when it is created
    _, net = _pair(monkeypatch, chain=chain)
    B = 32
    cond = seeded(4, B, 294, 1, 143).to(DEV)
    x_T, noises = cpu_noise_sequence(9, (B, 294, 1, 143), 6)

    def run(n):
        diff = make_diffusion(6)
        diff.fused_chunk = 3
        diff.noise_source = lambda step, like: (x_T if step == -1 else noises[step])
        return diff.p_sample_loop(n, {'cond': cond}, [B, 294, 1, 143])
    want = run(plain)
    nat = net.native(torch.device(DEV))
    nat.inject_exchange_fault(1)
    with pytest.warns(UserWarning, match='ran into its bound'):
        got = run(net)
    assert nat.exchange_mode & 51 == 0 and nat.exchange_mode & 8
    assert torch.equal(got, want)


@pytest.mark.parametrize('B', [64, 32, 48, 128])
def test_one_launch_per_denoising_step_matches_the_three_launch_step(B, monkeypatch):
    """Round 6: in the sampling loop the stack closes with OutputProcess (model/heads.py:171-176), the ancestral update
    (gaussian_diffusion_posenet.py:212-234,426-434) and the next step's pack -- ONE `rohm::` kernel per un-guided denoising step at
    the configs' batch sizes (single-round launches).  Against the same library with ROHM_POSENET_STACK_TAIL=0 (stack + stream-K
    head + finish_pack): the head sums K in another order, so agreement is to rounding (1e-5 after 7 steps on |x| ~ 4), the returned
    sample, the last pred_xstart (early_stop) and the input of the last step (batch['x_t']) alike; bit-reproducible; and the launch
    profiler sees exactly one label per step.  B = 48: one part-filled round of 4-part workgroups; B = 128: TWO rounds -- the pass
    accounting (host-numbered steps + pending word) must hold when late workgroups of a launch start after early ones have finished."""
    from rohm_amd import _lib
    steps = 7
    x_T, noises = cpu_noise_sequence(17, (B, 294, 1, 143), steps)
    cond = seeded(18, B, 294, 1, 143).to(DEV)

    def run(tail, early_stop, chunk):
        with monkeypatch.context() as m:
            m.setenv('ROHM_POSENET_STACK_TAIL', tail)
            net, _ = make_posenet(9)
            nat = net.native(torch.device(DEV))
        if nat.exchange_mode & 32 == 0:
            pytest.skip(f'no encoder stack on this device: {nat.exchange_guard}')
        diff = make_diffusion(steps)
        diff.noise_source = lambda step, like: (x_T if step == -1 else noises[step])
        diff.fused_chunk = chunk
        diff._indices = lambda skip=0, early_stop=False: list(range(steps))[::-1]
        batch = {'cond': cond}
        _lib.profile_start(1)
        y = diff.p_sample_loop(net, batch, [B, 294, 1, 143], early_stop=early_stop)
        torch.cuda.synchronize()
        prof = _lib.profile_stop()
        return y.clone(), batch['x_t'].clone(), prof

    for early_stop, chunk in ((False, 50), (True, 3)):
        want, want_in, prof0 = run('0', early_stop, chunk)
        got, got_in, prof1 = run('1', early_stop, chunk)
        again, _, _ = run('1', early_stop, chunk)
        err, err_in = max_abs(got, want), max_abs(got_in, want_in)
        print(f'B={B} early_stop={early_stop} chunk={chunk}: max|one launch - three launches| = {err:.3e} (input of the last step {err_in:.3e})')
        assert err < 1e-5 and err_in < 1e-5
        assert torch.equal(got, again)
        assert prof1['gemm_stack_tail']['launches'] == steps and 'finish_pack' not in prof1 and 'finish_ddpm' not in prof1
        assert not any(k.startswith('gemm_out_t') for k in prof1), sorted(prof1)
        assert prof0['gemm_stack']['launches'] == steps and 'gemm_stack_tail' not in prof0
        n_launch = sum(v['launches'] for v in prof1.values())
        n_calls = -(-steps // chunk)
        # per call: cond pack + cond embed (+ memset) + the first x_t pack; per step: ONE launch
        print('   launches with the closing phase:', {k: v['launches'] for k, v in prof1.items()}, '| without:', {k: v['launches'] for k, v in prof0.items()})
        assert n_launch <= steps + 4 * n_calls


def test_the_stack_is_used_where_whole_rounds_of_its_workgroups_fit():
    """ADVICE r5 / profiles/r6_f_tail_ab_and_batch_sweep.json: the persistent one-per-CU workgroups of the stack cost whole rounds of
    256, so by default it runs at B = 32 (256 8-part workgroups), 48 .. 64 (one round of 4-part ones), 128 (two full rounds) -- and
    NOT at 40, 72 or 96, where one launch per GEMM is 7-27 % faster."""
    from rohm_amd import _lib
    net, _ = make_posenet(5)
    nat = net.native(torch.device(DEV))
    if nat.exchange_mode & 32 == 0:
        pytest.skip(f'no encoder stack on this device: {nat.exchange_guard}')
    for B, stacked in ((32, True), (40, False), (48, True), (56, True), (64, True), (72, False), (96, False), (128, True)):
        x, c, t = _inputs(B)
        net({'x_t': x, 'cond': c}, t)
        _lib.profile_start(1)
        y = net({'x_t': x, 'cond': c}, t)
        torch.cuda.synchronize()
        prof = _lib.profile_stop()
        assert torch.isfinite(y).all()
        assert ('gemm_stack' in prof) == stacked, (B, sorted(prof))
    net.check_exchange()


@pytest.mark.parametrize('B', [32, 64])
def test_stack_phase_timeline_is_consistent(B):
    """`rohm_posenet_set_stack_timeline` / rohm_amd.stack_timeline (bench.py `roofline.attention`): the stamps the kernel writes about
    itself add up -- the phase spans of a workgroup tile its launch, the launch span is the step the events time, the attention phase's
    rate is a plausible fraction of the fp32-MFMA peak -- and the switch really is off afterwards (no stamp is written by later launches)."""
    from rohm_amd import stack_timeline
    net, _ = make_posenet(5)
    rec = stack_timeline.measure(net, B, reps=3, device=DEV)
    if 'error' in rec:
        pytest.skip(rec['error'])
    ph = rec['phases']
    covered = sum(v['us_per_launch'] for k, v in ph.items() if k != 'finish_skew')
    print(f"B={B}: span {rec['launch_span_us']:.1f} us, phases sum to {covered:.1f} us, step by events {rec['step_wall_us_by_events']:.1f} us, "
          f"attention {rec['attention_in_stack']['frac']:.3f} of peak")
    assert 0.90 * rec['launch_span_us'] < covered <= 1.001 * rec['launch_span_us']            # mean workgroup: its phases + meetings, minus its lateness
    assert 0.9 * rec['launch_span_us'] < rec['step_wall_us_by_events'] < 1.15 * rec['launch_span_us']
    assert 0.3 < rec['attention_in_stack']['frac'] < 0.9 and 0.04 < rec['attention_in_stack']['share_of_launch'] < 0.15
    assert 'head_update_pack' in ph and ph['in_proj_next']['frac_of_fp32_mfma_peak'] > ph['out_proj_norm1']['frac_of_fp32_mfma_peak']
    # off again: a later forward leaves a poisoned buffer alone
    from rohm_amd._lib import check, lib, ptr
    nat = net.native(torch.device(DEV))
    n = lib().rohm_posenet_stack_timeline_bytes(B)
    buf = torch.full((n // 8,), -7, dtype=torch.int64, device=DEV)
    x, c, t = _inputs(B)
    net({'x_t': x, 'cond': c}, t)
    torch.cuda.synchronize()
    assert bool((buf == -7).all())
    check(lib().rohm_posenet_set_stack_timeline(nat.handle, ptr(buf), n, B), 'rohm_posenet_set_stack_timeline')
    net({'x_t': x, 'cond': c}, t)
    torch.cuda.synchronize()
    check(lib().rohm_posenet_set_stack_timeline(nat.handle, None, 0, 0), 'rohm_posenet_set_stack_timeline')
    assert int((buf != -7).sum()) > 0


def test_one_launch_steps_recorded_into_a_graph_draw_fresh_tags_on_every_replay():
    """The fused sampling loop with its closing phase, RECORDED: the host bakes each step's index into its launch (`pass_add`), the
    device-side counter advances once per call by 1 + the steps the previous call consumed (the x_t pack folds the pending word in) --
    so every replay of the recorded call draws tags no earlier launch used, and its samples equal the eager call's bit for bit."""
    import numpy as np
    from rohm_amd import _lib
    B, T, n = 32, 143, 4
    net, _ = make_posenet(5)
    nat = net.native(torch.device(DEV))
    if nat.exchange_mode & 32 == 0:
        pytest.skip(f'no encoder stack on this device: {nat.exchange_guard}')
    g = torch.Generator(device=DEV).manual_seed(3)
    x0 = torch.randn(B, 294, 1, T, device=DEV, generator=g)
    cond = torch.randn(B, 294, 1, T, device=DEV, generator=g)
    noise = torch.randn(n, B, 294, 1, T, device=DEV, generator=g)
    coef = np.asarray([[0.05, 0.95, 0.1]] * n, np.float32)
    ts = [400, 399, 398, 397]
    eager = x0.clone()
    net.sample_loop_native(eager, cond, ts, coef, noise)
    eager2 = eager.clone()
    net.sample_loop_native(eager2, cond, ts, coef, noise)          # a second call continues from the first one's output
    net.check_exchange()
    side = torch.cuda.Stream()
    xs = x0.clone()
    with torch.cuda.stream(side):
        warm = x0.clone()
        net.sample_loop_native(warm, cond, ts, coef, noise)        # this stream's workspace exists and is armed before the capture
        ws = nat.workspace(B, T)
    torch.cuda.synchronize()
    assert torch.equal(warm, eager)
    off = _lib.lib().rohm_posenet_status_offset(nat.handle, B, T)
    word = ws[off:off + 16].view(torch.int32)
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph, stream=side):
        net.sample_loop_native(xs, cond, ts, coef, noise)
    passes = []
    xs.copy_(x0)
    for want in (eager, eager2):
        graph.replay()
        torch.cuda.synchronize()
        passes.append((int(word[2]), int(word[3])))
        if int(word[0]) != 0:      # a bounded wait expired on this box (a recorded call has no host loop to fall back and re-run): the caller's
            pytest.skip(f'an in-kernel exchange failed during a replay (status {int(word[0])}): the documented remedy is the eager loop')      # check says so
        assert torch.equal(xs, want), (len(passes), max_abs(xs, want), passes)
    # per call: the pack advances the counter by 1 + the pending count of the call before it; the last launch leaves n pending
    assert passes[1][0] - passes[0][0] == 1 + n and passes[0][1] == passes[1][1] == n, passes
    assert int(word[0]) == 0
    with torch.cuda.stream(side):
        net.check_exchange()
