"""The in-kernel exchanges of the default PoseNet path (LayerNorm inside the out-projection / FF2 GEMMs, stream-K output head:
csrc/gemm_f32.hip, csrc/exchange.hip) under the conditions they must survive: a sabotaged exchange (the waits really expire), a CU
mask that the layout guard must notice, a CU mask it is told to ignore, recycled workspace memory.  In every case the sampling loop
must return the result of the exchange-free launches -- through the guard or through fallback + re-run -- never wrong samples.
Reference work under test: model/posenet.py:63-69 (post-norm tails), model/heads.py:171-176 (OutputProcess)."""
import json
import os
import subprocess
import sys

import pytest
import torch

from helpers import cpu_noise_sequence, golden, max_abs, seeded
from test_gpu_posenet import DEV, make_diffusion, make_posenet

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _loop8(net, B, chunk=None):
    """The 8-step loop of tests/golden/posenet_loop8.npz, widened to B clips (clips 0, 1 are the golden's)."""
    g = golden('posenet_loop8.npz')
    steps = int(g['steps'])
    cond = seeded(int(g['cond_seed']), 2, 294, 1, 143)
    x_T, noises = cpu_noise_sequence(int(g['torch_seed']), (2, 294, 1, 143), steps)
    if B > 2:
        rep = lambda a, s: torch.cat([a, seeded(s, B - 2, 294, 1, 143)])
        cond, x_T, noises = rep(cond, 901), rep(x_T, 902), [rep(n, 903 + i) for i, n in enumerate(noises)]
    diff = make_diffusion(steps)
    diff.noise_source = lambda step, like: (x_T if step == -1 else noises[step])
    if chunk:
        diff.fused_chunk = chunk
    y = diff.p_sample_loop(net, {'cond': cond.to(DEV)}, [B, 294, 1, 143])
    return y, torch.from_numpy(g['y'])


@pytest.mark.parametrize('B', [2, 32, 64])
def test_a_failed_exchange_is_survived_by_fallback_and_rerun(B, monkeypatch):
    """rohm_posenet_inject_exchange_fault makes ONE LayerNorm-carrying launch publish a column tile's statistics under a wrong tag:
    its partners' waits expire for real (~0.2 s), the error word is set by the kernel, later launches of the chunk give up at once,
    the loop switches the handle to the GEMM + LayerNorm pair / plain tiles and repeats the chunk from its saved input.  Result:
    bit-equal to a handle that never used the exchanging launches, and on the reference's golden."""
    with monkeypatch.context() as m:
        m.setenv('ROHM_POSENET_LN_FUSED', '0')
        m.setenv('ROHM_POSENET_HEAD_SK', '0')
        plain, _ = make_posenet(int(golden('posenet_loop8.npz')['weight_seed']))
        want, gold = _loop8(plain, B, chunk=3)
    assert plain.native(torch.device(DEV)).exchange_mode == 0
    assert max_abs(want[:2].cpu(), gold) < 1e-4
    net, _ = make_posenet(int(golden('posenet_loop8.npz')['weight_seed']))
    nat = net.native(torch.device(DEV))
    if nat.exchange_mode & 1 == 0:
        pytest.skip(f'the layout guard refused the exchanging launches on this device: {nat.exchange_guard}')
    fused, _ = _loop8(net, B, chunk=3)                       # healthy: fused launches, no warning
    assert nat.exchange_mode & 8 == 0 and max_abs(fused[:2].cpu(), gold) < 1e-4
    nat.inject_exchange_fault(1)
    # chunks of 3 steps: the fault hits the first chunk only
    with pytest.warns(UserWarning, match='ran into its bound'):
        got, _ = _loop8(net, B, chunk=3)
    assert nat.exchange_mode & 3 == 0 and nat.exchange_mode & 8
    assert torch.equal(got, want)
    again, _ = _loop8(net, B, chunk=3)                       # the fallback is sticky and silent afterwards
    assert torch.equal(again, want)


def test_a_failed_exchange_in_a_stepwise_forward_is_rerun():
    """The step-wise path (p_sample / p_sample_with_grad, the guided tails) polls after every forward."""
    net, _ = make_posenet(5)
    nat = net.native(torch.device(DEV))
    if nat.exchange_mode & 1 == 0:
        pytest.skip(f'the layout guard refused the exchanging launches on this device: {nat.exchange_guard}')
    B = 4
    x, c = seeded(1, B, 294, 1, 143).to(DEV), seeded(2, B, 294, 1, 143).to(DEV)
    t = torch.full((B,), 500, device=DEV, dtype=torch.int64)
    dif = make_diffusion(1000)
    nz = seeded(3, B, 294, 1, 143)
    dif.noise_source = lambda step, like: nz
    want = dif.p_sample(net, {'cond': c}, x, t)['sample']
    nat.inject_exchange_fault(2)
    with pytest.warns(UserWarning, match='ran into its bound'):
        got = dif.p_sample(net, {'cond': c}, x, t)['sample']
    assert nat.exchange_mode & 8 and max_abs(got, want) < 2e-5       # fused vs un-fused LayerNorm: summation order only


@pytest.mark.parametrize('fill', [0xAB, 0x00, 0x01, 0xFF])
def test_recycled_workspace_memory_is_not_trusted(fill):
    """the workspace is torch.empty memory.  Whatever a recycled block holds -- small integers that look like tags of a
    young process included -- the first call zeroes the slot regions (a real tag never has a zero XCD field) and the tags carry a
    per-handle salt: the forward is bit-equal to the one on a fresh workspace and the status stays clean."""
    net, _ = make_posenet(5)
    B, T = 32, 143
    x, c = seeded(1, B, 294, 1, T).to(DEV), seeded(2, B, 294, 1, T).to(DEV)
    t = torch.tensor([(37 * i + 1) % 1000 for i in range(B)], device=DEV)
    y0 = net({'x_t': x, 'cond': c}, t)
    nat = net.native(torch.device(DEV))
    ws = nat.workspace(B, T)
    for k in range(3):
        ws.fill_(fill)
        if fill == 0x01:
            ws.view(torch.int32).fill_(k + 1)          # every word = a small integer
        y = net({'x_t': x, 'cond': c}, t)
        assert torch.equal(y, y0)
        net.check_exchange()


CHILD = r'''
import json, os, sys, warnings
sys.path.insert(0, {root!r}); sys.path.insert(0, os.path.join({root!r}, 'tests'))
import torch
from test_gpu_exchange import _loop8
from test_gpu_posenet import make_posenet, DEV
from helpers import golden
out = {{}}
for B in (32, 64):
    net, _ = make_posenet(int(golden('posenet_loop8.npz')['weight_seed']))
    with warnings.catch_warnings(record=True) as w:
        warnings.simplefilter('always')
        y, gold = _loop8(net, B, chunk=4)
    nat = net.native(torch.device(DEV))
    out[B] = dict(mode=nat.exchange_mode, guard=nat.exchange_guard, warned=[str(x.message)[:80] for x in w],
                  err_golden=float((y[:2].cpu() - gold).abs().max()))
    torch.save(y.cpu(), {tmp!r} + f'/y{{B}}.pt')
print('RESULT ' + json.dumps(out))
'''


@pytest.mark.parametrize('guard', ['default', 'probe', 'off'])
def test_loops_under_a_cu_mask_return_the_exchange_free_result(guard, tmp_path, monkeypatch):
    """A CU mask that leaves 24 of the 32 CUs of every XCD (HSA_CU_MASK, honoured by the runtime at queue creation) breaks what the
    exchanging launches assume: 256 co-resident one-per-CU workgroups.  'default': the guard sees the variable and never uses them.
    'probe': the environment shortcut is skipped and the PROBE LAUNCH must notice.  'off': the guard is told to trust the device --
    the launches run under the mask; whether their waits expire or not, the loop must still return the right samples (fallback +
    re-run).  B = 32 (8 partner tiles) and B = 64 (4), against the un-fused handle bit for bit or the golden at 1e-4."""
    with monkeypatch.context() as m:
        m.setenv('ROHM_POSENET_LN_FUSED', '0')
        m.setenv('ROHM_POSENET_HEAD_SK', '0')
        plain, _ = make_posenet(int(golden('posenet_loop8.npz')['weight_seed']))
        want = {B: _loop8(plain, B, chunk=4)[0].cpu() for B in (32, 64)}
    env = dict(os.environ)
    # CUs are numbered round-robin over the shader engines of an XCD; any 192-of-256 mask takes CUs away from every XCD
    env['HSA_CU_MASK'] = '0:0-191'
    if guard != 'default':
        env['ROHM_EXCHANGE_GUARD'] = guard
    script = tmp_path / 'child.py'
    script.write_text(CHILD.format(root=ROOT, tmp=str(tmp_path)))
    r = subprocess.run([sys.executable, str(script)], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    res = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith('RESULT ')][-1][7:])
    print(guard, json.dumps(res))
    for B in (32, 64):
        info = res[str(B)]
        y = torch.load(str(tmp_path / f'y{B}.pt'))
        assert info['err_golden'] < 1e-4
        if guard == 'default':
            assert info['mode'] & 3 == 0 and info['mode'] & 4 and 'CU mask' in info['guard']
        if info['mode'] & 3 == 0:      # refused by the guard, or fallen back after a failed wait: the exchange-free kernels ran
            assert torch.equal(y, want[B])
        else:                          # the exchanging launches ran and completed under the mask: summation order only
            assert max_abs(y, want[B]) < 1e-4 and not info['warned']


def test_standalone_exchange_calls_never_probe_under_a_graph_capture():
    """the stand-alone exchanging entry points (rohm_gemm_res_layernorm_f32, rohm_output_process_f32 with a scratch) used to
    run the layout probe -- hipMalloc, a null-stream launch, a device synchronisation -- lazily on their first call, which breaks a
    graph capture and then caches "probe launch failed" for the life of the process.  Now: on a device nobody has probed, a call on a
    capturing stream probes nothing and caches nothing (UNSUPPORTED / plain tiles); `rohm_exchange_probe` is the explicit set-up call;
    after it the same calls are recorded with their exchanges and replay bit-equal to the eager result.  Own interpreter: the device
    must be un-probed."""
    import os
    import subprocess
    import sys
    import textwrap
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = textwrap.dedent(f"""\
        import sys
        sys.path.insert(0, {root!r}); sys.path.insert(0, {root!r} + '/tests')
        import torch
        from helpers import seeded
        from rohm_amd import _lib, ops
        d = 'cuda:0'
        M, N, K = 144 * 64, 512, 512
        a, w, r = seeded(1, M, K).to(d), (seeded(2, N, K) * 0.05).to(d), seeded(3, M, N).to(d)
        b, gm, bt = seeded(4, N).to(d), seeded(5, N).to(d), seeded(6, N).to(d)
        h, wo, bo = seeded(7, M, 512).to(d), (seeded(8, 272, 512) * 0.05).to(d), seeded(9, 272).to(d)
        side = torch.cuda.Stream()
        torch.cuda.synchronize()
        graph = torch.cuda.CUDAGraph()
        refused = False
        with torch.cuda.graph(graph, stream=side):
            try:
                ops.gemm_res_layernorm(a, w, b, r, gm, bt)
            except _lib.RohmHipError as e:
                refused = 'has not been probed' in str(e)
            y_plain = ops.output_process(h, wo, bo, 64, 143)          # un-probed + capturing: plain tiles, still recordable
        assert refused, 'an un-probed device under capture must be refused, not probed'
        graph.replay(); torch.cuda.synchronize()
        ok, why = ops.exchange_probe(d)
        print('probe:', ok, why)
        if not ok:
            print('OK (device refused by the guard: nothing more to check)'); sys.exit(0)
        eager = ops.gemm_res_layernorm(a, w, b, r, gm, bt)
        eager_head = ops.output_process(h, wo, bo, 64, 143)
        torch.cuda.synchronize()
        assert (eager_head - y_plain).abs().max() < 1e-4
        g2 = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g2, stream=side):
            y = ops.gemm_res_layernorm(a, w, b, r, gm, bt)           # probed: recorded WITH its exchange
        for _ in range(3):
            g2.replay(); torch.cuda.synchronize()
            assert torch.equal(y, eager)
        ok2, _ = ops.exchange_probe(d)                               # re-probe on demand, same verdict on a quiet device
        assert ok2
        print('OK')
        """)
    p = subprocess.run([sys.executable, '-c', code], capture_output=True, text=True, timeout=600, cwd='/tmp')
    assert p.returncode == 0 and 'OK' in p.stdout, p.stdout[-2000:] + p.stderr[-3000:]


def test_stack_loop_next_to_a_second_busy_process_on_the_same_gpu(tmp_path):
    """The multi-tenant case (what 8 ranks + stragglers, or a shared box, look like to one rank): while ANOTHER process keeps the same
    GPU busy with long kernels, the B = 64 encoder-stack loop -- 256 one-per-CU workgroups that meet through L2 -- must end with the
    exact samples: either undisturbed (the tenant's workgroups leave the stack's partners co-resident) or through the documented
    path (a bounded wait expires -> warning -> exchange-free launches -> the chunk re-run), never with a hang or wrong numbers.
    The handle is created BEFORE the tenant starts (a tenant present at create is the guard's case, tested above).  Own interpreter
    under a hard timeout."""
    tenant = tmp_path / 'tenant.py'
    tenant.write_text(
        'import time\nimport torch\n'
        'a = torch.randn(8192, 8192, device="cuda:0")\nb = torch.randn(8192, 8192, device="cuda:0")\n'
        'torch.mm(a, b)\ntorch.cuda.synchronize()\nprint("READY", flush=True)\nt0 = time.time()\n'
        'while time.time() - t0 < 25.0:\n    for _ in range(20):\n        c = torch.mm(a, b)\n    torch.cuda.synchronize()\n')
    main = tmp_path / 'main.py'
    main.write_text(f"""
import subprocess, sys, time, warnings
sys.path.insert(0, {ROOT!r}); sys.path.insert(0, {ROOT!r} + '/tests')
import torch
from test_gpu_exchange import _loop8
from test_gpu_posenet import DEV, make_posenet
from helpers import golden, max_abs
net, _ = make_posenet(int(golden('posenet_loop8.npz')['weight_seed']))
clean, gold = _loop8(net, 64, chunk=3)
nat = net.native(torch.device(DEV))
mode0 = nat.exchange_mode
torch.cuda.synchronize()
tenant = subprocess.Popen([sys.executable, {str(tenant)!r}], stdout=subprocess.PIPE, text=True)
assert tenant.stdout.readline().strip() == 'READY'
outs = []
t0 = time.time()
with warnings.catch_warnings(record=True) as w:
    warnings.simplefilter('always')
    for rep in range(6):
        y, _ = _loop8(net, 64, chunk=3)
        outs.append(y.clone())
    warned = [str(x.message)[:120] for x in w]
took = time.time() - t0
alive = tenant.poll() is None
tenant.kill(); tenant.wait()
for y in outs:
    assert torch.isfinite(y).all()
    assert max_abs(y, clean) < 1e-4, max_abs(y, clean)        # exact up to the fallback path's summation order
    assert max_abs(y[:2].cpu(), gold) < 1e-4
print('tenant alive during the runs:', alive, '| 6 x 8-step B = 64 loops took %.2f s' % took, '| exchange_mode', mode0, '->',
      nat.exchange_mode, '| warnings:', warned)
print('OK')
""")
    p = subprocess.run([sys.executable, str(main)], capture_output=True, text=True, timeout=420, cwd='/tmp')
    print(p.stdout[-1500:])
    assert p.returncode == 0 and 'OK' in p.stdout, p.stdout[-2000:] + p.stderr[-3000:]
