"""GPU: each hand-written kernel against a float64 CPU evaluation of the same op (through the C ABI)."""
import math

import pytest
import torch

from helpers import max_abs, seeded
from oracle import nets

pytestmark = pytest.mark.gpu


def _dev():
    return torch.device('cuda', 0)


@pytest.mark.parametrize('M,N,K', [(144, 512, 512), (288, 1536, 512), (144 * 3, 512, 1024), (100, 70, 64),
                                   (272, 288, 512), (1, 16, 64), (144 * 64, 512, 512),
                                   (144 * 32, 1536, 512), (144 * 64, 1536, 512), (144 * 64, 1024, 512)])   # 144x192 / x384 / x256 tiles
@pytest.mark.parametrize('epi', [0, 1, 2])
def test_gemm(M, N, K, epi):
    from rohm_amd import ops
    a, w = seeded(M + N, M, K), seeded(K + 7, N, K) / math.sqrt(K)
    bias, res = seeded(3, N), seeded(4, M, N)
    ref = a.double() @ w.double().T + bias.double()
    if epi == 1:
        ref = nets.gelu_erf(ref)
    if epi == 2:
        ref = ref + res.double()
    d = _dev()
    out = ops.gemm(a.to(d), w.to(d), bias.to(d), res.to(d) if epi == 2 else None, epi)
    # A = I check with asymmetric W (catches transposed C writes, guide G9)
    assert max_abs(out.cpu(), ref) < 2e-5 * math.sqrt(K / 32)


def test_gemm_identity_asymmetric():
    from rohm_amd import ops
    d = _dev()
    a = torch.eye(144, 192).contiguous()
    w = torch.arange(96 * 192, dtype=torch.float32).reshape(96, 192) / 100.0
    out = ops.gemm(a.to(d), w.to(d))
    assert torch.equal(out.cpu(), w[:, :144].T.contiguous())


@pytest.mark.parametrize('M,N,K', [(144 * 64, 512, 512), (144 * 64, 512, 1024),      # B = 64: 144 x 128 tiles, 4 partner tiles
                                   (144 * 32, 512, 512), (144 * 32, 512, 1024),      # B = 32: 144 x 64 tiles, 8 partner tiles
                                   (144 * 2, 512, 512), (144 * 13, 512, 1024),       # row tiles not a multiple of 8 (surplus workgroups leave)
                                   (144 * 128, 512, 512),                            # two rounds of workgroups
                                   (144 * 5, 256, 256), (144 * 3, 1024, 512)])       # other widths: 4 and 8 partner tiles of 64 / 128
def test_gemm_res_layernorm(M, N, K):
    """nn.TransformerEncoderLayer's post-norm tail x = norm(x + sublayer(x)) (model/posenet.py:63-69) as ONE launch: the column
    tiles of a row tile exchange row statistics through L2 while the kernel runs (gemm_f32.hip EPI_BIAS_RES_LN)."""
    from rohm_amd import ops
    a, w = seeded(M + N, M, K), seeded(K + 7, N, K) / math.sqrt(K)
    bias, res = seeded(3, N), seeded(4, M, N) * 2 + 0.3
    g, b = seeded(5, N) * 0.5 + 1.0, seeded(6, N)
    ref = nets.layer_norm(a.double() @ w.double().T + bias.double() + res.double(), g.double(), b.double())
    d = _dev()
    args = [t.to(d) for t in (a, w, bias, res, g, b)]
    out, scratch = ops.gemm_res_layernorm(*args, return_scratch=True)
    assert max_abs(out.cpu(), ref) < 2e-5 * math.sqrt(K / 32)
    # the two-kernel path it replaces
    two = ops.layernorm_(ops.gemm(args[0], args[1], args[2], args[3], 2), args[4], args[5])
    assert max_abs(out.cpu(), two.cpu()) < 1e-5
    torch.cuda.synchronize()
    words = scratch.cpu()
    assert int(words[0]) == 0, 'a partner wait ran into its bound'
    # the exchange goes through ONE L2: every column tile of a row tile must have run on the same XCD (the kernel's block -> tile
    # map relies on the hardware placing block b on XCD b % 8); the kernel records where each tile ran
    tm = M // 144
    ok = lambda bn: N % bn == 0 and N // bn in (1, 2, 4, 8)
    bn = 128 if (ok(128) and tm * (N // 128) >= 256) else (64 if ok(64) else 128)            # gemm_f32.hip ln_tile_width
    tiles_n = N // bn
    xcc = words[16 + tm * 8 * 144 * 4:16 + tm * 8 * 144 * 4 + tm * 8].view(tm, 8)[:, :tiles_n]
    assert int(xcc.min()) >= 1 and bool((xcc == xcc[:, :1]).all()), 'partner tiles ran on different XCDs'
    if tm >= 8:
        assert len(set(xcc[:, 0].tolist())) == 8, 'row tiles should be spread over all 8 XCDs'
    # race screen: 50 launches, same bits (statistics summed in tile order whatever the arrival order)
    for _ in range(50):
        assert torch.equal(ops.gemm_res_layernorm(*args), out)


@pytest.mark.parametrize('M', [144 * 64, 144 * 32])
def test_gemm_res_layernorm_rows_far_from_zero(M):
    """Rows whose mean is hundreds of standard deviations away from zero: the fused statistics are two-pass per tile and combined by
    Chan's update across the partner tiles, so they are as stable as nn.LayerNorm's own kernel -- a one-pass E[x^2] - mu^2 over fp32
    sums would lose the variance here."""
    from rohm_amd import ops
    N, K = 512, 512
    a, w = seeded(M + 1, M, K), seeded(K + 8, N, K) / math.sqrt(K)
    bias, g, b = seeded(3, N), seeded(5, N) * 0.5 + 1.0, seeded(6, N)
    res = seeded(4, M, N) + (seeded(9, M, 1) * 150.0)              # per-row offsets of up to ~600
    ref = nets.layer_norm(a.double() @ w.double().T + bias.double() + res.double(), g.double(), b.double())
    d = _dev()
    args = [t.to(d) for t in (a, w, bias, res, g, b)]
    out = ops.gemm_res_layernorm(*args)
    two = ops.layernorm_(ops.gemm(args[0], args[1], args[2], args[3], 2), args[4], args[5])
    e_fused, e_two = max_abs(out.cpu(), ref), max_abs(two.cpu(), ref)
    assert e_two < 5e-4 and e_fused < 5e-4, (e_fused, e_two)       # fp32 resolution of x ~ 600 is 6e-5: both paths sit at it


def test_gemm_res_layernorm_refuses_other_shapes():
    from rohm_amd import _lib, ops
    d = _dev()
    for M, N, K in ((100, 512, 512), (144, 320, 512), (144, 2048, 64)):
        a, w = torch.zeros(M, K, device=d), torch.zeros(N, K, device=d)
        with pytest.raises(_lib.RohmHipError):
            ops.gemm_res_layernorm(a, w, torch.zeros(N, device=d), torch.zeros(M, N, device=d), torch.ones(N, device=d),
                                   torch.zeros(N, device=d))


def _sk_plan(B, T=143, D=512, c_out=272):
    """gemm_f32.hip gemm_sk_plan: does the output head of this shape run as a stream-K launch?"""
    tiles, nk = -(-c_out // 144) * -(-(B * (T + 1)) // 64), D // 32
    if tiles % 8 or nk < 2 or (tiles // 8 * nk) % 32:
        return False
    u, rounds = tiles // 8 * nk // 32, -(-tiles // 256)
    return 2 * u >= nk and rounds * nk - u >= 4


@pytest.mark.parametrize('B', [64, 96, 128, 32, 80, 48, 3])      # stream-K: 64 (18 units per workgroup), 96 (27), 128 (36), 32 (9: tiles in up to three pieces); plain tiles: 80, 48, 3
def test_output_process(B):
    """OutputProcess.forward (model/heads.py:171-176) as PoseNet stores it (posenet.py:94-96): Linear 512 -> 272 of every frame token,
    written to channels 22.. of [B, 294, 1, 143].  At B = 64 the 288 tiles run as ONE round of 256 workgroups, each contracting 18
    consecutive (tile, K chunk) units; a tile cut in two is finished by the workgroup holding its tail (gemm_f32.hip stream-K)."""
    from rohm_amd import ops
    T, D, C, off, tot = 143, 512, 272, 22, 294
    h, w, b = seeded(B, B * (T + 1), D) * 2 + 0.1, seeded(B + 1, C, D) / math.sqrt(D), seeded(B + 2, C)
    ref = (h.double() @ w.double().T + b.double()).view(B, T + 1, C)[:, 1:].permute(0, 2, 1)          # [B, C, T]
    d = _dev()
    args = [t.to(d) for t in (h, w, b)]
    assert _sk_plan(B) == (B in (64, 96, 128, 32))
    outs, errs = {}, {}
    for sk in (True, False):
        out = torch.full((B, tot, 1, T), 7.5, device=d)
        out, scratch = ops.output_process(*args, B, T, ch_off=off, c_total=tot, out=out, stream_k=sk, return_scratch=True)
        torch.cuda.synchronize()
        o = out.cpu()
        errs[sk] = max_abs(o[:, off:, 0], ref)
        assert errs[sk] < 2e-5 * math.sqrt(D / 32)
        assert bool((o[:, :off] == 7.5).all()), 'the trajectory channels are not this operator\'s to write'
        if sk:
            assert int(scratch[0]) == 0, 'stream-K exchange error word'
        outs[sk] = out
    # the two launch shapes add the same products; a cut tile sums its two K ranges separately (|out| reaches ~10: one ulp is 1e-6)
    both = max_abs(outs[True].cpu(), outs[False].cpu())
    print(f'B={B}: stream-K {errs[True]:.2e}, tiles {errs[False]:.2e} vs fp64; between them {both:.2e}')
    assert both < 4e-5
    if not _sk_plan(B):
        assert torch.equal(outs[True], outs[False])
    for _ in range(50):      # race screen: every run the same bits (fixed owner, fixed order of the two partial sums)
        assert torch.equal(ops.output_process(*args, B, T, ch_off=off, c_total=tot, out=torch.full((B, tot, 1, T), 7.5, device=d)),
                           outs[True])


def test_output_process_other_widths():
    """Other head shapes (the class takes any latent_dim / feature count): ragged channel and token tiles, K of one chunk."""
    from rohm_amd import ops
    d = _dev()
    for B, T, D, C in ((5, 47, 256, 100), (64, 143, 256, 272), (40, 143, 64, 300), (2, 10, 32, 3)):
        h, w, b = seeded(B + T, B * (T + 1), D), seeded(C, C, D) / math.sqrt(D), seeded(D, C)
        ref = (h.double() @ w.double().T + b.double()).view(B, T + 1, C)[:, 1:].permute(0, 2, 1)
        out, scratch = ops.output_process(h.to(d), w.to(d), b.to(d), B, T, return_scratch=True)
        torch.cuda.synchronize()
        assert max_abs(out.cpu()[:, :, 0], ref) < 2e-5 * math.sqrt(max(D, 32) / 32), (B, T, D, C)
        assert int(scratch[0]) == 0


@pytest.mark.parametrize('M', [4, 144, 1000])
def test_layernorm(M):
    from rohm_amd import ops
    x, g, b = seeded(M, M, 512) * 3 + 0.5, seeded(1, 512), seeded(2, 512)
    ref = nets.layer_norm(x.double(), g.double(), b.double())
    d = _dev()
    out = ops.layernorm_(x.to(d), g.to(d), b.to(d))
    assert max_abs(out.cpu(), ref) < 5e-6


@pytest.mark.parametrize('n_seq,n_head', [(1, 4), (3, 4), (2, 2), (32, 4), (33, 4), (64, 4), (100, 4)])   # split / full launch shapes
def test_attention(n_seq, n_head):
    from rohm_amd import ops
    D = n_head * 128
    qkv = seeded(n_seq * 10 + n_head, n_seq * 144, 3 * D)
    q, k, v = qkv.double().view(n_seq, 144, 3, n_head, 128).permute(2, 0, 3, 1, 4)
    # the kernel expects q pre-scaled; scores up to ~|N(0,128)| exercise the max-subtraction
    ref = torch.softmax(q @ k.transpose(-1, -2), dim=-1) @ v
    ref = ref.permute(0, 2, 1, 3).reshape(n_seq * 144, D)
    out = ops.attention(qkv.to(_dev()), n_seq, n_head)
    # |score| reaches ~40 here, so fp32 rounding of the score itself (~2e-6) is amplified by exp
    assert max_abs(out.cpu(), ref) < 1e-4
    qkv2 = qkv.clone()
    qkv2[:, :D] *= 128 ** -0.5
    q2 = qkv2.double().view(n_seq, 144, 3, n_head, 128).permute(2, 0, 3, 1, 4)[0]
    ref2 = (torch.softmax(q2 @ k.transpose(-1, -2), dim=-1) @ v).permute(0, 2, 1, 3).reshape(n_seq * 144, D)
    out2 = ops.attention(qkv2.to(_dev()), n_seq, n_head)
    assert max_abs(out2.cpu(), ref2) < 5e-6


@pytest.mark.parametrize('n_seq,n_head,S,dh', [(2, 4, 100, 64), (1, 2, 77, 128), (3, 4, 144, 64), (2, 4, 200, 128),
                                                (1, 1, 5, 64), (2, 3, 64, 128), (1, 4, 145, 128)])
def test_attention_general_shapes(n_seq, n_head, S, dh):
    """Any sequence length, head dim 64 / 128 (the reference class takes any clip length and defaults to 4 x 64)."""
    from rohm_amd import ops
    D = n_head * dh
    qkv = seeded(n_seq * 7 + S, n_seq * S, 3 * D)
    qkv[:, :D] *= dh ** -0.5
    q, k, v = qkv.double().view(n_seq, S, 3, n_head, dh).permute(2, 0, 3, 1, 4)
    ref = (torch.softmax(q @ k.transpose(-1, -2), dim=-1) @ v).permute(0, 2, 1, 3).reshape(n_seq * S, D)
    out = ops.attention(qkv.to(_dev()), n_seq, n_head, S, dh)
    assert max_abs(out.cpu(), ref) < 5e-6
    # un-scaled scores (|s| up to ~40): the running-maximum rescaling is exercised
    qkv2 = seeded(n_seq * 7 + S + 1, n_seq * S, 3 * D)
    q, k, v = qkv2.double().view(n_seq, S, 3, n_head, dh).permute(2, 0, 3, 1, 4)
    ref = (torch.softmax(q @ k.transpose(-1, -2), dim=-1) @ v).permute(0, 2, 1, 3).reshape(n_seq * S, D)
    assert max_abs(ops.attention(qkv2.to(_dev()), n_seq, n_head, S, dh).cpu(), ref) < 1e-4


@pytest.mark.parametrize('n_seq', [64, 32, 7])
def test_attention_is_bit_reproducible(n_seq):
    """Race screen of the LDS-DMA / counted-vmcnt / raw-barrier pipeline (both launch shapes): the same launch repeated
    100 times must return the same bits every time."""
    from rohm_amd import ops
    qkv = seeded(n_seq, n_seq * 144, 1536).to(_dev())
    ref = ops.attention(qkv, n_seq, 4).clone()
    for _ in range(100):
        assert torch.equal(ops.attention(qkv, n_seq, 4), ref)


def test_ddpm_step_kernels():
    from rohm_amd import ops
    from oracle import diffusion as odiff
    d = _dev()
    B, n = 3, 294 * 143
    x_t, x0, nz, g1, g2 = (seeded(s, B, n) for s in range(5))
    out = ops.ddpm_step(x_t.to(d), x0.to(d), nz.to(d), 0.25, 0.75, 0.5, g1.to(d), 2.0)
    assert max_abs(out.cpu(), 0.25 * x0 + 0.75 * x_t + 2.0 * g1 + 0.5 * nz) < 1e-6
    tab = odiff.tables(odiff.cosine_betas(1000))
    import numpy as np
    tabs = np.stack([tab['coef1'], tab['coef2'], tab['variance'], tab['log_variance']], 1).astype(np.float32)
    t = torch.tensor([999, 0, 37])
    out = ops.ddpm_step_table(x_t.to(d), x0.to(d), nz.to(d), torch.from_numpy(tabs).to(d), t.to(d),
                              g1.to(d), 3e5, g2.to(d), 1e5)
    tt = torch.from_numpy(tabs)[t]
    ref = tt[:, 0:1] * x0 + tt[:, 1:2] * x_t + 3e5 * tt[:, 2:3] * g1 + 1e5 * tt[:, 2:3] * g2 \
        + (t != 0).float()[:, None] * torch.exp(0.5 * tt[:, 3:4]) * nz
    assert max_abs(out.cpu(), ref) < 1e-4 * float(ref.abs().max())
