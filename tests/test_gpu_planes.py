"""GPU: the opt-in split-bf16 path kernel by kernel, through the C ABI.

Producers of bf16 planes (rohm_planes_split, the LayerNorm / attention / GELU-GEMM plane outputs) are held BIT-EXACTLY to
"cut of the fp32 result" (oracle/planes.py restates cut and layout); the GEMM on planes is held to a float64 evaluation
at the exact-fp32 kernel's own bar (tests/test_gpu_kernels.py::test_gemm) for three planes, and to the north-star class
for two.  The whole-network proofs (reference goldens, 1000 steps, 64 clips) are in tests/test_gpu_precision_ladder.py."""
import math
import os

import numpy as np
import pytest
import torch

from helpers import max_abs, seeded
from oracle import nets
from oracle import planes as oplanes

pytestmark = pytest.mark.gpu


def _dev():
    return torch.device('cuda', 0)


@pytest.mark.parametrize('nplane', [3, 2, 16])
def test_planes_split_is_the_cut_in_fragment_major_layout(nplane):
    from rohm_amd import ops
    x = seeded(11, 144 * 2, 96) * 3.0
    x[0, :8] = torch.tensor([0.0, -0.0, 1.0, -1.0, 1e-20, 3e20 if nplane != 16 else 6.0e4, 0.333333343, -7.0])
    x[1, :4] = torch.tensor([1e-5, -3e-6, 6.1e-5, 2049.0])          # fp16 subnormal range / a tie of round-to-nearest-even
    got = ops.planes_split(x.to(_dev()), nplane).cpu().numpy()
    assert np.array_equal(got, oplanes.encode(x.numpy(), nplane))
    if nplane == 3:
        assert np.array_equal(oplanes.decode(got, 288, 96, 3).astype(np.float64).sum(0), x.numpy().astype(np.float64))
    if nplane == 16:
        rec = oplanes.value(oplanes.decode(got, 288, 96, 16), 16)
        assert np.abs(rec - x.numpy().astype(np.float64)).max() <= 2.0 ** -22 * np.abs(x.numpy()).max()


# shapes: one tile / several; K chunk counts 1, 2, 3, 4, 16, 17, 18, 32 (every remainder of the 3-stage ring);
# 144 x 128 tiles (>= 192 of them) and 144 x 64 tiles (fewer)
# -- and for the persistent stream kernel (K >= 192): one, two and three tiles per workgroup, uneven tile counts (260, 320, 528
# tiles on 256 workgroups), tiles of five to seven chunks
_ALL_SHAPES = [(144, 64, 32), (144, 128, 64), (288, 192, 96), (144, 512, 128), (144 * 3, 512, 512), (144 * 2, 512, 544),
          (144 * 2, 256, 576), (144 * 3, 512, 1024), (144 * 64, 512, 512), (144 * 64, 1536, 512), (144 * 64, 1024, 512),
          (144 * 64, 512, 1024), (144 * 32, 512, 512), (144 * 32, 1536, 512), (144 * 65, 512, 192), (144 * 40, 512, 224),
          (144 * 66, 1024, 160), (144 * 66, 1024, 512), (144 * 65, 512, 320), (144 * 64, 1536, 352)]
# the child run of the one-workgroup-per-tile kernel only needs the shapes the default path gives to the stream kernel (K >= 192)
SHAPES = [sh for sh in _ALL_SHAPES if sh[2] >= 192 and sh[0] * sh[1] <= 144 * 66 * 1024] if os.environ.get('ROHM_PP_STREAM') == '0' else _ALL_SHAPES


@pytest.mark.parametrize('M,N,K', SHAPES)
@pytest.mark.parametrize('epi', [0, 1, 2, 3])
@pytest.mark.parametrize('mode', [3, 16])
def test_gemm_planes_bf16x6_and_fp16x3_meet_the_fp32_bar(M, N, K, epi, mode):
    from rohm_amd import ops
    if M * N > 144 * 64 * 512 and epi in (0, 3) and K > 512:
        pytest.skip('covered by the other epilogues at this size')
    a, w = seeded(M + N, M, K), seeded(K + 7, N, K) / math.sqrt(K)
    bias, res = seeded(3, N), seeded(4, M, N)
    ref = a.double() @ w.double().T + bias.double()
    if epi == 1:
        ref = nets.gelu_erf(ref)
    if epi == 2:
        ref = ref + res.double()
    if epi == 3:
        ref[:, :N // 2] *= 0.25
    d = _dev()
    ws = 256.0 if mode == 16 else 1.0      # fp16 weight planes are cut from 2^8 w, the accumulator is scaled back (posenet.hip)
    ap, wp = ops.planes_split(a.to(d), mode), ops.planes_split(w.to(d), mode, scale=ws)
    out, _ = ops.gemm_planes(ap, wp, M, N, K, mode, bias.to(d), res.to(d) if epi == 2 else None, epi, qcols=N // 2, qscale=0.25,
                             acc_scale=1.0 / ws)
    assert max_abs(out.cpu(), ref) < 2e-5 * math.sqrt(K / 32)       # the exact-fp32 kernel's bar


@pytest.mark.parametrize('M,N,K', [(144 * 3, 512, 512), (144 * 64, 512, 512), (144 * 2, 192, 1024)])
def test_gemm_planes_bf16x3_is_2_pow_minus_16_class(M, N, K):
    from rohm_amd import ops
    a, w = seeded(M + N, M, K), seeded(K + 7, N, K) / math.sqrt(K)
    ref = a.double() @ w.double().T
    d = _dev()
    out, _ = ops.gemm_planes(ops.planes_split(a.to(d), 2), ops.planes_split(w.to(d), 2), M, N, K, 2)
    err = max_abs(out.cpu(), ref)
    assert err < 3e-4 * math.sqrt(K / 512), err


def test_gemm_planes_identity_asymmetric():
    """A = I against an asymmetric W: catches transposed or permuted output tiles and a wrong k order inside a fragment."""
    from rohm_amd import ops
    d = _dev()
    a = torch.eye(144, 192).contiguous()
    w = (torch.arange(128 * 192, dtype=torch.float32).reshape(128, 192) / 128.0).contiguous()      # exact in 3 planes
    out, _ = ops.gemm_planes(ops.planes_split(a.to(d)), ops.planes_split(w.to(d)), 144, 128, 192)
    assert torch.equal(out.cpu(), w[:, :144].T.contiguous())


@pytest.mark.parametrize('M,N,K', [(144 * 64, 1024, 512), (144 * 2, 1024, 512), (144 * 64, 512, 64), (144 * 65, 1024, 192),
                                   (144 * 40, 1024, 224), (144 * 65, 1024, 320)])
@pytest.mark.parametrize('nplane', [3, 2, 16])
@pytest.mark.parametrize('flags', [0, 1])
def test_gemm_plane_output_is_the_cut_of_the_fp32_output(M, N, K, nplane, flags):
    """The GELU GEMM that feeds FF2 writes planes only; both store forms (lane-swapped 16-byte units / 8-byte halves) and
    both tile widths must give exactly cut(fp32 result)."""
    from rohm_amd import ops
    a, w, bias = seeded(M + N, M, K), seeded(K + 7, N, K) / math.sqrt(K), seeded(3, N)
    d = _dev()
    ap, wp = ops.planes_split(a.to(d), nplane), ops.planes_split(w.to(d), nplane)
    out, cp = ops.gemm_planes(ap, wp, M, N, K, nplane, bias.to(d), None, 1, out_planes=True, flags=flags)
    only, cp2 = ops.gemm_planes(ap, wp, M, N, K, nplane, bias.to(d), None, 1, out_f32=False, out_planes=True, flags=flags)
    assert only is None and torch.equal(cp, cp2)
    assert np.array_equal(cp.cpu().numpy(), oplanes.encode(out.cpu().numpy(), nplane))


@pytest.mark.parametrize('M', [16, 144, 144 * 7])
@pytest.mark.parametrize('nplane', [3, 2, 16])
def test_layernorm_planes(M, nplane):
    from rohm_amd import ops
    x, g, b = seeded(M, M, 512) * 3 + 0.5, seeded(1, 512), seeded(2, 512)
    d = _dev()
    plain = ops.layernorm_(x.to(d), g.to(d), b.to(d))
    xx = x.to(d)
    pl = ops.layernorm_planes_(xx, g.to(d), b.to(d), nplane)
    assert torch.equal(xx, plain)                      # same arithmetic, same order
    assert np.array_equal(pl.cpu().numpy(), oplanes.encode(plain.cpu().numpy(), nplane))
    assert max_abs(plain.cpu(), nets.layer_norm(x.double(), g.double(), b.double())) < 5e-6


@pytest.mark.parametrize('n_seq,n_head', [(1, 4), (3, 4), (32, 4), (33, 4), (64, 4)])      # split / full launch shapes
@pytest.mark.parametrize('nplane', [3, 2, 16])
def test_attention_planes(n_seq, n_head, nplane):
    """Plane output runs the P.V MFMAs with exchanged operands (the same products in the same order): the planes must be the
    cut of what the fp32 kernel stores."""
    from rohm_amd import ops
    D = n_head * 128
    qkv = seeded(n_seq * 10 + n_head, n_seq * 144, 3 * D).to(_dev())
    ctx = ops.attention(qkv, n_seq, n_head)
    pl = ops.attention_planes(qkv, n_seq, n_head, nplane).cpu().numpy()
    dec = oplanes.decode(pl, n_seq * 144, D, nplane)
    want = np.stack(oplanes.cut(ctx.cpu().numpy(), nplane))
    if not np.array_equal(dec, want):                  # tolerate a different rounding of the exchanged MFMA, nothing more
        assert nplane in (3, 16)
        assert np.abs(oplanes.value(dec, nplane) - ctx.cpu().numpy()).max() < 2e-6


def test_shape_errors_are_raised_before_any_launch():
    from rohm_amd import _lib, ops
    d = _dev()
    with pytest.raises(_lib.RohmHipError):
        ops.planes_split(torch.zeros(10, 32, device=d))                 # rows % 16
    ap, wp = ops.planes_split(torch.zeros(144, 32, device=d)), ops.planes_split(torch.zeros(64, 32, device=d))
    with pytest.raises(_lib.RohmHipError):
        ops.gemm_planes(ap, wp, 100, 64, 32)                            # M % 144
    with pytest.raises(_lib.RohmHipError):
        ops.gemm_planes(ap, wp, 144, 64, 32, nplane=4)
    with pytest.raises(_lib.RohmHipError):
        ops.planes_split(torch.zeros(16, 32, device=d), 8)


def test_the_one_tile_per_workgroup_kernel_too():
    """ROHM_PP_STREAM=0 (read at the first launch -> a child process) selects gemm_pp_kernel, which the default path only uses
    for K < 192: the GEMM tests of this file must hold for it as well."""
    import subprocess
    import sys
    if 'ROHM_PP_STREAM' in os.environ:
        pytest.skip('already the child')
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, '-m', 'pytest', '-q', '-x', '-m', 'gpu', '-p', 'no:cacheprovider', 'tests/test_gpu_planes.py',
                        '-k', 'gemm and not ln_fold'], cwd=root, env=dict(os.environ, ROHM_PP_STREAM='0'), capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-1000:]


def _row_stats(x):
    """[M, D] -> the partial (sum, sum of squares) pairs per 16 columns in the layout of rohm_gemm_planes_ln: [M/16, D/16, 16, 2]."""
    m, d = x.shape
    g = x.double().reshape(m // 16, 16, d // 16, 16)
    return torch.stack([g.sum(-1), (g * g).sum(-1)], -1).permute(0, 2, 1, 3).contiguous()


@pytest.mark.parametrize('M,N,K', [(144 * 3, 512, 512), (144 * 64, 512, 1024), (144 * 65, 1024, 192), (144 * 40, 256, 224)])
@pytest.mark.parametrize('mode', [16, 2])
def test_gemm_ln_fold_producer_writes_row_statistics_of_its_result(M, N, K, mode):
    """epilogue 2 + out_stats: C is what the plain call gives, bit for bit, and the statistics are the row sums of C."""
    from rohm_amd import ops
    a, w, bias, res = seeded(M + N, M, K), seeded(K + 7, N, K) / math.sqrt(K), seeded(3, N), seeded(4, M, N)
    d = _dev()
    ws = 256.0 if mode == 16 else 1.0
    ap, wp = ops.planes_split(a.to(d), mode), ops.planes_split(w.to(d), mode, scale=ws)
    plain, _ = ops.gemm_planes(ap, wp, M, N, K, mode, bias.to(d), res.to(d), 2, acc_scale=1.0 / ws)
    c, cp, st = ops.gemm_planes_ln(ap, wp, M, N, K, mode, bias.to(d), res.to(d), 2, acc_scale=1.0 / ws, want_stats=True, ln_dim=N,
                                   out_planes=True)
    assert torch.equal(c, plain)
    assert np.array_equal(cp.cpu().numpy(), oplanes.encode(c.cpu().numpy(), mode))
    want = _row_stats(c.cpu())
    assert max_abs(st.cpu(), want) < 1e-5 * float(want.abs().max())


@pytest.mark.parametrize('M,N,K', [(144 * 3, 1536, 512), (144 * 64, 1024, 512), (144 * 65, 512, 256), (144 * 40, 1024, 384)])
@pytest.mark.parametrize('epi', [1, 3])
def test_gemm_ln_fold_consumer_equals_layernorm_then_gemm(M, N, K, epi):
    """ln_stats: planes of RAW x, gamma folded into the weight planes, c / d vectors -> LN(x) W^T + b at the fp32 bar
    (x has a row mean of about a third of its spread, as the residual stream of the network has)."""
    from rohm_amd import ops
    x = seeded(M + N, M, K) * 1.5 + 0.5
    w, bias, gam, bet = seeded(K + 7, N, K) / math.sqrt(K), seeded(3, N), 1.0 + 0.2 * seeded(5, K), 0.1 * seeded(6, K)
    ref = nets.layer_norm(x.double(), gam.double(), bet.double()) @ w.double().T + bias.double()
    if epi == 1:
        ref = nets.gelu_erf(ref)
    else:
        ref[:, :N // 2] *= 0.25
    wg = (w.double() * gam.double()).float()
    cvec, dvec = wg.double().sum(1).float(), (bias.double() + w.double() @ bet.double()).float()
    d = _dev()
    c, _, _ = ops.gemm_planes_ln(ops.planes_split(x.to(d), 16), ops.planes_split(wg.to(d), 16, scale=256.0), M, N, K, 16, dvec.to(d),
                                 None, epi, qcols=N // 2, qscale=0.25, acc_scale=1.0 / 256.0, ln_stats=_row_stats(x).float().to(d),
                                 ln_c=cvec.to(d), ln_dim=K)
    assert max_abs(c.cpu(), ref) < 4e-5 * math.sqrt(K / 32)          # twice the plain bar: (acc - mu c) rstd cancels mu c


@pytest.mark.parametrize('M,N,K', [(144 * 3, 512, 512), (144 * 64, 512, 1024), (144 * 65, 512, 192), (144 * 40, 256, 224)])
def test_gemm_ln_fold_residual_is_normalised_on_the_fly(M, N, K):
    from rohm_amd import ops
    a, w, bias = seeded(M + N, M, K), seeded(K + 7, N, K) / math.sqrt(K), seeded(3, N)
    raw, gam, bet = seeded(4, M, N) * 1.5 + 0.5, 1.0 + 0.2 * seeded(5, N), 0.1 * seeded(6, N)
    ref = a.double() @ w.double().T + bias.double() + nets.layer_norm(raw.double(), gam.double(), bet.double())
    d = _dev()
    c, _, st = ops.gemm_planes_ln(ops.planes_split(a.to(d), 16), ops.planes_split(w.to(d), 16, scale=256.0), M, N, K, 16, bias.to(d),
                                  raw.to(d), 2, acc_scale=1.0 / 256.0, r_stats=_row_stats(raw).float().to(d), r_gamma=gam.to(d),
                                  r_beta=bet.to(d), want_stats=True, ln_dim=N)
    assert max_abs(c.cpu(), ref) < 2e-5 * math.sqrt(K / 32)
    want = _row_stats(c.cpu())
    assert max_abs(st.cpu(), want) < 1e-5 * float(want.abs().max())


def test_gemm_ln_fold_refuses_what_it_cannot_do():
    from rohm_amd import _lib, ops
    d = _dev()
    ap, wp = ops.planes_split(torch.zeros(144, 160, device=d), 16), ops.planes_split(torch.zeros(128, 160, device=d), 16)
    z = torch.zeros(128, device=d)
    with pytest.raises(_lib.RohmHipError):          # K = 160 < 192: no stream kernel
        ops.gemm_planes_ln(ap, wp, 144, 128, 160, 16, z, torch.zeros(144, 128, device=d), 2, want_stats=True, ln_dim=128)
    ap, wp = ops.planes_split(torch.zeros(144, 224, device=d), 16), ops.planes_split(torch.zeros(128, 224, device=d), 16)
    with pytest.raises(_lib.RohmHipError):          # two statistics slots of 144 x 1024 / 16 pairs do not fit the 160 KiB of LDS
        ops.gemm_planes_ln(ap, ops.planes_split(torch.zeros(1024, 224, device=d), 16), 144, 1024, 224, 16, torch.zeros(1024, device=d),
                           torch.zeros(144, 1024, device=d), 2, r_stats=torch.zeros(9, 64, 16, 2, device=d),
                           r_gamma=torch.zeros(1024, device=d), r_beta=torch.zeros(1024, device=d), ln_dim=1024)
    with pytest.raises(_lib.RohmHipError):          # ln_dim = K = 224 is not a multiple of 128
        ops.gemm_planes_ln(ap, wp, 144, 128, 224, 16, z, None, 1, ln_stats=torch.zeros(9, 14, 16, 2, device=d), ln_c=z, ln_dim=224)
