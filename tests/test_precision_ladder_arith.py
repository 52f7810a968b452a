"""CPU check of the arithmetic behind the opt-in split-bf16 GEMM (rohm_amd/csrc/gemm_pp.hip, planes.h): the truncation planes
h = upper 16 bits of x, m = upper 16 bits of (x - h), l = x - h - m reproduce x EXACTLY, every plane is a bf16 value, and
the six (three) plane products kept by bf16x6 (bf16x3) leave an error of fp32-rounding size (~2^-16) on a dot product.
The kernel's own results are held to the fp32 kernel's bars by tests/test_gpu_precision_ladder.py on the GPU."""
import numpy as np


def planes(x):
    x = np.asarray(x, np.float32)
    h = (x.view(np.uint32) & np.uint32(0xffff0000)).view(np.float32)
    r1 = (x - h).astype(np.float32)
    m = (r1.view(np.uint32) & np.uint32(0xffff0000)).view(np.float32)
    l = (r1 - m).astype(np.float32)
    return h, m, l


def is_bf16(v):
    return np.all((np.asarray(v, np.float32).view(np.uint32) & np.uint32(0xffff)) == 0)


def test_three_truncation_planes_are_exact_and_bf16():
    rng = np.random.default_rng(0)
    x = np.concatenate([rng.standard_normal(200000), rng.standard_normal(1000) * 1e-25, rng.standard_normal(1000) * 1e30,
                        [0.0, -0.0, 1.0, -1.0, 3.4e38]]).astype(np.float32)
    h, m, l = planes(x)
    assert is_bf16(h) and is_bf16(m) and is_bf16(l)
    assert np.array_equal((h.astype(np.float64) + m.astype(np.float64) + l.astype(np.float64)).astype(np.float32), x)
    assert np.array_equal(h.astype(np.float64) + m.astype(np.float64) + l.astype(np.float64), x.astype(np.float64))
    # (below ~2^-100 the remainders become denormal: the planes still sum to x, but l may need more than bf16's 8 bits --
    # twenty orders of magnitude under anything the networks produce)
    tiny = np.float32(4.6645464e-35)
    th, tm, tl = planes(np.asarray([tiny]))
    assert float(th[0]) + float(tm[0]) + float(tl[0]) == float(tiny)
    # two planes: relative error of x - (h + m) below 2^-15 (8 + 8 significant bits kept, truncation)
    nz = x != 0
    assert np.max(np.abs(l[nz].astype(np.float64) / x[nz])) < 2.0 ** -15


def test_six_products_are_fp32_class_three_are_2_pow_minus_16():
    rng = np.random.default_rng(1)
    K = 1024
    a = rng.standard_normal((64, K)).astype(np.float32)
    w = rng.standard_normal((64, K)).astype(np.float32)
    exact = np.einsum('ik,jk->ij', a.astype(np.float64), w.astype(np.float64))
    pa, pw = planes(a), planes(w)

    def emu(pairs):        # every bf16 x bf16 product is exact in fp32; sums taken in float64 to isolate the truncation error
        return sum(np.einsum('ik,jk->ij', pa[i].astype(np.float64), pw[j].astype(np.float64)) for i, j in pairs)
    six = emu([(0, 0), (0, 1), (1, 0), (1, 1), (0, 2), (2, 0)])
    three = emu([(0, 0), (0, 1), (1, 0)])
    scale = np.sqrt(K)     # size of a typical dot product of unit normals
    fp32 = (a @ w.T).astype(np.float64)
    e6, e3, e32 = np.abs(six - exact).max() / scale, np.abs(three - exact).max() / scale, np.abs(fp32 - exact).max() / scale
    assert e6 < 2.0 ** -21, e6                 # dropped m.l + l.m + l.l ~ 2^-23 per product: below an fp32 GEMM's own rounding
    assert e6 < 4 * max(e32, 2.0 ** -24)
    assert 2.0 ** -22 < e3 < 2.0 ** -13, e3    # three products: ~2^-16 per product, summed over K


def test_plane_layout_restatement_round_trips():
    """oracle/planes.py (the checker of the GPU plane producers): encode -> decode gives the cut back, the cut sums to x,
    and the unit index is the one planes.h states."""
    from oracle import planes as op
    rng = np.random.default_rng(2)
    for nplane in (3, 2):
        x = rng.standard_normal((48, 96)).astype(np.float32)
        buf = op.encode(x, nplane)
        assert buf.size * 2 == 48 * 96 * 2 * nplane
        dec = op.decode(buf, 48, 96, nplane)
        for got, want in zip(dec, op.cut(x, nplane)):
            assert np.array_equal(got, want)
        if nplane == 3:
            assert np.array_equal(dec.astype(np.float64).sum(0), x.astype(np.float64))
        # spot-check the address arithmetic: element (row 37, k 70) of plane 1
        row, k, p = 37, 70, 1
        unit = (((row // 16) * (96 // 32) + k // 32) * nplane + p) * 64 + ((k % 32) // 8) * 16 + row % 16
        got = (np.uint32(buf.view(np.uint16)[unit * 8 + k % 8]) << np.uint32(16)).view(np.float32)
        assert got == op.cut(x, nplane)[p][row, k]
    h, m = op.cut(np.float32([1.2345678]), 2)
    assert is_bf16(h) and is_bf16(m) and abs(float(h[0]) + float(m[0]) - 1.2345678) < 1.2345678 * 2.0 ** -15


def test_fp16_two_plane_split_is_fp32_class_with_three_products():
    """The fp16x3 mode (planes.h MODE 16): h = fp16(x), l' = fp16((x - h) 2^11) reproduce x to 2^-24 |x| over fp16's normal range, and
    the three products it keeps (a_h w_h; a_h w_l' + a_l' w_h at weight 2^-11) leave ~2^-22 per product -- between bf16x6
    (2^-23) and fp32's own rounding, four orders of magnitude below bf16x3."""
    from oracle import planes as op
    rng = np.random.default_rng(3)
    x = np.concatenate([rng.standard_normal(200000) * 3, rng.standard_normal(2000) * 1e-3, rng.uniform(-6e4, 6e4, 2000)]).astype(np.float32)
    h, lp = op.cut(x, 16)
    assert np.all(h == h.astype(np.float16).astype(np.float32)) and np.all(lp == lp.astype(np.float16).astype(np.float32))
    ok = np.abs(x) >= 6.2e-5
    rel = np.abs(op.value(np.stack([h, lp]), 16) - x.astype(np.float64))[ok] / np.abs(x[ok])
    assert rel.max() < 2.0 ** -22 and np.percentile(rel, 99) < 2.0 ** -23      # l' carries 11 more bits behind h's 11
    assert np.abs(lp).max() <= np.abs(x).max()                                 # scaled remainder sits next to h: no underflow, no overflow
    K = 1024
    a = rng.standard_normal((64, K)).astype(np.float32)
    w = (rng.standard_normal((64, K)) * 0.03).astype(np.float32)
    exact = np.einsum('ik,jk->ij', a.astype(np.float64), w.astype(np.float64))
    (ah, al), (wh, wl) = op.cut(a, 16), op.cut(w * np.float32(256.0), 16)      # weights are cut from 2^8 w (posenet.hip)
    d = lambda p, q: np.einsum('ik,jk->ij', p.astype(np.float64), q.astype(np.float64))
    emu = (d(ah, wh) + (d(ah, wl) + d(al, wh)) / 2048.0) / 256.0
    scale = np.sqrt(K) * 0.03
    e16 = np.abs(emu - exact).max() / scale
    e32 = np.abs((a @ w.T).astype(np.float64) - exact).max() / scale
    assert e16 < 2.0 ** -20, e16
    assert e16 < 8 * max(e32, 2.0 ** -24)


def test_layernorm_fold_algebra_and_statistics_layout():
    """The LayerNorm fold of the two-plane modes (gemm_pp.hip, posenet.hip): with c_n = sum_k gamma_k W_nk and
    d_n = b_n + sum_k beta_k W_nk,  (x W_gamma^T - mu c) rstd + d  ==  LN(x) W^T + b,  where (mu, rstd) come from the partial
    (sum, sum of squares) pairs per 16 columns that a producer GEMM writes in the layout [row / 16][D / 16][row % 16][2]."""
    rng = np.random.default_rng(3)
    M, D, N = 48, 512, 96
    x = (rng.standard_normal((M, D)) * 1.5 + 0.5)
    w, b = rng.standard_normal((N, D)) / np.sqrt(D), rng.standard_normal(N)
    gam, bet = 1.0 + 0.2 * rng.standard_normal(D), 0.1 * rng.standard_normal(D)
    mu, var = x.mean(-1, keepdims=True), x.var(-1, keepdims=True)
    want = ((x - mu) / np.sqrt(var + 1e-5) * gam + bet) @ w.T + b
    # the producer's statistics: per row and 16-column part, in the kernel's layout
    parts = x.reshape(M // 16, 16, D // 16, 16)
    stats = np.stack([parts.sum(-1), (parts ** 2).sum(-1)], -1).transpose(0, 2, 1, 3)        # [M/16][D/16][16][2]
    assert stats.shape == (M // 16, D // 16, 16, 2)
    su = stats[..., 0].sum(1).reshape(M)          # over the parts: [M/16][16] -> rows in order
    sq = stats[..., 1].sum(1).reshape(M)
    mu_k = su / D
    rstd_k = 1.0 / np.sqrt(np.maximum(sq / D - mu_k ** 2, 0.0) + 1e-5)      # biased variance, eps inside the root, clamp as the kernel does
    assert np.allclose(mu_k, mu[:, 0], rtol=0, atol=1e-12) and np.allclose(rstd_k, 1.0 / np.sqrt(var[:, 0] + 1e-5), rtol=1e-10)
    wg = w * gam
    c, d = wg.sum(1), b + w @ bet
    got = ((x @ wg.T) - mu_k[:, None] * c) * rstd_k[:, None] + d
    assert np.abs(got - want).max() < 1e-11
    # in float32 the cancellation (acc - mu c) costs about |mu| / sigma ulps of the accumulator: the GPU test allows twice the plain bar
    x32, wg32 = x.astype(np.float32), wg.astype(np.float32)
    got32 = ((x32 @ wg32.T) - mu_k.astype(np.float32)[:, None] * c.astype(np.float32)) * rstd_k.astype(np.float32)[:, None] + d.astype(np.float32)
    assert np.abs(got32 - want).max() < 4e-5 * np.sqrt(D / 32)
