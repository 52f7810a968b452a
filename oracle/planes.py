"""TEST INFRASTRUCTURE (never imported by rohm_amd/): numpy restatement of the bf16 PLANES behind the opt-in split-bf16
GEMM -- the cut (rohm_amd/csrc/planes.h::plane_cut4) and the fragment-major layout (planes.h::plane_unit) -- so that the
GPU tests can hold every producer (rohm_planes_split, the LayerNorm / attention / GELU-GEMM plane outputs) bit-exactly to
"cut of the fp32 result".  There is no reference counterpart: the reference computes these Linears in fp32
(model/posenet.py:63-69); the planes are an implementation detail of the labelled second line."""
import numpy as np


F16_MODE, F16_LOW_SCALE = 16, 2048.0


def n_planes(mode):
    return 2 if mode == F16_MODE else mode


def cut(x, nplane=3):
    """fp32 array -> the planes as fp32 arrays.  nplane 3 / 2: bf16 values by truncation (x = h + m + l exactly for 3);
    16: two FP16 values h = fp16(x) (round to nearest even), l' = fp16((x - h) * 2^11) (planes.h MODE 16)."""
    x = np.ascontiguousarray(x, np.float32)
    if nplane == F16_MODE:
        h = x.astype(np.float16)
        lp = ((x - h.astype(np.float32)) * np.float32(F16_LOW_SCALE)).astype(np.float16)
        return [h.astype(np.float32), lp.astype(np.float32)]
    h = (x.view(np.uint32) & np.uint32(0xffff0000)).view(np.float32)
    r1 = (x - h).astype(np.float32)
    if nplane == 2:
        return [h, (r1.view(np.uint32) & np.uint32(0xffff0000)).view(np.float32)]      # the second plane truncates r1
    m = (r1.view(np.uint32) & np.uint32(0xffff0000)).view(np.float32)
    return [h, m, (r1 - m).astype(np.float32)]


def encode(x, nplane=3):
    """fp32 [rows, K] -> the int16 buffer rohm_planes_split produces (rows % 16 == 0, K % 32 == 0)."""
    rows, K = x.shape
    assert rows % 16 == 0 and K % 32 == 0
    if nplane == F16_MODE:
        pl = np.stack([p.astype(np.float16).view(np.uint16) for p in cut(x, nplane)])
    else:
        pl = np.stack([(p.view(np.uint32) >> 16).astype(np.uint16) for p in cut(x, nplane)])    # [np, rows, K]
    # unit index = ((rb * nkc + kc) * np + p) * 64 + g * 16 + i ; 8 x 16 bit per unit
    v = pl.reshape(n_planes(nplane), rows // 16, 16, K // 32, 4, 8)          # p, rb, i, kc, g, e
    return np.ascontiguousarray(v.transpose(1, 3, 0, 4, 2, 5)).reshape(-1).view(np.int16)


def decode(buf, rows, K, nplane=3):
    """int16 buffer in the fragment-major layout -> fp32 planes [nplane, rows, K]."""
    npl = n_planes(nplane)
    v = np.asarray(buf).view(np.uint16).reshape(rows // 16, K // 32, npl, 4, 16, 8)     # rb, kc, p, g, i, e
    pl = np.ascontiguousarray(v.transpose(2, 0, 4, 1, 3, 5)).reshape(npl, rows, K)
    if nplane == F16_MODE:
        return pl.view(np.float16).astype(np.float32)
    return (pl.astype(np.uint32) << 16).view(np.float32)


def value(planes, nplane=3):
    """What a set of decoded planes stands for: h + m (+ l), or h + 2^-11 l' in the fp16 mode (float64)."""
    p = np.asarray(planes, np.float64)
    return p[0] + p[1] / F16_LOW_SCALE if nplane == F16_MODE else p.sum(0)
