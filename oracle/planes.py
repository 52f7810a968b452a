"""TEST INFRASTRUCTURE (never imported by rohm_amd/): numpy restatement of the bf16 PLANES behind the opt-in split-bf16
GEMM -- the cut (rohm_amd/csrc/planes.h::plane_cut4) and the fragment-major layout (planes.h::plane_unit) -- so that the
GPU tests can hold every producer (rohm_planes_split, the LayerNorm / attention / GELU-GEMM plane outputs) bit-exactly to
"cut of the fp32 result".  There is no reference counterpart: the reference computes these Linears in fp32
(model/posenet.py:63-69); the planes are an implementation detail of the labelled second line."""
import numpy as np


def cut(x, nplane=3):
    """fp32 array -> nplane fp32 arrays, each a bf16 value: truncation planes (x = h + m + l exactly for nplane = 3)."""
    x = np.ascontiguousarray(x, np.float32)
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
    pl = np.stack([(p.view(np.uint32) >> 16).astype(np.uint16) for p in cut(x, nplane)])        # [np, rows, K]
    # unit index = ((rb * nkc + kc) * np + p) * 64 + g * 16 + i ; 8 bf16 per unit
    v = pl.reshape(nplane, rows // 16, 16, K // 32, 4, 8)          # p, rb, i, kc, g, e
    return np.ascontiguousarray(v.transpose(1, 3, 0, 4, 2, 5)).reshape(-1).view(np.int16)


def decode(buf, rows, K, nplane=3):
    """int16 buffer in the fragment-major layout -> fp32 planes [nplane, rows, K]."""
    v = np.asarray(buf).view(np.uint16).reshape(rows // 16, K // 32, nplane, 4, 16, 8)     # rb, kc, p, g, i, e
    pl = np.ascontiguousarray(v.transpose(2, 0, 4, 1, 3, 5)).reshape(nplane, rows, K)
    return (pl.astype(np.uint32) << 16).view(np.float32)
