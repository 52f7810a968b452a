// 16-bit PLANES (bf16 or fp16) of fp32 tensors (opt-in precision ladder): layout and the cut, shared by every kernel that
// produces or consumes them (gemm_pp.hip, elementwise.hip LayerNorm, attention_f32.hip, posenet.hip).
//
// MODE 3 / 2 -- bf16 planes cut by truncation:  h = upper 16 bits of x,  m = upper 16 bits of x - h,  l = x - h - m.  Every
// remainder is exact in fp32 and each plane carries 8 significant bits, so x = h + m + l EXACTLY for |x| >= 2^-100
// (tests/test_precision_ladder_arith.py).  Two planes: x = h + m up to 2^-15 |x|.
// MODE 16 -- two FP16 planes (fp16 carries 11 significant bits):  h = fp16(x) (round to nearest),  l' = fp16((x - h) * 2^11): the
// remainder x - h is exact in fp32 and at most 2^-12 |x|, scaled by 2^11 it sits next to h in magnitude (no fp16 underflow), and
// x = h + 2^-11 l' up to 2^-24 |x| -- fp32's own resolution -- for 6e-5 <= |x| <= 65504.  A product a.w then needs THREE fp16 MFMA
// products: a_h w_h into one accumulator, a_h w_l' + a_l' w_h into a second one that enters with weight 2^-11 (the dropped
// a_l' w_l' term is 2^-22 of the product).  Half the matrix-core work of six bf16 products at ~fp32 accuracy.
//
// Layout ("fragment-major"), for a matrix X[rows][K], rows % 16 == 0, K % 32 == 0, NP planes:
//   16-byte unit (row block rb = row / 16, K chunk kc = k / 32, plane p, lane = ((k % 32) / 8) * 16 + row % 16)
//   holds the 8 bf16 of k = 8 (k / 8) .. + 7;   unit index = ((rb * (K / 32) + kc) * NP + p) * 64 + lane.
// A (row block, chunk, plane) is 1 KiB = exactly the operand fragment of v_mfma_f32_16x16x32_bf16 (lane (i, g) holds
// row i, k-group g), so the consuming GEMM moves it with ONE LDS-DMA instruction (contiguous source, linear LDS image) and
// reads it back with ONE conflict-free ds_read_b128 -- or, for the weights, loads it straight into registers; the row
// strip of a row block is contiguous over K.
#pragma once
#include "common.h"

namespace rohm {

typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));

constexpr int kModeF16 = 16;                       // "nplane" / mode values: 3 = bf16x6, 2 = bf16x3, 16 = fp16x3
constexpr int mode_planes(int mode) { return mode == kModeF16 ? 2 : mode; }
inline bool mode_ok(int mode) { return mode == 2 || mode == 3 || mode == kModeF16; }
constexpr float kF16LowScale = 2048.0f;             // 2^11: the scale of the low fp16 plane

inline size_t plane_tensor_bytes(int rows, int K, int mode) {
    return (size_t)((rows + 15) / 16) * 16 * (size_t)K * 2 * (size_t)mode_planes(mode);
}

#ifdef __HIPCC__
// 16-byte unit index of (row, k-group kg = k / 8)
__device__ __forceinline__ size_t plane_unit(int row, int kg, int nkc, int np, int pl) {
    return ((size_t)((row >> 4) * nkc + (kg >> 2)) * np + pl) * 64 + (kg & 3) * 16 + (row & 15);
}

// four fp32 -> two packed dwords (4 x 16 bit) per plane; NP is the MODE (2 / 3 bf16 planes, 16 = two fp16 planes)
template <int NP>
__device__ __forceinline__ void plane_cut4(const f32x4& x, u32x2 (&o)[mode_planes(NP)]) {
    if constexpr (NP == kModeF16) {
        typedef _Float16 h2 __attribute__((ext_vector_type(2)));
        _Float16 h[4], l[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            h[i] = (_Float16)x[i];                                   // round to nearest even (v_cvt_f16_f32)
            l[i] = (_Float16)((x[i] - (float)h[i]) * kF16LowScale);
        }
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            o[0][j] = __builtin_bit_cast(unsigned, h2{h[2 * j], h[2 * j + 1]});
            o[1][j] = __builtin_bit_cast(unsigned, h2{l[2 * j], l[2 * j + 1]});
        }
    } else {
        unsigned hb[4], mb[4], lb[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            hb[i] = __float_as_uint(x[i]) & 0xffff0000u;
            const float r1 = x[i] - __uint_as_float(hb[i]);
            if constexpr (NP == 3) {
                mb[i] = __float_as_uint(r1) & 0xffff0000u;
                lb[i] = __float_as_uint(r1 - __uint_as_float(mb[i]));
            } else {
                mb[i] = __float_as_uint(r1);
                lb[i] = 0u;
            }
        }
#pragma unroll
        for (int j = 0; j < 2; ++j) {        // (hi & 0xffff0000) | (lo >> 16): one v_perm_b32
            o[0][j] = __builtin_amdgcn_perm(hb[2 * j + 1], hb[2 * j], 0x07060302u);
            o[1][j] = __builtin_amdgcn_perm(mb[2 * j + 1], mb[2 * j], 0x07060302u);
            if constexpr (NP == 3) o[2][j] = __builtin_amdgcn_perm(lb[2 * j + 1], lb[2 * j], 0x07060302u);
        }
    }
}

// eight consecutive fp32 of one row (k = 8 kg .. 8 kg + 7) -> the row's 16-byte unit of every plane
template <int NP>
__device__ __forceinline__ void plane_store8(char* planes, int row, int kg, int nkc, const f32x4& lo, const f32x4& hi) {
    constexpr int NPL = mode_planes(NP);
    u32x2 a[NPL], b[NPL];
    plane_cut4<NP>(lo, a);
    plane_cut4<NP>(hi, b);
#pragma unroll
    for (int pl = 0; pl < NPL; ++pl)
        *reinterpret_cast<u32x4*>(planes + plane_unit(row, kg, nkc, NPL, pl) * 16) = u32x4{a[pl][0], a[pl][1], b[pl][0], b[pl][1]};
}

// four consecutive fp32 (k = k0 .. k0 + 3, k0 % 4 == 0) -> half a unit of every plane
template <int NP>
__device__ __forceinline__ void plane_store4(char* planes, int row, int k0, int nkc, const f32x4& v) {
    constexpr int NPL = mode_planes(NP);
    u32x2 a[NPL];
    plane_cut4<NP>(v, a);
#pragma unroll
    for (int pl = 0; pl < NPL; ++pl)
        *reinterpret_cast<u32x2*>(planes + plane_unit(row, k0 >> 3, nkc, NPL, pl) * 16 + (k0 & 4) * 2) = a[pl];
}
#endif

// ---- pre-planed split-bf16 GEMM (gemm_pp.hip):  C[M,N] = epi(A[M,K] . W[N,K]^T), both operands given as planes -------------
struct PlaneGemmParams {
    const void* Ap;          // planes of A over [M][K]
    const void* Wp;          // planes of W over [N][K]
    float* C; int ldc;       // fp32 result (nullptr: not stored)
    void* Cp;                // planes of the result over [M][N] (nullptr: not stored)
    int M, N, K;
    const float* bias;
    const float* R; int ldr; // EPI_BIAS_RES
    int qcols; float qscale; // EPI_QKV
    float acc_scale;         // the accumulator is multiplied by this before the epilogue (0 = 1: undoes a power-of-two scale the
                             // weight planes were cut with, fp16 mode)
    int no_swap;             // diagnostic: plane output through 8-byte stores instead of the lane-swapped 16-byte form
    // ---- LayerNorm folded into the GEMMs around it (stream kernel, two-plane modes; posenet.hip).  Row statistics travel as
    // partial (sum, sum of squares) pairs per 16 columns, row-block major:  stats[row / 16][ln_dim / 16][row % 16][2].
    //   consumer (EPI_QKV / EPI_BIAS_GELU, ln_stats != null): the A planes are those of the RAW (un-normalised) tensor x, the
    //     weight planes carry gamma (W_nk gamma_k), `bias` carries d_n = b_n + sum_k beta_k W_nk, ln_c carries c_n = sum_k gamma_k
    //     W_nk, and the epilogue finishes LN(x) W^T + b = (acc - mu_m c_n) rstd_m + d_n;  ln_dim = K.
    //   EPI_BIAS_RES, r_stats != null: the residual R is raw too; LN(R) = (R - mu) rstd r_gamma + r_beta is what is added.
    //   EPI_BIAS_RES, out_stats != null: the partial statistics of the RESULT rows are written (the next consumer's input).
    //     For both, ln_dim = N.
    const float* ln_stats; const float* ln_c;
    const float* r_stats; const float* r_gamma; const float* r_beta;
    float* out_stats;
    int ln_dim; float ln_eps;
};
// epi: EPI_BIAS / EPI_BIAS_GELU / EPI_BIAS_RES / EPI_QKV.  M % 144 == 0, N % 64 == 0, K % 32 == 0.  nplane = the mode (2, 3, 16).
int launch_gemm_pp(const PlaneGemmParams& p, int epi, int nplane, hipStream_t s);
// scale * X[rows][K] fp32 (row stride ld floats) -> planes; rows % 16 == 0, K % 32 == 0; scale = 1 for activations
int launch_plane_split(const float* X, int ld, int rows, int K, int nplane, float scale, void* out, hipStream_t s);
// LayerNorm in place on x [M][D] (M % 16 == 0) that also writes the planes of its result
int launch_layernorm_planes(float* x, const float* g, const float* b, int M, int D, int nplane, void* planes, hipStream_t s);
// attention (S = 144, d_h = 128 only) writing the planes of ctx [n_seq * 144][n_head * 128] instead of fp32
int launch_attention_planes(const float* qkv, void* ctx_planes, int n_seq, int n_head, int nplane, hipStream_t s);

}  // namespace rohm
