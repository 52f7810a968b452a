// fp32 MFMA GEMM for gfx950:  C[M,N] = epi(A[M,K] . W[N,K]^T)
//
// Every Linear of the PoseNet path (model/posenet.py:63-69, model/heads.py:154,169) runs on this
// kernel.  Numerics: v_mfma_f32_16x16x4_f32 is an exact-fp32 fma chain (guide §3), i.e. the same
// rounding class as the CPU reference's fp32 GEMM; only the summation order differs.
//
// Tiling (MI355X-first, not a warp-shaped port):
//   * workgroup tile 144 x BN (BN = 128 or 64), 4 waves (one per SIMD).  144 = 9 x 16 is exactly one
//     PoseNet clip (143 frames + timestep token), so M = B*144 tiles with no remainder and, at the
//     headline batch B = 64, N = 512 gives 64 x 4 = 256 tiles = one per CU.
//   * wave w owns columns [w*BN/4, (w+1)*BN/4): 9 x (BN/64) accumulator blocks of 16x16.
//   * K is walked in 32-wide chunks, register-staged global->LDS with two LDS buffers: the loads
//     for chunk k+1 are in flight while chunk k is multiplied, one barrier per chunk.
//   * both operands are K-contiguous, so a lane's MFMA fragments for four consecutive k-steps are
//     one ds_read_b128: lane (i = l&15, g = l>>4) holds X[i][16*ks + 4*g + j], j = 0..3, and MFMA j
//     contracts k = {4g + j}: a fixed permutation of k inside each 16-chunk, identical for A and W.
//   * LDS rows are 128 B (32 floats); the 16-byte slot index is XOR-swizzled with (row>>1)&7, which
//     makes every ds_read_b128 lane group hit 16 distinct 16-byte slots (conflict-free, guide §2).
//   * blockIdx -> tile mapping is XCD-aware: each XCD gets a contiguous run of tiles that share A
//     panels, so an A panel is fetched into one L2 instead of eight.
#include "common.h"

namespace rohm {

constexpr int BM = 144;
constexpr int BK = 32;
constexpr int NRB = BM / 16;        // 9 row blocks
constexpr int A_UNITS = BM * 8;     // 16-byte units per A chunk (1152)
constexpr int A_ITERS = (A_UNITS + 255) / 256;  // 5 (last one half full)

__device__ __forceinline__ int lds_off(int row, int slot) {   // float index inside a [rows][32] tile
    return row * BK + ((slot ^ ((row >> 1) & 7)) << 2);
}

__device__ __forceinline__ float gelu_erf(float x) {
    return 0.5f * x * (1.0f + erff(x * 0.70710678118654752440f));
}

template <int BN, int EPI>
__global__ __launch_bounds__(256) void gemm_f32_kernel(GemmParams p) {
    constexpr int WN = BN / 4;        // columns per wave
    constexpr int NCB = WN / 16;      // 16-wide column blocks per wave (2 or 1)
    constexpr int B_UNITS = BN * 8;
    constexpr int B_ITERS = B_UNITS / 256;   // 4 or 2

    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* As = smem;                       // [2][BM*BK]
    float* Bs = smem + 2 * BM * BK;         // [2][BN*BK]

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int li = lane & 15, lg = lane >> 4;

    const int tiles_n = (p.N + BN - 1) / BN;
    const int tiles_m = (p.M + BM - 1) / BM;
    const int tile = xcd_remap(blockIdx.x, tiles_m * tiles_n);
    const int m0 = (tile / tiles_n) * BM;
    const int n0 = (tile % tiles_n) * BN;

    // ---- global -> register staging ---------------------------------------------------------
    // Loads are unconditional (out-of-range rows are clamped to a valid row and zeroed at the LDS
    // store): a predicated load makes hipcc branch and drain vmcnt per element (guide §5 trap (c)).
    f32x4 ra[A_ITERS], rb[B_ITERS];
    const float* a_ptr[A_ITERS];
    bool a_ok[A_ITERS];
#pragma unroll
    for (int i = 0; i < A_ITERS; ++i) {
        const int u = tid + i * 256;
        const int row = (u < A_UNITS) ? (u >> 3) : 0, slot = u & 7;
        a_ok[i] = (m0 + row < p.M);
        const int grow = a_ok[i] ? m0 + row : p.M - 1;
        a_ptr[i] = p.A + (size_t)grow * p.lda + slot * 4;
    }
    const float* b_ptr[B_ITERS];
    bool b_ok[B_ITERS];
#pragma unroll
    for (int i = 0; i < B_ITERS; ++i) {
        const int u = tid + i * 256;
        const int row = u >> 3, slot = u & 7;
        b_ok[i] = (n0 + row < p.N);
        const int grow = b_ok[i] ? n0 + row : p.N - 1;
        b_ptr[i] = p.W + (size_t)grow * p.ldw + slot * 4;
    }
    auto g_load = [&](int k0) {
#pragma unroll
        for (int i = 0; i < A_ITERS; ++i) ra[i] = *reinterpret_cast<const f32x4*>(a_ptr[i] + k0);
#pragma unroll
        for (int i = 0; i < B_ITERS; ++i) rb[i] = *reinterpret_cast<const f32x4*>(b_ptr[i] + k0);
    };
    auto s_store = [&](int buf) {
        float* as = As + buf * (BM * BK);
        float* bs = Bs + buf * (BN * BK);
#pragma unroll
        for (int i = 0; i < A_ITERS; ++i) {
            const int u = tid + i * 256;
            const f32x4 v = a_ok[i] ? ra[i] : f32x4{0.f, 0.f, 0.f, 0.f};
            if (u < A_UNITS) *reinterpret_cast<f32x4*>(as + lds_off(u >> 3, u & 7)) = v;
        }
#pragma unroll
        for (int i = 0; i < B_ITERS; ++i) {
            const int u = tid + i * 256;
            *reinterpret_cast<f32x4*>(bs + lds_off(u >> 3, u & 7)) = b_ok[i] ? rb[i] : f32x4{0.f, 0.f, 0.f, 0.f};
        }
    };

    f32x4 acc[NRB][NCB];
#pragma unroll
    for (int r = 0; r < NRB; ++r)
#pragma unroll
        for (int c = 0; c < NCB; ++c) acc[r][c] = f32x4{0.f, 0.f, 0.f, 0.f};

    const int nk = p.K / BK;
    g_load(0);
    s_store(0);
    __syncthreads();

    for (int kc = 0; kc < nk; ++kc) {
        const int buf = kc & 1;
        if (kc + 1 < nk) g_load((kc + 1) * BK);
        const float* as = As + buf * (BM * BK);
        const float* bs = Bs + buf * (BN * BK) + wave * WN * BK;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            const int slot = ks * 4 + lg;
            f32x4 bf[NCB];
#pragma unroll
            for (int c = 0; c < NCB; ++c)
                bf[c] = *reinterpret_cast<const f32x4*>(bs + lds_off(c * 16 + li, slot));
#pragma unroll
            for (int r = 0; r < NRB; ++r) {
                const f32x4 af = *reinterpret_cast<const f32x4*>(as + lds_off(r * 16 + li, slot));
#pragma unroll
                for (int c = 0; c < NCB; ++c) {
#pragma unroll
                    for (int j = 0; j < 4; ++j)
                        acc[r][c] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[j], bf[c][j], acc[r][c], 0, 0, 0);
                }
            }
        }
        if (kc + 1 < nk) s_store(buf ^ 1);
        __syncthreads();
    }

    // ---- epilogue: acc[r][c][q] = C[m0 + r*16 + lg*4 + q][n0 + wave*WN + c*16 + li] ----------
#pragma unroll
    for (int c = 0; c < NCB; ++c) {
        const int n = n0 + wave * WN + c * 16 + li;
        if (n >= p.N) continue;
        if constexpr (EPI == EPI_OUT_T) {
            // rows = output channels, cols = tokens; store transposed into [B, C_total, 1, T]
            const int b = n / p.S, tok = n % p.S;
            if (tok == 0) continue;
            float* dst = p.C + ((size_t)b * p.C_total + p.ch_off) * p.T + (tok - 1);
#pragma unroll
            for (int r = 0; r < NRB; ++r)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int m = m0 + r * 16 + lg * 4 + q;
                    if (m < p.M) dst[(size_t)m * p.T] = acc[r][c][q] + p.bias[m];
                }
        } else {
            const float bias = p.bias ? p.bias[n] : 0.f;
#pragma unroll
            for (int r = 0; r < NRB; ++r)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int m = m0 + r * 16 + lg * 4 + q;
                    if (m >= p.M) continue;
                    float v = acc[r][c][q] + bias;
                    if constexpr (EPI == EPI_BIAS_GELU) v = gelu_erf(v);
                    if constexpr (EPI == EPI_BIAS_RES) v += p.R[(size_t)m * p.ldr + n];
                    if constexpr (EPI == EPI_QKV) v = (n < p.qcols) ? v * p.qscale : v;
                    if constexpr (EPI == EPI_EMBED) {
                        const int bidx = m / p.S, tok = m % p.S;
                        v = (tok == 0) ? p.tab0[(size_t)bidx * p.ldtab0 + n]
                                       : acc[r][c][q] + p.tab[(size_t)tok * p.ldtab + n];
                    }
                    p.C[(size_t)m * p.ldc + n] = v;
                }
        }
    }
}

template <int BN, int EPI>
static int launch_t(const GemmParams& p, hipStream_t s) {
    const int tiles = ((p.M + BM - 1) / BM) * ((p.N + BN - 1) / BN);
    const size_t lds = (size_t)2 * (BM + BN) * BK * sizeof(float);
    static bool attr_set[64] = {};
    int dev = 0;
    ROHM_HIP_CHECK(hipGetDevice(&dev));
    if (dev < 64 && !attr_set[dev]) {
        ROHM_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_f32_kernel<BN, EPI>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        attr_set[dev] = true;
    }
    static const char* const kNames[] = {"gemm_bias", "gemm_bias_gelu", "gemm_bias_res", "gemm_qkv", "gemm_embed",
                                         "gemm_out_t"};
    static const char* const kNames64[] = {"gemm_bias/64", "gemm_bias_gelu/64", "gemm_bias_res/64", "gemm_qkv/64",
                                           "gemm_embed/64", "gemm_out_t/64"};
    prof::Scope ps(BN == 128 ? kNames[EPI] : kNames64[EPI], 2.0 * p.M * p.N * p.K,
                   4.0 * ((double)p.M * p.K + (double)p.N * p.K + (double)p.M * p.N), s);
    hipLaunchKernelGGL((gemm_f32_kernel<BN, EPI>), dim3(tiles), dim3(256), lds, s, p);
    ROHM_LAUNCH_CHECK();
    return ROHM_OK;
}

template <int EPI>
static int launch_bn(const GemmParams& p, hipStream_t s) {
    // pick the wider tile only when it still yields at least one tile per CU
    const int tiles128 = ((p.M + BM - 1) / BM) * ((p.N + 127) / 128);
    if (tiles128 >= 256 && p.N % 128 == 0) return launch_t<128, EPI>(p, s);
    return launch_t<64, EPI>(p, s);
}

int launch_gemm(const GemmParams& p, int epi, hipStream_t s) {
    ROHM_ARG_CHECK(p.K > 0 && p.K % BK == 0, "gemm: K=%d must be a positive multiple of %d", p.K, BK);
    ROHM_ARG_CHECK(p.lda % 4 == 0 && p.ldw % 4 == 0, "gemm: lda/ldw must be multiples of 4 floats");
    ROHM_ARG_CHECK(((uintptr_t)p.A % 16) == 0 && ((uintptr_t)p.W % 16) == 0, "gemm: A/W must be 16-byte aligned");
    ROHM_ARG_CHECK(p.M > 0 && p.N > 0, "gemm: empty problem");
    switch (epi) {
        case EPI_BIAS: return launch_bn<EPI_BIAS>(p, s);
        case EPI_BIAS_GELU: return launch_bn<EPI_BIAS_GELU>(p, s);
        case EPI_BIAS_RES: return launch_bn<EPI_BIAS_RES>(p, s);
        case EPI_QKV: return launch_bn<EPI_QKV>(p, s);
        case EPI_EMBED: return launch_bn<EPI_EMBED>(p, s);
        case EPI_OUT_T: return launch_bn<EPI_OUT_T>(p, s);
    }
    set_error("gemm: unknown epilogue %d", epi);
    return ROHM_ERR_ARG;
}

}  // namespace rohm
