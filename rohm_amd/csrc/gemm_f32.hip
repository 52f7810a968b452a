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
#include <stdlib.h>
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

template <int BN, int EPI, int VAR = 0, bool FULL = false, bool CONV = false>
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

    // ---- global -> LDS staging by LDS-DMA (global_load_lds_dwordx4) ---------------------------
    // The DMA writes LDS linearly (wave-uniform base + lane*16 B), so the XOR swizzle is applied to
    // the per-lane SOURCE address instead (guide rule 21): the lane whose LDS unit is (row, pslot)
    // fetches logical slot pslot ^ ((row>>1)&7) of that row -- still inside the row's one 128-B line.
    // Rows past M / N are clamped to a valid row: they only feed accumulator rows / columns that the
    // epilogue never stores, so no zero fill is needed and there is no predicated load.
    const float* a_src[A_ITERS];
    int a_b[A_ITERS], a_tq[A_ITERS], a_slot[A_ITERS];      // conv gather: clip, output frame, 16-B slot
#pragma unroll
    for (int i = 0; i < A_ITERS; ++i) {
        const int u = tid + i * 256;
        const int row = (u < A_UNITS) ? (u >> 3) : 0;
        const int slot = (u & 7) ^ ((row >> 1) & 7);
        const int grow = (m0 + row < p.M) ? m0 + row : p.M - 1;
        a_src[i] = p.A + (size_t)grow * p.lda + slot * 4;
        if constexpr (CONV) {
            a_b[i] = grow / p.conv_tq;
            a_tq[i] = grow % p.conv_tq;
            a_slot[i] = slot;
        }
    }
    // conv gather: source pointer of unit i for the K chunk starting at k0 (a chunk never straddles taps
    // because cin_pad is a multiple of 32); taps that fall outside the clip read a page of zeros.
    auto conv_src = [&](int i, int k0) -> const float* {
        const int j = k0 / p.conv_cin_pad, ci0 = k0 - j * p.conv_cin_pad;
        const int tin = a_tq[i] * p.conv_stride + p.conv_off[j];
        const bool ok = (tin >= 0) && (tin < p.conv_tin);
        return ok ? p.A + ((size_t)a_b[i] * p.conv_tin + tin) * p.lda + ci0 + a_slot[i] * 4
                  : p.zero_page + a_slot[i] * 4;
    };
    const float* b_src[B_ITERS];
#pragma unroll
    for (int i = 0; i < B_ITERS; ++i) {
        const int u = tid + i * 256;
        const int row = u >> 3;
        const int slot = (u & 7) ^ ((row >> 1) & 7);
        const int grow = (n0 + row < p.N) ? n0 + row : p.N - 1;
        b_src[i] = p.W + (size_t)grow * p.ldw + slot * 4;
    }
    const int wave_u = __builtin_amdgcn_readfirstlane(wave) * 64;   // provably wave-uniform LDS base
    auto dma = [&](int buf, int k0) {
        float* as = As + buf * (BM * BK);
        float* bs = Bs + buf * (BN * BK);
#pragma unroll
        for (int i = 0; i < A_ITERS; ++i) {
            if (i < A_ITERS - 1 || wave_u < A_UNITS - (A_ITERS - 1) * 256)   // last pass: waves 0-1 only
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(CONV ? conv_src(i, k0) : a_src[i] + k0),
                                                 (__attribute__((address_space(3))) void*)(as + (i * 256 + wave_u) * 4),
                                                 16, 0, 0);
        }
#pragma unroll
        for (int i = 0; i < B_ITERS; ++i)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(b_src[i] + k0),
                                             (__attribute__((address_space(3))) void*)(bs + (i * 256 + wave_u) * 4),
                                             16, 0, 0);
    };

    f32x4 acc[NRB][NCB];
#pragma unroll
    for (int r = 0; r < NRB; ++r)
#pragma unroll
        for (int c = 0; c < NCB; ++c) acc[r][c] = f32x4{0.f, 0.f, 0.f, 0.f};

    // Operand order: SWAP puts the weight fragment on the MFMA "A" side, so a lane ends up with four
    // CONSECUTIVE output columns of one row (one 16-byte store); the transposed output head keeps the
    // natural order because its stores are contiguous along the token axis instead.
    constexpr bool SWAP = (EPI != EPI_OUT_T);

    const int nk = p.K / BK;

    // Fragment registers for the two k16 halves of a chunk.  fa0/fb0 feed half 0, fa1/fb1 half 1.
    f32x4 fa0[NRB], fb0[NCB], fa1[NRB], fb1[NCB];
    auto read_frags = [&](f32x4 (&fa)[NRB], f32x4 (&fb)[NCB], int buf, int ks) {
        const float* as = As + buf * (BM * BK);
        const float* bs = Bs + buf * (BN * BK) + wave * WN * BK;
        const int slot = ks * 4 + lg;
#pragma unroll
        for (int c = 0; c < NCB; ++c) fb[c] = *reinterpret_cast<const f32x4*>(bs + lds_off(c * 16 + li, slot));
#pragma unroll
        for (int r = 0; r < NRB; ++r) fa[r] = *reinterpret_cast<const f32x4*>(as + lds_off(r * 16 + li, slot));
    };
    auto mma_half = [&](const f32x4 (&fa)[NRB], const f32x4 (&fb)[NCB]) {
#pragma unroll
        for (int r = 0; r < NRB; ++r)
#pragma unroll
            for (int c = 0; c < NCB; ++c)
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    if constexpr (SWAP)
                        acc[r][c] = __builtin_amdgcn_mfma_f32_16x16x4f32(fb[c][j], fa[r][j], acc[r][c], 0, 0, 0);
                    else
                        acc[r][c] = __builtin_amdgcn_mfma_f32_16x16x4f32(fa[r][j], fb[c][j], acc[r][c], 0, 0, 0);
                }
    };

    // Schedule (one barrier per 32-wide K chunk, placed BETWEEN the chunk's two MFMA halves):
    //   top of chunk k : issue LDS reads of half 1 (chunk k is already visible)      } latency hidden
    //   MFMA half 0    : 72 MFMAs on registers read during the previous chunk         } under MFMAs
    //   vmcnt(0) + barrier : chunk k+1 (DMA issued one chunk ago) is now visible to everyone, and every
    //                    wave has finished READING chunk k into registers
    //   issue DMA of chunk k+2 into chunk k's buffer; issue LDS reads of chunk k+1's half 0
    //   MFMA half 1    : 72 MFMAs
    // so neither the HBM/L2 latency nor the LDS latency is ever exposed in steady state.
    // VAR (diagnostic builds, ROHM_GEMM_VARIANT): 0 = shipped; 5 = MFMA only + no epilogue; 6 = no epilogue.
    dma(0, 0);
    if (nk > 1) dma(1, BK);
    if (nk > 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(A_ITERS + B_ITERS) : "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    read_frags(fa0, fb0, 0, 0);
    for (int kc = 0; kc < nk; ++kc) {
        const int buf = kc & 1;
        if constexpr (VAR != 5) read_frags(fa1, fb1, buf, 1);
        mma_half(fa0, fb0);
        if constexpr (VAR != 5) {
            // MFMAs are register-only, so hipcc would otherwise sink them below the barrier (guide rule 18)
            __builtin_amdgcn_sched_barrier(0);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            __builtin_amdgcn_sched_barrier(0);
            if (kc + 2 < nk) dma(buf, (kc + 2) * BK);
            if (kc + 1 < nk) read_frags(fa0, fb0, buf ^ 1, 0);
            mma_half(fa1, fb1);
            __builtin_amdgcn_sched_barrier(0);
        } else {
            mma_half(fa0, fb0);
        }
    }
    if constexpr (VAR == 5 || VAR == 6) {
        // diagnostics: skip the epilogue unless an impossible value appears (keeps the MFMAs live)
        if (acc[0][0][0] != 123456.789f) return;
    }

    // ---- epilogue ------------------------------------------------------------------------------
    if constexpr (EPI == EPI_OUT_T) {
        // natural operand order: acc[r][c][q] = C[m = m0 + r*16 + lg*4 + q][n = n0 + wave*WN + c*16 + li];
        // rows = output channels, cols = tokens; stored transposed into [B, C_total, 1, T]
#pragma unroll
        for (int c = 0; c < NCB; ++c) {
            const int n = n0 + wave * WN + c * 16 + li;
            if (n >= p.N) continue;
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
        }
    } else {
        // swapped operand order: acc[r][c][q] = C[m = m0 + r*16 + li][n = nb + q], nb = n0 + wave*WN + c*16 + lg*4
        // FULL: the launcher proved M % 144 == 0, N % BN == 0 and 16-byte alignment of C / R / bias / tables,
        // so the hot instantiation carries no edge masks and only 16-byte accesses.
        const bool vec_ok = FULL || ((p.N % 4 == 0) && (p.ldc % 4 == 0) && (((uintptr_t)p.C & 15) == 0));
#pragma unroll
        for (int c = 0; c < NCB; ++c) {
            const int nb = n0 + wave * WN + c * 16 + lg * 4;
            if (!FULL && nb >= p.N) continue;
            f32x4 bias4 = f32x4{0.f, 0.f, 0.f, 0.f};
            if constexpr (FULL) {
                if (p.bias) bias4 = *reinterpret_cast<const f32x4*>(p.bias + nb);
            } else if (p.bias) {
#pragma unroll
                for (int q = 0; q < 4; ++q) bias4[q] = (nb + q < p.N) ? p.bias[nb + q] : 0.f;
            }
#pragma unroll
            for (int r = 0; r < NRB; ++r) {
                const int m = m0 + r * 16 + li;
                if (!FULL && m >= p.M) continue;
                f32x4 v;
#pragma unroll
                for (int q = 0; q < 4; ++q) v[q] = acc[r][c][q] + bias4[q];
                if constexpr (EPI == EPI_BIAS_GELU) {
#pragma unroll
                    for (int q = 0; q < 4; ++q) v[q] = gelu_erf(v[q]);
                }
                if constexpr (EPI == EPI_BIAS_RES) {
                    const float* rp = p.R + (size_t)m * p.ldr + nb;
                    if (FULL || (vec_ok && (p.ldr % 4 == 0) && (((uintptr_t)p.R & 15) == 0))) {
                        const f32x4 rr = *reinterpret_cast<const f32x4*>(rp);
#pragma unroll
                        for (int q = 0; q < 4; ++q) v[q] += rr[q];
                    } else {
#pragma unroll
                        for (int q = 0; q < 4; ++q)
                            if (nb + q < p.N) v[q] += rp[q];
                    }
                }
                if constexpr (EPI == EPI_QKV) {
                    if (nb < p.qcols) {      // qcols is a multiple of 4
#pragma unroll
                        for (int q = 0; q < 4; ++q) v[q] *= p.qscale;
                    }
                }
                if constexpr (EPI == EPI_EMBED) {
                    const int bidx = m / p.S, tok = m % p.S;
                    const float* tp = (tok == 0) ? p.tab0 + (size_t)bidx * p.ldtab0 + nb
                                                 : p.tab + (size_t)tok * p.ldtab + nb;
                    if constexpr (FULL) {
                        const f32x4 tt = *reinterpret_cast<const f32x4*>(tp);
#pragma unroll
                        for (int q = 0; q < 4; ++q) v[q] = (tok == 0 ? 0.f : acc[r][c][q]) + tt[q];
                    } else {
#pragma unroll
                        for (int q = 0; q < 4; ++q) v[q] = (tok == 0 ? 0.f : acc[r][c][q]) + tp[q];
                    }
                }
                const size_t crow = CONV ? (size_t)m * (p.orow_mul_m1 + 1) + p.orow_add : (size_t)m;
                float* cp = p.C + crow * p.ldc + nb;
                if (vec_ok) {
                    *reinterpret_cast<f32x4*>(cp) = v;
                } else {
#pragma unroll
                    for (int q = 0; q < 4; ++q)
                        if (nb + q < p.N) cp[q] = v[q];
                }
            }
        }
    }
}

static int gemm_variant() {
    static int v = -1;
    if (v < 0) {
        const char* e = getenv("ROHM_GEMM_VARIANT");
        v = e ? atoi(e) : 0;
    }
    return v;
}

template <int BN, int EPI, int VAR = 0, bool FULL = false, bool CONV = false>
static int launch_one(const GemmParams& p, hipStream_t s) {
    const int tiles = ((p.M + BM - 1) / BM) * ((p.N + BN - 1) / BN);
    static int lds_pad = -1;
    if (lds_pad < 0) { const char* e = getenv("ROHM_GEMM_LDS_PAD"); lds_pad = e ? atoi(e) : 0; }
    // One workgroup per CU on purpose: with two co-resident workgroups the hardware hands BOTH freed slots of
    // a CU to the next tiles, so a 3-tiles-per-CU GEMM (QKV at B = 64) degenerates to 4 + 2 (measured 158 us vs
    // 128 us); the schedule above already hides LDS / L2 latency without a partner wave.  Requesting more than
    // half of the 160 KiB LDS pins the residency to one.
    size_t lds = (size_t)2 * (BM + BN) * BK * sizeof(float) + lds_pad;
    if (lds < 84 * 1024) lds = 84 * 1024;
    static bool attr_set[64] = {};
    int dev = 0;
    ROHM_HIP_CHECK(hipGetDevice(&dev));
    if (dev < 64 && !attr_set[dev]) {
        ROHM_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_f32_kernel<BN, EPI, VAR, FULL, CONV>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        attr_set[dev] = true;
    }
    static const char* const kNames[] = {"gemm_bias", "gemm_bias_gelu", "gemm_bias_res", "gemm_qkv", "gemm_embed",
                                         "gemm_out_t"};
    if (CONV) {
        prof::Scope ps(BN == 128 ? "conv_gemm" : "conv_gemm/64", 2.0 * p.M * p.N * p.K,
                       4.0 * ((double)p.M * p.K + (double)p.N * p.K + (double)p.M * p.N), s);
        hipLaunchKernelGGL((gemm_f32_kernel<BN, EPI, VAR, FULL, CONV>), dim3(tiles), dim3(256), lds, s, p);
        ROHM_LAUNCH_CHECK();
        return ROHM_OK;
    }
    static const char* const kNames64[] = {"gemm_bias/64", "gemm_bias_gelu/64", "gemm_bias_res/64", "gemm_qkv/64",
                                           "gemm_embed/64", "gemm_out_t/64"};
    prof::Scope ps(BN == 128 ? kNames[EPI] : kNames64[EPI], 2.0 * p.M * p.N * p.K,
                   4.0 * ((double)p.M * p.K + (double)p.N * p.K + (double)p.M * p.N), s);
    hipLaunchKernelGGL((gemm_f32_kernel<BN, EPI, VAR, FULL, CONV>), dim3(tiles), dim3(256), lds, s, p);
    ROHM_LAUNCH_CHECK();
    return ROHM_OK;
}

static inline bool al16(const void* q) { return (((uintptr_t)q) & 15) == 0; }

template <int BN, int EPI, int VAR = 0>
static int launch_t(const GemmParams& p, hipStream_t s) {
    bool full = (p.M % BM == 0) && (p.N % BN == 0) && (p.ldc % 4 == 0) && al16(p.C) && al16(p.bias);
    if (EPI == EPI_BIAS_RES) full = full && (p.ldr % 4 == 0) && al16(p.R);
    if (EPI == EPI_EMBED) full = full && (p.ldtab % 4 == 0) && (p.ldtab0 % 4 == 0) && al16(p.tab) && al16(p.tab0);
    if (EPI == EPI_QKV) full = full && (p.qcols % 4 == 0);
    if (EPI == EPI_OUT_T || VAR != 0) full = false;
    if (p.conv_taps > 0) {
        if constexpr (EPI == EPI_BIAS && VAR == 0) {
            return full ? launch_one<BN, EPI, 0, true, true>(p, s) : launch_one<BN, EPI, 0, false, true>(p, s);
        } else {
            set_error("gemm: conv gather supports the bias epilogue only");
            return ROHM_ERR_UNSUPPORTED;
        }
    }
    if (full) {
        if constexpr (EPI != EPI_OUT_T && VAR == 0) return launch_one<BN, EPI, 0, true>(p, s);
    }
    return launch_one<BN, EPI, VAR, false>(p, s);
}

template <int EPI>
static int launch_bn(const GemmParams& p, hipStream_t s) {
    // pick the wider tile only when it still yields at least one tile per CU
    const int tiles128 = ((p.M + BM - 1) / BM) * ((p.N + 127) / 128);
    const int var = gemm_variant();
    const int force_bn = var / 10;            // diagnostics: 6x -> BN 64, 12x -> BN 128
    if constexpr (EPI == EPI_BIAS) {
        switch (var % 10) {                   // schedule variants exist for the plain epilogue only
            case 5: return force_bn == 6 ? launch_t<64, EPI, 5>(p, s) : launch_t<128, EPI, 5>(p, s);
            case 6: return force_bn == 6 ? launch_t<64, EPI, 6>(p, s) : launch_t<128, EPI, 6>(p, s);
            default: break;
        }
    }
    if (force_bn == 6) return launch_t<64, EPI>(p, s);
    if (force_bn == 12) return launch_t<128, EPI>(p, s);
    if (tiles128 >= 256 && p.N % 128 == 0) return launch_t<128, EPI>(p, s);
    return launch_t<64, EPI>(p, s);
}

int launch_gemm(const GemmParams& p, int epi, hipStream_t s) {
    ROHM_ARG_CHECK(p.K > 0 && p.K % BK == 0, "gemm: K=%d must be a positive multiple of %d", p.K, BK);
    ROHM_ARG_CHECK(p.lda % 4 == 0 && p.ldw % 4 == 0, "gemm: lda/ldw must be multiples of 4 floats");
    ROHM_ARG_CHECK(((uintptr_t)p.A % 16) == 0 && ((uintptr_t)p.W % 16) == 0, "gemm: A/W must be 16-byte aligned");
    ROHM_ARG_CHECK(p.M > 0 && p.N > 0, "gemm: empty problem");
    if (p.conv_taps > 0) {
        ROHM_ARG_CHECK(p.conv_taps <= 5 && p.conv_cin_pad % BK == 0 && p.K == p.conv_taps * p.conv_cin_pad,
                       "gemm: conv gather needs cin_pad %% 32 == 0 and K == taps * cin_pad");
        ROHM_ARG_CHECK(p.zero_page && p.conv_tq > 0 && p.conv_tin > 0 && p.M % p.conv_tq == 0,
                       "gemm: bad conv gather geometry");
    }
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
