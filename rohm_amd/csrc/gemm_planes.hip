// Split-bf16 GEMM for gfx950 (opt-in precision ladder):  C[M,N] = epi(A[M,K] . W[N,K]^T) with every fp32
// product a.w emulated by bf16 MFMA products of bf16 PLANES of the operands, fp32 accumulation.
//
//   ROHM_GEMM_PRECISION=bf16x6: three planes per operand, the six products of weight >= 2^-16 (fp32-class accuracy: every
//                               GEMM / PoseNet parity test passes at the exact-fp32 kernel's bars)
//   ROHM_GEMM_PRECISION=bf16x3: two planes, three products (~2^-16 per product)
//
// The default and every headline number stay on the exact-fp32 MFMA kernel (gemm_f32.hip); this one is the labelled
// second line.  Same Linears as there (model/posenet.py:63-69, model/heads.py:154,169).
//
// Planes by TRUNCATION: h = the upper 16 bits of x, m = the upper 16 bits of x - h, l = x - h - m.  Every remainder is
// exact in fp32, h and m carry 8 significant bits each and the last remainder has at most 8 left, so x = h + m + l holds
// EXACTLY for every |x| >= 2^-100 (below that the remainders are denormal; tests/test_precision_ladder_arith.py) (two
// planes: x = h + m up to 2^-15 |x|).  Per value: and, sub, and, sub on the VALU plus half a v_perm_b32 per
// plane to pack two bf16 into a dword.
//
// Design.  The operands arrive as fp32 (4 B per element for an MFMA that takes a quarter of the fp32 MFMA's time), so per
// MFMA cycle this kernel issues 2.7x the LDS-DMA pieces, waits and fragment reads of the fp32 kernel, plus the plane
// arithmetic; with ONE wave per SIMD none of that hides behind 17-cycle MFMAs (measured: a 4-wave version, per-wave or
// shared splitting alike, ran at 1.3x the fp32 kernel).  Hence:
//   * 8 waves = TWO per SIMD on a 144 x 128 tile (2 x 4 wave grid: wave (wm, wn) owns row blocks 0-4 / 5-8 and 32
//     columns; the two row halves of a column group share a SIMD, so the 5 : 4 split does not unbalance the SIMDs): while
//     one wave issues DMA pieces, waits on LDS or cuts planes, the other feeds the matrix core;
//   * every element of the A tile is cut ONCE per workgroup: each thread splits 2.25 sixteen-byte units of the raw tile and
//     writes the planes to a shared LDS image laid out as ready-made bf16x8 MFMA fragments (row, k-group), which all waves
//     read back with one ds_read_b128 per (row block, plane); the B planes of a wave's own 32 columns stay in registers;
//   * chunk k+1 is cut while chunk k is multiplied (its raw tile landed during chunk k-1); with three raw stages the DMA of
//     chunk k+2 targets a stage that was consumed two barriers ago, so its pieces are issued behind the MFMAs of the first
//     two steps instead of in front of them: ONE barrier per chunk, nothing but the first fragment read between it and the
//     first MFMA.
// k-permutation: a lane (i = l & 15, g = l >> 4) holds the chunk's k = {4g .. 4g+3, 16+4g .. 16+4g+3} as its bf16x8, the
// same assignment on both operands, so the contraction is a permutation of the 32 k-values.  Operand order is swapped
// (weights on the MFMA "A" side): a lane ends with 4 consecutive output columns of one row -> 16-byte epilogue accesses.
#include "common.h"

namespace rohm {
namespace {

constexpr int PBM = 144, PBN = 128, PBK = 32, PNT = 512;
constexpr int PA_UNITS = PBM * 8, PB_UNITS = PBN * 8;          // 16-byte units per raw chunk: 1152 + 1024
constexpr int PA_PASS = (PA_UNITS + PNT - 1) / PNT;            // 3 (the last one is a quarter populated)
constexpr int PB_PASS = PB_UNITS / PNT;                        // 2
constexpr int PPIECES = PA_PASS + PB_PASS;                     // LDS-DMA instructions per wave per chunk
constexpr int AP_PLANE = PBM * 64;                             // bytes of one plane of one A chunk: 144 rows x (4 groups x 8 bf16)
constexpr int PSTAGE = 3;                                      // raw fp32 stages

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));

// s_waitcnt vmcnt(n) lgkmcnt(0) (gfx9 encoding: vmcnt [3:0] + [15:14], expcnt [6:4] = 7, lgkmcnt [11:8])
#define PL_WAIT_VM_LGKM0(n) __builtin_amdgcn_s_waitcnt(((n) & 15) | (7 << 4) | (0 << 8) | (((n) >> 4) << 14))

// Plane image: row r holds its four 16-byte k-groups at 64 r + 16 (g ^ T[(r >> 2) & 3]), T = {0, 2, 3, 1}.  A ds_read_b128
// is served in four groups of 16 lanes which, for lane = (row i, group g), hold the rows {0-3, 12-15} of one g and {4-11} of the
// next: with 64-byte rows the bank base is 16 (r mod 4) dwords, so the four rows of a residue class must sit in four
// different k-group slots -- this T does that for all four lane groups (unswizzled the reads were 2-way conflicted).
__device__ __forceinline__ int pl_group_swz(int row) {
    const int x = (row >> 2) & 3;
    return (((x >> 1) ^ x) & 1) << 1 | (x >> 1);
}

__device__ __forceinline__ int pl_lds_off(int row, int slot) {   // float index inside a [rows][32] raw tile (as gemm_f32.hip)
    return row * PBK + ((slot ^ ((row >> 1) & 7)) << 2);
}

template <int NP>
struct Planes { u32x4 p[NP]; };

// the four values of one 16-byte fragment -> dwords 2 * HALF, 2 * HALF + 1 of every plane
template <int NP, int HALF>
__device__ __forceinline__ void split4(const f32x4& x, Planes<NP>& o) {
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
        o.p[0][2 * HALF + j] = __builtin_amdgcn_perm(hb[2 * j + 1], hb[2 * j], 0x07060302u);
        o.p[1][2 * HALF + j] = __builtin_amdgcn_perm(mb[2 * j + 1], mb[2 * j], 0x07060302u);
        if constexpr (NP == 3) o.p[2][2 * HALF + j] = __builtin_amdgcn_perm(lb[2 * j + 1], lb[2 * j], 0x07060302u);
    }
}

template <int EPI, int NP>
__global__ __launch_bounds__(PNT) void gemm_planes_kernel(GemmParams p) {
    constexpr int NPROD = (NP == 3) ? 6 : 3;
    constexpr int ia[6] = {2, 0, 1, 1, 0, 0}, ib[6] = {0, 2, 1, 0, 1, 0};    // plane pairs, smallest terms first
    constexpr int NCB = 2;                 // 16-wide column blocks per wave (32 columns)
    constexpr int NRW = 5;                 // row-block steps per wave (the second row half uses four)

    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* As = smem;                                   // [3][PBM * PBK]   raw fp32 stages
    float* Bs = smem + PSTAGE * PBM * PBK;              // [3][PBN * PBK]
    char* Ap = reinterpret_cast<char*>(smem + PSTAGE * (PBM + PBN) * PBK);     // [2][NP][PBM][64 B]   shared A planes
    float* lds_dummy = smem + PSTAGE * (PBM + PBN) * PBK + 2 * NP * AP_PLANE / 4;   // 2 KiB landing zone

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wn = wave & 3, wm = wave >> 2;
    const int li = lane & 15, lg = lane >> 4;
    const int wave_u = wave * 64;
    const int r0 = wm * NRW;               // first row block of this wave: 0 or 5

    const int tiles_n = p.N / PBN;
    const int tiles_m = p.M / PBM;
    const int tile = xcd_remap(blockIdx.x, tiles_m * tiles_n);
    const int m0 = (tile / tiles_n) * PBM;
    const int n0 = (tile % tiles_n) * PBN;

    // ---- global -> LDS staging by LDS-DMA (linear LDS image, XOR swizzle on the per-lane source slot) ------------------------
    const float* a_src[PA_PASS];
#pragma unroll
    for (int i = 0; i < PA_PASS; ++i) {
        const int u = tid + i * PNT;
        const int row = (u < PA_UNITS) ? (u >> 3) : 0;
        const int slot = (u & 7) ^ ((row >> 1) & 7);
        a_src[i] = p.A + (size_t)(m0 + row) * p.lda + slot * 4;
    }
    const float* b_src[PB_PASS];
#pragma unroll
    for (int i = 0; i < PB_PASS; ++i) {
        const int u = tid + i * PNT;
        const int row = u >> 3;
        const int slot = (u & 7) ^ ((row >> 1) & 7);
        b_src[i] = p.W + (size_t)(n0 + row) * p.ldw + slot * 4;
    }
    auto dma_a = [&](int st, int k0) {
#pragma unroll
        for (int i = 0; i < PA_PASS; ++i) {
            float* dst = As + st * (PBM * PBK) + (i * PNT + wave_u) * 4;
            // the last A pass is a quarter populated: the other waves fetch a (valid) row into the landing zone
            if (i == PA_PASS - 1) dst = (wave_u < PA_UNITS - (PA_PASS - 1) * PNT) ? dst : lds_dummy + (wave_u & 64) * 4;
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(a_src[i] + k0),
                                             (__attribute__((address_space(3))) void*)dst, 16, 0, 0);
        }
    };
    auto dma_b = [&](int st, int k0) {
#pragma unroll
        for (int i = 0; i < PB_PASS; ++i)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(b_src[i] + k0),
                                             (__attribute__((address_space(3))) void*)(Bs + st * (PBN * PBK) + (i * PNT + wave_u) * 4),
                                             16, 0, 0);
    };

    f32x4 acc[NRW * NCB];
#pragma unroll
    for (int i = 0; i < NRW * NCB; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};

    // this thread's i-th unit of the raw A tile (linear LDS position u): row u >> 3, LOGICAL 16-byte slot (u & 7) ^ swizzle ->
    // k = 4 slot .. 4 slot + 3 -> lane group g = slot & 3, half = slot >> 2 of the (row, g) bf16x8 fragment
    int ap_dst[PA_PASS];
#pragma unroll
    for (int i = 0; i < PA_PASS; ++i) {
        const int u = tid + i * PNT;
        const int row = u >> 3, slot = (u & 7) ^ ((row >> 1) & 7);
        ap_dst[i] = row * 64 + ((slot & 3) ^ pl_group_swz(row)) * 16 + (slot >> 2) * 8;
    }
    auto split_a_unit = [&](int i, int st, int pbuf) {            // raw stage st -> plane image pbuf
        if (i < PA_PASS - 1 || wave_u < PA_UNITS - (PA_PASS - 1) * PNT) {   // wave-uniform
            const f32x4 x = *reinterpret_cast<const f32x4*>(As + st * (PBM * PBK) + (tid + i * PNT) * 4);
            Planes<NP> t;
            split4<NP, 0>(x, t);
#pragma unroll
            for (int pl = 0; pl < NP; ++pl)
                *reinterpret_cast<u32x2*>(Ap + (pbuf * NP + pl) * AP_PLANE + ap_dst[i]) = u32x2{t.p[pl][0], t.p[pl][1]};
        }
    };
    const int ap_rd = (lg ^ pl_group_swz(li)) * 16;       // (r * 16 + li) >> 2 and li >> 2 agree modulo 4
    auto a_planes = [&](int pbuf, int r, Planes<NP>& o) {
#pragma unroll
        for (int pl = 0; pl < NP; ++pl)
            o.p[pl] = *reinterpret_cast<const u32x4*>(Ap + (pbuf * NP + pl) * AP_PLANE + (r * 16 + li) * 64 + ap_rd);
    };
    auto b_frag = [&](int st, int c, int ks) -> f32x4 {
        return *reinterpret_cast<const f32x4*>(Bs + st * (PBN * PBK) + pl_lds_off(wn * 32 + c * 16 + li, ks * 4 + lg));
    };

    // One K chunk: up to five row-block steps; step j multiplies row block r0 + j (NPROD MFMAs per 16 x 16 block, product-
    // major so that consecutive MFMAs hit different accumulators) and cuts a share of chunk k+1: A unit j (j < 3) into the
    // other plane image, half a column block of this wave's B fragments (j < 4) into the other register set.
    auto chunk = [&](Planes<NP> (&pbc)[NCB], Planes<NP> (&pbn)[NCB], int cur, int nxt, int sn, int sd, int kd) {
        // cur / nxt: plane images of this / the next chunk; sn: raw stage of the next chunk (cut here); sd, kd: raw stage
        // and K offset of the chunk after it, whose DMA pieces are issued behind the MFMAs of the first two steps
        Planes<NP> pa0, pa1;
        a_planes(cur, r0, pa0);
#pragma unroll
        for (int j = 0; j < NRW; ++j) {
            if (j == NRW - 1 && wm != 0) break;          // wave-uniform: the second row half has four row blocks
            Planes<NP>& pac = (j & 1) ? pa1 : pa0;
            Planes<NP>& pan = (j & 1) ? pa0 : pa1;
            if (j + 1 < NRW - 1 || (j + 1 == NRW - 1 && wm == 0)) a_planes(cur, r0 + j + 1, pan);
            f32x4 xb = f32x4{0.f, 0.f, 0.f, 0.f};
            if (j < 2 * NCB) xb = b_frag(sn, j >> 1, j & 1);
#pragma unroll
            for (int q = 6 - NPROD; q < 6; ++q)
#pragma unroll
                for (int c = 0; c < NCB; ++c)
                    acc[j * NCB + c] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, pbc[c].p[ib[q]]),
                                                                               __builtin_bit_cast(bf16x8, pac.p[ia[q]]),
                                                                               acc[j * NCB + c], 0, 0, 0);
            if (j == 0) dma_a(sd, kd);
            if (j == 1) dma_b(sd, kd);
            if (j < PA_PASS) split_a_unit(j, sn, nxt);
            if (j < 2 * NCB) {
                if (j & 1) split4<NP, 1>(xb, pbn[j >> 1]);
                else split4<NP, 0>(xb, pbn[j >> 1]);
            }
        }
    };

    // ---- prologue: chunks 0 and 1 in flight; chunk 0 cut into plane image 0 / the first B register set -------------------------
    const int nk = p.K / PBK;
    dma_a(0, 0);
    dma_b(0, 0);
    if (nk > 1) { dma_a(1, PBK); dma_b(1, PBK); }
    if (nk > 1) PL_WAIT_VM_LGKM0(PPIECES);
    else PL_WAIT_VM_LGKM0(0);
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
    Planes<NP> pbA[NCB], pbB[NCB];
#pragma unroll
    for (int i = 0; i < PA_PASS; ++i) split_a_unit(i, 0, 0);
#pragma unroll
    for (int c = 0; c < NCB; ++c) {
        split4<NP, 0>(b_frag(0, c, 0), pbA[c]);
        split4<NP, 1>(b_frag(0, c, 1), pbA[c]);
    }
    int s1 = 1, s2 = 2;                   // raw stages of chunks kc+1 and kc+2
    auto step = [&](Planes<NP> (&pbc)[NCB], Planes<NP> (&pbn)[NCB], int kc) {
        const int cur = kc & 1, nxt = cur ^ 1;
        __builtin_amdgcn_sched_barrier(0);
        PL_WAIT_VM_LGKM0(0);              // raw chunk kc+1 has landed (issued early in the previous chunk); my plane writes are done
        __builtin_amdgcn_s_barrier();     // ... for every wave: the plane image of chunk kc is complete
        __builtin_amdgcn_sched_barrier(0);
        // chunk kc+2 goes to the stage whose chunk (kc-1) was cut two barriers ago: its DMA needs no barrier of its own and
        // rides inside the MFMA steps.  The last two chunks re-fetch the last chunk instead of being predicated.
        chunk(pbc, pbn, cur, nxt, s1, s2, ((kc + 2 < nk) ? (kc + 2) : (nk - 1)) * PBK);
        s1 = s2;
        s2 = (s2 == 2) ? 0 : s2 + 1;
    };
    int kc = 0;
    for (; kc + 1 < nk; kc += 2) {
        step(pbA, pbB, kc);
        step(pbB, pbA, kc + 1);
    }
    if (kc < nk) step(pbA, pbB, kc);

    // ---- epilogue (whole tiles, 16-byte accesses): lane holds C[m][nb .. nb+3] ----------------------------------------------------
#pragma unroll
    for (int j = 0; j < NRW; ++j) {
        if (j == NRW - 1 && wm != 0) break;
#pragma unroll
        for (int c = 0; c < NCB; ++c) {
            const int m = m0 + (r0 + j) * 16 + li;
            const int nb = n0 + wn * 32 + c * 16 + lg * 4;
            const f32x4 a = acc[j * NCB + c];
            f32x4 v = a;
            if (p.bias) {
                const f32x4 b4 = *reinterpret_cast<const f32x4*>(p.bias + nb);
#pragma unroll
                for (int q = 0; q < 4; ++q) v[q] += b4[q];
            }
            if constexpr (EPI == EPI_BIAS_GELU) {
#pragma unroll
                for (int q = 0; q < 4; ++q) v[q] = gelu_erf(v[q]);
            }
            if constexpr (EPI == EPI_BIAS_RES) {
                const f32x4 rr = *reinterpret_cast<const f32x4*>(p.R + (size_t)m * p.ldr + nb);
#pragma unroll
                for (int q = 0; q < 4; ++q) v[q] += rr[q];
            }
            if constexpr (EPI == EPI_QKV) {
                if (nb < p.qcols) {      // qcols is a multiple of 4
#pragma unroll
                    for (int q = 0; q < 4; ++q) v[q] *= p.qscale;
                }
            }
            if constexpr (EPI == EPI_EMBED) {
                const int bidx = m / p.S, tok = m % p.S;
                const float* tp = (tok == 0) ? p.tab0 + (size_t)bidx * p.ldtab0 + nb : p.tab + (size_t)tok * p.ldtab + nb;
                const f32x4 tt = *reinterpret_cast<const f32x4*>(tp);
#pragma unroll
                for (int q = 0; q < 4; ++q) v[q] = (tok == 0 ? 0.f : a[q]) + tt[q];
            }
            *reinterpret_cast<f32x4*>(p.C + (size_t)m * p.ldc + nb) = v;
        }
    }
}

inline bool al16(const void* q) { return (((uintptr_t)q) & 15) == 0; }

template <int EPI, int NP>
int launch_planes(const GemmParams& p, hipStream_t s) {
    const int tiles = (p.M / PBM) * (p.N / PBN);
    const size_t lds = (size_t)PSTAGE * (PBM + PBN) * PBK * sizeof(float) + (size_t)2 * NP * AP_PLANE + 2048;
    static bool attr_set[64] = {};
    int dev = 0;
    ROHM_HIP_CHECK(hipGetDevice(&dev));
    if (dev < 64 && !attr_set[dev]) {
        ROHM_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_planes_kernel<EPI, NP>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        attr_set[dev] = true;
    }
    static const char* const kNames[] = {"gemm_bias", "gemm_bias_gelu", "gemm_bias_res", "gemm_qkv", "gemm_embed"};
    prof::Scope ps(kNames[EPI], 2.0 * p.M * p.N * p.K, 4.0 * ((double)p.M * p.K + (double)p.N * p.K + (double)p.M * p.N), s);
    hipLaunchKernelGGL((gemm_planes_kernel<EPI, NP>), dim3(tiles), dim3(PNT), lds, s, p);
    ROHM_LAUNCH_CHECK();
    return ROHM_OK;
}

template <int NP>
int launch_planes_epi(const GemmParams& p, int epi, hipStream_t s) {
    switch (epi) {
        case EPI_BIAS: return launch_planes<EPI_BIAS, NP>(p, s);
        case EPI_BIAS_GELU: return launch_planes<EPI_BIAS_GELU, NP>(p, s);
        case EPI_BIAS_RES: return launch_planes<EPI_BIAS_RES, NP>(p, s);
        case EPI_QKV: return launch_planes<EPI_QKV, NP>(p, s);
        case EPI_EMBED: return launch_planes<EPI_EMBED, NP>(p, s);
    }
    set_error("gemm_planes: unsupported epilogue %d", epi);
    return ROHM_ERR_UNSUPPORTED;
}

}  // namespace

// Whole 144 x 128 tiles, 16-byte aligned operands, enough tiles to fill three quarters of the chip: everything else
// (ragged problems, the transposed output head, split-K, convolutions) stays on the exact-fp32 kernel.
bool planes_gemm_applies(const GemmParams& p, int epi) {
    if (epi < EPI_BIAS || epi > EPI_EMBED) return false;
    if (p.M % PBM || p.N % PBN || p.K % PBK || p.K <= 0) return false;
    if ((long)(p.M / PBM) * (p.N / PBN) < 192) return false;
    if (p.lda % 4 || p.ldw % 4 || p.ldc % 4 || !al16(p.A) || !al16(p.W) || !al16(p.C) || !al16(p.bias)) return false;
    if (epi == EPI_BIAS_RES && (!p.R || p.ldr % 4 || !al16(p.R))) return false;
    if (epi == EPI_EMBED && (p.ldtab % 4 || p.ldtab0 % 4 || !al16(p.tab) || !al16(p.tab0) || p.S <= 0)) return false;
    if (epi == EPI_QKV && p.qcols % 4) return false;
    return true;
}

int launch_gemm_planes(const GemmParams& p, int epi, int nplane, hipStream_t s) {
    if (!planes_gemm_applies(p, epi)) {
        set_error("gemm_planes: problem %d x %d x %d (epilogue %d) is not made of whole 144 x 128 tiles", p.M, p.N, p.K, epi);
        return ROHM_ERR_UNSUPPORTED;
    }
    return nplane == 3 ? launch_planes_epi<3>(p, epi, s) : launch_planes_epi<2>(p, epi, s);
}

}  // namespace rohm
