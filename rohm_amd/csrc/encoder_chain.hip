// The GEMM chain of one nn.TransformerEncoderLayer as ONE launch (model/posenet.py:63-69, post-norm):
//
//     y   = norm1(h + out_proj(ctx))                      phase A   tile 144 x 512/G   (+ residual + LayerNorm, EPI_BIAS_RES_LN's form)
//     ff  = gelu(linear1(y))                              phase B   tile 144 x 1024/G
//     h   = norm2(y + linear2(ff))                        phase C   tile 144 x 512/G   (K = 1024)
//     qkv = in_proj(h)   of the NEXT layer (q pre-scaled) phase D   tile 144 x 1536/G  (absent behind the last layer)
//
// Why: with one launch per GEMM every launch pays ~2 us of prologue (first operands from L2 / HBM with idle matrix cores), its
// epilogue's store burst with all 256 CUs idle on the matrix side, and the ramp-down until the slowest of 256 tiles is done --
// 5-7 us of 36-115 us, four times per layer ("persistent per-layer kernel").  The dependency between these GEMMs is
// all-to-all inside a CLIP (the next GEMM contracts over the full 512 / 1024 columns of the clip's 144 rows) and nil across clips,
// and at B = 64 (32) every one of them is cut into exactly G = 4 (8) column tiles per clip.  So a workgroup is (clip g, part tn) for
// the whole chain: it computes its column tile of every phase, and between phases only the G workgroups of a clip meet -- a flag
// exchange through the L2 of the XCD they share (the block -> tile map of EPI_BIAS_RES_LN: partners dispatched back to back onto one
// XCD), while the next phase's WEIGHT chunks, which depend on nobody, are already on their way into LDS underneath the epilogue.
//
// Arithmetic: every tile is computed exactly as gemm_f32_kernel<BN, EPI, 0, true> computes it (same fragments, same k order, same
// LayerNorm statistics tree) -- the chain and the launch-per-GEMM path agree to the last bit or two (fma contraction of the epilogue
// expressions is the compiler's choice per instantiation; tests/test_gpu_chain.py).
// Exchange discipline: gemm_f32.hip's -- tags from (salt + index) + 64 x the workspace's pass counter, bounded waits that report
// into the error word, layout guard at create, fallback + re-run by the host (exchange.hip).
#include <type_traits>
#include "common.h"
#include "gemm_sched.h"
#include "attention_tile.h"

namespace rohm {

typedef float f32x16c __attribute__((ext_vector_type(16)));

namespace chain {

constexpr int BM = 144, BK = kGemmBK, NRB = BM / 16;
constexpr int A_UNITS = BM * 8, A_ITERS = (A_UNITS + 255) / 256;      // 16-byte units per A chunk; the fifth pass is half populated
constexpr int kMaxBN = 384;
// LDS: [A: 2 x 144 x 32] [W: 2 x BN x 32, BN <= 384] and, at a FIXED place behind the widest W buffers, the DMA landing zone and
// the statistics / bias-row zone -- so that the next phase's weight chunks can land while this phase's epilogue still uses its zone
constexpr int kZone = 2 * (BM + kMaxBN) * BK;                          // float offset
constexpr int kLdsFloats = kZone + 512 + 4 * BM * 2;
#ifdef ROHM_CHAIN_NO_SC1      // TIMING experiments only (results may be stale): what does the device-scope policy of the operand loads cost?
constexpr int kSc1 = 0;
#else
constexpr int kSc1 = 16;                                               // cache policy of a load that must come from L2 (device scope)
#endif

struct PhaseArgs {
    const float* A; int lda;
    const float* W; int ldw;
    float* C; int ldc;
    int K;
    const float* bias;
    const float* R; int ldr;                       // RES_LN
    const float* gamma; const float* beta; float eps; int ln_dim;
    int qcols; float qscale;                       // QKV
    int S; const float* tab; const float* tab0; int ldtab, ldtab0, tab_by_row;      // EMBED (common.h GemmParams)
    int m0, n0;
    float* row_stats; unsigned tag28; int tn, tiles_n;      // RES_LN: this row tile's slots, tag of this exchange
    unsigned* err; unsigned xcc1; int fault;
};

// Issue the first two K chunks of a phase's weight tile (rows n0 .. n0 + BN of W) into the two W buffers: 2 x BN * 8 / 256 LDS-DMA
// instructions per wave, no register staging.  Called while the PREVIOUS phase's epilogue is still to run.
template <int BN>
__device__ __forceinline__ void prefetch_w(const float* W, int ldw, int n0, float* smem, int tid, int wave_u) {
    constexpr int B_ITERS = BN * 8 / 256;
    float* Bs = smem + 2 * BM * BK;
#pragma unroll
    for (int buf = 0; buf < 2; ++buf)
#pragma unroll
        for (int i = 0; i < B_ITERS; ++i) {
            const int u = tid + i * 256;
            const int row = u >> 3;
            const int slot = (u & 7) ^ ((row >> 1) & 7);
            const float* src = W + (size_t)(n0 + row) * ldw + slot * 4 + buf * BK;
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                             (__attribute__((address_space(3))) void*)(Bs + buf * (BN * BK) + (i * 256 + wave_u) * 4), 16, 0, 0);
        }
}

// The G workgroups of a clip meet: everybody's stores of the phase are in L2 before anybody reads them.  flags[k] = (XCD id + 1) << 32
// | tag of workgroup k.  One lane per partner polls; bounded like every wait of this library, reported, never silently survived.
__device__ __forceinline__ void group_sync(unsigned long long* flags, int tn, int G, unsigned tag, unsigned xcc1, unsigned* err, int tid) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // this wave's stores are acknowledged by L2
    __syncthreads();
    // The flag by a PLAIN 8-byte store: it stays in the XCD's L2, where the partners' device-scope polls are served.  A device-scope (sc1)
    // store writes through to the memory side and drops the line from L2, so every poll pays the trip out (measured in
    // csrc/trajnet_resident.hip: 2.0 -> 1.2 us per 32-partner meeting; here -DROHM_CHAIN_FLAG_SC1 builds the old form for A/B runs).
    if (tid == 0) {
#ifdef ROHM_CHAIN_FLAG_SC1
        __hip_atomic_store(flags + tn, ((unsigned long long)xcc1 << 32) | tag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#else
        flags[tn] = ((unsigned long long)xcc1 << 32) | tag;
        asm volatile("" ::: "memory");
#endif
    }
#ifndef ROHM_CHAIN_NO_MEET      // TIMING experiment only (results may be stale): what do the meetings cost?
    if (tid < G && tid != tn) {
        for (int it = 0;; ++it) {
            const unsigned long long f = __hip_atomic_load(flags + tid, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if ((unsigned)f == tag && (f >> 32) != 0ull) {
                if ((unsigned)(f >> 32) != xcc1) __hip_atomic_store(err, 2u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                break;
            }
#ifndef ROHM_CHAIN_NO_SLEEP
            __builtin_amdgcn_s_sleep(1);
#endif
            if ((it & 127) == 127 && __hip_atomic_load(err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u) break;
            if (it > (1 << 19)) {
                __hip_atomic_store(err, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                break;
            }
        }
    }
#endif
    __syncthreads();
    // No acquire fence here: at agent scope it is `buffer_inv sc1`, which on this multi-XCD part also drops the L2's lines of ordinary
    // memory -- weights and activations come back from HBM, measured +5 us per meeting (gemm_chain 322 -> 344 us per layer,
    // profiles/r5_d_*).  Staleness is excluded where it could arise instead: every operand another workgroup wrote during this launch is
    // fetched by LDS-DMA with sc1 (device scope: never served by this CU's L1), and the plain loads behind a meeting read constants
    // (bias, gamma, beta) or the tile this very CU stored (residuals: a CU's own stores are coherent with its L1).
}

// One tile of one phase: C[m0.., n0..n0+BN) = epi(A[m0.., :K] . W[n0.., :K]^T).  PREF: the first two W chunks are in LDS (or on
// their way) already; SC1: A was written by partner workgroups of this launch -- fetch it from L2, never from this CU's L1.
// `after_loop()` runs once all waves are done with the staging buffers (the place to start the next phase's weights).
template <int BN, int EPI, bool PREF, bool SC1, bool REFETCH, typename AfterLoop>
__device__ __forceinline__ void gemm_phase(const PhaseArgs& p, float* smem, const int tid, AfterLoop&& after_loop) {
    static_assert(EPI == EPI_BIAS_RES_LN || EPI == EPI_BIAS_GELU || EPI == EPI_QKV || EPI == EPI_EMBED, "chain phases: LN tail, GELU, QKV, embed");
    constexpr int WN = BN / 4;
    // Every phase on v_mfma_f32_16x16x4_f32.  gemm_f32.hip's own launches run rows 0..127 of their 256- / 384-wide tiles on
    // v_mfma_f32_32x32x2_f32 (half the operand-register traffic per flop; measured faster there in round 2); INSIDE the stack the
    // all-16x16 form is the faster one: same box, two legs each, 22.36 -> 22.70 clips/s, stack launch 2812 -> 2773 us
    // (profiles/r5_j_*).  -DROHM_CHAIN_M32 builds the mixed form for A/B runs.
#ifdef ROHM_CHAIN_M32
    constexpr bool M32 = WN >= 64;
#else
    constexpr bool M32 = false;
#endif
    constexpr int NCB = WN / 16, NCB32 = WN / 32;
    constexpr int B_ITERS = BN * 8 / 256;
    constexpr int PIECES = A_ITERS + B_ITERS;
    constexpr int AUX = SC1 ? kSc1 : 0;
    static_assert(!(EPI == EPI_BIAS_RES_LN && M32), "the LayerNorm tail exists for the 16x16 layouts (BN <= 128)");

    float* As = smem;
    float* Bs = smem + 2 * BM * BK;
    float* lds_dummy = smem + kZone;
    float* const zone = lds_dummy + 512;          // statistics of the LayerNorm tail / bias row of the 384-wide tile

    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int li = lane & 15, lg = lane >> 4;
    const int li32 = lane & 31, lg32 = lane >> 5;
    const int wave_u = __builtin_amdgcn_readfirstlane(wave) * 64;
    const int m0 = p.m0, n0 = p.n0;

    // ---- staging -----------------------------------------------------------------------------------------------------------
    const float* a_src[A_ITERS];
#pragma unroll
    for (int i = 0; i < A_ITERS; ++i) {
        const int u = tid + i * 256;
        const int row = (u < A_UNITS) ? (u >> 3) : 0;
        const int slot = (u & 7) ^ ((row >> 1) & 7);
        a_src[i] = p.A + (size_t)(m0 + row) * p.lda + slot * 4;
    }
    const float* b_src[B_ITERS];
#pragma unroll
    for (int i = 0; i < B_ITERS; ++i) {
        const int u = tid + i * 256;
        const int row = u >> 3;
        const int slot = (u & 7) ^ ((row >> 1) & 7);
        b_src[i] = p.W + (size_t)(n0 + row) * p.ldw + slot * 4;
    }
    auto dma_a = [&](int buf, int k0) {
#pragma unroll
        for (int piece = 0; piece < A_ITERS; ++piece) {
            float* dst = As + buf * (BM * BK) + (piece * 256 + wave_u) * 4;
            if (piece == A_ITERS - 1 && A_UNITS % 256 != 0)
                dst = (wave_u < A_UNITS - (A_ITERS - 1) * 256) ? dst : lds_dummy + (wave_u & 64) * 4;
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(a_src[piece] + k0),
                                             (__attribute__((address_space(3))) void*)dst, 16, 0, AUX);
        }
    };
#ifdef ROHM_CHAIN_W_DEAD64        // TIMING experiment only (WRONG results): the 64-wide tiles' weight chunks as ordinary global loads into registers
    f32x4 wdead[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};      // that nobody multiplies: what does the register path cost to ISSUE?
#endif
    auto dma_b = [&](int buf, int k0) {
#ifdef ROHM_CHAIN_NO_W_DMA64      // TIMING experiment only (WRONG results: stale weights in LDS): what would taking W off the LDS-DMA path of the 64-wide tiles buy?
        if constexpr (BN == 64) return;
#endif
#ifdef ROHM_CHAIN_W_DEAD64
        if constexpr (BN == 64) {
            asm volatile("" ::"v"(wdead[0]), "v"(wdead[1]));      // "use" of the previous chunk's loads (they have landed: vmcnt(0) + barrier just above)
#pragma unroll
            for (int i = 0; i < B_ITERS; ++i) wdead[i] = *reinterpret_cast<const f32x4*>(b_src[i] + k0);
            return;
        }
#endif
#pragma unroll
        for (int i = 0; i < B_ITERS; ++i)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(b_src[i] + k0),
                                             (__attribute__((address_space(3))) void*)(Bs + buf * (BN * BK) + (i * 256 + wave_u) * 4), 16, 0, 0);
    };

    // ---- accumulators and fragments (gemm_f32.hip's layouts) ---------------------------------------------------------------------
    constexpr int N32 = M32 ? 4 * NCB32 : 1;
    constexpr int N16 = M32 ? NCB : NRB * NCB;
    f32x16c acc32[N32];
    f32x4 acc16[N16];
#pragma unroll
    for (int i = 0; i < N32; ++i)
#pragma unroll
        for (int q = 0; q < 16; ++q) acc32[i][q] = 0.f;
#pragma unroll
    for (int i = 0; i < N16; ++i) acc16[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    constexpr int FA = M32 ? 9 : NRB;
    constexpr int FB = M32 ? 2 * NCB32 + NCB : NCB;
    constexpr int READS = FA + FB;
    constexpr int MFMAS = M32 ? 32 * NCB32 + 4 * NCB : 4 * NRB * NCB;
    struct Frag { f32x4 a[FA]; f32x4 b[FB]; };
    Frag f0, f1;
    auto read_frags = [&](Frag& f, int buf, int ks) {
        const float* as = As + buf * (BM * BK);
        const float* bs = Bs + buf * (BN * BK) + wave * WN * BK;
        if constexpr (M32) {
#pragma unroll
            for (int s8 = 0; s8 < 2; ++s8) {
                const int slot = ks * 4 + s8 * 2 + lg32;
#pragma unroll
                for (int cb = 0; cb < NCB32; ++cb)
                    f.b[s8 * NCB32 + cb] = *reinterpret_cast<const f32x4*>(bs + lds_off(cb * 32 + li32, slot));
#pragma unroll
                for (int rb = 0; rb < 4; ++rb)
                    f.a[s8 * 4 + rb] = *reinterpret_cast<const f32x4*>(as + lds_off(rb * 32 + li32, slot));
            }
            const int slot16 = ks * 4 + lg;
            f.a[8] = *reinterpret_cast<const f32x4*>(as + lds_off(128 + li, slot16));
#pragma unroll
            for (int c = 0; c < NCB; ++c)
                f.b[2 * NCB32 + c] = *reinterpret_cast<const f32x4*>(bs + lds_off(c * 16 + li, slot16));
        } else {
            const int slot = ks * 4 + lg;
#pragma unroll
            for (int c = 0; c < NCB; ++c) f.b[c] = *reinterpret_cast<const f32x4*>(bs + lds_off(c * 16 + li, slot));
#pragma unroll
            for (int r = 0; r < NRB; ++r) f.a[r] = *reinterpret_cast<const f32x4*>(as + lds_off(r * 16 + li, slot));
        }
    };
    auto mma_half = [&](const Frag& f) {      // weights on the MFMA "A" side: a lane ends with 4 consecutive output columns
        if constexpr (M32) {
#pragma unroll
            for (int s8 = 0; s8 < 2; ++s8)
#pragma unroll
                for (int j = 0; j < 4; ++j)
#pragma unroll
                    for (int rb = 0; rb < 4; ++rb)
#pragma unroll
                        for (int cb = 0; cb < NCB32; ++cb)
                            acc32[rb * NCB32 + cb] = __builtin_amdgcn_mfma_f32_32x32x2f32(f.b[s8 * NCB32 + cb][j], f.a[s8 * 4 + rb][j],
                                                                                          acc32[rb * NCB32 + cb], 0, 0, 0);
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int c = 0; c < NCB; ++c)
                    acc16[c] = __builtin_amdgcn_mfma_f32_16x16x4f32(f.b[2 * NCB32 + c][j], f.a[8][j], acc16[c], 0, 0, 0);
        } else {
#pragma unroll
            for (int r = 0; r < NRB; ++r)
#pragma unroll
                for (int c = 0; c < NCB; ++c)
#pragma unroll
                    for (int j = 0; j < 4; ++j)
                        acc16[r * NCB + c] = __builtin_amdgcn_mfma_f32_16x16x4f32(f.b[c][j], f.a[r][j], acc16[r * NCB + c], 0, 0, 0);
        }
    };

    // ---- prologue -----------------------------------------------------------------------------------------------------------
    const int nk = p.K / BK;                       // >= 16 here
#ifdef ROHM_CHAIN_NO_COL_LDS      // experiment builds: with the all-16x16 form a 384-wide tile has 6 column groups per lane, not 18
    constexpr bool COL_LDS = false;
#else
    constexpr bool COL_LDS = BN >= 384;            // the 384-wide tile takes its bias row through LDS (gemm_f32.hip's choice for that width)
#endif
    if constexpr (!PREF) { dma_b(0, 0); dma_b(1, BK); }
    if constexpr (COL_LDS) {
        if (wave == 0) {
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const int unit = h * 64 + lane;
                const float* src = p.bias + n0 + (unit < BN / 4 ? unit : BN / 4 - 1) * 4;
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                                 (__attribute__((address_space(3))) void*)(zone + h * 256), 16, 0, 0);
            }
        }
    }
    dma_a(0, 0);
    dma_a(1, BK);

    // ---- epilogue operands (requested at the top of the peeled last chunk: they land under its MFMAs) --------------------------
    const int nw = n0 + wave * WN;
#ifdef ROHM_CHAIN_NO_COL_LDS
    constexpr bool COL_LDS_TILE = false;
#else
    constexpr bool COL_LDS_TILE = BN >= 384;       // (= COL_LDS of the prologue: its bias row comes from LDS, nothing to request early)
#endif
    struct ColOps { f32x4 bias, g4, b4; };
    constexpr bool RES = (EPI == EPI_BIAS_RES_LN);
    constexpr bool HAS_RES = RES || EPI == EPI_EMBED;      // a per-unit operand from memory: the residual, or the table row of the embedding
    // the last chunk's iteration is peeled (no wait / barrier / prefetch in it, the epilogue's operands requested at its top) for tiles up to
    // ROHM_CHAIN_PEEL_MAX columns: 256 like gemm_f32.hip (measured: peeling the 384-wide tile too loses 0.6 %, profiles/r5_k_*)
#ifndef ROHM_CHAIN_PEEL_MAX
#define ROHM_CHAIN_PEEL_MAX 256
#endif
    constexpr bool PEEL = BN <= ROHM_CHAIN_PEEL_MAX;
    constexpr bool EARLY = PEEL && !COL_LDS_TILE;
    const f32x4 zero4 = f32x4{0.f, 0.f, 0.f, 0.f};
    auto load_col = [&](int nb) __attribute__((always_inline)) {
        ColOps o{zero4, zero4, zero4};
        if constexpr (EPI == EPI_EMBED) { /* the biases are part of the table rows */ }
        else if constexpr (COL_LDS) o.bias = *reinterpret_cast<const f32x4*>(zone + (nb - n0));
        else o.bias = *reinterpret_cast<const f32x4*>(p.bias + nb);
        if constexpr (RES) {
            o.g4 = *reinterpret_cast<const f32x4*>(p.gamma + nb);
            o.b4 = *reinterpret_cast<const f32x4*>(p.beta + nb);
        }
        return o;
    };
    constexpr int NUNIT = M32 ? 16 * NCB32 + NCB : NCB * NRB;
    constexpr int NCG = M32 ? 4 * NCB32 + NCB : NCB;
    auto for_units = [&](auto&& fn) __attribute__((always_inline)) {
        if constexpr (M32) {
#pragma unroll
            for (int rb = 0; rb < 4; ++rb)
#pragma unroll
                for (int cb = 0; cb < NCB32; ++cb)
#pragma unroll
                    for (int qq = 0; qq < 4; ++qq) {
                        const f32x16c& a = acc32[rb * NCB32 + cb];
                        fn((rb * NCB32 + cb) * 4 + qq, cb * 4 + qq, m0 + rb * 32 + li32, nw + cb * 32 + 8 * qq + 4 * lg32,
                           f32x4{a[4 * qq], a[4 * qq + 1], a[4 * qq + 2], a[4 * qq + 3]});
                    }
#pragma unroll
            for (int c = 0; c < NCB; ++c) fn(16 * NCB32 + c, 4 * NCB32 + c, m0 + 128 + li, nw + c * 16 + lg * 4, acc16[c]);
        } else {
            // row-major over the lane's units: the NCB 64-byte pieces of a row's WN columns are stored by consecutive instructions, so
            // the halves of a 128-byte line reach L2 together (column-major order -- gemm_f32.hip's -- left 9 stores between them and
            // the wide tiles' HBM-side write traffic 20 % above the algorithmic, profiles/r5_n_pmc_traffic.txt)
#ifndef ROHM_CHAIN_STORE_COLMAJOR
#pragma unroll
            for (int r = 0; r < NRB; ++r)
#pragma unroll
                for (int c = 0; c < NCB; ++c) fn(c * NRB + r, c, m0 + r * 16 + li, nw + c * 16 + lg * 4, acc16[r * NCB + c]);
#else
#pragma unroll
            for (int c = 0; c < NCB; ++c)
#pragma unroll
                for (int r = 0; r < NRB; ++r) fn(c * NRB + r, c, m0 + r * 16 + li, nw + c * 16 + lg * 4, acc16[r * NCB + c]);
#endif
        }
    };
    ColOps col[NCG];
    f32x4 res[HAS_RES ? NUNIT : 1];
    auto request_ops = [&]() __attribute__((always_inline)) {
        if constexpr (M32) {
#pragma unroll
            for (int cb = 0; cb < NCB32; ++cb)
#pragma unroll
                for (int qq = 0; qq < 4; ++qq) col[cb * 4 + qq] = load_col(nw + cb * 32 + 8 * qq + 4 * lg32);
#pragma unroll
            for (int c = 0; c < NCB; ++c) col[4 * NCB32 + c] = load_col(nw + c * 16 + lg * 4);
        } else {
#pragma unroll
            for (int c = 0; c < NCB; ++c) col[c] = load_col(nw + c * 16 + lg * 4);
        }
        if constexpr (RES)
            for_units([&](int i, int, int m, int nb, f32x4) __attribute__((always_inline)) {
                res[i] = *reinterpret_cast<const f32x4*>(p.R + (size_t)m * p.ldr + nb);
            });
        if constexpr (EPI == EPI_EMBED)      // gemm_f32.hip load_res: the timestep token's row for token 0, else the positional (or per-row) table
            for_units([&](int i, int, int m, int nb, f32x4) __attribute__((always_inline)) {
                const int bidx = m / p.S, tok = m % p.S;
                const float* tp = (tok == 0) ? p.tab0 + (size_t)bidx * p.ldtab0 + nb : p.tab + (size_t)(p.tab_by_row ? m : tok) * p.ldtab + nb;
                res[i] = *reinterpret_cast<const f32x4*>(tp);
            });
    };

    // chunk 0 has landed when at most the A pieces of chunk 1 are outstanding (everything older -- the prefetched or just issued W
    // chunks, the bias row, A chunk 0 -- retires first: loads return in order, and no store of this wave is in flight here)
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(A_ITERS) : "memory");
    __syncthreads();
    read_frags(f0, 0, 0);
    constexpr int NG = READS;
    constexpr int MF = (MFMAS + NG - 1) / NG;
    // chunk kc: [reads of half 1 | MFMAs of half 0] wait + barrier [DMA of chunk kc + 2, reads of chunk kc + 1's half 0 | MFMAs of half 1].
    // LAST: the final chunk, peeled -- nothing to wait for or prefetch, the epilogue's operands requested at its top.  DMA = false: an
    // iteration behind which there is no chunk kc + 2: gemm_f32.hip re-fetches the last chunk there to keep ONE loop body (REFETCH); or the
    // body is instantiated once more without the DMA.  Measured inside the stack, same box, two rounds (profiles/r5_m_*): without the
    // redundant DMA +0.5 % at B = 64 (4 parts per clip), -1.0 % at B = 32 (8 parts: its 64- / 128- / 192-wide tiles) -- so the kernel
    // picks per G.
    auto chunk = [&](int kc, auto last_tag, auto dma_tag) __attribute__((always_inline)) {
        constexpr bool LAST = decltype(last_tag)::value;
        constexpr bool DMA = decltype(dma_tag)::value;
        const int buf = kc & 1;
        if constexpr (LAST && EARLY) request_ops();
        read_frags(f1, buf, 1);
        mma_half(f0);
        SchedGroups<0, NG, MF, READS, 0, 0>::run();
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (!LAST) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            __builtin_amdgcn_sched_barrier(0);
            if constexpr (DMA) {
                const int kn = ((kc + 2 < nk) ? (kc + 2) : (nk - 1)) * BK;
                dma_a(buf, kn);
                dma_b(buf, kn);
            }
            read_frags(f0, buf ^ 1, 0);
            mma_half(f1);
            SchedGroups<0, NG, MF, READS, DMA ? PIECES : 0, 1>::run();
        } else {
            mma_half(f1);
        }
        __builtin_amdgcn_sched_barrier(0);
    };
    using DmaTail = std::integral_constant<bool, REFETCH>;      // do the iterations without a chunk kc + 2 issue a (redundant) DMA?
    if constexpr (PEEL) {
        for (int kc = 0; kc + 2 < nk; ++kc) chunk(kc, std::false_type{}, std::true_type{});
        chunk(nk - 2, std::false_type{}, DmaTail{});
        chunk(nk - 1, std::true_type{}, std::false_type{});
    } else {
        for (int kc = 0; kc + 2 < nk; ++kc) chunk(kc, std::false_type{}, std::true_type{});
        chunk(nk - 2, std::false_type{}, DmaTail{});
        chunk(nk - 1, std::false_type{}, DmaTail{});
    }
    // every DMA of this phase has landed (the re-fetched last chunk included) and every wave is done reading the staging buffers:
    // they belong to the next phase from here on
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    after_loop();
    if constexpr (!EARLY) request_ops();

    // ---- epilogue -----------------------------------------------------------------------------------------------------------
    if constexpr (EPI == EPI_BIAS_RES_LN) {
        // gemm_f32.hip EPI_BIAS_RES_LN, statement for statement (same statistics, same merge tree: bit-identical results)
#pragma unroll
        for (int c = 0; c < NCB; ++c)
#pragma unroll
            for (int r = 0; r < NRB; ++r) {
                f32x4& a = acc16[r * NCB + c];
                const f32x4 rr = res[c * NRB + r];
#pragma unroll
                for (int q = 0; q < 4; ++q) a[q] = (a[q] + col[c].bias[q]) + rr[q];
            }
#ifdef ROHM_CHAIN_NO_LN_EPI      // TIMING experiment only (WRONG results: no LayerNorm): what does the LayerNorm part of the epilogue cost?
#pragma unroll
        for (int r = 0; r < NRB; ++r)
#pragma unroll
            for (int c = 0; c < NCB; ++c)
                *reinterpret_cast<f32x4*>(p.C + (size_t)(m0 + r * 16 + li) * p.ldc + nw + c * 16 + lg * 4) = acc16[r * NCB + c];
        return;
#endif
        typedef unsigned rohm_u2 __attribute__((ext_vector_type(2)));
        auto merge_swap = [](float& m, float& q2, float n, bool far) __attribute__((always_inline)) {
            rohm_u2 tm, tq;
            if (far) { tm = __builtin_amdgcn_permlane32_swap(__float_as_uint(m), __float_as_uint(m), false, false);
                       tq = __builtin_amdgcn_permlane32_swap(__float_as_uint(q2), __float_as_uint(q2), false, false); }
            else     { tm = __builtin_amdgcn_permlane16_swap(__float_as_uint(m), __float_as_uint(m), false, false);
                       tq = __builtin_amdgcn_permlane16_swap(__float_as_uint(q2), __float_as_uint(q2), false, false); }
            const float ma = __uint_as_float(tm[0]), mb = __uint_as_float(tm[1]);
            const float d = mb - ma;
            m = 0.5f * (ma + mb);
            q2 = (__uint_as_float(tq[0]) + __uint_as_float(tq[1])) + 0.5f * n * d * d;
        };
        float* const part = zone;                       // [4 waves][BM][2]
        constexpr float kLane = (float)(4 * NCB);
#pragma unroll
        for (int r = 0; r < NRB; ++r) {
            float sm = 0.f;
#pragma unroll
            for (int c = 0; c < NCB; ++c)
#pragma unroll
                for (int k = 0; k < 4; ++k) sm += acc16[r * NCB + c][k];
            float m = sm * (1.0f / kLane), q2 = 0.f;
#pragma unroll
            for (int c = 0; c < NCB; ++c)
#pragma unroll
                for (int k = 0; k < 4; ++k) { const float d = acc16[r * NCB + c][k] - m; q2 += d * d; }
            merge_swap(m, q2, kLane, true);
            merge_swap(m, q2, 2.0f * kLane, false);
            if (lg == 0) *reinterpret_cast<f32x2*>(part + (wave * BM + r * 16 + li) * 2) = f32x2{m, q2};
        }
        __syncthreads();
        const int tn = p.tn, tiles_n = p.tiles_n;
        const unsigned xcc1 = p.xcc1, ep28 = p.tag28;
        const unsigned tag = (ep28 << 4) | xcc1;
        float* const row_stats = p.row_stats;
        if (tid < BM) {
            auto merge = [](float ma, float qa, float mb, float qb, float n, float& m, float& q2) __attribute__((always_inline)) {
                const float d = mb - ma;
                m = 0.5f * (ma + mb);
                q2 = (qa + qb) + 0.5f * n * d * d;
            };
            f32x2 w4[4];
#pragma unroll
            for (int w = 0; w < 4; ++w) w4[w] = *reinterpret_cast<const f32x2*>(part + (w * BM + tid) * 2);
            float m01, q01, m23, q23, a, b;
            merge(w4[0][0], w4[0][1], w4[1][0], w4[1][1], (float)WN, m01, q01);
            merge(w4[2][0], w4[2][1], w4[3][0], w4[3][1], (float)WN, m23, q23);
            merge(m01, q01, m23, q23, 2.0f * (float)WN, a, b);
            const unsigned pub = (p.fault && tn == 0) ? (tag ^ 0x80000000u) : tag;
            *reinterpret_cast<f32x4*>(row_stats + ((size_t)tn * BM + tid) * 4) = f32x4{a, __uint_as_float(pub), b, __uint_as_float(pub)};
            float mk[8], qk[8];
            for (int it = 0;; ++it) {
                unsigned long long lo[8], hi[8];
#pragma unroll
                for (int k = 0; k < 8; ++k) {
                    lo[k] = hi[k] = 0ull;
                    if (k < tiles_n && k != tn) {
                        const unsigned long long* sp = reinterpret_cast<const unsigned long long*>(row_stats + ((size_t)k * BM + tid) * 4);
                        lo[k] = __hip_atomic_load(sp, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        hi[k] = __hip_atomic_load(sp + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    }
                }
                bool ok = true, same_xcd = true;
#pragma unroll
                for (int k = 0; k < 8; ++k) {
                    mk[k] = 0.f; qk[k] = 0.f;
                    if (k >= tiles_n) continue;
                    if (k == tn) { mk[k] = a; qk[k] = b; continue; }
                    const unsigned tl = (unsigned)(lo[k] >> 32), th = (unsigned)(hi[k] >> 32);
                    ok = ok && (tl >> 4) == ep28 && (th >> 4) == ep28 && (tl & 15u) != 0u && (th & 15u) != 0u;
                    same_xcd = same_xcd && (tl & 15u) == xcc1 && (th & 15u) == xcc1;
                    mk[k] = __uint_as_float((unsigned)lo[k]);
                    qk[k] = __uint_as_float((unsigned)hi[k]);
                }
                if (ok) {
                    if (!same_xcd) __hip_atomic_store(p.err, 2u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    break;
                }
#ifdef ROHM_CHAIN_NO_LN_WAIT      // TIMING experiment only (WRONG statistics: whatever the slots hold): what does waiting for the partners' statistics cost?
                break;
#endif
                __builtin_amdgcn_s_sleep(1);
                if ((it & 127) == 127 && __hip_atomic_load(p.err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u) break;
                if (it > (1 << 19)) {
                    __hip_atomic_store(p.err, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    break;
                }
            }
            constexpr float kT = (float)BN;
            if (tiles_n > 1) {
                merge(mk[0], qk[0], mk[1], qk[1], kT, mk[0], qk[0]);
                if (tiles_n > 2) merge(mk[2], qk[2], mk[3], qk[3], kT, mk[2], qk[2]);
                if (tiles_n > 4) { merge(mk[4], qk[4], mk[5], qk[5], kT, mk[4], qk[4]); merge(mk[6], qk[6], mk[7], qk[7], kT, mk[6], qk[6]); }
            }
            if (tiles_n > 2) {
                merge(mk[0], qk[0], mk[2], qk[2], 2.0f * kT, mk[0], qk[0]);
                if (tiles_n > 4) merge(mk[4], qk[4], mk[6], qk[6], 2.0f * kT, mk[4], qk[4]);
            }
            if (tiles_n > 4) merge(mk[0], qk[0], mk[4], qk[4], 4.0f * kT, mk[0], qk[0]);
            const float mu = mk[0];
            const float var = qk[0] / (float)p.ln_dim;
            *reinterpret_cast<f32x2*>(part + tid * 2) = f32x2{mu, 1.0f / sqrtf(var + p.eps)};
        }
        __syncthreads();
#pragma unroll
        for (int r = 0; r < NRB; ++r) {
            const f32x2 mrv = *reinterpret_cast<const f32x2*>(part + (r * 16 + li) * 2);
#pragma unroll
            for (int c = 0; c < NCB; ++c) {
                const f32x4 a = acc16[r * NCB + c];
                f32x4 v;
#pragma unroll
                for (int q = 0; q < 4; ++q) v[q] = (a[q] - mrv[0]) * mrv[1] * col[c].g4[q] + col[c].b4[q];
                *reinterpret_cast<f32x4*>(p.C + (size_t)(m0 + r * 16 + li) * p.ldc + nw + c * 16 + lg * 4) = v;
            }
        }
    } else if constexpr (EPI == EPI_EMBED) {
        for_units([&](int i, int, int m, int nb, f32x4 a) __attribute__((always_inline)) {
            const int tok = m % p.S;
            f32x4 v;
#pragma unroll
            for (int q = 0; q < 4; ++q) v[q] = (tok == 0 ? 0.f : a[q]) + res[i][q];
            *reinterpret_cast<f32x4*>(p.C + (size_t)m * p.ldc + nb) = v;
        });
    } else {
        for_units([&](int, int cg, int m, int nb, f32x4 a) __attribute__((always_inline)) {
            f32x4 v;
#pragma unroll
            for (int q = 0; q < 4; ++q) v[q] = a[q] + col[cg].bias[q];
#ifndef ROHM_CHAIN_NO_GELU      // TIMING experiment only (WRONG results): what does the erf-form GELU cost in linear1's epilogue?
            if constexpr (EPI == EPI_BIAS_GELU) {
#pragma unroll
                for (int q = 0; q < 4; ++q) v[q] = gelu_erf(v[q]);
            }
#endif
            if constexpr (EPI == EPI_QKV) {
                if (nb < p.qcols) {
#pragma unroll
                    for (int q = 0; q < 4; ++q) v[q] *= p.qscale;
                }
            }
            *reinterpret_cast<f32x4*>(p.C + (size_t)m * p.ldc + nb) = v;
        });
    }
}

// The stack's closing phase in the sampling loop (StackParams::tail): OutputProcess + DDPM update + the next step's pack for clip g.
// out^T is not formed: the MFMAs take the TOKENS on their row side, so a lane ends with four consecutive tokens of one channel --
// contiguous in the [B, C, 1, T] tensors that x_t, the noise, x_prev and x0 live in.  Work split: the clip's 17 column blocks (272
// channels) x 9 row blocks of 16 x 16 outputs; every wave owns one column block over 9 (G = 4) or 5 / 4 (G = 8) row blocks, and the
// 17th column block's nine blocks go one each to the waves with the lightest load -- at most 10 (5) blocks per wave.  Staging like
// gemm_phase (h by sc1 LDS-DMA: the partners wrote it in this launch), a plain two-barrier double-buffered loop: the phase is ~1 %
// of the launch.
template <int G>
__device__ __forceinline__ void head_tail_phase(const StackParams& p, const int g, const int tn, float* smem, const int tid) {
    constexpr int NC_OWN = 16 / G;                 // own column blocks of this workgroup
    constexpr int NW = NC_OWN * 16 + 16;           // weight rows staged per chunk: own + the shared 17th block
    constexpr int NRMAX = (G == 4) ? 9 : 5;
    constexpr int W_UNITS = NW * 8, W_ITERS = (W_UNITS + 255) / 256;
    constexpr int PIECES = A_ITERS + W_ITERS;
    // three staging buffers [A: 144 x 32 | W: NW x 32] each: chunk kc + 2 is issued at the top of iteration kc into the buffer that
    // iteration kc - 1 read, so ONE barrier per chunk orders everything (86 KB of the 135 KB the GEMM phases stage in)
    constexpr int kBuf = (BM + NW) * BK;
    static_assert(3 * kBuf <= kZone, "head phase: staging buffers overrun the zone");
    float* lds_dummy = smem + kZone;
    const int lane = tid & 63, li = lane & 15, lg = lane >> 4;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wave_u = wave * 64;
    const int D = p.D, m0 = g * BM;
    int c_loc, r0, nr, e;
    if constexpr (G == 4) { c_loc = wave; r0 = 0; nr = 9; e = tn * 4 + wave; }
    else { c_loc = wave >> 1; r0 = (wave & 1) ? 5 : 0; nr = (wave & 1) ? 4 : 5; e = (wave & 1) ? tn * 2 + (wave >> 1) : 99; }
    const bool has_e = e < NRB;

    const float* a_src[A_ITERS];
#pragma unroll
    for (int i = 0; i < A_ITERS; ++i) {
        const int u = tid + i * 256;
        const int row = (u < A_UNITS) ? (u >> 3) : 0;
        const int slot = (u & 7) ^ ((row >> 1) & 7);
        a_src[i] = p.h + (size_t)(m0 + row) * D + slot * 4;
    }
    const float* w_src[W_ITERS];
#pragma unroll
    for (int i = 0; i < W_ITERS; ++i) {
        const int u = tid + i * 256;
        const int lrow = (u < W_UNITS) ? (u >> 3) : 0;
        const int slot = (u & 7) ^ ((lrow >> 1) & 7);
        const int grow = (lrow < NC_OWN * 16) ? tn * NC_OWN * 16 + lrow : 256 + (lrow - NC_OWN * 16);
        w_src[i] = p.t_out_w + (size_t)grow * D + slot * 4;
    }
    auto dma = [&](int buf, int k0) {
#pragma unroll
        for (int i = 0; i < A_ITERS; ++i) {
            float* dst = smem + buf * kBuf + (i * 256 + wave_u) * 4;
            if (i == A_ITERS - 1 && A_UNITS % 256 != 0)
                dst = (wave_u < A_UNITS - (A_ITERS - 1) * 256) ? dst : lds_dummy + (wave_u & 64) * 4;
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(a_src[i] + k0),
                                             (__attribute__((address_space(3))) void*)dst, 16, 0, kSc1);
        }
#pragma unroll
        for (int i = 0; i < W_ITERS; ++i) {
            float* dst = smem + buf * kBuf + BM * BK + (i * 256 + wave_u) * 4;
            if (i == W_ITERS - 1 && W_UNITS % 256 != 0)
                dst = (wave_u < W_UNITS - (W_ITERS - 1) * 256) ? dst : lds_dummy + (wave_u & 64) * 4;
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(w_src[i] + k0),
                                             (__attribute__((address_space(3))) void*)dst, 16, 0, 0);
        }
    };

    f32x4 acc[NRMAX], acc_e = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int r = 0; r < NRMAX; ++r) acc[r] = f32x4{0.f, 0.f, 0.f, 0.f};
    const int nk = D / BK;
    dma(0, 0);
    dma(1, BK);

    // The epilogue's operands -- x_t and the noise of this lane's outputs: four consecutive tokens of one channel per block, i.e. 16
    // contiguous bytes of the [B, C, 1, T] tensors (rows of T = 143 floats: 4-byte aligned only, which global loads of this target
    // accept) -- are requested HERE, behind the first two chunks' DMAs, and land underneath the main loop.  Block (0, lg 0) starts at
    // token 0, which has no output: its first element is the float in front of the row (never stored).
    typedef float f32x4u __attribute__((ext_vector_type(4), aligned(4)));
    const int C = p.t_C, T = p.t_T, traj = p.t_traj, S = BM;
    float* const x = p.t_x;
    const float* const nzp = p.t_noise;
    const size_t row_own = ((size_t)g * C + traj + (tn * NC_OWN + c_loc) * 16 + li) * T;
    const size_t row_sh = ((size_t)g * C + traj + 256 + li) * T;
    f32x4 xt[NRMAX], nz[NRMAX], xt_e = f32x4{0.f, 0.f, 0.f, 0.f}, nz_e = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int r = 0; r < NRMAX; ++r) {
        xt[r] = nz[r] = f32x4{0.f, 0.f, 0.f, 0.f};
        if (r < nr) {
            const size_t idx = row_own + (size_t)((r0 + r) * 16 + 4 * lg) - 1;
            xt[r] = *reinterpret_cast<const f32x4u*>(x + idx);
            if (nzp) nz[r] = *reinterpret_cast<const f32x4u*>(nzp + idx);
        }
    }
    if (has_e) {
        const size_t idx = row_sh + (size_t)(e * 16 + 4 * lg) - 1;
        xt_e = *reinterpret_cast<const f32x4u*>(x + idx);
        if (nzp) nz_e = *reinterpret_cast<const f32x4u*>(nzp + idx);
    }
    const float bias_own = p.t_out_b[(tn * NC_OWN + c_loc) * 16 + li], bias_sh = p.t_out_b[256 + li];
    // Software pipeline (gemm_phase's order): the fragments of the NEXT half chunk are requested before the MFMAs of the current one, so
    // no MFMA group waits for LDS: chunk kc: [reads of half 1 | MFMAs of half 0]  wait + barrier  [DMA of chunk kc + 2, reads of chunk
    // kc + 1's half 0 | MFMAs of half 1].  hipcc left to itself emits read - wait - 4 MFMAs - read - wait ... (every fragment's LDS latency
    // exposed: the phase ran at 0.55 of its MFMA floor); the waits go through the builtin so that it sees the queue drain, sched_barrier
    // keeps the stages apart.
    struct HFrag { f32x4 a[NRMAX]; f32x4 ae, w, wsh; };
    auto read_h = [&](HFrag& f, int b, int ks) __attribute__((always_inline)) {
        const float* as = smem + b * kBuf;
        const float* ws = as + BM * BK;
        const int slot = ks * 4 + lg;
        f.w = *reinterpret_cast<const f32x4*>(ws + lds_off(c_loc * 16 + li, slot));
#pragma unroll
        for (int r = 0; r < NRMAX; ++r)
            if (r < nr) f.a[r] = *reinterpret_cast<const f32x4*>(as + lds_off((r0 + r) * 16 + li, slot));
        if (has_e) {
            f.ae = *reinterpret_cast<const f32x4*>(as + lds_off(e * 16 + li, slot));
            f.wsh = *reinterpret_cast<const f32x4*>(ws + lds_off(NC_OWN * 16 + li, slot));
        }
    };
    auto mma_h = [&](const HFrag& f) __attribute__((always_inline)) {
#pragma unroll
        for (int r = 0; r < NRMAX; ++r)
            if (r < nr) {
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[r] = __builtin_amdgcn_mfma_f32_16x16x4f32(f.a[r][j], f.w[j], acc[r], 0, 0, 0);
            }
        if (has_e) {
#pragma unroll
            for (int j = 0; j < 4; ++j) acc_e = __builtin_amdgcn_mfma_f32_16x16x4f32(f.ae[j], f.wsh[j], acc_e, 0, 0, 0);
        }
    };
    HFrag f0, f1;
    f0.ae = f0.wsh = f1.ae = f1.wsh = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int r = 0; r < NRMAX; ++r) f0.a[r] = f1.a[r] = f32x4{0.f, 0.f, 0.f, 0.f};
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // chunks 0 and 1 (and the epilogue operands behind them) have landed
    __syncthreads();
    read_h(f0, 0, 0);
    int buf = 0;
    for (int kc = 0; kc < nk; ++kc) {
        AT_WAIT_LGKM0();                               // f0 is in registers (requested one MFMA stage ago)
        __builtin_amdgcn_sched_barrier(0);
        read_h(f1, buf, 1);
        __builtin_amdgcn_sched_barrier(0);
        mma_h(f0);
        __builtin_amdgcn_sched_barrier(0);
        const int nbuf = buf == 2 ? 0 : buf + 1;
        if (kc + 1 < nk) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // chunk kc + 1 (issued a whole chunk of MFMAs ago)
            __syncthreads();                           // ... is visible to every wave; every wave has read chunk kc - 1 and half 0 of chunk kc
            __builtin_amdgcn_sched_barrier(0);
            if (kc + 2 < nk) dma(buf >= 1 ? buf - 1 : 2, (kc + 2) * BK);      // (kc + 2) % 3: the buffer chunk kc - 1 lived in
            AT_WAIT_LGKM0();                           // f1 is in registers
            __builtin_amdgcn_sched_barrier(0);
            read_h(f0, nbuf, 0);
            __builtin_amdgcn_sched_barrier(0);
        } else {
            AT_WAIT_LGKM0();
        }
        mma_h(f1);
        __builtin_amdgcn_sched_barrier(0);
        buf = nbuf;
    }

    // ---- epilogue: bias, x0, the ancestral update in place, the next step's pack -------------------------------------------------
    const float c1 = p.t_c1, c2 = p.t_c2, sg = p.t_sigma;
    float* const x0o = p.t_x0;
    float* const apk = p.t_apack;
    auto emit = [&](int rblk, size_t row, int ch, float bias, const f32x4& a, const f32x4& xv, const f32x4& nv) __attribute__((always_inline)) {
        const int tok0 = rblk * 16 + 4 * lg;
        f32x4 val, o;
#pragma unroll
        for (int v = 0; v < 4; ++v) {
            val[v] = a[v] + bias;
            o[v] = c1 * val[v] + c2 * xv[v];
            if (nzp) o[v] += sg * nv[v];
        }
        const size_t idx = row + (size_t)tok0 - 1;
        if (tok0 != 0) {
            if (x0o) *reinterpret_cast<f32x4u*>(x0o + idx) = val;
            *reinterpret_cast<f32x4u*>(x + idx) = o;
        } else {                                       // token 0 (the timestep token) has no output: elements 1 .. 3 only
#pragma unroll
            for (int v = 1; v < 4; ++v) {
                if (x0o) x0o[idx + v] = val[v];
                x[idx + v] = o[v];
            }
        }
        if (apk) {
            float* const d = apk + ((size_t)g * S + tok0) * p.t_lda + traj + ch;
#pragma unroll
            for (int v = 0; v < 4; ++v)
                if (tok0 + v != 0) d[(size_t)v * p.t_lda] = o[v];
        }
    };
#pragma unroll
    for (int r = 0; r < NRMAX; ++r)
        if (r < nr) emit(r0 + r, row_own, (tn * NC_OWN + c_loc) * 16 + li, bias_own, acc[r], xt[r], nz[r]);
    if (has_e) emit(e, row_sh, 256 + li, bias_sh, acc_e, xt_e, nz_e);
    // trajectory channels: x0 = cond there (model/posenet.py:94-95); the clip's traj x T elements are dealt over its G workgroups
    for (int i = tn * 256 + tid; i < traj * T; i += G * 256) {
        const int ch = i / T, t = i - ch * T;
        const size_t idx = ((size_t)g * C + ch) * T + t;
        const float val = p.t_cond[idx];
        if (x0o) x0o[idx] = val;
        float o = c1 * val + c2 * x[idx];
        if (nzp) o += sg * nzp[idx];
        x[idx] = o;
        if (apk) apk[((size_t)g * S + t + 1) * p.t_lda + ch] = o;
    }
}

}  // namespace chain

// G = column tiles per clip of every phase = partner workgroups of a clip: 4 (B = 64: tiles 144 x 128 / 256 / 128 / 384) or 8 (B = 32:
// 144 x 64 / 128 / 64 / 192)
template <int G>
__global__ __launch_bounds__(256) void encoder_chain_kernel(ChainParams p) {
    using namespace chain;
    constexpr int BNL = 512 / G, BNF = 1024 / G, BNQ = 1536 / G;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int tid = threadIdx.x;
    const int wave_u = __builtin_amdgcn_readfirstlane(tid >> 6) * 64;
    // block -> (clip, part): the G parts of a clip are consecutive workgroups of ONE XCD (hardware deals block b to XCD b % 8 in
    // block order) -- co-resident, one L2 (gemm_f32.hip EPI_BIAS_RES_LN's map)
    const int x = blockIdx.x % kNumXCD, j = blockIdx.x / kNumXCD;
    const int g = (j / G) * kNumXCD + x, tn = j % G;
    if (g >= p.tiles_m) return;
    const unsigned xcc1 = 1u + (unsigned)__builtin_amdgcn_s_getreg((3 << 11) | 20);
    const unsigned ep = p.epoch + 64u * *p.xln_pass;      // the pass counter was advanced by an EARLIER kernel of the stream
    if (tid == 0) p.xln_xcc[g * 8 + tn] = xcc1;
    unsigned long long* const flags = p.flags + (size_t)g * 9 * 5 * 8;

    PhaseArgs a{};
    a.m0 = g * BM; a.err = p.xln_err; a.xcc1 = xcc1; a.tn = tn; a.tiles_n = G;
    a.row_stats = p.xln_stats + ((size_t)g * G * BM) * 4;
    a.eps = p.ln_eps; a.ln_dim = p.D;

    // ---- A: y = norm1(h + ctx . Wo^T + bo) ---------------------------------------------------------------------------------------
    a.A = p.ctx; a.lda = p.D; a.W = p.out_w; a.ldw = p.D; a.C = p.y; a.ldc = p.D; a.K = p.D; a.bias = p.out_b;
    a.R = p.h; a.ldr = p.D; a.gamma = p.n1_w; a.beta = p.n1_b; a.n0 = tn * BNL; a.tag28 = ep & 0x0fffffffu; a.fault = p.fault & 1;
    gemm_phase<BNL, EPI_BIAS_RES_LN, false, false, G == 8>(a, smem, tid, [&]() { prefetch_w<BNF>(p.l1_w, p.D, tn * BNF, smem, tid, wave_u); });
    group_sync(flags, tn, G, ep, xcc1, p.xln_err, tid);

    // ---- B: ff = gelu(y . W1^T + b1) ---------------------------------------------------------------------------------------------
    a.A = p.y; a.lda = p.D; a.W = p.l1_w; a.ldw = p.D; a.C = p.ff; a.ldc = p.F; a.K = p.D; a.bias = p.l1_b; a.n0 = tn * BNF;
    gemm_phase<BNF, EPI_BIAS_GELU, true, true, G == 8>(a, smem, tid, [&]() { prefetch_w<BNL>(p.l2_w, p.F, tn * BNL, smem, tid, wave_u); });
    group_sync(flags + 8, tn, G, ep, xcc1, p.xln_err, tid);

    // ---- C: h = norm2(y + ff . W2^T + b2) ----------------------------------------------------------------------------------------
    a.A = p.ff; a.lda = p.F; a.W = p.l2_w; a.ldw = p.F; a.C = p.h; a.ldc = p.D; a.K = p.F; a.bias = p.l2_b;
    a.R = p.y; a.ldr = p.D; a.gamma = p.n2_w; a.beta = p.n2_b; a.n0 = tn * BNL; a.tag28 = (ep + 1u) & 0x0fffffffu; a.fault = (p.fault >> 1) & 1;
    gemm_phase<BNL, EPI_BIAS_RES_LN, true, true, G == 8>(a, smem, tid, [&]() { if (p.qkv) prefetch_w<BNQ>(p.in_w, p.D, tn * BNQ, smem, tid, wave_u); });
    if (p.qkv == nullptr) return;      // last layer: the output head follows as its own launch (uniform over the launch)
    group_sync(flags + 16, tn, G, ep, xcc1, p.xln_err, tid);

    // ---- D: qkv = h . Win^T + bin of the next layer, q pre-scaled ------------------------------------------------------------------
    a.A = p.h; a.lda = p.D; a.W = p.in_w; a.ldw = p.D; a.C = p.qkv; a.ldc = 3 * p.D; a.K = p.D; a.bias = p.in_b; a.n0 = tn * BNQ;
    a.qcols = p.D; a.qscale = p.qscale;
    gemm_phase<BNQ, EPI_QKV, true, true, G == 8>(a, smem, tid, []() {});
}

// The whole encoder: for every layer  [qkv of the clip complete] attention  [ctx complete]  A  B  C  [D = the next layer's QKV].
// Workgroup (clip g, part tn) computes, per layer, head tn (G = 4: a whole (clip, head) item on four waves, two owned query blocks
// per wave) or half tn & 1 of head tn >> 1 (G = 8: the SPLIT shape of attention_f32.hip), then its column tile of every GEMM phase.
// Every operand another workgroup wrote earlier in THIS launch is fetched past the L1 (sc1 LDS-DMA): the same addresses were read one
// layer earlier.  With front != 0 two leading phases turn the packed input into h and layer 0's qkv first (InputProcess + in_proj_0).
template <int G>
__global__ __launch_bounds__(256) void encoder_stack_kernel(StackParams p) {
    using namespace chain;
    constexpr int BNL = 512 / G, BNF = 1024 / G, BNQ = 1536 / G;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int tid0 = threadIdx.x;
    const int x = blockIdx.x % kNumXCD, j = blockIdx.x / kNumXCD;
    const int g = (j / G) * kNumXCD + x, tn = j % G;
    if (g >= p.tiles_m) return;
    const unsigned xcc1 = 1u + (unsigned)__builtin_amdgcn_s_getreg((3 << 11) | 20);
    const unsigned ep = p.epoch + 64u * (*p.xln_pass + p.pass_add);
    if (tid0 == 0) p.xln_xcc[g * 8 + tn] = xcc1;

    PhaseArgs a{};
    a.m0 = g * BM; a.err = p.xln_err; a.xcc1 = xcc1; a.tn = tn; a.tiles_n = G;
    a.row_stats = p.xln_stats + ((size_t)g * G * BM) * 4;
    a.eps = p.ln_eps; a.ln_dim = p.D;
    a.qcols = p.D; a.qscale = p.qscale;
    // diagnostics only (p.timeline is null on every product path: one scalar test per seam)
    unsigned long long* const tl = p.timeline ? p.timeline + (size_t)blockIdx.x * kStackTimelineLayers * kStackTimelineStamps : nullptr;
    auto stamp = [&](int layer, int k) __attribute__((always_inline)) {
        if (tl != nullptr && threadIdx.x == 0) tl[layer * kStackTimelineStamps + k] = __builtin_amdgcn_s_memrealtime();
    };
    if (p.front) {
        // ---- E: h = [x_t | cond] . We^T + table rows;  D0: qkv = in_proj_0(h) -- flags of "layer" 8 ------------------------------------
        int tid = tid0;
        asm volatile("" : "+v"(tid));
        const int wave_u = __builtin_amdgcn_readfirstlane(tid >> 6) * 64;
        unsigned long long* const fl = p.flags + ((size_t)g * 9 + 8) * 5 * 8;
        a.A = p.apack; a.lda = p.lda_pack; a.W = p.w_embed; a.ldw = p.ldw_embed; a.C = p.h; a.ldc = p.D; a.K = p.k_embed; a.n0 = tn * BNL;
        a.S = p.S; a.tab = p.tab; a.tab0 = p.tab0; a.ldtab = p.ldtab; a.ldtab0 = p.ldtab0; a.tab_by_row = p.tab_by_row;
        stamp(8, 0);
        gemm_phase<BNL, EPI_EMBED, false, false, G == 8>(a, smem, tid, [&]() { prefetch_w<BNQ>(p.layer[0].in_w, p.D, tn * BNQ, smem, tid, wave_u); });
        stamp(8, 1);
        group_sync(fl, tn, G, ep, xcc1, p.xln_err, tid);
        stamp(8, 2);
        a.A = p.h; a.lda = p.D; a.W = p.layer[0].in_w; a.ldw = p.D; a.C = p.qkv; a.ldc = 3 * p.D; a.K = p.D; a.bias = p.layer[0].in_b;
        a.n0 = tn * BNQ;
        gemm_phase<BNQ, EPI_QKV, true, true, G == 8>(a, smem, tid, []() {});
        stamp(8, 3);
        group_sync(fl + 8, tn, G, ep, xcc1, p.xln_err, tid);
        stamp(8, 4);
    }
#pragma unroll 1
    for (int l = 0; l < p.L; ++l) {
        // Everything a phase derives from the thread index -- operand addresses of four GEMM shapes and the attention item -- is
        // layer-invariant; hoisted out of this loop it is several hundred live registers (measured: 363 spilled).  The index is made
        // opaque per layer so that each phase computes its addresses where it uses them.
        int tid = tid0;
        asm volatile("" : "+v"(tid));
        const int wave_u = __builtin_amdgcn_readfirstlane(tid >> 6) * 64;
        const StackLayerW& w = p.layer[l];
        unsigned long long* const fl = p.flags + ((size_t)g * 9 + l) * 5 * 8;
        stamp(l, 0);
        if (l > 0) group_sync(fl, tn, G, ep, xcc1, p.xln_err, tid);                  // the clip's qkv of this layer is complete
        stamp(l, 1);
        if constexpr (G == 4) attention_item<4, 0, kSc1, 2>(p.qkv, p.ctx, p.n_head, g * p.n_head + tn, 0, AT_NB, smem, tid);
        else attention_item<4, 0, kSc1, 1>(p.qkv, p.ctx, p.n_head, g * p.n_head + (tn >> 1), (tn & 1) ? 5 : 0, (tn & 1) ? 4 : 5, smem, tid);
        stamp(l, 2);
        group_sync(fl + 8, tn, G, ep, xcc1, p.xln_err, tid);                         // ... its ctx
        stamp(l, 3);

        a.A = p.ctx; a.lda = p.D; a.W = w.out_w; a.ldw = p.D; a.C = p.y; a.ldc = p.D; a.K = p.D; a.bias = w.out_b;
        a.R = p.h; a.ldr = p.D; a.gamma = w.n1_w; a.beta = w.n1_b; a.n0 = tn * BNL; a.tag28 = (ep + 4u * l) & 0x0fffffffu;
        a.fault = (l == 0) ? (p.fault & 1) : 0;
        gemm_phase<BNL, EPI_BIAS_RES_LN, false, true, G == 8>(a, smem, tid, [&]() { prefetch_w<BNF>(w.l1_w, p.D, tn * BNF, smem, tid, wave_u); });
        stamp(l, 4);
        group_sync(fl + 16, tn, G, ep, xcc1, p.xln_err, tid);
        stamp(l, 5);

        a.A = p.y; a.lda = p.D; a.W = w.l1_w; a.ldw = p.D; a.C = p.ff; a.ldc = p.F; a.K = p.D; a.bias = w.l1_b; a.n0 = tn * BNF;
        gemm_phase<BNF, EPI_BIAS_GELU, true, true, G == 8>(a, smem, tid, [&]() { prefetch_w<BNL>(w.l2_w, p.F, tn * BNL, smem, tid, wave_u); });
        stamp(l, 6);
        group_sync(fl + 24, tn, G, ep, xcc1, p.xln_err, tid);
        stamp(l, 7);

        const bool more = l + 1 < p.L;
        a.A = p.ff; a.lda = p.F; a.W = w.l2_w; a.ldw = p.F; a.C = p.h; a.ldc = p.D; a.K = p.F; a.bias = w.l2_b;
        a.R = p.y; a.ldr = p.D; a.gamma = w.n2_w; a.beta = w.n2_b; a.n0 = tn * BNL; a.tag28 = (ep + 4u * l + 1u) & 0x0fffffffu;
        a.fault = (l == 0) ? ((p.fault >> 1) & 1) : 0;
        gemm_phase<BNL, EPI_BIAS_RES_LN, true, true, G == 8>(a, smem, tid, [&]() { if (more) prefetch_w<BNQ>(p.layer[l + 1].in_w, p.D, tn * BNQ, smem, tid, wave_u); });
        stamp(l, 8);
        if (!more) {
            if (p.tail) {      // sampling loop: head + DDPM update + the next step's pack close the launch (uniform over the launch)
                group_sync(fl + 32, tn, G, ep, xcc1, p.xln_err, tid);                // the clip's h is complete
                stamp(l, 9);
                head_tail_phase<G>(p, g, tn, smem, tid);
                stamp(l, 10);
                // the passes this call has consumed so far, for the first kernel of the next pass (the counter itself stays untouched:
                // common.h StackParams::pass_add)
                if (p.t_pass_ctr != nullptr && blockIdx.x == 0 && threadIdx.x == 0) p.t_pass_ctr[1] = p.pass_add + 1u;
            }
            break;
        }
        group_sync(fl + 32, tn, G, ep, xcc1, p.xln_err, tid);
        stamp(l, 9);

        a.A = p.h; a.lda = p.D; a.W = p.layer[l + 1].in_w; a.ldw = p.D; a.C = p.qkv; a.ldc = 3 * p.D; a.K = p.D; a.bias = p.layer[l + 1].in_b;
        a.n0 = tn * BNQ;
        gemm_phase<BNQ, EPI_QKV, true, true, G == 8>(a, smem, tid, []() {});
        stamp(l, 10);
    }
}

int launch_encoder_stack(const StackParams& p, hipStream_t s) {
    const int G = encoder_chain_parts(p.M, p.D, p.F);
    ROHM_ARG_CHECK(G != 0 && p.n_head == 4 && p.L >= 1 && p.L <= 8, "encoder_stack: shape (M %d, D %d, F %d, %d heads, %d layers) has no stack form",
                   p.M, p.D, p.F, p.n_head, p.L);
    ROHM_ARG_CHECK(p.h && p.y && p.ff && p.qkv && p.ctx && p.xln_stats && p.xln_err && p.xln_pass && p.xln_xcc && p.flags, "encoder_stack: null operand");
    ROHM_ARG_CHECK(!p.front || (p.apack && p.w_embed && p.tab && p.tab0 && p.S == chain::BM && p.k_embed >= 2 * chain::BK && p.k_embed % chain::BK == 0 &&
                                p.lda_pack % 4 == 0 && p.ldw_embed % 4 == 0 && p.ldtab % 4 == 0 && p.ldtab0 % 4 == 0),
                   "encoder_stack: bad operands of the leading embed phase");
    StackParams q = p;
    q.tiles_m = p.M / chain::BM;
    const int groups8 = (q.tiles_m + kNumXCD - 1) / kNumXCD * kNumXCD;
    ROHM_ARG_CHECK(!p.tail || (p.D == 512 && p.t_out_w && p.t_out_b && p.t_x && p.t_cond && p.t_C - p.t_traj == 272 &&
                               p.t_T == chain::BM - 1 && p.t_traj > 0 && (!p.t_apack || p.t_lda >= p.t_C)),
                   "encoder_stack: the closing head / update / pack phase needs whole 144-token clips, d_model 512 and 272 predicted channels");
    constexpr int kFloats = AT_LDS_FLOATS > chain::kLdsFloats ? AT_LDS_FLOATS : chain::kLdsFloats;
    const size_t lds = (size_t)kFloats * sizeof(float);
    static bool attr_set[64][2] = {};
    int dev = 0;
    ROHM_HIP_CHECK(hipGetDevice(&dev));
    if (dev < 64 && !attr_set[dev][G == 8]) {
        if (G == 4) ROHM_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&encoder_stack_kernel<4>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        else ROHM_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&encoder_stack_kernel<8>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        attr_set[dev][G == 8] = true;
    }
    const double MM = (double)p.M, D = p.D, F = p.F, L = p.L;
    // algorithmic work of the launch: L x (attention 4 S^2 d_h per (clip, head) + out-proj + FF1 + FF2) + (L - 1) QKV projections
    const double flops = L * (4.0 * 144.0 * 128.0 * MM * p.n_head + 2.0 * MM * (D * D + 2.0 * D * F)) + (L - 1.0 + (p.front ? 1.0 : 0.0)) * 2.0 * MM * 3.0 * D * D +
                         (p.front ? 2.0 * MM * D * p.k_embed : 0.0);
    const double bytes = 4.0 * (L * (MM * (3.0 * D + D + 4.0 * D + 2.0 * F + D) + D * D + 2.0 * D * F) + (L - 1.0) * (MM * 3.0 * D + 3.0 * D * D));
    const double tail_flops = p.tail ? 2.0 * MM * D * (p.t_C - p.t_traj) : 0.0;
    prof::Scope ps(p.tail ? "gemm_stack_tail" : "gemm_stack", flops + tail_flops, bytes, s);
    if (G == 4) hipLaunchKernelGGL(encoder_stack_kernel<4>, dim3(groups8 * 4), dim3(256), lds, s, q);
    else hipLaunchKernelGGL(encoder_stack_kernel<8>, dim3(groups8 * 8), dim3(256), lds, s, q);
    ROHM_LAUNCH_CHECK();
    return ROHM_OK;
}

int encoder_chain_parts(int M, int D, int F) {      // 0: this shape has no chain form
    if (D != 512 || F != 1024 || M <= 0 || M % chain::BM != 0) return 0;
    const int tm = M / chain::BM;
    // 4 parts per clip (144 x 128 / 256 / 128 / 384 tiles) from 48 clips on, 8 (144 x 64 / 128 / 64 / 192) below: 48 .. 64 clips fit ONE
    // round of 4-part workgroups (192 .. 256 of them), whose wider tiles beat two rounds of 8-part ones even with CUs left idle
    // (measured, profiles/r6_f_*: B = 56 as 448 8-part workgroups 3.12 ms per step, as 224 4-part ones the 2.83 ms of B = 64)
    return tm >= 48 ? 4 : 8;
}

// Does the chain / stack form pay at this batch size?  Its workgroups are persistent, one per CU, so a launch costs whole ROUNDS of 256
// workgroups: B = 40 (320 8-part workgroups) or B = 72 (288 4-part ones) pay two rounds for little more than one round of work, and
// one launch per GEMM -- which picks a tile width per GEMM -- is 7-27 % faster there (measured at B = 40 / 48 / 56 / 72 / 96 / 128,
// profiles/r6_f_tail_ab_and_batch_sweep.json).  So: a single round of 4-part workgroups (48 .. 64 clips), a single round of
// 8-part ones that fills the chip (25 .. 32 clips; the callers also ask for >= 32), or several rounds that are >= 93 % full
// (profiles/r6_g_stack_gate_sweep.json).
bool encoder_chain_pays(int B) {
    if (B <= 0) return false;
    if (B < 48) return B > 24 && B <= 32;
    if (B <= 64) return true;
    const int wgs = (B + kNumXCD - 1) / kNumXCD * kNumXCD * 4, rounds = (wgs + 255) / 256;
    return 4 * B * 100 >= 93 * rounds * 256;      // B = 120 (two rounds, 94 % full): the stack is 2.4 % faster; B = 112 stays per GEMM
}

// [row tile][layer <= 8, + 1 for the leading embed / QKV phases][meeting point <= 5][part <= 8] (the per-layer chain uses the first 3 x 8
// words of a row tile's block)
size_t encoder_chain_flag_bytes(int M) { return (size_t)((M + chain::BM - 1) / chain::BM) * 9 * 5 * 8 * sizeof(unsigned long long); }

int launch_encoder_chain(const ChainParams& p, hipStream_t s) {
    const int G = encoder_chain_parts(p.M, p.D, p.F);
    ROHM_ARG_CHECK(G != 0, "encoder_chain: shape (M %d, D %d, F %d) has no chain form", p.M, p.D, p.F);
    ROHM_ARG_CHECK(p.ctx && p.h && p.y && p.ff && p.out_w && p.out_b && p.n1_w && p.n1_b && p.l1_w && p.l1_b && p.l2_w && p.l2_b &&
                       p.n2_w && p.n2_b && p.xln_stats && p.xln_err && p.xln_pass && p.xln_xcc && p.flags && (!p.qkv || (p.in_w && p.in_b)),
                   "encoder_chain: null operand");
    ChainParams q = p;
    q.tiles_m = p.M / chain::BM;
    const int groups8 = (q.tiles_m + kNumXCD - 1) / kNumXCD * kNumXCD;
    const size_t lds = (size_t)chain::kLdsFloats * sizeof(float);
    static bool attr_set[64][2] = {};
    int dev = 0;
    ROHM_HIP_CHECK(hipGetDevice(&dev));
    if (dev < 64 && !attr_set[dev][G == 8]) {
        if (G == 4) ROHM_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&encoder_chain_kernel<4>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        else ROHM_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&encoder_chain_kernel<8>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        attr_set[dev][G == 8] = true;
    }
    const double MM = (double)p.M, D = p.D, F = p.F;
    const double flops = 2.0 * MM * (D * D + 2.0 * D * F + (p.qkv ? 3.0 * D * D : 0.0));
    const double bytes = 4.0 * (MM * (4.0 * D + 2.0 * F + D + (p.qkv ? 3.0 * D : 0.0)) + D * D + 2.0 * D * F + (p.qkv ? 3.0 * D * D : 0.0));
    prof::Scope ps(p.qkv ? "gemm_chain" : "gemm_chain_last", flops, bytes, s);
    if (G == 4) hipLaunchKernelGGL(encoder_chain_kernel<4>, dim3(groups8 * 4), dim3(256), lds, s, q);
    else hipLaunchKernelGGL(encoder_chain_kernel<8>, dim3(groups8 * 8), dim3(256), lds, s, q);
    ROHM_LAUNCH_CHECK();
    return ROHM_OK;
}

}  // namespace rohm
