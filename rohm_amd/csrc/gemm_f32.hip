// fp32 MFMA GEMM for gfx950:  C[M,N] = epi(A[M,K] . W[N,K]^T)
//
// Every Linear of the PoseNet path (model/posenet.py:63-69, model/heads.py:154,169) and, with a tap-gathered
// A operand, every Conv1d / ConvTranspose1d of TrajNet (model/heads.py:72-106) runs on this kernel.
// Numerics: v_mfma_f32_32x32x2_f32 / v_mfma_f32_16x16x4_f32 are exact-fp32 fma chains (guide §3), i.e. the
// same rounding class as the CPU reference's fp32 GEMM; only the summation order differs.
//
// Design (MI355X-first, not a warp-shaped port):
//   * workgroup tile 144 x BN (BN = 64 / 128 / 256 / 384), 4 waves = one per SIMD, ONE workgroup per CU.
//     144 = one PoseNet clip (143 frames + timestep token), so M = B*144 has no remainder and at the headline
//     batch B = 64 the widest tile that still gives all 256 CUs a tile is picked per GEMM: N = 512 -> 144x128,
//     N = 1024 -> 144x256, N = 1536 -> 144x384, each exactly 256 tiles.
//   * wave w owns columns [w*BN/4, (w+1)*BN/4).  Rows 0..127 of the tile are four 32-row blocks multiplied
//     with v_mfma_f32_32x32x2_f32 (measured 154.7 TFLOP/s as a bare stream with one wave per SIMD), rows
//     128..143 one 16-row block on v_mfma_f32_16x16x4_f32 (138.8 TFLOP/s bare; it cannot be issued back to back
//     at its nominal 32 cycles).  So 8/9 of the flops run on the instruction that reaches peak.  (BN = 64 keeps
//     the all-16x16 form: a wave's 16 columns are narrower than a 32x32 block.)
//   * both operands are K-contiguous, so a lane's fragments for several consecutive k-steps are ONE
//     ds_read_b128: 32x32x2: lane (i = l&31, g = l>>5) holds X[i][8*s + 4g + j], j = 0..3, MFMA j contracts
//     k = {j, 4 + j}; 16x16x4: lane (i = l&15, g = l>>4) holds X[i][16*s + 4g + j], MFMA j contracts
//     k = {4g + j}: fixed permutations of k, identical for both operands.
//   * operands are SWAPPED (weights on the MFMA "A" side) so a lane ends with 4 consecutive output columns of one
//     row -> 16-byte epilogue stores (the 4-byte form cost 10 us per tile).
//   * staging: LDS-DMA (global_load_lds_dwordx4), K chunks of 32, two LDS buffers; the XOR swizzle that makes
//     every ds_read_b128 conflict-free is applied to the per-lane SOURCE address (the DMA writes LDS linearly).
//     No predicated loads: out-of-range rows are clamped (they only feed rows / columns never stored).
//   * schedule per chunk: [LDS reads of half 1 | MFMAs of half 0] vmcnt(0)+barrier [DMA of chunk k+2 + LDS reads
//     of chunk k+1's half 0 | MFMAs of half 1]: HBM/L2 latency has a whole chunk to land, LDS latency hides
//     behind MFMAs.  The loop body is one basic block (the DMA of the last two iterations is redirected, not
//     predicated) and sched_group_barrier spreads every non-MFMA instruction between MFMAs: with one wave per
//     SIMD a clustered non-MFMA issue slot is an MFMA-pipe bubble.
//   * XCD-aware tile order: each XCD gets a contiguous run of tiles sharing A panels.
//   * epilogue: the operands a unit needs from memory (bias, residual, embed-table row) are read for ALL units before the first store
//     -- GemmParams members cannot be restrict, so behind a store to C the compiler must re-read them, one vmcnt(0) round trip per
//     unit -- and, for tiles up to 256 wide, already at the top of the peeled last K chunk (they land under its MFMAs); the 384-wide
//     tile takes its bias row through LDS.  The FULL instantiation has no conditional operand loads (LayerNorm folding lives in the
//     general one).
#include <stdlib.h>
#include <type_traits>
#include <string.h>
#include "common.h"
#include "gemm_sched.h"

namespace rohm {

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int BM = 144;
constexpr int BK = 32;
constexpr int NRB = BM / 16;        // 9 row blocks of 16 (BN = 64 path)
constexpr int kSkWorkgroups = 256;  // stream-K launches: one workgroup per CU (32 per XCD)
constexpr int A_UNITS = BM * 8;     // 16-byte units per A chunk (1152)
constexpr int A_ITERS = (A_UNITS + 255) / 256;  // 5 (the last pass is half populated)

// (mu, rstd) of one row from its partial (sum, sum of squares) pairs -- biased variance, eps inside the root, like
// nn.LayerNorm.  The pairs of a row are contiguous (parts x 2 floats, parts a multiple of 2): independent 16-byte
// loads, summed in index order.
__device__ __forceinline__ void row_mu_rstd(const float* __restrict__ stats, int parts, int row, int dim, float eps,
                                            float& mu, float& rstd) {
    const f32x4* sp = reinterpret_cast<const f32x4*>(stats + (size_t)row * parts * 2);
    f32x4 v[4];                      // parts <= 8 (launch_gemm checks)
#pragma unroll
    for (int i = 0; i < 4; ++i) v[i] = (2 * i < parts) ? sp[i] : f32x4{0.f, 0.f, 0.f, 0.f};
    float s = 0.f, q = 0.f;
#pragma unroll
    for (int i = 0; i < 4; ++i) { s += v[i][0]; q += v[i][1]; s += v[i][2]; q += v[i][3]; }
    const float inv = 1.0f / (float)dim;
    mu = s * inv;
    const float var = fmaxf(q * inv - mu * mu, 0.f);
    rstd = 1.0f / sqrtf(var + eps);
}

#ifndef ROHM_GEMM_MIXED
#define ROHM_GEMM_MIXED 1
#endif

template <int BN, int EPI, int VAR = 0, bool FULL = false, bool CONV = false>
__global__ __launch_bounds__(256) void gemm_f32_kernel(GemmParams p) {
    constexpr int WN = BN / 4;             // columns per wave
    constexpr bool M32 = (WN >= 64) && (ROHM_GEMM_MIXED != 0);   // mixed 32x32x2 + 16x16x4 path
    constexpr int NCB = WN / 16;           // 16-wide column blocks per wave
    constexpr int NCB32 = WN / 32;         // 32-wide column blocks per wave (M32)
    constexpr int B_UNITS = BN * 8;
    constexpr int B_ITERS = B_UNITS / 256; // 2, 4, 8 or 12
    constexpr int PIECES = A_ITERS + B_ITERS;   // LDS-DMA instructions per wave per chunk
    // Operand order: SWAP puts the weight fragment on the MFMA "A" side, so a lane ends up with four CONSECUTIVE
    // output columns of one row (one 16-byte store); the transposed output head keeps the natural order because
    // its stores are contiguous along the token axis instead.
    constexpr bool SWAP = (EPI != EPI_OUT_T);

    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* As = smem;                       // [2][BM*BK]
    float* Bs = smem + 2 * BM * BK;         // [2][BN*BK]
    float* lds_dummy = smem + 2 * (BM + BN) * BK;     // 2 KiB landing zone (inside the padded LDS request)

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int li = lane & 15, lg = lane >> 4;       // 16x16x4 fragment coordinates
    const int li32 = lane & 31, lg32 = lane >> 5;   // 32x32x2 fragment coordinates
    const int wave_u = __builtin_amdgcn_readfirstlane(wave) * 64;   // provably wave-uniform LDS base

    unsigned long long ts_wall[5] = {0, 0, 0, 0, 0}, ts_cyc[2] = {0, 0};   // VAR 7 only (timeline diagnostics)
    if constexpr (VAR == 7) ts_wall[0] = wall_clock64();
    const int tiles_n = (p.N + BN - 1) / BN;
    const int tiles_m = (p.M + BM - 1) / BM;
    int ksplit = (EPI == EPI_BIAS && p.ksplit > 1) ? p.ksplit : 1;
    int split = (ksplit > 1) ? (int)(blockIdx.x % ksplit) : 0;
    int m0, n0;
    // ---- stream-K (GemmParams::sk_*; the output head at B = 64: 288 tiles of 16 chunks on 256 CUs) -------------------------------
    // The (tile, chunk) units of the launch are dealt out evenly: XCD x owns tiles [x T8, (x + 1) T8) and its j-th workgroup
    // (block 8 j + x) the units [j u, (j + 1) u) of them -- a workgroup's range is the TAIL of one tile, whole tiles, and the HEAD of
    // another (or, when u < chunks per tile, a piece from the middle of one: B = 32, 144 tiles, u = 9); a workgroup produces at most
    // one partial tile.  Segments are processed LAST first: a segment that does not reach its tile's last chunk leaves its raw
    // accumulators in this workgroup's slot of `sk_part` (role 1: producer) early; the segment that does comes last, and its
    // workgroup (role 2: owner) adds the slots of the blocks 8, 16, .. below it that hold the tile's earlier chunks -- same XCD,
    // dispatched EARLIER, written long before -- and runs the epilogue.  Dependencies point to lower block indices only: no
    // workgroup waits for one that may not have started.
    constexpr bool SK = (EPI == EPI_OUT_T) && !FULL && !CONV && BN == 64 && VAR == 0;
    int sk_lo = 0, sk_hi = 0, sk_role = -1;      // role of the current segment: 0 whole tile, 1 producer, 2 owner; -1 before the first
    // Tag of this launch's exchange slots (LayerNorm statistics / stream-K flags): the host's part (a per-handle salt + the launch's
    // index within a network pass) plus 64 x the workspace's pass counter, a DEVICE word that the first kernel of every pass
    // increments -- so a launch recorded into a hipGraph carries a fresh tag on every replay (GemmParams::xln_pass).
    unsigned xln_ep = 0;
    if constexpr (EPI == EPI_BIAS_RES_LN || SK) {
        xln_ep = p.xln_epoch;
        if (p.xln_pass) xln_ep += 64u * *p.xln_pass;      // written by an EARLIER kernel of the stream: a plain (scalar) load
    }
    // chunks of K this launch's tiles contract over, and where they start: all of K -- except for the 1x1 residual columns of a
    // fused [conv5 | residual] launch (trajnet.hip), whose weights are zero outside the centre tap: those tiles walk the centre
    // tap's chunks only, with their own split count
    int nk_all = p.K / BK, k_first = 0;
    if (CONV && EPI == EPI_BIAS && p.res_col0 > 0) {
        const int tiles_nc = p.res_col0 / BN, tiles_nr = tiles_n - tiles_nc;
        const int wg_c = tiles_m * tiles_nc * ksplit;
        if ((int)blockIdx.x < wg_c) {
            const int tile = xcd_remap(blockIdx.x / ksplit, tiles_m * tiles_nc);
            m0 = (tile / tiles_nc) * BM;
            n0 = (tile % tiles_nc) * BN;
        } else {
            const int rb = (int)blockIdx.x - wg_c;
            ksplit = p.res_ksplit > 1 ? p.res_ksplit : 1;
            split = rb % ksplit;
            const int tile = xcd_remap(rb / ksplit, tiles_m * tiles_nr);
            m0 = (tile / tiles_nr) * BM;
            n0 = (tiles_nc + tile % tiles_nr) * BN;
            nk_all = p.res_nk;
            k_first = p.res_k0;
        }
    } else if constexpr (EPI == EPI_BIAS_RES_LN) {
        // The tiles_n column tiles of a row tile exchange their row statistics while they run, so they must sit on ONE XCD (one
        // L2: the exchange never leaves it) and be handed out back to back (co-resident: nobody waits for a tile that cannot
        // start).  Hardware places block b on XCD b % 8 in block order, so the j-th block of XCD x takes column tile
        // j % tiles_n of row tile (j / tiles_n) * 8 + x; the launch rounds tiles_m up to a multiple of 8, the surplus leaves.
        const int x = blockIdx.x % kNumXCD, j = blockIdx.x / kNumXCD;
        const int g = (j / tiles_n) * kNumXCD + x;
        if (g >= tiles_m) return;
        m0 = g * BM;
        n0 = (j % tiles_n) * BN;
    } else if (SK && p.sk_units > 0) {
        m0 = n0 = 0;                       // set per segment below
        sk_lo = (int)(blockIdx.x / kNumXCD) * p.sk_units;
        sk_hi = sk_lo + p.sk_units;
    } else {
        const int tile = xcd_remap(ksplit > 1 ? blockIdx.x / ksplit : blockIdx.x, tiles_m * tiles_n);
        m0 = (tile / tiles_n) * BM;
        n0 = (tile % tiles_n) * BN;
    }

    // Everything below is the life of ONE (tile, K range).  A stream-K workgroup walks its segments through it, last segment first;
    // every other launch passes once (the loop condition is a compile-time `false` for them).
    do {
    if constexpr (SK) {
        if (p.sk_units > 0) {
            if (sk_role >= 0) {            // not the first segment: the previous one's LDS reads and global stores are done
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                __syncthreads();
            }
            const int nkt = p.K / BK;                                // chunks per tile
            const int t = (sk_hi - 1) / nkt;                         // this XCD's tile holding the last unit still to do
            const int c_lo = sk_lo > t * nkt ? sk_lo - t * nkt : 0;  // ... and the chunks [c_lo, c_hi) of it that are this workgroup's
            const int c_hi = sk_hi - t * nkt;
            sk_role = c_hi < nkt ? 1 : (c_lo > 0 ? 2 : 0);
            const int gt = (int)(blockIdx.x % kNumXCD) * p.sk_tiles8 + t;
            m0 = (gt % tiles_m) * BM;      // the row tiles of one column tile are neighbours: its W rows are read once per L2
            n0 = (gt / tiles_m) * BN;
            nk_all = c_hi - c_lo;
            k_first = c_lo * BK;
            sk_hi = t * nkt + c_lo;
        }
    }

    // ---- global -> LDS staging by LDS-DMA ---------------------------------------------------------------------
    const float* a_src[A_ITERS];
    // conv gather, per unit: the zero-page address of its 16-B slot, the distance from there to its row at tap offset 0,
    // and tq * stride (for the in-clip test)
    uintptr_t a_cdiff[A_ITERS], a_zero[A_ITERS];      // a_cdiff = (row address at tap offset 0) - (zero-page address)
    int a_tqs[A_ITERS];
#pragma unroll
    for (int i = 0; i < A_ITERS; ++i) {
        const int u = tid + i * 256;
        const int row = (u < A_UNITS) ? (u >> 3) : 0;
        const int slot = (u & 7) ^ ((row >> 1) & 7);
        const int grow = (m0 + row < p.M) ? m0 + row : p.M - 1;
        a_src[i] = p.A + (size_t)grow * p.lda + slot * 4;
        if constexpr (CONV) {
            const int b = grow / p.conv_tq;
            a_tqs[i] = (grow - b * p.conv_tq) * p.conv_stride;
            a_zero[i] = (uintptr_t)(p.zero_page + slot * 4);
            a_cdiff[i] = (uintptr_t)(p.A + ((size_t)b * p.conv_tin + a_tqs[i]) * p.lda + slot * 4) - a_zero[i];
        }
    }
    // conv gather: source pointer of unit i for the K chunk starting at k0 (a chunk never straddles taps because
    // cin_pad is a multiple of 32); taps that fall outside the clip read a page of zeros.  Everything that depends on the
    // chunk only (tap index, its row offset, the channel base) is wave-uniform scalar arithmetic (no table: an indexed
    // kernel-argument load costs an s_waitcnt lgkmcnt(0) inside the loop), a unit adds that scalar byte offset to its
    // precomputed distance, and the in-clip / zero-page choice is a masked blend of the two addresses: a branch here
    // splits the loop body that sched_group_barrier interleaves (both happened: the gather's loop ran at ~1.5 us per
    // chunk against 1.05 us of MFMA).
    const int conv_cp = p.conv_cin_pad;
    auto conv_src = [&](int i, int k0) -> const float* {
        const int j = (k0 >= conv_cp) + (k0 >= 2 * conv_cp) + (k0 >= 3 * conv_cp) + (k0 >= 4 * conv_cp);
        const int off = p.conv_off0 + j * p.conv_dstep;                       // tap offsets are an arithmetic progression
        const long long d = ((long long)off * p.lda + (k0 - j * conv_cp)) * (long long)sizeof(float);
        // blend through an opaque lane mask: written as a select, hipcc turns the choice back into an exec-masked branch
        unsigned m32 = ((unsigned)(a_tqs[i] + off) < (unsigned)p.conv_tin) ? 0xffffffffu : 0u;
        asm("" : "+v"(m32));
        const uintptr_t m = ((uintptr_t)m32 << 32) | m32;
        return reinterpret_cast<const float*>(a_zero[i] + ((a_cdiff[i] + (uintptr_t)d) & m));
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
    auto dma_piece = [&](int piece, int buf, int k0) {
        if (piece < A_ITERS) {
            // the last A pass is half populated: waves 2-3 fetch a (valid) dummy row into the landing zone instead
            // of being predicated off -- a branch here would split the block the scheduler interleaves
            float* dst = As + buf * (BM * BK) + (piece * 256 + wave_u) * 4;
            if (piece == A_ITERS - 1 && A_UNITS % 256 != 0)
                dst = (wave_u < A_UNITS - (A_ITERS - 1) * 256) ? dst : lds_dummy + (wave_u & 64) * 4;
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(CONV ? conv_src(piece, k0) : a_src[piece] + k0),
                                             (__attribute__((address_space(3))) void*)dst, 16, 0, 0);
        } else {
            const int i = piece - A_ITERS;
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(b_src[i] + k0),
                                             (__attribute__((address_space(3))) void*)(Bs + buf * (BN * BK) + (i * 256 + wave_u) * 4),
                                             16, 0, 0);
        }
    };
    auto dma = [&](int buf, int k0) {
#pragma unroll
        for (int i = 0; i < PIECES; ++i) dma_piece(i, buf, k0);
    };

    // ---- accumulators and fragments --------------------------------------------------------------------------
    // M32:  acc32[rb*NCB32 + cb] = 32x32 blocks of rows rb*32.. (rb < 4), acc16[c] = 16x16 blocks of rows 128..143
    // !M32: acc16[r*NCB + c]     = 16x16 blocks of rows r*16..
    constexpr int N32 = M32 ? 4 * NCB32 : 1;
    constexpr int N16 = M32 ? NCB : NRB * NCB;
    f32x16 acc32[N32];
    f32x4 acc16[N16];
#pragma unroll
    for (int i = 0; i < N32; ++i)
#pragma unroll
        for (int q = 0; q < 16; ++q) acc32[i][q] = 0.f;
#pragma unroll
    for (int i = 0; i < N16; ++i) acc16[i] = f32x4{0.f, 0.f, 0.f, 0.f};

    // one k16 half of a chunk: M32: a = A32[s8*4 + rb] (8) + A16 (1), b = B32[s8*NCB32 + cb] + B16[c]
    //                         !M32: a = A16[r] (9), b = B16[c]
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
    auto mma_half = [&](const Frag& f) {
        if constexpr (M32) {
#pragma unroll
            for (int s8 = 0; s8 < 2; ++s8)
#pragma unroll
                for (int j = 0; j < 4; ++j)
#pragma unroll
                    for (int rb = 0; rb < 4; ++rb)
#pragma unroll
                        for (int cb = 0; cb < NCB32; ++cb) {
                            const float wv = f.b[s8 * NCB32 + cb][j], av = f.a[s8 * 4 + rb][j];
                            if constexpr (SWAP)
                                acc32[rb * NCB32 + cb] = __builtin_amdgcn_mfma_f32_32x32x2f32(wv, av, acc32[rb * NCB32 + cb], 0, 0, 0);
                            else
                                acc32[rb * NCB32 + cb] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, wv, acc32[rb * NCB32 + cb], 0, 0, 0);
                        }
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int c = 0; c < NCB; ++c) {
                    const float wv = f.b[2 * NCB32 + c][j], av = f.a[8][j];
                    if constexpr (SWAP) acc16[c] = __builtin_amdgcn_mfma_f32_16x16x4f32(wv, av, acc16[c], 0, 0, 0);
                    else acc16[c] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, wv, acc16[c], 0, 0, 0);
                }
        } else {
#pragma unroll
            for (int r = 0; r < NRB; ++r)
#pragma unroll
                for (int c = 0; c < NCB; ++c)
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        if constexpr (SWAP)
                            acc16[r * NCB + c] = __builtin_amdgcn_mfma_f32_16x16x4f32(f.b[c][j], f.a[r][j], acc16[r * NCB + c], 0, 0, 0);
                        else
                            acc16[r * NCB + c] = __builtin_amdgcn_mfma_f32_16x16x4f32(f.a[r][j], f.b[c][j], acc16[r * NCB + c], 0, 0, 0);
                    }
        }
    };

    // ---- main loop -------------------------------------------------------------------------------------------
    // VAR (diagnostic builds, ROHM_GEMM_VARIANT): 0 = shipped; 5 = MFMA only + no epilogue; 6 = no epilogue;
    // 7 = shipped schedule + per-workgroup phase timestamps written to p.R (scripts/gemm_timeline.py).
    // split-K: this workgroup owns chunks [kc0, kc0 + nk) of the K / BK chunks (the first K/BK % ksplit splits
    // take one more)
    const int nk = nk_all / ksplit + (split < nk_all % ksplit ? 1 : 0);
    const int kc0 = split * (nk_all / ksplit) + (split < nk_all % ksplit ? split : nk_all % ksplit);
    const int kbase = k_first + kc0 * BK;
    // 384-wide tile (QKV): no registers to keep 18 column groups of bias during the loop, and a global load behind the loop is a
    // round trip in front of 54 stores per lane -- the tile's bias row goes through LDS instead (an LDS read is not ordered
    // against the stores to C): two LDS-DMA pieces of wave 0 in front of chunk 0's (older, so the prologue's counted wait covers
    // them; through registers the ds_write drew a vmcnt(0) that drained chunk 1 as well)
    constexpr bool COL_LDS = FULL && !CONV && BN >= 384 && EPI != EPI_EMBED && EPI != EPI_OUT_T;
    float* const bias_s = lds_dummy + 512;          // the row-statistics zone of the general instantiation (unused here), 4.5 KiB
    if constexpr (COL_LDS) {
        if (wave == 0) {
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const int unit = h * 64 + lane;                               // 16-byte unit of the tile's bias row (BN / 4 of them)
                const float* src = p.bias + n0 + (unit < BN / 4 ? unit : BN / 4 - 1) * 4;      // the tail re-reads the last one
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                                 (__attribute__((address_space(3))) void*)(bias_s + h * 256), 16, 0, 0);
            }
        }
    }
    dma(0, kbase);
    if (nk > 1) dma(1, kbase + BK);
    // ---- LayerNorm folding (common.h): row statistics, fetched under the prologue's DMA latency ----------------------
    // per-lane row slots -- !M32: slot r = row m0 + 16 r + li; M32: slots 0..3 = rows m0 + 32 rb + li32, slot 4 = row
    // m0 + 128 + li.
    constexpr int NSLOT = M32 ? 5 : NRB;
    // LayerNorm folding (opt-in, measured slower than the LayerNorm kernel) lives in the general instantiation only: the FULL one
    // carries no run-time choices in its epilogue (every conditional operand load is a value merge the compiler resolves with a
    // vmcnt(0) right behind the load)
    constexpr bool LN_CONSUMER = (EPI == EPI_QKV || EPI == EPI_BIAS_GELU) && !FULL;
    constexpr bool LN_PRODUCER = (EPI == EPI_BIAS_RES) && !M32 && !FULL;
    float amu[NSLOT], ars[NSLOT], rmu[NSLOT], rrs[NSLOT], osum[NSLOT], osq[NSLOT];
    auto slot_row = [&](int sl) {
        int m;
        if constexpr (M32) m = (sl < 4) ? m0 + sl * 32 + li32 : m0 + 128 + li;
        else m = m0 + sl * 16 + li;
        return m < p.M ? m : p.M - 1;
    };
    if constexpr (LN_CONSUMER) {
        if (p.ln_stats) {
#pragma unroll
            for (int sl = 0; sl < NSLOT; ++sl)
                row_mu_rstd(p.ln_stats, p.ln_parts, slot_row(sl), p.ln_dim, p.ln_eps, amu[sl], ars[sl]);
        }
    }
    if constexpr (EPI == EPI_BIAS_RES && !FULL) {
        if (p.r_stats) {
#pragma unroll
            for (int sl = 0; sl < NSLOT; ++sl)
                row_mu_rstd(p.r_stats, p.r_parts, slot_row(sl), p.ln_dim, p.ln_eps, rmu[sl], rrs[sl]);
        }
#pragma unroll
        for (int sl = 0; sl < NSLOT; ++sl) { osum[sl] = 0.f; osq[sl] = 0.f; }
    }
    // ---- epilogue operands (defined here: the last chunk's iteration already requests them) -------------------------------
    const int nw = n0 + wave * WN;
    const bool vec_ok = FULL || ((p.N % 4 == 0) && (p.ldc % 4 == 0) && (((uintptr_t)p.C & 15) == 0));
    // An output unit (4 consecutive columns of one row) needs, besides its accumulators, values from memory: the bias, the
    // fold vectors, the residual.  They are read in `load_ops`, used in `finish`.  The struct members are plain pointers (no
    // restrict), so hipcc must assume that a store to C changes them and keeps every load behind the store before it: emitted
    // unit by unit, each unit waits vmcnt(0) for its own loads AND for the acknowledgement of the store before -- one
    // dependent round trip to memory per unit (18 to 54 per lane).  The hot instantiations (PRE) therefore read the operands of
    // ALL units first (identical addresses -- the bias of a column group -- collapse into one load) and then only compute
    // and store.
    struct ColOps { f32x4 bias, c4, g4, b4; };      // what depends on the column group only
    // (not where it would spill: the 384-wide tile keeps 216 accumulators per lane; its QKV form fits, 475 VGPRs)
    constexpr bool RES = (EPI == EPI_BIAS_RES || EPI == EPI_BIAS_RES_LN);      // epilogues that add R[m][n]
    constexpr bool PRE = !(BN >= 384 && (EPI == EPI_BIAS || RES || EPI == EPI_EMBED));
    const f32x4 zero4 = f32x4{0.f, 0.f, 0.f, 0.f};
    auto load_col = [&](int nb) __attribute__((always_inline)) {
        ColOps o{zero4, zero4, zero4, zero4};
        if (!FULL && nb >= p.N) return o;
        if constexpr (EPI == EPI_EMBED) {
            // no bias operand: it is part of the table rows (load_res)
        } else if constexpr (COL_LDS) {
            o.bias = *reinterpret_cast<const f32x4*>(bias_s + (nb - n0));
        } else if constexpr (FULL) {
            o.bias = *reinterpret_cast<const f32x4*>(p.bias + nb);        // the launcher proved bias != null
        } else if (p.bias) {
#pragma unroll
            for (int q = 0; q < 4; ++q) o.bias[q] = (nb + q < p.N) ? p.bias[nb + q] : 0.f;
        }
        if constexpr (LN_CONSUMER) {
            if (p.ln_stats) {
#pragma unroll
                for (int q = 0; q < 4; ++q) o.c4[q] = (nb + q < p.N) ? p.ln_c[nb + q] : 0.f;
            }
        }
        if constexpr (EPI == EPI_BIAS_RES && !FULL) {
            if (p.r_stats) {
#pragma unroll
                for (int q = 0; q < 4; ++q)
                    if (nb + q < p.N) { o.g4[q] = p.r_gamma[nb + q]; o.b4[q] = p.r_beta[nb + q]; }
            }
        }
        if constexpr (EPI == EPI_BIAS_RES_LN) {      // FULL only: the affine part of the LayerNorm this epilogue applies
            o.g4 = *reinterpret_cast<const f32x4*>(p.ln_gamma + nb);
            o.b4 = *reinterpret_cast<const f32x4*>(p.ln_beta + nb);
        }
        return o;
    };
    auto load_res = [&](int m, int nb) __attribute__((always_inline)) {
        f32x4 rr = zero4;
        if constexpr (EPI == EPI_EMBED) {      // the row of the positional / timestep table this unit adds
            if (!FULL && (m >= p.M || nb >= p.N)) return rr;
            const int bidx = m / p.S, tok = m % p.S;
            const float* tp = (tok == 0) ? p.tab0 + (size_t)bidx * p.ldtab0 + nb : p.tab + (size_t)(p.tab_by_row ? m : tok) * p.ldtab + nb;
            if constexpr (FULL) {
                rr = *reinterpret_cast<const f32x4*>(tp);
            } else {
#pragma unroll
                for (int q = 0; q < 4; ++q)
                    if (nb + q < p.N) rr[q] = tp[q];
            }
        }
        if constexpr (RES) {
            if (!FULL && (m >= p.M || nb >= p.N)) return rr;
            const float* rp = p.R + (size_t)m * p.ldr + nb;
            if (FULL || (vec_ok && (p.ldr % 4 == 0) && (((uintptr_t)p.R & 15) == 0))) {
                rr = *reinterpret_cast<const f32x4*>(rp);
            } else {
#pragma unroll
                for (int q = 0; q < 4; ++q)
                    if (nb + q < p.N) rr[q] = rp[q];
            }
        }
        return rr;
    };
    auto finish = [&](int m, int nb, f32x4 a, int sl, const ColOps& o, f32x4 rr) __attribute__((always_inline)) {
        if (!FULL && (m >= p.M || nb >= p.N)) return;
        if constexpr (EPI == EPI_BIAS) {
            if (ksplit > 1) {      // raw partial tile; ld_partial is a multiple of 4 and covers N rounded up
                *reinterpret_cast<f32x4*>(p.partial + ((size_t)split * p.M + m) * p.ld_partial + nb) = a;
                return;
            }
        }
        f32x4 v;
        bool folded = false;
        if constexpr (LN_CONSUMER) {
            if (p.ln_stats) {       // (acc - mu c_n) rstd + d_n ; d_n arrives as the bias
                folded = true;
#pragma unroll
                for (int q = 0; q < 4; ++q) v[q] = (a[q] - amu[sl] * o.c4[q]) * ars[sl] + o.bias[q];
            }
        }
        if (!folded) {
#pragma unroll
            for (int q = 0; q < 4; ++q) v[q] = a[q] + o.bias[q];
        }
        if constexpr (EPI == EPI_BIAS_GELU) {
#pragma unroll
            for (int q = 0; q < 4; ++q) v[q] = gelu_erf(v[q]);
        }
        if constexpr (EPI == EPI_BIAS_RES) {
            if constexpr (!FULL) {
                if (p.r_stats) {        // the residual is LN(raw): normalise it on the fly
#pragma unroll
                    for (int q = 0; q < 4; ++q) rr[q] = (rr[q] - rmu[sl]) * rrs[sl] * o.g4[q] + o.b4[q];
                }
            }
#pragma unroll
            for (int q = 0; q < 4; ++q) v[q] += rr[q];
            if constexpr (LN_PRODUCER) {
#pragma unroll
                for (int q = 0; q < 4; ++q)
                    if (FULL || nb + q < p.N) { osum[sl] += v[q]; osq[sl] += v[q] * v[q]; }
            }
        }
        if constexpr (EPI == EPI_QKV) {
            if (nb < p.qcols) {      // qcols is a multiple of 4
#pragma unroll
                for (int q = 0; q < 4; ++q) v[q] *= p.qscale;
            }
        }
        if constexpr (EPI == EPI_EMBED) {
            const int tok = m % p.S;
#pragma unroll
            for (int q = 0; q < 4; ++q) v[q] = (tok == 0 ? 0.f : a[q]) + rr[q];
        }
        const size_t crow = CONV ? (size_t)m * (p.orow_mul_m1 + 1) + p.orow_add : (size_t)m;
        float* cp = p.C + crow * p.ldc + nb + ((CONV && p.ncol_split && nb >= p.ncol_split) ? p.ncol_jump : 0);
        if (vec_ok) {
            *reinterpret_cast<f32x4*>(cp) = v;
        } else {
#pragma unroll
            for (int q = 0; q < 4; ++q)
                if (nb + q < p.N) cp[q] = v[q];
        }
    };
    // every unit of this lane, in a fixed order with compile-time indices: fn(unit, column group, row, first column, accumulators, row slot)
    constexpr int NUNIT = M32 ? 16 * NCB32 + NCB : NCB * NRB;
    constexpr int NCG = M32 ? 4 * NCB32 + NCB : NCB;
    auto for_units = [&](auto&& fn) __attribute__((always_inline)) {
        if constexpr (M32) {
            // acc32[rb][cb][4q' + r] = C[m0 + rb*32 + li32][nw + cb*32 + 8q' + 4*lg32 + r]
#pragma unroll
            for (int rb = 0; rb < 4; ++rb)
#pragma unroll
                for (int cb = 0; cb < NCB32; ++cb)
#pragma unroll
                    for (int qq = 0; qq < 4; ++qq) {
                        const f32x16& a = acc32[rb * NCB32 + cb];
                        fn((rb * NCB32 + cb) * 4 + qq, cb * 4 + qq, m0 + rb * 32 + li32, nw + cb * 32 + 8 * qq + 4 * lg32,
                           f32x4{a[4 * qq], a[4 * qq + 1], a[4 * qq + 2], a[4 * qq + 3]}, rb);
                    }
            // acc16[c][r] = C[m0 + 128 + li][nw + c*16 + 4*lg + r]
#pragma unroll
            for (int c = 0; c < NCB; ++c) fn(16 * NCB32 + c, 4 * NCB32 + c, m0 + 128 + li, nw + c * 16 + lg * 4, acc16[c], 4);
        } else {
#pragma unroll
            for (int c = 0; c < NCB; ++c)
#pragma unroll
                for (int r = 0; r < NRB; ++r) fn(c * NRB + r, c, m0 + r * 16 + li, nw + c * 16 + lg * 4, acc16[r * NCB + c], r);
        }
    };
    // EARLY: the operands are requested at the top of the LAST chunk's iteration and land under its MFMAs (2 us of them); the
    // 384-wide tile has no registers to spare during the loop and requests them behind it.
    constexpr bool HAS_RES = (RES || EPI == EPI_EMBED);      // a per-unit operand besides the column operands
    constexpr bool PEEL = BN <= 256;
    constexpr bool EARLY = PRE && PEEL && FULL && EPI != EPI_OUT_T && !(BN >= 256 && HAS_RES);
    ColOps col[PRE ? NCG : 1];
    f32x4 res[(PRE && HAS_RES) ? NUNIT : 1];
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
        if constexpr (HAS_RES)
            for_units([&](int i, int, int m, int nb, f32x4, int) __attribute__((always_inline)) { res[i] = load_res(m, nb); });
    };
    // every wave issues exactly PIECES loads per chunk (the statistics loads above are younger: waiting for
    // vmcnt <= PIECES still means chunk 0 has landed)
    if (nk > 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PIECES) : "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if constexpr (VAR == 7) { ts_wall[1] = wall_clock64(); ts_cyc[0] = __builtin_readcyclecounter(); }
    read_frags(f0, 0, 0);
    constexpr int NG = READS;                       // one LDS read per scheduling group
    constexpr int MF = (MFMAS + NG - 1) / NG;       // MFMAs per group (the tail groups run dry, harmless)
    // The last chunk's iteration is peeled: it prefetches nothing (no wait, no barrier -- nobody overwrites a buffer any more),
    // and it is where the epilogue's operands are requested (EARLY).
    auto chunk = [&](int kc, auto last_tag) __attribute__((always_inline)) {
        constexpr bool LAST = decltype(last_tag)::value;
        const int buf = kc & 1;
        if constexpr (VAR != 5) {
            if constexpr (LAST && EARLY) request_ops();
            read_frags(f1, buf, 1);
            mma_half(f0);
            SchedGroups<0, NG, MF, READS, 0, 0, CONV ? 6 : 0>::run();
            __builtin_amdgcn_sched_barrier(0);
            if constexpr (!LAST) {
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                         // chunk k+1 landed
                __syncthreads();
                __builtin_amdgcn_sched_barrier(0);
                // the last iterations re-fetch the last chunk (into a buffer nobody reads again) instead of being predicated:
                // the body stays one basic block
                const int kn = kbase + ((kc + 2 < nk) ? (kc + 2) * BK : (nk - 1) * BK);
                dma(buf, kn);
                read_frags(f0, buf ^ 1, 0);
                mma_half(f1);
                SchedGroups<0, NG, MF, READS, PIECES, 1, CONV ? 6 : 0>::run();
            } else {
                mma_half(f1);
            }
            __builtin_amdgcn_sched_barrier(0);
        } else {
            mma_half(f0);
            mma_half(f0);
        }
    };
    if constexpr (PEEL) {
        for (int kc = 0; kc + 1 < nk; ++kc) chunk(kc, std::false_type{});
        chunk(nk - 1, std::true_type{});
    } else {                                  // 384-wide tile: the peeled copy costs it registers it does not have (82 spills)
        for (int kc = 0; kc < nk; ++kc) chunk(kc, std::false_type{});
    }
    // the re-fetched chunk of the second to last iteration must have landed before the workgroup gives its LDS back; with EARLY the
    // (younger) operand loads are waited for below, which covers it
    if constexpr (!EARLY) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if constexpr (VAR == 7) { ts_wall[2] = wall_clock64(); ts_cyc[1] = __builtin_readcyclecounter(); }
    if constexpr (VAR == 5 || VAR == 6) {
        // diagnostics: skip the epilogue unless an impossible value appears (keeps the MFMAs live)
        if (acc16[0][0] != 123456.789f) return;
    }

    // ---- epilogue --------------------------------------------------------------------------------------------
    if constexpr (EPI == EPI_BIAS_RES_LN) {
        // C = LayerNorm(acc + bias + R): see GemmParams::xln_*.  FULL, 16x16 layouts (BN <= 128) only -- the launcher sees to it.
        static_assert(!M32 && !CONV && FULL && PRE, "EPI_BIAS_RES_LN: 144 x 64 / 144 x 128 tiles of whole problems only");
        if constexpr (!EARLY) request_ops();
        // 1. x = (acc + bias) + R in the accumulator registers (the order of the un-fused path)
#pragma unroll
        for (int c = 0; c < NCB; ++c)
#pragma unroll
            for (int r = 0; r < NRB; ++r) {
                f32x4& a = acc16[r * NCB + c];
                const f32x4 rr = res[c * NRB + r];
#pragma unroll
                for (int q = 0; q < 4; ++q) a[q] = (a[q] + col[c].bias[q]) + rr[q];
            }
        // 2. row statistics as (mean, M2 = sum of squares about that mean) of equal-sized parts, merged pairwise by Chan's update
        //        mean = (m_a + m_b) / 2,   M2 = M2_a + M2_b + n (m_b - m_a)^2 / 2        (n = elements per part)
        //    at every level: the lane's own 4 NCB columns (two-pass in registers), the 4 lanes li + 16 lg of a row (two lane swaps on
        //    the VALU), the 4 waves (LDS), the partner tiles (L2).  As stable as nn.LayerNorm's own two-pass kernel -- no
        //    E[x^2] - mu^2 cancellation for rows far from zero -- at the price of a few VALU operations, no extra barrier.
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
        float* const part = lds_dummy + 512;            // [4 waves][BM][2] (mean, M2) of a wave's columns; afterwards [BM][2] = (mu, rstd)
        constexpr float kLane = (float)(4 * NCB);       // elements per lane and row
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
            merge_swap(m, q2, kLane, true);             // lanes l and l ^ 32
            merge_swap(m, q2, 2.0f * kLane, false);     // ... and l ^ 16: the wave's WN columns of the row
            if (lg == 0) *reinterpret_cast<f32x2*>(part + (wave * BM + r * 16 + li) * 2) = f32x2{m, q2};
        }
        __syncthreads();
        const int g = m0 / BM, tn = n0 / BN;
        // which XCD this tile ran on: recorded for the tests, and part of the tag -- a partner that publishes from another XCD is
        // reported (status 2), not trusted silently
        const unsigned xcc1 = 1u + (unsigned)__builtin_amdgcn_s_getreg((3 << 11) | 20);      // 1 .. 8
        if (tid == 0) p.xln_xcc[g * 8 + tn] = xcc1;
        // 3. publish this tile's (mean, M2) of every row and collect the partner tiles'.  A slot is 16 bytes (mean, tag, M2, tag)
        //    written by ONE store: each 8-byte half carries the launch's tag, so a reader that sees both tags has the data (no
        //    separate flag, no wait for the store's acknowledgement, no counter to re-arm: three dependent trips to L2 less than
        //    publish / count / poll).  tag = (epoch of the launch, 28 bits) << 4 | (XCD id + 1): never 0, so zeroed memory is stale.
        //    Tiles are merged in a fixed tree over the column-tile index with the own pair taken from registers at its place: every
        //    tile of the row gets bit-identical statistics whatever the arrival order.
        float* const row_stats = p.xln_stats + ((size_t)g * tiles_n * BM) * 4;
        const unsigned ep28 = xln_ep & 0x0fffffffu;
        const unsigned tag = (ep28 << 4) | xcc1;
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
            merge(m01, q01, m23, q23, 2.0f * (float)WN, a, b);                     // this tile's BN columns of row tid
            // (test hook, rohm_posenet_inject_exchange_fault: column tile 0 publishes under a tag nobody waits for)
            const unsigned pub = (p.xln_fault && tn == 0) ? (tag ^ 0x80000000u) : tag;
            *reinterpret_cast<f32x4*>(row_stats + ((size_t)tn * BM + tid) * 4) = f32x4{a, __uint_as_float(pub), b, __uint_as_float(pub)};
            float mk[8], qk[8];
            for (int it = 0;; ++it) {
                unsigned long long lo[8], hi[8];
#pragma unroll
                for (int k = 0; k < 8; ++k) {
                    lo[k] = hi[k] = 0ull;
                    if (k < tiles_n && k != tn) {          // past the L1: the partner's store went to L2
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
                    if (!same_xcd) __hip_atomic_store(p.xln_err, 2u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    break;
                }
                __builtin_amdgcn_s_sleep(1);
                // a wait that another launch of this pass (or an earlier row) already lost is not worth 0.2 s again: the results since
                // the last status check are void anyway, the host falls back and re-runs (rohm_amd/diffusion/ddpm.py)
                if ((it & 127) == 127 && __hip_atomic_load(p.xln_err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u) break;
                if (it > (1 << 19)) {                   // ~0.2 s: never on a healthy device (the partner tiles are co-resident); do not hang it
                    __hip_atomic_store(p.xln_err, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    break;
                }
            }
            // tiles_n is 1, 2, 4 or 8: a balanced tree over the tile index (static indices: the arrays stay in registers)
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
            *reinterpret_cast<f32x2*>(part + tid * 2) = f32x2{mu, 1.0f / sqrtf(var + p.ln_eps)};      // own row of wave 0's zone: read above
        }
        __syncthreads();
        // 4. normalise in registers, store LN(x) once
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
    } else if constexpr (EPI == EPI_OUT_T) {
        if constexpr (SK) {
            constexpr size_t kSlot = (size_t)N16 * 256 * 4;        // floats per workgroup slot: accumulator i of thread t at (i * 256 + t) * 4
            if (sk_role == 1) {
                // head of a tile: leave the raw accumulators for the tile's owner.  The stores are acknowledged by L2 (vmcnt) before
                // the flag goes out; the owner sits on the same XCD, whose L2 is the point both meet at.
                float* slot = p.sk_part + (size_t)blockIdx.x * kSlot;
#pragma unroll
                for (int i = 0; i < N16; ++i) *reinterpret_cast<f32x4*>(slot + ((size_t)i * 256 + tid) * 4) = acc16[i];
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                __syncthreads();
                if (tid == 0) {
                    const unsigned long long xcc = 1u + (unsigned)__builtin_amdgcn_s_getreg((3 << 11) | 20);
                    __hip_atomic_store(p.sk_flag + blockIdx.x, (xcc << 32) | xln_ep, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
                continue;
            }
            if (sk_role == 2) {
                // tail of a tile: add what the workgroups in front of us (8, 16, .. blocks below: same XCD) accumulated over the
                // tile's earlier chunks -- nearest first, a fixed order.  With sk_units >= chunks per tile that is one workgroup.
                const int tile_u0 = (n0 / BN * tiles_m + m0 / BM - (int)(blockIdx.x % kNumXCD) * p.sk_tiles8) * (p.K / BK);
                for (int jb = (int)(blockIdx.x / kNumXCD) - 1; jb >= 0 && (jb + 1) * p.sk_units > tile_u0; --jb) {
                    const unsigned src = (unsigned)jb * kNumXCD + blockIdx.x % kNumXCD;
                    if (tid == 0) {
                        const unsigned long long xcc = 1u + (unsigned)__builtin_amdgcn_s_getreg((3 << 11) | 20);
                        for (int it = 0;; ++it) {
                            const unsigned long long f = __hip_atomic_load(p.sk_flag + src, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                            if ((unsigned)f == xln_ep && (f >> 32) != 0ull) {      // (XCD id + 1 is never 0: zeroed memory is stale)
                                // a producer on another XCD would have left its partial in another L2: never with the hardware's
                                // round-robin placement, and not survivable silently
                                if ((f >> 32) != xcc) __hip_atomic_store(p.xln_err, 2u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                                break;
                            }
                            __builtin_amdgcn_s_sleep(1);
                            if ((it & 127) == 127 && __hip_atomic_load(p.xln_err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u) break;
                            if (it > (1 << 19)) {                   // ~0.2 s: do not hang the device; the host reads the word
                                __hip_atomic_store(p.xln_err, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                                break;
                            }
                        }
                    }
                    __syncthreads();
                    const unsigned long long* slot = reinterpret_cast<const unsigned long long*>(p.sk_part + (size_t)src * kSlot);
                    unsigned long long lo[N16], hi[N16];
#pragma unroll
                    for (int i = 0; i < N16; ++i) {                 // past the L1
                        lo[i] = __hip_atomic_load(slot + ((size_t)i * 256 + tid) * 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        hi[i] = __hip_atomic_load(slot + ((size_t)i * 256 + tid) * 2 + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    }
#pragma unroll
                    for (int i = 0; i < N16; ++i) {
                        acc16[i][0] += __uint_as_float((unsigned)lo[i]);
                        acc16[i][1] += __uint_as_float((unsigned)(lo[i] >> 32));
                        acc16[i][2] += __uint_as_float((unsigned)hi[i]);
                        acc16[i][3] += __uint_as_float((unsigned)(hi[i] >> 32));
                    }
                }
            }
        }
        // natural operand order: rows = output channels m, cols = tokens n; stored transposed into [B, C_total, 1, T]
        // LN fold: the normalised operand is the token (column) side; a lane's columns are fixed per column block
        constexpr int NCOL = M32 ? NCB32 + NCB : NCB;
        float cmu[NCOL], crs[NCOL];
        if (p.ln_stats) {
#pragma unroll
            for (int c = 0; c < NCOL; ++c) {
                int n;
                if constexpr (M32) n = (c < NCB32) ? nw + c * 32 + li32 : nw + (c - NCB32) * 16 + li;
                else n = nw + c * 16 + li;
                n = n < p.N ? n : p.N - 1;
                row_mu_rstd(p.ln_stats, p.ln_parts, n, p.ln_dim, p.ln_eps, cmu[c], crs[c]);
            }
        }
        // bias[m] / ln_c[m] of every row of this lane are read BEFORE the first store (see the note on `finish` below: behind a
        // store to C the compiler must re-read them, one dependent round trip per value)
        auto row_ops = [&](int m, float& bm, float& cm) __attribute__((always_inline)) {
            const int mc = m < p.M ? m : p.M - 1;
            bm = p.bias[mc];
            cm = p.ln_stats ? p.ln_c[mc] : 0.f;
        };
        auto put = [&](int m, int n, float v, int cslot, float bm, float cm) __attribute__((always_inline)) {
            if (m >= p.M || n >= p.N) return;
            const int b = n / p.S, tok = n % p.S;
            if (tok == 0) return;
            if (p.ln_stats) v = (v - cmu[cslot] * cm) * crs[cslot];
            p.C[((size_t)b * p.C_total + p.ch_off + m) * p.T + (tok - 1)] = v + bm;
        };
        if constexpr (M32) {
            float bm32[4][16], cm32[4][16], bm16[4], cm16[4];
#pragma unroll
            for (int rb = 0; rb < 4; ++rb)
#pragma unroll
                for (int q = 0; q < 16; ++q) row_ops(m0 + rb * 32 + 8 * (q >> 2) + 4 * lg32 + (q & 3), bm32[rb][q], cm32[rb][q]);
#pragma unroll
            for (int q = 0; q < 4; ++q) row_ops(m0 + 128 + lg * 4 + q, bm16[q], cm16[q]);
#pragma unroll
            for (int rb = 0; rb < 4; ++rb)
#pragma unroll
                for (int cb = 0; cb < NCB32; ++cb)
#pragma unroll
                    for (int q = 0; q < 16; ++q)
                        put(m0 + rb * 32 + 8 * (q >> 2) + 4 * lg32 + (q & 3), nw + cb * 32 + li32, acc32[rb * NCB32 + cb][q], cb,
                            bm32[rb][q], cm32[rb][q]);
#pragma unroll
            for (int c = 0; c < NCB; ++c)
#pragma unroll
                for (int q = 0; q < 4; ++q) put(m0 + 128 + lg * 4 + q, nw + c * 16 + li, acc16[c][q], NCB32 + c, bm16[q], cm16[q]);
        } else {
            float bm[NRB][4], cm[NRB][4];
#pragma unroll
            for (int r = 0; r < NRB; ++r)
#pragma unroll
                for (int q = 0; q < 4; ++q) row_ops(m0 + r * 16 + lg * 4 + q, bm[r][q], cm[r][q]);
#pragma unroll
            for (int r = 0; r < NRB; ++r)
#pragma unroll
                for (int c = 0; c < NCB; ++c)
#pragma unroll
                    for (int q = 0; q < 4; ++q) put(m0 + r * 16 + lg * 4 + q, nw + c * 16 + li, acc16[r * NCB + c][q], c, bm[r][q], cm[r][q]);
        }
    } else {
        // swapped operand order: a lane holds C[m][nb .. nb+3].  FULL: the launcher proved M % 144 == 0,
        // N % BN == 0 and 16-byte alignment of C / R / bias / tables, so the hot instantiation carries no edge
        // masks and only 16-byte accesses.
        if constexpr (PRE) {
            if constexpr (!EARLY) request_ops();
            if constexpr (HAS_RES)
                for_units([&](int i, int cg, int m, int nb, f32x4 a, int sl) __attribute__((always_inline)) { finish(m, nb, a, sl, col[cg], res[i]); });
            else
                for_units([&](int, int cg, int m, int nb, f32x4 a, int sl) __attribute__((always_inline)) { finish(m, nb, a, sl, col[cg], zero4); });
        } else {
            for_units([&](int, int, int m, int nb, f32x4 a, int sl) __attribute__((always_inline)) { finish(m, nb, a, sl, load_col(nb), load_res(m, nb)); });
        }
        if constexpr (LN_PRODUCER) {
            if (p.out_stats) {
                // a row's columns of this tile live in the 4 lanes (li, lg = 0..3) of each of the 4 waves: two shuffles,
                // then the waves meet in LDS (a zone past the staging buffers and the DMA landing zone)
                float* part = lds_dummy + 512;                  // [4 waves][BM][2]
#pragma unroll
                for (int sl = 0; sl < NSLOT; ++sl) {
                    float a = osum[sl], b = osq[sl];
                    a += __shfl_xor(a, 16); b += __shfl_xor(b, 16);
                    a += __shfl_xor(a, 32); b += __shfl_xor(b, 32);
                    if (lg == 0) { part[(wave * BM + sl * 16 + li) * 2] = a; part[(wave * BM + sl * 16 + li) * 2 + 1] = b; }
                }
                __syncthreads();
                if (tid < BM && m0 + tid < p.M) {
                    float a = 0.f, b = 0.f;
#pragma unroll
                    for (int w = 0; w < 4; ++w) { a += part[(w * BM + tid) * 2]; b += part[(w * BM + tid) * 2 + 1]; }
                    // one slot per 64 columns whatever BN the launcher picked: this tile fills its first slot and
                    // zeroes the others it covers
                    float* o = p.out_stats + ((size_t)(m0 + tid) * p.out_parts + (n0 / 64)) * 2;
                    o[0] = a; o[1] = b;
#pragma unroll
                    for (int k = 1; k < BN / 64; ++k)
                        if (n0 / 64 + k < p.out_parts) { o[2 * k] = 0.f; o[2 * k + 1] = 0.f; }
                }
            }
        }
    }
    if constexpr (VAR == 7) {
        ts_wall[3] = wall_clock64();
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");        // stores acknowledged
        ts_wall[4] = wall_clock64();
        if (tid == 0) {
            unsigned long long* ts = reinterpret_cast<unsigned long long*>(const_cast<float*>(p.R)) + (size_t)blockIdx.x * 8;
            for (int i = 0; i < 5; ++i) ts[i] = ts_wall[i];
            ts[5] = ts_cyc[1] - ts_cyc[0];
            ts[6] = __builtin_amdgcn_s_getreg((3 << 11) | 20);   // XCC_ID
            ts[7] = __builtin_amdgcn_s_getreg((31 << 11) | 4);   // HW_ID
        }
    }
    } while (SK && sk_hi > sk_lo);
}

// Second pass of a split-K GEMM: C[crow(m)][n] = bias[n] + sum_s partial[s][m][n], splits added in index order
// (deterministic).  One thread per 4 columns.
__global__ __launch_bounds__(256) void splitk_reduce_kernel(const float* __restrict__ partial, int S, int M, int N,
                                                            int ldp, const float* __restrict__ bias, float* __restrict__ C,
                                                            int ldc, int orow_mul_m1, int orow_add, int ncol_split,
                                                            int ncol_jump) {
    const int n4 = (N + 3) / 4;
    const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (long)M * n4) return;
    const int m = (int)(idx / n4), nb = (int)(idx % n4) * 4;
    f32x4 acc = *reinterpret_cast<const f32x4*>(partial + (size_t)m * ldp + nb);
    for (int s = 1; s < S; ++s) {
        const f32x4 v = *reinterpret_cast<const f32x4*>(partial + ((size_t)s * M + m) * ldp + nb);
#pragma unroll
        for (int q = 0; q < 4; ++q) acc[q] += v[q];
    }
    float* cp = C + ((size_t)m * (orow_mul_m1 + 1) + orow_add) * ldc + nb + ((ncol_split && nb >= ncol_split) ? ncol_jump : 0);
#pragma unroll
    for (int q = 0; q < 4; ++q)
        if (nb + q < N) cp[q] = acc[q] + (bias ? bias[nb + q] : 0.f);
}

// Diagnostic builds only (`ROHM_DIAG=1 python -m rohm_amd.build` defines ROHM_GEMM_DIAGNOSTICS): schedule variants
// 5 / 6 / 7, forced tile widths and the occupancy / LDS-padding / target-workgroup knobs used by scripts/gemm_*.  The
// shipped library has none of them: no environment lookups on the launch path, no diagnostic kernels in the binary.
#ifdef ROHM_GEMM_DIAGNOSTICS
static int gemm_variant() {
    static int v = -1;
    if (v < 0) {
        const char* e = getenv("ROHM_GEMM_VARIANT");
        v = e ? atoi(e) : 0;
    }
    return v;
}
static int diag_env_int(const char* name, int dflt) {
    const char* e = getenv(name);
    return (e && *e) ? atoi(e) : dflt;
}
#else
static constexpr int gemm_variant() { return 0; }
#endif

template <int BN, int EPI, int VAR = 0, bool FULL = false, bool CONV = false>
static int launch_one(const GemmParams& p, hipStream_t s) {
    const int ksplit = (EPI == EPI_BIAS && p.ksplit > 1) ? p.ksplit : 1;
    int tiles = ((p.M + BM - 1) / BM) * ((p.N + BN - 1) / BN) * ksplit;
    if (CONV && EPI == EPI_BIAS && p.res_col0 > 0)      // conv tiles x their splits, then the residual-column tiles x theirs
        tiles = ((p.M + BM - 1) / BM) * ((p.res_col0 / BN) * ksplit + (((p.N + BN - 1) / BN) - p.res_col0 / BN) * (p.res_ksplit > 1 ? p.res_ksplit : 1));
    if (EPI == EPI_BIAS_RES_LN)      // row tiles dealt round-robin to the XCDs, rounded up to a multiple of 8 (the kernel's own map)
        tiles = (((p.M + BM - 1) / BM + kNumXCD - 1) / kNumXCD) * kNumXCD * ((p.N + BN - 1) / BN);
    if (EPI == EPI_OUT_T && p.sk_units > 0) {      // stream-K: one workgroup per CU, the units dealt out evenly
        if (!(BN == 64 && !FULL && !CONV && VAR == 0)) { set_error("gemm: stream-K exists for the 144 x 64 output-head tiles only"); return ROHM_ERR_ARG; }
        tiles = kSkWorkgroups;
    }
#ifdef ROHM_GEMM_DIAGNOSTICS
    static const int lds_pad = diag_env_int("ROHM_GEMM_LDS_PAD", 0);
    static const bool occ2 = getenv("ROHM_GEMM_OCC2") != nullptr;      // allow two workgroups per CU
#else
    constexpr int lds_pad = 0;
    constexpr bool occ2 = false;
#endif
    // One workgroup per CU on purpose: with two co-resident workgroups the hardware hands BOTH freed slots of
    // a CU to the next tiles, so a 3-tiles-per-CU GEMM degenerates to 4 + 2 (measured 158 us vs 128 us); the
    // schedule already hides LDS / L2 latency without a partner wave.  Requesting more than half of the
    // 160 KiB LDS pins the residency to one.
    size_t lds = (size_t)2 * (BM + BN) * BK * sizeof(float) + lds_pad;
    lds += 2048;                           // landing zone of the dummy DMA pieces
    lds += 4 * BM * 2 * sizeof(float);     // row-stat exchange of the LayerNorm-producing epilogues
    const size_t lds_need = lds;
    if (lds < 84 * 1024 && !occ2 && p.wg_per_cu < 2) lds = 84 * 1024;
    static bool attr_set[64] = {};
    int dev = 0;
    ROHM_HIP_CHECK(hipGetDevice(&dev));
    if (dev < 64 && !attr_set[dev]) {
        const size_t lds_max = lds_need > 84 * 1024 ? lds_need : 84 * 1024;     // either residency of later launches
        ROHM_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_f32_kernel<BN, EPI, VAR, FULL, CONV>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_max));
        attr_set[dev] = true;
    }
    static const char* const kNames[] = {"gemm_bias", "gemm_bias_gelu", "gemm_bias_res", "gemm_qkv", "gemm_embed",
                                         "gemm_out_t", "gemm_bias_res_ln"};
    static const char* const kNames64[] = {"gemm_bias/64", "gemm_bias_gelu/64", "gemm_bias_res/64", "gemm_qkv/64",
                                           "gemm_embed/64", "gemm_out_t/64", "gemm_bias_res_ln/64"};
    const char* label = CONV ? (BN == 64 ? "conv_gemm/64" : "conv_gemm") : (BN == 64 ? kNames64[EPI] : kNames[EPI]);
    if (prof::detail()) {
        char buf[48];
        snprintf(buf, sizeof(buf), "%s M%d N%d K%d S%d", label, p.M, p.N, p.K, ksplit);
        label = prof::intern(buf);
    }
    {
    prof::Scope ps(label, 2.0 * p.M * p.N * p.K, 4.0 * ((double)p.M * p.K + (double)p.N * p.K + (double)p.M * p.N), s);
    hipLaunchKernelGGL((gemm_f32_kernel<BN, EPI, VAR, FULL, CONV>), dim3(tiles), dim3(256), lds, s, p);
    ROHM_LAUNCH_CHECK();
    }
    if ((ksplit > 1 || (p.res_col0 > 0 && p.res_ksplit > 1)) && !p.ksplit_defer) {
        if (p.res_col0 > 0) { set_error("gemm: a fused [conv | residual] launch must leave its split-K slabs to the consumer"); return ROHM_ERR_ARG; }
        prof::Scope pr("splitk_reduce", 0.0, 4.0 * ((double)ksplit + 1.0) * p.M * p.N, s);
        const long n = (long)p.M * ((p.N + 3) / 4);
        hipLaunchKernelGGL(splitk_reduce_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, p.partial, ksplit, p.M,
                           p.N, p.ld_partial, p.bias, p.C, p.ldc, CONV ? p.orow_mul_m1 : 0, CONV ? p.orow_add : 0,
                           CONV ? p.ncol_split : 0, CONV ? p.ncol_jump : 0);
        ROHM_LAUNCH_CHECK();
    }
    return ROHM_OK;
}

static inline bool al16(const void* q) { return (((uintptr_t)q) & 15) == 0; }

template <int BN, int EPI, int VAR = 0>
static int launch_t(const GemmParams& p, hipStream_t s) {
    bool full = (p.M % BM == 0) && (p.N % BN == 0) && (p.ldc % 4 == 0) && al16(p.C) && al16(p.bias) && (p.bias || EPI == EPI_EMBED);
    if (p.ln_stats || p.r_stats || p.out_stats) full = false;      // LayerNorm folding: general instantiation only
    if (EPI == EPI_BIAS_RES) full = full && (p.ldr % 4 == 0) && al16(p.R);
    if (EPI == EPI_EMBED) full = full && (p.ldtab % 4 == 0) && (p.ldtab0 % 4 == 0) && al16(p.tab) && al16(p.tab0);
    if (EPI == EPI_QKV) full = full && (p.qcols % 4 == 0);
    full = full && al16(p.ln_c) && al16(p.r_gamma) && al16(p.r_beta);      // 16-byte loads of the LayerNorm vectors
    if (EPI == EPI_OUT_T || (VAR != 0 && VAR != 7)) full = false;
    if (p.conv_taps > 0) {
        if constexpr (EPI == EPI_BIAS && VAR == 0 && BN <= 128) {
            return full ? launch_one<BN, EPI, 0, true, true>(p, s) : launch_one<BN, EPI, 0, false, true>(p, s);
        } else {
            set_error("gemm: conv gather supports the bias epilogue and BN <= 128 only");
            return ROHM_ERR_UNSUPPORTED;
        }
    }
    if (full) {
        if constexpr (EPI != EPI_OUT_T && (VAR == 0 || VAR == 7)) return launch_one<BN, EPI, VAR, true>(p, s);
    }
    return launch_one<BN, EPI, VAR, false>(p, s);
}

// Workgroups a launch should at least produce before a wider tile is preferred: one per CU of the MI355X, or
// fewer when the caller runs several independent launches side by side (ROHM_GEMM_TARGET_WGS).
static int target_wgs() {
#ifdef ROHM_GEMM_DIAGNOSTICS
    static const int v = diag_env_int("ROHM_GEMM_TARGET_WGS", 256) > 0 ? diag_env_int("ROHM_GEMM_TARGET_WGS", 256) : 256;
    return v;
#else
    return 256;
#endif
}

template <int EPI>
static int launch_bn(const GemmParams& p, hipStream_t s) {
    const int tm = (p.M + BM - 1) / BM;
    const int tiles128 = tm * ((p.N + 127) / 128);
    const int want = target_wgs();
#ifdef ROHM_GEMM_DIAGNOSTICS
    const int var = gemm_variant();
    const int force_bn = var / 10;            // 6x -> BN 64, 12x -> BN 128
    if constexpr (EPI == EPI_BIAS) {
        switch (var % 10) {                   // schedule variants exist for the plain epilogue only
            case 5: return force_bn == 6 ? launch_t<64, EPI, 5>(p, s) : launch_t<128, EPI, 5>(p, s);
            case 6: return force_bn == 6 ? launch_t<64, EPI, 6>(p, s) : launch_t<128, EPI, 6>(p, s);
            case 7:                           // same tile choice as the shipped path
                if (!p.R) break;
                if (force_bn == 12) return launch_t<128, EPI, 7>(p, s);
                if (p.N % 384 == 0 && tm * (p.N / 384) >= want) return launch_t<384, EPI, 7>(p, s);
                if (p.N % 256 == 0 && tm * (p.N / 256) >= want) return launch_t<256, EPI, 7>(p, s);
                return launch_t<128, EPI, 7>(p, s);
            default: break;
        }
    }
    if (force_bn == 6) return launch_t<64, EPI>(p, s);
    if (force_bn == 12) return launch_t<128, EPI>(p, s);
#endif
    // Tile width: minimise  rounds x (BN + per-tile overhead),  rounds = ceil(tiles / workgroup slots).  Wider
    // tiles move fewer LDS-DMA bytes and fragment reads per MFMA and amortise the per-chunk barrier, but only while
    // every CU still gets a tile.  B = 64: N = 1536 -> 144x384, 1024 -> x256, 512 -> x128 (256 tiles each);
    // B = 32 (the per-GPU batch of the full-scheme configs): 1536 -> x192, 1024 -> x128, 512 -> x64.
    int best = 64;
    if (p.conv_taps == 0 && EPI != EPI_OUT_T) {
        long best_cost = -1;
        const int cand[5] = {384, 256, 192, 128, 64};
        for (int bn : cand) {
            if (bn > 64 && p.N % bn != 0) continue;
            if (p.out_stats && bn > 128) continue;      // the row-stat epilogue exists for the 16x16 layouts only
            const long tiles = (long)tm * ((p.N + bn - 1) / bn);
            const long cost = ((tiles + want - 1) / want) * (bn + 24);
            if (best_cost < 0 || cost < best_cost) { best_cost = cost; best = bn; }
        }
    } else if (tiles128 >= want && p.N % 128 == 0) {
        best = 128;
    }
    if constexpr (EPI != EPI_OUT_T) {
        if (p.conv_taps == 0) {
            if (best == 384) return launch_t<384, EPI>(p, s);
            if (best == 256) return launch_t<256, EPI>(p, s);
            if (best == 192) return launch_t<192, EPI>(p, s);
        }
    }
    if (best == 128) return launch_t<128, EPI>(p, s);
    return launch_t<64, EPI>(p, s);
}

// ---- EPI_BIAS_RES_LN: LayerNorm inside the producer GEMM --------------------------------------------------------------------
// Tile width: 144 x 128 while that still gives every CU a tile, else 144 x 64; the N / BN column tiles of a row tile exchange
// statistics, so there may be 1, 2, 4 or 8 of them (a divisor of the 32 CUs of an XCD: partner tiles are dispatched back to back
// onto one XCD and never straddle a round of workgroups).
static int ln_tile_width(int tm, int N) {
    auto ok = [&](int bn) { const int t = N / bn; return N % bn == 0 && t >= 1 && t <= 8 && 32 % t == 0; };
    if (ok(128) && (long)tm * (N / 128) >= target_wgs()) return 128;
    if (ok(64)) return 64;
    if (ok(128)) return 128;
    return 0;
}

bool gemm_ln_supported(int M, int N, int K) {
    return M > 0 && M % BM == 0 && K > 0 && K % BK == 0 && N > 0 && ln_tile_width(M / BM, N) != 0;
}

// scratch layout: [exchange header, 64 B: error word, magic, pass counter] [statistics: tiles_m x 8 column tiles x 144 rows x (mean, tag, M2, tag)] [XCD id + 1 of every tile:
// tiles_m x 8 words]
static size_t ln_stats_bytes(int M) { return (size_t)((M + BM - 1) / BM) * 8 * BM * 4 * sizeof(float); }
size_t gemm_ln_scratch_bytes(int M, int N) {
    (void)N;
    return 64 + ln_stats_bytes(M) + (size_t)((M + BM - 1) / BM) * 8 * sizeof(unsigned);
}
void gemm_ln_bind(GemmParams& p, void* scratch) {
    char* c = static_cast<char*>(scratch);
    p.xln_err = reinterpret_cast<unsigned*>(c);
    p.xln_pass = reinterpret_cast<unsigned*>(c) + 2;      // exchange header (exchange.hip): [0] error, [1] magic, [2] pass counter
    p.xln_stats = reinterpret_cast<float*>(c + 64);
    p.xln_xcc = reinterpret_cast<unsigned*>(c + 64 + ln_stats_bytes(p.M));
}

// The tag of a launch's exchange slots (LayerNorm statistics, stream-K flags) = GemmParams::xln_epoch (host part: a salt of the
// caller + the launch's index within a pass of its network) + 64 x *xln_pass, a device word the caller's first kernel of every pass
// increments (exchange.hip).  Nothing is cleared or re-armed between launches, and because the changing part lives in device memory
// a launch recorded into a hipGraph gets a fresh tag on every replay.

// ---- stream-K for EPI_OUT_T ----------------------------------------------------------------------------------------------------
constexpr size_t kSkSlotBytes = (size_t)NRB * 256 * 4 * sizeof(float);      // the 9 accumulator quads of 256 threads (144 x 64 tile)
size_t gemm_sk_scratch_bytes() { return (size_t)kSkWorkgroups * 8 + (size_t)kSkWorkgroups * kSkSlotBytes; }
void gemm_sk_bind(GemmParams& p, void* scratch, unsigned* err) {
    p.sk_flag = static_cast<unsigned long long*>(scratch);
    p.sk_part = reinterpret_cast<float*>(static_cast<char*>(scratch) + (size_t)kSkWorkgroups * 8);
    p.xln_err = err;
    p.xln_pass = err + 2;      // `err` is the first word of an exchange header
}
// Possible when every XCD gets the same number of tiles and every workgroup the same number of units, with no tile in more than
// three pieces (units per workgroup >= half the chunks of a tile); worth it when that is at least four chunks (~5 us) shorter than
// the rounds of whole tiles it replaces: B = 64 (288 tiles): 18 units against 2 x 16; B = 32 (144 tiles on 256 CUs): 9 against 16.
bool gemm_sk_plan(int M, int N, int K, int* units, int* tiles8) {
    const int tiles = ((M + BM - 1) / BM) * ((N + 63) / 64), nk = K / BK, per_xcd = kSkWorkgroups / kNumXCD;
    if (tiles % kNumXCD != 0 || K % BK != 0 || nk < 2) return false;
    const int t8 = tiles / kNumXCD;
    if ((t8 * nk) % per_xcd != 0) return false;
    const int u = (t8 * nk) / per_xcd, rounds = (tiles + kSkWorkgroups - 1) / kSkWorkgroups;
    if (2 * u < nk || rounds * nk - u < 4) return false;
    if (units) *units = u;
    if (tiles8) *tiles8 = t8;
    return true;
}

bool stream_is_capturing(hipStream_t s) {
    hipStreamCaptureStatus st = hipStreamCaptureStatusNone;
    if (hipStreamIsCapturing(s, &st) != hipSuccess) { (void)hipGetLastError(); return false; }
    return st != hipStreamCaptureStatusNone;
}

static int launch_ln(const GemmParams& p, hipStream_t s) {
    ROHM_ARG_CHECK(gemm_ln_supported(p.M, p.N, p.K), "gemm: shape (%d, %d, %d) has no in-kernel LayerNorm form", p.M, p.N, p.K);
    ROHM_ARG_CHECK(p.bias && p.R && p.ln_gamma && p.ln_beta && p.xln_stats && p.xln_err && p.xln_pass, "gemm: LayerNorm epilogue: null operand");
    ROHM_ARG_CHECK(p.ln_dim == p.N && p.ldc % 4 == 0 && p.ldr % 4 == 0 && al16(p.C) && al16(p.R) && al16(p.bias) && al16(p.ln_gamma) &&
                       al16(p.ln_beta) && (((uintptr_t)p.xln_stats) & 15) == 0 && p.ksplit <= 1 && p.conv_taps == 0,
                   "gemm: LayerNorm epilogue needs ln_dim == N and 16-byte aligned operands");
    if (ln_tile_width(p.M / BM, p.N) == 128) return launch_one<128, EPI_BIAS_RES_LN, 0, true>(p, s);
    return launch_one<64, EPI_BIAS_RES_LN, 0, true>(p, s);
}

int launch_gemm(const GemmParams& p, int epi, hipStream_t s) {
    ROHM_ARG_CHECK(p.K > 0 && p.K % BK == 0, "gemm: K=%d must be a positive multiple of %d", p.K, BK);
    ROHM_ARG_CHECK(p.lda % 4 == 0 && p.ldw % 4 == 0, "gemm: lda/ldw must be multiples of 4 floats");
    ROHM_ARG_CHECK(((uintptr_t)p.A % 16) == 0 && ((uintptr_t)p.W % 16) == 0, "gemm: A/W must be 16-byte aligned");
    ROHM_ARG_CHECK(p.M > 0 && p.N > 0, "gemm: empty problem");
    if (p.out_stats) ROHM_ARG_CHECK(epi == EPI_BIAS_RES && p.out_parts == (p.N + 63) / 64 && p.out_parts <= 8 &&
                                        p.out_parts % 2 == 0, "gemm: bad row-stat request");
    ROHM_ARG_CHECK((!p.ln_stats || (p.ln_parts <= 8 && p.ln_parts % 2 == 0)) && (!p.r_stats || (p.r_parts <= 8 && p.r_parts % 2 == 0)),
                   "gemm: LayerNorm folding supports 2..8 statistic slots (64 columns each)");
    if (p.ln_stats) ROHM_ARG_CHECK((epi == EPI_QKV || epi == EPI_BIAS_GELU || epi == EPI_OUT_T) && p.ln_c && p.ln_dim > 0,
                                   "gemm: LayerNorm folding needs ln_c / ln_dim and a supporting epilogue");
    if (p.r_stats) ROHM_ARG_CHECK(epi == EPI_BIAS_RES && p.r_gamma && p.r_beta && p.ln_dim > 0, "gemm: bad residual LN");
    if (p.ksplit > 1) {
        ROHM_ARG_CHECK(epi == EPI_BIAS, "gemm: split-K supports the plain bias epilogue only");
        ROHM_ARG_CHECK(p.partial && p.ld_partial % 4 == 0 && p.ld_partial >= ((p.N + 3) / 4) * 4 &&
                           (((uintptr_t)p.partial) & 15) == 0 && p.ksplit <= p.K / BK,
                       "gemm: bad split-K workspace (ksplit=%d, K chunks=%d)", p.ksplit, p.K / BK);
    }
    if (p.res_col0 > 0)
        ROHM_ARG_CHECK(epi == EPI_BIAS && p.conv_taps > 0 && p.res_col0 % 64 == 0 && p.res_col0 < p.N && p.res_nk > 0 && p.res_k0 % BK == 0 &&
                           p.res_k0 + p.res_nk * BK <= p.K && (p.res_ksplit <= 1 || (p.res_ksplit <= p.res_nk && p.res_ksplit <= p.ksplit)) &&
                           (long)((p.M + BM - 1) / BM) * ((p.N + 127) / 128) < 256,
                       "gemm: bad fused-residual request");
    if (p.conv_taps > 0) {
        ROHM_ARG_CHECK(p.conv_taps <= 5 && p.conv_cin_pad % BK == 0 && p.K == p.conv_taps * p.conv_cin_pad,
                       "gemm: conv gather needs cin_pad %% 32 == 0 and K == taps * cin_pad");
        ROHM_ARG_CHECK(p.zero_page && p.conv_tq > 0 && p.conv_tin > 0 && p.M % p.conv_tq == 0,
                       "gemm: bad conv gather geometry");
        for (int j = 0; j < p.conv_taps; ++j)
            ROHM_ARG_CHECK(p.conv_off[j] == p.conv_off0 + j * p.conv_dstep, "gemm: conv tap offsets must be off0 + j * dstep");
    }
    switch (epi) {
        case EPI_BIAS: return launch_bn<EPI_BIAS>(p, s);
        case EPI_BIAS_GELU: return launch_bn<EPI_BIAS_GELU>(p, s);
        case EPI_BIAS_RES: return launch_bn<EPI_BIAS_RES>(p, s);
        case EPI_QKV: return launch_bn<EPI_QKV>(p, s);
        case EPI_EMBED: return launch_bn<EPI_EMBED>(p, s);
        case EPI_OUT_T: {
            GemmParams q = p;
            q.sk_units = q.sk_tiles8 = 0;
            if (p.sk_part && gemm_sk_plan(p.M, p.N, p.K, &q.sk_units, &q.sk_tiles8)) {
                ROHM_ARG_CHECK(p.sk_flag && p.xln_err && p.xln_pass && al16(p.sk_part) && p.bias, "gemm: stream-K: bad scratch / null bias");
                return launch_t<64, EPI_OUT_T>(q, s);
            }
            return launch_bn<EPI_OUT_T>(q, s);
        }
        case EPI_BIAS_RES_LN: return launch_ln(p, s);
    }
    set_error("gemm: unknown epilogue %d", epi);
    return ROHM_ERR_ARG;
}

}  // namespace rohm
