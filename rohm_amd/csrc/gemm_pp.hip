// Split GEMM on READY-MADE 16-bit planes for gfx950 (opt-in precision ladder):
//   C[M,N] = epi(A[M,K] . W[N,K]^T),  every fp32 product a.w emulated by 16-bit MFMA products of the PLANES of the operands
//   (planes.h), fp32 accumulation.  Same Linears as gemm_f32.hip (model/posenet.py:63-69).  The template parameter NP is the MODE:
//
//   NP = 16 (fp16x3): two fp16 planes h, l' = (x - h) 2^11; three products: a_h w_h -> accumulator, a_h w_l' + a_l' w_h -> a second
//                     accumulator that enters with weight 2^-11 when the tile is handed over (~2^-22 per product)
//   NP = 3  (bf16x6): three bf16 planes; the six plane products of weight >= 2^-16 -- fp32-class accuracy
//   NP = 2  (bf16x3): two bf16 planes, three products (~2^-16 per product)
//
// The row-block layout of wave (wm, wn) in both kernels: row blocks 4 wm .. 4 wm + 3 of its CB column blocks, plus of row block 8
// the column block(s) of half wm -- every wave multiplies 4 CB + CB / 2 blocks per chunk, every SIMD (waves w, w + 4) nine row
// blocks' worth.
//
// Round 2 cut the planes INSIDE the GEMM (fp32 tiles staged through LDS, cut by the VALU, written back to LDS as fragments):
// 221 KB of LDS traffic per K chunk for 1.0 us of MFMA -- the LDS pipe, not the matrix core, set the pace (MFMA busy 0.36).
// Here nobody cuts: the PRODUCERS (LayerNorm, attention, the GELU epilogue of this kernel) write planes of their outputs in
// fragment-major layout, the weights are cut once at rohm_posenet_create, and
//   * the A fragments of a chunk (9 row blocks x NP planes x 1 KiB) go HBM/L2 -> LDS by LDS-DMA, ONE instruction per
//     fragment, contiguous source, linear image, read back with one conflict-free ds_read_b128 per (row block, plane);
//   * the B fragments never touch LDS: a wave loads the planes of its own 32 (16) columns straight into registers
//     (1 KiB contiguous per load), two chunks ahead;
//   -> 27 KB of DMA + 108 KB of fragment reads per chunk at 144 x 128, no VALU work in the loop at all.
//   * 8 waves = two per SIMD on a 144 x (64 | 128) tile, 2 x 4 wave grid (wave (wm, wn): row half wm of column group wn);
//     waves w and w + 4 share a SIMD, so every SIMD carries nine row blocks and the partner issues MFMAs while a wave sits
//     in a DMA issue, an LDS wait or the barrier;
//   * three LDS stages, DMA two chunks ahead, ONE barrier per chunk; the barrier sits between two MFMA groups whose
//     operands are already in registers (the first row block of chunk k is multiplied AFTER the barrier that publishes chunk
//     k + 1), so no LDS latency is exposed behind it; waits are counted (vmcnt(ops of the youngest chunk)).
// Operand order is swapped (weights on the MFMA "A" side): a lane ends with 4 consecutive output columns of one row.  Plane
// output (the GELU GEMM feeding FF2): lanes 16 apart exchange half their packed planes (v_permlane16_swap) so that every lane
// stores one complete 16-byte unit -- 1 KiB contiguous per store instruction.
#include <stdlib.h>
#include <type_traits>
#include "planes.h"

namespace rohm {
namespace {

constexpr int QM = 144, QRB = 9, QNT = 512, QSTAGE = 3;
constexpr int kStreamCUs = 256;      // MI355X: persistent workgroups of the stream kernel

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

// s_waitcnt vmcnt(n) lgkmcnt(0) (gfx9 encoding: vmcnt [3:0] + [15:14], expcnt [6:4] = 7, lgkmcnt [11:8])
#define PP_WAIT_VM_LGKM0(n) __builtin_amdgcn_s_waitcnt(((n) & 15) | (7 << 4) | (0 << 8) | (((n) >> 4) << 14))

template <int NP>
struct Frag { u32x4 p[NP]; };

template <int EPI, int NP, int CB, bool POUT>
__global__ __launch_bounds__(QNT) void gemm_pp_kernel(PlaneGemmParams p) {
    constexpr int NPL = mode_planes(NP);           // planes per operand (NP is the mode: 3 / 2 bf16 planes, 16 = two fp16 planes)
    constexpr bool F16 = (NP == kModeF16);
    constexpr int NPROD = (NP == 3) ? 6 : 3;
    constexpr int ia[6] = {2, 0, 1, 1, 0, 0}, ib[6] = {0, 2, 1, 0, 1, 0};    // plane pairs (activation, weight), smallest terms first
    constexpr int BN = 4 * CB * 16;
    constexpr int NF = QRB * NPL;                  // A fragments (1 KiB) per chunk
    constexpr int PIECES = (NF + 7) / 8;          // LDS-DMA instructions per wave per chunk
    constexpr int VMOPS = PIECES + CB * NPL;       // vector-memory operations per wave per chunk
    constexpr int STAGE_B = NF * 1024;

    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* lds_dummy = smem + QSTAGE * STAGE_B;    // 1 KiB landing zone of the unpopulated DMA pieces

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wn = wave & 3, wm = wave >> 2;
    const int li = lane & 15, lg = lane >> 4;
    // Rows: wave (wm, wn) owns row blocks 4 wm .. 4 wm + 3 of its CB column blocks, and of row block 8 the column block(s)
    // of half wm -- every wave multiplies 4 CB + CB / 2 blocks per chunk, every SIMD (waves w, w + 4) nine row blocks.
    const int r0 = wm * 4;
    const int tiles_n = p.N / BN;
    const int tile = xcd_remap(blockIdx.x, (p.M / QM) * tiles_n);
    const int mblk0 = (tile / tiles_n) * QRB;     // first 16-row block of the tile
    const int n0 = (tile % tiles_n) * BN;
    const int nkc = p.K / 32;
    const size_t chunk_b = (size_t)NPL * 1024;     // bytes of one (row block, chunk) in either operand

    // ---- A: LDS-DMA, fragment f = wave + 8 j of the chunk (row block f / NP, plane f % NP) --------------------------------
    const char* a_src[PIECES];
    int a_dst[PIECES];                            // byte offset inside a stage, -1: landing zone
#pragma unroll
    for (int j = 0; j < PIECES; ++j) {
        const int f = wave + 8 * j;
        const bool ok = f < NF;
        const int rb = ok ? f / NPL : 0, pl = ok ? f % NPL : 0;
        a_src[j] = reinterpret_cast<const char*>(p.Ap) + ((size_t)(mblk0 + rb) * nkc * NPL + pl) * 1024 + lane * 16;
        a_dst[j] = ok ? f * 1024 : -1;
    }
    auto dma_piece = [&](int stage, int kc, int j) {
        char* dst = (a_dst[j] >= 0) ? smem + stage * STAGE_B + a_dst[j] : lds_dummy;
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(a_src[j] + (size_t)kc * chunk_b),
                                         (__attribute__((address_space(3))) void*)dst, 16, 0, 0);
    };
    auto dma = [&](int stage, int kc) {
#pragma unroll
        for (int j = 0; j < PIECES; ++j) dma_piece(stage, kc, j);
    };
    // ---- B: straight into registers, the planes of this wave's CB column blocks ------------------------------------------
    const char* b_src[CB];
#pragma unroll
    for (int c = 0; c < CB; ++c)
        b_src[c] = reinterpret_cast<const char*>(p.Wp) + (size_t)((n0 >> 4) + wn * CB + c) * nkc * chunk_b + lane * 16;
    Frag<NPL> breg[QSTAGE][CB];
    auto bload = [&](Frag<NPL> (&dst)[CB], int kc) {
#pragma unroll
        for (int c = 0; c < CB; ++c)
#pragma unroll
            for (int pl = 0; pl < NPL; ++pl)
                dst[c].p[pl] = *reinterpret_cast<const u32x4*>(b_src[c] + (size_t)kc * chunk_b + pl * 1024);
    };
    auto aread = [&](Frag<NPL>& dst, int stage, int rb) {     // row block rb of the chunk in `stage`
#pragma unroll
        for (int pl = 0; pl < NPL; ++pl)
            dst.p[pl] = *reinterpret_cast<const u32x4*>(smem + stage * STAGE_B + (rb * NPL + pl) * 1024 + lane * 16);
    };

    f32x4 acc[4][CB], acc8 = f32x4{0.f, 0.f, 0.f, 0.f};
    f32x4 accx[4][CB], acc8x = f32x4{0.f, 0.f, 0.f, 0.f};       // cross-term accumulators (fp16 mode only; dead otherwise)
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int c = 0; c < CB; ++c) { acc[j][c] = f32x4{0.f, 0.f, 0.f, 0.f}; accx[j][c] = f32x4{0.f, 0.f, 0.f, 0.f}; }
    // main + 2^-11 cross, times the power-of-two scale the weight planes were cut with
    const float acc_scale = (p.acc_scale != 0.f) ? p.acc_scale : 1.0f;
    auto combine = [&](const f32x4& m, const f32x4& x) {
        f32x4 r;
#pragma unroll
        for (int q = 0; q < 4; ++q) r[q] = F16 ? (m[q] + x[q] * (1.0f / kF16LowScale)) * acc_scale : m[q] * acc_scale;
        return r;
    };
    auto mfma = [](const u32x4& w, const u32x4& a, const f32x4& c) {
        if constexpr (F16) return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, w), __builtin_bit_cast(f16x8, a), c, 0, 0, 0);
        else return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, w), __builtin_bit_cast(bf16x8, a), c, 0, 0, 0);
    };
    // fp16 mode: products 3, 4 of the table (a_l' w_h, a_h w_l') are the cross terms: they go to the second accumulator
    // set, which enters the result with weight 2^-11; product 5 (a_h w_h) goes to the first
    auto is_cross = [](int q) { return F16 && q < 5; };
    // MFMAs [lo, hi) of one row block, product-major (consecutive MFMAs hit different accumulators): t = product * CB + column
    auto mm_range = [&](f32x4 (&d)[CB], f32x4 (&dx)[CB], const Frag<NPL>& a, const Frag<NPL> (&b)[CB], int lo, int hi) {
#pragma unroll
        for (int t = 0; t < NPROD * CB; ++t) {
            if (t < lo || t >= hi) continue;
            const int q = 6 - NPROD + t / CB, c = t % CB;
            if (is_cross(q)) dx[c] = mfma(b[c].p[ib[q]], a.p[ia[q]], dx[c]);
            else d[c] = mfma(b[c].p[ib[q]], a.p[ia[q]], d[c]);
        }
    };
    auto mm = [&](f32x4 (&d)[CB], f32x4 (&dx)[CB], const Frag<NPL>& a, const Frag<NPL> (&b)[CB]) { mm_range(d, dx, a, b, 0, NPROD * CB); };
    // the wave's share of row block 8: column block wm of its two (CB = 2), or the only one for wm = 0 (CB = 1)
    auto b8 = [&](const Frag<NPL> (&b)[CB]) {
        Frag<NPL> r;
#pragma unroll
        for (int pl = 0; pl < NPL; ++pl)
#pragma unroll
            for (int q = 0; q < 4; ++q) r.p[pl][q] = wm ? b[CB - 1].p[pl][q] : b[0].p[pl][q];
        return r;
    };
    // [last full row block | the share of row block 8], CB = 2: three accumulators round-robin, MFMAs [lo, hi) of 3 NPROD
    auto mm_last = [&](f32x4 (&d)[CB], f32x4 (&dx)[CB], const Frag<NPL>& a, const Frag<NPL>& a8, const Frag<NPL> (&b)[CB], const Frag<NPL>& bh,
                       int lo, int hi) {
#pragma unroll
        for (int t = 0; t < 3 * NPROD; ++t) {
            if (t < lo || t >= hi) continue;
            const int q = 6 - NPROD + t / 3, c = t % 3;
            if (c < 2) {
                if (is_cross(q)) dx[c] = mfma(b[c].p[ib[q]], a.p[ia[q]], dx[c]);
                else d[c] = mfma(b[c].p[ib[q]], a.p[ia[q]], d[c]);
            } else {
                if (is_cross(q)) acc8x = mfma(bh.p[ib[q]], a8.p[ia[q]], acc8x);
                else acc8 = mfma(bh.p[ib[q]], a8.p[ia[q]], acc8);
            }
        }
    };

    // ---- prologue: chunks 0 and 1 in flight -----------------------------------------------------------------------------------
    dma(0, 0);
    bload(breg[0], 0);
    const int k1 = (nkc > 1) ? 1 : 0;
    dma(1, k1);
    bload(breg[1], k1);
    PP_WAIT_VM_LGKM0(VMOPS);                      // chunk 0 has landed (mine); the barrier publishes everybody's
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);

    // One K chunk (stage S = kc % 3), entered right after the barrier that published it.  `a0` carries the wave's first row
    // block of the PREVIOUS chunk (read before that barrier): its MFMAs run first, while the fragment reads of this chunk are
    // in flight.  Every section opens with ONE MFMA of its row block, then issues the fragment reads of the next section,
    // then the rest: hipcc waits lgkmcnt(0) at the first use of a fragment while LDS-DMA is pending, and placed like this
    // that wait never has anything to wait for.  DMA pieces stay where the source puts them (side effects), so they are dealt
    // out between MFMA groups by hand; the B loads are placed by sched_group_barrier.
    constexpr int kMfma = 0x008, kVmem = 0x010;
    constexpr int NM = CB * NPROD;                // MFMAs of one full row block
#define PP_SB() __builtin_amdgcn_sched_barrier(0)
    Frag<NPL> a0, a1, a2, a3;
    auto step = [&](auto s_tag, auto first_tag, int kc) {
        constexpr int S = decltype(s_tag)::value;
        constexpr bool FIRST = decltype(first_tag)::value;
        constexpr int SP = (S + 2) % 3;           // stage of chunk kc - 1 = target of chunk kc + 2
        const int kn = (kc + 2 < nkc) ? kc + 2 : nkc - 1;     // the last two chunks re-fetch the last one (uniform counts)
        // -- section 1: [previous chunk, first row block] + DMA of chunk kc + 2
        if constexpr (FIRST) {
            aread(a1, S, r0 + 1);
            dma(SP, kn);
        } else {
            constexpr int G = (NM - 1) / PIECES;  // MFMAs between two DMA pieces
            mm_range(acc[0], accx[0], a0, breg[SP], 0, 1);
            PP_SB();
            aread(a1, S, r0 + 1);
            PP_SB();
#pragma unroll
            for (int i = 0; i < PIECES; ++i) {
                mm_range(acc[0], accx[0], a0, breg[SP], 1 + i * G, 1 + (i + 1) * G);
                PP_SB();
                dma_piece(SP, kn, i);
                PP_SB();
            }
            mm_range(acc[0], accx[0], a0, breg[SP], 1 + PIECES * G, NM);
        }
        PP_SB();
        // -- section 2: row block r0 + 1, B fragments of chunk kc + 2 (their registers were last read in section 1)
        mm_range(acc[1], accx[1], a1, breg[S], 0, 1);
        PP_SB();
        aread(a2, S, r0 + 2);
        PP_SB();
        bload(breg[SP], kn);
        mm_range(acc[1], accx[1], a1, breg[S], 1, NM);
#pragma unroll
        for (int i = 0; i < CB * NPL; ++i) {
            __builtin_amdgcn_sched_group_barrier(kMfma, (NM - 1) / (CB * NPL), 1);
            __builtin_amdgcn_sched_group_barrier(kVmem, 1, 1);
        }
        PP_SB();
        // -- section 3: row block r0 + 2
        mm_range(acc[2], accx[2], a2, breg[S], 0, 1);
        PP_SB();
        aread(a1, S, r0 + 3);
        aread(a3, S, 8);
        PP_SB();
        mm_range(acc[2], accx[2], a2, breg[S], 1, NM);
        PP_SB();
        // -- section 4: row block r0 + 3 and the share of row block 8; the first row block is read for the next step
        if constexpr (CB == 2) {
            const Frag<NPL> bh = b8(breg[S]);
            mm_last(acc[3], accx[3], a1, a3, breg[S], bh, 0, 1);
            PP_SB();
            aread(a0, S, r0);
            PP_SB();
            mm_last(acc[3], accx[3], a1, a3, breg[S], bh, 1, 3 * NPROD);
        } else {
            mm_range(acc[3], accx[3], a1, breg[S], 0, 1);
            PP_SB();
            aread(a0, S, r0);
            PP_SB();
            mm_range(acc[3], accx[3], a1, breg[S], 1, NM);
            if (wm == 0) {          // wave-uniform
#pragma unroll
                for (int q = 6 - NPROD; q < 6; ++q) {
                    if (is_cross(q)) acc8x = mfma(breg[S][0].p[ib[q]], a3.p[ia[q]], acc8x);
                    else acc8 = mfma(breg[S][0].p[ib[q]], a3.p[ia[q]], acc8);
                }
            }
        }
        PP_SB();
        PP_WAIT_VM_LGKM0(VMOPS);                  // chunk kc + 1 has landed; every fragment read of chunk kc has returned
        __builtin_amdgcn_s_barrier();
        PP_SB();
    };
    using I0 = std::integral_constant<int, 0>;
    using I1 = std::integral_constant<int, 1>;
    using I2 = std::integral_constant<int, 2>;
    step(I0{}, std::true_type{}, 0);
    int kc = 1;
    for (; kc + 3 <= nkc; kc += 3) {
        step(I1{}, std::false_type{}, kc);
        step(I2{}, std::false_type{}, kc + 1);
        step(I0{}, std::false_type{}, kc + 2);
    }
    // remainder: 0, 1 or 2 chunks, then the first row block of the last chunk
    const int rem = nkc - kc;
    if (rem >= 1) step(I1{}, std::false_type{}, kc);
    if (rem == 2) step(I2{}, std::false_type{}, kc + 1);
    if (rem == 0) mm(acc[0], accx[0], a0, breg[0]);
    else if (rem == 1) mm(acc[0], accx[0], a0, breg[1]);
    else mm(acc[0], accx[0], a0, breg[2]);
    PP_WAIT_VM_LGKM0(0);                          // the re-fetched tail chunks: nothing may land in LDS after the workgroup ends

    // ---- epilogue: lane holds C[m][nb .. nb + 3] ------------------------------------------------------------------------------
    const int nkc_out = p.N / 32;
    char* const cp = reinterpret_cast<char*>(p.Cp);
    auto finish = [&](f32x4 v, int m, int nb) {
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
        if (p.C) *reinterpret_cast<f32x4*>(p.C + (size_t)m * p.ldc + nb) = v;
        return v;
    };
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int m = (mblk0 + r0 + j) * 16 + li;
        f32x4 v[CB];
#pragma unroll
        for (int c = 0; c < CB; ++c) v[c] = finish(combine(acc[j][c], accx[j][c]), m, n0 + (wn * CB + c) * 16 + lg * 4);
        if constexpr (POUT) {
            if (CB == 2 && !p.no_swap) {
                // lanes (li, lg) and (li, lg ^ 1) trade: the even one ends with columns 4 lg .. 4 lg + 7 of block 0, the odd
                // one with columns 4 (lg - 1) .. + 7 of block 1 -- one complete 16-byte unit per lane and plane
                u32x2 c0[NPL], c1[NPL];
                plane_cut4<NP>(v[0], c0);
                plane_cut4<NP>(v[CB - 1], c1);
                const int col = n0 + wn * 32 + ((lg & 1) ? 16 + (lg - 1) * 4 : lg * 4);
#pragma unroll
                for (int pl = 0; pl < NPL; ++pl) {
                    const u32x2 s0 = __builtin_amdgcn_permlane16_swap(c0[pl][0], c1[pl][0], false, false);
                    const u32x2 s1 = __builtin_amdgcn_permlane16_swap(c0[pl][1], c1[pl][1], false, false);
                    *reinterpret_cast<u32x4*>(cp + plane_unit(m, col >> 3, nkc_out, NPL, pl) * 16) = u32x4{s0[0], s1[0], s0[1], s1[1]};
                }
            } else {
#pragma unroll
                for (int c = 0; c < CB; ++c) plane_store4<NP>(cp, m, n0 + (wn * CB + c) * 16 + lg * 4, nkc_out, v[c]);
            }
        }
    }
    if (CB == 2 || wm == 0) {        // the share of row block 8
        const int m = (mblk0 + 8) * 16 + li;
        const int nb = n0 + (wn * CB + (CB == 2 ? wm : 0)) * 16 + lg * 4;
        const f32x4 v = finish(combine(acc8, acc8x), m, nb);
        if constexpr (POUT) plane_store4<NP>(cp, m, nb, nkc_out, v);
    }
}

// ---- the same GEMM as ONE STREAM OF CHUNKS per workgroup ---------------------------------------------------------------------
// gemm_pp_kernel pays, per tile, a prologue (first chunk from L2 / HBM with nothing to multiply) and an epilogue in which all
// 256 CUs store at once while no matrix core runs: measured 18-30 us of a 43-81 us launch at B = 64 (the K loop itself runs
// at ~1.04 us per chunk).  Here a workgroup is persistent (grid = min(tiles, CUs)): it walks its tiles t = b, b + G, ... as one
// flat sequence of K chunks.  The prefetch cursor (DMA + B loads two chunks ahead) simply runs on into the next tile, so only
// the first tile of a workgroup has a prologue; and when a tile's last chunk has been multiplied its accumulators are handed to
// a second register set (`pend`) and the next tile starts at once -- the stores of the finished tile are DRIPPED out one unit
// (one 16-row block of the wave) per chunk of the next tile, between its MFMAs.  Only the last tile of a workgroup has an
// exposed epilogue.
// What was measured around this kernel in round 3 and did NOT pay (NOTES.md, profiles/r3_f_*): eight waves side by side along N
// (no B fragment loaded twice: 51 instead of 75 KB of operands per chunk) with the A tile by LDS-DMA and, in a second variant, through
// registers + ds_write_b128 -- both 1.39-1.49 us per chunk against 1.30-1.41 here: the loop is bound by the MFMA issue rate under DVFS
// (a bare stream of these MFMAs on random operands runs at 1.7-2.0 GHz, 8.4-10.3 ns per MFMA; this loop needs 12.2-13 ns), not by
// the L2 -> CU path, the LDS-DMA rate or LDS bandwidth.
// Counted waits with stores in the stream: on gfx9 stores share vmcnt with loads; loads return in order among loads, so
// `vmcnt(n)` with n <= (loads issued since the data we need) can never be satisfied while an older load is outstanding,
// whatever order the stores complete in.  Every step waits vmcnt(VMOPS) = the DMA pieces + B loads of that step.
template <int EPI, int NP, int CB, bool POUT>
__global__ __launch_bounds__(QNT) void gemm_pp_stream_kernel(PlaneGemmParams p) {
    constexpr int NPL = mode_planes(NP);           // planes per operand (NP is the mode: 3 / 2 bf16 planes, 16 = two fp16 planes)
    constexpr bool F16 = (NP == kModeF16);
    constexpr int NPROD = (NP == 3) ? 6 : 3;
    constexpr int ia[6] = {2, 0, 1, 1, 0, 0}, ib[6] = {0, 2, 1, 0, 1, 0};
    constexpr int BN = 4 * CB * 16;
    constexpr int NF = QRB * NPL;
    constexpr int PIECES = (NF + 7) / 8;
    constexpr int VMOPS = PIECES + CB * NPL;
    constexpr int STAGE_B = NF * 1024;
    constexpr int NM = CB * NPROD;

    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* lds_dummy = smem + QSTAGE * STAGE_B;
    // the bias vector lives in LDS for the whole launch: an ordinary global load inside the dripped epilogue would make hipcc
    // wait vmcnt(0) at its use and drain the prefetch pipeline in the middle of a step
    float* bias_s = reinterpret_cast<float*>(smem + QSTAGE * STAGE_B + 1024);
    for (int i = threadIdx.x; i < p.N; i += QNT) bias_s[i] = p.bias ? p.bias[i] : 0.f;
    // LayerNorm folding (planes.h): the vectors of the fold next to the bias, then two slots for the row statistics of a tile
    // (the current tile's block is fetched by LDS-DMA in its first step; the pending tile's block is read by the dripped epilogue)
    constexpr bool CONSUMER = (EPI == EPI_QKV || EPI == EPI_BIAS_GELU);
    const float* const stats_src = CONSUMER ? p.ln_stats : (EPI == EPI_BIAS_RES ? p.r_stats : nullptr);
    const bool has_stats = stats_src != nullptr;
    const int parts = has_stats || (EPI == EPI_BIAS_RES && p.out_stats) ? p.ln_dim / 16 : 0;
    float* const vec1_s = bias_s + p.N;              // c_n (consumer) / gamma of the residual's LayerNorm
    float* const vec2_s = vec1_s + p.N;              // beta of the residual's LayerNorm
    char* const stats_s = reinterpret_cast<char*>(vec2_s + p.N);
    const int stats_tile_b = QM * parts * 8;         // bytes of one tile's statistics (144 rows)
    // (mu, rstd) of the 144 rows of a tile, two slots: reduced ONCE per tile by 144 threads in the tile's second step (the partial
    // pairs of a row are 256 B apart: read per unit by every wave they were a 16-way bank conflict eight waves wide -- measured:
    // the QKV GEMM went from 48 to 92 us)
    float* const murs_s = reinterpret_cast<float*>(stats_s + 2 * stats_tile_b);
    if (has_stats) {
        const float* v1 = CONSUMER ? p.ln_c : p.r_gamma;
        for (int i = threadIdx.x; i < p.N; i += QNT) {
            vec1_s[i] = v1[i];
            if (!CONSUMER) vec2_s[i] = p.r_beta[i];
        }
    }

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wn = wave & 3, wm = wave >> 2;
    const int li = lane & 15, lg = lane >> 4;
    const int r0 = wm * 4;
    const int tiles_n = p.N / BN;
    const int ntiles = (p.M / QM) * tiles_n;
    const int G = gridDim.x;
    const int my_tiles = (ntiles - (int)blockIdx.x + G - 1) / G;      // >= 1 (grid <= tiles)
    const int nkc = p.K / 32;
    const int total = my_tiles * nkc;                                 // chunks of this workgroup
    const size_t chunk_b = (size_t)NPL * 1024;

    // tile seq -> (first row block, first column); linear ids of one workgroup keep their XCD (G is a multiple of 8 or == tiles)
    auto tile_coords = [&](int seq, int& mblk0, int& n0) {
        const int t = xcd_remap((int)blockIdx.x + seq * G, ntiles);
        mblk0 = (t / tiles_n) * QRB;
        n0 = (t % tiles_n) * BN;
    };

    // ---- operand addresses: tile-independent per-lane bases + a scalar offset for (tile, chunk) ------------------------------
    const char* a_base[PIECES];
    int a_dst[PIECES];
#pragma unroll
    for (int j = 0; j < PIECES; ++j) {
        const int f = wave + 8 * j;
        const bool ok = f < NF;
        const int rb = ok ? f / NPL : 0, pl = ok ? f % NPL : 0;
        a_base[j] = reinterpret_cast<const char*>(p.Ap) + ((size_t)rb * nkc * NPL + pl) * 1024 + lane * 16;
        a_dst[j] = ok ? f * 1024 : -1;
    }
    const char* b_base[CB];
#pragma unroll
    for (int c = 0; c < CB; ++c)
        b_base[c] = reinterpret_cast<const char*>(p.Wp) + (size_t)(wn * CB + c) * nkc * chunk_b + lane * 16;
    // prefetch cursor: chunk pf_kc of tile pf_seq; offsets of that chunk in A / W
    int pf_seq = 0, pf_kc = 0;
    size_t pf_a = 0, pf_b = 0;
    {
        int mb, n0;
        tile_coords(0, mb, n0);
        pf_a = (size_t)mb * nkc * chunk_b;
        pf_b = (size_t)(n0 >> 4) * nkc * chunk_b;
    }
    auto pf_advance = [&]() {                     // to the next chunk of the stream; past the end it stays on the last chunk
        if (pf_kc + 1 < nkc) {
            ++pf_kc; pf_a += chunk_b; pf_b += chunk_b;
        } else if (pf_seq + 1 < my_tiles) {
            ++pf_seq; pf_kc = 0;
            int mb, n0;
            tile_coords(pf_seq, mb, n0);
            pf_a = (size_t)mb * nkc * chunk_b;
            pf_b = (size_t)(n0 >> 4) * nkc * chunk_b;
        }
    };
    auto dma_piece = [&](int stage, int j) {
        char* dst = (a_dst[j] >= 0) ? smem + stage * STAGE_B + a_dst[j] : lds_dummy;
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(a_base[j] + pf_a),
                                         (__attribute__((address_space(3))) void*)dst, 16, 0, 0);
    };
    auto dma = [&](int stage) {
#pragma unroll
        for (int j = 0; j < PIECES; ++j) dma_piece(stage, j);
    };
    Frag<NPL> breg[QSTAGE][CB];
    auto bload = [&](Frag<NPL> (&dst)[CB]) {
#pragma unroll
        for (int c = 0; c < CB; ++c)
#pragma unroll
            for (int pl = 0; pl < NPL; ++pl)
                dst[c].p[pl] = *reinterpret_cast<const u32x4*>(b_base[c] + pf_b + pl * 1024);
    };
    auto aread = [&](Frag<NPL>& dst, int stage, int rb) {
#pragma unroll
        for (int pl = 0; pl < NPL; ++pl)
            dst.p[pl] = *reinterpret_cast<const u32x4*>(smem + stage * STAGE_B + (rb * NPL + pl) * 1024 + lane * 16);
    };

    f32x4 acc[4][CB], acc8 = f32x4{0.f, 0.f, 0.f, 0.f};
    f32x4 accx[4][CB], acc8x = f32x4{0.f, 0.f, 0.f, 0.f};       // cross-term accumulators (fp16 mode only; dead otherwise)
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int c = 0; c < CB; ++c) { acc[j][c] = f32x4{0.f, 0.f, 0.f, 0.f}; accx[j][c] = f32x4{0.f, 0.f, 0.f, 0.f}; }
    // main + 2^-11 cross, times the power-of-two scale the weight planes were cut with
    const float acc_scale = (p.acc_scale != 0.f) ? p.acc_scale : 1.0f;
    auto combine = [&](const f32x4& m, const f32x4& x) {
        f32x4 r;
#pragma unroll
        for (int q = 0; q < 4; ++q) r[q] = F16 ? (m[q] + x[q] * (1.0f / kF16LowScale)) * acc_scale : m[q] * acc_scale;
        return r;
    };
    auto mfma = [](const u32x4& w, const u32x4& a, const f32x4& c) {
        if constexpr (F16) return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, w), __builtin_bit_cast(f16x8, a), c, 0, 0, 0);
        else return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, w), __builtin_bit_cast(bf16x8, a), c, 0, 0, 0);
    };
    // fp16 mode: products 3, 4 of the table (a_l' w_h, a_h w_l') are the cross terms: they go to the second accumulator
    // set, which enters the result with weight 2^-11; product 5 (a_h w_h) goes to the first
    auto is_cross = [](int q) { return F16 && q < 5; };
    auto mm_range = [&](f32x4 (&d)[CB], f32x4 (&dx)[CB], const Frag<NPL>& a, const Frag<NPL> (&b)[CB], int lo, int hi) {
#pragma unroll
        for (int t = 0; t < NPROD * CB; ++t) {
            if (t < lo || t >= hi) continue;
            const int q = 6 - NPROD + t / CB, c = t % CB;
            if (is_cross(q)) dx[c] = mfma(b[c].p[ib[q]], a.p[ia[q]], dx[c]);
            else d[c] = mfma(b[c].p[ib[q]], a.p[ia[q]], d[c]);
        }
    };
    auto b8 = [&](const Frag<NPL> (&b)[CB]) {
        Frag<NPL> r;
#pragma unroll
        for (int pl = 0; pl < NPL; ++pl)
#pragma unroll
            for (int q = 0; q < 4; ++q) r.p[pl][q] = wm ? b[CB - 1].p[pl][q] : b[0].p[pl][q];
        return r;
    };
    auto mm_last = [&](f32x4 (&d)[CB], f32x4 (&dx)[CB], const Frag<NPL>& a, const Frag<NPL>& a8, const Frag<NPL> (&b)[CB], const Frag<NPL>& bh,
                       int lo, int hi) {
#pragma unroll
        for (int t = 0; t < 3 * NPROD; ++t) {
            if (t < lo || t >= hi) continue;
            const int q = 6 - NPROD + t / 3, c = t % 3;
            if (c < 2) {
                if (is_cross(q)) dx[c] = mfma(b[c].p[ib[q]], a.p[ia[q]], dx[c]);
                else d[c] = mfma(b[c].p[ib[q]], a.p[ia[q]], d[c]);
            } else {
                if (is_cross(q)) acc8x = mfma(bh.p[ib[q]], a8.p[ia[q]], acc8x);
                else acc8 = mfma(bh.p[ib[q]], a8.p[ia[q]], acc8);
            }
        }
    };

    // ---- epilogue of the PENDING tile (pm0 / pn0), one unit per step.  Units queue in the order they complete: row blocks
    // r0 + 1, r0 + 2, r0 + 3, the share of row block 8, row block r0 (its last MFMAs run one step later).  `pendq[0]` is always the
    // unit that goes out next; after a store the queue moves up by one (register moves: the arms of a switch over five statically
    // named units would put five copies of the epilogue code into every step).
    int pm0 = 0, pn0 = 0;                 // pending tile: first row block, first column
    int drip = 5;                         // units stored so far of the pending tile (5 = nothing pending)
    const int nkc_out = p.N / 32;
    char* const cp = reinterpret_cast<char*>(p.Cp);
    f32x4 pendq[5][CB];
#pragma unroll
    for (int u = 0; u < 5; ++u)
#pragma unroll
        for (int c = 0; c < CB; ++c) pendq[u][c] = f32x4{0.f, 0.f, 0.f, 0.f};
    f32x4 rpre[CB];                       // residual of the unit about to be stored (EPI_BIAS_RES), loaded at the top of the step
#pragma unroll
    for (int c = 0; c < CB; ++c) rpre[c] = f32x4{0.f, 0.f, 0.f, 0.f};
    // position d in the queue order -> (row of this lane, first column of column block c, is the unit's block c real?)
    auto unit_geom = [&](int d, int c, int& m, int& nb, bool& real) __attribute__((always_inline)) {
        const bool blk8 = (d == 3);
        const int rb = blk8 ? 8 : r0 + (d == 4 ? 0 : d + 1);
        m = (pm0 + rb) * 16 + li;
        const int cb = blk8 ? (CB == 2 ? wm : 0) : c;
        nb = pn0 + (wn * CB + cb) * 16 + lg * 4;
        real = !blk8 || (c == 0 && (CB == 2 || wm == 0));
    };
    auto load_residual = [&](int d) __attribute__((always_inline)) {
        if constexpr (EPI == EPI_BIAS_RES) {
            if (d < 5) {
#pragma unroll
                for (int c = 0; c < CB; ++c) {
                    int m, nb;
                    bool real;
                    unit_geom(d, c, m, nb, real);
                    if (real) rpre[c] = *reinterpret_cast<const f32x4*>(p.R + (size_t)m * p.ldr + nb);
                }
            }
        }
    };
    auto finish = [&](f32x4 v, int m, int nb, f32x4 rr, float mu, float rstd) __attribute__((always_inline)) {
        {
            const f32x4 b4 = *reinterpret_cast<const f32x4*>(bias_s + nb);
            if (CONSUMER && has_stats) {       // LN(x) W^T + b = (acc - mu c_n) rstd + d_n
                const f32x4 c4 = *reinterpret_cast<const f32x4*>(vec1_s + nb);
#pragma unroll
                for (int q = 0; q < 4; ++q) v[q] = (v[q] - mu * c4[q]) * rstd + b4[q];
            } else {
#pragma unroll
                for (int q = 0; q < 4; ++q) v[q] += b4[q];
            }
        }
        if constexpr (EPI == EPI_BIAS_GELU) {
#pragma unroll
            for (int q = 0; q < 4; ++q) v[q] = gelu_erf(v[q]);
        }
        if constexpr (EPI == EPI_BIAS_RES) {
            if (has_stats) {                   // the residual is LN(raw): normalise it on the fly
                const f32x4 g4 = *reinterpret_cast<const f32x4*>(vec1_s + nb), be4 = *reinterpret_cast<const f32x4*>(vec2_s + nb);
#pragma unroll
                for (int q = 0; q < 4; ++q) rr[q] = (rr[q] - mu) * rstd * g4[q] + be4[q];
            }
#pragma unroll
            for (int q = 0; q < 4; ++q) v[q] += rr[q];
        }
        if constexpr (EPI == EPI_QKV) {
            if (nb < p.qcols) {
#pragma unroll
                for (int q = 0; q < 4; ++q) v[q] *= p.qscale;
            }
        }
        return v;
    };
    // store the front unit (queue position d = drip) and move the queue up
    int pslot = 0;                        // statistics slot of the pending tile
    auto store_front = [&](int d) __attribute__((always_inline)) {
        f32x4 v[CB];
        int m = 0, nb0 = 0;
        bool real0 = false;
        float mu = 0.f, rstd = 1.f;
        if (has_stats) {                  // this lane's row of the pending tile
            const int rb = (d == 3) ? 8 : r0 + (d == 4 ? 0 : d + 1);
            const float* mr = murs_s + (pslot * QM + rb * 16 + li) * 2;
            mu = mr[0];
            rstd = mr[1];
        }
#pragma unroll
        for (int c = 0; c < CB; ++c) {
            int mm_, nb;
            bool real;
            unit_geom(d, c, mm_, nb, real);
            if (c == 0) { m = mm_; nb0 = nb; real0 = real; }
            v[c] = finish(pendq[0][c], mm_, nb, rpre[c], mu, rstd);
            if (real && p.C) *reinterpret_cast<f32x4*>(p.C + (size_t)mm_ * p.ldc + nb) = v[c];
            if constexpr (EPI == EPI_BIAS_RES) {
                if (p.out_stats) {        // partial (sum, sum of squares) of this row over the 16 columns of block c
                    const float su = (v[c][0] + v[c][1]) + (v[c][2] + v[c][3]);
                    const float sq = (v[c][0] * v[c][0] + v[c][1] * v[c][1]) + (v[c][2] * v[c][2] + v[c][3] * v[c][3]);
                    // the four lanes of a row (16 apart) summed on the VALU (two lane swaps; __shfl_xor goes through the LDS crossbar
                    // and its lgkmcnt wait also waits for the ring's LDS-DMA): lane rows 0 / 2 end with the sum, 1 / 3 with the squares
                    const u32x2 s16 = __builtin_amdgcn_permlane16_swap(__float_as_uint(su), __float_as_uint(sq), false, false);
                    const float t = __uint_as_float(s16[0]) + __uint_as_float(s16[1]);
                    const u32x2 s32 = __builtin_amdgcn_permlane32_swap(__float_as_uint(t), __float_as_uint(t), false, false);
                    const float tot = __uint_as_float(s32[0]) + __uint_as_float(s32[1]);
                    if (real && (lg >> 1) == c)      // [row block][part][row in block][2]: the sixteen rows of a unit share a 128-byte line
                        p.out_stats[((((size_t)(mm_ >> 4) * parts + (nb >> 4)) * 16 + li) << 1) + (lg & 1)] = tot;
                }
            }
        }
        if constexpr (POUT) {
            if (d != 3 && CB == 2 && !p.no_swap) {
                u32x2 c0[NPL], c1[NPL];
                plane_cut4<NP>(v[0], c0);
                plane_cut4<NP>(v[CB - 1], c1);
                const int col = pn0 + wn * 32 + ((lg & 1) ? 16 + (lg - 1) * 4 : lg * 4);
#pragma unroll
                for (int pl = 0; pl < NPL; ++pl) {
                    const u32x2 s0 = __builtin_amdgcn_permlane16_swap(c0[pl][0], c1[pl][0], false, false);
                    const u32x2 s1 = __builtin_amdgcn_permlane16_swap(c0[pl][1], c1[pl][1], false, false);
                    *reinterpret_cast<u32x4*>(cp + plane_unit(m, col >> 3, nkc_out, NPL, pl) * 16) = u32x4{s0[0], s1[0], s0[1], s1[1]};
                }
            } else if (d != 3) {
#pragma unroll
                for (int c = 0; c < CB; ++c) plane_store4<NP>(cp, m, nb0 + c * 16, nkc_out, v[c]);
            } else if (real0) {
                plane_store4<NP>(cp, m, nb0, nkc_out, v[0]);
            }
        }
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
            for (int c = 0; c < CB; ++c) pendq[u][c] = pendq[u + 1][c];
    };

    // The two row halves run two instances of the whole stream (EARLY = true / false), chosen once per wave: inside the steps
    // nothing depends on wm any more, so every step stays straight-line code (a run-time test there makes hipcc merge the
    // wait-count state of the two paths and drain vmcnt(0) in the loop).  Both instances execute the same barriers.
    auto run = [&](auto early_tag) {
        constexpr bool EARLY = decltype(early_tag)::value;
        // ---- prologue of the stream: chunks 0 and 1 in flight ---------------------------------------------------------------------------
        dma(0);
        bload(breg[0]);
        pf_advance();
        dma(1);
        bload(breg[1]);
        pf_advance();
        PP_WAIT_VM_LGKM0(VMOPS);
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);

        constexpr int kMfma = 0x008, kVmem = 0x010;
        Frag<NPL> a0, a1, a2, a3;
        int kc = 0;                           // chunk of the current tile being multiplied
        int cm0 = 0, cn0 = 0, cseq = 0;       // current tile
        tile_coords(0, cm0, cn0);
        auto step = [&](auto s_tag, auto first_tag) {
            constexpr int S = decltype(s_tag)::value;
            constexpr bool FIRST = decltype(first_tag)::value;
            constexpr int SP = (S + 2) % 3;
            const bool tile_first = (kc == 0), tile_last = (kc + 1 == nkc);
            if (kc == 2 && has_stats && threadIdx.x < 2 * QM) {
                // the block requested at the end of the tile's first step has landed (the closing wait of the second step covers it:
                // everything older than that step's own VMOPS prefetches) and is visible (its barrier).  Two threads per row (the even
                // and the odd partial pairs: 128 bytes apart, different banks) sum in index order and are combined over the lane
                // pair; biased variance, eps inside the root (nn.LayerNorm).  [row block][part][row][2]: consecutive rows are
                // consecutive pairs.  The chain of dependent LDS reads is what this costs (about 0.5 us per tile), hence two threads.
                const int row = (int)threadIdx.x >> 1, half = (int)threadIdx.x & 1;
                const f32x2* sp = reinterpret_cast<const f32x2*>(stats_s + (cseq & 1) * stats_tile_b) +
                                  ((row >> 4) * parts + half) * 16 + (row & 15);
                float su = 0.f, sq = 0.f;
                for (int i = 0; i < parts; i += 4) {          // parts % 8 == 0 (launcher); two reads in flight (more spill: 256 VGPRs)
                    const f32x2 t0 = sp[i * 16], t1 = sp[(i + 2) * 16];
                    su += t0[0] + t1[0];
                    sq += t0[1] + t1[1];
                }
                su += __uint_as_float(__builtin_amdgcn_update_dpp(0u, __float_as_uint(su), 0xB1, 0xf, 0xf, true));     // lane ^ 1
                sq += __uint_as_float(__builtin_amdgcn_update_dpp(0u, __float_as_uint(sq), 0xB1, 0xf, 0xf, true));
                const float inv = 1.0f / (float)p.ln_dim;
                const float mu_ = su * inv;
                if (half == 0) {
                    float* mr = murs_s + ((cseq & 1) * QM + row) * 2;
                    mr[0] = mu_;
                    mr[1] = 1.0f / sqrtf(fmaxf(sq * inv - mu_ * mu_, 0.f) + p.ln_eps);
                }
            }
            if constexpr (!FIRST) load_residual(drip);
            // The two waves of a SIMD (row halves wm = 0 / 1) leave every barrier together and run the same code: without help they
            // would issue their DMA pieces and B loads at the same moment and the matrix core would idle behind both.  So the
            // halves are shifted against each other: wm = 0 issues its DMA in section 1 and its B loads in section 2, wm = 1 in
            // sections 3 and 4 -- while one wave of the SIMD feeds the memory pipeline the other one feeds the matrix core.
            // -- section 1: [previous chunk, first row block] (+ DMA of chunk g + 2)
            if constexpr (FIRST) {
                aread(a1, S, r0 + 1);
                dma(SP);
            } else {
                constexpr int GQ = (NM - 1) / PIECES;
                mm_range(acc[0], accx[0], a0, breg[SP], 0, 1);
                PP_SB();
                aread(a1, S, r0 + 1);
                PP_SB();
                if constexpr (EARLY) {
    #pragma unroll
                    for (int i = 0; i < PIECES; ++i) {
                        mm_range(acc[0], accx[0], a0, breg[SP], 1 + i * GQ, 1 + (i + 1) * GQ);
                        PP_SB();
                        dma_piece(SP, i);
                        PP_SB();
                    }
                    mm_range(acc[0], accx[0], a0, breg[SP], 1 + PIECES * GQ, NM);
                } else {
                    mm_range(acc[0], accx[0], a0, breg[SP], 1, NM);
                }
                if (tile_first) {             // that was the last chunk of the previous tile: its first row block is complete now
    #pragma unroll                        // (no unit of that tile has gone out yet: the queue's last slot is still position 4)
                    for (int c = 0; c < CB; ++c) {
                        pendq[4][c] = combine(acc[0][c], accx[0][c]);
                        acc[0][c] = f32x4{0.f, 0.f, 0.f, 0.f}; accx[0][c] = f32x4{0.f, 0.f, 0.f, 0.f};
                    }
                }
            }
            PP_SB();
            // -- section 2: row block r0 + 1 (+ B fragments of chunk g + 2: their registers were last read in section 1)
            mm_range(acc[1], accx[1], a1, breg[S], 0, 1);
            PP_SB();
            aread(a2, S, r0 + 2);
            PP_SB();
            if constexpr (FIRST || EARLY) {
                bload(breg[SP]);
                mm_range(acc[1], accx[1], a1, breg[S], 1, NM);
    #pragma unroll
                for (int i = 0; i < CB * NPL; ++i) {
                    __builtin_amdgcn_sched_group_barrier(kMfma, (NM - 1) / (CB * NPL), 1);
                    __builtin_amdgcn_sched_group_barrier(kVmem, 1, 1);
                }
            } else {
                mm_range(acc[1], accx[1], a1, breg[S], 1, NM);
            }
            PP_SB();
            // -- section 3: row block r0 + 2 (+ DMA, second row half)
            mm_range(acc[2], accx[2], a2, breg[S], 0, 1);
            PP_SB();
            aread(a1, S, r0 + 3);
            aread(a3, S, 8);
            PP_SB();
            if constexpr (!FIRST && !EARLY) {
                constexpr int GQ = (NM - 1) / PIECES;
    #pragma unroll
                for (int i = 0; i < PIECES; ++i) {
                    mm_range(acc[2], accx[2], a2, breg[S], 1 + i * GQ, 1 + (i + 1) * GQ);
                    PP_SB();
                    dma_piece(SP, i);
                    PP_SB();
                }
                mm_range(acc[2], accx[2], a2, breg[S], 1 + PIECES * GQ, NM);
            } else {
                mm_range(acc[2], accx[2], a2, breg[S], 1, NM);
            }
            PP_SB();
            // -- section 4: row block r0 + 3 and the share of row block 8 (+ B loads, second row half); the first row block is read
            //    for the next step
            if constexpr (CB == 2) {
                const Frag<NPL> bh = b8(breg[S]);
                mm_last(acc[3], accx[3], a1, a3, breg[S], bh, 0, 1);
                PP_SB();
                aread(a0, S, r0);
                PP_SB();
                if constexpr (!FIRST && !EARLY) {
                    bload(breg[SP]);
                    mm_last(acc[3], accx[3], a1, a3, breg[S], bh, 1, 3 * NPROD);
    #pragma unroll
                    for (int i = 0; i < CB * NPL; ++i) {
                        __builtin_amdgcn_sched_group_barrier(kMfma, (3 * NPROD - 1) / (CB * NPL), 3);
                        __builtin_amdgcn_sched_group_barrier(kVmem, 1, 3);
                    }
                } else {
                    mm_last(acc[3], accx[3], a1, a3, breg[S], bh, 1, 3 * NPROD);
                }
            } else {
                mm_range(acc[3], accx[3], a1, breg[S], 0, 1);
                PP_SB();
                aread(a0, S, r0);
                PP_SB();
                if constexpr (!FIRST && !EARLY) bload(breg[SP]);
                mm_range(acc[3], accx[3], a1, breg[S], 1, NM);
                if (wm == 0) {
    #pragma unroll
                    for (int q = 6 - NPROD; q < 6; ++q) {
                    if (is_cross(q)) acc8x = mfma(breg[S][0].p[ib[q]], a3.p[ia[q]], acc8x);
                    else acc8 = mfma(breg[S][0].p[ib[q]], a3.p[ia[q]], acc8);
                }
                }
            }
            PP_SB();
            pf_advance();
            // -- one unit of the pending tile goes out
            if constexpr (!FIRST) {
                if (drip < 5) {
                    // the residual of this unit was requested at the top of the step, before the step's VMOPS prefetch operations
                    if constexpr (EPI == EPI_BIAS_RES) PP_WAIT_VM_LGKM0(VMOPS);
                    store_front(drip);
                    ++drip;
                }
            }
            // The statistics of this tile's 144 rows -> slot cseq & 1 (1 KiB pieces).  Requested LAST in the tile's first step, behind
            // the step's own prefetches: the closing wait below (all but the newest VMOPS operations) then leaves them in flight for
            // one more step -- requested at the top of the step they had to land within it, a whole HBM latency on the critical
            // path of every tile (all of it for the one-tile-per-workgroup GEMMs).
            if (tile_first && has_stats) {
                const char* src = reinterpret_cast<const char*>(stats_src) + (size_t)cm0 * 16 * parts * 8 + lane * 16;
                for (int f = wave; f * 1024 < stats_tile_b; f += 8)
                    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + f * 1024),
                                                     (__attribute__((address_space(3))) void*)(stats_s + (cseq & 1) * stats_tile_b + f * 1024),
                                                     16, 0, 0);
            }
            if (tile_last) {
                // hand the finished blocks over (the first row block follows after section 1 of the next step).  The launcher
                // guarantees nkc >= 6, so the five units of the tile before have gone out by now.
    #pragma unroll
                for (int j = 1; j < 4; ++j)
    #pragma unroll
                    for (int c = 0; c < CB; ++c) {
                        pendq[j - 1][c] = combine(acc[j][c], accx[j][c]);
                        acc[j][c] = f32x4{0.f, 0.f, 0.f, 0.f}; accx[j][c] = f32x4{0.f, 0.f, 0.f, 0.f};
                    }
                pendq[3][0] = combine(acc8, acc8x); acc8 = f32x4{0.f, 0.f, 0.f, 0.f}; acc8x = f32x4{0.f, 0.f, 0.f, 0.f};
                pm0 = cm0; pn0 = cn0; drip = 0; pslot = cseq & 1;
                kc = 0;
                if (cseq + 1 < my_tiles) { ++cseq; tile_coords(cseq, cm0, cn0); }
            } else {
                ++kc;
            }
            PP_SB();
            PP_WAIT_VM_LGKM0(VMOPS);
            __builtin_amdgcn_s_barrier();
            PP_SB();
        };
        using I0 = std::integral_constant<int, 0>;
        using I1 = std::integral_constant<int, 1>;
        using I2 = std::integral_constant<int, 2>;
        step(I0{}, std::true_type{});
        int g = 1;
        for (; g + 3 <= total; g += 3) {              // straight-line body: with steps under conditions inside the loop hipcc's wait
            step(I1{}, std::false_type{});            // bookkeeping merges the paths and drains vmcnt(0) at the loop header
            step(I2{}, std::false_type{});
            step(I0{}, std::false_type{});
        }
        if (total - g >= 1) step(I1{}, std::false_type{});
        if (total - g == 2) step(I2{}, std::false_type{});
        const int last = (total - 1) % 3;             // stage of the last chunk: its first row block is still to be multiplied
        if (last == 0) mm_range(acc[0], accx[0], a0, breg[0], 0, NM);
        else if (last == 1) mm_range(acc[0], accx[0], a0, breg[1], 0, NM);
        else mm_range(acc[0], accx[0], a0, breg[2], 0, NM);
        PP_WAIT_VM_LGKM0(0);                          // the re-fetched tail chunks: nothing may land in LDS after the workgroup ends
        // the last tile was handed to the queue by its last step, except for the first row block
    #pragma unroll
        for (int c = 0; c < CB; ++c) pendq[4][c] = combine(acc[0][c], accx[0][c]);      // drip == 0 here: the tile ended with the last step
        // The flush of the last tile (for the one-tile-per-workgroup GEMMs: the whole epilogue).  All five units' residuals are
        // requested before the first unit is finished: as a loop of (load, store) pairs every unit waited vmcnt(0) -- for its own
        // residual AND for the stores of the unit before -- five dependent round trips to memory at the end of every workgroup.
        if constexpr (EPI == EPI_BIAS_RES) {
            f32x4 rq[5][CB];
    #pragma unroll
            for (int d = 0; d < 5; ++d) {
                load_residual(d);
    #pragma unroll
                for (int c = 0; c < CB; ++c) rq[d][c] = rpre[c];
            }
    #pragma unroll
            for (int d = 0; d < 5; ++d) {
    #pragma unroll
                for (int c = 0; c < CB; ++c) rpre[c] = rq[d][c];
                store_front(d);
            }
        } else {
    #pragma unroll
            for (int d = 0; d < 5; ++d) store_front(d);
        }
        drip = 5;
    };
    if (wm == 0) run(std::true_type{});
    else run(std::false_type{});
}

// X[rows][K] fp32 -> planes; one thread per 16-byte unit, consecutive threads = consecutive lanes of a fragment
template <int NP>
__global__ __launch_bounds__(256) void plane_split_kernel(const float* __restrict__ X, int ld, int rows, int K, float scale,
                                                          char* __restrict__ out) {
    const size_t u = (size_t)blockIdx.x * 256 + threadIdx.x;
    const int nkc = K / 32;
    const size_t units = (size_t)(rows / 16) * nkc * 64;
    if (u >= units) return;
    const int lane = (int)(u & 63);
    const size_t blk = u >> 6;
    const int kc = (int)(blk % nkc), rb = (int)(blk / nkc);
    const int row = rb * 16 + (lane & 15), kg = kc * 4 + (lane >> 4);
    const float* src = X + (size_t)row * ld + kg * 8;
    f32x4 lo = *reinterpret_cast<const f32x4*>(src), hi = *reinterpret_cast<const f32x4*>(src + 4);
#pragma unroll
    for (int q = 0; q < 4; ++q) { lo[q] *= scale; hi[q] *= scale; }      // a power of two: exact
    plane_store8<NP>(out, row, kg, nkc, lo, hi);
}

inline bool al16(const void* q) { return (((uintptr_t)q) & 15) == 0; }

static bool pp_stream_enabled() {
    static const bool on = [] {
        const char* e = getenv("ROHM_PP_STREAM");      // diagnostic A/B switch: 0 = one workgroup per tile (gemm_pp_kernel)
        return !(e && e[0] == '0');
    }();
    return on;
}

template <int EPI, int NP, int CB, bool POUT>
int launch_pp(const PlaneGemmParams& p, hipStream_t s) {
    constexpr int BN = 4 * CB * 16;
    const int tiles = (p.M / QM) * (p.N / BN);
    const bool fold = p.ln_stats || p.r_stats || p.out_stats;     // LayerNorm folding: stream kernel only (launch_gemm_pp checks)
    const size_t lds = (size_t)QSTAGE * QRB * mode_planes(NP) * 1024 + 1024 + (size_t)3 * p.N * sizeof(float) +
                       ((p.ln_stats || p.r_stats) ? (size_t)2 * QM * (p.ln_dim / 16) * 8 + 2 * QM * 2 * sizeof(float) : 0);
    const bool stream = pp_stream_enabled() && p.K / 32 >= 6;     // the dripped epilogue needs five steps of the next tile
    if (fold && (!stream || lds > 160 * 1024)) {
        set_error("gemm_pp: LayerNorm folding needs the stream kernel (K >= 192) and %zu bytes of LDS (<= 160 KiB)", lds);
        return ROHM_ERR_UNSUPPORTED;
    }
    static bool attr_set[64] = {};
    int dev = 0;
    ROHM_HIP_CHECK(hipGetDevice(&dev));
    if (dev < 64 && !attr_set[dev]) {
        const int lds_max = QSTAGE * QRB * mode_planes(NP) * 1024 + 1024 + 3 * 16384;      // + bias / fold vectors (N <= 4096, checked at entry)
        ROHM_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_pp_kernel<EPI, NP, CB, POUT>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, lds_max));
        ROHM_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_pp_stream_kernel<EPI, NP, CB, POUT>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));

        attr_set[dev] = true;
    }
    static const char* const kNames[] = {"gemm_bias", "gemm_bias_gelu", "gemm_bias_res", "gemm_qkv"};
    static const char* const kNames64[] = {"gemm_bias/64", "gemm_bias_gelu/64", "gemm_bias_res/64", "gemm_qkv/64"};
    prof::Scope ps(CB == 1 ? kNames64[EPI] : kNames[EPI], 2.0 * p.M * p.N * p.K,
                   2.0 * mode_planes(NP) * ((double)p.M * p.K + (double)p.N * p.K) + (p.C ? 4.0 : 0.0) * p.M * p.N +
                       (p.Cp ? 2.0 * mode_planes(NP) : 0.0) * p.M * p.N, s);
    const int grid = tiles < kStreamCUs ? tiles : kStreamCUs;      // persistent: one workgroup per CU (a multiple of 8: a
                                                                   // workgroup's tiles stay on its XCD), or one per tile if fewer
    if (stream) {
        hipLaunchKernelGGL((gemm_pp_stream_kernel<EPI, NP, CB, POUT>), dim3(grid), dim3(QNT), lds, s, p);
    } else {
        hipLaunchKernelGGL((gemm_pp_kernel<EPI, NP, CB, POUT>), dim3(tiles), dim3(QNT), lds, s, p);
    }
    ROHM_LAUNCH_CHECK();
    return ROHM_OK;
}

template <int EPI, int NP>
int launch_pp_shape(const PlaneGemmParams& p, hipStream_t s) {
    // 144 x 128 tiles while they give (nearly) every CU a tile, 144 x 64 below that (the per-GPU batch of 32 clips and less)
    const bool wide = (p.N % 128 == 0) && (long)(p.M / QM) * (p.N / 128) >= 192;
    if (p.Cp) return wide ? launch_pp<EPI, NP, 2, true>(p, s) : launch_pp<EPI, NP, 1, true>(p, s);
    return wide ? launch_pp<EPI, NP, 2, false>(p, s) : launch_pp<EPI, NP, 1, false>(p, s);
}

template <int NP>
int launch_pp_epi(const PlaneGemmParams& p, int epi, hipStream_t s) {
    switch (epi) {
        case EPI_BIAS: return launch_pp_shape<EPI_BIAS, NP>(p, s);
        case EPI_BIAS_GELU: return launch_pp_shape<EPI_BIAS_GELU, NP>(p, s);
        case EPI_BIAS_RES: return launch_pp_shape<EPI_BIAS_RES, NP>(p, s);
        case EPI_QKV: return launch_pp_shape<EPI_QKV, NP>(p, s);
    }
    set_error("gemm_pp: unsupported epilogue %d", epi);
    return ROHM_ERR_UNSUPPORTED;
}

}  // namespace

int launch_gemm_pp(const PlaneGemmParams& p, int epi, int nplane, hipStream_t s) {
    ROHM_ARG_CHECK(mode_ok(nplane), "gemm_pp: mode must be 3 (bf16x6), 2 (bf16x3) or 16 (fp16x3), got %d", nplane);
    ROHM_ARG_CHECK(p.Ap && p.Wp && (p.C || p.Cp), "gemm_pp: null operand / no output");
    ROHM_ARG_CHECK(p.N <= 4096, "gemm_pp: N = %d exceeds the 4096 columns the kernel keeps a bias copy for", p.N);
    ROHM_ARG_CHECK(p.M > 0 && p.M % QM == 0 && p.N > 0 && p.N % 64 == 0 && p.K > 0 && p.K % 32 == 0,
                   "gemm_pp: %d x %d x %d is not made of whole 144 x 64 tiles / 32-wide K chunks", p.M, p.N, p.K);
    ROHM_ARG_CHECK(al16(p.Ap) && al16(p.Wp) && al16(p.C) && al16(p.Cp) && al16(p.bias) && p.ldc % 4 == 0,
                   "gemm_pp: operands must be 16-byte aligned");
    if (epi == EPI_BIAS_RES) ROHM_ARG_CHECK(p.R && al16(p.R) && p.ldr % 4 == 0, "gemm_pp: bad residual");
    if (epi == EPI_QKV) ROHM_ARG_CHECK(p.qcols % 4 == 0, "gemm_pp: qcols must be a multiple of 4");
    if (p.ln_stats) ROHM_ARG_CHECK((epi == EPI_QKV || epi == EPI_BIAS_GELU) && p.ln_c && p.bias && p.ln_dim == p.K && p.K % 32 == 0,
                                   "gemm_pp: a LayerNorm-consuming GEMM needs epilogue 1 / 3, ln_c, bias (= d) and ln_dim == K");
    if (p.r_stats) ROHM_ARG_CHECK(epi == EPI_BIAS_RES && p.r_gamma && p.r_beta && p.ln_dim == p.N, "gemm_pp: bad residual LayerNorm");
    if (p.out_stats) ROHM_ARG_CHECK(epi == EPI_BIAS_RES && p.ln_dim == p.N && p.N % 128 == 0, "gemm_pp: row statistics need epilogue 2, ln_dim == N");
    if (p.ln_stats || p.r_stats || p.out_stats)      // the statistics of a tile move as whole 1 KiB pieces (144 x ln_dim / 16 x 8 bytes)
        ROHM_ARG_CHECK(p.ln_dim > 0 && p.ln_dim % 128 == 0 && p.ln_eps > 0.f, "gemm_pp: LayerNorm folding needs ln_dim %% 128 == 0 (got %d) and eps > 0", p.ln_dim);
    if (nplane == 3) return launch_pp_epi<3>(p, epi, s);
    if (nplane == 2) return launch_pp_epi<2>(p, epi, s);
    return launch_pp_epi<kModeF16>(p, epi, s);
}

int launch_plane_split(const float* X, int ld, int rows, int K, int nplane, float scale, void* out, hipStream_t s) {
    ROHM_ARG_CHECK(X && out && rows > 0 && rows % 16 == 0 && K > 0 && K % 32 == 0 && ld % 4 == 0 && al16(X) && al16(out),
                   "plane_split: rows %% 16, K %% 32, 16-byte aligned operands required (rows=%d, K=%d)", rows, K);
    ROHM_ARG_CHECK(mode_ok(nplane), "plane_split: mode must be 3, 2 or 16 (got %d)", nplane);
    const size_t units = (size_t)(rows / 16) * (K / 32) * 64;
    prof::Scope ps("plane_split", 0.0, (4.0 + 2.0 * mode_planes(nplane)) * rows * K, s);
    const dim3 grid((unsigned)((units + 255) / 256));
    if (nplane == 3) hipLaunchKernelGGL(plane_split_kernel<3>, grid, dim3(256), 0, s, X, ld, rows, K, scale, (char*)out);
    else if (nplane == 2) hipLaunchKernelGGL(plane_split_kernel<2>, grid, dim3(256), 0, s, X, ld, rows, K, scale, (char*)out);
    else hipLaunchKernelGGL(plane_split_kernel<kModeF16>, grid, dim3(256), 0, s, X, ld, rows, K, scale, (char*)out);
    ROHM_LAUNCH_CHECK();
    return ROHM_OK;
}

}  // namespace rohm
