// fp32 multi-head self-attention for the PoseNet sequence shape: S = 144 tokens (143 frames + the
// timestep token), head dim 128, no mask.  Replaces the scaled-dot-product inside
// nn.MultiheadAttention (model/posenet.py:63-69; op inventory SURVEY.md §2a).
//
// Work item = one (clip, head): 9 query blocks of 16 rows against 144 keys.  Two launch shapes:
//   FULL   one workgroup of 8 waves per item (2 waves per SIMD): waves 0..7 OWN query blocks 0..7, block 8 is
//          computed COOPERATIVELY by all eight waves (split along the keys), so every SIMD carries 2 + 1/4 blocks --
//          the 9-waves-on-4-SIMDs layout of round 1 left one SIMD with 3 blocks (75 % of the MFMA time at best).
//   SPLIT  two workgroups of 4 waves per item (query blocks 0..4 / 5..8; the first one's fifth block cooperative) on
//          the SAME XCD (blocks b and b + 8 share an L2), used while the items alone cannot fill the 256 CUs
//          (B = 32 clips x 4 heads = 128 items).
// Data movement: K and V tiles [144 x 128] go HBM -> LDS by LDS-DMA (global_load_lds_dwordx4, no staging registers),
// all issued in the first microsecond; K arrives in three 48-key groups and the QK^T MFMAs of a group start as soon as
// that group has landed (counted s_waitcnt vmcnt(N) + raw s_barrier), V lands underneath the QK^T phase.  Round 1
// loaded everything, THEN computed: with one workgroup per CU the 56 MB of q/k/v (B = 64) cost ~12 us of idle MFMA.
// The Q rows are staged through the (still empty) V buffer so that no ordinary global load is outstanding beside the
// DMAs (hipcc would wait vmcnt(0) at its first use and drain the DMA queue).
//   LDS images: K [144][128] with the 16-byte chunk index XOR-ed by (row & 15) -- applied on the per-lane SOURCE
//   address, LDS-DMA writes lane-linear -- so the QK^T fragment reads (16 rows x one chunk per lane group) are
//   conflict-free ds_read_b128; V [144][128] plain: the PV fragment read is 16 consecutive chunks of 4 rows, already
//   conflict-free.  147,456 B + 2 KiB scratch.
//   QK^T  S^T = K . Q^T on v_mfma_f32_16x16x4_f32: the accumulator of key block kb holds
//         S^T[key = 16kb + 4g + r][query = l & 15]: a lane owns 36 of the 144 scores of ONE query, the softmax
//         reduction is in-register plus two cross-lane steps; scores never touch LDS or HBM.
//   PV    O = P . V with P straight from those accumulators (they ARE the A-operand layout) and ONE ds_read_b128 of
//         V[key][64 db + 4 li .. + 3] feeding FOUR MFMAs (output columns 64 db + 4 li + m, m = 0..3): a lane ends
//         with 4 consecutive output floats per row = 16-byte stores.  (Round 1 issued one ds_read_b32 per MFMA.)
//   Cooperative block: wave w takes the key tiles kb = w, w + NW, ...; it keeps an un-normalised P_w = exp(s - m_w)
//         with its own row maximum m_w and row sum l_w, computes O_w = P_w V over its keys, and the waves combine
//         O = sum_w e^(m_w - M) O_w / sum_w e^(m_w - M) l_w through the (by then free) K buffer.
#include "common.h"
#include "planes.h"

namespace rohm {

constexpr int AT_S = 144;      // tokens
constexpr int AT_DH = 128;     // head dim
constexpr int AT_NB = 9;       // 16-row blocks
constexpr int AT_TILE = AT_S * AT_DH;              // floats per K / V image
constexpr int AT_LDS_FLOATS = 2 * AT_TILE + 512;   // + 1 KiB DMA landing zone + 1 KiB statistics

// s_waitcnt vmcnt(n) only (gfx9 encoding: vmcnt [3:0] + [15:14], expcnt [6:4] = 7, lgkmcnt [11:8] = 15), through the
// builtin so that hipcc's own wait-count bookkeeping sees the DMA queue drain: while it believes an LDS-DMA is pending
// it degrades every LDS wait to lgkmcnt(0).
#define AT_WAIT_VM(n) __builtin_amdgcn_s_waitcnt(((n) & 15) | (7 << 4) | (15 << 8) | (((n) >> 4) << 14))
#define AT_WAIT_LGKM0() __builtin_amdgcn_s_waitcnt(15 | (7 << 4) | (0 << 8) | (3 << 14))

__device__ __forceinline__ void at_dma16(const float* src, float* lds_dst) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                     (__attribute__((address_space(3))) void*)lds_dst, 16, 0, 0);
}

// Phase stamps for scripts/probes/attn_timeline.hip (compiled only there, with -DAT_TIMELINE).
#ifdef AT_TIMELINE
#define AT_TL_PARAM , unsigned long long* __restrict__ tl
#define AT_STAMP(i)                                                                                          \
    do {                                                                                                     \
        if (lane == 0) tl[((size_t)blockIdx.x * NW + wave) * 16 + (i)] = __builtin_amdgcn_s_memtime();     \
    } while (0)
#else
#define AT_TL_PARAM
#define AT_STAMP(i)
#endif

// NPO = 0: ctx is the fp32 matrix [n_seq * 144][n_head * 128]; NPO = 2 / 3 / 16: ctx receives the bf16 (fp16) PLANES of that matrix
// instead (planes.h; the consumer is the out-projection of the split-bf16 path, gemm_pp.hip).  For plane output the P.V
// MFMAs run with their operands exchanged (O^T = V^T P^T: the same products summed in the same order), which leaves a lane
// with 16 CONSECUTIVE output columns of one query row -- two complete 16-byte units per plane, 256 contiguous bytes per 16 lanes.
template <int NW, int NPO = 0>
__global__ __launch_bounds__(NW * 64) void attention_f32_kernel(const float* __restrict__ qkv,
                                                                float* __restrict__ ctx, int n_head, int n_items,
                                                                int split AT_TL_PARAM) {
    constexpr int OWNED = NW;                       // query blocks owned by one wave each
    constexpr int NQP = (NW == 8) ? 9 : 10;         // Q staging pieces (1 KiB) per wave
    constexpr int NKP = 24 / NW;                    // K pieces per wave per 48-key group
    constexpr int NVP = 72 / NW;                    // V pieces per wave
    constexpr int NV3 = NVP / 3;                    // ... per instalment
    constexpr int NT = (AT_NB + NW - 1) / NW;       // cooperative key tiles per wave (max)
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* Ks = smem;
    float* Vs = smem + AT_TILE;
    float* dummy = smem + 2 * AT_TILE;              // 256 floats
    float* stats = dummy + 256;                     // [NW][16][2]

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int li = lane & 15, lg = lane >> 4;

    int item, q0, nq;
    if (split) {      // blocks b and b + 8 (same XCD) are the two halves of one item
        const int b = blockIdx.x;
        item = (b >> 4) * 8 + (b & 7);
        const int half = (b >> 3) & 1;
        q0 = half ? 5 : 0;
        nq = half ? 4 : 5;
    } else {
        item = blockIdx.x;
        q0 = 0;
        nq = AT_NB;
    }
    if (item >= n_items) return;
    AT_STAMP(0);
    const bool coop = nq > OWNED;
    const int seq = item / n_head, head = item % n_head;
    const int D = n_head * AT_DH;
    const size_t ldq = (size_t)3 * D;
    const float* qg = qkv + (size_t)seq * AT_S * ldq + head * AT_DH;
    const float* kg = qg + D;
    const float* vg = qg + 2 * D;

    // ---- issue: Q rows of this workgroup -> V buffer (swizzled like K), all of K -> K buffer ------------------
    const int half_row = lane >> 5, cphys = lane & 31;
#pragma unroll
    for (int i = 0; i < NQP; ++i) {
        const int lp = i * NW + wave;                                   // local piece = local rows 2lp, 2lp + 1
        const bool ok = lp < nq * 8;
        const int lr = ok ? 2 * lp + half_row : half_row;
        const float* src = qg + (size_t)(q0 * 16 + lr) * ldq + ((cphys ^ (lr & 15)) << 2);
        at_dma16(src, ok ? Vs + lp * 256 : dummy);
    }
    // Only what the first MFMA needs goes out first (Q + the first 48 keys): with the rest of K and V queued behind
    // them every CU's first bytes arrive later (measured: first MFMA at 7.2 us instead of ~5 at B = 64).
    auto issue_k = [&](int G) {
#pragma unroll
        for (int i = 0; i < NKP; ++i) {
            const int piece = 24 * G + i * NW + wave;
            const int row = 2 * piece + half_row;
            at_dma16(kg + (size_t)row * ldq + ((cphys ^ (row & 15)) << 2), Ks + piece * 256);
        }
    };
    issue_k(0);
    AT_WAIT_VM(NKP);                     // this wave's Q pieces have landed
    __builtin_amdgcn_s_barrier();        // ... and everybody else's
    AT_STAMP(1);

    // Q fragments: lane (query li, g) holds Q[q][16 ks + 4 g + j], j = 0..3
    f32x4 qf[8], qc[8];
#pragma unroll
    for (int ks = 0; ks < 8; ++ks) {
        qf[ks] = *reinterpret_cast<const f32x4*>(Vs + (wave * 16 + li) * AT_DH + (((ks * 4 + lg) ^ li) << 2));
        qc[ks] = *reinterpret_cast<const f32x4*>(Vs + ((coop ? OWNED : 0) * 16 + li) * AT_DH + (((ks * 4 + lg) ^ li) << 2));
    }
    AT_WAIT_LGKM0();
    __builtin_amdgcn_s_barrier();        // every wave has its Q: the V buffer may be overwritten
    // The rest of K and V is issued in three instalments, one ahead of each QK^T group: issuing all 27 (54) DMA
    // instructions here stalls every wave in the issue loop until the CU's memory queue has drained (measured: first
    // MFMA at 16k cycles although Q had landed at 8k).
    auto issue_v = [&](int part) {
#pragma unroll
        for (int i = part * NV3; i < (part + 1) * NV3; ++i) {
            const int piece = i * NW + wave;
            at_dma16(vg + (size_t)(2 * piece + half_row) * ldq + (cphys << 2), Vs + piece * 256);
        }
    };
    // Issue order behind Q and the first key group.  AT_KFIRST (default): K1, K2 | V0, V1 | V2 -- all of K ahead of all of V: the
    // QK^T phase is gated by K's arrival (timeline at B = 32, profiles/r5_attn_split_*: the third key group landed at 11.8k cycles
    // with V instalments queued in between, 9.2k without), while V is not needed before the softmax is done and still lands
    // underneath QK^T.  AT_KFIRST=0: the round-2 order K1, V0 | K2, V1 | V2.
#ifndef AT_KFIRST
#define AT_KFIRST 1
#endif
    issue_k(1);
    if constexpr (AT_KFIRST != 0) issue_k(2);
    else issue_v(0);

    // ---- QK^T of the owned block, one 48-key group at a time as K lands ----------------------------------------
    f32x4 sacc[AT_NB];
#pragma unroll
    for (int kb = 0; kb < AT_NB; ++kb) sacc[kb] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int G = 0; G < 3; ++G) {
        // (vmcnt retires in issue order)
        if constexpr (AT_KFIRST != 0) {              // issue order: Q, K0 | K1, K2 | V0, V1 | V2
            if (G == 0) {
                AT_WAIT_VM(2 * NKP);                 // K0 landed; K1, K2 may be in flight
            } else if (G == 1) {
                issue_v(0);
                issue_v(1);
                AT_WAIT_VM(NKP + 2 * NV3);           // K1 landed; K2, V0, V1 in flight
            } else {
                issue_v(2);
                AT_WAIT_VM(3 * NV3);                 // K2 landed; V0, V1, V2 in flight
            }
        } else if (G == 0) {                         // issue order: Q, K0 | K1, V0 | K2, V1 | V2
            AT_WAIT_VM(NKP + NV3);                   // K0 landed; K1, V0 may be in flight
        } else if (G == 1) {
            issue_k(2);
            issue_v(1);
            AT_WAIT_VM(NV3 + NKP + NV3);             // K1 landed; V0, K2, V1 in flight
        } else {
            issue_v(2);
            AT_WAIT_VM(2 * NV3);                     // K2 (and V0) landed; V1, V2 in flight
        }
        __builtin_amdgcn_s_barrier();
        AT_STAMP(2 + G);
        auto kread = [&](f32x4* kf, int ks) {
#pragma unroll
            for (int c = 0; c < 3; ++c)
                kf[c] = *reinterpret_cast<const f32x4*>(Ks + ((3 * G + c) * 16 + li) * AT_DH + (((ks * 4 + lg) ^ li) << 2));
        };
        // While a DMA is in flight hipcc waits lgkmcnt(0) before the first MFMA of every step, so the prefetch of step
        // ks + 1 is issued AFTER that wait (behind the first three MFMAs) and has nine MFMAs to land.
        f32x4 kf[2][3];
        kread(kf[0], 0);
#pragma unroll
        for (int ks = 0; ks < 8; ++ks) {
#pragma unroll
            for (int c = 0; c < 3; ++c)
                sacc[3 * G + c] = __builtin_amdgcn_mfma_f32_16x16x4f32(kf[ks & 1][c][0], qf[ks][0], sacc[3 * G + c], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
            if (ks + 1 < 8) kread(kf[(ks + 1) & 1], ks + 1);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int j = 1; j < 4; ++j)
#pragma unroll
                for (int c = 0; c < 3; ++c)
                    sacc[3 * G + c] = __builtin_amdgcn_mfma_f32_16x16x4f32(kf[ks & 1][c][j], qf[ks][j], sacc[3 * G + c], 0, 0, 0);
        }
    }
    // cooperative block: this wave's key tiles
    f32x4 cs[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) cs[t] = f32x4{0.f, 0.f, 0.f, 0.f};
    if (coop) {
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            const int kb = wave + t * NW;
            if (kb < AT_NB) {
#pragma unroll
                for (int ks = 0; ks < 8; ++ks) {
                    const f32x4 kf = *reinterpret_cast<const f32x4*>(Ks + (kb * 16 + li) * AT_DH + (((ks * 4 + lg) ^ li) << 2));
#pragma unroll
                    for (int j = 0; j < 4; ++j)
                        cs[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(kf[j], qc[ks][j], cs[t], 0, 0, 0);
                }
            }
        }
    }

    AT_STAMP(5);
    // ---- softmax of the owned block over the 144 keys of query li (normalised before P.V, like the reference) ----
    constexpr float LOG2E = 1.4426950408889634f;
    {
        float mx = sacc[0][0];
#pragma unroll
        for (int kb = 0; kb < AT_NB; ++kb)
#pragma unroll
            for (int r = 0; r < 4; ++r) mx = fmaxf(mx, sacc[kb][r]);
        mx = fmaxf(mx, __shfl_xor(mx, 16));
        mx = fmaxf(mx, __shfl_xor(mx, 32));
        const float mb = mx * LOG2E;
        float sum = 0.f;
#pragma unroll
        for (int kb = 0; kb < AT_NB; ++kb)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float e = __builtin_amdgcn_exp2f(fmaf(sacc[kb][r], LOG2E, -mb));
                sacc[kb][r] = e;
                sum += e;
            }
        sum += __shfl_xor(sum, 16);
        sum += __shfl_xor(sum, 32);
        const float inv = 1.0f / sum;
#pragma unroll
        for (int kb = 0; kb < AT_NB; ++kb)
#pragma unroll
            for (int r = 0; r < 4; ++r) sacc[kb][r] *= inv;
    }
    // cooperative block: un-normalised P_w with this wave's own maximum / sum
    float c_m = -INFINITY, c_l = 0.f;
    if (coop) {
#pragma unroll
        for (int t = 0; t < NT; ++t)
            if (wave + t * NW < AT_NB)
#pragma unroll
                for (int r = 0; r < 4; ++r) c_m = fmaxf(c_m, cs[t][r]);
        c_m = fmaxf(c_m, __shfl_xor(c_m, 16));
        c_m = fmaxf(c_m, __shfl_xor(c_m, 32));
        const float mb = c_m * LOG2E;
#pragma unroll
        for (int t = 0; t < NT; ++t)
            if (wave + t * NW < AT_NB)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float e = __builtin_amdgcn_exp2f(fmaf(cs[t][r], LOG2E, -mb));
                    cs[t][r] = e;
                    c_l += e;
                }
        c_l += __shfl_xor(c_l, 16);
        c_l += __shfl_xor(c_l, 32);
    }

    AT_STAMP(6);
    AT_WAIT_VM(0);                       // V has landed
    __builtin_amdgcn_s_barrier();        // ... for every wave; and every wave is done reading K
    AT_STAMP(7);

    // ---- cooperative block first: partial O_w over this wave's keys -> LDS (the K buffer is free now); the combine
    // comes after the owned block, so the LDS writes and the statistics land underneath 288 MFMAs ---------------------
    if (coop) {
        float* part = Ks + wave * (16 * AT_DH);     // [16][128]
#pragma unroll
        for (int db = 0; db < 2; ++db) {
            f32x4 oacc[4];
#pragma unroll
            for (int m = 0; m < 4; ++m) oacc[m] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int t = 0; t < NT; ++t) {
                const int kb = wave + t * NW;
                if (kb < AT_NB) {
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const f32x4 vf = *reinterpret_cast<const f32x4*>(Vs + (kb * 16 + lg * 4 + j) * AT_DH + db * 64 + li * 4);
#pragma unroll
                        for (int m = 0; m < 4; ++m)
                            oacc[m] = __builtin_amdgcn_mfma_f32_16x16x4f32(cs[t][j], vf[m], oacc[m], 0, 0, 0);
                    }
                }
            }
#pragma unroll
            for (int r = 0; r < 4; ++r)
                *reinterpret_cast<f32x4*>(part + (lg * 4 + r) * AT_DH + db * 64 + li * 4) =
                    f32x4{oacc[0][r], oacc[1][r], oacc[2][r], oacc[3][r]};
        }
        if (lg == 0) {
            stats[(wave * 16 + li) * 2 + 0] = c_m;
            stats[(wave * 16 + li) * 2 + 1] = c_l;
        }
    }
    AT_STAMP(9);

    // ---- P.V of the owned block ----------------------------------------------------------------------------------
    {
        auto pv_mfma = [](float pval, float vval, const f32x4& c) {
            if constexpr (NPO == 0) return __builtin_amdgcn_mfma_f32_16x16x4f32(pval, vval, c, 0, 0, 0);
            else return __builtin_amdgcn_mfma_f32_16x16x4f32(vval, pval, c, 0, 0, 0);
        };
        float* out = ctx + ((size_t)seq * AT_S + (q0 + wave) * 16) * D + head * AT_DH;
#pragma unroll
        for (int db = 0; db < 2; ++db) {
            f32x4 oacc[4];
#pragma unroll
            for (int m = 0; m < 4; ++m) oacc[m] = f32x4{0.f, 0.f, 0.f, 0.f};
            auto vread = [&](int st) {      // step st = 4 kb + j: keys 16 kb + 4 g + j
                return *reinterpret_cast<const f32x4*>(Vs + ((st >> 2) * 16 + lg * 4 + (st & 3)) * AT_DH + db * 64 + li * 4);
            };
            // steps are taken in pairs (8 MFMAs); the two reads of the next pair go out behind the first MFMA
            f32x4 vf[2][2];
            vf[0][0] = vread(0);
            vf[0][1] = vread(1);
#pragma unroll
            for (int sp = 0; sp < 2 * AT_NB; ++sp) {
                const int st = 2 * sp;
                oacc[0] = pv_mfma(sacc[st >> 2][st & 3], vf[sp & 1][0][0], oacc[0]);
                __builtin_amdgcn_sched_barrier(0);
                if (sp + 1 < 2 * AT_NB) {
                    vf[(sp + 1) & 1][0] = vread(st + 2);
                    vf[(sp + 1) & 1][1] = vread(st + 3);
                }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int m = 1; m < 4; ++m)
                    oacc[m] = pv_mfma(sacc[st >> 2][st & 3], vf[sp & 1][0][m], oacc[m]);
#pragma unroll
                for (int m = 0; m < 4; ++m)
                    oacc[m] = pv_mfma(sacc[(st + 1) >> 2][(st + 1) & 3], vf[sp & 1][1][m], oacc[m]);
            }
            if constexpr (NPO == 0) {
                // oacc[m][r] = O[query 4g + r][d = 64 db + 4 li + m]
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    *reinterpret_cast<f32x4*>(out + (size_t)(lg * 4 + r) * D + db * 64 + li * 4) =
                        f32x4{oacc[0][r], oacc[1][r], oacc[2][r], oacc[3][r]};
            } else {
                // exchanged operands: oacc[m][r] = O[query li][d = 64 db + 16 g + 4 r + m]
                const int row = seq * AT_S + (q0 + wave) * 16 + li;
                const int kg0 = (head * AT_DH + db * 64 + lg * 16) >> 3;
#pragma unroll
                for (int h = 0; h < 2; ++h)
                    plane_store8<(NPO ? NPO : 2)>(reinterpret_cast<char*>(ctx), row, kg0 + h, D / 32,
                                                  f32x4{oacc[0][2 * h], oacc[1][2 * h], oacc[2][2 * h], oacc[3][2 * h]},
                                                  f32x4{oacc[0][2 * h + 1], oacc[1][2 * h + 1], oacc[2][2 * h + 1], oacc[3][2 * h + 1]});
            }
        }
    }
    AT_STAMP(8);
    if (!coop) return;                   // uniform per workgroup

    AT_WAIT_LGKM0();
    __builtin_amdgcn_s_barrier();
    AT_STAMP(10);
    {
        float* out = ctx + ((size_t)seq * AT_S + (q0 + OWNED) * 16) * D + head * AT_DH;
        for (int u = tid; u < 16 * 32; u += NW * 64) {
            const int q = u >> 5, c4 = u & 31;
            float M = -INFINITY;
#pragma unroll
            for (int w = 0; w < NW; ++w) M = fmaxf(M, stats[(w * 16 + q) * 2]);
            float e[NW], L = 0.f;
#pragma unroll
            for (int w = 0; w < NW; ++w) {
                e[w] = __builtin_amdgcn_exp2f((stats[(w * 16 + q) * 2] - M) * LOG2E);
                L = fmaf(e[w], stats[(w * 16 + q) * 2 + 1], L);
            }
            const float inv = 1.0f / L;
            f32x4 o = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int w = 0; w < NW; ++w) {
                const f32x4 pv = *reinterpret_cast<const f32x4*>(Ks + w * (16 * AT_DH) + q * AT_DH + c4 * 4);
                const float s = e[w] * inv;
                o[0] = fmaf(s, pv[0], o[0]);
                o[1] = fmaf(s, pv[1], o[1]);
                o[2] = fmaf(s, pv[2], o[2]);
                o[3] = fmaf(s, pv[3], o[3]);
            }
            if constexpr (NPO == 0) *reinterpret_cast<f32x4*>(out + (size_t)q * D + c4 * 4) = o;
            else plane_store4<(NPO ? NPO : 2)>(reinterpret_cast<char*>(ctx), seq * AT_S + (q0 + OWNED) * 16 + q, head * AT_DH + c4 * 4, D / 32, o);
        }
    }
    AT_STAMP(11);
}

#ifndef AT_TIMELINE

// ---- any sequence length / head dim 64 or 128 ----------------------------------------------------------------------
// The reference class takes any clip length and its default width is 256 = 4 heads x 64 (model/posenet.py:12-20); the
// kernel above is specialised to the shape of every released configuration (S = 144, d_h = 128).  This one is the
// general path: one workgroup of four waves per (item, group of four 16-query blocks), keys streamed through LDS in
// chunks of 64 with an online softmax (running row maximum / sum, accumulators rescaled when the maximum moves), keys
// beyond S masked, rows beyond S not stored.  Same MFMA, same fragment layouts; correctness first (register-staged
// loads, b32 V fragment reads), ~half the speed of the specialised kernel at its own shape.
template <int DH>
__global__ __launch_bounds__(256) void attention_f32_generic_kernel(const float* __restrict__ qkv, float* __restrict__ ctx,
                                                                    int n_head, int S) {
    constexpr int KS = DH + 8, VS = DH + 4, CH = 64, NDB = DH / 16;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* Ks = smem;                 // [CH][KS]
    float* Vs = smem + CH * KS;       // [CH][VS]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int li = lane & 15, lg = lane >> 4;
    const int nqg = (S + 63) / 64;
    const int item = blockIdx.x / nqg, qg = blockIdx.x % nqg;
    const int seq = item / n_head, head = item % n_head;
    const int D = n_head * DH;
    const size_t ldq = (size_t)3 * D;
    const float* qg_ = qkv + (size_t)seq * S * ldq + head * DH;
    const float* kg = qg_ + D;
    const float* vg = qg_ + 2 * D;
    const int q_row = qg * 64 + wave * 16 + li;                       // this lane's query
    f32x4 qf[DH / 16];
#pragma unroll
    for (int ks = 0; ks < DH / 16; ++ks)
        qf[ks] = (q_row < S) ? *reinterpret_cast<const f32x4*>(qg_ + (size_t)q_row * ldq + ks * 16 + lg * 4)
                             : f32x4{0.f, 0.f, 0.f, 0.f};
    f32x4 oacc[NDB];
#pragma unroll
    for (int db = 0; db < NDB; ++db) oacc[db] = f32x4{0.f, 0.f, 0.f, 0.f};
    float m_run = -INFINITY, l_run = 0.f;
    constexpr float LOG2E = 1.4426950408889634f;
    for (int k0 = 0; k0 < S; k0 += CH) {
        __syncthreads();                                              // previous chunk fully consumed
        for (int u = tid; u < CH * (DH / 4); u += 256) {
            const int r = u / (DH / 4), c4 = u % (DH / 4);
            const bool ok = k0 + r < S;
            const f32x4 z = f32x4{0.f, 0.f, 0.f, 0.f};
            const f32x4 kv = ok ? *reinterpret_cast<const f32x4*>(kg + (size_t)(k0 + r) * ldq + c4 * 4) : z;
            const f32x4 vv = ok ? *reinterpret_cast<const f32x4*>(vg + (size_t)(k0 + r) * ldq + c4 * 4) : z;
            *reinterpret_cast<f32x4*>(Ks + r * KS + c4 * 4) = kv;
            *reinterpret_cast<f32x4*>(Vs + r * VS + c4 * 4) = vv;
        }
        __syncthreads();
        const int nkb = (min(CH, S - k0) + 15) / 16;
        for (int kb = 0; kb < nkb; ++kb) {
            f32x4 sc = f32x4{0.f, 0.f, 0.f, 0.f};                    // S^T[key 16kb + 4g + r][query li]
#pragma unroll
            for (int ks = 0; ks < DH / 16; ++ks) {
                const f32x4 kf = *reinterpret_cast<const f32x4*>(Ks + (kb * 16 + li) * KS + ks * 16 + lg * 4);
#pragma unroll
                for (int j = 0; j < 4; ++j) sc = __builtin_amdgcn_mfma_f32_16x16x4f32(kf[j], qf[ks][j], sc, 0, 0, 0);
            }
            float mx = -INFINITY;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                if (k0 + kb * 16 + lg * 4 + r >= S) sc[r] = -INFINITY;
                mx = fmaxf(mx, sc[r]);
            }
            mx = fmaxf(mx, __shfl_xor(mx, 16));
            mx = fmaxf(mx, __shfl_xor(mx, 32));
            const float m_new = fmaxf(m_run, mx);                     // finite: the first block always has a valid key
            const float corr = __builtin_amdgcn_exp2f((m_run - m_new) * LOG2E);      // exp(-inf) = 0 on the first block
            float ps = 0.f;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                sc[r] = __builtin_amdgcn_exp2f((sc[r] - m_new) * LOG2E);
                ps += sc[r];
            }
            ps += __shfl_xor(ps, 16);
            ps += __shfl_xor(ps, 32);
            l_run = l_run * corr + ps;
            m_run = m_new;
#pragma unroll
            for (int db = 0; db < NDB; ++db) {
#pragma unroll
                for (int r = 0; r < 4; ++r) oacc[db][r] *= corr;
                const float* vp = Vs + (kb * 16 + lg * 4) * VS + db * 16 + li;
#pragma unroll
                for (int j = 0; j < 4; ++j)      // O^T = V^T P^T: the lane ends with O[query li][16 db + 4 g + r]
                    oacc[db] = __builtin_amdgcn_mfma_f32_16x16x4f32(vp[j * VS], sc[j], oacc[db], 0, 0, 0);
            }
        }
    }
    if (q_row < S) {
        const float inv = 1.0f / l_run;
        float* out = ctx + ((size_t)seq * S + q_row) * D + head * DH;
#pragma unroll
        for (int db = 0; db < NDB; ++db)
            *reinterpret_cast<f32x4*>(out + db * 16 + lg * 4) =
                f32x4{oacc[db][0] * inv, oacc[db][1] * inv, oacc[db][2] * inv, oacc[db][3] * inv};
    }
}

template <int DH>
static int launch_generic(const float* qkv, float* ctx, int n_seq, int n_head, int S, hipStream_t s) {
    const size_t lds = (size_t)64 * (2 * DH + 12) * sizeof(float);
    static bool attr_set[64] = {};
    int dev = 0;
    ROHM_HIP_CHECK(hipGetDevice(&dev));
    if (dev < 64 && !attr_set[dev]) {
        ROHM_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&attention_f32_generic_kernel<DH>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        attr_set[dev] = true;
    }
    const int items = n_seq * n_head;
    prof::Scope ps("attention_generic", 4.0 * S * S * DH * (double)items, 4.0 * 4.0 * S * DH * (double)items, s);
    hipLaunchKernelGGL(attention_f32_generic_kernel<DH>, dim3(items * ((S + 63) / 64)), dim3(256), lds, s, qkv, ctx, n_head, S);
    ROHM_LAUNCH_CHECK();
    return ROHM_OK;
}

template <int NW, int NPO>
static int set_lds_attr(int dev) {
    static bool attr_set[64] = {};
    if (dev < 64 && !attr_set[dev]) {
        ROHM_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&attention_f32_kernel<NW, NPO>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize,
                                           (int)(AT_LDS_FLOATS * sizeof(float))));
        attr_set[dev] = true;
    }
    return ROHM_OK;
}

// specialised shape (S = 144, d_h = 128); NPO = 0: fp32 ctx, 2 / 3: planes of ctx
template <int NPO>
static int launch_special(const float* qkv, float* ctx, int n_seq, int n_head, hipStream_t s) {
    const size_t lds = AT_LDS_FLOATS * sizeof(float);
    int dev = 0;
    ROHM_HIP_CHECK(hipGetDevice(&dev));
    const int items = n_seq * n_head;
    // One 8-wave workgroup per item takes ~1.6x the time of a 4-wave half-item workgroup: split while the halves
    // still fit the chip in fewer (weighted) rounds.
    constexpr int kCUs = 256;
    const int rounds_full = (items + kCUs - 1) / kCUs, rounds_split = (2 * items + kCUs - 1) / kCUs;
    const bool split = 10 * rounds_split <= 16 * rounds_full;
    prof::Scope ps("attention", 4.0 * AT_S * AT_S * AT_DH * (double)items, 4.0 * 4.0 * AT_S * AT_DH * (double)items, s);
    if (split) {
        if (int e = set_lds_attr<4, NPO>(dev)) return e;
        const int grid = ((items + 7) / 8) * 16;
        hipLaunchKernelGGL((attention_f32_kernel<4, NPO>), dim3(grid), dim3(256), lds, s, qkv, ctx, n_head, items, 1);
    } else {
        if (int e = set_lds_attr<8, NPO>(dev)) return e;
        hipLaunchKernelGGL((attention_f32_kernel<8, NPO>), dim3(items), dim3(512), lds, s, qkv, ctx, n_head, items, 0);
    }
    ROHM_LAUNCH_CHECK();
    return ROHM_OK;
}

int launch_attention(const float* qkv, float* ctx, int n_seq, int n_head, int n_tok, int head_dim, hipStream_t s) {
    ROHM_ARG_CHECK(n_seq > 0 && n_head > 0 && n_tok > 0, "attention: empty problem");
    ROHM_ARG_CHECK(((uintptr_t)qkv % 16) == 0 && ((uintptr_t)ctx % 16) == 0, "attention: qkv / ctx must be 16-byte aligned");
    if (n_tok != AT_S || head_dim != AT_DH) {
        if (head_dim == 128) return launch_generic<128>(qkv, ctx, n_seq, n_head, n_tok, s);
        if (head_dim == 64) return launch_generic<64>(qkv, ctx, n_seq, n_head, n_tok, s);
        set_error("attention: head dim must be 64 or 128 (got %d)", head_dim);
        return ROHM_ERR_UNSUPPORTED;
    }
    return launch_special<0>(qkv, ctx, n_seq, n_head, s);
}

int launch_attention_planes(const float* qkv, void* ctx_planes, int n_seq, int n_head, int nplane, hipStream_t s) {
    ROHM_ARG_CHECK(n_seq > 0 && n_head > 0 && qkv && ctx_planes, "attention_planes: empty problem / null pointer");
    ROHM_ARG_CHECK(((uintptr_t)qkv % 16) == 0 && ((uintptr_t)ctx_planes % 16) == 0, "attention_planes: operands must be 16-byte aligned");
    ROHM_ARG_CHECK(mode_ok(nplane), "attention_planes: mode must be 3, 2 or 16");
    if (nplane == 3) return launch_special<3>(qkv, (float*)ctx_planes, n_seq, n_head, s);
    if (nplane == 2) return launch_special<2>(qkv, (float*)ctx_planes, n_seq, n_head, s);
    return launch_special<kModeF16>(qkv, (float*)ctx_planes, n_seq, n_head, s);
}
#endif  // AT_TIMELINE

}  // namespace rohm
