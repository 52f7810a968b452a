// The specialised attention work item (S = 144 tokens, head dim 128) as a device function: attention_f32.hip's kernels are thin
// wrappers around it, and encoder_chain.hip runs it between its GEMM phases.  See attention_f32.hip for the design notes.
#pragma once
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

// AUX: cache policy of the load (0: default; 16 = sc1: device scope, never served by this CU's L1 -- for operands that other
// workgroups of the SAME launch have written, encoder_chain.hip)
template <int AUX>
__device__ __forceinline__ void at_dma16(const float* src, float* lds_dst) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                     (__attribute__((address_space(3))) void*)lds_dst, 16, 0, AUX);
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
// One work item = query blocks [q0, q0 + nq) of (clip, head) `item`, computed by the NW waves of the calling workgroup: every wave
// OWNS NPASS query blocks (q0 + ps * NW + wave), a last block beyond NW * NPASS is computed cooperatively.  NPASS = 2 (NW = 4: a whole
// item on four waves, encoder_chain.hip at 4 workgroups per clip): all QK^T first, then the softmaxes, then all P.V.
template <int NW, int NPO = 0, int AUX = 0, int NPASS = 1>
__device__ __forceinline__ void attention_item(const float* __restrict__ qkv, float* __restrict__ ctx, int n_head, int item, int q0,
                                               int nq, float* smem, const int tid AT_TL_PARAM) {
    constexpr int OWNED = NW * NPASS;               // query blocks owned by one wave each
    constexpr int NQP = (NPASS == 2) ? 72 / NW : ((NW == 8) ? 9 : 10);      // Q staging pieces (1 KiB) per wave
    constexpr int NKP = 24 / NW;                    // K pieces per wave per 48-key group
    constexpr int NVP = 72 / NW;                    // V pieces per wave
    constexpr int NV3 = NVP / 3;                    // ... per instalment
    constexpr int NT = (AT_NB + NW - 1) / NW;       // cooperative key tiles per wave (max)
    float* Ks = smem;
    float* Vs = smem + AT_TILE;
    float* dummy = smem + 2 * AT_TILE;              // 256 floats
    float* stats = dummy + 256;                     // [NW][16][2]

    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int li = lane & 15, lg = lane >> 4;

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
        at_dma16<AUX>(src, ok ? Vs + lp * 256 : dummy);
    }
    // Only what the first MFMA needs goes out first (Q + the first 48 keys): with the rest of K and V queued behind
    // them every CU's first bytes arrive later (measured: first MFMA at 7.2 us instead of ~5 at B = 64).
    auto issue_k = [&](int G) {
#pragma unroll
        for (int i = 0; i < NKP; ++i) {
            const int piece = 24 * G + i * NW + wave;
            const int row = 2 * piece + half_row;
            at_dma16<AUX>(kg + (size_t)row * ldq + ((cphys ^ (row & 15)) << 2), Ks + piece * 256);
        }
    };
    issue_k(0);
    AT_WAIT_VM(NKP);                     // this wave's Q pieces have landed
    __builtin_amdgcn_s_barrier();        // ... and everybody else's
    AT_STAMP(1);

    // Q fragments: lane (query li, g) holds Q[q][16 ks + 4 g + j], j = 0..3
    f32x4 qf[NPASS][8], qc[8];
#pragma unroll
    for (int ks = 0; ks < 8; ++ks) {
#pragma unroll
        for (int ps = 0; ps < NPASS; ++ps)
            qf[ps][ks] = *reinterpret_cast<const f32x4*>(Vs + ((ps * NW + wave) * 16 + li) * AT_DH + (((ks * 4 + lg) ^ li) << 2));
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
            at_dma16<AUX>(vg + (size_t)(2 * piece + half_row) * ldq + (cphys << 2), Vs + piece * 256);
        }
    };
    // Issue order behind Q and the first key group.  AT_KFIRST (default): K1, K2 | V0, V1 | V2 -- all of K ahead of all of V;
    // AT_KFIRST=0: the round-2 order K1, V0 | K2, V1 | V2.  Measured (profiles/r5_a_attn_split_timeline.txt): the kernel is not
    // gated by the arrival of the later key groups in either order (their stamps follow the previous group's 96 MFMAs), the K-first
    // order is 0.7 % (B = 32) / 1.5 % (B = 64) faster.
#ifndef AT_KFIRST
#define AT_KFIRST 1
#endif
    issue_k(1);
    if constexpr (AT_KFIRST != 0) issue_k(2);
    else issue_v(0);

    // ---- QK^T of the owned block, one 48-key group at a time as K lands ----------------------------------------
    f32x4 sacc[NPASS][AT_NB];
#pragma unroll
    for (int ps = 0; ps < NPASS; ++ps)
#pragma unroll
        for (int kb = 0; kb < AT_NB; ++kb) sacc[ps][kb] = f32x4{0.f, 0.f, 0.f, 0.f};
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
                sacc[0][3 * G + c] = __builtin_amdgcn_mfma_f32_16x16x4f32(kf[ks & 1][c][0], qf[0][ks][0], sacc[0][3 * G + c], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
            if (ks + 1 < 8) kread(kf[(ks + 1) & 1], ks + 1);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int j = 1; j < 4; ++j)
#pragma unroll
                for (int c = 0; c < 3; ++c)
                    sacc[0][3 * G + c] = __builtin_amdgcn_mfma_f32_16x16x4f32(kf[ks & 1][c][j], qf[0][ks][j], sacc[0][3 * G + c], 0, 0, 0);
        }
    }
    // second owned block of every wave (NPASS = 2): all of K is resident, same loop without waits
    if constexpr (NPASS == 2) {
#pragma unroll
        for (int G = 0; G < 3; ++G) {
            auto kread = [&](f32x4* kf, int ks) {
#pragma unroll
                for (int c = 0; c < 3; ++c)
                    kf[c] = *reinterpret_cast<const f32x4*>(Ks + ((3 * G + c) * 16 + li) * AT_DH + (((ks * 4 + lg) ^ li) << 2));
            };
            f32x4 kf[2][3];
            kread(kf[0], 0);
#pragma unroll
            for (int ks = 0; ks < 8; ++ks) {
#pragma unroll
                for (int c = 0; c < 3; ++c)
                    sacc[NPASS - 1][3 * G + c] = __builtin_amdgcn_mfma_f32_16x16x4f32(kf[ks & 1][c][0], qf[NPASS - 1][ks][0], sacc[NPASS - 1][3 * G + c], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
                if (ks + 1 < 8) kread(kf[(ks + 1) & 1], ks + 1);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int j = 1; j < 4; ++j)
#pragma unroll
                    for (int c = 0; c < 3; ++c)
                        sacc[NPASS - 1][3 * G + c] = __builtin_amdgcn_mfma_f32_16x16x4f32(kf[ks & 1][c][j], qf[NPASS - 1][ks][j], sacc[NPASS - 1][3 * G + c], 0, 0, 0);
            }
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
#pragma unroll
    for (int ps = 0; ps < NPASS; ++ps) {
        float mx = sacc[ps][0][0];
#pragma unroll
        for (int kb = 0; kb < AT_NB; ++kb)
#pragma unroll
            for (int r = 0; r < 4; ++r) mx = fmaxf(mx, sacc[ps][kb][r]);
        mx = fmaxf(mx, __shfl_xor(mx, 16));
        mx = fmaxf(mx, __shfl_xor(mx, 32));
        const float mb = mx * LOG2E;
        float sum = 0.f;
#pragma unroll
        for (int kb = 0; kb < AT_NB; ++kb)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float e = __builtin_amdgcn_exp2f(fmaf(sacc[ps][kb][r], LOG2E, -mb));
                sacc[ps][kb][r] = e;
                sum += e;
            }
        sum += __shfl_xor(sum, 16);
        sum += __shfl_xor(sum, 32);
        const float inv = 1.0f / sum;
#pragma unroll
        for (int kb = 0; kb < AT_NB; ++kb)
#pragma unroll
            for (int r = 0; r < 4; ++r) sacc[ps][kb][r] *= inv;
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
#pragma unroll
        for (int ps = 0; ps < NPASS; ++ps) {
        float* out = ctx + ((size_t)seq * AT_S + (q0 + ps * NW + wave) * 16) * D + head * AT_DH;
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
                oacc[0] = pv_mfma(sacc[ps][st >> 2][st & 3], vf[sp & 1][0][0], oacc[0]);
                __builtin_amdgcn_sched_barrier(0);
                if (sp + 1 < 2 * AT_NB) {
                    vf[(sp + 1) & 1][0] = vread(st + 2);
                    vf[(sp + 1) & 1][1] = vread(st + 3);
                }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int m = 1; m < 4; ++m)
                    oacc[m] = pv_mfma(sacc[ps][st >> 2][st & 3], vf[sp & 1][0][m], oacc[m]);
#pragma unroll
                for (int m = 0; m < 4; ++m)
                    oacc[m] = pv_mfma(sacc[ps][(st + 1) >> 2][(st + 1) & 3], vf[sp & 1][1][m], oacc[m]);
            }
            if constexpr (NPO == 0) {
                // oacc[m][r] = O[query 4g + r][d = 64 db + 4 li + m]
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    *reinterpret_cast<f32x4*>(out + (size_t)(lg * 4 + r) * D + db * 64 + li * 4) =
                        f32x4{oacc[0][r], oacc[1][r], oacc[2][r], oacc[3][r]};
            } else {
                // exchanged operands: oacc[m][r] = O[query li][d = 64 db + 16 g + 4 r + m]
                const int row = seq * AT_S + (q0 + ps * NW + wave) * 16 + li;
                const int kg0 = (head * AT_DH + db * 64 + lg * 16) >> 3;
#pragma unroll
                for (int h = 0; h < 2; ++h)
                    plane_store8<(NPO ? NPO : 2)>(reinterpret_cast<char*>(ctx), row, kg0 + h, D / 32,
                                                  f32x4{oacc[0][2 * h], oacc[1][2 * h], oacc[2][2 * h], oacc[3][2 * h]},
                                                  f32x4{oacc[0][2 * h + 1], oacc[1][2 * h + 1], oacc[2][2 * h + 1], oacc[3][2 * h + 1]});
            }
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

}  // namespace rohm
