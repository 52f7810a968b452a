// fp32 multi-head self-attention for the PoseNet sequence shape: S = 144 tokens (143 frames + the
// timestep token), head dim 128, no mask.  Replaces the scaled-dot-product inside
// nn.MultiheadAttention (model/posenet.py:63-69; op inventory SURVEY.md §2a).
//
// One workgroup per (clip, head), nine waves, wave w owns the 16 query rows [16w, 16w+16):
//   phase 0  K tile [144 x 128] -> LDS (row stride 136 floats: conflict-free ds_read_b128);
//            V tile is fetched into registers and parked there while phase 1 runs.
//   phase 1  S^T = K . Q^T on v_mfma_f32_16x16x4_f32 ("swapped QK^T"): the accumulator of key
//            block kb holds S^T[key = 16kb + 4g + r][query = l & 15], i.e. each lane owns 36 of the
//            144 scores of ONE query row, so the softmax row reduction is in-register plus two
//            cross-lane steps (lanes l, l^16, l^32, l^48 share a query).
//   phase 2  softmax in fp32 (max-subtracted, normalised before P.V like the reference).
//   phase 3  O^T = V^T . P^T (operands swapped so each lane ends with 4 consecutive output floats = one
//            16-byte store): the S^T accumulator layout *is* the MFMA operand layout of P
//            (lane (i = query, g) register j = P[query][16kb + 4g + j]), so P never leaves
//            registers; V comes from LDS (row stride 132 floats: conflict-free ds_read_b32).
// Scores never touch LDS or HBM.  LDS: (136 + 132) * 144 * 4 = 154,368 B (of 160 KiB).
#include "common.h"

namespace rohm {

constexpr int AT_S = 144;      // tokens
constexpr int AT_DH = 128;     // head dim
constexpr int AT_NB = 9;       // 16-row blocks
constexpr int AT_KS = 136;     // K row stride (floats)
constexpr int AT_VS = 132;     // V row stride (floats)
constexpr int AT_THREADS = 576;
constexpr int AT_UNITS = AT_S * AT_DH / 4 / AT_THREADS;   // 16-byte units per thread per tile = 8

__global__ __launch_bounds__(AT_THREADS) void attention_f32_kernel(const float* __restrict__ qkv,
                                                                   float* __restrict__ ctx, int n_head) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* Ks = smem;                    // [144][136]
    float* Vs = smem + AT_S * AT_KS;     // [144][132]

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;           // query block
    const int li = lane & 15, lg = lane >> 4;
    const int seq = blockIdx.x / n_head, head = blockIdx.x % n_head;
    const int D = n_head * AT_DH;
    const size_t ldq = (size_t)3 * D;
    const float* base = qkv + (size_t)seq * AT_S * ldq + head * AT_DH;
    const float* qg = base;
    const float* kg = base + D;
    const float* vg = base + 2 * D;

    // ---- phase 0: K -> LDS, V -> registers ----------------------------------------------------
    f32x4 kr[AT_UNITS], vr[AT_UNITS];
#pragma unroll
    for (int i = 0; i < AT_UNITS; ++i) {
        const int u = tid + i * AT_THREADS;
        const int row = u >> 5, c4 = u & 31;
        kr[i] = *reinterpret_cast<const f32x4*>(kg + (size_t)row * ldq + c4 * 4);
    }
    // Q fragments for this wave's 16 queries: lane (query li, g) holds Q[q][16*ks + 4g + j]
    f32x4 qf[8];
#pragma unroll
    for (int ks = 0; ks < 8; ++ks)
        qf[ks] = *reinterpret_cast<const f32x4*>(qg + (size_t)(wave * 16 + li) * ldq + ks * 16 + lg * 4);
#pragma unroll
    for (int i = 0; i < AT_UNITS; ++i) {
        const int u = tid + i * AT_THREADS;
        const int row = u >> 5, c4 = u & 31;
        *reinterpret_cast<f32x4*>(Ks + row * AT_KS + c4 * 4) = kr[i];
    }
#pragma unroll
    for (int i = 0; i < AT_UNITS; ++i) {
        const int u = tid + i * AT_THREADS;
        const int row = u >> 5, c4 = u & 31;
        vr[i] = *reinterpret_cast<const f32x4*>(vg + (size_t)row * ldq + c4 * 4);
    }
    __syncthreads();

    // ---- phase 1: S^T[key][query] = sum_d K[key][d] Q[query][d] -------------------------------
    f32x4 sacc[AT_NB];
#pragma unroll
    for (int kb = 0; kb < AT_NB; ++kb) sacc[kb] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int ks = 0; ks < 8; ++ks) {
#pragma unroll
        for (int kb = 0; kb < AT_NB; ++kb) {
            const f32x4 kf = *reinterpret_cast<const f32x4*>(Ks + (kb * 16 + li) * AT_KS + ks * 16 + lg * 4);
#pragma unroll
            for (int j = 0; j < 4; ++j)
                sacc[kb] = __builtin_amdgcn_mfma_f32_16x16x4f32(kf[j], qf[ks][j], sacc[kb], 0, 0, 0);
        }
    }

    // park V in LDS now (its global loads were in flight during phase 1)
#pragma unroll
    for (int i = 0; i < AT_UNITS; ++i) {
        const int u = tid + i * AT_THREADS;
        const int row = u >> 5, c4 = u & 31;
        *reinterpret_cast<f32x4*>(Vs + row * AT_VS + c4 * 4) = vr[i];
    }

    // ---- phase 2: softmax over the 144 keys of query li ---------------------------------------
    float mx = sacc[0][0];
#pragma unroll
    for (int kb = 0; kb < AT_NB; ++kb)
#pragma unroll
        for (int r = 0; r < 4; ++r) mx = fmaxf(mx, sacc[kb][r]);
    mx = fmaxf(mx, __shfl_xor(mx, 16));
    mx = fmaxf(mx, __shfl_xor(mx, 32));
    float sum = 0.f;
#pragma unroll
    for (int kb = 0; kb < AT_NB; ++kb)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const float e = expf(sacc[kb][r] - mx);
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

    __syncthreads();   // V visible

    // ---- phase 3: O[query][d] = sum_key P[query][key] V[key][d] -------------------------------
    float* out = ctx + ((size_t)seq * AT_S + wave * 16) * D + head * AT_DH;
#pragma unroll
    for (int db = 0; db < 8; ++db) {
        f32x4 oacc = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int kb = 0; kb < AT_NB; ++kb) {
            const float* vp = Vs + (kb * 16 + lg * 4) * AT_VS + db * 16 + li;
#pragma unroll
            for (int j = 0; j < 4; ++j)   // operands swapped (V on the "A" side): the tile comes out as O^T
                oacc = __builtin_amdgcn_mfma_f32_16x16x4f32(vp[j * AT_VS], sacc[kb][j], oacc, 0, 0, 0);
        }
        // oacc[r] = O[query = li][d = 16*db + 4*lg + r]: four consecutive floats -> one 16-byte store
        *reinterpret_cast<f32x4*>(out + (size_t)li * D + db * 16 + lg * 4) = oacc;
    }
}

int launch_attention(const float* qkv, float* ctx, int n_seq, int n_head, hipStream_t s) {
    ROHM_ARG_CHECK(n_seq > 0 && n_head > 0, "attention: empty problem");
    ROHM_ARG_CHECK(((uintptr_t)qkv % 16) == 0, "attention: qkv must be 16-byte aligned");
    const size_t lds = (size_t)AT_S * (AT_KS + AT_VS) * sizeof(float);
    static bool attr_set[64] = {};
    int dev = 0;
    ROHM_HIP_CHECK(hipGetDevice(&dev));
    if (dev < 64 && !attr_set[dev]) {
        ROHM_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&attention_f32_kernel),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        attr_set[dev] = true;
    }
    prof::Scope ps("attention", 4.0 * AT_S * AT_S * AT_DH * (double)n_seq * n_head,
                   4.0 * 4.0 * AT_S * AT_DH * (double)n_seq * n_head, s);
    hipLaunchKernelGGL(attention_f32_kernel, dim3(n_seq * n_head), dim3(AT_THREADS), lds, s, qkv, ctx, n_head);
    ROHM_LAUNCH_CHECK();
    return ROHM_OK;
}

}  // namespace rohm
