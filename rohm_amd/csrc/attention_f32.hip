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
#include "attention_tile.h"

namespace rohm {

template <int NW, int NPO = 0>
__global__ __launch_bounds__(NW * 64) void attention_f32_kernel(const float* __restrict__ qkv,
                                                                float* __restrict__ ctx, int n_head, int n_items,
                                                                int split AT_TL_PARAM) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
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
#ifdef AT_TIMELINE
    attention_item<NW, NPO>(qkv, ctx, n_head, item, q0, nq, smem, (int)threadIdx.x, tl);
#else
    attention_item<NW, NPO>(qkv, ctx, n_head, item, q0, nq, smem, (int)threadIdx.x);
#endif
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
