// Full SMPL-X linear blend skinning: vertices [N, V, 3] (+ all J joints) of smplx.SMPLX.forward.
//
// Reference: third-party smplx==0.1.28 `lbs` (blend_shapes, vertices2joints, batch_rodrigues,
// batch_rigid_transform, skinning) as called from data_loaders/motion_representation.py:379-396 with
// `return_verts=True` (test_amass_full.py:283,405,415,425: the post-loop meshes).
// The denoising loops and the guidance never need vertices (smplx.hip); this is the mesh path.
//
//   lbs_pose_kernel   one thread per frame: Rodrigues, rest joints from the folded regressor, the 55-joint chain,
//                     relative transforms A[m][j] = [G_j | P_j - G_j J_j] and the pose feature (R_j - I, j >= 1).
//   gemm_f32_kernel   shape + pose blendshapes as ONE fp32-MFMA GEMM: v_posed[N, V*3] = [pose_feature(486) | beta(10) | 1] .
//                     [posedirs ; shapedirs ; v_template]  (K = 497 -> 512; 30.5 + 0.7 MFLOP per frame).
//   lbs_skin_mfma_kernel  DENSE skinning weights: the blend T = W[V, 55] . A[55, 12 F] is a dense contraction (13.8 MFLOP per
//                     frame, 660 fma per vertex-frame on the VALU before) -- on the matrix core: a workgroup keeps the 12 x 16
//                     transform rows of 16 frames in LDS, streams 144-vertex tiles of W through a double buffer, and applies
//                     v = T [p; 1] + transl in the epilogue.  Algorithmic traffic 12 B read + 12 B written per vertex-frame.
//   lbs_skin_ell_kernel   SPARSE skinning weights (a real SMPLX_NEUTRAL.npz: a handful of non-zero joints per vertex; chosen at
//                     rohm_smplx_set_skinning when >= 75 % of lbs_weights are zero): per vertex an ELL row (joint index, weight) of
//                     the widest vertex's length, one thread per vertex x 8 frames, transforms from LDS.
// Expression coefficients are taken as zero (every reference call site passes zeros, :383-388).
#include "common.h"
#include "smplx_fk.h"

namespace rohm {

constexpr int kMaxJ = 64;

__global__ __launch_bounds__(64) void lbs_pose_kernel(const float* __restrict__ pose, int n_pose, int pose_kind,
                                                      const float* __restrict__ betas, const float* __restrict__ Jt,
                                                      const float* __restrict__ Js, const int* __restrict__ parents,
                                                      int J, float* __restrict__ A, float* __restrict__ feat, int KP,
                                                      int N) {
    const int n = blockIdx.x * blockDim.x + threadIdx.x;
    if (n >= N) return;
    float beta[NBETA];
#pragma unroll
    for (int k = 0; k < NBETA; ++k) beta[k] = betas[(size_t)n * NBETA + k];
    float* An = A + (size_t)n * J * 12;          // pass 1: (G_j [9], P_j [3]); pass 2: P_j -> P_j - G_j J_j
    float* fn = feat + (size_t)n * KP;
    // columns (J-1)*9 .. +9: the shape coefficients, then a 1: the shape blend and v_template ride in the blendshape GEMM
    // (their rows follow posedirs in d_pdT)
    for (int k = (J - 1) * 9; k < KP; ++k) fn[k] = (k < (J - 1) * 9 + NBETA) ? beta[k - (J - 1) * 9] : (k == (J - 1) * 9 + NBETA ? 1.f : 0.f);
    auto rest = [&](int j, float* o) {
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            float v = Jt[j * 3 + c];
#pragma unroll
            for (int k = 0; k < NBETA; ++k) v = fmaf(Js[(j * 3 + c) * NBETA + k], beta[k], v);
            o[c] = v;
        }
    };
    for (int j = 0; j < J; ++j) {
        float R[9];
        if (j < n_pose && pose_kind == 1) {          // interleaved 6-D (quaternion.py:482-501), as the representation stores it
            float x6[6];
#pragma unroll
            for (int k = 0; k < 6; ++k) x6[k] = pose[((size_t)n * n_pose + j) * 6 + k];
            rot6d_fwd(x6, R);
        } else if (j < n_pose) {
            const float r[3] = {pose[((size_t)n * n_pose + j) * 3], pose[((size_t)n * n_pose + j) * 3 + 1],
                                pose[((size_t)n * n_pose + j) * 3 + 2]};
            rodrigues(r, R);
        } else {
#pragma unroll
            for (int i = 0; i < 9; ++i) R[i] = (i % 4 == 0) ? 1.f : 0.f;
        }
        if (j >= 1) {
#pragma unroll
            for (int i = 0; i < 9; ++i) fn[(j - 1) * 9 + i] = R[i] - ((i % 4 == 0) ? 1.f : 0.f);
        }
        float Jj[3];
        rest(j, Jj);
        float* o = An + j * 12;
        if (j == 0) {
#pragma unroll
            for (int i = 0; i < 9; ++i) o[i] = R[i];
#pragma unroll
            for (int c = 0; c < 3; ++c) o[9 + c] = Jj[c];
        } else {
            const int p = parents[j];
            const float* gp = An + p * 12;
            float Gp[9], Pp[3], Jp[3], G[9], w[3];
#pragma unroll
            for (int i = 0; i < 9; ++i) Gp[i] = gp[i];
#pragma unroll
            for (int c = 0; c < 3; ++c) Pp[c] = gp[9 + c];
            rest(p, Jp);
            mat_mul(Gp, R, G);
            const float off[3] = {Jj[0] - Jp[0], Jj[1] - Jp[1], Jj[2] - Jp[2]};
            mat_vec(Gp, off, w);
#pragma unroll
            for (int i = 0; i < 9; ++i) o[i] = G[i];
#pragma unroll
            for (int c = 0; c < 3; ++c) o[9 + c] = Pp[c] + w[c];
        }
    }
}

// joints out (posed + transl) and P_j -> t_j = P_j - G_j J_j; separate pass so children above read the parents' P_j
// Row of the skinning GEMM's transform operand that holds component `comp` (0..11) of frame `f`: groups of 16 frames = 192 rows; inside
// a group the row order makes one MFMA lane end up with all 12 components of ONE frame: wave w = f_in / 4 owns rows [48 w, 48 w + 48),
// column block c = comp / 4, lane group lg = f_in % 4, q = comp % 4  (see lbs_skin_mfma_kernel).
__host__ __device__ __forceinline__ size_t skin_row(int f, int comp) {
    const int g = f >> 4, fi = f & 15;
    return (size_t)g * 192 + (fi >> 2) * 48 + (comp >> 2) * 16 + (fi & 3) * 4 + (comp & 3);
}
constexpr int SKIN_K = 64;        // joints padded to two 32-wide K chunks (J <= 64)

__global__ __launch_bounds__(64) void lbs_finish_pose_kernel(const float* __restrict__ betas, const float* __restrict__ transl,
                                                             const float* __restrict__ Jt, const float* __restrict__ Js,
                                                             int J, float* __restrict__ A, float* __restrict__ Tm,
                                                             float* __restrict__ joints, int n_joints_out, int N) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= N * J) return;
    const int n = idx / J, j = idx % J;
    float* o = A + (size_t)idx * 12;
    float Jj[3];
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        float v = Jt[j * 3 + c];
#pragma unroll
        for (int k = 0; k < NBETA; ++k) v = fmaf(Js[(j * 3 + c) * NBETA + k], betas[(size_t)n * NBETA + k], v);
        Jj[c] = v;
    }
    if (joints && j < n_joints_out)
#pragma unroll
        for (int c = 0; c < 3; ++c) joints[((size_t)n * n_joints_out + j) * 3 + c] = o[9 + c] + transl[(size_t)n * 3 + c];
    float w[3];
    mat_vec(o, Jj, w);
#pragma unroll
    for (int c = 0; c < 3; ++c) o[9 + c] -= w[c];
    if (Tm) {      // the same 12 numbers as rows of the skinning GEMM's operand [frames x 12 (permuted), 64 joints]
#pragma unroll
        for (int c = 0; c < 12; ++c) Tm[skin_row(n, c) * SKIN_K + j] = o[c];
    }
}

// Skinning.  One thread per vertex, SKIN_F consecutive frames per block.  What belongs to the vertex -- its J skinning weights,
// v_template, the ten shape directions -- is loaded ONCE per block and reused for all its frames (a weight lives in one register
// while it is blended into the SKIN_F transforms the thread accumulates); what belongs to a frame -- the relative joint transforms
// A[n][j], betas, transl -- is staged in LDS and read as broadcasts (every lane wants the same 16 bytes).  Per vertex-frame that
// leaves 12 B read (off) + 12 B written against J x 12 = 660 v_fma: the VALU, not memory, bounds the kernel.  (Round 2: one block
// per frame, 364 B of L2 traffic per vertex-frame for the same 24 B of payload, 2.74 ms for 4576 frames.  Also measured and dropped:
// joint transforms through the scalar cache instead of LDS -- latency-bound, 4.1 ms.)  Sums run over j in ascending order with fmaf:
// bit-identical to the one-frame kernel.
constexpr int SKIN_F = 8;

__global__ __launch_bounds__(256) void lbs_skin_kernel(const float* __restrict__ vt, const float* __restrict__ sd,
                                                       const float* __restrict__ wT, const float* __restrict__ off, int ldo,
                                                       const float* __restrict__ A, const float* __restrict__ betas,
                                                       const float* __restrict__ transl, int J, int V, int N,
                                                       float* __restrict__ verts) {
    extern __shared__ __attribute__((aligned(16))) float sA[];       // [J][SKIN_F][12], then betas [SKIN_F][NBETA], transl [SKIN_F][3]
    const int n0 = blockIdx.y * SKIN_F;
    const int nf = (N - n0 < SKIN_F) ? N - n0 : SKIN_F;              // frames of this block (uniform)
    float* sb = sA + SKIN_F * J * 12;
    float* st = sb + SKIN_F * NBETA;
    // joint-major image: the SKIN_F transforms of joint j are contiguous (one run of broadcast reads per joint); frames past
    // the end of the batch are zero-filled (their results are not stored)
    for (int i = threadIdx.x; i < SKIN_F * J * 12; i += blockDim.x) {
        const int j = i / (SKIN_F * 12), r = i % (SKIN_F * 12), f = r / 12, c = r % 12;
        sA[i] = (f < nf) ? A[((size_t)(n0 + f) * J + j) * 12 + c] : 0.f;
    }
    for (int i = threadIdx.x; i < SKIN_F * NBETA; i += blockDim.x) sb[i] = (i < nf * NBETA) ? betas[(size_t)n0 * NBETA + i] : 0.f;
    for (int i = threadIdx.x; i < SKIN_F * 3; i += blockDim.x) st[i] = (i < nf * 3) ? transl[(size_t)n0 * 3 + i] : 0.f;
    __syncthreads();
    const int v = blockIdx.x * blockDim.x + threadIdx.x;
    if (v >= V) return;
    float T[SKIN_F][12];
#pragma unroll
    for (int f = 0; f < SKIN_F; ++f)
#pragma unroll
        for (int i = 0; i < 12; ++i) T[f][i] = 0.f;
    for (int j = 0; j < J; ++j) {
        const float w = wT[(size_t)j * V + v];
        const float* a = sA + j * (SKIN_F * 12);
#pragma unroll
        for (int f = 0; f < SKIN_F; ++f)
#pragma unroll
            for (int i = 0; i < 12; ++i) T[f][i] = fmaf(w, a[f * 12 + i], T[f][i]);
    }
    float base[3], sdv[3][NBETA];
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        base[c] = vt[v * 3 + c];
#pragma unroll
        for (int k = 0; k < NBETA; ++k) sdv[c][k] = sd[((size_t)v * 3 + c) * NBETA + k];
    }
#pragma unroll
    for (int f = 0; f < SKIN_F; ++f) {
        if (f < nf) {
            const int n = n0 + f;
            float pq[3];
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                float x = base[c];
#pragma unroll
                for (int k = 0; k < NBETA; ++k) x = fmaf(sdv[c][k], sb[f * NBETA + k], x);
                pq[c] = x + off[(size_t)n * ldo + v * 3 + c];
            }
#pragma unroll
            for (int c = 0; c < 3; ++c)
                verts[((size_t)n * V + v) * 3 + c] =
                    T[f][c * 3] * pq[0] + T[f][c * 3 + 1] * pq[1] + T[f][c * 3 + 2] * pq[2] + T[f][9 + c] + st[f * 3 + c];
        }
    }
}

// dst[c][r] = src[r][c] for r < R, c < Cc; dst rows padded to ldd (pre-zeroed)
__global__ void lbs_transpose_kernel(const float* __restrict__ src, int R, int Cc, float* __restrict__ dst, int ldd) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (long)R * Cc) return;
    const int r = (int)(i / Cc), c = (int)(i % Cc);
    dst[(size_t)c * ldd + r] = src[i];
}
__global__ void lbs_copy_shape_kernel(const float* __restrict__ src, int n_total, float* __restrict__ dst, long rows) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= rows * NBETA) return;
    dst[i] = src[(i / NBETA) * n_total + (i % NBETA)];
}

static inline size_t al64(size_t n) { return (n + 63) / 64 * 64; }

}  // namespace rohm

using namespace rohm;

extern "C" {

int rohm_smplx_set_skinning(rohm_smplx_t* h, const float* v_template, const float* shapedirs, int n_shape_total,
                            const float* posedirs, int n_pose_feat, const float* lbs_weights) {
    ROHM_ARG_CHECK(h && v_template && shapedirs && posedirs && lbs_weights, "smplx_set_skinning: null argument");
    ROHM_ARG_CHECK(n_pose_feat == (h->J - 1) * 9, "smplx_set_skinning: posedirs must have (J-1)*9 = %d rows (got %d)",
                   (h->J - 1) * 9, n_pose_feat);
    ROHM_ARG_CHECK(h->J <= kMaxJ && n_shape_total >= NBETA, "smplx_set_skinning: bad sizes");
    ROHM_HIP_CHECK(hipSetDevice(h->device));
    const int V = h->V, J = h->J;
    h->P = n_pose_feat;
    h->KP = (n_pose_feat + 31) / 32 * 32;
    h->NP = (V * 3 + 383) / 384 * 384;
    float *d_pd = nullptr, *d_w = nullptr, *d_sdin = nullptr;
    hipError_t e = hipSuccess;
    auto bad = [&](hipError_t err) {
        if (d_pd) (void)hipFree(d_pd);
        if (d_w) (void)hipFree(d_w);
        if (d_sdin) (void)hipFree(d_sdin);
        set_error("smplx_set_skinning: %s", hipGetErrorString(err));
        return ROHM_ERR_HIP;
    };
#define TRY(x) if ((e = (x)) != hipSuccess) return bad(e)
    if (!h->d_vt) TRY(hipMalloc(&h->d_vt, (size_t)V * 3 * sizeof(float)));
    if (!h->d_sd) TRY(hipMalloc(&h->d_sd, (size_t)V * 3 * NBETA * sizeof(float)));
    if (!h->d_pdT) TRY(hipMalloc(&h->d_pdT, (size_t)h->NP * h->KP * sizeof(float)));
    if (!h->d_wT) TRY(hipMalloc(&h->d_wT, (size_t)J * V * sizeof(float)));
    TRY(hipMalloc(&d_pd, (size_t)n_pose_feat * V * 3 * sizeof(float)));
    TRY(hipMalloc(&d_w, (size_t)V * J * sizeof(float)));
    TRY(hipMalloc(&d_sdin, (size_t)V * 3 * n_shape_total * sizeof(float)));
    TRY(hipMemcpy(h->d_vt, v_template, (size_t)V * 3 * sizeof(float), hipMemcpyDefault));
    TRY(hipMemcpy(d_sdin, shapedirs, (size_t)V * 3 * n_shape_total * sizeof(float), hipMemcpyDefault));
    TRY(hipMemcpy(d_pd, posedirs, (size_t)n_pose_feat * V * 3 * sizeof(float), hipMemcpyDefault));
    TRY(hipMemcpy(d_w, lbs_weights, (size_t)V * J * sizeof(float), hipMemcpyDefault));
    TRY(hipMemset(h->d_pdT, 0, (size_t)h->NP * h->KP * sizeof(float)));
    {
        const long n = (long)V * 3 * NBETA;
        hipLaunchKernelGGL(lbs_copy_shape_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, 0, d_sdin, n_shape_total,
                           h->d_sd, (long)V * 3);
        const long n2 = (long)n_pose_feat * V * 3;     // posedirs [P, V*3] -> [V*3 (pad NP), KP]
        hipLaunchKernelGGL(lbs_transpose_kernel, dim3((unsigned)((n2 + 255) / 256)), dim3(256), 0, 0, d_pd, n_pose_feat,
                           V * 3, h->d_pdT, h->KP);
        const long n3 = (long)V * J;                   // weights [V, J] -> [J, V]
        hipLaunchKernelGGL(lbs_transpose_kernel, dim3((unsigned)((n3 + 255) / 256)), dim3(256), 0, 0, d_w, V, J, h->d_wT, V);
    }
    TRY(hipDeviceSynchronize());
#undef TRY
    (void)hipFree(d_pd); (void)hipFree(d_w); (void)hipFree(d_sdin);
    return ROHM_OK;
}

size_t rohm_smplx_lbs_workspace_bytes(const rohm_smplx_t* h, int N) {
    if (!h || N <= 0 || !h->d_pdT) return 0;
    return (al64((size_t)N * h->KP) + al64((size_t)N * h->J * 12) + al64((size_t)N * h->NP)) * sizeof(float);
}

int rohm_smplx_forward(const rohm_smplx_t* h, const float* pose, int n_pose, int pose_kind, const float* betas, const float* transl,
                       int N, float* joints, int n_joints_out, float* verts, void* ws, size_t ws_bytes,
                       rohm_stream_t stream) {
    ROHM_ARG_CHECK(h && pose && betas && transl && ws, "smplx_forward: null argument");
    ROHM_ARG_CHECK(h->d_pdT, "smplx_forward: call rohm_smplx_set_skinning first (posedirs / lbs_weights)");
    ROHM_ARG_CHECK(n_pose >= 1 && n_pose <= h->J, "smplx_forward: n_pose must be in [1, %d]", h->J);
    ROHM_ARG_CHECK(pose_kind == 0 || pose_kind == 1, "smplx_forward: pose_kind must be 0 (axis-angle) or 1 (6-D)");
    ROHM_ARG_CHECK(!joints || (n_joints_out >= 1 && n_joints_out <= h->J), "smplx_forward: n_joints_out must be in [1, %d]", h->J);
    ROHM_ARG_CHECK(((uintptr_t)ws % 256) == 0, "smplx_forward: workspace must be 256-byte aligned");
    if (N <= 0) return ROHM_OK;
    ROHM_ARG_CHECK(N <= 65535, "smplx_forward: at most 65535 frames per call (got %d)", N);
    ROHM_ARG_CHECK(ws_bytes >= rohm_smplx_lbs_workspace_bytes(h, N), "smplx_forward: workspace too small");
    hipStream_t s = (hipStream_t)stream;
    float* feat = (float*)ws;
    float* A = feat + al64((size_t)N * h->KP);
    float* off = A + al64((size_t)N * h->J * 12);
    {
        prof::Scope ps("lbs_pose", 0.0, 4.0 * N * (n_pose * 3 + 10 + h->J * 12 + h->KP), s);
        hipLaunchKernelGGL(lbs_pose_kernel, dim3((N + 63) / 64), dim3(64), 0, s, pose, n_pose, pose_kind, betas, h->d_Jt, h->d_Js,
                           h->d_parents, h->J, A, feat, h->KP, N);
        const int nj = N * h->J;
        hipLaunchKernelGGL(lbs_finish_pose_kernel, dim3((nj + 63) / 64), dim3(64), 0, s, betas, transl, h->d_Jt, h->d_Js,
                           h->J, A, joints, n_joints_out, N);
        ROHM_LAUNCH_CHECK();
    }
    if (!verts) return ROHM_OK;
    GemmParams g{};
    g.A = feat; g.lda = h->KP; g.W = h->d_pdT; g.ldw = h->KP; g.C = off; g.ldc = h->NP;
    g.M = N; g.N = h->NP; g.K = h->KP; g.bias = nullptr;
    int rc = launch_gemm(g, EPI_BIAS, s);
    if (rc) return rc;
    prof::Scope ps("lbs_skin", 2.0 * N * h->V * (h->J * 12 + 30 + 12), 24.0 * N * h->V, s);
    const dim3 sgrid((h->V + 255) / 256, (N + SKIN_F - 1) / SKIN_F);
    const size_t slds = (size_t)SKIN_F * (h->J * 12 + NBETA + 3) * sizeof(float);      // 21.5 KB for 55 joints
    hipLaunchKernelGGL(lbs_skin_kernel, sgrid, dim3(256), slds, s, h->d_vt, h->d_sd, h->d_wT, off, h->NP, A, betas, transl, h->J,
                       h->V, N, verts);
    ROHM_LAUNCH_CHECK();
    return ROHM_OK;
}

}  // extern "C"
