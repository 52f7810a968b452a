// SMPL-X joints-only forward kinematics shared by the guidance kernels (smplx.hip) and the trajectory
// re-derivation (rederive.hip): 6-D -> R (Gram-Schmidt), Rodrigues, rest joints from the folded regressor,
// the kinematic chain and its hand-derived reverse pass.  Reference citations are on each function.
#pragma once
#include "common.h"


namespace rohm {
constexpr int NJ = 22;          // body joints used by the hot path
constexpr int NBETA = 10;
constexpr int C_TOTAL = 294;    // utils/other_utils.py:17-37
// channel offsets of the 294-d representation
constexpr int CH_ROOT_ANG = 0, CH_ROOT_POS = 2, CH_ROOT_H = 6, CH_ROT6D = 7, CH_TRANS = 16, CH_LOCAL = 22,
              CH_POSE6D = 154, CH_BETAS = 280, CH_CONTACT = 290;
}  // namespace rohm

struct rohm_smplx {
    int V, J, device;
    float* d_Jt;       // [J, 3]        J_regressor . v_template
    float* d_Js;       // [J, 3, 10]    J_regressor . shapedirs[:, :, :10]
    int* d_parents;    // [J]
    int parents[64];
    int max_depth22;   // depth of the 22-joint body tree (guidance kernels run it level by level)
    // ---- full linear blend skinning (lbs.hip; optional, set by rohm_smplx_set_skinning) ----
    int P, KP, NP;     // pose-blendshape rows (J-1)*9, padded to a multiple of 32; V*3 padded to a multiple of 384
    float* d_vt;       // [V, 3]          v_template
    float* d_sd;       // [V, 3, 10]      shapedirs[:, :, :10]
    float* d_pdT;      // [NP, KP]        posedirs transposed (GEMM weight layout), zero padded
    float* d_wT;       // [tiles_v * 144, 64]  lbs_weights, joints padded to 64, rows to whole 144-vertex tiles (MFMA skinning operand)
    int tiles_v;       // 144-vertex tiles
    int skin_mode;     // 0 = dense weights on the matrix core, 1 = sparse weights (ELL rows of the non-zeros), 2 = ELL rows of all joints
    int ell_width;     // entries per ELL row
    int* d_ell_j;      // [ell_width, V]  joint indices
    float* d_ell_w;    // [ell_width, V]  weights (0 = padding)
    float* d_zero_bias; // [NP] zeros: the blendshape GEMM's bias operand
};

namespace rohm {

struct Mat3 { float m[9]; };

__device__ __forceinline__ void mat_mul(const float* a, const float* b, float* c) {   // c = a b
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j) c[i * 3 + j] = a[i * 3] * b[j] + a[i * 3 + 1] * b[3 + j] + a[i * 3 + 2] * b[6 + j];
}
__device__ __forceinline__ void mat_vec(const float* a, const float* v, float* o) {   // o = a v
#pragma unroll
    for (int i = 0; i < 3; ++i) o[i] = a[i * 3] * v[0] + a[i * 3 + 1] * v[1] + a[i * 3 + 2] * v[2];
}
__device__ __forceinline__ void matT_vec(const float* a, const float* v, float* o) {  // o = a^T v
#pragma unroll
    for (int i = 0; i < 3; ++i) o[i] = a[i] * v[0] + a[3 + i] * v[1] + a[6 + i] * v[2];
}

// Gram-Schmidt of the interleaved 6-D vector x = (a1x a2x a1y a2y a1z a2z) (quaternion.py:494-501).
// R columns are b1, b2, b3; R is row-major.
__device__ __forceinline__ void rot6d_fwd(const float* x, float* R) {
    const float a1[3] = {x[0], x[2], x[4]}, a2[3] = {x[1], x[3], x[5]};
    const float n1 = fmaxf(sqrtf(a1[0] * a1[0] + a1[1] * a1[1] + a1[2] * a1[2]), 1e-12f);
    const float b1[3] = {a1[0] / n1, a1[1] / n1, a1[2] / n1};
    const float d = b1[0] * a2[0] + b1[1] * a2[1] + b1[2] * a2[2];
    const float u[3] = {a2[0] - d * b1[0], a2[1] - d * b1[1], a2[2] - d * b1[2]};
    const float n2 = fmaxf(sqrtf(u[0] * u[0] + u[1] * u[1] + u[2] * u[2]), 1e-12f);
    const float b2[3] = {u[0] / n2, u[1] / n2, u[2] / n2};
    const float b3[3] = {b1[1] * b2[2] - b1[2] * b2[1], b1[2] * b2[0] - b1[0] * b2[2], b1[0] * b2[1] - b1[1] * b2[0]};
#pragma unroll
    for (int i = 0; i < 3; ++i) { R[i * 3] = b1[i]; R[i * 3 + 1] = b2[i]; R[i * 3 + 2] = b3[i]; }
}

// Reverse pass of rot6d_fwd: dR (row-major, dL/dR) -> dx[6].
__device__ __forceinline__ void rot6d_bwd(const float* x, const float* dR, float* dx) {
    const float a1[3] = {x[0], x[2], x[4]}, a2[3] = {x[1], x[3], x[5]};
    const float n1 = fmaxf(sqrtf(a1[0] * a1[0] + a1[1] * a1[1] + a1[2] * a1[2]), 1e-12f);
    const float b1[3] = {a1[0] / n1, a1[1] / n1, a1[2] / n1};
    const float d = b1[0] * a2[0] + b1[1] * a2[1] + b1[2] * a2[2];
    const float u[3] = {a2[0] - d * b1[0], a2[1] - d * b1[1], a2[2] - d * b1[2]};
    const float n2 = fmaxf(sqrtf(u[0] * u[0] + u[1] * u[1] + u[2] * u[2]), 1e-12f);
    const float b2[3] = {u[0] / n2, u[1] / n2, u[2] / n2};
    float g1[3] = {dR[0], dR[3], dR[6]}, g2[3] = {dR[1], dR[4], dR[7]};
    const float g3[3] = {dR[2], dR[5], dR[8]};
    // b3 = b1 x b2:  db1 += b2 x g3,  db2 += g3 x b1
    g1[0] += b2[1] * g3[2] - b2[2] * g3[1]; g1[1] += b2[2] * g3[0] - b2[0] * g3[2]; g1[2] += b2[0] * g3[1] - b2[1] * g3[0];
    g2[0] += g3[1] * b1[2] - g3[2] * b1[1]; g2[1] += g3[2] * b1[0] - g3[0] * b1[2]; g2[2] += g3[0] * b1[1] - g3[1] * b1[0];
    // b2 = u / |u|
    const float s2 = b2[0] * g2[0] + b2[1] * g2[1] + b2[2] * g2[2];
    const float du[3] = {(g2[0] - s2 * b2[0]) / n2, (g2[1] - s2 * b2[1]) / n2, (g2[2] - s2 * b2[2]) / n2};
    // u = a2 - (b1.a2) b1
    const float sb = du[0] * b1[0] + du[1] * b1[1] + du[2] * b1[2];
    const float da2[3] = {du[0] - sb * b1[0], du[1] - sb * b1[1], du[2] - sb * b1[2]};
    // d/db1 of u = a2 - (b1.a2) b1 contracted with du:  -(d du + (b1.du) a2)
#pragma unroll
    for (int i = 0; i < 3; ++i) g1[i] += -d * du[i] - sb * a2[i];
    // b1 = a1 / |a1|
    const float s1 = b1[0] * g1[0] + b1[1] * g1[1] + b1[2] * g1[2];
    const float da1[3] = {(g1[0] - s1 * b1[0]) / n1, (g1[1] - s1 * b1[1]) / n1, (g1[2] - s1 * b1[2]) / n1};
    dx[0] = da1[0]; dx[2] = da1[1]; dx[4] = da1[2];
    dx[1] = da2[0]; dx[3] = da2[1]; dx[5] = da2[2];
}

// Rodrigues as smplx.lbs.batch_rodrigues: angle = |r + 1e-8|, R = I + sin K + (1 - cos) K^2.
__device__ __forceinline__ void rodrigues(const float* r, float* R) {
    const float ex = r[0] + 1e-8f, ey = r[1] + 1e-8f, ez = r[2] + 1e-8f;
    const float ang = sqrtf(ex * ex + ey * ey + ez * ez);
    const float x = r[0] / ang, y = r[1] / ang, z = r[2] / ang;
    const float s = sinf(ang), c1 = 1.f - cosf(ang);
    // K = [[0,-z,y],[z,0,-x],[-y,x,0]];  K^2 = r r^T - |dir|^2 I (dir may be slightly non-unit, keep exact form)
    const float K[9] = {0.f, -z, y, z, 0.f, -x, -y, x, 0.f};
    float K2[9];
    mat_mul(K, K, K2);
#pragma unroll
    for (int i = 0; i < 9; ++i) R[i] = ((i % 4 == 0) ? 1.f : 0.f) + s * K[i] + c1 * K2[i];
}

struct FkCtx {
    float R[NJ][9];     // local rotations
    float G[NJ][9];     // world rotations
    float Jr[NJ][3];    // rest joints for this frame's betas
    float P[NJ][3];     // posed joints (without transl)
};

__device__ __forceinline__ void rest_joints(const float* __restrict__ Jt, const float* __restrict__ Js,
                                            const float* beta, float (*Jr)[3]) {
    for (int j = 0; j < NJ; ++j)
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            float v = Jt[j * 3 + c];
#pragma unroll
            for (int k = 0; k < NBETA; ++k) v = fmaf(Js[(j * 3 + c) * NBETA + k], beta[k], v);
            Jr[j][c] = v;
        }
}

__device__ __forceinline__ void fk_forward(FkCtx& f, const int* __restrict__ parents) {
#pragma unroll
    for (int i = 0; i < 9; ++i) f.G[0][i] = f.R[0][i];
#pragma unroll
    for (int c = 0; c < 3; ++c) f.P[0][c] = f.Jr[0][c];
    for (int j = 1; j < NJ; ++j) {
        const int p = parents[j];
        mat_mul(f.G[p], f.R[j], f.G[j]);
        const float off[3] = {f.Jr[j][0] - f.Jr[p][0], f.Jr[j][1] - f.Jr[p][1], f.Jr[j][2] - f.Jr[p][2]};
        float w[3];
        mat_vec(f.G[p], off, w);
#pragma unroll
        for (int c = 0; c < 3; ++c) f.P[j][c] = f.P[p][c] + w[c];
    }
}

// Reverse pass of fk_forward.  gP[j] = dL/dP[j] on entry (overwritten).  Outputs dR[j] (dL/dR_j, j >= 1; the
// global orientation lives in a zeroed channel range) and dJr (dL/dJrest).
__device__ __forceinline__ void fk_backward(const FkCtx& f, const int* __restrict__ parents, float (*gP)[3],
                                            float (*dR)[9], float (*dJr)[3]) {
    float dG[NJ][9];
    for (int j = 0; j < NJ; ++j) {
#pragma unroll
        for (int i = 0; i < 9; ++i) dG[j][i] = 0.f;
#pragma unroll
        for (int c = 0; c < 3; ++c) dJr[j][c] = 0.f;
    }
    for (int j = NJ - 1; j >= 1; --j) {
        const int p = parents[j];
        // P[j] = P[p] + G[p] off
        const float off[3] = {f.Jr[j][0] - f.Jr[p][0], f.Jr[j][1] - f.Jr[p][1], f.Jr[j][2] - f.Jr[p][2]};
        float doff[3];
        matT_vec(f.G[p], gP[j], doff);
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            gP[p][c] += gP[j][c];
            dJr[j][c] += doff[c];
            dJr[p][c] -= doff[c];
        }
#pragma unroll
        for (int a = 0; a < 3; ++a)
#pragma unroll
            for (int b = 0; b < 3; ++b) dG[p][a * 3 + b] += gP[j][a] * off[b];
        // G[j] = G[p] R[j]:  dG[p] += dG[j] R[j]^T,  dR[j] = G[p]^T dG[j]
#pragma unroll
        for (int a = 0; a < 3; ++a)
#pragma unroll
            for (int b = 0; b < 3; ++b) {
                float s = 0.f, r = 0.f;
#pragma unroll
                for (int k = 0; k < 3; ++k) {
                    s += dG[j][a * 3 + k] * f.R[j][b * 3 + k];
                    r += f.G[p][k * 3 + a] * dG[j][k * 3 + b];
                }
                dG[p][a * 3 + b] += s;
                dR[j][a * 3 + b] = r;
            }
    }
#pragma unroll
    for (int c = 0; c < 3; ++c) dJr[0][c] += gP[0][c];
}

// Loads + de-normalises the channels the SMPL-X path needs for frame (b, t) and runs FK.
struct FrameIn {
    float x6[NJ][6];     // 6-D rotations (0 = global orient)
    float beta[NBETA];
    float trans[3];
};

__device__ __forceinline__ float ld(const float* __restrict__ x0, const float* __restrict__ mean,
                                    const float* __restrict__ stdv, size_t base, int T, int c) {
    return x0[base + (size_t)c * T] * stdv[c] + mean[c];
}

__device__ __forceinline__ void load_frame(const float* __restrict__ x0, const float* __restrict__ mean,
                                           const float* __restrict__ stdv, size_t base, int T, FrameIn& in) {
#pragma unroll
    for (int k = 0; k < 6; ++k) in.x6[0][k] = ld(x0, mean, stdv, base, T, CH_ROT6D + k);
    for (int j = 1; j < NJ; ++j)
#pragma unroll
        for (int k = 0; k < 6; ++k) in.x6[j][k] = ld(x0, mean, stdv, base, T, CH_POSE6D + (j - 1) * 6 + k);
#pragma unroll
    for (int k = 0; k < NBETA; ++k) in.beta[k] = ld(x0, mean, stdv, base, T, CH_BETAS + k);
#pragma unroll
    for (int k = 0; k < 3; ++k) in.trans[k] = ld(x0, mean, stdv, base, T, CH_TRANS + k);
}

__device__ __forceinline__ void smplx_fk(const FrameIn& in, const float* Jt, const float* Js, const int* parents,
                                         FkCtx& f) {
    for (int j = 0; j < NJ; ++j) rot6d_fwd(in.x6[j], f.R[j]);
    rest_joints(Jt, Js, in.beta, f.Jr);
    fk_forward(f, parents);
}

// abs-trajectory joint j >= 1 (recover_from_repr_smpl 'joint_abs_traj'): qrot(qinv(q), v) + (x, y, 0),
// q = (cos a, 0, 0, sin a).  With u = (0, 0, -sin a), w = cos a:  v' = v + 2 (w (u x v) + u x (u x v)).
__device__ __forceinline__ void abs_joint(float ang, const float* pos, const float* v, float* o) {
    const float w = cosf(ang), uz = -sinf(ang);
    const float uv[3] = {-uz * v[1], uz * v[0], 0.f};
    const float uuv[3] = {-uz * uv[1], uz * uv[0], 0.f};
    o[0] = v[0] + 2.f * (w * uv[0] + uuv[0]) + pos[0];
    o[1] = v[1] + 2.f * (w * uv[1] + uuv[1]) + pos[1];
    o[2] = v[2];
}
// transpose of the linear map above applied to g (gradient wrt v)
__device__ __forceinline__ void abs_joint_T(float ang, const float* g, float* o) {
    const float w = cosf(ang), uz = sinf(ang);   // conjugate quaternion
    const float uv[3] = {-uz * g[1], uz * g[0], 0.f};
    const float uuv[3] = {-uz * uv[1], uz * uv[0], 0.f};
    o[0] = g[0] + 2.f * (w * uv[0] + uuv[0]);
    o[1] = g[1] + 2.f * (w * uv[1] + uuv[1]);
    o[2] = g[2];
}

static __constant__ int kFoot[4] = {7, 10, 8, 11};                       // model/posenet.py:31
static __constant__ int kProj[10] = {16, 18, 20, 17, 19, 21, 4, 5, 7, 8}; // model/posenet.py:308

}  // namespace rohm
