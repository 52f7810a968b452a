// SMPL-X joints-only forward kinematics and the two test-time guidance gradients, with an ANALYTIC backward.
//
// Reference: model/posenet.py:196-317 (guide_skating_with_smpl, guide_2d_projection_with_smpl),
// data_loaders/motion_representation.py:285-398 (recover_from_repr_smpl), data_loaders/common/
// quaternion.py:482-501 (rot6d_to_rotmat) and the third-party smplx==0.1.28 `lbs`.
//
// What the reference does per guided step: full SMPL-X LBS (10 475 vertices, 54 MFLOP / frame) under
// autograd, although the losses read 4 (skating) or 10 (re-projection) of the first 22 joints.  Those joints
// depend only on  J_rest = J_regressor.v_template + (J_regressor.shapedirs).beta  and the kinematic chain, so:
//   * create():  fold J_regressor into a [J,3] template and a [J,3,10] shape basis once (fp64 accumulate);
//   * per step:  one thread per frame does 6-D -> R (Gram-Schmidt), FK, loss gradient and the hand-derived
//                reverse pass, reading the [B, 294, 1, T] tensor with T-contiguous (coalesced) accesses and
//                writing the full 294-channel gradient column (zeros where the reference zeroes).
// The reference goes R -> quaternion -> axis-angle -> Rodrigues between Gram-Schmidt and FK; that round trip is
// the identity on SO(3) and its Jacobian restricted to the tangent space is the identity too, so FK consumes
// the Gram-Schmidt matrices directly (agreement with the full chain is checked against the oracle).
#include <vector>
#include "common.h"
#include "smplx_fk.h"

namespace rohm {


// ---------------------------------------------------------------------------------------- skating guidance
// pass 1: foot joints of both recoveries for every frame -> feet[B, T, 2, 4, 3]
__global__ __launch_bounds__(64) void skating_fwd_kernel(const float* __restrict__ x0, const float* __restrict__ mean,
                                                         const float* __restrict__ stdv, const float* __restrict__ Jt,
                                                         const float* __restrict__ Js, const int* __restrict__ parents,
                                                         float* __restrict__ feet, int B, int T) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= B * T) return;
    const int b = idx / T, t = idx % T;
    const size_t base = (size_t)b * C_TOTAL * T + t;
    FrameIn in;
    load_frame(x0, mean, stdv, base, T, in);
    FkCtx f;
    smplx_fk(in, Jt, Js, parents, f);
    const float ang = ld(x0, mean, stdv, base, T, CH_ROOT_ANG);
    const float pos[3] = {ld(x0, mean, stdv, base, T, CH_ROOT_POS), ld(x0, mean, stdv, base, T, CH_ROOT_POS + 1),
                          ld(x0, mean, stdv, base, T, CH_ROOT_H)};
    float* o = feet + (size_t)idx * 24;
    for (int k = 0; k < 4; ++k) {
        const int j = kFoot[k];
        const float v[3] = {ld(x0, mean, stdv, base, T, CH_LOCAL + 3 * j), ld(x0, mean, stdv, base, T, CH_LOCAL + 3 * j + 1),
                            ld(x0, mean, stdv, base, T, CH_LOCAL + 3 * j + 2)};
        float a[3];
        abs_joint(ang, pos, v, a);
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            o[k * 3 + c] = a[c];
            o[12 + k * 3 + c] = f.P[j][c] + in.trans[c];
        }
    }
}

// pass 2: mask counts of both recoveries (posenet.py:224-246).  counts[0] = abs-traj, counts[1] = smplx.
__global__ __launch_bounds__(256) void skating_count_kernel(const float* __restrict__ x0, const float* __restrict__ mean,
                                                            const float* __restrict__ stdv,
                                                            const float* __restrict__ feet, float* __restrict__ counts,
                                                            int B, int T) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    float c0 = 0.f, c1 = 0.f;
    if (idx < B * (T - 1) * 4) {
        const int k = idx % 4, t = (idx / 4) % (T - 1), b = idx / (4 * (T - 1));
        const size_t base = (size_t)b * C_TOTAL * T + t;
        const bool contact = ld(x0, mean, stdv, base, T, CH_CONTACT + k) > 0.5f;
        const float* f0 = feet + ((size_t)b * T + t) * 24;
        const float* f1 = f0 + 24;
#pragma unroll
        for (int path = 0; path < 2; ++path) {
            const float vx = (f1[path * 12 + k * 3] - f0[path * 12 + k * 3]) * 30.f;
            const float vy = (f1[path * 12 + k * 3 + 1] - f0[path * 12 + k * 3 + 1]) * 30.f;
            const float vz = (f1[path * 12 + k * 3 + 2] - f0[path * 12 + k * 3 + 2]) * 30.f;
            const float n = sqrtf(vx * vx + vy * vy + vz * vz);
            const float m = (contact && (n - 0.1f > 0.f)) ? 1.f : 0.f;
            if (path == 0) c0 = m; else c1 = m;
        }
    }
    // counts are small integers: fp32 atomic accumulation is exact and order-independent
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { c0 += __shfl_xor(c0, o); c1 += __shfl_xor(c1, o); }
    if ((threadIdx.x & 63) == 0) {
        if (c0 != 0.f) atomicAdd(&counts[0], c0);
        if (c1 != 0.f) atomicAdd(&counts[1], c1);
    }
}

// d(-loss)/dJ_foot for frame t of one recovery: the frame appears in the pairs (t-1, t) and (t, t+1).
__device__ __forceinline__ void skating_djoint(const float* __restrict__ x0, const float* __restrict__ mean,
                                               const float* __restrict__ stdv, const float* __restrict__ feet,
                                               int b, int t, int T, int path, int k, float inv_cnt, float* g) {
    g[0] = g[1] = g[2] = 0.f;
    if (inv_cnt == 0.f) return;
#pragma unroll
    for (int s = 0; s < 2; ++s) {
        const int t0 = t - 1 + s;                 // pair (t0, t0 + 1)
        if (t0 < 0 || t0 + 1 >= T) continue;
        const size_t base = (size_t)b * C_TOTAL * T + t0;
        if (!(ld(x0, mean, stdv, base, T, CH_CONTACT + k) > 0.5f)) continue;
        const float* f0 = feet + ((size_t)b * T + t0) * 24 + path * 12 + k * 3;
        const float* f1 = f0 + 24;
        const float v[3] = {(f1[0] - f0[0]) * 30.f, (f1[1] - f0[1]) * 30.f, (f1[2] - f0[2]) * 30.f};
        const float n = sqrtf(v[0] * v[0] + v[1] * v[1] + v[2] * v[2]);
        if (!(n - 0.1f > 0.f)) continue;
        // loss = sum(mask n) / cnt;  dn/dJ[t0+1] = +30 v/n, dn/dJ[t0] = -30 v/n;  we return d(-loss)
        const float sgn = (s == 0) ? -1.f : 1.f;   // this frame is J[t0+1] when s == 0
        const float sc = sgn * 30.f * inv_cnt / n;
#pragma unroll
        for (int c = 0; c < 3; ++c) g[c] += sc * v[c];
    }
}

// Writes one gradient column (all 294 channels of frame (b, t)) from the SMPL-X reverse pass results.
__device__ __forceinline__ void write_column(float* __restrict__ grad, const float* __restrict__ stdv, size_t base, int T,
                                             const float* dlocal /*[NJ*3] or null*/, const float (*d6)[6],
                                             const float* dbeta) {
    for (int c = 0; c < C_TOTAL; ++c) {
        float v = 0.f;
        if (c >= CH_LOCAL && c < CH_LOCAL + NJ * 3) v = dlocal ? dlocal[c - CH_LOCAL] : 0.f;
        else if (c >= CH_POSE6D && c < CH_POSE6D + 126) v = d6[1 + (c - CH_POSE6D) / 6][(c - CH_POSE6D) % 6];
        else if (c >= CH_BETAS && c < CH_BETAS + NBETA) v = dbeta[c - CH_BETAS];
        // channels [0, 22) and [290, 294) are zeroed by the reference (posenet.py:251-252); local_vel never
        // enters either recovery.
        grad[base + (size_t)c * T] = v * stdv[c];
    }
}

// pass 3: gradient column of every frame
__global__ __launch_bounds__(64) void skating_bwd_kernel(const float* __restrict__ x0, const float* __restrict__ mean,
                                                         const float* __restrict__ stdv, const float* __restrict__ Jt,
                                                         const float* __restrict__ Js, const int* __restrict__ parents,
                                                         const float* __restrict__ feet, const float* __restrict__ counts,
                                                         float* __restrict__ grad, int B, int T) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= B * T) return;
    const int b = idx / T, t = idx % T;
    const size_t base = (size_t)b * C_TOTAL * T + t;
    const float inv0 = counts[0] > 0.f ? 1.f / counts[0] : 0.f;
    const float inv1 = counts[1] > 0.f ? 1.f / counts[1] : 0.f;
    // abs-trajectory recovery: only local_positions of the foot joints receive gradient
    float dlocal[NJ * 3];
    for (int i = 0; i < NJ * 3; ++i) dlocal[i] = 0.f;
    const float ang = ld(x0, mean, stdv, base, T, CH_ROOT_ANG);
    for (int k = 0; k < 4; ++k) {
        float g[3], gl[3];
        skating_djoint(x0, mean, stdv, feet, b, t, T, 0, k, inv0, g);
        abs_joint_T(ang, g, gl);
#pragma unroll
        for (int c = 0; c < 3; ++c) dlocal[kFoot[k] * 3 + c] = gl[c];
    }
    // SMPL-X recovery
    FrameIn in;
    load_frame(x0, mean, stdv, base, T, in);
    FkCtx f;
    smplx_fk(in, Jt, Js, parents, f);
    float gP[NJ][3];
    for (int j = 0; j < NJ; ++j) gP[j][0] = gP[j][1] = gP[j][2] = 0.f;
    for (int k = 0; k < 4; ++k) skating_djoint(x0, mean, stdv, feet, b, t, T, 1, k, inv1, gP[kFoot[k]]);
    float dR[NJ][9], dJr[NJ][3], d6[NJ][6], dbeta[NBETA];
    fk_backward(f, parents, gP, dR, dJr);
    for (int j = 1; j < NJ; ++j) rot6d_bwd(in.x6[j], dR[j], d6[j]);
    for (int k = 0; k < NBETA; ++k) {
        float s = 0.f;
        for (int j = 0; j < NJ; ++j)
#pragma unroll
            for (int c = 0; c < 3; ++c) s = fmaf(dJr[j][c], Js[(j * 3 + c) * NBETA + k], s);
        dbeta[k] = s;
    }
    write_column(grad, stdv, base, T, dlocal, d6, dbeta);
}

// ---------------------------------------------------------------------------------------- 2-D re-projection
__device__ __forceinline__ void inv3(const float* a, float* o) {
    const float c00 = a[4] * a[8] - a[5] * a[7], c01 = a[5] * a[6] - a[3] * a[8], c02 = a[3] * a[7] - a[4] * a[6];
    const float det = a[0] * c00 + a[1] * c01 + a[2] * c02;
    const float id = 1.f / det;
    o[0] = c00 * id; o[1] = (a[2] * a[7] - a[1] * a[8]) * id; o[2] = (a[1] * a[5] - a[2] * a[4]) * id;
    o[3] = c01 * id; o[4] = (a[0] * a[8] - a[2] * a[6]) * id; o[5] = (a[2] * a[3] - a[0] * a[5]) * id;
    o[6] = c02 * id; o[7] = (a[1] * a[6] - a[0] * a[7]) * id; o[8] = (a[0] * a[4] - a[1] * a[3]) * id;
}

// cam2[b] = {Rc (3x3 of inv(transf_matrix)), Tc (3), Ri = inv(cam_R) (9)}: 21 floats per clip.
// transf_matrix is affine ([R t; 0 1]), so inv = [R^-1, -R^-1 t] (torch.linalg.inv, posenet.py:286).
__global__ void proj_prep_kernel(const float* __restrict__ transf, const float* __restrict__ camR, float* __restrict__ cam2,
                                 int B) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= B) return;
    const float* m = transf + (size_t)b * 16;
    const float R[9] = {m[0], m[1], m[2], m[4], m[5], m[6], m[8], m[9], m[10]};
    const float tr[3] = {m[3], m[7], m[11]};
    float* o = cam2 + (size_t)b * 21;
    inv3(R, o);
    float tt[3];
    mat_vec(o, tr, tt);
    o[9] = -tt[0]; o[10] = -tt[1]; o[11] = -tt[2];
    inv3(camR, o + 12);
}

__global__ __launch_bounds__(64) void proj2d_kernel(const float* __restrict__ x0, const float* __restrict__ mean,
                                                    const float* __restrict__ stdv, const float* __restrict__ Jt,
                                                    const float* __restrict__ Js, const int* __restrict__ parents,
                                                    const float* __restrict__ cam2, const float* __restrict__ camT,
                                                    const float* __restrict__ focal, const float* __restrict__ center,
                                                    const float* __restrict__ kp2d, int kp_frames,
                                                    float* __restrict__ grad, int B, int T) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= B * T) return;
    const int b = idx / T, t = idx % T;
    const size_t base = (size_t)b * C_TOTAL * T + t;
    FrameIn in;
    load_frame(x0, mean, stdv, base, T, in);
    FkCtx f;
    smplx_fk(in, Jt, Js, parents, f);
    const float* Rc = cam2 + (size_t)b * 21;
    const float* Tc = Rc + 9;
    const float* Ri = Rc + 12;
    const float fx = focal[b * 2], fy = focal[b * 2 + 1], cx = center[b * 2], cy = center[b * 2 + 1];
    const float inv_n = 1.f / ((float)B * (float)T * 20.f);      // .mean() over B*T*10*2 (posenet.py:308-309)
    float gP[NJ][3];
    for (int j = 0; j < NJ; ++j) gP[j][0] = gP[j][1] = gP[j][2] = 0.f;
    for (int k = 0; k < 10; ++k) {
        const int j = kProj[k];
        const float p[3] = {f.P[j][0] + in.trans[0], f.P[j][1] + in.trans[1], f.P[j][2] + in.trans[2]};
        float sc[3], cm[3];
        mat_vec(Rc, p, sc);
        const float d[3] = {sc[0] + Tc[0] - camT[0], sc[1] + Tc[1] - camT[1], sc[2] + Tc[2] - camT[2]};
        mat_vec(Ri, d, cm);
        const float iz = 1.f / cm[2];
        const float u = fx * (cm[0] * iz) + cx, v = fy * (cm[1] * iz) + cy;
        const float* kp = kp2d + (((size_t)b * kp_frames + t) * NJ + j) * 3;
        const float conf = kp[2];
        const float du = u - kp[0], dv = v - kp[1];
        // d(-mean |.| conf): -sign(diff) conf / N
        const float gu = -((du > 0.f) - (du < 0.f)) * conf * inv_n;
        const float gv = -((dv > 0.f) - (dv < 0.f)) * conf * inv_n;
        const float gc[3] = {gu * fx * iz, gv * fy * iz, -(gu * fx * cm[0] + gv * fy * cm[1]) * iz * iz};
        float gs[3];
        matT_vec(Ri, gc, gs);
        matT_vec(Rc, gs, gP[j]);
    }
    float dR[NJ][9], dJr[NJ][3], d6[NJ][6], dbeta[NBETA];
    fk_backward(f, parents, gP, dR, dJr);
    for (int j = 1; j < NJ; ++j) rot6d_bwd(in.x6[j], dR[j], d6[j]);
    for (int k = 0; k < NBETA; ++k) {
        float s = 0.f;
        for (int j = 0; j < NJ; ++j)
#pragma unroll
            for (int c = 0; c < 3; ++c) s = fmaf(dJr[j][c], Js[(j * 3 + c) * NBETA + k], s);
        dbeta[k] = s;
    }
    write_column(grad, stdv, base, T, nullptr, d6, dbeta);
}

// ---------------------------------------------------------------------------------------- body-model forward
// joints[n, 0:n_out] of smplx.SMPLX.forward from axis-angle poses (motion_representation.py:379-396):
// pose [N, n_pose, 3] (global orient first, joints >= n_pose have zero rotation), betas [N,10], transl [N,3].
__global__ __launch_bounds__(64) void smplx_joints_kernel(const float* __restrict__ pose, int n_pose,
                                                          const float* __restrict__ betas,
                                                          const float* __restrict__ transl,
                                                          const float* __restrict__ Jt, const float* __restrict__ Js,
                                                          const int* __restrict__ parents, float* __restrict__ out,
                                                          int n_out, int N) {
    const int n = blockIdx.x * blockDim.x + threadIdx.x;
    if (n >= N) return;
    FkCtx f;
    for (int j = 0; j < NJ; ++j) {
        float r[3] = {0.f, 0.f, 0.f};
        if (j < n_pose) { r[0] = pose[((size_t)n * n_pose + j) * 3]; r[1] = pose[((size_t)n * n_pose + j) * 3 + 1]; r[2] = pose[((size_t)n * n_pose + j) * 3 + 2]; }
        rodrigues(r, f.R[j]);
    }
    float beta[NBETA];
    for (int k = 0; k < NBETA; ++k) beta[k] = betas[(size_t)n * NBETA + k];
    rest_joints(Jt, Js, beta, f.Jr);
    fk_forward(f, parents);
    for (int j = 0; j < n_out; ++j)
#pragma unroll
        for (int c = 0; c < 3; ++c) out[((size_t)n * n_out + j) * 3 + c] = f.P[j][c] + transl[(size_t)n * 3 + c];
}


// ---- dataset-side per-frame work -------------------------------------------------------------
// data_loaders/dataloader_video.py:121-142 / :282-300 run, for every frame of a recording: one SMPL-X forward ->
// joints in camera coordinates -> joints to world -> update_globalRT_for_smplx (utils/other_utils.py:221-240: the
// global orientation / translation re-expressed in the world frame; float64 numpy + scipy Rotation).  Here: one thread
// per frame, FK in float32 (as smplx computes it), the rigid-transform part in float64 with scipy's own formulas
// (from_rotvec / from_matrix / as_rotvec incl. their small-angle series) so results agree to rounding.
__device__ __forceinline__ void rotvec_to_matrix_f64(const double* rv, double* M) {
    const double a2 = rv[0] * rv[0] + rv[1] * rv[1] + rv[2] * rv[2];
    const double a = sqrt(a2);
    const double sc = (a <= 1e-3) ? 0.5 - a2 / 48.0 + a2 * a2 / 3840.0 : sin(a / 2.0) / a;
    const double x = sc * rv[0], y = sc * rv[1], z = sc * rv[2], w = cos(a / 2.0);
    const double x2 = x * x, y2 = y * y, z2 = z * z, w2 = w * w;
    const double xy = x * y, zw = z * w, xz = x * z, yw = y * w, yz = y * z, xw = x * w;
    M[0] = x2 - y2 - z2 + w2; M[1] = 2 * (xy - zw);       M[2] = 2 * (xz + yw);
    M[3] = 2 * (xy + zw);       M[4] = -x2 + y2 - z2 + w2; M[5] = 2 * (yz - xw);
    M[6] = 2 * (xz - yw);       M[7] = 2 * (yz + xw);       M[8] = -x2 - y2 + z2 + w2;
}

__device__ __forceinline__ void matrix_to_rotvec_f64(const double* M, double* rv) {
    // scipy Rotation.from_matrix (Markley's quaternion extraction) followed by as_rotvec
    double dec[4] = {M[0], M[4], M[8], M[0] + M[4] + M[8]};
    int choice = 0;
    for (int i = 1; i < 4; ++i)
        if (dec[i] > dec[choice]) choice = i;
    double q[4];
    if (choice != 3) {
        const int i = choice, j = (i + 1) % 3, k = (j + 1) % 3;
        q[i] = 1 - dec[3] + 2 * M[i * 3 + i];
        q[j] = M[j * 3 + i] + M[i * 3 + j];
        q[k] = M[k * 3 + i] + M[i * 3 + k];
        q[3] = M[k * 3 + j] - M[j * 3 + k];
    } else {
        q[0] = M[7] - M[5];
        q[1] = M[2] - M[6];
        q[2] = M[3] - M[1];
        q[3] = 1 + dec[3];
    }
    const double n = sqrt(q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3]);
    for (int i = 0; i < 4; ++i) q[i] /= n;
    if (q[3] < 0) for (int i = 0; i < 4; ++i) q[i] = -q[i];
    const double ang = 2 * atan2(sqrt(q[0] * q[0] + q[1] * q[1] + q[2] * q[2]), q[3]);
    const double a2 = ang * ang;
    const double sc = (ang <= 1e-3) ? 2 + a2 / 12 + 7 * a2 * a2 / 2880 : ang / sin(ang / 2);
    rv[0] = sc * q[0]; rv[1] = sc * q[1]; rv[2] = sc * q[2];
}

__global__ __launch_bounds__(64) void frames_to_world_kernel(const float* __restrict__ global_orient,
                                                             const float* __restrict__ body_pose,
                                                             const float* __restrict__ betas,
                                                             const float* __restrict__ transl,
                                                             const float* __restrict__ rigid,      // [4,4] row-major
                                                             const float* __restrict__ Jt, const float* __restrict__ Js,
                                                             const int* __restrict__ parents,
                                                             float* __restrict__ joints_world,     // [N,22,3]
                                                             double* __restrict__ orient_transl,   // [N,6]
                                                             int N) {
    const int n = blockIdx.x * blockDim.x + threadIdx.x;
    if (n >= N) return;
    FkCtx f;
    rodrigues(global_orient + (size_t)n * 3, f.R[0]);
    for (int j = 1; j < NJ; ++j) rodrigues(body_pose + ((size_t)n * (NJ - 1) + (j - 1)) * 3, f.R[j]);
    float beta[NBETA];
    for (int k = 0; k < NBETA; ++k) beta[k] = betas[(size_t)n * NBETA + k];
    rest_joints(Jt, Js, beta, f.Jr);
    fk_forward(f, parents);
    const float t[3] = {transl[(size_t)n * 3], transl[(size_t)n * 3 + 1], transl[(size_t)n * 3 + 2]};
    float pelvis[3];
    for (int j = 0; j < NJ; ++j) {
        float pc[3];
#pragma unroll
        for (int c = 0; c < 3; ++c) pc[c] = f.P[j][c] + t[c];                 // camera coordinates (float32, as smplx)
        if (j == 0) { pelvis[0] = pc[0]; pelvis[1] = pc[1]; pelvis[2] = pc[2]; }
#pragma unroll
        for (int c = 0; c < 3; ++c)                                           // torch.matmul(cam_R, joints^T)^T + cam_t
            joints_world[((size_t)n * NJ + j) * 3 + c] =
                rigid[c * 4] * pc[0] + rigid[c * 4 + 1] * pc[1] + rigid[c * 4 + 2] * pc[2] + rigid[c * 4 + 3];
    }
    // update_globalRT_for_smplx: delta_T and (transl + delta_T) are float32 there, the rest float64
    float dT[3], tp[3];
#pragma unroll
    for (int c = 0; c < 3; ++c) { dT[c] = pelvis[c] - t[c]; tp[c] = t[c] + dT[c]; }
    const double rv[3] = {(double)global_orient[(size_t)n * 3], (double)global_orient[(size_t)n * 3 + 1],
                          (double)global_orient[(size_t)n * 3 + 2]};
    double Rb[9], Rn[9], rvn[3];
    rotvec_to_matrix_f64(rv, Rb);
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j)
            Rn[i * 3 + j] = (double)rigid[i * 4] * Rb[j] + (double)rigid[i * 4 + 1] * Rb[3 + j] + (double)rigid[i * 4 + 2] * Rb[6 + j];
    matrix_to_rotvec_f64(Rn, rvn);
    for (int c = 0; c < 3; ++c) {
        const double tn = (double)rigid[c * 4] * tp[0] + (double)rigid[c * 4 + 1] * tp[1] + (double)rigid[c * 4 + 2] * tp[2] +
                          (double)rigid[c * 4 + 3];
        orient_transl[(size_t)n * 6 + c] = rvn[c];
        orient_transl[(size_t)n * 6 + 3 + c] = tn - (double)dT[c];
    }
}

// fold of the joint regressor: one block per (joint, coord[, beta]) row, fp64 accumulation
__global__ __launch_bounds__(256) void fold_regressor_kernel(const float* __restrict__ Jreg, const float* __restrict__ src,
                                                             int src_stride, int src_off, int V, float* __restrict__ out,
                                                             int rows_per_joint) {
    // out[j * rows_per_joint + r] = sum_v Jreg[j, v] * src[(v*3 + c) * src_stride + src_off + k], r = c*K + k
    const int row = blockIdx.x;
    const int j = row / rows_per_joint, r = row % rows_per_joint;
    const int K = rows_per_joint / 3, c = r / K, k = r % K;
    double s = 0.0;
    for (int v = threadIdx.x; v < V; v += blockDim.x)
        s += (double)Jreg[(size_t)j * V + v] * (double)src[((size_t)v * 3 + c) * src_stride + src_off + k];
    __shared__ double sh[256];
    sh[threadIdx.x] = s;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if (threadIdx.x < o) sh[threadIdx.x] += sh[threadIdx.x + o];
        __syncthreads();
    }
    if (threadIdx.x == 0) out[row] = (float)sh[0];
}

}  // namespace rohm

using namespace rohm;

extern "C" {

int rohm_smplx_create(rohm_smplx_t** out, const float* v_template, const float* shapedirs, int n_shape_total,
                      const float* J_regressor, const int32_t* parents, int V, int J, int device) {
    ROHM_ARG_CHECK(out && v_template && shapedirs && J_regressor && parents, "smplx_create: null argument");
    ROHM_ARG_CHECK(J >= NJ && J <= 64 && V > 0 && n_shape_total >= NBETA, "smplx_create: bad sizes (J=%d V=%d)", J, V);
    ROHM_HIP_CHECK(hipSetDevice(device));
    rohm_smplx* h = new rohm_smplx();
    h->V = V; h->J = J; h->device = device;
    float *d_vt = nullptr, *d_sd = nullptr, *d_jr = nullptr;
    auto fail = [&](const char* what, hipError_t e) {
        set_error("smplx_create: %s: %s", what, hipGetErrorString(e));
        if (d_vt) (void)hipFree(d_vt);
        if (d_sd) (void)hipFree(d_sd);
        if (d_jr) (void)hipFree(d_jr);
        if (h->d_Jt) (void)hipFree(h->d_Jt);
        if (h->d_Js) (void)hipFree(h->d_Js);
        if (h->d_parents) (void)hipFree(h->d_parents);
        delete h;
        return ROHM_ERR_HIP;
    };
    hipError_t e;
#define TRY(x) if ((e = (x)) != hipSuccess) return fail(#x, e)
    TRY(hipMalloc(&d_vt, (size_t)V * 3 * sizeof(float)));
    TRY(hipMalloc(&d_sd, (size_t)V * 3 * n_shape_total * sizeof(float)));
    TRY(hipMalloc(&d_jr, (size_t)J * V * sizeof(float)));
    TRY(hipMalloc(&h->d_Jt, (size_t)J * 3 * sizeof(float)));
    TRY(hipMalloc(&h->d_Js, (size_t)J * 3 * NBETA * sizeof(float)));
    TRY(hipMalloc(&h->d_parents, (size_t)J * sizeof(int)));
    TRY(hipMemcpy(d_vt, v_template, (size_t)V * 3 * sizeof(float), hipMemcpyDefault));
    TRY(hipMemcpy(d_sd, shapedirs, (size_t)V * 3 * n_shape_total * sizeof(float), hipMemcpyDefault));
    TRY(hipMemcpy(d_jr, J_regressor, (size_t)J * V * sizeof(float), hipMemcpyDefault));
    TRY(hipMemcpy(h->d_parents, parents, (size_t)J * sizeof(int), hipMemcpyDefault));
    TRY(hipMemcpy(h->parents, parents, (size_t)J * sizeof(int), hipMemcpyDefault));
    for (int j = 1; j < NJ; ++j)
        if (h->parents[j] < 0 || h->parents[j] >= j) {
            set_error("smplx_create: parents[%d] = %d is not an earlier joint", j, h->parents[j]);
            return fail("kinematic tree", hipSuccess);
        }
    hipLaunchKernelGGL(fold_regressor_kernel, dim3(J * 3), dim3(256), 0, 0, d_jr, d_vt, 1, 0, V, h->d_Jt, 3);
    hipLaunchKernelGGL(fold_regressor_kernel, dim3(J * 3 * NBETA), dim3(256), 0, 0, d_jr, d_sd, n_shape_total, 0, V,
                       h->d_Js, 3 * NBETA);
    TRY(hipDeviceSynchronize());
#undef TRY
    (void)hipFree(d_vt); (void)hipFree(d_sd); (void)hipFree(d_jr);
    *out = h;
    return ROHM_OK;
}

void rohm_smplx_destroy(rohm_smplx_t* h) {
    if (!h) return;
    (void)hipFree(h->d_Jt); (void)hipFree(h->d_Js); (void)hipFree(h->d_parents);
    if (h->d_vt) (void)hipFree(h->d_vt);
    if (h->d_sd) (void)hipFree(h->d_sd);
    if (h->d_pdT) (void)hipFree(h->d_pdT);
    if (h->d_wT) (void)hipFree(h->d_wT);
    delete h;
}

int rohm_smplx_joints(const rohm_smplx_t* h, const float* pose, int n_pose, const float* betas, const float* transl,
                      int N, float* joints, int n_out, rohm_stream_t stream) {
    ROHM_ARG_CHECK(h && pose && betas && transl && joints, "smplx_joints: null argument");
    ROHM_ARG_CHECK(n_pose >= 1 && n_out >= 1 && n_out <= NJ, "smplx_joints: n_out must be in [1, %d]", NJ);
    if (N <= 0) return ROHM_OK;
    prof::Scope ps("smplx_joints", 0.0, 4.0 * N * (n_pose * 3 + 13 + n_out * 3), (hipStream_t)stream);
    hipLaunchKernelGGL(smplx_joints_kernel, dim3((N + 63) / 64), dim3(64), 0, (hipStream_t)stream, pose, n_pose, betas,
                       transl, h->d_Jt, h->d_Js, h->d_parents, joints, n_out, N);
    ROHM_LAUNCH_CHECK();
    return ROHM_OK;
}

int rohm_smplx_frames_to_world(const rohm_smplx_t* h, const float* global_orient, const float* body_pose,
                               const float* betas, const float* transl, const float* rigid, int N, float* joints_world,
                               double* orient_transl_world, rohm_stream_t stream) {
    ROHM_ARG_CHECK(h && global_orient && body_pose && betas && transl && rigid && joints_world && orient_transl_world,
                   "smplx_frames_to_world: null argument");
    if (N <= 0) return ROHM_OK;
    prof::Scope ps("frames_to_world", 0.0, 4.0 * N * (3 + 63 + 10 + 3 + 66) + 8.0 * N * 6, (hipStream_t)stream);
    hipLaunchKernelGGL(frames_to_world_kernel, dim3((N + 63) / 64), dim3(64), 0, (hipStream_t)stream, global_orient,
                       body_pose, betas, transl, rigid, h->d_Jt, h->d_Js, h->d_parents, joints_world, orient_transl_world, N);
    ROHM_LAUNCH_CHECK();
    return ROHM_OK;
}

size_t rohm_guidance_workspace_bytes(int B, int T) {
    if (B <= 0 || T <= 0) return 0;
    return ((size_t)B * T * 24 + (size_t)B * 21 + 64) * sizeof(float);
}

// The skating loss normalises by mask counts taken over the whole batch (model/posenet.py:231,243), so under clip
// sharding the reference-at-full-batch result needs the two counts summed over the ranks between the two halves:
//   prepare: foot joints of both recoveries + the local counts;   apply: gradient with the counts it is GIVEN.
int rohm_guidance_skating_prepare(const rohm_smplx_t* h, const float* x0, const float* mean294, const float* std294,
                                  int B, int T, float* counts2, void* ws, size_t ws_bytes, rohm_stream_t stream) {
    ROHM_ARG_CHECK(h && x0 && mean294 && std294 && counts2 && ws, "guidance_skating: null argument");
    ROHM_ARG_CHECK(B > 0 && T > 1, "guidance_skating: need B > 0 and T > 1");
    ROHM_ARG_CHECK(ws_bytes >= rohm_guidance_workspace_bytes(B, T), "guidance_skating: workspace too small");
    hipStream_t s = (hipStream_t)stream;
    float* feet = (float*)ws;
    prof::Scope ps("guidance_skating_fwd", 0.0, 4.0 * B * T * (C_TOTAL + 24), s);
    ROHM_HIP_CHECK(hipMemsetAsync(counts2, 0, 2 * sizeof(float), s));
    const int nf = B * T;
    hipLaunchKernelGGL(skating_fwd_kernel, dim3((nf + 63) / 64), dim3(64), 0, s, x0, mean294, std294, h->d_Jt, h->d_Js,
                       h->d_parents, feet, B, T);
    const int np = B * (T - 1) * 4;
    hipLaunchKernelGGL(skating_count_kernel, dim3((np + 255) / 256), dim3(256), 0, s, x0, mean294, std294, feet,
                       counts2, B, T);
    ROHM_LAUNCH_CHECK();
    return ROHM_OK;
}

int rohm_guidance_skating_apply(const rohm_smplx_t* h, const float* x0, const float* mean294, const float* std294,
                                int B, int T, const float* counts2, float* grad_out, void* ws, size_t ws_bytes,
                                rohm_stream_t stream) {
    ROHM_ARG_CHECK(h && x0 && mean294 && std294 && grad_out && counts2 && ws, "guidance_skating: null argument");
    ROHM_ARG_CHECK(B > 0 && T > 1, "guidance_skating: need B > 0 and T > 1");
    ROHM_ARG_CHECK(ws_bytes >= rohm_guidance_workspace_bytes(B, T), "guidance_skating: workspace too small");
    hipStream_t s = (hipStream_t)stream;
    prof::Scope ps("guidance_skating_bwd", 0.0, 4.0 * B * T * (2 * C_TOTAL + 24), s);
    const int nf = B * T;
    hipLaunchKernelGGL(skating_bwd_kernel, dim3((nf + 63) / 64), dim3(64), 0, s, x0, mean294, std294, h->d_Jt, h->d_Js,
                       h->d_parents, (const float*)ws, counts2, grad_out, B, T);
    ROHM_LAUNCH_CHECK();
    return ROHM_OK;
}

int rohm_guidance_skating_grad(const rohm_smplx_t* h, const float* x0, const float* mean294, const float* std294,
                               int B, int T, float* grad_out, float* counts2, void* ws, size_t ws_bytes,
                               rohm_stream_t stream) {
    ROHM_ARG_CHECK(grad_out, "guidance_skating: null argument");
    int rc = rohm_guidance_skating_prepare(h, x0, mean294, std294, B, T, counts2, ws, ws_bytes, stream);
    if (rc) return rc;
    return rohm_guidance_skating_apply(h, x0, mean294, std294, B, T, counts2, grad_out, ws, ws_bytes, stream);
}

int rohm_guidance_proj2d_grad(const rohm_smplx_t* h, const float* x0, const float* mean294, const float* std294,
                              const float* transf_matrix, const float* cam_R, const float* cam_t, const float* focal,
                              const float* center, const float* kp2d, int kp_frames, int B, int T, float* grad_out,
                              void* ws, size_t ws_bytes, rohm_stream_t stream) {
    ROHM_ARG_CHECK(h && x0 && mean294 && std294 && transf_matrix && cam_R && cam_t && focal && center && kp2d &&
                       grad_out && ws, "guidance_proj2d: null argument");
    ROHM_ARG_CHECK(B > 0 && T > 0 && kp_frames >= T, "guidance_proj2d: keypoints must cover T frames");
    ROHM_ARG_CHECK(ws_bytes >= rohm_guidance_workspace_bytes(B, T), "guidance_proj2d: workspace too small");
    hipStream_t s = (hipStream_t)stream;
    float* cam2 = (float*)ws + (size_t)B * T * 24;
    prof::Scope ps("guidance_proj2d", 0.0, 8.0 * B * T * C_TOTAL, s);
    hipLaunchKernelGGL(proj_prep_kernel, dim3((B + 63) / 64), dim3(64), 0, s, transf_matrix, cam_R, cam2, B);
    const int nf = B * T;
    hipLaunchKernelGGL(proj2d_kernel, dim3((nf + 63) / 64), dim3(64), 0, s, x0, mean294, std294, h->d_Jt, h->d_Js,
                       h->d_parents, cam2, cam_t, focal, center, kp2d, kp_frames, grad_out, B, T);
    ROHM_LAUNCH_CHECK();
    return ROHM_OK;
}

}  // extern "C"
