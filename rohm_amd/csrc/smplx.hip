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

// ---------------------------------------------------------------------------------------- 2-D re-projection
__device__ __forceinline__ void inv3(const float* a, float* o) {
    const float c00 = a[4] * a[8] - a[5] * a[7], c01 = a[5] * a[6] - a[3] * a[8], c02 = a[3] * a[7] - a[4] * a[6];
    const float det = a[0] * c00 + a[1] * c01 + a[2] * c02;
    const float id = 1.f / det;
    o[0] = c00 * id; o[1] = (a[2] * a[7] - a[1] * a[8]) * id; o[2] = (a[1] * a[5] - a[2] * a[4]) * id;
    o[3] = c01 * id; o[4] = (a[0] * a[8] - a[2] * a[6]) * id; o[5] = (a[2] * a[3] - a[0] * a[5]) * id;
    o[6] = c02 * id; o[7] = (a[1] * a[6] - a[0] * a[7]) * id; o[8] = (a[0] * a[4] - a[1] * a[3]) * id;
}

// ---------------------------------------------------------------------------------------- lanes over joints
// Round 1 gave every frame ONE thread: a 22-joint FK context (R, G, rest joints, posed joints = 528 floats) plus the
// reverse-pass arrays per thread live in scratch memory, and B*T = 4576 threads are 72 waves on a 256-CU chip: 150-180
// us per kernel at B = 32, two orders of magnitude off the 5 MB they move.  Here a frame is 32 lanes (lane = joint, 22
// used), a workgroup is a tile of 8 consecutive frames of one clip:
//   * the 294 channels of the tile are staged through LDS with coalesced loads (de-normalised on the way in) and the
//     gradient tile goes back the same way (every channel written, zeros where the reference zeroes);
//   * FK runs level by level down the kinematic tree, a lane fetching its parent's world rotation / position with
//     wave shuffles (a frame never leaves its half-wave: no barriers);
//   * the reverse pass runs the levels upwards, a parent gathering the 15 numbers (dP, -dJrest, dG) each of its <= 3
//     children contributes; 6-D Gram-Schmidt forward / backward are lane-local; d(beta) is a 5-step butterfly.
// Everything stays in registers; the grid is B * ceil(T / 8) workgroups of 256 threads.  The tree (parent, depth,
// children of each joint) travels in the kernel arguments: no pointer chasing through global memory.
constexpr int GF = 8;                  // frames per workgroup (B = 32: 576 workgroups of 256 threads, 2-3 per CU)
constexpr int GMAXKID = 3;             // children per joint (SMPL-X body: pelvis and spine3 have three)

struct GuideArgs {
    const float *x0, *mean, *stdv, *Jt, *Js;
    signed char parent[NJ], depth[NJ], kid[NJ][GMAXKID];     // kinematic tree of the 22 body joints (-1 = none)
    int max_depth;
    int B, T;
    float* feet;                 // [B, T, 2, 4, 3]      MODE 0 out, MODE 1 in
    const float* counts;         // [2]                  MODE 1
    float* grad;                 // [B, 294, 1, T]       MODE 1 / 2 out
    // MODE 2 (guide_2d_projection_with_smpl, model/posenet.py:260-317)
    const float *transf, *camR, *camT, *focal, *center, *kp2d;
    int kp_frames;
};

__device__ __forceinline__ float shfl_in(float v, int src_lane) { return __shfl(v, src_lane, 64); }

// MODE 0: foot joints of both recoveries (skating, pass 1);  1: skating gradient;  2: 2-D re-projection gradient
template <int MODE>
__global__ __launch_bounds__(GF * 32) void guide_lanes_kernel(GuideArgs a) {
    __shared__ float xs[C_TOTAL * GF];          // input tile, then gradient tile
    __shared__ float cam[21];                   // MODE 2: Rc (9), Tc (3), Ri (9)
    const int tid = threadIdx.x;
    const int j = tid & 31, fl = tid >> 5;      // joint lane, frame within the tile
    const int half = (tid & 63) & 32;           // first lane of this frame's half-wave
    const int ntile = (a.T + GF - 1) / GF;
    const int b = blockIdx.x / ntile, t0 = (blockIdx.x % ntile) * GF;
    const int t = t0 + fl, T = a.T;
    const bool active = (t < T) && (j < NJ);
    const int jj = (j < NJ) ? j : 0;            // idle lanes shadow joint 0 (never written anywhere)

    for (int idx = tid; idx < C_TOTAL * GF; idx += GF * 32) {
        const int c = idx / GF, f = idx % GF;
        xs[idx] = (t0 + f < T) ? a.x0[((size_t)b * C_TOTAL + c) * T + t0 + f] * a.stdv[c] + a.mean[c] : 0.f;
    }
    if (MODE == 2 && tid == 0) {
        // cano -> scene: inv of the affine transf_matrix = [R^-1, -R^-1 t] (torch.linalg.inv, posenet.py:286); inv(cam_R)
        const float* m = a.transf + (size_t)b * 16;
        const float R[9] = {m[0], m[1], m[2], m[4], m[5], m[6], m[8], m[9], m[10]};
        const float tr[3] = {m[3], m[7], m[11]};
        float o[9], tt[3], ri[9];
        inv3(R, o);
        mat_vec(o, tr, tt);
        inv3(a.camR, ri);
#pragma unroll
        for (int i = 0; i < 9; ++i) { cam[i] = o[i]; cam[12 + i] = ri[i]; }
        cam[9] = -tt[0]; cam[10] = -tt[1]; cam[11] = -tt[2];
    }
    __syncthreads();
    auto X = [&](int c) { return xs[c * GF + fl]; };

    // ---- lane-local: rotation from the 6-D vector, rest joint, place in the tree --------------------------------
    float x6[6], R[9], Jr[3], trans[3];
#pragma unroll
    for (int k = 0; k < 6; ++k) x6[k] = X(jj == 0 ? CH_ROT6D + k : CH_POSE6D + (jj - 1) * 6 + k);
    rot6d_fwd(x6, R);
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        float v = a.Jt[jj * 3 + c];
#pragma unroll
        for (int k = 0; k < NBETA; ++k) v = fmaf(a.Js[(jj * 3 + c) * NBETA + k], X(CH_BETAS + k), v);
        Jr[c] = v;
        trans[c] = X(CH_TRANS + c);
    }
    const int par = (jj > 0) ? a.parent[jj] : 0;
    const int depth = a.depth[jj];
    int kid[GMAXKID];
#pragma unroll
    for (int s2 = 0; s2 < GMAXKID; ++s2) kid[s2] = (j < NJ) ? a.kid[jj][s2] : -1;

    // ---- forward kinematics, level by level ------------------------------------------------------------------------
    float off[3], G[9], P[3], Gp[9];
#pragma unroll
    for (int c = 0; c < 3; ++c) off[c] = Jr[c] - shfl_in(Jr[c], half + par);
#pragma unroll
    for (int i = 0; i < 9; ++i) { G[i] = R[i]; Gp[i] = 0.f; }
#pragma unroll
    for (int c = 0; c < 3; ++c) P[c] = Jr[c];
    for (int d = 1; d <= a.max_depth; ++d) {
        float gi[9], pi[3];
#pragma unroll
        for (int i = 0; i < 9; ++i) gi[i] = shfl_in(G[i], half + par);
#pragma unroll
        for (int c = 0; c < 3; ++c) pi[c] = shfl_in(P[c], half + par);
        if (depth == d) {
            float w[3];
#pragma unroll
            for (int i = 0; i < 9; ++i) Gp[i] = gi[i];
            mat_mul(Gp, R, G);
            mat_vec(Gp, off, w);
#pragma unroll
            for (int c = 0; c < 3; ++c) P[c] = pi[c] + w[c];
        }
    }

    if constexpr (MODE == 0) {
        // feet[b, t, path, k, :]: path 0 = abs-trajectory recovery, 1 = SMPL-X recovery (posenet.py:213-216)
        if (t < T) {
            float* o = a.feet + ((size_t)b * T + t) * 24;
            if (j < 4) {
                const int fj = kFoot[j];
                const float pos[3] = {X(CH_ROOT_POS), X(CH_ROOT_POS + 1), X(CH_ROOT_H)};
                const float v[3] = {X(CH_LOCAL + 3 * fj), X(CH_LOCAL + 3 * fj + 1), X(CH_LOCAL + 3 * fj + 2)};
                float r[3];
                abs_joint(X(CH_ROOT_ANG), pos, v, r);
#pragma unroll
                for (int c = 0; c < 3; ++c) o[j * 3 + c] = r[c];
            }
#pragma unroll
            for (int k = 0; k < 4; ++k)
                if (j == kFoot[k]) {
#pragma unroll
                    for (int c = 0; c < 3; ++c) o[12 + k * 3 + c] = P[c] + trans[c];
                }
        }
        return;
    } else {
        // ---- dL/dP of this lane's joint ----------------------------------------------------------------------------
        float gP[3] = {0.f, 0.f, 0.f}, dl[3] = {0.f, 0.f, 0.f};
        int foot_k = -1;
        if constexpr (MODE == 1) {
            const float inv0 = a.counts[0] > 0.f ? 1.f / a.counts[0] : 0.f;
            const float inv1 = a.counts[1] > 0.f ? 1.f / a.counts[1] : 0.f;
#pragma unroll
            for (int k = 0; k < 4; ++k)
                if (j == kFoot[k]) foot_k = k;
            if (active && foot_k >= 0) {
                float g[3];
                skating_djoint(a.x0, a.mean, a.stdv, a.feet, b, t, T, 0, foot_k, inv0, g);
                abs_joint_T(X(CH_ROOT_ANG), g, dl);       // abs-trajectory recovery: only local_positions get gradient
                skating_djoint(a.x0, a.mean, a.stdv, a.feet, b, t, T, 1, foot_k, inv1, gP);
            }
        } else {
            bool proj = false;
#pragma unroll
            for (int k = 0; k < 10; ++k) proj = proj || (j == kProj[k]);
            if (active && proj) {
                const float* Rc = cam;
                const float* Tc = cam + 9;
                const float* Ri = cam + 12;
                const float fx = a.focal[b * 2], fy = a.focal[b * 2 + 1], cx = a.center[b * 2], cy = a.center[b * 2 + 1];
                const float inv_n = 1.f / ((float)a.B * (float)T * 20.f);      // .mean() over B*T*10*2 (posenet.py:308-309)
                const float p[3] = {P[0] + trans[0], P[1] + trans[1], P[2] + trans[2]};
                float sc[3], cm[3];
                mat_vec(Rc, p, sc);
                const float dd[3] = {sc[0] + Tc[0] - a.camT[0], sc[1] + Tc[1] - a.camT[1], sc[2] + Tc[2] - a.camT[2]};
                mat_vec(Ri, dd, cm);
                const float iz = 1.f / cm[2];
                const float u = fx * (cm[0] * iz) + cx, v = fy * (cm[1] * iz) + cy;
                const float* kp = a.kp2d + (((size_t)b * a.kp_frames + t) * NJ + j) * 3;
                const float conf = kp[2];
                const float du = u - kp[0], dv = v - kp[1];
                // d(-mean |.| conf): -sign(diff) conf / N
                const float gu = -((du > 0.f) - (du < 0.f)) * conf * inv_n;
                const float gv = -((dv > 0.f) - (dv < 0.f)) * conf * inv_n;
                const float gc[3] = {gu * fx * iz, gv * fy * iz, -(gu * fx * cm[0] + gv * fy * cm[1]) * iz * iz};
                float gs[3];
                matT_vec(Ri, gc, gs);
                matT_vec(Rc, gs, gP);
            }
        }

        // ---- reverse pass of the kinematic chain, levels upwards (fk_backward of smplx_fk.h, one joint per lane) -----
        float dG[9], dR[9], dJr[3] = {0.f, 0.f, 0.f};
#pragma unroll
        for (int i = 0; i < 9; ++i) { dG[i] = 0.f; dR[i] = 0.f; }
        for (int d = a.max_depth; d >= 1; --d) {
            float cb[15];
#pragma unroll
            for (int i = 0; i < 15; ++i) cb[i] = 0.f;
            if (active && depth == d) {
                // P[j] = P[p] + G[p] off,  G[j] = G[p] R[j]
                float doff[3];
                matT_vec(Gp, gP, doff);
#pragma unroll
                for (int c = 0; c < 3; ++c) { dJr[c] += doff[c]; cb[c] = gP[c]; cb[3 + c] = -doff[c]; }
#pragma unroll
                for (int r = 0; r < 3; ++r)
#pragma unroll
                    for (int c = 0; c < 3; ++c) {
                        float sg = 0.f, sr = 0.f;
#pragma unroll
                        for (int k = 0; k < 3; ++k) {
                            sg += dG[r * 3 + k] * R[c * 3 + k];        // dG[j] R[j]^T
                            sr += Gp[k * 3 + r] * dG[k * 3 + c];       // G[p]^T dG[j]
                        }
                        cb[6 + r * 3 + c] = gP[r] * off[c] + sg;
                        dR[r * 3 + c] = sr;
                    }
            }
#pragma unroll
            for (int s2 = 0; s2 < GMAXKID; ++s2) {
                const bool take = active && kid[s2] >= 0 && depth + 1 == d;
                const int src = half + (kid[s2] >= 0 ? kid[s2] : jj);
#pragma unroll
                for (int i = 0; i < 15; ++i) {
                    const float v = shfl_in(cb[i], src);
                    if (take) {
                        if (i < 3) gP[i] += v;
                        else if (i < 6) dJr[i - 3] += v;
                        else dG[i - 6] += v;
                    }
                }
            }
        }
        if (j == 0) {
#pragma unroll
            for (int c = 0; c < 3; ++c) dJr[c] += gP[c];
        }
        float d6[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        if (active && j >= 1) rot6d_bwd(x6, dR, d6);
        float dbeta[NBETA];
#pragma unroll
        for (int k = 0; k < NBETA; ++k) {
            float v = 0.f;
            if (active) {
#pragma unroll
                for (int c = 0; c < 3; ++c) v = fmaf(dJr[c], a.Js[(jj * 3 + c) * NBETA + k], v);
            }
#pragma unroll
            for (int o = 16; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);      // stays inside the frame's 32 lanes
            dbeta[k] = v;
        }

        // ---- gradient tile: every channel written (zeros where the reference zeroes: [0, 22), [290, 294); local_vel
        // never enters either recovery), scaled by d(denormalised)/d(normalised) = Std --------------------------------
        __syncthreads();
        for (int idx = tid; idx < C_TOTAL * GF; idx += GF * 32) xs[idx] = 0.f;
        __syncthreads();
        if (active) {
            if (j >= 1) {
#pragma unroll
                for (int k = 0; k < 6; ++k) xs[(CH_POSE6D + (j - 1) * 6 + k) * GF + fl] = d6[k];
            } else {
#pragma unroll
                for (int k = 0; k < NBETA; ++k) xs[(CH_BETAS + k) * GF + fl] = dbeta[k];
            }
            if (MODE == 1 && foot_k >= 0) {
#pragma unroll
                for (int c = 0; c < 3; ++c) xs[(CH_LOCAL + 3 * j + c) * GF + fl] = dl[c];
            }
        }
        __syncthreads();
        for (int idx = tid; idx < C_TOTAL * GF; idx += GF * 32) {
            const int c = idx / GF, f = idx % GF;
            if (t0 + f < T) a.grad[((size_t)b * C_TOTAL + c) * T + t0 + f] = xs[idx] * a.stdv[c];
        }
    }
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

static void fill_tree(const rohm_smplx* h, GuideArgs& ga) {
    int n[NJ] = {0};
    for (int j = 0; j < NJ; ++j)
        for (int s2 = 0; s2 < GMAXKID; ++s2) ga.kid[j][s2] = -1;
    ga.parent[0] = -1; ga.depth[0] = 0;
    for (int j = 1; j < NJ; ++j) {
        ga.parent[j] = (signed char)h->parents[j];
        int d = 0;
        for (int q = j; q > 0; q = h->parents[q]) ++d;
        ga.depth[j] = (signed char)d;
    }
    for (int j = NJ - 1; j >= 1; --j) {          // descending: the order the serial reverse pass visited the children
        const int p = h->parents[j];
        ga.kid[p][n[p]++] = (signed char)j;
    }
    ga.max_depth = h->max_depth22;
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
    {   // the guidance kernels walk the 22-joint tree level by level, at most GMAXKID children per joint
        int nkid[NJ] = {0};
        h->max_depth22 = 0;
        for (int j = 1; j < NJ; ++j) {
            int d = 0;
            for (int q = j; q > 0; q = h->parents[q]) ++d;
            if (d > h->max_depth22) h->max_depth22 = d;
            if (++nkid[h->parents[j]] > GMAXKID) {
                set_error("smplx_create: joint %d has more than %d children among the first %d joints", h->parents[j], GMAXKID, NJ);
                return fail("kinematic tree", hipSuccess);
            }
        }
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
    if (h->d_ell_j) (void)hipFree(h->d_ell_j);
    if (h->d_ell_w) (void)hipFree(h->d_ell_w);
    if (h->d_zero_bias) (void)hipFree(h->d_zero_bias);
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
    GuideArgs ga{};
    ga.x0 = x0; ga.mean = mean294; ga.stdv = std294; ga.Jt = h->d_Jt; ga.Js = h->d_Js;
    fill_tree(h, ga); ga.B = B; ga.T = T; ga.feet = feet;
    hipLaunchKernelGGL(guide_lanes_kernel<0>, dim3(B * ((T + GF - 1) / GF)), dim3(GF * 32), 0, s, ga);
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
    GuideArgs ga{};
    ga.x0 = x0; ga.mean = mean294; ga.stdv = std294; ga.Jt = h->d_Jt; ga.Js = h->d_Js;
    fill_tree(h, ga); ga.B = B; ga.T = T; ga.feet = (float*)ws; ga.counts = counts2; ga.grad = grad_out;
    hipLaunchKernelGGL(guide_lanes_kernel<1>, dim3(B * ((T + GF - 1) / GF)), dim3(GF * 32), 0, s, ga);
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
    prof::Scope ps("guidance_proj2d", 0.0, 8.0 * B * T * C_TOTAL, s);
    GuideArgs ga{};
    ga.x0 = x0; ga.mean = mean294; ga.stdv = std294; ga.Jt = h->d_Jt; ga.Js = h->d_Js;
    fill_tree(h, ga); ga.B = B; ga.T = T; ga.grad = grad_out;
    ga.transf = transf_matrix; ga.camR = cam_R; ga.camT = cam_t; ga.focal = focal; ga.center = center; ga.kp2d = kp2d;
    ga.kp_frames = kp_frames;
    hipLaunchKernelGGL(guide_lanes_kernel<2>, dim3(B * ((T + GF - 1) / GF)), dim3(GF * 32), 0, s, ga);
    ROHM_LAUNCH_CHECK();
    return ROHM_OK;
}

}  // extern "C"
