// Between-stage trajectory re-derivation on the device.
//
// Reference: the drivers' inference-iteration loop, test_amass_full.py:262-311 / test_prox_egobody.py:238-287:
//   de-normalise TrajNet's output representation -> SMPL-X joints of every frame
//   (recover_from_repr_smpl 'smplx_params', data_loaders/motion_representation.py:373-398; the 10 475-vertex
//   mesh it also builds is never read) -> D2H -> per sequence get_repr_smplx (motion_representation.py:187-282)
//   -> re-normalise with the pose dataset's statistics -> keep channels 0..21 -> H2D into PoseNet's `cond`.
// The 22 trajectory channels depend only on five joints (pelvis, both hips, both shoulders), the global
// orientation and the translation, so the whole detour is one small kernel: one workgroup per clip, one thread
// per frame, no host round trip.
//
// Precision follows the reference's dtype flow (oracle/rederive.py): positions and the quaternion algebra in
// float32 (qbetween_np / qmul_np / qrot_np cast to float32, quaternion.py:21-23,126-135,397-406) WITHOUT fma
// contraction, the facing direction, the angular velocity and the normalisation in float64, one rounding to
// float32 at the store (the reference rounds when it assigns into the float32 `cond`, test_amass_full.py:336).
// The reference's R -> axis-angle -> scipy Rodrigues (float64) round trip for the global orientation is the
// identity on SO(3); the Gram-Schmidt matrix is used directly.
#include "common.h"
#include "smplx_fk.h"

namespace rohm {

constexpr int kTrajCh = 22;
// get_repr_smplx unpacks face_joint_indx = [2, 1, 17, 16] as `l_hip, r_hip, sdr_r, sdr_l`
// (motion_representation.py:14,201): inside the function r_hip = 1, l_hip = 2.
constexpr int kRHip = 1, kLHip = 2, kSdrR = 17, kSdrL = 16;

struct FrameState {      // what frame t+1 contributes to frame t, parked in LDS
    float q[4];          // root quaternion (w, x, y, z)
    float p0[3];         // pelvis position
    float R[9];          // global orientation, row-major
    float tr[3];         // translation
};

__device__ __forceinline__ float mul(float a, float b) { return __fmul_rn(a, b); }
__device__ __forceinline__ float add(float a, float b) { return __fadd_rn(a, b); }
__device__ __forceinline__ float sub(float a, float b) { return __fsub_rn(a, b); }

__device__ __forceinline__ void cross_rn(const float* a, const float* b, float* o) {
    o[0] = sub(mul(a[1], b[2]), mul(a[2], b[1]));
    o[1] = sub(mul(a[2], b[0]), mul(a[0], b[2]));
    o[2] = sub(mul(a[0], b[1]), mul(a[1], b[0]));
}

// qrot (quaternion.py:52-71): v + 2 (w (u x v) + u x (u x v)), u = q.xyz
__device__ __forceinline__ void qrot_rn(const float* q, const float* v, float* o) {
    float uv[3], uuv[3];
    cross_rn(q + 1, v, uv);
    cross_rn(q + 1, uv, uuv);
#pragma unroll
    for (int c = 0; c < 3; ++c) o[c] = add(v[c], mul(2.f, add(mul(q[0], uv[c]), uuv[c])));
}

__global__ __launch_bounds__(256) void traj_rederive_kernel(
    const float* __restrict__ repr, long long isb, long long ist, long long isc, const float* __restrict__ mean_in,
    const float* __restrict__ std_in, const float* __restrict__ mean_out, const float* __restrict__ std_out,
    const float* __restrict__ Jt, const float* __restrict__ Js, const int* __restrict__ parents, float* __restrict__ out,
    long long osb, long long ost, long long osc, int T) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    FrameState* st = reinterpret_cast<FrameState*>(smem_raw);
    int* first_nan = reinterpret_cast<int*>(st + T);
    const int b = blockIdx.x;
    if (threadIdx.x == 0) *first_nan = T;
    __syncthreads();

    // ---- phase 1: joints of every frame, root quaternion from the facing direction ---------------------------
    for (int t = threadIdx.x; t < T; t += blockDim.x) {
        const float* x = repr + (size_t)b * isb + (size_t)t * ist;
        auto ldc = [&](int c) { return add(mul(x[(size_t)c * isc], std_in[c]), mean_in[c]); };   // x * Std + Mean
        FrameIn in;
#pragma unroll
        for (int k = 0; k < 6; ++k) in.x6[0][k] = ldc(CH_ROT6D + k);
        for (int j = 1; j < NJ; ++j)
#pragma unroll
            for (int k = 0; k < 6; ++k) in.x6[j][k] = ldc(CH_POSE6D + (j - 1) * 6 + k);
#pragma unroll
        for (int k = 0; k < NBETA; ++k) in.beta[k] = ldc(CH_BETAS + k);
#pragma unroll
        for (int k = 0; k < 3; ++k) in.trans[k] = ldc(CH_TRANS + k);
        FkCtx f;
        smplx_fk(in, Jt, Js, parents, f);
        float p[4][3];   // r_hip, l_hip, sdr_r, sdr_l with the translation added (smplx adds transl to joints)
        const int idx[4] = {kRHip, kLHip, kSdrR, kSdrL};
#pragma unroll
        for (int k = 0; k < 4; ++k)
#pragma unroll
            for (int c = 0; c < 3; ++c) p[k][c] = add(f.P[idx[k]][c], in.trans[c]);
        FrameState s;
#pragma unroll
        for (int c = 0; c < 3; ++c) { s.p0[c] = add(f.P[0][c], in.trans[c]); s.tr[c] = in.trans[c]; }
#pragma unroll
        for (int i = 0; i < 9; ++i) s.R[i] = f.R[0][i];
        // across = (r_hip - l_hip) + (sdr_r - sdr_l), normalised in float32 (motion_representation.py:202-205)
        float ac[3];
#pragma unroll
        for (int c = 0; c < 3; ++c) ac[c] = add(sub(p[0][c], p[1][c]), sub(p[2][c], p[3][c]));
        const float an = sqrtf(add(add(mul(ac[0], ac[0]), mul(ac[1], ac[1])), mul(ac[2], ac[2])));
#pragma unroll
        for (int c = 0; c < 3; ++c) ac[c] = __fdiv_rn(ac[c], an);
        // forward = z x across, normalised in float64 (np.cross with an int64 array promotes, :206-207)
        double fw[3] = {-(double)ac[1], (double)ac[0], 0.0};
        const double fn = sqrt(__dadd_rn(__dadd_rn(__dmul_rn(fw[0], fw[0]), __dmul_rn(fw[1], fw[1])), 0.0));
        const float v0[3] = {(float)(fw[0] / fn), (float)(fw[1] / fn), (float)(fw[2] / fn)};
        // qbetween(forward, +y) in float32 (quaternion.py:385-394)
        const float v1[3] = {0.f, 1.f, 0.f};
        float v[3];
        cross_rn(v0, v1, v);
        const float n0 = add(add(mul(v0[0], v0[0]), mul(v0[1], v0[1])), mul(v0[2], v0[2]));
        const float dt = add(add(mul(v0[0], v1[0]), mul(v0[1], v1[1])), mul(v0[2], v1[2]));
        const float w = add(sqrtf(mul(n0, 1.f)), dt);
        const float qn = sqrtf(add(add(add(mul(w, w), mul(v[0], v[0])), mul(v[1], v[1])), mul(v[2], v[2])));
        s.q[0] = __fdiv_rn(w, qn); s.q[1] = __fdiv_rn(v[0], qn); s.q[2] = __fdiv_rn(v[1], qn); s.q[3] = __fdiv_rn(v[2], qn);
        if (isnan(s.q[0]) || isnan(s.q[1]) || isnan(s.q[2]) || isnan(s.q[3])) atomicMin(first_nan, t);
        st[t] = s;
    }
    __syncthreads();
    // only the FIRST NaN frame is patched with its predecessor (numpy index -1 = last frame when it is frame 0),
    // then frame 0 becomes the identity (motion_representation.py:212-216)
    if (threadIdx.x == 0) {
        const int k = *first_nan;
        if (k < T) {
            const int src = (k == 0) ? T - 1 : k - 1;
#pragma unroll
            for (int i = 0; i < 4; ++i) st[k].q[i] = st[src].q[i];
        }
        st[0].q[0] = 1.f; st[0].q[1] = 0.f; st[0].q[2] = 0.f; st[0].q[3] = 0.f;
    }
    __syncthreads();

    // ---- phase 2: the 22 trajectory channels of frames 0 .. T-2 ----------------------------------------------
    for (int t = threadIdx.x; t < T - 1; t += blockDim.x) {
        const FrameState a = st[t], n = st[t + 1];
        double ch[kTrajCh];
        ch[0] = (double)atan2f(a.q[3], a.q[0]);                                    // root_rot_angle
        // qmul(q[t+1], qinv(q[t])) (quaternion.py:31-49): terms[i][j] = r_i q_j with r = qinv(q[t]), q = q[t+1]
        const float r[4] = {a.q[0], -a.q[1], -a.q[2], -a.q[3]};
        const float* q = n.q;
        const float vw = sub(sub(sub(mul(r[0], q[0]), mul(r[1], q[1])), mul(r[2], q[2])), mul(r[3], q[3]));
        const float vz = add(add(sub(mul(r[0], q[3]), mul(r[1], q[2])), mul(r[2], q[1])), mul(r[3], q[0]));
        ch[1] = (double)atan2f(vz, vw);                                            // root_rot_angle_vel
        ch[2] = a.p0[0]; ch[3] = a.p0[1];                                          // root_l_pos
        const float dp[3] = {sub(n.p0[0], a.p0[0]), sub(n.p0[1], a.p0[1]), sub(n.p0[2], a.p0[2])};
        float lv[3];
        qrot_rn(n.q, dp, lv);                                                      // rotated by the NEXT frame's q (:222-223)
        ch[4] = lv[0]; ch[5] = lv[1];                                              // root_l_vel
        ch[6] = a.p0[2];                                                           // root_height
        ch[7] = a.R[0]; ch[8] = a.R[1]; ch[9] = a.R[3]; ch[10] = a.R[4]; ch[11] = a.R[6]; ch[12] = a.R[7];   // smplx_rot_6d
        // angular velocity: vee((R[t+1] - R[t]) R[t]^T), float64 (utils/other_utils.py:264-277)
        double dR[9], W[9];
#pragma unroll
        for (int i = 0; i < 9; ++i) dR[i] = (double)n.R[i] - (double)a.R[i];
#pragma unroll
        for (int i = 0; i < 3; ++i)
#pragma unroll
            for (int j = 0; j < 3; ++j)
                W[i * 3 + j] = dR[i * 3] * (double)a.R[j * 3] + dR[i * 3 + 1] * (double)a.R[j * 3 + 1] +
                               dR[i * 3 + 2] * (double)a.R[j * 3 + 2];
        ch[13] = (-W[5] + W[7]) / 2.0; ch[14] = (W[2] - W[6]) / 2.0; ch[15] = (-W[1] + W[3]) / 2.0;   // smplx_rot_vel
#pragma unroll
        for (int c = 0; c < 3; ++c) { ch[16 + c] = a.tr[c]; ch[19 + c] = (double)sub(n.tr[c], a.tr[c]); }
        float* o = out + (size_t)b * osb + (size_t)t * ost;
#pragma unroll
        for (int c = 0; c < kTrajCh; ++c) o[(size_t)c * osc] = (float)((ch[c] - (double)mean_out[c]) / (double)std_out[c]);
    }
}

// joints [B, T, 22, 3] of recover_from_repr_smpl (motion_representation.py:332-398) straight from the 294-channel
// representation: mode 0 = 'smplx_params' (6-D rotations -> FK + transl), mode 1 = 'joint_abs_traj' (local joint
// positions rotated back by the root angle and moved to the root position).  mean/stdv may be null (input already
// de-normalised).
__global__ __launch_bounds__(64) void repr_joints_kernel(const float* __restrict__ repr, long long isb, long long ist,
                                                         long long isc, const float* __restrict__ mean,
                                                         const float* __restrict__ stdv, const float* __restrict__ Jt,
                                                         const float* __restrict__ Js, const int* __restrict__ parents,
                                                         float* __restrict__ out, int mode, int B, int T) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= B * T) return;
    const int b = idx / T, t = idx % T;
    const float* x = repr + (size_t)b * isb + (size_t)t * ist;
    auto ldc = [&](int c) {
        const float v = x[(size_t)c * isc];
        return mean ? add(mul(v, stdv[c]), mean[c]) : v;
    };
    float* o = out + (size_t)idx * NJ * 3;
    if (mode == 0) {
        FrameIn in;
#pragma unroll
        for (int k = 0; k < 6; ++k) in.x6[0][k] = ldc(CH_ROT6D + k);
        for (int j = 1; j < NJ; ++j)
#pragma unroll
            for (int k = 0; k < 6; ++k) in.x6[j][k] = ldc(CH_POSE6D + (j - 1) * 6 + k);
#pragma unroll
        for (int k = 0; k < NBETA; ++k) in.beta[k] = ldc(CH_BETAS + k);
#pragma unroll
        for (int k = 0; k < 3; ++k) in.trans[k] = ldc(CH_TRANS + k);
        FkCtx f;
        smplx_fk(in, Jt, Js, parents, f);
        for (int j = 0; j < NJ; ++j)
#pragma unroll
            for (int c = 0; c < 3; ++c) o[j * 3 + c] = add(f.P[j][c], in.trans[c]);
    } else {
        const float ang = ldc(CH_ROOT_ANG);
        const float pos[3] = {ldc(CH_ROOT_POS), ldc(CH_ROOT_POS + 1), ldc(CH_ROOT_H)};
        o[0] = pos[0]; o[1] = pos[1]; o[2] = pos[2];
        for (int j = 1; j < NJ; ++j) {
            const float v[3] = {ldc(CH_LOCAL + 3 * j), ldc(CH_LOCAL + 3 * j + 1), ldc(CH_LOCAL + 3 * j + 2)};
            abs_joint(ang, pos, v, o + j * 3);
        }
    }
}

// mode 2 = 'joint_rel_traj' (motion_representation.py:312-329,349-371): the root angle / position are RUNNING SUMS over
// the earlier frames (torch.cumsum, float32, sequential) -- one workgroup per clip: thread 0 scans the T frames into
// LDS in the reference's order, then one thread per frame places the local joints.
constexpr int CH_ROOT_ANG_VEL = 1, CH_ROOT_VEL = 4;
__global__ __launch_bounds__(256) void repr_joints_rel_kernel(const float* __restrict__ repr, long long isb, long long ist,
                                                              long long isc, const float* __restrict__ mean,
                                                              const float* __restrict__ stdv, float* __restrict__ out,
                                                              int T) {
    extern __shared__ __attribute__((aligned(16))) float sroot[];     // [T][4] = angle, x, y, (unused)
    const int b = blockIdx.x;
    const float* xb = repr + (size_t)b * isb;
    auto ldc = [&](int t, int c) {
        const float v = xb[(size_t)t * ist + (size_t)c * isc];
        return mean ? add(mul(v, stdv[c]), mean[c]) : v;
    };
    if (threadIdx.x == 0) {
        float ang = 0.f, px = 0.f, py = 0.f;
        for (int t = 0; t < T; ++t) {
            float vx = 0.f, vy = 0.f;
            if (t > 0) {
                ang = add(ang, ldc(t - 1, CH_ROOT_ANG_VEL));
                vx = ldc(t - 1, CH_ROOT_VEL);
                vy = ldc(t - 1, CH_ROOT_VEL + 1);
            }
            // r_pos increment = qrot(qinv(q_t), (vx, vy, 0)), q_t = (cos ang, 0, 0, sin ang); then cumsum
            const float q[4] = {cosf(ang), 0.f, 0.f, -sinf(ang)};
            const float v[3] = {vx, vy, 0.f};
            float r[3];
            qrot_rn(q, v, r);
            px = add(px, r[0]);
            py = add(py, r[1]);
            sroot[t * 4] = ang; sroot[t * 4 + 1] = px; sroot[t * 4 + 2] = py;
        }
    }
    __syncthreads();
    for (int t = threadIdx.x; t < T; t += blockDim.x) {
        const float ang = sroot[t * 4];
        const float pos[3] = {sroot[t * 4 + 1], sroot[t * 4 + 2], ldc(t, CH_ROOT_H)};
        float* o = out + ((size_t)b * T + t) * NJ * 3;
        o[0] = pos[0]; o[1] = pos[1]; o[2] = pos[2];
        for (int j = 1; j < NJ; ++j) {
            const float v[3] = {ldc(t, CH_LOCAL + 3 * j), ldc(t, CH_LOCAL + 3 * j + 1), ldc(t, CH_LOCAL + 3 * j + 2)};
            abs_joint(ang, pos, v, o + j * 3);
        }
    }
}

}  // namespace rohm

using namespace rohm;

extern "C" int rohm_repr_joints(const rohm_smplx_t* h, const float* repr, long long in_stride_b, long long in_stride_t,
                                long long in_stride_c, const float* mean294, const float* std294, int B, int T, int mode,
                                float* joints, rohm_stream_t stream) {
    ROHM_ARG_CHECK(repr && joints, "repr_joints: null argument");
    ROHM_ARG_CHECK(mode != 0 || h, "repr_joints: mode 0 ('smplx_params') needs a body-model handle");
    ROHM_ARG_CHECK(mode >= 0 && mode <= 2, "repr_joints: mode must be 0 (smplx_params), 1 (joint_abs_traj) or 2 (joint_rel_traj)");
    ROHM_ARG_CHECK((mean294 == nullptr) == (std294 == nullptr), "repr_joints: pass both mean and std or neither");
    if (B <= 0 || T <= 0) return ROHM_OK;
    const int n = B * T;
    prof::Scope ps("repr_joints", 0.0, 4.0 * n * (155 + 66), (hipStream_t)stream);
    if (mode == 2) {
        ROHM_ARG_CHECK(T <= 2048, "repr_joints: 'joint_rel_traj' supports T <= 2048");
        hipLaunchKernelGGL(repr_joints_rel_kernel, dim3(B), dim3(256), (size_t)T * 4 * sizeof(float), (hipStream_t)stream, repr,
                           in_stride_b, in_stride_t, in_stride_c, mean294, std294, joints, T);
        ROHM_LAUNCH_CHECK();
        return ROHM_OK;
    }
    hipLaunchKernelGGL(repr_joints_kernel, dim3((n + 63) / 64), dim3(64), 0, (hipStream_t)stream, repr, in_stride_b,
                       in_stride_t, in_stride_c, mean294, std294, h ? h->d_Jt : nullptr, h ? h->d_Js : nullptr,
                       h ? h->d_parents : nullptr, joints, mode, B, T);
    ROHM_LAUNCH_CHECK();
    return ROHM_OK;
}

extern "C" int rohm_traj_rederive(const rohm_smplx_t* h, const float* repr, long long in_stride_b, long long in_stride_t,
                                  long long in_stride_c, const float* mean_in, const float* std_in, const float* mean_out,
                                  const float* std_out, int B, int T, float* out, long long out_stride_b,
                                  long long out_stride_t, long long out_stride_c, rohm_stream_t stream) {
    ROHM_ARG_CHECK(h && repr && mean_in && std_in && mean_out && std_out && out, "traj_rederive: null argument");
    ROHM_ARG_CHECK(B > 0 && T >= 2 && T <= 800, "traj_rederive: need B > 0 and 2 <= T <= 800 (got B=%d T=%d)", B, T);
    const size_t lds = (size_t)T * sizeof(FrameState) + 16;
    prof::Scope ps("traj_rederive", 0.0, 4.0 * B * ((double)T * 155 + (double)(T - 1) * kTrajCh), (hipStream_t)stream);
    hipLaunchKernelGGL(traj_rederive_kernel, dim3(B), dim3(256), lds, (hipStream_t)stream, repr, in_stride_b, in_stride_t,
                       in_stride_c, mean_in, std_in, mean_out, std_out, h->d_Jt, h->d_Js, h->d_parents, out, out_stride_b,
                       out_stride_t, out_stride_c, T);
    ROHM_LAUNCH_CHECK();
    return ROHM_OK;
}
