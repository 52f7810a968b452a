// Evaluation metrics of the AMASS driver on the device: eval_amass_full.py:67-147.
//
// The reference copies the recovered joints of every clip to the host, pickles them and evaluates with numpy.  The
// quantities are per-clip reductions over [clip_len, 22, 3] joint tracks, so one workgroup per clip produces the
// partial sums in a single pass (after a min-height pre-pass); the host only divides sums by counts.  HBM-bound:
// 2 x clip_len x 22 x 3 x 4 B read per clip, 10 doubles written.
//
// Comparisons are done in float32 exactly as numpy does them on the float32 joint arrays (velocity and height
// thresholds of the skating test, the -0.05 m penetration test); sums are accumulated in float64.
#include "common.h"

namespace rohm {

constexpr int kMJ = 22;
constexpr int kNMetric = 10;     // layout documented in include/rohm_hip.h
__constant__ int kMFoot[4] = {7, 10, 8, 11};      // eval_amass_full.py:103

__device__ __forceinline__ double block_sum(double v, double* sh) {
    const int tid = threadIdx.x;
    __syncthreads();
    sh[tid] = v;
    __syncthreads();
    for (int o = blockDim.x / 2; o > 0; o >>= 1) {
        if (tid < o) sh[tid] += sh[tid + o];
        __syncthreads();
    }
    return sh[0];
}

__device__ __forceinline__ bool skating_at(const float* __restrict__ j, int t, float min_h) {
    // both joints of both feet faster than 0.10 m/s horizontally and lower than 0.15 m (ankle) / 0.10 m (toe)
    bool ok = true;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const float* a = j + ((size_t)t * kMJ + kMFoot[k]) * 3;
        const float* b = a + kMJ * 3;
        const float dx = __fsub_rn(b[0], a[0]), dy = __fsub_rn(b[1], a[1]);
        const float vel = __fmul_rn(sqrtf(__fadd_rn(__fmul_rn(dx, dx), __fmul_rn(dy, dy))), 30.f);
        const float h = __fsub_rn(a[2], min_h);
        const float hmax = (k & 1) ? 0.10f : (float)(0.10 + 0.05);
        ok = ok && (vel > 0.10f) && (h < hmax);
    }
    return ok;
}

__global__ __launch_bounds__(256) void amass_metrics_kernel(const float* __restrict__ jc, const float* __restrict__ jr,
                                                            const float* __restrict__ cc, long long cc_st,
                                                            const float* __restrict__ cr, long long cr_st,
                                                            unsigned occ_joint_mask, int occ_start, int occ_end,
                                                            double* __restrict__ out, int T) {
    __shared__ double sh[256];
    __shared__ float sh_min;
    const int b = blockIdx.x, tid = threadIdx.x;
    const float* c = jc + (size_t)b * T * kMJ * 3;
    const float* r = jr + (size_t)b * T * kMJ * 3;
    // min height of the clean clip over all frames and joints (:105)
    float mn = INFINITY;
    for (int i = tid; i < T * kMJ; i += blockDim.x) mn = fminf(mn, c[(size_t)i * 3 + 2]);
    __shared__ float shf[256];
    shf[tid] = mn;
    __syncthreads();
    for (int o = blockDim.x / 2; o > 0; o >>= 1) {
        if (tid < o) shf[tid] = fminf(shf[tid], shf[tid + o]);
        __syncthreads();
    }
    if (tid == 0) sh_min = shf[0];
    __syncthreads();
    const float min_h = sh_min;

    double s_all = 0, s_occ = 0, n_occ = 0, n_contact = 0, n_sk_gt = 0, n_sk_rec = 0, s_acc = 0, n_pene = 0, s_pene = 0;
    for (int i = tid; i < T * kMJ; i += blockDim.x) {
        const int t = i / kMJ, j = i % kMJ;
        const float* pc = c + (size_t)i * 3;
        const float* pr = r + (size_t)i * 3;
        const float dx = pc[0] - pr[0], dy = pc[1] - pr[1], dz = pc[2] - pr[2];
        const double e = sqrt((double)dx * dx + (double)dy * dy + (double)dz * dz);
        s_all += e;
        const bool occ = ((occ_joint_mask >> j) & 1u) || (t >= occ_start && t < occ_end);
        if (occ) { s_occ += e; n_occ += 1.0; }
        if (t + 2 < T) {      // acceleration error (:135-139): second differences x fps^2
            const size_t s1 = (size_t)kMJ * 3, s2 = 2 * s1;
            double a2 = 0;
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                const float ar = (pr[s2 + k] - 2.f * pr[s1 + k] + pr[k]) * 900.f;
                const float ac = (pc[s2 + k] - 2.f * pc[s1 + k] + pc[k]) * 900.f;
                const double d = (double)ar - (double)ac;
                a2 += d * d;
            }
            s_acc += sqrt(a2);
        }
        if (j == 10 || j == 11) {   // ground penetration of the toes of the reconstruction (:141-147)
            const float d = __fsub_rn(pr[2], min_h);
            if (d < -0.05f) n_pene += 1.0;
            if (d < 0.f) s_pene += (double)d;
        }
    }
    for (int t = tid; t < T - 1; t += blockDim.x) {
        if (skating_at(c, t, min_h)) n_sk_gt += 1.0;
        if (skating_at(r, t, min_h)) n_sk_rec += 1.0;
    }
    for (int i = tid; i < T * 4; i += blockDim.x) {   // contact labels (:92-98): rec thresholded at 0.5
        const int t = i / 4, k = i % 4;
        const float lr = cr[((size_t)b * T + t) * cr_st + k] > 0.5f ? 1.f : 0.f;
        if (cc[((size_t)b * T + t) * cc_st + k] == lr) n_contact += 1.0;
    }
    double vals[kNMetric] = {s_all, s_occ, n_occ, n_contact, n_sk_gt, n_sk_rec, s_acc, n_pene, s_pene, (double)min_h};
    for (int k = 0; k < kNMetric - 1; ++k) {
        const double v = block_sum(vals[k], sh);
        if (tid == 0) out[(size_t)b * kNMetric + k] = v;
    }
    if (tid == 0) out[(size_t)b * kNMetric + kNMetric - 1] = (double)min_h;
}

}  // namespace rohm

using namespace rohm;

extern "C" int rohm_amass_metrics(const float* joints_clean, const float* joints_rec, const float* contact_clean,
                                  long long contact_clean_stride, const float* contact_rec, long long contact_rec_stride,
                                  unsigned occ_joint_mask, int occ_start, int occ_end, int B, int T, double* out,
                                  rohm_stream_t stream) {
    ROHM_ARG_CHECK(joints_clean && joints_rec && contact_clean && contact_rec && out, "amass_metrics: null argument");
    ROHM_ARG_CHECK(B > 0 && T >= 3, "amass_metrics: need B > 0 and T >= 3 (got B=%d T=%d)", B, T);
    ROHM_ARG_CHECK(occ_start >= 0 && occ_end >= occ_start, "amass_metrics: bad occluded frame range");
    prof::Scope ps("amass_metrics", 0.0, 8.0 * B * T * (kMJ * 3 + 4), (hipStream_t)stream);
    hipLaunchKernelGGL(amass_metrics_kernel, dim3(B), dim3(256), 0, (hipStream_t)stream, joints_clean, joints_rec,
                       contact_clean, contact_clean_stride, contact_rec, contact_rec_stride, occ_joint_mask, occ_start,
                       occ_end, out, T);
    ROHM_LAUNCH_CHECK();
    return ROHM_OK;
}
