// Private to the TrajNet sources (trajnet.hip: create / forward / the launch-per-layer sample loop; trajnet_resident.hip: the
// clip-resident step kernel): the handle, the re-laid-out weights, the workspace map.
#pragma once
#include "common.h"

namespace rohm {

constexpr int kPadC = 64;      // row width of <=32-channel activations that are GEMM inputs
constexpr int kPadCtl = 320;   // 272 control channels padded to the 64-wide K chunk

struct ConvW {          // re-laid-out conv weight: [cout, taps * cin_pad] + bias [cout]
    float* w = nullptr;
    float* b = nullptr;
    int cin = 0, cin_pad = 0, cout = 0, taps = 0;
};
struct UpW { ConvW even, odd, both; };    // ConvTranspose1d(k4, s2, p1) as two 2-tap phases; `both` = the two phases as ONE
                                          // 3-tap GEMM with 2 C output columns (phase 1 stored to the next output row)
struct BlockW {                           // Conv1dBlock: conv5 + GroupNorm(8)
    ConvW conv;
    float *g = nullptr, *be = nullptr;
};
struct ResW {                             // ResidualTemporalBlock (b0res: block-0 conv and the 1x1 residual conv as ONE GEMM)
    BlockW b0, b1;
    ConvW res;                            // 1x1 when cin != cout (taps == 0 -> absent)
    ConvW b0res;                          // [2 cout, 5 cin_pad]: rows [0, cout) = b0.conv, rows [cout, 2 cout) = res at the centre tap
    bool has_res = false;
    int tb_off = -1;                      // offset of this block's time bias inside tb_all, -1 = no time input
    int cin = 0, cout = 0;
};

}  // namespace rohm

struct rohm_trajnet {
    int mid, tdim, ctraj, cctrl, control, device;
    float* arena = nullptr;
    size_t arena_floats = 0;
    float* zero_page = nullptr;
    // time path
    float *t_w1T, *t_b1, *t_w3T, *t_b3;   // time_mlp.{1,3} stored [in][out]
    float *tb_wT, *tb_b;                  // all per-block time Linears stacked: [tdim][tb_total], [tb_total]
    int tb_total = 0;
    rohm::ResW cond_enc[4], diff_enc[4], mid_blk[2], dec[4], c_enc[4], c_mid[2];
    rohm::ConvW cond_down[3], diff_down[4], c_down[4];
    rohm::UpW up[4];
    rohm::BlockW final_blk;
    rohm::ConvW final_conv, c_zero0, c_zero[4], c_zero_mid;
};

namespace rohm {

__device__ __forceinline__ float mishf(float x) {
    // x * tanh(softplus(x)), softplus with torch's threshold 20 (model/heads.py:104, nn.Mish).  With e = exp(x):
    // tanh(log(1 + e)) = ((1 + e)^2 - 1) / ((1 + e)^2 + 1) = n / (n + 2), n = e (e + 2) -- one exp and one division instead
    // of expf + log1pf + tanhf (~150 VALU instructions per value in the library form, the bulk of the GroupNorm
    // kernel's time with one wave per SIMD); within 1 ulp of the library form over [-100, 100] (checked on the host).
    const float e = __expf(fminf(x, 20.f));
    const float n = e * (e + 2.f);
    const float r = x * __fdividef(n, n + 2.f);
    return (x > 20.f) ? x : r;
}

constexpr int kTbSteps = 128;             // loop steps whose time path is evaluated by one launch

struct Scratch { float *ya, *hb, *rc; };   // conv output, block-0 activation, 1x1 residual

// ---- workspace ---------------------------------------------------------------------------------------------
struct TWs {
    float *xin, *cin, *ctl;                 // padded inputs [M,32], [M,32], [M,288]
    float *cat[4], *dcat[4], *ccat[4];      // concat buffers (see forward)
    float *cdn[3], *ddn[4], *kdn[4];        // outputs of the down convs (cond / diff / control)
    float *mid_a, *mid_b, *kmid_a, *kmid_b;
    float *d[4];                            // decoder block outputs
    float *cz, *ctrl[4], *ctrl_mid;         // control residuals
    float *ctrl_b[4], *ctrl_mid_b;          // ... second set: the ControlNet branch of the sample loop runs one step ahead on its own stream
    Scratch sc_ctl;                         // ... with its own block scratch and split-K slabs
    float *splitk_ctl, *splitk_res_ctl;
    float *fin;                             // final conv block output [M, 32]
    float *tb_all;                          // [B or 1][tb_total]
    float *tb_steps;                        // [kTbSteps][tb_total]: time biases of a run of loop steps (one launch)
    Scratch sc;
    float *x0, *cond_keep;                  // loop: network output [B,T,13]
    float *splitk, *splitk_res;             // split-K partial tiles (plan_split)
    float *step_coef;                       // graph replay: (c1, c2, sigma) per step, timesteps, step counter
    int64_t* step_t;
    int* step_ctr;
    float* resident;                        // clip-resident sample loop: layer list, meeting flags, statistics slots, x_T (trajnet_resident.hip)
    size_t floats;
};

// ---- clip-resident sample loop (trajnet_resident.hip) ---------------------------------------------------------------------------
// One launch per denoising step: the workgroups of an XCD stay with that XCD's clips for the whole U-Net (+ ControlNet branch) and
// meet through its L2 between the layers.  resident_floats: extra workspace (program, flags, statistics slots).
size_t resident_floats(int B, int T);
// time biases of `run` consecutive loop steps (timesteps t[0 .. run)) into w.tb_steps: one launch (trajnet.hip)
int launch_time_path_steps(const rohm_trajnet* h, const TWs& w, const int64_t* t, int run, hipStream_t s);
// Can (and should) this loop run resident?  Shape, device layout (exchange guard), environment (ROHM_TRAJ_RESIDENT=0 switches it off).
bool resident_ok(const rohm_trajnet* h, int B, int T, int n_steps, hipStream_t s);
// The n_steps ancestral steps (cond encodings, the projected control input and the padded x_T are in the workspace already).
// Returns ROHM_ERR_UNSUPPORTED if a bounded wait expired (x restored to x_T: the caller runs the launch-per-layer loop instead).
int resident_loop(const rohm_trajnet* h, const TWs& w, float* x, const float* noise, const int64_t* t_model, const float* coef,
                  float* x0_last, float* x_in_last, int n_steps, int B, int T, hipStream_t s);

}  // namespace rohm
