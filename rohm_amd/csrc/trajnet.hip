// TrajNet / TrajControl on MI355X: 1-D conv U-Net denoiser of the 13-channel trajectory
// (model/trajnet.py:10-275, model/heads.py:12-106) and its 100-step DDPM loop
// (diffusion/gaussian_diffusion_trajnet.py:440-466, 559-627).
//
// Layout: activations are CHANNELS-LAST, row m = b*T_level + t of a [B*T_level, C] matrix -- the layout the
// drivers already hand over ([B, T, 13] / [B, T, 272]), so there is no 'b h t -> b t h' transpose at all.
// Every convolution is the fp32-MFMA GEMM of gemm_f32.hip with a tap-gathered A operand (K = taps * C_in;
// zero padding comes from a page of zeros), channel concatenations are column slices of preallocated wide
// buffers that the producers write into directly, and GroupNorm + Mish + time bias + residual (+ control
// residual) are one fused kernel per conv block.  Everything that depends on the timestep only -- the
// sinusoid -> MLP embedding and the per-block time biases -- is one small launch per step.
// In the TrajControl sample loop the ControlNet branch (timestep-dependent, x_t-independent) runs on a second stream, one step ahead.
#include <math.h>
#include <algorithm>
#include <string.h>
#include <string>
#include <vector>
#include <atomic>
#include "trajnet_priv.h"

namespace rohm {

// ------------------------------------------------------------------------------------------------ kernels
// pad [rows, cin] -> [rows, cpad] (zeros beyond cin)
__global__ __launch_bounds__(256) void pad_rows_kernel(const float* __restrict__ src, float* __restrict__ dst,
                                                       size_t rows, int cin, int cpad) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= rows * cpad) return;
    const size_t r = i / cpad;
    const int c = (int)(i % cpad);
    dst[i] = (c < cin) ? src[r * cin + c] : 0.f;
}

// Timestep path of one step: SinusoidalPosEmb(32) -> Linear(32,128) -> Mish -> Linear(128,32)
// (trajnet.py:120-125, heads.py:57-69), then every block's `Mish -> Linear(32, C_out)` (heads.py:34-38)
// stacked into one [tb_total] vector.  One block per row (sample, or 1 when the batch shares t).
__global__ __launch_bounds__(256) void time_path_kernel(const int64_t* __restrict__ t_dev, int64_t t_host,
                                                        const int64_t* __restrict__ t_tab, const int* __restrict__ step_ctr,
                                                        int tdim,
                                                        const float* __restrict__ w1T, const float* __restrict__ b1,
                                                        const float* __restrict__ w3T, const float* __restrict__ b3,
                                                        const float* __restrict__ tbwT, const float* __restrict__ tbb,
                                                        int tb_total, float* __restrict__ tb_all) {
    __shared__ float e[64], h[256], te[64];
    const int tid = threadIdx.x;
    // t: per sample (t_dev), shared by the batch (t_host), or the entry of a device table selected by the step counter
    // of a replayed hipGraph
    const float tval = (float)(t_tab ? t_tab[*step_ctr] : (t_dev ? t_dev[blockIdx.x] : t_host));
    const int half = tdim / 2;
    if (tid < tdim) {
        const int k = tid % half;
        const float f = expf((float)k * -(logf(10000.f) / (float)(half - 1)));
        e[tid] = (tid < half) ? sinf(tval * f) : cosf(tval * f);
    }
    __syncthreads();
    const int hid = 4 * tdim;
    if (tid < hid) {
        float a = b1[tid];
        for (int k = 0; k < tdim; ++k) a = fmaf(e[k], w1T[k * hid + tid], a);
        h[tid] = mishf(a);
    }
    __syncthreads();
    if (tid < tdim) {
        float a = b3[tid];
        for (int k = 0; k < hid; ++k) a = fmaf(h[k], w3T[k * tdim + tid], a);
        te[tid] = mishf(a);            // every consumer applies Mish first (heads.py:35)
    }
    __syncthreads();
    for (int o = tid; o < tb_total; o += blockDim.x) {
        float a = tbb[o];
        for (int k = 0; k < tdim; ++k) a = fmaf(te[k], tbwT[(size_t)k * tb_total + o], a);
        tb_all[(size_t)blockIdx.x * tb_total + o] = a;
    }
}

// Fused GroupNorm(8 groups, eps 1e-5, biased variance over (C/8 x T) per sample) -> Mish -> (+ time bias)
// -> (+ residual) -> (+ second residual), channels-last (heads.py:98-104, 50-54; trajnet.py:240,259-271).
// One block per (sample, group).
struct GnArgs {
    const float* y; int ldy;            // conv output [B*T, ldy] -- or, with ksplit > 1, `ksplit` partial slabs of it
    int ksplit; size_t slab;            //   (slab s at y + s * slab, row stride ldy) that still lack the conv bias
    const float* cbias;
    const float *gamma, *beta;
    int C, T;
    const float* tb; int ldtb;          // [rows or 1][...] time bias slice for this block (nullable)
    const float* res; int ldres;        // residual (nullable) -- or `res_ksplit` partial slabs of a 1x1 residual conv
    int res_ksplit; size_t res_slab; const float* res_bias;
    const float* add2; int ldadd2;      // control residual (nullable)
    float* dst; int lddst;
    float* dst2; int lddst2;            // optional second destination (skip / concat copy)
};

// One block per (sample, group).  A group is (C/8 channels) x T values (1152 at every level of a 144-frame clip, up to
// 4096 supported): each thread keeps its values in
// registers (ONE pass over memory; round 1 made three, with two 8-step block reductions), the statistics are two
// wave-shuffle reductions + one LDS exchange each.  With split-K the block sums the partial slabs itself -- the
// separate reduction kernel (35 of the 98 launches of a denoising step) disappears for every conv that feeds a
// GroupNorm, and so does the round trip of the reduced tensor through memory.
constexpr int kGnMaxPerThread = 16;      // scalar units per thread: groups of up to 4096 values (T <= 512 at 8 channels per group)
constexpr int kGnVecPerThread = 4;       // 16-byte units per thread: the same 4096 values
__device__ __forceinline__ float block_sum_256(float v, float* sh) {
    v = wave_sum64(v);
    const int w = threadIdx.x >> 6;
    __syncthreads();                    // sh may still be read from the previous call
    if ((threadIdx.x & 63) == 0) sh[w] = v;
    __syncthreads();
    return (sh[0] + sh[1]) + (sh[2] + sh[3]);
}

// VEC: 16-byte accesses (every leading dimension a multiple of 4 floats, 16-byte aligned bases -- the layouts this
// file builds); the scalar variant is the fallback for foreign strides.
//
// The kernel is a latency chain (B x 8 blocks, a few values per thread), so its memory round trips are what it costs:
// EVERY load the block needs -- the conv output or its split-K slabs (8 slabs in flight per unit, summed in slab order),
// the residual or ITS slabs, gamma / beta / time bias / control residual -- is issued before the first reduction, and the
// statistics are computed while the epilogue operands are still in flight.
template <bool VEC>
__global__ __launch_bounds__(256) void gn_mish_kernel(GnArgs a) {
    constexpr int W = VEC ? 4 : 1;                  // floats per unit
    constexpr int NU = VEC ? kGnVecPerThread : kGnMaxPerThread;   // units per thread (unused ones are predicated off)
    constexpr int SB = VEC ? 8 : 2;                 // slabs in flight per unit
    const int b = blockIdx.x, g = blockIdx.y;
    const int cg = a.C >> 3;                        // channels per group: a power of two (4 .. 64)
    const int ug = cg / W;                          // units per row of the group
    const int sh_ug = 31 - __clz(ug);
    const int n = cg * a.T, nu = n / W;
    __shared__ float sh[4];
    auto ldv = [&](const float* p, float* o) {
        if constexpr (VEC) { const f32x4 t = *reinterpret_cast<const f32x4*>(p); o[0] = t[0]; o[1] = t[1]; o[2] = t[2]; o[3] = t[3]; }
        else o[0] = *p;
    };
    auto active = [&](int k) { return (int)threadIdx.x + k * 256 < nu; };
    auto unit_col = [&](int k) { return g * cg + (((int)threadIdx.x + k * 256) & (ug - 1)) * W; };
    auto unit_row = [&](int k) { return (size_t)b * a.T + (((int)threadIdx.x + k * 256) >> sh_ug); };

    // ---- phase 1: all loads ------------------------------------------------------------------------------------------
    float v[NU][W];
#pragma unroll
    for (int k = 0; k < NU; ++k) {
#pragma unroll
        for (int e = 0; e < W; ++e) v[k][e] = 0.f;
        if (active(k)) ldv(a.y + unit_row(k) * a.ldy + unit_col(k), v[k]);
    }
    if (a.ksplit > 1) {       // v = ((((slab 0 + slab 1) + slab 2) + ...) + bias: fixed order, deterministic
        for (int sp = 1; sp < a.ksplit; sp += SB) {
            float t[NU][SB][W];
#pragma unroll
            for (int k = 0; k < NU; ++k)
                if (active(k)) {
                    const float* p = a.y + unit_row(k) * a.ldy + unit_col(k);
#pragma unroll
                    for (int j = 0; j < SB; ++j) {     // past the last slab: re-read it (an L2 hit), weight 0
                        const int sj = (sp + j < a.ksplit) ? sp + j : a.ksplit - 1;
                        ldv(p + (size_t)sj * a.slab, t[k][j]);
                    }
                }
#pragma unroll
            for (int k = 0; k < NU; ++k)
                if (active(k)) {
#pragma unroll
                    for (int j = 0; j < SB; ++j) {
                        const bool ok = sp + j < a.ksplit;
#pragma unroll
                        for (int e = 0; e < W; ++e) v[k][e] += ok ? t[k][j][e] : 0.f;
                    }
                }
        }
#pragma unroll
        for (int k = 0; k < NU; ++k)
            if (active(k)) {
                float cb[W];
                ldv(a.cbias + unit_col(k), cb);
#pragma unroll
                for (int e = 0; e < W; ++e) v[k][e] += cb[e];
            }
    }
    // epilogue operands: issued now, consumed after the two reductions
    float ga[NU][W], be[NU][W], tbv[NU][W], rr[NU][W], a2[NU][W];
#pragma unroll
    for (int k = 0; k < NU; ++k) {
#pragma unroll
        for (int e = 0; e < W; ++e) { ga[k][e] = 0.f; be[k][e] = 0.f; tbv[k][e] = 0.f; rr[k][e] = 0.f; a2[k][e] = 0.f; }
        if (active(k)) {
            const int c = unit_col(k);
            ldv(a.gamma + c, ga[k]);
            ldv(a.beta + c, be[k]);
            if (a.tb) ldv(a.tb + (size_t)(a.ldtb ? b : 0) * a.ldtb + c, tbv[k]);
            if (a.res) ldv(a.res + unit_row(k) * a.ldres + c, rr[k]);
            if (a.add2) ldv(a.add2 + unit_row(k) * a.ldadd2 + c, a2[k]);
        }
    }
    if (a.res && a.res_ksplit > 1) {          // residual = 1x1 conv left as split-K slabs: same fixed-order sum + its bias
        constexpr int SR = VEC ? 4 : 2;
        for (int sp = 1; sp < a.res_ksplit; sp += SR) {
            float t[NU][SR][W];
#pragma unroll
            for (int k = 0; k < NU; ++k)
                if (active(k)) {
                    const float* p = a.res + unit_row(k) * a.ldres + unit_col(k);
#pragma unroll
                    for (int j = 0; j < SR; ++j) {
                        const int sj = (sp + j < a.res_ksplit) ? sp + j : a.res_ksplit - 1;
                        ldv(p + (size_t)sj * a.res_slab, t[k][j]);
                    }
                }
#pragma unroll
            for (int k = 0; k < NU; ++k)
                if (active(k)) {
#pragma unroll
                    for (int j = 0; j < SR; ++j) {
                        const bool ok = sp + j < a.res_ksplit;
#pragma unroll
                        for (int e = 0; e < W; ++e) rr[k][e] += ok ? t[k][j][e] : 0.f;
                    }
                }
        }
#pragma unroll
        for (int k = 0; k < NU; ++k)
            if (active(k)) {
                float x[W];
                ldv(a.res_bias + unit_col(k), x);
#pragma unroll
                for (int e = 0; e < W; ++e) rr[k][e] += x[e];
            }
    }

    // ---- phase 2: statistics (two-pass: mean, then the centred sum of squares; biased variance) ------------------------
    float s = 0.f;
#pragma unroll
    for (int k = 0; k < NU; ++k)
        if (active(k))
#pragma unroll
            for (int e = 0; e < W; ++e) s += v[k][e];
    const float mean = block_sum_256(s, sh) / (float)n;
    float q = 0.f;
#pragma unroll
    for (int k = 0; k < NU; ++k)
        if (active(k))
#pragma unroll
            for (int e = 0; e < W; ++e) {
                const float d = v[k][e] - mean;
                q += d * d;
            }
    const float rstd = 1.0f / sqrtf(block_sum_256(q, sh) / (float)n + 1e-5f);

    // ---- phase 3: normalise, Mish, (+ time bias) (+ residual) (+ control residual), store -------------------------------
#pragma unroll
    for (int k = 0; k < NU; ++k) {
        if (!active(k)) continue;
        const int c = unit_col(k);
        const size_t row = unit_row(k);
        float o[W];
#pragma unroll
        for (int e = 0; e < W; ++e) o[e] = mishf((v[k][e] - mean) * rstd * ga[k][e] + be[k][e]);
        if (a.tb) {
#pragma unroll
            for (int e = 0; e < W; ++e) o[e] += tbv[k][e];
        }
        if (a.res) {
#pragma unroll
            for (int e = 0; e < W; ++e) o[e] += rr[k][e];
        }
        if (a.add2) {
#pragma unroll
            for (int e = 0; e < W; ++e) o[e] += a2[k][e];
        }
        if constexpr (VEC) {
            *reinterpret_cast<f32x4*>(a.dst + row * a.lddst + c) = f32x4{o[0], o[1], o[2], o[3]};
            if (a.dst2) *reinterpret_cast<f32x4*>(a.dst2 + row * a.lddst2 + c) = f32x4{o[0], o[1], o[2], o[3]};
        } else {
            a.dst[row * a.lddst + c] = o[0];
            if (a.dst2) a.dst2[row * a.lddst2 + c] = o[0];
        }
    }
}

// out = a + b over [rows, C] (control residual after the middle blocks, trajnet.py:240)
__global__ __launch_bounds__(256) void add_kernel(float* __restrict__ a, const float* __restrict__ b, size_t n) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) a[i] += b[i];
}

// The last three launches of a sampling-loop step as ONE: the head's Conv1d(32, c_traj, 1) (trajnet.py:158-161), the
// ancestral update x_{t-1} = c1 x0 + c2 x_t + sigma noise (gaussian_diffusion_trajnet.py:440-466) and the zero-padded copy
// of x_{t-1} that the next step's first convolution reads.  One thread per (row, channel); the per-step values are either
// immediates or -- for the replayed hipGraph -- entries of device tables selected by the step counter.
struct TailArgs {
    const float* fin; int ldfin;         // final block output [M, ldfin]
    const float* w; int ldw; int cin;    // head weight [c_traj, ldw] (cin <= 64 used columns), bias [c_traj]
    const float* bias; int ctraj;
    float* x;                            // [M, c_traj]: x_t in, x_{t-1} out
    float* xin; int ldxin;               // padded copy [M, ldxin] (columns >= c_traj stay zero)
    float* x0_out;                       // nullable: the head's output itself
    const float* noise;                  // immediates: this step's noise (nullable)
    float c1, c2, sigma;
    const float* noise_base; const float* coef_tab; const int* step_ctr;   // tables (coef_tab != nullptr selects them)
    size_t M;
};
__global__ __launch_bounds__(256) void traj_tail_kernel(TailArgs a) {
    __shared__ float ws[32 * 65];
    for (int i = threadIdx.x; i < a.ctraj * a.cin; i += 256) ws[(i / a.cin) * 65 + i % a.cin] = a.w[(size_t)(i / a.cin) * a.ldw + i % a.cin];
    __syncthreads();
    float c1 = a.c1, c2 = a.c2, sigma = a.sigma;
    const float* noise = a.noise;
    const size_t n = a.M * a.ctraj;
    if (a.coef_tab) {
        const int k = *a.step_ctr;
        c1 = a.coef_tab[3 * k]; c2 = a.coef_tab[3 * k + 1]; sigma = a.coef_tab[3 * k + 2];
        noise = a.noise_base ? a.noise_base + (size_t)k * n : nullptr;
    }
    if (sigma == 0.f) noise = nullptr;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        const size_t m = i / a.ctraj;
        const int c = (int)(i - m * a.ctraj);
        const float* f = a.fin + m * a.ldfin;
        const float* wc = ws + c * 65;
        float acc = 0.f;
        for (int k = 0; k < a.cin; k += 4) {
            const f32x4 fv = *reinterpret_cast<const f32x4*>(f + k);
            acc = fmaf(fv[0], wc[k], acc); acc = fmaf(fv[1], wc[k + 1], acc);
            acc = fmaf(fv[2], wc[k + 2], acc); acc = fmaf(fv[3], wc[k + 3], acc);
        }
        const float x0 = acc + a.bias[c];
        if (a.x0_out) a.x0_out[i] = x0;
        float v = c1 * x0 + c2 * a.x[i];
        if (noise) v += sigma * noise[i];
        a.x[i] = v;
        a.xin[m * a.ldxin + c] = v;
    }
}

// ---- weight re-layout (create time) --------------------------------------------------------------------
__global__ void relayout_conv_kernel(const float* __restrict__ w, float* __restrict__ out, int cout, int cin, int k,
                                     int cin_pad, int tap0, int tap_step, int ntap, int transposed) {
    // out[co][jj*cin_pad + ci] = w[co][ci][tap0 + jj*tap_step]   (Conv1d, weight [cout, cin, k])
    //                          = w[ci][co][tap0 + jj*tap_step]   (ConvTranspose1d, weight [cin, cout, k])
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    const int K = ntap * cin_pad;
    if (i >= cout * K) return;
    const int co = i / K, r = i % K, jj = r / cin_pad, ci = r % cin_pad;
    float v = 0.f;
    if (ci < cin) {
        const int j = tap0 + jj * tap_step;
        v = transposed ? w[((size_t)ci * cout + co) * k + j] : w[((size_t)co * cin + ci) * k + j];
    }
    out[i] = v;
}

__global__ void transpose2_kernel(const float* __restrict__ src, float* __restrict__ dst, int R, int Cc, int ldd,
                                  int col0) {
    // dst[c][col0 + r] = src[r][c]
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < R * Cc) {
        const int r = i / Cc, c = i % Cc;
        dst[(size_t)c * ldd + col0 + r] = src[i];
    }
}

// ---- launch helpers --------------------------------------------------------------------------------------
static inline size_t al(size_t n) { return (n + 63) / 64 * 64; }

// Split-K plan for the few-tile, long-K convolutions of the deep levels (T <= 36: 16-32 output tiles on 256 CUs):
// enough splits to give every CU a workgroup, at least 4 K chunks per split.  The partial-tile buffer is part of the
// caller's workspace; forward() publishes it here for the launch helpers of this host thread.
constexpr size_t kSplitKFloats = (size_t)256 * 144 * 128;
constexpr size_t kFuseMaxWork = (size_t)64 * 144 * 64;   // rows x channels of a level up to which the fused conv forms pay (B <= 64)
constexpr int kGraphMaxSteps = 1024;       // steps per sample-loop call that the captured-graph path accepts
static thread_local int tl_loop_mode = 0;             // form the last sample loop of this host thread ran in (rohm_trajnet_loop_mode)
static thread_local float* tl_splitk = nullptr;       // partial slabs of the conv feeding the next kernel
static thread_local float* tl_splitk_res = nullptr;   // partial slabs of a block's 1x1 residual conv (alive until its 2nd GroupNorm)

// A conv whose consumer is the GroupNorm kernel leaves its split-K slabs un-reduced; this says where they are.
struct SplitInfo { int S = 1; size_t slab = 0; int ldp = 0; const float* base = nullptr; };

// Launch shape of the latency-bound convolutions (rohm_trajnet_tune): workgroups per CU the conv GEMMs may occupy (a
// 144 x 64 tile needs 60 KB of LDS, so two fit a CU; one is pinned by padding the LDS request) and the fewest K chunks a
// split-K slice may get.
static std::atomic<int> g_conv_wg_per_cu{1};      // launch-shape knobs (rohm_trajnet_tune): read by every launch, possibly from
static std::atomic<int> g_split_min_chunks{2};    // several host threads (one stream / workspace each) -> atomics, relaxed
// g_split_min_chunks:      // measured: B = 32 loop 69.0 ms at 4, 66.7 at 2; B = 1 54.1 / 49.6
static std::atomic<int> g_split_pow2{0};      // 1: split counts are powers of two, so that split = block % S stays tied to XCD = block % 8

static void plan_split(GemmParams& g, float* buf = nullptr, SplitInfo* defer = nullptr) {
    const int tm = (g.M + 143) / 144;
    const int bn = (tm * ((g.N + 127) / 128) >= 256 && g.N % 128 == 0) ? 128 : 64;
    const int tiles = tm * ((g.N + bn - 1) / bn);
    const int wg_per_cu = g_conv_wg_per_cu.load(std::memory_order_relaxed);
    const int slots = 256 * wg_per_cu;
    g.wg_per_cu = (tiles <= slots) ? wg_per_cu : 1;      // multi-round grids keep one workgroup per CU (gemm_f32.hip)
    if (!buf) buf = tl_splitk;
    if (!buf) return;
    const int nk = g.K / 32;
    int S = std::min(slots / tiles, nk / g_split_min_chunks.load(std::memory_order_relaxed));
    if (g_split_pow2.load(std::memory_order_relaxed) && S > 1) S = 1 << (31 - __builtin_clz((unsigned)S));
    const int ldp = (g.N + 3) / 4 * 4;
    if (tiles > slots / 2 || S < 2 || (size_t)S * g.M * ldp > kSplitKFloats) return;
    g.ksplit = S; g.partial = buf; g.ld_partial = ldp;
    if (defer) {
        g.ksplit_defer = 1;
        defer->S = S; defer->slab = (size_t)g.M * ldp; defer->ldp = ldp; defer->base = buf;
    }
}

// Split plan of a fused [conv5 | 1x1 residual] launch (res_block): the residual columns' weights are zero outside the centre tap, so
// their tiles walk K / 5 chunks and the workgroup slots they do not need go to the conv tiles as extra splits.  Returns false (plain
// plan_split) where the two column ranges share tiles (cout % 64) or nothing is split anyway.
static std::atomic<int> g_res_centre_tap{1};      // ROHM_TRAJ_RES_TAP=0 (read at create) keeps the uniform plan
static bool plan_fused_residual(GemmParams& g, int co, float* buf, SplitInfo* defer, SplitInfo* defer_res) {
    if (!g_res_centre_tap.load(std::memory_order_relaxed) || !buf || !defer || !defer_res || co % 64 != 0 || g.conv_taps != 5) return false;
    const int tm = (g.M + 143) / 144, t = tm * (co / 64);            // conv tiles = residual tiles = t (144 x 64 tiles)
    const int slots = 256, nkr = g.conv_cin_pad / 32, nkc = g.K / 32, minc = g_split_min_chunks.load(std::memory_order_relaxed);
    if (2 * t > slots / 2) return false;
    // split counts that fill the slots and balance the chunks per workgroup of the two kinds of tile: smallest max(conv, residual) chunks
    int Sr = 1, Sc = 0, best = 1 << 30;
    for (int r = 1; r <= std::max(1, nkr / minc) && t * (r + 2) <= slots; ++r) {
        const int c = std::min((slots - t * r) / t, nkc / minc);
        if (c < 2 || c < r) continue;
        const int cost = std::max((nkc + c - 1) / c, (nkr + r - 1) / r);
        if (cost < best) { best = cost; Sr = r; Sc = c; }
    }
    if (Sc < 2) return false;
    const int ldp = (g.N + 3) / 4 * 4;
    if ((size_t)Sc * g.M * ldp > kSplitKFloats) return false;
    g.wg_per_cu = 1;
    g.ksplit = Sc; g.partial = buf; g.ld_partial = ldp; g.ksplit_defer = 1;
    g.res_col0 = co; g.res_ksplit = Sr; g.res_k0 = 2 * g.conv_cin_pad; g.res_nk = nkr;
    defer->S = Sc; defer->slab = (size_t)g.M * ldp; defer->ldp = ldp; defer->base = buf;
    *defer_res = *defer;
    defer_res->S = Sr;             // Sr == 1: the residual columns are final (bias added) in C, not in a slab
    return true;
}

static int conv_gemm(const rohm_trajnet* h, const ConvW& w, const float* x, int ldx, int B, int tin, int tq,
                     int stride, const int* offs, float* out, int ldo, int orow_mul, int orow_add, hipStream_t s,
                     SplitInfo* defer = nullptr, float* splitk_buf = nullptr, int ncol_split = 0, int ncol_jump = 0,
                     int fused_res_co = 0, SplitInfo* defer_res = nullptr) {
    GemmParams g{};
    g.A = x; g.lda = ldx; g.W = w.w; g.ldw = w.taps * w.cin_pad; g.C = out; g.ldc = ldo;
    g.M = B * tq; g.N = w.cout; g.K = w.taps * w.cin_pad; g.bias = w.b;
    g.conv_taps = w.taps; g.conv_cin_pad = w.cin_pad; g.conv_tin = tin; g.conv_tq = tq; g.conv_stride = stride;
    for (int j = 0; j < w.taps; ++j) g.conv_off[j] = offs[j];
    g.conv_off0 = offs[0]; g.conv_dstep = (w.taps > 1) ? offs[1] - offs[0] : 0;
    g.zero_page = h->zero_page; g.orow_mul_m1 = orow_mul - 1; g.orow_add = orow_add;
    g.ncol_split = ncol_split; g.ncol_jump = ncol_jump;
    if (!(fused_res_co > 0 && plan_fused_residual(g, fused_res_co, splitk_buf ? splitk_buf : tl_splitk, defer, defer_res))) {
        plan_split(g, splitk_buf, defer);
        if (defer_res && defer) *defer_res = *defer;
    }
    return launch_gemm(g, EPI_BIAS, s);
}
static int conv5(const rohm_trajnet* h, const ConvW& w, const float* x, int ldx, int B, int T, float* out, int ldo,
                 hipStream_t s, SplitInfo* defer = nullptr) {
    static const int offs[5] = {-2, -1, 0, 1, 2};
    return conv_gemm(h, w, x, ldx, B, T, T, 1, offs, out, ldo, 1, 0, s, defer);
}
static int conv1(const ConvW& w, const float* x, int ldx, int M, float* out, int ldo, hipStream_t s,
                 SplitInfo* defer = nullptr) {
    GemmParams g{};
    g.A = x; g.lda = ldx; g.W = w.w; g.ldw = w.cin_pad; g.C = out; g.ldc = ldo; g.M = M; g.N = w.cout;
    g.K = w.cin_pad; g.bias = w.b;
    plan_split(g, defer ? tl_splitk_res : nullptr, defer);
    return launch_gemm(g, EPI_BIAS, s);
}
static int down(const rohm_trajnet* h, const ConvW& w, const float* x, int ldx, int B, int T, float* out, int ldo,
                hipStream_t s) {   // Conv1d(k3, s2, p1): heads.py:72-78
    static const int offs[3] = {-1, 0, 1};
    return conv_gemm(h, w, x, ldx, B, T, T / 2, 2, offs, out, ldo, 1, 0, s);
}
static int upsample(const rohm_trajnet* h, const UpW& w, const float* x, int ldx, int B, int Tq, float* out, int ldo,
                    hipStream_t s) {   // ConvTranspose1d(k4, s2, p1): heads.py:81-87
    // out[2t] = W1 x[t] + W3 x[t-1],  out[2t+1] = W0 x[t+1] + W2 x[t]: one GEMM over the taps {-1, 0, +1} with the even
    // phase in columns [0, C) and the odd phase in [C, 2C); row m = (b, t) starts at output row 2 m and the odd columns
    // jump to row 2 m + 1 (ldo - C floats further)
    static const int offs[3] = {-1, 0, 1};
    const int Cc = w.even.cout;
    if ((size_t)B * Tq * Cc * 2 <= kFuseMaxWork)
        return conv_gemm(h, w.both, x, ldx, B, Tq, Tq, 1, offs, out, ldo, 2, 0, s, nullptr, nullptr, Cc, ldo - Cc);
    static const int off_even[2] = {0, -1}, off_odd[2] = {1, 0};      // compute-bound batches: the two 2-tap phases
    int rc = conv_gemm(h, w.even, x, ldx, B, Tq, Tq, 1, off_even, out, ldo, 2, 0, s);
    if (rc) return rc;
    return conv_gemm(h, w.odd, x, ldx, B, Tq, Tq, 1, off_odd, out, ldo, 2, 1, s);
}
// Column views of a fused conv result (plain tensor, or un-reduced split-K slabs) for the GroupNorm kernel.
struct GnFused {
    SplitInfo si; const float* plain; int ldplain;
    const float* y(int col) const { return si.S > 1 ? nullptr : plain + col; }
    int ld() const { return ldplain; }
    SplitInfo split(int col) const { SplitInfo o = si; if (o.S > 1) o.base = si.base + col; return o; }
};

static int gn(const BlockW& bw, const float* y, int ldy, const SplitInfo& sy, int B, int T, const float* tb, int ldtb,
              const float* res, int ldres, const SplitInfo& sr, const float* res_bias, const float* add2, int ldadd2,
              float* dst, int lddst, float* dst2, int lddst2, hipStream_t s) {
    GnArgs a{};
    a.y = sy.S > 1 ? sy.base : y; a.ldy = sy.S > 1 ? sy.ldp : ldy; a.ksplit = sy.S; a.slab = sy.slab; a.cbias = bw.conv.b;
    a.gamma = bw.g; a.beta = bw.be; a.C = bw.conv.cout; a.T = T; a.tb = tb; a.ldtb = ldtb;
    a.res = sr.S > 1 ? sr.base : res; a.ldres = sr.S > 1 ? sr.ldp : ldres; a.res_ksplit = sr.S; a.res_slab = sr.slab;
    a.res_bias = res_bias;
    a.add2 = add2; a.ldadd2 = ldadd2; a.dst = dst; a.lddst = lddst; a.dst2 = dst2; a.lddst2 = lddst2;
    if ((bw.conv.cout >> 3) * T > 256 * kGnMaxPerThread || (bw.conv.cout & (bw.conv.cout - 1)) != 0) {
        set_error("trajnet: GroupNorm group of %d x %d values is not supported", bw.conv.cout >> 3, T);
        return ROHM_ERR_UNSUPPORTED;
    }
    auto ok4 = [](const void* p, long ld) { return p == nullptr || ((((uintptr_t)p) & 15) == 0 && ld % 4 == 0); };
    const bool vec = ok4(a.y, a.ldy) && ok4(a.y, (long)(a.slab % 4)) && ok4(a.res, a.ldres) && ok4(a.res, (long)(a.res_slab % 4)) &&
                     ok4(a.add2, a.ldadd2) && ok4(a.dst, a.lddst) && ok4(a.dst2, a.lddst2) && ok4(a.tb, a.ldtb) &&
                     ok4(a.gamma, 0) && ok4(a.beta, 0) && ok4(a.cbias, 0) && ok4(a.res_bias, 0) && (a.C >> 3) % 4 == 0;
    const char* label = "gn_mish";
    if (prof::detail()) {
        char buf[48];
        snprintf(buf, sizeof(buf), "gn_mish C%d T%d S%d R%d", bw.conv.cout, T, sy.S, res ? sr.S : 0);
        label = prof::intern(buf);
    }
    prof::Scope ps(label, 0.0, 4.0 * B * T * bw.conv.cout * (2.0 + sy.S), s);
    if (vec) hipLaunchKernelGGL(gn_mish_kernel<true>, dim3(B, 8), dim3(256), 0, s, a);
    else hipLaunchKernelGGL(gn_mish_kernel<false>, dim3(B, 8), dim3(256), 0, s, a);
    ROHM_LAUNCH_CHECK();
    return ROHM_OK;
}


// ResidualTemporalBlock (heads.py:43-54)
static int res_block(const rohm_trajnet* h, const ResW& r, const float* x, int ldx, int B, int T, const float* tb_all,
                     int ldtb, const float* add2, int ldadd2, float* dst, int lddst, float* dst2, int lddst2,
                     const Scratch& sc, hipStream_t s) {
    int rc;
    const int co = r.cout;
    SplitInfo s0, s1, sr, none;
    const float* tb = (r.tb_off >= 0) ? tb_all + r.tb_off : nullptr;
    // the fused forms trade flops (zero taps) for launches: only while the step is latency-bound (measured: B = 256 loses
    // 10 % with them, B <= 32 gains 8-11 %); T * cout is the same at every level, so this is a bound on the batch
    if (r.has_res && (size_t)B * T * co <= kFuseMaxWork) {
        // block-0 conv and the 1x1 residual conv in one launch: C = [conv5(x) | res(x)], 2 co columns.  Its split-K slabs go
        // to the residual buffer (they must outlive the second conv, which re-uses the other one).
        // The residual columns walk the centre tap only and may be split fewer ways than the conv columns (plan_fused_residual): two views
        SplitInfo sf, sfr;
        static const int offs[5] = {-2, -1, 0, 1, 2};
        if ((rc = conv_gemm(h, r.b0res, x, ldx, B, T, T, 1, offs, sc.ya, 2 * co, 1, 0, s, &sf, tl_splitk_res, 0, 0, co, &sfr))) return rc;
        GnFused f0{sf, sc.ya, 2 * co}, fr{sfr, sc.ya, 2 * co};
        if ((rc = gn(r.b0, f0.y(0), f0.ld(), f0.split(0), B, T, tb, ldtb, nullptr, 0, none, nullptr, nullptr, 0, sc.hb, co,
                     nullptr, 0, s)))
            return rc;
        if ((rc = conv5(h, r.b1.conv, sc.hb, co, B, T, sc.rc, co, s, &s1))) return rc;
        return gn(r.b1, sc.rc, co, s1, B, T, nullptr, 0, fr.y(co), fr.ld(), fr.split(co), r.b0res.b + co, add2, ldadd2, dst,
                  lddst, dst2, lddst2, s);
    }
    if ((rc = conv5(h, r.b0.conv, x, ldx, B, T, sc.ya, co, s, &s0))) return rc;
    const float* res = x;
    int ldres = ldx;
    if (r.has_res) {
        if ((rc = conv1(r.res, x, ldx, B * T, sc.rc, co, s, &sr))) return rc;
        res = sc.rc; ldres = co;
    }
    if ((rc = gn(r.b0, sc.ya, co, s0, B, T, tb, ldtb, nullptr, 0, none, nullptr, nullptr, 0, sc.hb, co, nullptr, 0, s))) return rc;
    if ((rc = conv5(h, r.b1.conv, sc.hb, co, B, T, sc.ya, co, s, &s1))) return rc;
    return gn(r.b1, sc.ya, co, s1, B, T, nullptr, 0, res, ldres, sr, r.res.b, add2, ldadd2, dst, lddst, dst2, lddst2, s);
}


static TWs carve_t(const rohm_trajnet* h, int B, int T, float* base) {
    const int m = h->mid;
    const int ch[4] = {m / 8, m / 4, m / 2, m};
    TWs w;
    size_t off = 0;
    auto take = [&](size_t n) { float* p = base ? base + off : nullptr; off += al(n); return p; };
    const size_t M = (size_t)B * T;
    w.xin = take(M * kPadC); w.cin = take(M * kPadC); w.ctl = take(M * kPadCtl);
    for (int i = 0; i < 4; ++i) {
        const size_t Mi = M >> i;
        w.cat[i] = take(Mi * 2 * ch[i]);
        w.dcat[i] = take(Mi * 2 * ch[i]);
        w.ccat[i] = take(Mi * 2 * ch[i]);
        if (i < 3) w.cdn[i] = take((Mi / 2) * ch[i]);
        w.ddn[i] = take((Mi / 2) * 2 * ch[i]);
        w.kdn[i] = take((Mi / 2) * 2 * ch[i]);
        w.d[i] = take(Mi * (i == 0 ? kPadC : ch[i - 1]));   // d[0]: 32 channels in a 64-wide zero-padded row
        w.ctrl[i] = take(Mi * (i == 0 ? 32 : ch[i - 1]));
    }
    const size_t M16 = M >> 4;
    w.mid_a = take(M16 * m); w.mid_b = take(M16 * m); w.kmid_a = take(M16 * m); w.kmid_b = take(M16 * m);
    w.ctrl_mid = take(M16 * m);
    w.cz = take(M * kPadC);
    w.fin = take(M * kPadC);
    w.tb_all = take((size_t)B * h->tb_total);
    w.tb_steps = take((size_t)kTbSteps * h->tb_total);
    w.sc.ya = take(2 * (M * (m / 8) > (M >> 3) * m ? M * (m / 8) : (M >> 3) * m));     // fused [conv | residual] result
    w.sc.hb = take(M * (m / 8) > (M >> 3) * m ? M * (m / 8) : (M >> 3) * m);
    w.sc.rc = take(M * (m / 8) > (M >> 3) * m ? M * (m / 8) : (M >> 3) * m);
    w.x0 = take(M * h->ctraj);
    w.cond_keep = take(16);
    w.splitk = take(kSplitKFloats);
    w.splitk_res = take(kSplitKFloats);
    for (int i = 0; i < 4; ++i) w.ctrl_b[i] = nullptr;
    w.ctrl_mid_b = nullptr; w.sc_ctl = Scratch{}; w.splitk_ctl = w.splitk_res_ctl = nullptr;
    if (h->control) {
        for (int i = 0; i < 4; ++i) w.ctrl_b[i] = take((M >> i) * (i == 0 ? 32 : ch[i - 1]));
        w.ctrl_mid_b = take(M16 * m);
        const size_t blk = M * (m / 8) > (M >> 3) * m ? M * (m / 8) : (M >> 3) * m;
        w.sc_ctl.ya = take(2 * blk); w.sc_ctl.hb = take(blk); w.sc_ctl.rc = take(blk);
        w.splitk_ctl = take(kSplitKFloats);
        w.splitk_res_ctl = take(kSplitKFloats);
    }
    w.step_coef = take(3 * (size_t)kGraphMaxSteps);
    w.step_t = reinterpret_cast<int64_t*>(take(2 * (size_t)kGraphMaxSteps));
    w.step_ctr = reinterpret_cast<int*>(take(16));
    w.resident = take(resident_floats(B, T));
    w.floats = off;
    return w;
}

static int check_t(const rohm_trajnet* h, int B, int T) {
    ROHM_ARG_CHECK(h != nullptr, "trajnet: null handle");
    ROHM_ARG_CHECK(B > 0, "trajnet: batch must be positive");
    ROHM_ARG_CHECK(T > 0 && T % 16 == 0, "trajnet: T must be a multiple of 16 (got %d)", T);
    return ROHM_OK;
}

// cond encoder (no time input): trajnet.py:192-208.  Writes h_cond[i] into cat[i][:, ch:] and ccat[i][:, ch:].
static int run_cond_encoder(const rohm_trajnet* h, const TWs& w, int B, int T, hipStream_t s) {
    const int m = h->mid;
    const int ch[4] = {m / 8, m / 4, m / 2, m};
    int rc;
    const float* x = w.cin;
    int ldx = kPadC;
    for (int i = 0; i < 4; ++i) {
        const int Ti = T >> i;
        float* dst = w.cat[i] + ch[i];
        float* dst2 = h->control ? w.ccat[i] + ch[i] : nullptr;
        if ((rc = res_block(h, h->cond_enc[i], x, ldx, B, Ti, nullptr, 0, nullptr, 0, dst, 2 * ch[i], dst2, 2 * ch[i],
                            w.sc, s)))
            return rc;
        if (i < 3) {
            if ((rc = down(h, h->cond_down[i], dst, 2 * ch[i], B, Ti, w.cdn[i], ch[i], s))) return rc;
            x = w.cdn[i]; ldx = ch[i];
        }
    }
    return ROHM_OK;
}

// everything that depends on x_t / t (and control_cond): trajnet.py:211-275
// ControlNet.forward, trajnet.py:43-75.  control_zero_conv_0(control_cond) depends on the (loop-invariant) control input only:
static int run_control_pre(const rohm_trajnet* h, const TWs& w, int B, int T, hipStream_t s) {
    return conv1(h->c_zero0, w.ctl, kPadCtl, B * T, w.cz, kPadC, s);   // cols 13..63 stay zero
}
// ... the rest depends on the timestep (not on x_t): residuals -> ctrl[0..3], ctrl_mid; intermediates in ccat / kdn / kmid and `sc`
static int run_control(const rohm_trajnet* h, const TWs& w, int B, int T, int ldtb, const float* tb, float* const* ctrl,
                       float* ctrl_mid, const Scratch& sc, hipStream_t s) {
    const int m = h->mid;
    const int ch[4] = {m / 8, m / 4, m / 2, m};
    int rc;
    const float* x = w.cz;
    int ldx = kPadC;
    for (int i = 0; i < 4; ++i) {
        const int Ti = T >> i;
        if ((rc = res_block(h, h->c_enc[i], x, ldx, B, Ti, tb, ldtb, nullptr, 0, w.ccat[i], 2 * ch[i], nullptr, 0, sc, s)))
            return rc;
        if ((rc = conv1(h->c_zero[i], w.ccat[i], 2 * ch[i], B * Ti, ctrl[i], i == 0 ? 32 : ch[i - 1], s))) return rc;
        if ((rc = down(h, h->c_down[i], w.ccat[i], 2 * ch[i], B, Ti, w.kdn[i], 2 * ch[i], s))) return rc;
        x = w.kdn[i]; ldx = 2 * ch[i];
    }
    const int T16 = T >> 4;
    if ((rc = res_block(h, h->c_mid[0], w.kdn[3], 2 * m, B, T16, tb, ldtb, nullptr, 0, w.kmid_a, m, nullptr, 0, sc, s))) return rc;
    if ((rc = res_block(h, h->c_mid[1], w.kmid_a, m, B, T16, tb, ldtb, nullptr, 0, w.kmid_b, m, nullptr, 0, sc, s))) return rc;
    return conv1(h->c_zero_mid, w.kmid_b, m, B * T16, ctrl_mid, m, s);
}

// The ControlNet residuals of a step when they come from a second stream (sample loop): which set, and the events that order
// the two streams -- `ready` is waited for before the first consumer (the second middle block), `consumed` recorded behind the last.
struct CtrlRef { const float* ctrl[4]; const float* mid; hipEvent_t ready, consumed; };

static int run_denoiser(const rohm_trajnet* h, const TWs& w, int B, int T, int ldtb, float* out, hipStream_t s,
                        const float* tb_row = nullptr, const CtrlRef* cr = nullptr) {
    const int m = h->mid;
    const int ch[4] = {m / 8, m / 4, m / 2, m};
    const float* tb = tb_row ? tb_row : w.tb_all;      // tb_row: this step's row of the table built at loop start
    int rc;
    const float* const* ctrl = cr ? cr->ctrl : w.ctrl;
    const float* ctrl_mid = cr ? cr->mid : w.ctrl_mid;
    if (h->control && !cr) {            // single stream: the ControlNet branch in front of the U-Net
        if ((rc = run_control_pre(h, w, B, T, s))) return rc;
        if ((rc = run_control(h, w, B, T, ldtb, tb, w.ctrl, w.ctrl_mid, w.sc, s))) return rc;
    }
    // U-Net encoder
    const float* x = w.xin;
    int ldx = kPadC;
    for (int i = 0; i < 4; ++i) {
        const int Ti = T >> i;
        // output goes to the left half of cat[i] (input of the down conv) and to the right half of dcat[i] (skip)
        if ((rc = res_block(h, h->diff_enc[i], x, ldx, B, Ti, tb, ldtb, nullptr, 0, w.cat[i], 2 * ch[i],
                            w.dcat[i] + ch[i], 2 * ch[i], w.sc, s)))
            return rc;
        if ((rc = down(h, h->diff_down[i], w.cat[i], 2 * ch[i], B, Ti, w.ddn[i], 2 * ch[i], s))) return rc;
        x = w.ddn[i]; ldx = 2 * ch[i];
    }
    const int T16 = T >> 4;
    if ((rc = res_block(h, h->mid_blk[0], w.ddn[3], 2 * m, B, T16, tb, ldtb, nullptr, 0, w.mid_a, m, nullptr, 0, w.sc, s)))
        return rc;
    if (cr) ROHM_HIP_CHECK(hipStreamWaitEvent(s, cr->ready, 0));
    if ((rc = res_block(h, h->mid_blk[1], w.mid_a, m, B, T16, tb, ldtb, h->control ? ctrl_mid : nullptr, m, w.mid_b, m,
                        nullptr, 0, w.sc, s)))
        return rc;
    // decoder
    x = w.mid_b; ldx = m;
    for (int i = 3; i >= 0; --i) {
        const int Ti = T >> i, Tq = Ti / 2;
        if ((rc = upsample(h, h->up[i], x, ldx, B, Tq, w.dcat[i], 2 * ch[i], s))) return rc;
        const int co = (i == 0) ? 32 : ch[i - 1];
        const int ldd = (i == 0) ? kPadC : co;
        if ((rc = res_block(h, h->dec[i], w.dcat[i], 2 * ch[i], B, Ti, tb, ldtb, h->control ? ctrl[i] : nullptr, co,
                            w.d[i], ldd, nullptr, 0, w.sc, s)))
            return rc;
        x = w.d[i]; ldx = ldd;
    }
    if (cr) ROHM_HIP_CHECK(hipEventRecord(cr->consumed, s));
    // head: Conv1dBlock(32, 32, k5) + Conv1d(32, 13, 1)  (trajnet.py:158-161)
    SplitInfo sf, none;
    if ((rc = conv5(h, h->final_blk.conv, w.d[0], kPadC, B, T, w.sc.ya, 32, s, &sf))) return rc;
    if ((rc = gn(h->final_blk, w.sc.ya, 32, sf, B, T, nullptr, 0, nullptr, 0, none, nullptr, nullptr, 0, w.fin, kPadC, nullptr, 0, s)))
        return rc;
    if (!out) return ROHM_OK;           // sampling loop: the head's 1x1 conv is part of traj_tail_kernel
    return conv1(h->final_conv, w.fin, kPadC, B * T, out, h->ctraj, s);
}

// head conv + DDPM update + padded copy of the new x (one launch, see traj_tail_kernel)
static int run_tail(const rohm_trajnet* h, const TWs& w, float* x, float* x0_out, const float* noise, float c1, float c2,
                    float sigma, const float* noise_base, const float* coef_tab, const int* step_ctr, size_t M,
                    hipStream_t s) {
    TailArgs a{};
    a.fin = w.fin; a.ldfin = kPadC; a.w = h->final_conv.w; a.ldw = h->final_conv.cin_pad; a.cin = h->final_conv.cin;
    a.bias = h->final_conv.b; a.ctraj = h->ctraj; a.x = x; a.xin = w.xin; a.ldxin = kPadC; a.x0_out = x0_out;
    a.noise = noise; a.c1 = c1; a.c2 = c2; a.sigma = sigma; a.noise_base = noise_base; a.coef_tab = coef_tab;
    a.step_ctr = step_ctr; a.M = M;
    if (a.cin > 64 || a.cin % 4 != 0 || a.ctraj > 32) { set_error("trajnet: unsupported head shape"); return ROHM_ERR_UNSUPPORTED; }
    const size_t n = M * h->ctraj;
    size_t blocks = (n + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    prof::Scope ps("traj_tail", 2.0 * n * a.cin, 4.0 * (M * a.cin + 4.0 * n), s);
    hipLaunchKernelGGL(traj_tail_kernel, dim3((unsigned)blocks), dim3(256), 0, s, a);
    ROHM_LAUNCH_CHECK();
    return ROHM_OK;
}

static int run_time_path(const rohm_trajnet* h, const TWs& w, const int64_t* t_dev, int64_t t_host, int B,
                         hipStream_t s, const int64_t* t_tab = nullptr, const int* step_ctr = nullptr) {
    const int rows = t_dev ? B : 1;
    prof::Scope ps("time_path", 0.0, 4.0 * h->tb_total * h->tdim, s);
    hipLaunchKernelGGL(time_path_kernel, dim3(rows), dim3(256), 0, s, t_dev, t_host, t_tab, step_ctr, h->tdim, h->t_w1T, h->t_b1, h->t_w3T,
                       h->t_b3, h->tb_wT, h->tb_b, h->tb_total, w.tb_all);
    ROHM_LAUNCH_CHECK();
    return ROHM_OK;
}

// the time path depends on t only (trajnet.py:120-125, heads.py:35-38): one launch covers a run of loop steps (37 us per step before)
int launch_time_path_steps(const rohm_trajnet* h, const TWs& w, const int64_t* t, int run, hipStream_t s) {
    ROHM_HIP_CHECK(hipMemcpyAsync(w.step_t, t, (size_t)run * sizeof(int64_t), hipMemcpyHostToDevice, s));
    prof::Scope ps("time_path", 0.0, 4.0 * h->tb_total * h->tdim * run, s);
    hipLaunchKernelGGL(time_path_kernel, dim3(run), dim3(256), 0, s, w.step_t, (int64_t)0, (const int64_t*)nullptr,
                       (const int*)nullptr, h->tdim, h->t_w1T, h->t_b1, h->t_w3T, h->t_b3, h->tb_wT, h->tb_b,
                       h->tb_total, w.tb_steps);
    ROHM_LAUNCH_CHECK();
    return ROHM_OK;
}

// 32-channel activations that feed a GEMM live in 64-wide rows (K chunks are 64 wide); their pad columns are
// never written by the producers, so they are cleared once per call.
static int zero_pads(const rohm_trajnet* h, const TWs& w, size_t M, hipStream_t s) {
    ROHM_HIP_CHECK(hipMemsetAsync(w.d[0], 0, M * kPadC * sizeof(float), s));
    ROHM_HIP_CHECK(hipMemsetAsync(w.fin, 0, M * kPadC * sizeof(float), s));
    if (h->control) ROHM_HIP_CHECK(hipMemsetAsync(w.cz, 0, M * kPadC * sizeof(float), s));
    return ROHM_OK;
}

static int pad_rows(const float* src, float* dst, size_t rows, int cin, int cpad, hipStream_t s) {
    const size_t n = rows * cpad;
    hipLaunchKernelGGL(pad_rows_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, src, dst, rows, cin, cpad);
    ROHM_LAUNCH_CHECK();
    return ROHM_OK;
}

// TrajControl sample loop: the ControlNet branch (27 of the step's 86 dependent launches) depends on the timestep and on
// loop-invariant inputs only, never on x_t -- it runs on a second stream, up to one step ahead of the U-Net (two sets of residual
// buffers), so its launch latencies hide behind the U-Net's instead of adding to them.  Same kernels, same split plans:
// bit-identical to the one-stream order (ROHM_TRAJ_CTRL_STREAM=0).
struct SideStream {
    hipStream_t s2 = nullptr;
    hipEvent_t ev_in = nullptr, ev_ctl[2] = {nullptr, nullptr}, ev_free[2] = {nullptr, nullptr};
    int device = -1;
};
static SideStream* side_stream(int device) {
    const char* e = getenv("ROHM_TRAJ_CTRL_STREAM");      // read per loop call (once per 100 steps): tests flip it
    if (e && e[0] == '0') return nullptr;
    static thread_local SideStream ss;
    if (ss.s2 && ss.device == device) return &ss;
    if (ss.s2) return nullptr;                  // a host thread that drives two devices keeps the one-stream order on the second
    if (hipStreamCreateWithFlags(&ss.s2, hipStreamNonBlocking) != hipSuccess) { ss.s2 = nullptr; (void)hipGetLastError(); return nullptr; }
    bool ok = hipEventCreateWithFlags(&ss.ev_in, hipEventDisableTiming) == hipSuccess;
    for (int i = 0; i < 2 && ok; ++i)
        ok = hipEventCreateWithFlags(&ss.ev_ctl[i], hipEventDisableTiming) == hipSuccess &&
             hipEventCreateWithFlags(&ss.ev_free[i], hipEventDisableTiming) == hipSuccess;
    if (!ok) { (void)hipGetLastError(); (void)hipStreamDestroy(ss.s2); ss.s2 = nullptr; return nullptr; }
    ss.device = device;
    return &ss;
}

// Opt-in (ROHM_TRAJNET_GRAPH=1): measured on ROCm 7.2 / MI355X the replayed graph is SLOWER than plain stream launches
// (100-step loop, B = 1: 96 ms vs 82 ms; B = 32: 106 vs 95; TrajControl B = 1: 152 vs 130) -- a graph kernel node costs
// ~9.8 us against ~8.4 us for a stream launch, and the plain loop's host time only tracks the GPU because the queue
// back-pressures.  The loop is bound by the GPU-side latency of ~98 small dependent kernels per step; fewer kernels
// (fusion), not graphs, is what would help.
static bool graph_replay_ok(int n_steps) {
    const char* e = getenv("ROHM_TRAJNET_GRAPH");
    return e && atoi(e) == 1 && !prof::enabled() && n_steps >= 3 && n_steps <= kGraphMaxSteps;
}

// Captured-graph sample loop (see rohm_trajnet_sample_loop).  Stream capture is not allowed on the legacy default
// stream torch hands out by default, so the loop runs on a private non-blocking stream fenced with events against
// the caller's stream.  Returns ROHM_ERR_UNSUPPORTED (nothing enqueued) if capture cannot start.
static int sample_loop_graph(const rohm_trajnet* h, const TWs& w, float* x, const float* noise, const int64_t* t_model,
                             const float* coef, float* x0_last, float* x_in_last, int n_steps, int B, int T, size_t M, size_t n,
                             hipStream_t caller) {
    static thread_local hipStream_t gs = nullptr;
    static thread_local hipEvent_t ev_in = nullptr, ev_out = nullptr;
    if (!gs) {
        if (hipStreamCreateWithFlags(&gs, hipStreamNonBlocking) != hipSuccess) { gs = nullptr; return ROHM_ERR_UNSUPPORTED; }
        if (hipEventCreateWithFlags(&ev_in, hipEventDisableTiming) != hipSuccess ||
            hipEventCreateWithFlags(&ev_out, hipEventDisableTiming) != hipSuccess)
            return ROHM_ERR_UNSUPPORTED;
    }
    ROHM_HIP_CHECK(hipEventRecord(ev_in, caller));
    ROHM_HIP_CHECK(hipStreamWaitEvent(gs, ev_in, 0));
    ROHM_HIP_CHECK(hipMemcpyAsync(w.step_coef, coef, (size_t)n_steps * 3 * sizeof(float), hipMemcpyHostToDevice, gs));
    ROHM_HIP_CHECK(hipMemcpyAsync(w.step_t, t_model, (size_t)n_steps * sizeof(int64_t), hipMemcpyHostToDevice, gs));
    ROHM_HIP_CHECK(hipMemsetAsync(w.step_ctr, 0, sizeof(int), gs));
    auto one_step = [&](hipStream_t s) -> int {
        int rc;
        if ((rc = run_time_path(h, w, nullptr, 0, B, s, w.step_t, w.step_ctr))) return rc;
        if ((rc = run_denoiser(h, w, B, T, 0, nullptr, s))) return rc;
        if ((rc = run_tail(h, w, x, w.x0, nullptr, 0.f, 0.f, 0.f, noise, w.step_coef, w.step_ctr, M, s))) return rc;
        return launch_advance_counter(w.step_ctr, s);
    };
    int rc = pad_rows(x, w.xin, M, h->ctraj, kPadC, gs);     // x_T; afterwards the tail kernel keeps the padded copy current
    if (!rc && x_in_last && n_steps == 1)
        ROHM_HIP_CHECK(hipMemcpyAsync(x_in_last, x, n * sizeof(float), hipMemcpyDeviceToDevice, gs));
    if (!rc) rc = one_step(gs);                // step 0 directly (also sets every kernel's launch attributes)
    hipGraph_t graph = nullptr;
    hipGraphExec_t exec = nullptr;
    bool captured = false;
    if (!rc) {
        if (hipStreamBeginCapture(gs, hipStreamCaptureModeThreadLocal) == hipSuccess) {
            const int rc_cap = one_step(gs);
            const hipError_t e = hipStreamEndCapture(gs, &graph);
            captured = !rc_cap && e == hipSuccess && graph &&
                       hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0) == hipSuccess;
        }
        if (!captured) (void)hipGetLastError();        // plain launches on the private stream instead
        for (int i = 1; i < n_steps && !rc; ++i) {
            if (x_in_last && i == n_steps - 1 &&
                hipMemcpyAsync(x_in_last, x, n * sizeof(float), hipMemcpyDeviceToDevice, gs) != hipSuccess) { rc = ROHM_ERR_HIP; break; }
            if (captured) rc = (hipGraphLaunch(exec, gs) == hipSuccess) ? ROHM_OK : ROHM_ERR_HIP;
            else rc = one_step(gs);
        }
    }
    if (!rc && x0_last)
        ROHM_HIP_CHECK(hipMemcpyAsync(x0_last, w.x0, n * sizeof(float), hipMemcpyDeviceToDevice, gs));
    (void)hipEventRecord(ev_out, gs);
    (void)hipStreamWaitEvent(caller, ev_out, 0);
    (void)hipStreamSynchronize(gs);            // the graph objects must outlive their launches
    if (exec) (void)hipGraphExecDestroy(exec);
    if (graph) (void)hipGraphDestroy(graph);
    if (rc == ROHM_ERR_HIP) set_error("trajnet_sample_loop: hipGraph launch failed");
    return rc;
}

}  // namespace rohm

using namespace rohm;

extern "C" {

// Weights arrive as a flat table in the reference's state_dict order (see rohm_hip.h).
int rohm_trajnet_create(rohm_trajnet_t** out, const rohm_trajnet_weights* wts, int mid_dim, int time_dim, int c_traj,
                        int c_ctrl, int trajcontrol, int device) {
    ROHM_ARG_CHECK(out && wts && wts->tensors, "trajnet_create: null argument");
    ROHM_ARG_CHECK(mid_dim >= 256 && mid_dim % 256 == 0, "trajnet_create: mid_dim must be a multiple of 256");
    ROHM_ARG_CHECK(time_dim == 32, "trajnet_create: time_dim must be 32");
    ROHM_ARG_CHECK(c_traj > 0 && c_traj <= 32 && c_ctrl > 0 && c_ctrl <= kPadCtl, "trajnet_create: bad channel counts");
    ROHM_HIP_CHECK(hipSetDevice(device));
    rohm_trajnet* h = new rohm_trajnet();
    h->mid = mid_dim; h->tdim = time_dim; h->ctraj = c_traj; h->cctrl = c_ctrl; h->control = trajcontrol ? 1 : 0;
    h->device = device;
    const int m = mid_dim;
    const int ch[4] = {m / 8, m / 4, m / 2, m};

    // ---- plan: walk the architecture twice (size, then fill) with the same code ------------------------------
    int cursor = 0;                       // index into wts->tensors
    size_t total = 0;
    float* base = nullptr;
    std::vector<float*> staged;           // device staging copies of source tensors (freed at the end)
    bool ok = true;
    std::string err;
    auto next = [&](size_t expect_elems) -> const float* {
        if (cursor >= wts->n_tensors) { ok = false; err = "weight table too short"; return nullptr; }
        const rohm_tensor_ref& t = wts->tensors[cursor++];
        if (t.numel != expect_elems) {
            ok = false;
            char buf[160];
            snprintf(buf, sizeof(buf), "tensor %d has %zu elements, expected %zu", cursor - 1, (size_t)t.numel, expect_elems);
            err = buf;
            return nullptr;
        }
        return t.data;
    };
    auto arena_take = [&](size_t n) { float* p = base ? base + total : nullptr; total += al(n); return p; };
    auto stage = [&](const float* src, size_t n) -> float* {   // device copy of a (host or device) tensor
        float* d = nullptr;
        if (hipMalloc(&d, n * sizeof(float)) != hipSuccess) { ok = false; err = "hipMalloc(staging) failed"; return nullptr; }
        staged.push_back(d);
        if (hipMemcpy(d, src, n * sizeof(float), hipMemcpyDefault) != hipSuccess) { ok = false; err = "hipMemcpy failed"; }
        return d;
    };
    auto pad32 = [](int c) { return (c + 63) / 64 * 64; };   // GEMM K chunks are 64 wide
    // conv: weight [cout, cin, k] (or [cin, cout, k] when transposed) + bias [cout]
    auto load_conv = [&](ConvW& c, int cout, int cin, int k, int tap0, int tap_step, int ntap, bool transposed,
                         const float* wsrc, const float* bsrc) {
        c.cout = cout; c.cin = cin; c.cin_pad = pad32(cin); c.taps = (k == 1) ? 0 : ntap;
        const int K = ntap * c.cin_pad;
        c.w = arena_take((size_t)cout * K);
        c.b = arena_take(cout);
        if (base && ok) {
            float* dw = stage(wsrc, (size_t)cout * cin * k);
            if (!ok) return;
            const int n = cout * K;
            hipLaunchKernelGGL(relayout_conv_kernel, dim3((n + 255) / 256), dim3(256), 0, 0, dw, c.w, cout, cin, k,
                               c.cin_pad, tap0, tap_step, ntap, transposed ? 1 : 0);
            if (hipMemcpy(c.b, bsrc, cout * sizeof(float), hipMemcpyDefault) != hipSuccess) { ok = false; err = "bias copy"; }
        }
    };
    auto conv_std = [&](ConvW& c, int cout, int cin, int k) {
        const float* wsrc = next((size_t)cout * cin * k);
        const float* bsrc = next(cout);
        if (!ok) return;
        load_conv(c, cout, cin, k, 0, 1, k, false, wsrc, bsrc);
    };
    auto vec = [&](float*& dst, int n) {
        const float* src = next(n);
        dst = arena_take(n);
        if (base && ok && hipMemcpy(dst, src, n * sizeof(float), hipMemcpyDefault) != hipSuccess) { ok = false; err = "vector copy"; }
    };
    std::vector<std::pair<const float*, const float*>> tb_src;   // (weight [cout, tdim], bias) per time block
    std::vector<int> tb_cout;
    int tb_total = 0;
    auto block = [&](BlockW& b, int cin, int cout) {
        conv_std(b.conv, cout, cin, 5);
        vec(b.g, cout);
        vec(b.be, cout);
    };
    auto resblk = [&](ResW& r, int cin, int cout, bool input_t) {
        r.cin = cin; r.cout = cout;
        block(r.b0, cin, cout);
        block(r.b1, cout, cout);
        if (input_t) {
            const float* tw = next((size_t)cout * time_dim);
            const float* tbias = next(cout);
            r.tb_off = tb_total;
            tb_total += cout;
            tb_src.push_back({tw, tbias});
            tb_cout.push_back(cout);
        }
        r.has_res = (cin != cout);
        if (r.has_res) {
            conv_std(r.res, cout, cin, 1);
            // the residual conv reads the same input as the block's first conv: one GEMM with 2 cout output columns, the
            // 1x1 weights sitting at the centre tap (the arena is zero-filled)
            ConvW& f = r.b0res;
            const ConvW& a0 = r.b0.conv;
            f.cout = 2 * cout; f.cin = cin; f.cin_pad = a0.cin_pad; f.taps = 5;
            const size_t K5 = (size_t)5 * a0.cin_pad;
            f.w = arena_take((size_t)2 * cout * K5);
            f.b = arena_take(2 * cout);
            if (base && ok) {
                if (hipDeviceSynchronize() != hipSuccess ||
                    hipMemcpy(f.w, a0.w, (size_t)cout * K5 * sizeof(float), hipMemcpyDeviceToDevice) != hipSuccess ||
                    hipMemcpy2D(f.w + (size_t)cout * K5 + 2 * a0.cin_pad, K5 * sizeof(float), r.res.w,
                                (size_t)r.res.cin_pad * sizeof(float), (size_t)r.res.cin_pad * sizeof(float), cout,
                                hipMemcpyDeviceToDevice) != hipSuccess ||
                    hipMemcpy(f.b, a0.b, cout * sizeof(float), hipMemcpyDeviceToDevice) != hipSuccess ||
                    hipMemcpy(f.b + cout, r.res.b, cout * sizeof(float), hipMemcpyDeviceToDevice) != hipSuccess) {
                    ok = false; err = "fused residual weights";
                }
            }
        }
    };
    const float *w1 = nullptr, *b1 = nullptr, *w3 = nullptr, *b3 = nullptr;
    const float* upw[4] = {nullptr, nullptr, nullptr, nullptr};
    const float* upb[4] = {nullptr, nullptr, nullptr, nullptr};

    for (int pass = 0; pass < 2 && ok; ++pass) {
        cursor = 0; total = 0; tb_total = 0; tb_src.clear(); tb_cout.clear();
        // --- state_dict order of model/trajnet.py (controlnet.* first when present) ---
        if (h->control) {
            conv_std(h->c_zero0, c_traj, c_ctrl, 1);
            int cin = c_traj;
            for (int i = 0; i < 4 && ok; ++i) {
                resblk(h->c_enc[i], cin, ch[i], true);
                conv_std(h->c_zero[i], i == 0 ? 32 : ch[i - 1], ch[i], 1);
                conv_std(h->c_down[i], 2 * ch[i], 2 * ch[i], 3);
                cin = 2 * ch[i];
            }
            resblk(h->c_mid[0], 2 * m, m, true);
            resblk(h->c_mid[1], m, m, true);
            conv_std(h->c_zero_mid, m, m, 1);
        }
        w1 = next((size_t)4 * time_dim * time_dim); b1 = next(4 * time_dim);
        w3 = next((size_t)time_dim * 4 * time_dim); b3 = next(time_dim);
        {
            int cin = c_traj;
            for (int i = 0; i < 4 && ok; ++i) {
                resblk(h->diff_enc[i], cin, ch[i], true);
                conv_std(h->diff_down[i], 2 * ch[i], 2 * ch[i], 3);
                cin = 2 * ch[i];
            }
        }
        resblk(h->mid_blk[0], 2 * m, m, true);
        resblk(h->mid_blk[1], m, m, true);
        for (int i = 3; i >= 0 && ok; --i) {          // upsample4, dec4, ..., upsample1, dec1
            upw[i] = next((size_t)ch[i] * ch[i] * 4);
            upb[i] = next(ch[i]);
            if (ok) {
                load_conv(h->up[i].even, ch[i], ch[i], 4, 1, 2, 2, true, upw[i], upb[i]);   // taps j = 1, 3
                load_conv(h->up[i].odd, ch[i], ch[i], 4, 0, 2, 2, true, upw[i], upb[i]);    // taps j = 0, 2
            }
            {   // even = [tap @0 | tap @-1], odd = [tap @+1 | tap @0]  ->  both = [@-1 | @0 | @+1] x {even rows, odd rows}
                UpW& u = h->up[i];
                const int Cc = ch[i], cp = u.even.cin_pad;
                u.both.cout = 2 * Cc; u.both.cin = Cc; u.both.cin_pad = cp; u.both.taps = 3;
                u.both.w = arena_take((size_t)2 * Cc * 3 * cp);
                u.both.b = arena_take(2 * Cc);
                if (base && ok) {
                    const size_t sp = (size_t)2 * cp * sizeof(float), dp = (size_t)3 * cp * sizeof(float), wd = (size_t)cp * sizeof(float);
                    float* de = u.both.w;
                    float* dod = u.both.w + (size_t)Cc * 3 * cp;
                    if (hipDeviceSynchronize() != hipSuccess ||
                        hipMemcpy2D(de, dp, u.even.w + cp, sp, wd, Cc, hipMemcpyDeviceToDevice) != hipSuccess ||          // @-1
                        hipMemcpy2D(de + cp, dp, u.even.w, sp, wd, Cc, hipMemcpyDeviceToDevice) != hipSuccess ||          // @0
                        hipMemcpy2D(dod + cp, dp, u.odd.w + cp, sp, wd, Cc, hipMemcpyDeviceToDevice) != hipSuccess ||     // @0
                        hipMemcpy2D(dod + 2 * cp, dp, u.odd.w, sp, wd, Cc, hipMemcpyDeviceToDevice) != hipSuccess ||      // @+1
                        hipMemcpy(u.both.b, u.even.b, Cc * sizeof(float), hipMemcpyDeviceToDevice) != hipSuccess ||
                        hipMemcpy(u.both.b + Cc, u.odd.b, Cc * sizeof(float), hipMemcpyDeviceToDevice) != hipSuccess) {
                        ok = false; err = "fused upsample weights";
                    }
                }
            }
            resblk(h->dec[i], 2 * ch[i], i == 0 ? 32 : ch[i - 1], true);
        }
        block(h->final_blk, 32, 32);
        conv_std(h->final_conv, c_traj, 32, 1);
        {
            int cin = c_traj;   // cond_dim == traj_feat_dim in every driver (test_amass_full.py:151-153)
            for (int i = 0; i < 4 && ok; ++i) {
                resblk(h->cond_enc[i], cin, ch[i], false);
                ConvW dummy;
                conv_std(i < 3 ? h->cond_down[i] : dummy, ch[i], ch[i], 3);   // cond_downsample4 exists but is never called
                cin = ch[i];
            }
        }
        if (ok && cursor != wts->n_tensors) { ok = false; err = "weight table has extra tensors"; }
        // time path storage
        h->tb_total = tb_total;
        h->t_w1T = arena_take((size_t)time_dim * 4 * time_dim); h->t_b1 = arena_take(4 * time_dim);
        h->t_w3T = arena_take((size_t)4 * time_dim * time_dim); h->t_b3 = arena_take(time_dim);
        h->tb_wT = arena_take((size_t)time_dim * tb_total); h->tb_b = arena_take(tb_total);
        h->zero_page = arena_take(64);
        if (pass == 0 && ok) {
            const size_t bytes = total * sizeof(float);
            hipError_t e = hipMalloc(&h->arena, bytes);
            if (e != hipSuccess) { ok = false; err = std::string("hipMalloc(arena) failed: ") + hipGetErrorString(e); break; }
            if (hipMemset(h->arena, 0, bytes) != hipSuccess) { ok = false; err = "hipMemset failed"; break; }
            base = h->arena;
        } else if (ok) {
            // fill the time path
            float* d1 = stage(w1, (size_t)4 * time_dim * time_dim);
            float* d3 = stage(w3, (size_t)time_dim * 4 * time_dim);
            if (ok) {
                const int n1 = 4 * time_dim * time_dim;
                hipLaunchKernelGGL(transpose2_kernel, dim3((n1 + 255) / 256), dim3(256), 0, 0, d1, h->t_w1T, 4 * time_dim,
                                   time_dim, 4 * time_dim, 0);
                hipLaunchKernelGGL(transpose2_kernel, dim3((n1 + 255) / 256), dim3(256), 0, 0, d3, h->t_w3T, time_dim,
                                   4 * time_dim, time_dim, 0);
                (void)hipMemcpy(h->t_b1, b1, 4 * time_dim * sizeof(float), hipMemcpyDefault);
                (void)hipMemcpy(h->t_b3, b3, time_dim * sizeof(float), hipMemcpyDefault);
                int col = 0;
                for (size_t k = 0; k < tb_src.size() && ok; ++k) {
                    const int co = tb_cout[k];
                    float* dw = stage(tb_src[k].first, (size_t)co * time_dim);
                    if (!ok) break;
                    const int n = co * time_dim;
                    hipLaunchKernelGGL(transpose2_kernel, dim3((n + 255) / 256), dim3(256), 0, 0, dw, h->tb_wT, co, time_dim,
                                       tb_total, col);
                    (void)hipMemcpy(h->tb_b + col, tb_src[k].second, co * sizeof(float), hipMemcpyDefault);
                    col += co;
                }
            }
        }
    }
    hipError_t se = hipDeviceSynchronize();
    for (float* d : staged) (void)hipFree(d);
    // the clip-resident sample loop meets through L2 inside a launch: is this device laid out the way it assumes?  (probe launch, once
    // per device and process: exchange.hip; a refusal only means the launch-per-layer loop is used)
    { const char* why = nullptr; (void)exchange_layout_ok(device, &why); }
    if (ok && se != hipSuccess) { ok = false; err = hipGetErrorString(se); }
    if (!ok) {
        set_error("trajnet_create: %s", err.c_str());
        if (h->arena) (void)hipFree(h->arena);
        delete h;
        return ROHM_ERR_ARG;
    }
    *out = h;
    return ROHM_OK;
}

int rohm_trajnet_loop_mode(void) { return tl_loop_mode; }

int rohm_trajnet_tune(int conv_wg_per_cu, int split_min_chunks, int split_pow2) {
    ROHM_ARG_CHECK(conv_wg_per_cu >= 1 && conv_wg_per_cu <= 2 && split_min_chunks >= 1 && split_min_chunks <= 64,
                   "trajnet_tune: conv_wg_per_cu must be 1 or 2, split_min_chunks 1..64");
    g_conv_wg_per_cu.store(conv_wg_per_cu, std::memory_order_relaxed);
    g_split_min_chunks.store(split_min_chunks, std::memory_order_relaxed);
    g_split_pow2.store(split_pow2 ? 1 : 0, std::memory_order_relaxed);
    return ROHM_OK;
}

void rohm_trajnet_destroy(rohm_trajnet_t* h) {
    if (!h) return;
    if (h->arena) (void)hipFree(h->arena);
    delete h;
}

size_t rohm_trajnet_workspace_bytes(const rohm_trajnet_t* h, int B, int T) {
    if (!h || B <= 0 || T <= 0 || T % 16) return 0;
    return carve_t(h, B, T, nullptr).floats * sizeof(float);
}

int rohm_trajnet_forward(const rohm_trajnet_t* h, const float* x_t, const float* cond, const float* control_cond,
                         const int64_t* t, float* x0_out, int B, int T, void* ws, size_t ws_bytes,
                         rohm_stream_t stream) {
    int rc = check_t(h, B, T);
    if (rc) return rc;
    ROHM_ARG_CHECK(x_t && cond && t && x0_out && ws, "trajnet_forward: null argument");
    ROHM_ARG_CHECK(!h->control || control_cond, "trajnet_forward: TrajControl needs control_cond");
    ROHM_ARG_CHECK(((uintptr_t)ws % 256) == 0, "trajnet_forward: workspace must be 256-byte aligned");
    hipStream_t s = (hipStream_t)stream;
    TWs w = carve_t(h, B, T, (float*)ws);
    if (w.floats * sizeof(float) > ws_bytes) {
        set_error("trajnet_forward: workspace too small (%zu < %zu)", ws_bytes, w.floats * sizeof(float));
        return ROHM_ERR_WORKSPACE;
    }
    tl_splitk = w.splitk;
    tl_splitk_res = w.splitk_res;
    { const char* e = getenv("ROHM_TRAJ_RES_TAP"); g_res_centre_tap.store(!(e && e[0] == '0'), std::memory_order_relaxed); }
    const size_t M = (size_t)B * T;
    if ((rc = pad_rows(x_t, w.xin, M, h->ctraj, kPadC, s))) return rc;
    if ((rc = pad_rows(cond, w.cin, M, h->ctraj, kPadC, s))) return rc;
    if ((rc = zero_pads(h, w, M, s))) return rc;
    if (h->control) {
        if ((rc = pad_rows(control_cond, w.ctl, M, h->cctrl, kPadCtl, s))) return rc;
    }
    if ((rc = run_time_path(h, w, t, 0, B, s))) return rc;
    if ((rc = run_cond_encoder(h, w, B, T, s))) return rc;
    return run_denoiser(h, w, B, T, h->tb_total, x0_out, s);
}

int rohm_trajnet_sample_loop(const rohm_trajnet_t* h, float* x, const float* cond, const float* control_cond,
                             const int64_t* t_model, const float* coef, const float* noise, float* x0_last,
                             float* x_in_last, int n_steps, int B, int T, void* ws, size_t ws_bytes, rohm_stream_t stream) {
    int rc = check_t(h, B, T);
    if (rc) return rc;
    ROHM_ARG_CHECK(x && cond && t_model && coef && ws, "trajnet_sample_loop: null argument");
    ROHM_ARG_CHECK(!h->control || control_cond, "trajnet_sample_loop: TrajControl needs control_cond");
    ROHM_ARG_CHECK(((uintptr_t)ws % 256) == 0, "trajnet_sample_loop: workspace must be 256-byte aligned");
    hipStream_t s = (hipStream_t)stream;
    // the side stream and its events belong to the handle's device: make it current for the duration of the call
    struct DeviceGuard {
        int prev = -1;
        explicit DeviceGuard(int want) { if (hipGetDevice(&prev) != hipSuccess || prev == want || hipSetDevice(want) != hipSuccess) prev = -1; }
        ~DeviceGuard() { if (prev >= 0) (void)hipSetDevice(prev); }
    } device_guard(h->device);
    TWs w = carve_t(h, B, T, (float*)ws);
    if (w.floats * sizeof(float) > ws_bytes) {
        set_error("trajnet_sample_loop: workspace too small (%zu < %zu)", ws_bytes, w.floats * sizeof(float));
        return ROHM_ERR_WORKSPACE;
    }
    tl_splitk = w.splitk;
    tl_splitk_res = w.splitk_res;
    tl_loop_mode = 0;
    { const char* e = getenv("ROHM_TRAJ_RES_TAP"); g_res_centre_tap.store(!(e && e[0] == '0'), std::memory_order_relaxed); }
    const size_t M = (size_t)B * T, n = M * h->ctraj;
    // cond / control_cond do not change over the loop: pad them and run the (time-free) cond encoder once
    if ((rc = pad_rows(cond, w.cin, M, h->ctraj, kPadC, s))) return rc;
    if ((rc = zero_pads(h, w, M, s))) return rc;
    if (h->control) {
        if ((rc = pad_rows(control_cond, w.ctl, M, h->cctrl, kPadCtl, s))) return rc;
    }
    if ((rc = run_cond_encoder(h, w, B, T, s))) return rc;
    for (int i = 0; i < n_steps; ++i)
        ROHM_ARG_CHECK(coef[3 * i + 2] == 0.f || noise, "trajnet_sample_loop: noise is required when sigma != 0");
    if (graph_replay_ok(n_steps)) {
        // hipGraph replay (opt-in, see graph_replay_ok): one denoising step is captured into a hipGraph whose kernels read
        // the per-step values (timestep, c1 / c2 / sigma, noise slice) from device tables through a device-side step
        // counter, and the graph is replayed for the remaining steps.
        rc = sample_loop_graph(h, w, x, noise, t_model, coef, x0_last, x_in_last, n_steps, B, T, M, n, s);
        if (rc == ROHM_OK) tl_loop_mode = 2;
        if (rc != ROHM_ERR_UNSUPPORTED) return rc;
        // capture not available on this stream / runtime: fall through to the plain loop
    }
    if ((rc = pad_rows(x, w.xin, M, h->ctraj, kPadC, s))) return rc;     // x_T; the tail kernel keeps the padded copy current
    bool control_pre_done = false;
    if (resident_ok(h, B, T, n_steps, s)) {
        // clip-resident form (trajnet_resident.hip): ONE launch per step, the XCD's workgroups stay with its clips through the whole
        // U-Net (+ ControlNet branch).  A wait that expired (another tenant, a CU mask set after the probe) hands the call back: x is x_T again.
        if (h->control) { if ((rc = run_control_pre(h, w, B, T, s))) return rc; control_pre_done = true; }
        rc = resident_loop(h, w, x, noise, t_model, coef, x0_last, x_in_last, n_steps, B, T, s);
        if (rc == ROHM_OK) tl_loop_mode = 1;
        if (rc != ROHM_ERR_UNSUPPORTED) return rc;
        if ((rc = pad_rows(x, w.xin, M, h->ctraj, kPadC, s))) return rc;
    }
    SideStream* ss = h->control ? side_stream(h->device) : nullptr;
    if (ss && !control_pre_done && (rc = run_control_pre(h, w, B, T, s))) return rc;           // control_zero_conv_0(control_cond): once per loop
    // an early return must not leave side-stream work behind that still reads / writes the caller-owned workspace un-ordered
    // against the caller's stream: join s2 into s first
    auto fail = [&](int code) {
        if (ss && hipEventRecord(ss->ev_in, ss->s2) == hipSuccess) (void)hipStreamWaitEvent(s, ss->ev_in, 0);
        tl_splitk = w.splitk; tl_splitk_res = w.splitk_res;
        return code;
    };
    for (int i = 0; i < n_steps; ++i) {
        prof::set_step(i);
        const float c1 = coef[3 * i], c2 = coef[3 * i + 1], sigma = coef[3 * i + 2];
        if (x_in_last && i == n_steps - 1)       // the reference keeps the input of the last step in batch['x_t']
            ROHM_HIP_CHECK(hipMemcpyAsync(x_in_last, x, n * sizeof(float), hipMemcpyDeviceToDevice, s));
        if (i % kTbSteps == 0) {
            // the time path depends on t only (trajnet.py:120-125, heads.py:35-38): one launch covers the next run of
            // steps (37 us per step before)
            const int run = (n_steps - i < kTbSteps) ? n_steps - i : kTbSteps;
            if ((rc = launch_time_path_steps(h, w, t_model + i, run, s))) return rc;
        }
        const float* tb_row = w.tb_steps + (size_t)(i % kTbSteps) * h->tb_total;
        if (ss) {
            const int b = i & 1;
            if (i % kTbSteps == 0) {          // the branch reads what this stream has just produced: the time table (and, at i = 0,
                ROHM_HIP_CHECK(hipEventRecord(ss->ev_in, s));                 // the cond encodings and the projected control input)
                ROHM_HIP_CHECK(hipStreamWaitEvent(ss->s2, ss->ev_in, 0));
            }
            if (i >= 2) ROHM_HIP_CHECK(hipStreamWaitEvent(ss->s2, ss->ev_free[b], 0));      // step i - 2 has consumed this set
            tl_splitk = w.splitk_ctl; tl_splitk_res = w.splitk_res_ctl;
            rc = run_control(h, w, B, T, 0, tb_row, b ? w.ctrl_b : w.ctrl, b ? w.ctrl_mid_b : w.ctrl_mid, w.sc_ctl, ss->s2);
            tl_splitk = w.splitk; tl_splitk_res = w.splitk_res;
            if (rc) return fail(rc);
            if (hipEventRecord(ss->ev_ctl[b], ss->s2) != hipSuccess) { set_error("trajnet_sample_loop: hipEventRecord failed"); return fail(ROHM_ERR_HIP); }
            CtrlRef cr{};
            for (int j = 0; j < 4; ++j) cr.ctrl[j] = b ? w.ctrl_b[j] : w.ctrl[j];
            cr.mid = b ? w.ctrl_mid_b : w.ctrl_mid;
            cr.ready = ss->ev_ctl[b]; cr.consumed = ss->ev_free[b];
            if ((rc = run_denoiser(h, w, B, T, 0, nullptr, s, tb_row, &cr))) return fail(rc);
        } else if ((rc = run_denoiser(h, w, B, T, 0, nullptr, s, tb_row))) {
            return rc;
        }
        // head conv + ancestral update + padded copy for the next step: one launch (three before)
        if ((rc = run_tail(h, w, x, (x0_last && i == n_steps - 1) ? x0_last : nullptr, noise ? noise + (size_t)i * n : nullptr,
                           c1, c2, sigma, nullptr, nullptr, nullptr, M, s)))
            return fail(rc);
    }
    return ROHM_OK;
}

}  // extern "C"
