// Bandwidth-bound helpers of the PoseNet step: LayerNorm, layout pack, timestep token, DDPM update.
// All are vectorised to 16 B per lane and sized so that >> 256 workgroups are in flight.
#include "common.h"
#include "planes.h"

namespace rohm {

// ------------------------------------------------------------------------------ LayerNorm
// One wave per row, whole row in registers (two-pass mean / variance like torch's CPU kernel):
// nn.LayerNorm(eps=1e-5) inside nn.TransformerEncoderLayer, post-norm (model/posenet.py:63-69).
template <int D>
__global__ __launch_bounds__(256) void layernorm_kernel(float* __restrict__ x, const float* __restrict__ g,
                                                        const float* __restrict__ b, int M) {
    constexpr int V = D / 256;   // float4 per lane
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= M) return;
    float* xr = x + (size_t)row * D;
    f32x4 v[V];
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < V; ++i) {
        v[i] = *reinterpret_cast<const f32x4*>(xr + (i * 64 + lane) * 4);
        s += (v[i][0] + v[i][1]) + (v[i][2] + v[i][3]);
    }
    s = wave_sum64(s);
    const float mean = s * (1.0f / D);
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < V; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const float d = v[i][j] - mean;
            q += d * d;
        }
    q = wave_sum64(q);
    const float rstd = 1.0f / sqrtf(q * (1.0f / D) + 1e-5f);
#pragma unroll
    for (int i = 0; i < V; ++i) {
        const f32x4 gg = *reinterpret_cast<const f32x4*>(g + (i * 64 + lane) * 4);
        const f32x4 bb = *reinterpret_cast<const f32x4*>(b + (i * 64 + lane) * 4);
        f32x4 o;
#pragma unroll
        for (int j = 0; j < 4; ++j) o[j] = (v[i][j] - mean) * rstd * gg[j] + bb[j];
        *reinterpret_cast<f32x4*>(xr + (i * 64 + lane) * 4) = o;
    }
}

int launch_layernorm(float* x, const float* g, const float* b, int M, int D, hipStream_t s) {
    ROHM_ARG_CHECK(M > 0, "layernorm: empty");
    const dim3 grid((M + 3) / 4), block(256);
    prof::Scope ps("layernorm", 0.0, 8.0 * M * D, s);
    if (D == 512) hipLaunchKernelGGL(layernorm_kernel<512>, grid, block, 0, s, x, g, b, M);
    else if (D == 256) hipLaunchKernelGGL(layernorm_kernel<256>, grid, block, 0, s, x, g, b, M);
    else if (D == 1024) hipLaunchKernelGGL(layernorm_kernel<1024>, grid, block, 0, s, x, g, b, M);
    else { set_error("layernorm: unsupported D=%d", D); return ROHM_ERR_UNSUPPORTED; }
    ROHM_LAUNCH_CHECK();
    return ROHM_OK;
}

// LayerNorm that also hands the consuming GEMM the bf16 planes of its result (precision ladder, planes.h).  A workgroup is
// one 16-row block: wave w normalises rows 4 w .. 4 w + 3 exactly as layernorm_kernel does (same arithmetic, same order ->
// the fp32 result is bit-identical), writes them back and parks them in LDS; then every thread cuts 16-byte units
// (8 consecutive k of one row) in fragment order, so each plane store instruction writes 1 KiB contiguous.
template <int D, int NP>
__global__ __launch_bounds__(256) void layernorm_planes_kernel(float* __restrict__ x, const float* __restrict__ g,
                                                               const float* __restrict__ b, char* __restrict__ planes) {
    constexpr int V = D / 256;        // float4 per lane
    constexpr int LD = D + 4;         // LDS row stride (floats)
    __shared__ __attribute__((aligned(16))) float tile[16 * LD];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    f32x4 gg[V], bb[V];
#pragma unroll
    for (int i = 0; i < V; ++i) {
        gg[i] = *reinterpret_cast<const f32x4*>(g + (i * 64 + lane) * 4);
        bb[i] = *reinterpret_cast<const f32x4*>(b + (i * 64 + lane) * 4);
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int lr = wave * 4 + r;
        float* xr = x + ((size_t)blockIdx.x * 16 + lr) * D;
        f32x4 v[V];
        float s = 0.f;
#pragma unroll
        for (int i = 0; i < V; ++i) {
            v[i] = *reinterpret_cast<const f32x4*>(xr + (i * 64 + lane) * 4);
            s += (v[i][0] + v[i][1]) + (v[i][2] + v[i][3]);
        }
        s = wave_sum64(s);
        const float mean = s * (1.0f / D);
        float q = 0.f;
#pragma unroll
        for (int i = 0; i < V; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float d = v[i][j] - mean;
                q += d * d;
            }
        q = wave_sum64(q);
        const float rstd = 1.0f / sqrtf(q * (1.0f / D) + 1e-5f);
#pragma unroll
        for (int i = 0; i < V; ++i) {
            f32x4 o;
#pragma unroll
            for (int j = 0; j < 4; ++j) o[j] = (v[i][j] - mean) * rstd * gg[i][j] + bb[i][j];
            *reinterpret_cast<f32x4*>(xr + (i * 64 + lane) * 4) = o;
            *reinterpret_cast<f32x4*>(tile + lr * LD + (i * 64 + lane) * 4) = o;
        }
    }
    __syncthreads();
    constexpr int UNITS = 16 * (D / 8);
#pragma unroll
    for (int u = threadIdx.x; u < UNITS; u += 256) {
        const int i = u & 15, kg = u >> 4;
        const float* src = tile + i * LD + kg * 8;
        plane_store8<NP>(planes, blockIdx.x * 16 + i, kg, D / 32, *reinterpret_cast<const f32x4*>(src),
                         *reinterpret_cast<const f32x4*>(src + 4));
    }
}

int launch_layernorm_planes(float* x, const float* g, const float* b, int M, int D, int nplane, void* planes, hipStream_t s) {
    ROHM_ARG_CHECK(M > 0 && M % 16 == 0 && planes, "layernorm_planes: M must be a positive multiple of 16");
    ROHM_ARG_CHECK(mode_ok(nplane), "layernorm_planes: mode must be 3, 2 or 16");
    const dim3 grid(M / 16), block(256);
    prof::Scope ps("layernorm", 0.0, (8.0 + 2.0 * mode_planes(nplane)) * M * D, s);
#define ROHM_LNP(DD)                                                                                                     \
    do {                                                                                                                 \
        if (nplane == 3) hipLaunchKernelGGL((layernorm_planes_kernel<DD, 3>), grid, block, 0, s, x, g, b, (char*)planes); \
        else if (nplane == 2) hipLaunchKernelGGL((layernorm_planes_kernel<DD, 2>), grid, block, 0, s, x, g, b, (char*)planes); \
        else hipLaunchKernelGGL((layernorm_planes_kernel<DD, kModeF16>), grid, block, 0, s, x, g, b, (char*)planes);      \
    } while (0)
    if (D == 512) ROHM_LNP(512);
    else if (D == 256) ROHM_LNP(256);
    else if (D == 1024) ROHM_LNP(1024);
    else { set_error("layernorm_planes: unsupported D=%d", D); return ROHM_ERR_UNSUPPORTED; }
#undef ROHM_LNP
    ROHM_LAUNCH_CHECK();
    return ROHM_OK;
}

// ------------------------------------------------------------------------------ DDPM update
// x_prev = c1*x0 + c2*x_t + gscale*grad + sigma*noise
// (q_posterior_mean_variance + p_sample, diffusion/gaussian_diffusion_posenet.py:212-234,426-434).
__global__ __launch_bounds__(256) void ddpm_step_kernel(const float* x_t, const float* __restrict__ x0,
                                                        const float* __restrict__ noise,
                                                        const float* __restrict__ grad, float c1, float c2,
                                                        float sigma, float gscale, float* out,
                                                        size_t n) {
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        float v = c1 * x0[i] + c2 * x_t[i];
        if (grad) v += gscale * grad[i];
        if (noise) v += sigma * noise[i];
        out[i] = v;
    }
}

int launch_ddpm_step(const float* x_t, const float* x0, const float* noise, const float* grad, float c1,
                     float c2, float sigma, float gscale, float* out, size_t n, hipStream_t s) {
    if (n == 0) return ROHM_OK;
    if (sigma == 0.f) noise = nullptr;
    size_t blocks = (n + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(ddpm_step_kernel, dim3((unsigned)blocks), dim3(256), 0, s, x_t, x0, noise, grad, c1, c2,
                       sigma, gscale, out, n);
    ROHM_LAUNCH_CHECK();
    return ROHM_OK;
}


// The same update with its per-step values read on the device: coef_tab[step] = (c1, c2, sigma), noise of step k at
// noise_base + k n, step = *step_ctr.  Lets one captured hipGraph of a denoising step be replayed for every step.
__global__ __launch_bounds__(256) void ddpm_step_indexed_kernel(const float* x_t, const float* __restrict__ x0,
                                                                const float* __restrict__ noise_base,
                                                                const float* __restrict__ coef_tab,
                                                                const int* __restrict__ step_ctr, float* out, size_t n) {
    const int k = *step_ctr;
    const float c1 = coef_tab[3 * k], c2 = coef_tab[3 * k + 1], sigma = coef_tab[3 * k + 2];
    const float* noise = (sigma != 0.f && noise_base) ? noise_base + (size_t)k * n : nullptr;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        float v = c1 * x0[i] + c2 * x_t[i];
        if (noise) v += sigma * noise[i];
        out[i] = v;
    }
}
__global__ void advance_counter_kernel(int* ctr) { *ctr += 1; }

int launch_ddpm_step_indexed(const float* x_t, const float* x0, const float* noise_base, const float* coef_tab,
                             const int* step_ctr, float* out, size_t n, hipStream_t s) {
    if (n == 0) return ROHM_OK;
    size_t blocks = (n + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(ddpm_step_indexed_kernel, dim3((unsigned)blocks), dim3(256), 0, s, x_t, x0, noise_base, coef_tab,
                       step_ctr, out, n);
    ROHM_LAUNCH_CHECK();
    return ROHM_OK;
}
int launch_advance_counter(int* step_ctr, hipStream_t s) {
    hipLaunchKernelGGL(advance_counter_kernel, dim3(1), dim3(1), 0, s, step_ctr);
    ROHM_LAUNCH_CHECK();
    return ROHM_OK;
}

// Per-sample timesteps, schedule tables on the device (p_sample_with_grad without a host sync).
__global__ __launch_bounds__(256) void ddpm_step_table_kernel(const float* x_t, const float* __restrict__ x0,
                                                              const float* __restrict__ noise,
                                                              const float* __restrict__ ga, float wa,
                                                              const float* __restrict__ gb, float wb,
                                                              const float* __restrict__ tables,
                                                              const int64_t* __restrict__ t, int n_steps,
                                                              float* out, size_t row_len) {
    const int b = blockIdx.y;
    int64_t tb = t[b];
    tb = tb < 0 ? 0 : (tb >= n_steps ? n_steps - 1 : tb);
    const float c1 = tables[tb * 4 + 0], c2 = tables[tb * 4 + 1], var = tables[tb * 4 + 2];
    const float sigma = (tb != 0) ? expf(0.5f * tables[tb * 4 + 3]) : 0.f;
    const size_t base = (size_t)b * row_len;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < row_len; i += stride) {
        const size_t k = base + i;
        float v = c1 * x0[k] + c2 * x_t[k];
        if (ga) v += wa * var * ga[k];
        if (gb) v += wb * var * gb[k];
        if (noise) v += sigma * noise[k];
        out[k] = v;
    }
}

int launch_ddpm_step_table(const float* x_t, const float* x0, const float* noise, const float* ga, float wa,
                           const float* gb, float wb, const float* tables, const int64_t* t, int n_steps,
                           float* out, int B, size_t row_len, hipStream_t s) {
    if (B <= 0 || row_len == 0) return ROHM_OK;
    size_t bx = (row_len + 255) / 256;
    if (bx > 256) bx = 256;
    hipLaunchKernelGGL(ddpm_step_table_kernel, dim3((unsigned)bx, (unsigned)B), dim3(256), 0, s, x_t, x0, noise, ga,
                       wa, gb, wb, tables, t, n_steps, out, row_len);
    ROHM_LAUNCH_CHECK();
    return ROHM_OK;
}

}  // namespace rohm
