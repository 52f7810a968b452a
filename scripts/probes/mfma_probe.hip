// Standalone probe: what stops a one-wave-per-SIMD fp32 MFMA stream from reaching peak on gfx950?
// build+run on the GPU box: hipcc --offload-arch=gfx950 -O3 mfma_probe.hip -o /tmp/mfma_probe && /tmp/mfma_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int VAR>
__global__ __launch_bounds__(256) void probe(float* out, const float* in, int iters, unsigned long long* clk) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int lane = threadIdx.x & 63;
    for (int i = threadIdx.x; i < 8192; i += 256) smem[i] = in[i];
    __syncthreads();
    const unsigned long long c0 = __builtin_readcyclecounter(), w0 = wall_clock64();
    if constexpr (VAR < 10) {
        f32x4 acc[18];
        for (int i = 0; i < 18; ++i) acc[i] = f32x4{0, 0, 0, 0};
        f32x4 a = *reinterpret_cast<f32x4*>(smem + lane * 4);
        f32x4 b = *reinterpret_cast<f32x4*>(smem + 1024 + lane * 4);
        f32x4 fr[9];
        for (int i = 0; i < 9; ++i) fr[i] = a;
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int r = 0; r < 9; ++r) {
#pragma unroll
                for (int c = 0; c < 2; ++c)
#pragma unroll
                    for (int j = 0; j < 4; ++j)
                        acc[r * 2 + c] = __builtin_amdgcn_mfma_f32_16x16x4f32(b[j], fr[r][j], acc[r * 2 + c], 0, 0, 0);
                if constexpr (VAR >= 1)   // one LDS fragment read per 8 MFMAs, consumed next iteration
                    fr[r] = *reinterpret_cast<f32x4*>(smem + ((it * 9 + r) & 15) * 256 + lane * 4);
                if constexpr (VAR >= 3) {
                    if (r < 9) __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(in + ((it + r) & 7) * 256 + lane * 4),
                                                                (__attribute__((address_space(3))) void*)(smem + 4096 + r * 256), 16, 0, 0);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
            if constexpr (VAR >= 2) {
                if constexpr (VAR >= 3) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                __syncthreads();
            }
        }
        f32x4 s = f32x4{0, 0, 0, 0};
        for (int i = 0; i < 18; ++i) s += acc[i];
        out[blockIdx.x * 256 + threadIdx.x] = s[0] + s[1] + s[2] + s[3];
    } else {
        // 32x32x2: 4 accumulators of 16 regs, same flops per iteration (4.5 blocks -> use 4: 128x32 per wave)
        f32x16 acc[4];
        for (int i = 0; i < 4; ++i) for (int k = 0; k < 16; ++k) acc[i][k] = 0;
        f32x4 a = *reinterpret_cast<f32x4*>(smem + lane * 4);
        f32x4 b = *reinterpret_cast<f32x4*>(smem + 1024 + lane * 4);
        f32x4 fa[4];
        for (int i = 0; i < 4; ++i) fa[i] = a;
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int rep = 0; rep < 4; ++rep) {
#pragma unroll
                for (int r = 0; r < 4; ++r)
#pragma unroll
                    for (int j = 0; j < 4; ++j) acc[r] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[r][j], b[j], acc[r], 0, 0, 0);
                if constexpr (VAR >= 11) {   // 2 LDS fragment reads per 16 MFMAs (same LDS bytes per flop as the GEMM)
                    fa[rep] = *reinterpret_cast<f32x4*>(smem + ((it * 4 + rep) & 15) * 256 + lane * 4);
                    b = *reinterpret_cast<f32x4*>(smem + 1024 + ((it * 4 + rep) & 7) * 256 + lane * 4);
                }
                if constexpr (VAR >= 13) {
#pragma unroll
                    for (int q = 0; q < 2; ++q)
                        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(in + ((it + rep * 2 + q) & 7) * 256 + lane * 4),
                                                         (__attribute__((address_space(3))) void*)(smem + 4096 + (rep * 2 + q) * 256), 16, 0, 0);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
            if constexpr (VAR >= 12) {
                if constexpr (VAR >= 13) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                __syncthreads();
            }
        }
        float s = 0;
        for (int i = 0; i < 4; ++i) for (int k = 0; k < 16; ++k) s += acc[i][k];
        out[blockIdx.x * 256 + threadIdx.x] = s;
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        clk[0] = __builtin_readcyclecounter() - c0;
        clk[1] = wall_clock64() - w0;
    }
}

template <int VAR>
void run(const char* name, float* out, float* in, double flops_per_iter_per_wave, int iters = 2000) {
    unsigned long long* clk;
    hipHostMalloc(&clk, 16);
    hipFuncSetAttribute(reinterpret_cast<const void*>(&probe<VAR>), hipFuncAttributeMaxDynamicSharedMemorySize, 100 * 1024);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(probe<VAR>, dim3(256), dim3(256), 100 * 1024, 0, out, in, 10, clk);
    hipEventRecord(e0);
    hipLaunchKernelGGL(probe<VAR>, dim3(256), dim3(256), 100 * 1024, 0, out, in, iters, clk);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    const double fl = flops_per_iter_per_wave * iters * 1024.0;
    printf("%-46s %8.3f ms  %7.1f TFLOP/s   shader clock %.3f GHz (s_memtime / 100 MHz wall clock)\n", name, ms,
           fl / (ms * 1e-3) / 1e12, (double)clk[0] / ((double)clk[1] * 10.0));
}

int main() {
    float *out, *in;
    hipMalloc(&out, 256 * 256 * 4);
    hipMalloc(&in, 65536 * 4);
    hipMemset(in, 0, 65536 * 4);
    const bool rnd = getenv("PROBE_RANDOM") != nullptr;
    if (rnd) {
        float* h = (float*)malloc(65536 * 4);
        unsigned x = 12345u;
        for (int i = 0; i < 65536; ++i) { x = x * 1664525u + 1013904223u; h[i] = ((x >> 8) & 0xffff) / 65536.0f - 0.5f; }
        hipMemcpy(in, h, 65536 * 4, hipMemcpyHostToDevice);
        free(h);
    }
    const int long_iters = getenv("PROBE_ITERS") ? atoi(getenv("PROBE_ITERS")) : 2000;
    printf("data: %s, iters %d\n", rnd ? "random" : "zero", long_iters);
    const double f16 = 72.0 * 2 * 16 * 16 * 4;       // 72 MFMAs 16x16x4 per iteration
    const double f32 = 64.0 * 2 * 32 * 32 * 2;       // 64 MFMAs 32x32x2 per iteration
    run<0>("16x16x4, 18 acc, MFMA only", out, in, f16, long_iters);
    run<1>("  + 1 ds_read_b128 per 8 MFMA", out, in, f16, long_iters);
    run<2>("  + barrier per 72 MFMA", out, in, f16, long_iters);
    run<3>("  + 9 LDS-DMA (1 KiB) per 72 MFMA", out, in, f16, long_iters);
    run<10>("32x32x2, 4 acc, MFMA only", out, in, f32, long_iters);
    run<11>("  + 2 ds_read_b128 per 16 MFMA", out, in, f32, long_iters);
    run<12>("  + barrier per 64 MFMA", out, in, f32, long_iters);
    run<13>("  + 8 LDS-DMA (1 KiB) per 64 MFMA", out, in, f32, long_iters);
    return 0;
}
