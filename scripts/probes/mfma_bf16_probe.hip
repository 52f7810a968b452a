// Standalone probe for the "precision ladder" item of DESIGN.md §7: what would a split-bf16 GEMM main loop sustain on
// gfx950?  One wave per SIMD, 144x384-tile-like mix per k16 step: 27 v_mfma_f32_32x32x16_bf16 (= the hi*hi, hi*lo,
// lo*hi products of 4.5 x 2 blocks... rounded to 9 x 3), 15 ds_read_b128 fragment reads, 8 LDS-DMA pieces per 40 MFMAs,
// random operands.  Reports TFLOP/s of bf16 MFMA work and the shader clock; fp32-equivalent = /3 (x3 split) or /6 (x6).
// build+run on the GPU box: hipcc --offload-arch=gfx950 -O3 -w mfma_bf16_probe.hip -o /tmp/p && /tmp/p
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int VAR>
__global__ __launch_bounds__(256) void probe(float* out, const float* in, int iters, unsigned long long* clk) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int lane = threadIdx.x & 63;
    for (int i = threadIdx.x; i < 16384; i += 256) smem[i] = in[i];
    __syncthreads();
    const unsigned long long c0 = __builtin_readcyclecounter(), w0 = wall_clock64();
    f32x16 acc[9];
    for (int i = 0; i < 9; ++i) for (int k = 0; k < 16; ++k) acc[i][k] = 0.f;
    bf16x8 fa[3], fb[3];
    for (int i = 0; i < 3; ++i) {
        fa[i] = *reinterpret_cast<bf16x8*>(smem + (i * 64 + lane) * 4);
        fb[i] = *reinterpret_cast<bf16x8*>(smem + 2048 + (i * 64 + lane) * 4);
    }
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int rep = 0; rep < 3; ++rep) {          // 3 x 9 = 27 MFMAs per k16 step
#pragma unroll
            for (int r = 0; r < 3; ++r)
#pragma unroll
                for (int c = 0; c < 3; ++c)
                    acc[r * 3 + c] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[(r + rep) % 3], fb[(c + rep) % 3], acc[r * 3 + c], 0, 0, 0);
            if constexpr (VAR >= 1) {                // 5 fragment reads per 9 MFMAs (15 per 27)
#pragma unroll
                for (int q = 0; q < 5; ++q) {
                    const bf16x8 v = *reinterpret_cast<bf16x8*>(smem + (((it * 3 + rep) * 5 + q) & 31) * 256 + lane * 4);
                    if (q < 3) fa[q] = v; else fb[q - 3] = v;
                }
            }
            if constexpr (VAR >= 2) {                // ~5.4 LDS-DMA pieces per 27 MFMAs -> 2 per 9
#pragma unroll
                for (int q = 0; q < 2; ++q)
                    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(in + ((it + rep * 2 + q) & 15) * 256 + lane * 4),
                                                     (__attribute__((address_space(3))) void*)(smem + 8192 + (rep * 2 + q) * 256), 16, 0, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        if constexpr (VAR >= 2) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
        }
    }
    float s = 0;
    for (int i = 0; i < 9; ++i) for (int k = 0; k < 16; ++k) s += acc[i][k];
    out[blockIdx.x * 256 + threadIdx.x] = s;
    if (blockIdx.x == 0 && threadIdx.x == 0) { clk[0] = __builtin_readcyclecounter() - c0; clk[1] = wall_clock64() - w0; }
}

template <int VAR>
void run(const char* name, float* out, float* in, int iters) {
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
    const double fl = 27.0 * 2 * 32 * 32 * 16 * (double)iters * 1024.0;
    const double tf = fl / (ms * 1e-3) / 1e12;
    printf("%-44s %8.3f ms  %7.1f TFLOP/s bf16  (= %6.1f fp32-equivalent at x3, %6.1f at x6)  clock %.3f GHz\n", name, ms, tf,
           tf / 3, tf / 6, (double)clk[0] / ((double)clk[1] * 10.0));
}

int main() {
    float *out, *in;
    hipMalloc(&out, 256 * 256 * 4);
    hipMalloc(&in, 65536 * 4);
    unsigned short* h = (unsigned short*)malloc(65536 * 4);
    unsigned x = 12345u;
    for (int i = 0; i < 131072; ++i) {           // random bf16 bit patterns in [-2, 2)
        x = x * 1664525u + 1013904223u;
        h[i] = (unsigned short)(((x >> 16) & 0x80ff) | 0x3f00 | ((x >> 9) & 0x0080));
    }
    hipMemcpy(in, h, 65536 * 4, hipMemcpyHostToDevice);
    const int iters = 20000;
    run<0>("32x32x16 bf16, 9 acc, MFMA only", out, in, iters);
    run<1>("  + 15 ds_read_b128 per 27 MFMA", out, in, iters);
    run<2>("  + 6 LDS-DMA (1 KiB) per 27 MFMA + barrier", out, in, iters);
    return 0;
}
