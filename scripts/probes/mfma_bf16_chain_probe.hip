// Probe: what does the bf16 MFMA pipe of gfx950 sustain for the instruction / dependency shapes of the plane GEMM?
//   v_mfma_f32_16x16x32_bf16 vs v_mfma_f32_32x32x16_bf16, NACC accumulators visited round-robin (distance between two MFMAs
//   on the same accumulator = NACC), one or two waves per SIMD, random operands (switching activity matters: DVFS).
// Reports ns and shader cycles per MFMA per SIMD, TFLOP/s and the shader clock (s_memtime cycles per wall-clock us).
// build + run on the GPU box: hipcc --offload-arch=gfx950 -O3 -w scripts/probes/mfma_bf16_chain_probe.hip -o /tmp/cp && /tmp/cp
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

template <int KIND, int NACC>
__global__ void probe(float* out, const float* in, int iters, unsigned long long* clk) {
    const int lane = threadIdx.x & 63;
    bf16x8 fa[4], fb[4];
    for (int i = 0; i < 4; ++i) {
        fa[i] = *reinterpret_cast<const bf16x8*>(in + (i * 64 + lane) * 4);
        fb[i] = *reinterpret_cast<const bf16x8*>(in + 2048 + (i * 64 + lane) * 4);
    }
    const unsigned long long c0 = __builtin_readcyclecounter(), w0 = wall_clock64();
    float s = 0.f;
    if constexpr (KIND == 16) {
        f32x4 acc[NACC];
        for (int i = 0; i < NACC; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int rep = 0; rep < 48 / NACC; ++rep)
#pragma unroll
                for (int a = 0; a < NACC; ++a)
                    acc[a] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[(a + rep) & 3], fb[(a * 3 + rep) & 3], acc[a], 0, 0, 0);
        }
        for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][3];
    } else {
        f32x16 acc[NACC];
        for (int i = 0; i < NACC; ++i) for (int k = 0; k < 16; ++k) acc[i][k] = 0.f;
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int rep = 0; rep < 48 / NACC; ++rep)
#pragma unroll
                for (int a = 0; a < NACC; ++a)
                    acc[a] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[(a + rep) & 3], fb[(a * 3 + rep) & 3], acc[a], 0, 0, 0);
        }
        for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][15];
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (blockIdx.x == 0 && threadIdx.x == 0) { clk[0] = __builtin_readcyclecounter() - c0; clk[1] = wall_clock64() - w0; }
}

template <int KIND, int NACC>
void run(int threads, float* out, float* in) {
    unsigned long long* clk;
    hipHostMalloc(&clk, 16);
    const int iters = 4000;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL((probe<KIND, NACC>), dim3(256), dim3(threads), 0, 0, out, in, 10, clk);
    hipEventRecord(e0);
    hipLaunchKernelGGL((probe<KIND, NACC>), dim3(256), dim3(threads), 0, 0, out, in, iters, clk);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    const double per_simd = 48.0 * iters * (threads / 256);                      // MFMAs issued on one SIMD
    const double ghz = (double)clk[0] / ((double)clk[1] * 10.0);
    const double ns = ms * 1e6 / per_simd;
    const double flop = (KIND == 16 ? 2.0 * 16 * 16 * 32 : 2.0 * 32 * 32 * 16);
    printf("%2dx%2d  acc %2d  waves/SIMD %d : %6.2f ns/MFMA/SIMD = %5.1f cycles at %.2f GHz   %7.1f TFLOP/s (%.0f %% of 2500)\n", KIND, KIND, NACC,
           threads / 256, ns, ns * ghz, ghz, flop * 1024.0 / ns / 1e3, flop * 1024.0 / ns / 1e3 / 25.0);
}

int main() {
    float *out, *in;
    hipMalloc(&out, 256 * 512 * 4);
    hipMalloc(&in, 65536 * 4);
    unsigned short* h = (unsigned short*)malloc(65536 * 4);
    unsigned x = 12345u;
    for (int i = 0; i < 131072; ++i) {           // random bf16 bit patterns in [-2, 2)
        x = x * 1664525u + 1013904223u;
        h[i] = (unsigned short)(((x >> 16) & 0x80ff) | 0x3f00 | ((x >> 9) & 0x0080));
    }
    hipMemcpy(in, h, 65536 * 4, hipMemcpyHostToDevice);
    for (int threads : {256, 512}) {
        run<16, 1>(threads, out, in); run<16, 2>(threads, out, in); run<16, 3>(threads, out, in); run<16, 4>(threads, out, in);
        run<16, 8>(threads, out, in); run<16, 12>(threads, out, in);
        run<32, 1>(threads, out, in); run<32, 2>(threads, out, in); run<32, 4>(threads, out, in); run<32, 6>(threads, out, in);
    }
    return 0;
}
