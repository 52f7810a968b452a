// How fast can the workgroups of ONE XCD stream weights, and how fast can all eight XCDs stream the SAME weights at the same time?
// (the two rates that bound the clip-resident TrajNet step, csrc/trajnet_resident.hip: every XCD reads all 90 MB of weights per step.)
// 256 one-per-CU workgroups (block b on XCD b % 8), LDS-DMA 16-byte loads like the kernels', DEPTH 1-KiB pieces per wave in flight.
//   mode 0: only XCD 0's 32 workgroups read the buffer (each 1/32 of it)           -> one XCD's pull rate
//   mode 1: every XCD's 32 workgroups read the WHOLE buffer (8 x the traffic)      -> the resident step's pattern at B >= 8
//   mode 2: 256 workgroups read 1/256 of the buffer each                             -> every byte once chip-wide (split-K launches)
// Build + run on the GPU box:  hipcc --offload-arch=gfx950 -O3 -std=c++17 scripts/probes/xcd_stream_probe.hip -o /tmp/xcd_stream && /tmp/xcd_stream
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

template <int DEPTH>
__global__ __launch_bounds__(256) void stream_kernel(const float* __restrict__ buf, size_t floats, int mode, float* sink) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int xcd = blockIdx.x % 8, j = blockIdx.x / 8;
    if (mode == 0 && xcd != 0) return;
    const int parts = (mode == 2) ? 256 : 32, part = (mode == 2) ? (int)blockIdx.x : j;
    const size_t per = floats / parts;                       // multiple of DEPTH * 1024 floats by construction
    const float* src = buf + (size_t)part * per;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    // a wave walks its quarter of the slice in bursts of DEPTH pieces of 1 KiB (64 lanes x 16 B)
    const size_t wper = per / 4;
    const float* wsrc = src + (size_t)wave * wper + lane * 4;
    for (size_t off = 0; off < wper; off += (size_t)DEPTH * 256) {
#pragma unroll
        for (int d = 0; d < DEPTH; ++d)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(wsrc + off + d * 256),
                                             (__attribute__((address_space(3))) void*)(lds + (wave * DEPTH + d) * 256), 16, 0, 0);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    if (sink && lds[threadIdx.x] == 12345.678f) sink[0] = 1.f;
}

template <int DEPTH>
static void run(const float* buf, size_t floats, float* sink) {
    const size_t lds = (size_t)4 * DEPTH * 1024;
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(&stream_kernel<DEPTH>), hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int mode = 0; mode < 3; ++mode) {
        for (int it = 0; it < 3; ++it) hipLaunchKernelGGL(stream_kernel<DEPTH>, dim3(256), dim3(256), lds > 84 * 1024 ? lds : 84 * 1024, 0, buf, floats, mode, sink);
        CK(hipEventRecord(e0));
        const int reps = 10;
        for (int it = 0; it < reps; ++it) hipLaunchKernelGGL(stream_kernel<DEPTH>, dim3(256), dim3(256), lds > 84 * 1024 ? lds : 84 * 1024, 0, buf, floats, mode, sink);
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms = 0.f;
        CK(hipEventElapsedTime(&ms, e0, e1));
        const double us = ms * 1e3 / reps, bytes = (double)floats * 4 * (mode == 1 ? 8 : 1);
        printf("depth %2d KiB/wave  mode %d (%s): %8.1f us per pass of %.0f MB  ->  %.2f TB/s%s\n", DEPTH, mode,
               mode == 0 ? "XCD 0 alone      " : mode == 1 ? "8 XCDs, same data" : "chip-wide once   ", us, floats * 4 / 1e6, bytes / us / 1e6,
               mode == 1 ? " aggregate" : "");
    }
}

int main() {
    const size_t floats = (size_t)96 * 256 * 1024;           // 101 MB (TrajNet: 90 MB of weights); per-wave slices are whole bursts at every depth
    float *buf, *sink;
    CK(hipMalloc(&buf, floats * 4)); CK(hipMalloc(&sink, 64));
    CK(hipMemset(buf, 0, floats * 4));
    run<2>(buf, floats, sink);
    run<8>(buf, floats, sink);
    run<16>(buf, floats, sink);
    return 0;
}
