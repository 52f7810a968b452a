// Probe: wave_sum64 (common.h: v_permlane32/16_swap + DPP row rotations) against the __shfl_xor butterfly it replaces -- the two must
// agree BIT FOR BIT in every lane, for random data, for data of mixed magnitudes and for NaN / inf payloads.
// build + run on a GPU box:  hipcc --offload-arch=gfx950 -O3 -std=c++17 -Irohm_amd/csrc -Iinclude -o /tmp/wsp scripts/probes/wave_sum_probe.hip && /tmp/wsp
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <math.h>
#include "common.h"

__global__ void probe(const float* in, unsigned* a, unsigned* b) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    float v = in[i];
    float x = v;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) x += __shfl_xor(x, o);
    a[i] = __float_as_uint(x);
    b[i] = __float_as_uint(rohm::wave_sum64(v));
}

int main() {
    const int n = 64 * 4096;
    float* h = (float*)malloc(n * sizeof(float));
    srand(7);
    for (int i = 0; i < n; ++i) {
        const float u = (float)rand() / RAND_MAX - 0.5f;
        const int e = rand() % 40 - 20;
        h[i] = (i / 64) % 3 == 0 ? u : ldexpf(u, e);                 // every third wave: same magnitude; others: 2^-20 .. 2^19
    }
    h[64 * 5 + 17] = INFINITY; h[64 * 6 + 3] = NAN; h[64 * 7 + 63] = -0.0f;
    float* d; unsigned *da, *db;
    hipMalloc(&d, n * 4); hipMalloc(&da, n * 4); hipMalloc(&db, n * 4);
    hipMemcpy(d, h, n * 4, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(probe, dim3(n / 256), dim3(256), 0, 0, d, da, db);
    unsigned* ha = (unsigned*)malloc(n * 4); unsigned* hb = (unsigned*)malloc(n * 4);
    hipMemcpy(ha, da, n * 4, hipMemcpyDeviceToHost); hipMemcpy(hb, db, n * 4, hipMemcpyDeviceToHost);
    int bad = 0, nan_waves = 0;
    for (int i = 0; i < n; ++i) {
        const bool both_nan = (ha[i] & 0x7fffffffu) > 0x7f800000u && (hb[i] & 0x7fffffffu) > 0x7f800000u;
        if (both_nan) { nan_waves += (i % 64 == 0); continue; }
        if (ha[i] != hb[i] && bad++ < 8) printf("lane %d of wave %d: butterfly %08x  wave_sum64 %08x\n", i % 64, i / 64, ha[i], hb[i]);
    }
    printf("wave_sum64 vs __shfl_xor butterfly: %d lanes compared, %d differ (%d NaN waves skipped)\n", n, bad, nan_waves);
    return bad != 0;
}
