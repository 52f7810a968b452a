// Phase timeline of attention_f32_kernel (s_memtime stamps per wave), plus event timing of the un-instrumented
// launch shape.  Build + run on the GPU box:
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -DAT_TIMELINE scripts/probes/attn_timeline.hip -o /tmp/attn_tl && /tmp/attn_tl 64
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#include <algorithm>
#include "../../rohm_amd/csrc/attention_f32.hip"

namespace rohm {
void set_error(const char*, ...) {}
namespace prof {
bool g_active = false;
Scope::Scope(const char*, double, double, hipStream_t s) : idx(-1), stream(s) {}
Scope::~Scope() {}
}  // namespace prof
}  // namespace rohm

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

template <int NW>
static void run(int B, int split) {
    using namespace rohm;
    const int n_head = 4, items = B * n_head, D = 512;
    const size_t nq = (size_t)B * 144 * 3 * D, no = (size_t)B * 144 * D;
    std::vector<float> h(nq);
    unsigned s = 12345;
    for (auto& v : h) { s = s * 1664525u + 1013904223u; v = ((s >> 8) / 16777216.0f - 0.5f) * 2.0f; }
    float *qkv, *ctx; unsigned long long* tl;
    CK(hipMalloc(&qkv, nq * 4)); CK(hipMalloc(&ctx, no * 4));
    const int grid = split ? ((items + 7) / 8) * 16 : items;
    CK(hipMalloc(&tl, (size_t)grid * NW * 16 * 8)); CK(hipMemset(tl, 0, (size_t)grid * NW * 16 * 8));
    CK(hipMemcpy(qkv, h.data(), nq * 4, hipMemcpyHostToDevice));
    const size_t lds = AT_LDS_FLOATS * sizeof(float);
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(&attention_f32_kernel<NW>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int it = 0; it < 5; ++it)
        hipLaunchKernelGGL(attention_f32_kernel<NW>, dim3(grid), dim3(NW * 64), lds, 0, qkv, ctx, n_head, items, split, tl);
    CK(hipEventRecord(e0));
    const int R = 50;
    for (int it = 0; it < R; ++it)
        hipLaunchKernelGGL(attention_f32_kernel<NW>, dim3(grid), dim3(NW * 64), lds, 0, qkv, ctx, n_head, items, split, tl);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    printf("B=%d NW=%d split=%d grid=%d: %.2f us per launch (instrumented), %.1f TFLOP/s\n", B, NW, split, grid, ms / R * 1e3,
           4.0 * 144 * 144 * 128 * items / (ms / R * 1e-3) / 1e12);
    std::vector<unsigned long long> t((size_t)grid * NW * 16);
    CK(hipMemcpy(t.data(), tl, t.size() * 8, hipMemcpyDeviceToHost));
    const char* names[12] = {"start", "Q landed+barrier", "K g0", "K g1", "K g2", "QK done", "softmax done", "V barrier", "PV owned done",
                             "coop partial (pre-PV)", "coop barrier", "end"};
    for (int half = 0; half < (split ? 2 : 1); ++half) {
        printf("  %s: phase end (cycles since wave start): mean over waves [min .. max]\n", split ? (half ? "second half (q 5-8)" : "first half (q 0-4, coop)") : "full");
        for (int i = 1; i < 12; ++i) {
            double sum = 0; unsigned long long mn = ~0ull, mx = 0; size_t n = 0;
            for (int b = 0; b < grid; ++b) {
                if (split && ((b >> 3) & 1) != half) continue;
                for (int w = 0; w < NW; ++w) {
                    const unsigned long long* r = &t[((size_t)b * NW + w) * 16];
                    if (r[0] == 0 || r[i] == 0) continue;
                    const unsigned long long d = r[i] - r[0];
                    sum += d; mn = std::min(mn, d); mx = std::max(mx, d); ++n;
                }
            }
            if (n) printf("    %-22s %9.0f [%8llu .. %8llu]  (n=%zu)\n", names[i], sum / n, mn, mx, n);
        }
    }
    CK(hipFree(qkv)); CK(hipFree(ctx)); CK(hipFree(tl));
}

int main(int argc, char** argv) {
    const int B = argc > 1 ? atoi(argv[1]) : 64;
    const int mode = argc > 2 ? atoi(argv[2]) : -1;      // 0 full, 1 split, -1 both
    if (mode != 1) run<8>(B, 0);
    if (mode != 0) run<4>(B, 1);
    return 0;
}
