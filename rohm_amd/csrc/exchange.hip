// In-kernel exchanges between workgroups of one launch (the LayerNorm statistics of EPI_BIAS_RES_LN, the stream-K partials of the
// output head: gemm_f32.hip) -- what has to hold around them so that they are safe to ship as the default:
//   * the scratch they meet in is armed on first use (zeroed slots are stale by construction: a tag's XCD field is never 0),
//   * their tags come from a device-side pass counter (recordable into a hipGraph),
//   * they are only used on a device whose layout the kernels' block -> tile maps assume (exchange_layout_ok: a probe launch, not a
//     guess), and a failed wait is reported, never silently survived (rohm_posenet_exchange_status; the Python loops fall back to
//     the exchange-free launches and re-run the chunk).
// Reference work this protects: the post-norm tails of nn.TransformerEncoderLayer, model/posenet.py:63-69, and OutputProcess,
// model/heads.py:171-176.
#include <fcntl.h>
#include <stdlib.h>
#include <string.h>
#include <sys/file.h>
#include <unistd.h>
#include <mutex>
#include "common.h"

namespace rohm {

// zero the slot regions of a scratch whose header does not carry the magic yet (16-byte units; grid-stride)
__global__ __launch_bounds__(256) void exchange_zero_kernel(const unsigned* __restrict__ header, f32x4* __restrict__ za, size_t na,
                                                            f32x4* __restrict__ zb, size_t nb, f32x4* __restrict__ zc, size_t nc) {
    if (header[1] == kExchangeMagic) return;
    const size_t stride = (size_t)gridDim.x * blockDim.x, i0 = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const f32x4 z = f32x4{0.f, 0.f, 0.f, 0.f};
    for (size_t i = i0; i < na; i += stride) za[i] = z;
    for (size_t i = i0; i < nb; i += stride) zb[i] = z;
    for (size_t i = i0; i < nc; i += stride) zc[i] = z;
}

__global__ void exchange_arm_kernel(unsigned* header, int bump) {
    if (header[1] != kExchangeMagic) { header[0] = 0u; header[2] = 0u; header[3] = 0u; header[1] = kExchangeMagic; }
    if (bump) header[2] += 1u;
}

int exchange_arm(unsigned* header, void* za, size_t za_bytes, void* zb, size_t zb_bytes, bool bump, hipStream_t s, void* zc, size_t zc_bytes) {
    ROHM_ARG_CHECK(header && (((uintptr_t)za | (uintptr_t)zb | (uintptr_t)zc | za_bytes | zb_bytes | zc_bytes) & 15) == 0,
                   "exchange_arm: misaligned scratch");
    const size_t units = (za_bytes + zb_bytes + zc_bytes) / 16;
    unsigned blocks = (unsigned)((units + 255) / 256);
    if (blocks > 256) blocks = 256;
    if (blocks == 0) blocks = 1;
    hipLaunchKernelGGL(exchange_zero_kernel, dim3(blocks), dim3(256), 0, s, header, static_cast<f32x4*>(za), za_bytes / 16,
                       static_cast<f32x4*>(zb), zb_bytes / 16, static_cast<f32x4*>(zc), zc_bytes / 16);
    ROHM_LAUNCH_CHECK();
    hipLaunchKernelGGL(exchange_arm_kernel, dim3(1), dim3(1), 0, s, header, bump ? 1 : 0);
    ROHM_LAUNCH_CHECK();
    return ROHM_OK;
}

// ---- layout probe ------------------------------------------------------------------------------------------------------------
// 256 workgroups that request more than half of a CU's LDS (one per CU, like the GEMMs): each records the XCD it runs on, announces
// itself and waits -- bounded -- until all 256 have: that only completes if 256 CUs are free for ONE launch at the same time.
constexpr int kProbeWgs = 256;
__global__ __launch_bounds__(256) void exchange_probe_kernel(unsigned* buf) {      // [0] arrivals, [1 + b] XCD id + 1, [257 + b] saw everybody
    extern __shared__ float probe_pad[];
    if (threadIdx.x != 0) return;
    probe_pad[0] = 0.f;
    buf[1 + blockIdx.x] = 1u + (unsigned)__builtin_amdgcn_s_getreg((3 << 11) | 20);
    __hip_atomic_fetch_add(buf, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
    unsigned ok = 0u;
    for (int it = 0; it < (1 << 15); ++it) {      // ~20 ms at the outside
        if (__hip_atomic_load(buf, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) >= (unsigned)kProbeWgs) { ok = 1u; break; }
        __builtin_amdgcn_s_sleep(8);
    }
    buf[1 + kProbeWgs + blockIdx.x] = ok;
}

static const char* probe_device(int device) {
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, device) != hipSuccess) { (void)hipGetLastError(); return "hipGetDeviceProperties failed"; }
    if (strncmp(prop.gcnArchName, "gfx950", 6) != 0) return "not a gfx950 device";
    if (prop.multiProcessorCount != kProbeWgs) return "the device does not expose 256 CUs (partitioned: CPX / DPX / QPX mode?)";
    const char* guard = getenv("ROHM_EXCHANGE_GUARD");
    const bool probe_only = guard && !strcmp(guard, "probe");
    if (!probe_only) {
        for (const char* name : {"HSA_CU_MASK", "ROC_GLOBAL_CU_MASK"}) {
            const char* v = getenv(name);
            if (v && *v) return "a CU mask is set in the environment (HSA_CU_MASK / ROC_GLOBAL_CU_MASK)";
        }
    }
    int prev = 0;
    if (hipGetDevice(&prev) != hipSuccess || hipSetDevice(device) != hipSuccess) { (void)hipGetLastError(); return "hipSetDevice failed"; }
    // One probe at a time per HOST, across processes: the ranks of a node create their handles at the same moment, each next to
    // its RCCL initialisation; eight 256-workgroup probes are cheap but their bounded waits should not be timed against a host
    // that is busy launching seven others.  Advisory lock, best effort (no lock file = no serialisation, never an error).
    int lock_fd = open("/tmp/.rohm_exchange_probe.lock", O_CREAT | O_RDWR | O_CLOEXEC, 0666);
    if (lock_fd >= 0 && flock(lock_fd, LOCK_EX) != 0) { close(lock_fd); lock_fd = -1; }
    const char* verdict = nullptr;
    unsigned* buf = nullptr;
    unsigned host[1 + 2 * kProbeWgs];
    const size_t lds = 84 * 1024;
    if (hipMalloc(&buf, sizeof(host)) != hipSuccess || hipMemset(buf, 0, sizeof(host)) != hipSuccess ||
        hipFuncSetAttribute(reinterpret_cast<const void*>(&exchange_probe_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) {
        verdict = "probe set-up failed";
    } else {
        hipLaunchKernelGGL(exchange_probe_kernel, dim3(kProbeWgs), dim3(256), lds, 0, buf);
        if (hipGetLastError() != hipSuccess || hipDeviceSynchronize() != hipSuccess ||
            hipMemcpy(host, buf, sizeof(host), hipMemcpyDeviceToHost) != hipSuccess) {
            verdict = "probe launch failed";
        } else {
            bool all = true, map = true;
            unsigned seen = 0u;
            for (int b = 0; b < kProbeWgs; ++b) {
                all = all && host[1 + kProbeWgs + b] == 1u;
                const unsigned x = host[1 + b];
                map = map && x >= 1u && x <= 8u && x == host[1 + b % kNumXCD];
                if (x >= 1u && x <= 8u) seen |= 1u << (x - 1u);
            }
            if (!all) verdict = "256 one-per-CU workgroups were not resident at the same time (CU mask, or another tenant on the device)";
            else if (!map || seen != 0xffu) verdict = "workgroups are not dealt round-robin over 8 XCDs";
        }
    }
    if (buf) (void)hipFree(buf);
    (void)hipGetLastError();
    if (lock_fd >= 0) { (void)flock(lock_fd, LOCK_UN); close(lock_fd); }
    (void)hipSetDevice(prev);
    return verdict;
}

// A verdict that says something about the DEVICE is kept; one that only says the probe itself could not run (set-up / launch failed:
// out of memory at that moment, a capturing null stream, a transient runtime error) is returned but not cached, so a later call
// probes again.
static bool verdict_is_transient(const char* v) {
    return v && (!strcmp(v, "probe set-up failed") || !strcmp(v, "probe launch failed") || !strcmp(v, "hipSetDevice failed") ||
                 !strcmp(v, "hipGetDeviceProperties failed"));
}

static std::mutex g_probe_mutex;      // handles may be created from several host threads: one probe launch per device at a time
static const char* g_reason[64];
static int g_state[64];               // 0 unknown, 1 fine, 2 refused

int exchange_layout_state(int device, const char** why) {
    const char* guard = getenv("ROHM_EXCHANGE_GUARD");
    if (guard && !strcmp(guard, "off")) { if (why) *why = "guard off"; return 1; }
    if (device < 0 || device >= 64) { if (why) *why = "device index >= 64"; return 2; }
    std::lock_guard<std::mutex> lock(g_probe_mutex);
    if (why) *why = g_state[device] == 0 ? "not probed yet" : (g_reason[device] ? g_reason[device] : "whole device, block b on XCD b % 8");
    return g_state[device];
}

bool exchange_layout_ok(int device, const char** why, bool reprobe) {
    static const char* const kOutOfRange = "device index >= 64";
    const char* guard = getenv("ROHM_EXCHANGE_GUARD");
    if (guard && !strcmp(guard, "off")) { if (why) *why = "guard off"; return true; }
    if (device < 0 || device >= 64) { if (why) *why = kOutOfRange; return false; }
    std::lock_guard<std::mutex> lock(g_probe_mutex);
    if (g_state[device] == 0 || reprobe) {
        const char* v = probe_device(device);
        if (verdict_is_transient(v)) {      // says nothing about the device: answer "no" now, ask again next time
            if (why) *why = v;
            g_state[device] = 0;
            return false;
        }
        g_reason[device] = v;
        g_state[device] = v ? 2 : 1;
    }
    if (why) *why = g_reason[device] ? g_reason[device] : "whole device, block b on XCD b % 8";
    return g_state[device] == 1;
}

int device_of_pointer(const void* ptr) {
    hipPointerAttribute_t attr;
    if (ptr && hipPointerGetAttributes(&attr, ptr) == hipSuccess && attr.device >= 0) return attr.device;
    (void)hipGetLastError();
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) { (void)hipGetLastError(); return 0; }
    return dev;
}

}  // namespace rohm
