// C-ABI glue: error channel + the building-block entry points declared in include/rohm_hip.h.
#include <mutex>
#include <string>
#include <string.h>
#include <vector>
#include "common.h"
#include "planes.h"

namespace rohm {
static thread_local std::string g_err;
void set_error(const char* fmt, ...) {
    char buf[1024];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof(buf), fmt, ap);
    va_end(ap);
    g_err = buf;
}
}  // namespace rohm

namespace rohm {
namespace prof {
bool g_active = false;
static bool g_enabled = false;
static int g_stride = 1;
struct Rec { const char* label; double flops, bytes; hipEvent_t a, b; };
static std::vector<Rec> g_recs;
static std::vector<hipEvent_t> g_pool;

static hipEvent_t get_event() {
    if (!g_pool.empty()) { hipEvent_t e = g_pool.back(); g_pool.pop_back(); return e; }
    hipEvent_t e = nullptr;
    (void)hipEventCreate(&e);
    return e;
}
Scope::Scope(const char* label, double flops, double bytes, hipStream_t s) : idx(-1), stream(s) {
    if (!g_active) return;
    Rec r{label, flops, bytes, get_event(), get_event()};
    (void)hipEventRecord(r.a, s);
    idx = (int)g_recs.size();
    g_recs.push_back(r);
}
Scope::~Scope() {
    if (idx >= 0) (void)hipEventRecord(g_recs[idx].b, stream);
}
void set_step(int step) { g_active = g_enabled && (step % g_stride == 0); }
bool enabled() { return g_enabled; }
static bool g_detail = false;
bool detail() { return g_detail && g_active; }
bool detail_requested() { return g_detail && g_enabled; }
const char* intern(const char* s) {        // stable storage for labels composed at launch time (detail mode only)
    static std::vector<std::string*> pool;
    static std::mutex mu;                  // launches may come from several host threads (one stream each)
    std::lock_guard<std::mutex> lock(mu);
    for (std::string* q : pool)
        if (*q == s) return q->c_str();
    pool.push_back(new std::string(s));
    return pool.back()->c_str();
}
}  // namespace prof
}  // namespace rohm

using namespace rohm;

extern "C" {

int rohm_profile_start(int step_stride) {
    prof::g_enabled = true;
    prof::g_stride = step_stride > 0 ? step_stride : 1;
    prof::g_active = true;
    return ROHM_OK;
}

int rohm_profile_detail(int on) {
    prof::g_detail = on != 0;
    return ROHM_OK;
}

int rohm_profile_stop(rohm_profile_row* rows, int max_rows, int* n_rows) {
    prof::g_enabled = false;
    prof::g_active = false;
    int n = 0;
    for (auto& r : prof::g_recs) {
        float ms = 0.f;
        hipError_t e = hipEventSynchronize(r.b);
        if (e == hipSuccess) e = hipEventElapsedTime(&ms, r.a, r.b);
        prof::g_pool.push_back(r.a);
        prof::g_pool.push_back(r.b);
        if (e != hipSuccess) continue;
        int k = 0;
        for (; k < n; ++k)
            if (strncmp(rows[k].name, r.label, sizeof(rows[k].name) - 1) == 0) break;
        if (k == n) {
            if (n >= max_rows) continue;
            memset(&rows[n], 0, sizeof(rows[n]));
            strncpy(rows[n].name, r.label, sizeof(rows[n].name) - 1);
            ++n;
        }
        rows[k].launches += 1;
        rows[k].total_ms += ms;
        rows[k].flops += r.flops;
        rows[k].bytes += r.bytes;
    }
    prof::g_recs.clear();
    if (n_rows) *n_rows = n;
    return ROHM_OK;
}

const char* rohm_last_error(void) { return g_err.c_str(); }
int rohm_version(void) { return 100; }

int rohm_gemm_f32(const float* A, int lda, const float* W, int ldw, float* C, int ldc, int M, int N, int K,
                  const float* bias, const float* R, int ldr, int epi, rohm_stream_t stream) {
    ROHM_ARG_CHECK(A && W && C, "gemm: null pointer");
    ROHM_ARG_CHECK(epi >= EPI_BIAS && epi <= EPI_BIAS_RES, "gemm: public epilogues are 0..2 (got %d)", epi);
    ROHM_ARG_CHECK(epi != EPI_BIAS_RES || R, "gemm: epilogue 2 needs a residual");
    GemmParams g{};
    g.A = A; g.lda = lda; g.W = W; g.ldw = ldw; g.C = C; g.ldc = ldc; g.M = M; g.N = N; g.K = K;
    g.bias = bias; g.R = R; g.ldr = ldr;
    return launch_gemm(g, epi, (hipStream_t)stream);
}

// Host part of the launch tags of the stand-alone exchanging entry points: a hash of the scratch ADDRESS (low six bits clear: the
// kernels add the launch index and the XCD field there).  A recycled allocator block whose old header survives (magic set) while its
// slot region overlaps what used to be ANOTHER scratch's slots cannot produce a matching tag: that scratch lived at another address.
static unsigned standalone_salt(const void* scratch) {
    unsigned v = (unsigned)(uintptr_t)scratch ^ (unsigned)((uintptr_t)scratch >> 32) ^ 0x5a17c0u;
    v ^= v >> 16; v *= 0x7feb352du; v ^= v >> 15; v *= 0x846ca68bu; v ^= v >> 16;
    return v & ~63u;
}

int rohm_exchange_probe(int device, const char** why) {
    const char* w = "";
    const bool ok = exchange_layout_ok(device, &w, true);
    if (why) *why = w;
    return ok ? 1 : 0;
}

size_t rohm_gemm_res_layernorm_scratch_bytes(int M, int N) { return gemm_ln_supported(M, N, 32) ? gemm_ln_scratch_bytes(M, N) : 0; }

int rohm_gemm_res_layernorm_f32(const float* A, int lda, const float* W, int ldw, float* C, int ldc, int M, int N, int K,
                                const float* bias, const float* R, int ldr, const float* gamma, const float* beta, float eps,
                                void* scratch, size_t scratch_bytes, rohm_stream_t stream) {
    ROHM_ARG_CHECK(A && W && C && bias && R && gamma && beta && scratch, "gemm_res_layernorm: null pointer");
    if (!gemm_ln_supported(M, N, K)) {
        set_error("gemm_res_layernorm: shape (%d, %d, %d) has no in-kernel LayerNorm form (M %% 144, K %% 32, N / 64 or N / 128 in 1, 2, 4, 8)", M, N, K);
        return ROHM_ERR_UNSUPPORTED;
    }
    ROHM_ARG_CHECK(scratch_bytes >= gemm_ln_scratch_bytes(M, N) && (((uintptr_t)scratch) & 63) == 0, "gemm_res_layernorm: scratch too small / misaligned");
    GemmParams g{};
    g.A = A; g.lda = lda; g.W = W; g.ldw = ldw; g.C = C; g.ldc = ldc; g.M = M; g.N = N; g.K = K;
    g.bias = bias; g.R = R; g.ldr = ldr; g.ln_gamma = gamma; g.ln_beta = beta; g.ln_dim = N; g.ln_eps = eps;
    // the column tiles meet in L2 while they run: only on a device laid out as the kernel's block -> tile map assumes (exchange.hip).
    // The device is the one that owns the scratch (not whatever hipGetDevice says); an un-probed device is probed HERE only when the
    // stream is not recording a graph -- the probe allocates and synchronises -- and a capturing call on an un-probed device is
    // refused without caching anything (rohm_exchange_probe(device) up front avoids both).
    const int dev = device_of_pointer(scratch);
    const char* why = "";
    int st = exchange_layout_state(dev, &why);
    if (st == 0) {
        if (stream_is_capturing((hipStream_t)stream)) {
            set_error("gemm_res_layernorm: device %d has not been probed and the stream is recording a graph: call rohm_exchange_probe(%d) "
                      "before the capture, or use rohm_gemm_f32(epi 2) + rohm_layernorm_f32", dev, dev);
            return ROHM_ERR_UNSUPPORTED;
        }
        st = exchange_layout_ok(dev, &why) ? 1 : 2;
    }
    if (st != 1) {
        set_error("gemm_res_layernorm: the in-kernel LayerNorm needs a whole MI355X (%s): use rohm_gemm_f32(epi 2) + rohm_layernorm_f32", why);
        return ROHM_ERR_UNSUPPORTED;
    }
    gemm_ln_bind(g, scratch);
    g.xln_epoch = standalone_salt(scratch);      // + 64 x the scratch's pass counter, advanced by exchange_arm
    int rc = exchange_arm(static_cast<unsigned*>(scratch), static_cast<char*>(scratch) + 64, gemm_ln_scratch_bytes(M, N) - 64, nullptr, 0, true,
                          (hipStream_t)stream);
    if (rc) return rc;
    return launch_gemm(g, EPI_BIAS_RES_LN, (hipStream_t)stream);
}

size_t rohm_output_process_scratch_bytes(void) { return 256 + gemm_sk_scratch_bytes(); }

int rohm_output_process_plan(int B, int T, int D, int C_out, int* units_per_workgroup, int* tiles_per_xcd) {
    int u = 0, t8 = 0;
    const bool sk = B > 0 && T > 0 && D > 0 && C_out > 0 && gemm_sk_plan(C_out, B * (T + 1), D, &u, &t8);
    if (units_per_workgroup) *units_per_workgroup = sk ? u : 0;
    if (tiles_per_xcd) *tiles_per_xcd = sk ? t8 : 0;
    return sk ? 1 : 0;
}

int rohm_output_process_f32(const float* h, const float* w, const float* b, float* out, int B, int T, int D, int C_out,
                            int ch_off, int C_total, void* scratch, size_t scratch_bytes, rohm_stream_t stream) {
    ROHM_ARG_CHECK(h && w && b && out, "output_process: null pointer");
    ROHM_ARG_CHECK(B > 0 && T > 0 && D > 0 && C_out > 0 && ch_off >= 0 && ch_off + C_out <= C_total, "output_process: bad shape");
    GemmParams g{};
    g.A = w; g.lda = D; g.W = h; g.ldw = D; g.C = out; g.M = C_out; g.N = B * (T + 1); g.K = D;
    g.bias = b; g.S = T + 1; g.ch_off = ch_off; g.C_total = C_total; g.T = T;
    if (scratch) {
        ROHM_ARG_CHECK(scratch_bytes >= rohm_output_process_scratch_bytes() && (((uintptr_t)scratch) & 255) == 0,
                       "output_process: scratch too small / misaligned");
        const int dev = device_of_pointer(scratch);
        int st = exchange_layout_state(dev, nullptr);
        if (st == 0 && !stream_is_capturing((hipStream_t)stream)) st = exchange_layout_ok(dev, nullptr) ? 1 : 2;
        if (st == 1) {      // else (refused, or un-probed under a graph capture -- not cached): plain tiles, like a call without scratch
            gemm_sk_bind(g, static_cast<char*>(scratch) + 256, static_cast<unsigned*>(scratch));
            g.xln_epoch = standalone_salt(scratch) + 1u;
            int rc = exchange_arm(static_cast<unsigned*>(scratch), static_cast<char*>(scratch) + 256, 256 * sizeof(unsigned long long), nullptr, 0,
                                  true, (hipStream_t)stream);
            if (rc) return rc;
        }
    }
    return launch_gemm(g, EPI_OUT_T, (hipStream_t)stream);
}

int rohm_layernorm_f32(float* x, const float* gamma, const float* beta, int M, int D, rohm_stream_t stream) {
    ROHM_ARG_CHECK(x && gamma && beta, "layernorm: null pointer");
    return launch_layernorm(x, gamma, beta, M, D, (hipStream_t)stream);
}

int rohm_attention_f32(const float* qkv, float* ctx, int n_seq, int n_head, int n_tok, int head_dim,
                       rohm_stream_t stream) {
    ROHM_ARG_CHECK(qkv && ctx, "attention: null pointer");
    return launch_attention(qkv, ctx, n_seq, n_head, n_tok, head_dim, (hipStream_t)stream);
}

size_t rohm_planes_bytes(int rows, int K, int nplane) {
    if (rows <= 0 || K <= 0 || nplane <= 0) return 0;
    return plane_tensor_bytes(rows, K, nplane);
}

int rohm_planes_split(const float* X, int ldx, int rows, int K, int nplane, float scale, void* planes, rohm_stream_t stream) {
    return launch_plane_split(X, ldx, rows, K, nplane, scale, planes, (hipStream_t)stream);
}

int rohm_gemm_planes(const void* Ap, const void* Wp, float* C, int ldc, void* Cp, int M, int N, int K,
                     const float* bias, const float* R, int ldr, int qcols, float qscale, float acc_scale, int epi, int nplane,
                     int flags, rohm_stream_t stream) {
    PlaneGemmParams g{};
    g.Ap = Ap; g.Wp = Wp; g.C = C; g.ldc = ldc; g.Cp = Cp; g.M = M; g.N = N; g.K = K;
    g.bias = bias; g.R = R; g.ldr = ldr; g.qcols = qcols; g.qscale = qscale; g.acc_scale = acc_scale; g.no_swap = flags & 1;
    return launch_gemm_pp(g, epi, nplane, (hipStream_t)stream);
}

int rohm_gemm_planes_ln(const void* Ap, const void* Wp, float* C, int ldc, void* Cp, int M, int N, int K,
                        const float* bias, const float* R, int ldr, int qcols, float qscale, float acc_scale, int epi, int nplane,
                        const float* ln_stats, const float* ln_c, const float* r_stats, const float* r_gamma,
                        const float* r_beta, float* out_stats, int ln_dim, float ln_eps, rohm_stream_t stream) {
    PlaneGemmParams g{};
    g.Ap = Ap; g.Wp = Wp; g.C = C; g.ldc = ldc; g.Cp = Cp; g.M = M; g.N = N; g.K = K;
    g.bias = bias; g.R = R; g.ldr = ldr; g.qcols = qcols; g.qscale = qscale; g.acc_scale = acc_scale;
    g.ln_stats = ln_stats; g.ln_c = ln_c; g.r_stats = r_stats; g.r_gamma = r_gamma; g.r_beta = r_beta; g.out_stats = out_stats;
    g.ln_dim = ln_dim; g.ln_eps = ln_eps;
    return launch_gemm_pp(g, epi, nplane, (hipStream_t)stream);
}

int rohm_layernorm_planes_f32(float* x, const float* gamma, const float* beta, int M, int D, int nplane,
                              void* planes, rohm_stream_t stream) {
    ROHM_ARG_CHECK(x && gamma && beta && planes, "layernorm_planes: null pointer");
    return launch_layernorm_planes(x, gamma, beta, M, D, nplane, planes, (hipStream_t)stream);
}

int rohm_attention_planes_f32(const float* qkv, void* ctx_planes, int n_seq, int n_head, int nplane,
                              rohm_stream_t stream) {
    return launch_attention_planes(qkv, ctx_planes, n_seq, n_head, nplane, (hipStream_t)stream);
}

int rohm_ddpm_step(const float* x_t, const float* x0, const float* noise, const float* guid_grad, float c1,
                   float c2, float sigma, float guid_scale, float* x_prev, size_t n, rohm_stream_t stream) {
    ROHM_ARG_CHECK(x_t && x0 && x_prev, "ddpm_step: null pointer");
    ROHM_ARG_CHECK(sigma == 0.f || noise, "ddpm_step: noise required when sigma != 0");
    return launch_ddpm_step(x_t, x0, noise, guid_grad, c1, c2, sigma, guid_scale, x_prev, n, (hipStream_t)stream);
}

int rohm_ddpm_step_table(const float* x_t, const float* x0, const float* noise, const float* grad_a, float w_a,
                         const float* grad_b, float w_b, const float* tables, const int64_t* t, int n_steps,
                         float* x_prev, int B, size_t row_len, rohm_stream_t stream) {
    ROHM_ARG_CHECK(x_t && x0 && x_prev && tables && t, "ddpm_step_table: null pointer");
    ROHM_ARG_CHECK(n_steps > 0, "ddpm_step_table: empty schedule");
    return launch_ddpm_step_table(x_t, x0, noise, grad_a, w_a, grad_b, w_b, tables, t, n_steps, x_prev, B, row_len,
                                  (hipStream_t)stream);
}

}  // extern "C"
