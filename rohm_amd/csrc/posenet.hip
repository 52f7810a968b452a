// PoseNet on MI355X: handle, forward (model/posenet.py:75-96) and the device-resident DDPM loop
// (diffusion/gaussian_diffusion_posenet.py:578-662).
//
// Data layout in HBM.  The reference keeps activations sequence-first [S, B, D] and pays a permute
// copy around every block (41 % of its CPU time).  Here everything between the input
// embed and the output head is TOKEN-MAJOR: row m = b*S + tok of a [B*S, D] matrix, S = T + 1 tokens,
// token 0 = timestep token (posenet.py:90).  A clip is 144 consecutive rows = exactly one GEMM
// M-tile and one attention workgroup per head.
//
// Per step:  pack(x_t | cond) -> [B*S, 608]   (transpose of the [B, C, 1, T] motion tensor)
//            timestep token    -> tab0[B, D]  (heads.py:145-146, + pe[0])
//            embed GEMM  K=608 (= 294 + 294 + pad; both InputProcess Linears fused), + pe
//            8 x { QKV GEMM (q pre-scaled) -> attention -> out-proj GEMM + residual -> LayerNorm
//                  -> FF1 GEMM + GELU -> FF2 GEMM + residual -> LayerNorm }
//            output head as a transposed GEMM (rows = 272 channels, cols = tokens) that stores
//            straight into the [B, C, 1, T] layout, then `finish` copies the trajectory channels
//            from cond (posenet.py:94-95) and applies the DDPM update.
#include <stdlib.h>
#include <string.h>
#include <vector>
#include <atomic>
#include "common.h"
#include "planes.h"

namespace rohm {

constexpr int kMaxTok = 256;
constexpr int kLoopChunk = 1024;   // max denoising steps per rohm_posenet_sample_loop call
constexpr float kF16WeightScale = 256.0f;      // fp16x3 mode: weight planes are cut from 2^8 w

struct LayerW {
    float *in_w, *in_b, *out_w, *out_b, *l1_w, *l1_b, *l2_w, *l2_b, *n1_w, *n1_b, *n2_w, *n2_b;
    // LayerNorm folding (common.h GemmParams): with folding on, l1_w is gamma1-scaled, l1_b holds d, l1_c holds c;
    // for layers >= 1 in_w is scaled by the PREVIOUS layer's gamma2, in_b holds d and in_c holds c.
    float *in_c, *l1_c;
    // bf16 planes of the four weight matrices (precision ladder, planes.h); null in the exact-fp32 mode
    char *in_wp, *out_wp, *l1_wp, *l2_wp;
    // LayerNorm folded into the plane GEMMs (pp_fold): planes of gamma-scaled weights and the fold vectors (planes.h PlaneGemmParams)
    //   l1_wpf = planes of W1 diag(gamma1), l1_cf / l1_df;  in_wpf = planes of Win diag(gamma2 of the PREVIOUS layer), in_cf / in_df
    char *l1_wpf, *in_wpf;
    float *l1_cf, *l1_df, *in_cf, *in_df;
};

}  // namespace rohm

struct rohm_posenet {
    int D, H, F, L, Cin, Cout, traj, device;
    int KP;            // padded K of the fused embed GEMM
    float* arena;      // one allocation holding every weight
    float* w_embed;    // [D, KP]   = [Wx | Wc | 0]
    int KX;            // padded K of the x_t half alone
    float* w_embed_x;  // [D, KX]   = [Wx | 0]: the per-step embed of the sampling loop, whose cond half is computed once per call
    bool cond_hoist;   // (default on; ROHM_POSENET_COND_HOIST=0: the full [x_t | cond] contraction every step)
    float* tab;        // [kMaxTok, D]  pe[tok] + bx + bc
    float* pe;         // [pe_len, D]
    int pe_len;
    float *t_w0T, *t_b0, *t_w2T, *t_b2;   // time MLP, weights stored [in][out]
    float* tok_table;  // [pe_len, D]  timestep token of every t (time MLP output + pe[0]), built once at create
    float *out_w, *out_b;                 // [Cout, D], [Cout]
    float *out_c;                         // LayerNorm folding of the last norm2 into the output head
    bool ln_fold;                         // LayerNorm folded into the CONSUMER GEMMs (opt-in, measured slower) or run as a kernel
    bool ln_fused;                        // LayerNorm inside the PRODUCER GEMMs (EPI_BIAS_RES_LN; default on, ROHM_POSENET_LN_FUSED=0: kernel)
    bool head_sk;                         // output head as a stream-K launch where the shape qualifies (default on, ROHM_POSENET_HEAD_SK=0: tiles)
    // The two launch forms above exchange data between workgroups of one launch (exchange.hip).  They are used only where the device
    // passed the layout guard at create (exch_allowed) and until an exchange failed on this handle (exch_fallback, set by
    // rohm_posenet_set_exchange: the Python loops then re-run the chunk on the exchange-free launches).
    bool chain_any;                       // chain at every batch size (tests: ROHM_POSENET_CHAIN_ANY=1)
    bool stack_front;                     // stack: input embedding + layer 0's in-projection as leading phases (ROHM_POSENET_STACK_FRONT=0: own launches)
    bool finish_pack;                     // sampling loop: DDPM update + the next step's pack as one kernel (ROHM_POSENET_FINISH_PACK=0: two)
    bool stack_tail;                      // sampling loop: output head + DDPM update + the next step's pack as the stack's closing phase -- ONE
                                          // launch per denoising step (single-round launches: B = 64 / 32; ROHM_POSENET_STACK_TAIL=0: head and
                                          // finish_pack as their own launches)
    int chain;                            // 0: one launch per GEMM; 1: the four GEMMs between two attention launches as ONE launch; 2 (default):
                                          // the whole encoder stack, attention included, as one launch (encoder_chain.hip; both need ln_fused;
                                          // ROHM_POSENET_CHAIN=0 | layer | stack)
    bool ln_fused_env, head_sk_env;       // what the environment asked for
    bool exch_allowed, exch_fallback;
    const char* exch_reason;              // why the guard refused (static string), or what it saw
    unsigned salt;                        // host part of this handle's launch tags
    mutable int fault_left;               // test hook (rohm_posenet_inject_exchange_fault): LayerNorm launches still to sabotage
    unsigned long long* stack_timeline;   // diagnostics (rohm_posenet_set_stack_timeline): phase stamps of the next stack launches, or null
    int nplane;                           // 0: exact fp32 MFMA (default); 3 / 2 / 16: split GEMMs on planes (bf16x6 / bf16x3 / fp16x3)
    char* wplanes;                        // one allocation holding the weight planes of every layer
    bool pp_fold;                         // plane modes: LayerNorm folded into the plane GEMMs (ROHM_PP_LNFOLD=1, two-plane modes)
    float* foldvec;                       // one allocation holding the fold vectors
    char* wplanes_fold;                   // ... and the planes of the gamma-scaled weights
    std::vector<rohm::LayerW> layers;
};

namespace rohm {

// ----------------------------------------------------------------------------- small kernels
// [B, C, T] (T contiguous) -> columns [col0, col0+C) of the token-major pack [B*S, KP], rows tok>=1.
// Also zeroes the tok = 0 row and (when zero_to > C) the pad columns [col0+C, col0+zero_to).
// `pass_ctr` (the x_t pack, i.e. the first kernel of a network pass): the workspace's pass counter, advanced by one thread -- the
// device part of the tags of this pass's exchanging launches (exchange.hip; every reader of the word is a LATER kernel of the stream).
__global__ __launch_bounds__(256) void pack_kernel(const float* __restrict__ src, float* __restrict__ dst,
                                                   int C, int T, int S, int KP, int col0, int zero_to, unsigned* pass_ctr) {
    __shared__ float tile[32][33];
    // [0] the pass counter, [1] the passes a preceding run of one-launch steps consumed beyond it (StackParams::pass_add: those launches
    // never write [0] -- their own late workgroups would read it -- they leave their count here for the next pass's first kernel)
    if (pass_ctr && blockIdx.x == 0 && blockIdx.y == 0 && blockIdx.z == 0 && threadIdx.x == 0) { *pass_ctr += 1u + pass_ctr[1]; pass_ctr[1] = 0u; }
    const int b = blockIdx.z;
    const int c0 = blockIdx.x * 32, t0 = blockIdx.y * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;   // 32 x 8
    const float* s = src + (size_t)b * C * T;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int c = c0 + ty + i * 8, t = t0 + tx;
        tile[ty + i * 8][tx] = (c < C && t < T) ? s[(size_t)c * T + t] : 0.f;
    }
    __syncthreads();
    float* d = dst + (size_t)b * S * KP + col0;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int t = t0 + ty + i * 8, c = c0 + tx;
        if (t < T && c < zero_to) d[(size_t)(t + 1) * KP + c] = tile[tx][ty + i * 8];
    }
    if (blockIdx.y == 0 && ty == 0) {
        const int c = c0 + tx;
        if (c < zero_to) d[c] = 0.f;
    }
}

// Timestep token: tab0[b] = W2 . silu(W0 . pe[t_b] + b0) + b2 + pe[0]   (heads.py:145-146, posenet.py:90-91)
// One block per row, D threads; weights are stored transposed ([in][out]) so reads coalesce.
__global__ __launch_bounds__(1024) void timestep_token_kernel(const float* __restrict__ pe, int pe_len,
                                                              const int64_t* __restrict__ t_dev, int64_t t_host,
                                                              const float* __restrict__ w0T,
                                                              const float* __restrict__ b0,
                                                              const float* __restrict__ w2T,
                                                              const float* __restrict__ b2,
                                                              float* __restrict__ tab0, int D) {
    extern __shared__ float sh[];   // [2*D]
    float* e = sh;
    float* h = sh + D;
    const int n = threadIdx.x;
    int64_t t = t_dev ? t_dev[blockIdx.x] : (t_host == -1 ? (int64_t)blockIdx.x : t_host);   // -1: row index = t (table build)
    if (t < 0) t = 0;
    if (t >= pe_len) t = pe_len - 1;
    e[n] = pe[(size_t)t * D + n];
    __syncthreads();
    float a = b0[n];
    for (int k = 0; k < D; ++k) a = fmaf(e[k], w0T[(size_t)k * D + n], a);
    h[n] = a / (1.0f + expf(-a));
    __syncthreads();
    float o = b2[n];
    for (int k = 0; k < D; ++k) o = fmaf(h[k], w2T[(size_t)k * D + n], o);
    tab0[(size_t)blockIdx.x * D + n] = o + pe[n];
}

// tab0[b] = tok_table[clamp(t_b)]: the timestep token of every possible t is computed ONCE at create (the time MLP only
// depends on t), so a forward with per-sample timesteps -- every guided step of the PROX / AMASS tails -- gathers B rows
// instead of running B x two 512x512 mat-vecs (77 us -> ~3 us per step at B = 32).
__global__ __launch_bounds__(256) void gather_tokens_kernel(const float* __restrict__ table, int rows,
                                                            const int64_t* __restrict__ t_dev,
                                                            float* __restrict__ tab0, int D) {
    int64_t t = t_dev[blockIdx.x];
    if (t < 0) t = 0;
    if (t >= rows) t = rows - 1;
    for (int n = threadIdx.x; n < D; n += blockDim.x) tab0[(size_t)blockIdx.x * D + n] = table[(size_t)t * D + n];
}

// x0[:, :traj] = cond[:, :traj] (posenet.py:94-95); optionally the DDPM update
// x_prev = c1*x0 + c2*x_t + sigma*noise (gaussian_diffusion_posenet.py:212-234,426-434).
__global__ __launch_bounds__(256) void finish_kernel(float* __restrict__ x0, const float* __restrict__ cond,
                                                     const float* x_t,
                                                     const float* __restrict__ noise, float* x_prev,
                                                     float c1, float c2, float sigma, int traj, int C, int T,
                                                     size_t n) {
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        const int c = (int)((i / T) % C);
        float v;
        if (c < traj) {
            v = cond[i];
            x0[i] = v;
        } else {
            v = x0[i];
        }
        if (x_prev) {
            float o = c1 * v + c2 * x_t[i];
            if (noise) o += sigma * noise[i];
            x_prev[i] = o;
        }
    }
}

// finish_kernel with the DDPM update AND pack_kernel of the next step in one pass (sampling loop, every step but the call's last):
// v = x0 (cond for the trajectory channels), x_prev = c1 v + c2 x_t + sigma noise is written back into x in place and, transposed
// through LDS, into the x_t half of the next step's token-major pack -- one read of x_t less and one launch less per step.  Also the
// first kernel of the NEXT pass: it advances the workspace's pass counter (see pack_kernel).
__global__ __launch_bounds__(256) void finish_pack_kernel(float* __restrict__ x0, const float* __restrict__ cond, float* x,
                                                          const float* __restrict__ noise, float* __restrict__ apack, float c1, float c2,
                                                          float sigma, int traj, int C, int T, int S, int KP, unsigned* pass_ctr) {
    __shared__ float tile[32][33];
    // [0] the pass counter, [1] the passes a preceding run of one-launch steps consumed beyond it (StackParams::pass_add: those launches
    // never write [0] -- their own late workgroups would read it -- they leave their count here for the next pass's first kernel)
    if (pass_ctr && blockIdx.x == 0 && blockIdx.y == 0 && blockIdx.z == 0 && threadIdx.x == 0) { *pass_ctr += 1u + pass_ctr[1]; pass_ctr[1] = 0u; }
    const int b = blockIdx.z;
    const int c0 = blockIdx.x * 32, t0 = blockIdx.y * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;   // 32 x 8
    const size_t base = (size_t)b * C * T;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int c = c0 + ty + i * 8, t = t0 + tx;
        float o = 0.f;
        if (c < C && t < T) {
            const size_t idx = base + (size_t)c * T + t;
            float v;
            if (c < traj) { v = cond[idx]; x0[idx] = v; }
            else v = x0[idx];
            o = c1 * v + c2 * x[idx];
            if (noise) o += sigma * noise[idx];
            x[idx] = o;
        }
        tile[ty + i * 8][tx] = o;
    }
    __syncthreads();
    float* d = apack + (size_t)b * S * KP;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int t = t0 + ty + i * 8, c = c0 + tx;
        if (t < T && c < C) d[(size_t)(t + 1) * KP + c] = tile[tx][ty + i * 8];
    }
}

__global__ void transpose_kernel(const float* __restrict__ src, float* __restrict__ dst, int R, int Cc) {
    // dst[c][r] = src[r][c]
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < R * Cc) {
        const int r = i / Cc, c = i % Cc;
        dst[(size_t)c * R + r] = src[i];
    }
}

__global__ void build_embed_kernel(const float* __restrict__ wx, const float* __restrict__ wc,
                                   float* __restrict__ w_embed, int D, int C, int KP) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < D * KP) {
        const int n = i / KP, k = i % KP;
        float v = 0.f;
        if (k < C) v = wx[(size_t)n * C + k];
        else if (wc && k < 2 * C) v = wc[(size_t)n * C + (k - C)];      // wc == null: [Wx | 0]
        w_embed[i] = v;
    }
}

__global__ void build_tab_kernel(const float* __restrict__ pe, const float* __restrict__ bx,
                                 const float* __restrict__ bc, float* __restrict__ tab, int rows, int D) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < rows * D) {
        const int n = i % D;
        tab[i] = pe[i] + (bx[n] + bc[n]);
    }
}

// LayerNorm folding, once at create: W[n][k] *= gamma[k];  c[n] = sum_k gamma[k] W0[n][k];
// bias[n] += sum_k beta[k] W0[n][k]  (W0 = the unscaled weight; float64 accumulation).  One block per row.
__global__ __launch_bounds__(256) void ln_fold_kernel(float* __restrict__ W, float* __restrict__ bias,
                                                      const float* __restrict__ gamma, const float* __restrict__ beta,
                                                      float* __restrict__ c, int K) {
    __shared__ double sc[256], sd[256];
    const int n = blockIdx.x;
    double ac = 0.0, ad = 0.0;
    for (int k = threadIdx.x; k < K; k += blockDim.x) {
        const float w = W[(size_t)n * K + k];
        ac += (double)gamma[k] * (double)w;
        ad += (double)beta[k] * (double)w;
        W[(size_t)n * K + k] = w * gamma[k];
    }
    sc[threadIdx.x] = ac; sd[threadIdx.x] = ad;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if (threadIdx.x < o) { sc[threadIdx.x] += sc[threadIdx.x + o]; sd[threadIdx.x] += sd[threadIdx.x + o]; }
        __syncthreads();
    }
    if (threadIdx.x == 0) { c[n] = (float)sc[0]; bias[n] = (float)((double)bias[n] + sd[0]); }
}

// ----------------------------------------------------------------------------- workspace
struct Workspace {
    float *apack, *h, *y, *qkv, *ctx, *ff, *tab0, *x0, *tok_all;
    char *hP, *yP, *ctxP, *ffP;   // planes of h / y / ctx / ff (split-bf16 mode; ctx and ff then exist as planes only)
    float *stats_a, *stats_b;     // row (sum, sum of squares) partials of y / h: [M][D/64][2]
    float* xln;                   // scratch of the LayerNorm-producing GEMMs (common.h gemm_ln_*): status words, statistics
    float* sk;                    // scratch of the stream-K output head (common.h gemm_sk_*): flags, partial tiles
    float* econd;                 // [M, D] cond half of the input embedding + positional table + biases (sampling loop)
    float* chain_flags;           // "my tile is stored" flags of the encoder chain (common.h ChainParams::flags)
    int64_t* t_all;
    size_t floats;
};

static inline size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

static Workspace carve(const rohm_posenet* p, int B, int T, float* base) {
    const size_t M = (size_t)B * (T + 1);
    Workspace w;
    size_t off = 0;
    auto take = [&](size_t n) {
        float* ptr = base ? base + off : nullptr;
        off += align_up(n, 64);
        return ptr;
    };
    w.apack = take(M * p->KP);
    w.h = take(M * p->D);
    w.y = take(M * p->D);
    w.qkv = take(M * 3 * p->D);
    if (p->nplane) {
        // M is a multiple of 16 only for whole clips of 144 tokens; other shapes run the fp32 path and need ctx / ff
        auto take_planes = [&](size_t cols) { return reinterpret_cast<char*>(take(plane_tensor_bytes((int)M, (int)cols, p->nplane) / 4)); };
        w.hP = take_planes(p->D); w.yP = take_planes(p->D); w.ctxP = take_planes(p->D); w.ffP = take_planes(p->F);
    } else {
        w.hP = w.yP = w.ctxP = w.ffP = nullptr;
    }
    w.ctx = take(M * p->D);
    w.ff = take(M * p->F);
    w.tab0 = take((size_t)B * p->D);
    w.x0 = take((size_t)B * p->Cin * T);
    w.tok_all = take((size_t)kLoopChunk * p->D);     // timestep tokens of one sample-loop call
    // row statistics: one (sum, sum of squares) pair per 64 columns (fp32 LayerNorm folding) or per 16 columns (plane modes)
    w.stats_a = take(M * (p->D / (p->nplane ? 16 : 64)) * 2);
    w.stats_b = take(M * (p->D / (p->nplane ? 16 : 64)) * 2);
    w.t_all = reinterpret_cast<int64_t*>(take(2 * (size_t)kLoopChunk));
    w.xln = take(gemm_ln_scratch_bytes((int)M, p->D) / sizeof(float));
    w.sk = take(gemm_sk_scratch_bytes() / sizeof(float));
    w.econd = take(M * p->D);
    w.chain_flags = take(encoder_chain_flag_bytes((int)M) / sizeof(float));
    w.floats = off;
    return w;
}

// The first words of w.xln are the workspace's exchange header (exchange.hip): [0] = error word of the in-kernel exchanges (LayerNorm
// statistics between column tiles, stream-K partials of the output head: a wait that ran into its bound, a partner on the wrong
// XCD), [1] = a magic that says the workspace has been armed, [2] = the pass counter.  Every entry point arms it -- on a workspace
// it sees for the first time that zeroes the error word, the counter, the statistics slots and the stream-K flags (torch.empty
// memory may hold anything) -- so an error stays until rohm_posenet_exchange_status() reads and clears it: no host synchronisation
// on the forward path, and no silent wrong statistics either.
static int arm_status(const rohm_posenet* p, const Workspace& w, int B, int T, hipStream_t s) {
    const size_t M = (size_t)B * (T + 1);
    return exchange_arm(reinterpret_cast<unsigned*>(w.xln), reinterpret_cast<char*>(w.xln) + 64, gemm_ln_scratch_bytes((int)M, p->D) - 64,
                        w.sk, 256 * sizeof(unsigned long long), false, s, w.chain_flags, encoder_chain_flag_bytes((int)M));
}
static inline unsigned* pass_counter(const Workspace& w) { return reinterpret_cast<unsigned*>(w.xln) + 2; }

static int check_shape(const rohm_posenet* p, int B, int T) {
    ROHM_ARG_CHECK(p != nullptr, "posenet: null handle");
    ROHM_ARG_CHECK(B > 0, "posenet: batch must be positive (got %d)", B);
    ROHM_ARG_CHECK(T + 1 <= kMaxTok && T + 1 <= p->pe_len, "posenet: sequence too long");
    return ROHM_OK;
}

static inline float qscale_of(const rohm_posenet* p) { return 1.0f / sqrtf((float)(p->D / p->H)); }

// What the sampling loop hands run_network so that a stacked launch can close the step itself (StackParams::tail, common.h).
struct TailArgs {
    float* x; const float* cond; const float* noise; float* x0; float* apack_next;
    float c1, c2, sigma;
    unsigned* pass_ctr;
    int step;      // index of the step within the call: the launch's tags use pass counter + step
};

// Network body: from packed input (w.apack complete) to x0 channels [traj, Cin) in `x0_out`.  With `tail` (sampling loop) and a launch
// plan that allows it -- the encoder stack with its leading phases, a single round of workgroups -- the stack launch also runs the
// output head, the DDPM update and the next step's pack; `*tail_ran` says whether it did (x0_out is then NOT written: tail->x0 is).
static int run_network(const rohm_posenet* p, const Workspace& w, const int64_t* t_dev, int64_t t_host,
                       const float* tok_pre, float* x0_out, int B, int T, hipStream_t s, bool cond_done = false,
                       const TailArgs* tail = nullptr, bool* tail_ran = nullptr) {
    if (tail_ran) *tail_ran = false;
    const int S = T + 1, D = p->D, M = B * S;
    // timestep token(s): per sample from device timesteps, or one precomputed row shared by the batch
    if (!tok_pre) {
        if (t_dev) {
            prof::Scope ps("gather_tokens", 0.0, 8.0 * D * B, s);
            hipLaunchKernelGGL(gather_tokens_kernel, dim3(B), dim3(256), 0, s, p->tok_table, p->pe_len, t_dev, w.tab0, D);
            ROHM_LAUNCH_CHECK();
        } else {
            int64_t t = t_host < 0 ? 0 : (t_host >= p->pe_len ? p->pe_len - 1 : t_host);
            tok_pre = p->tok_table + (size_t)t * D;
        }
    }
    int rc;
    const bool planes = p->nplane && S == 144 && D / p->H == 128;
    // the launch plan of the encoder (below): decided here because the stack also takes the input embedding and layer 0's in-projection
    const bool lnf_plan = p->ln_fused && !(p->ln_fold && !planes) && !planes && gemm_ln_supported(M, D, D) && gemm_ln_supported(M, D, p->F);
    const bool chained_plan = lnf_plan && p->chain && encoder_chain_parts(M, D, p->F) != 0 && 4 * p->L + 2 <= 60 && ((B >= 32 && encoder_chain_pays(B)) || p->chain_any);
    const bool stacked_plan = chained_plan && p->chain == 2 && p->H == 4 && p->L <= 8 && S == 144 && D / p->H == 128;
    GemmParams ge{};      // fused input embed (+cond embed, + biases, + positional table)
    ge.A = w.apack; ge.lda = p->KP; ge.W = p->w_embed; ge.ldw = p->KP; ge.C = w.h; ge.ldc = D;
    ge.M = M; ge.N = D; ge.K = p->KP; ge.S = S; ge.tab = p->tab; ge.tab0 = tok_pre ? tok_pre : w.tab0; ge.ldtab = D;
    ge.ldtab0 = (t_dev && !tok_pre) ? D : 0;
    if (cond_done) {      // w.econd = cond . Wc^T + bx + bc + pe[tok] already (embed_cond): contract x_t only and add it row by row
        ge.W = p->w_embed_x; ge.ldw = p->KX; ge.K = p->KX; ge.tab = w.econd; ge.tab_by_row = 1;
    }
    const bool front = stacked_plan && p->stack_front;      // the embedding and layer 0's in-projection as the stack's leading phases
    if (!front && (rc = launch_gemm(ge, EPI_EMBED, s))) return rc;
    float* h = w.h;
    float* y = w.y;
    if (planes) {
        // Split-bf16 mode: every producer hands its consumer bf16 planes (planes.h) -- LayerNorm writes fp32 (the residual)
        // AND planes, attention and the GELU GEMM write planes only; the embed output is cut by a small kernel (once per step).
        const int np = p->nplane;
        if ((rc = launch_plane_split(h, D, M, D, np, 1.0f, w.hP, s))) return rc;
        const float wsc = (np == kModeF16) ? 1.0f / kF16WeightScale : 0.f;
        const bool pf = p->pp_fold;       // LayerNorm folded into the GEMMs: h / y hold RAW (pre-norm) values from layer 0's out-projection on
        auto consumer = [&](PlaneGemmParams& g, const float* stats, const float* c, const float* d) {
            g.ln_stats = stats; g.ln_c = c; g.bias = d; g.ln_dim = D; g.ln_eps = 1e-5f;
        };
        for (int l = 0; l < p->L; ++l) {
            const LayerW& lw = p->layers[l];
            PlaneGemmParams g{};
            g.Ap = w.hP; g.Wp = lw.in_wp; g.C = w.qkv; g.ldc = 3 * D; g.M = M; g.N = 3 * D; g.K = D;
            g.bias = lw.in_b; g.qcols = D; g.qscale = 1.0f / sqrtf((float)(D / p->H)); g.acc_scale = wsc;
            if (pf && l > 0) { g.Wp = lw.in_wpf; consumer(g, w.stats_b, lw.in_cf, lw.in_df); }      // LN2 of layer l - 1 folded in
            if ((rc = launch_gemm_pp(g, EPI_QKV, np, s))) return rc;
            if ((rc = launch_attention_planes(w.qkv, w.ctxP, B, p->H, np, s))) return rc;
            g = PlaneGemmParams{};
            g.Ap = w.ctxP; g.Wp = lw.out_wp; g.C = y; g.ldc = D; g.M = M; g.N = D; g.K = D;
            g.bias = lw.out_b; g.R = h; g.ldr = D; g.acc_scale = wsc;
            if (pf) {
                g.ln_dim = D; g.ln_eps = 1e-5f;
                if (l > 0) { g.r_stats = w.stats_b; g.r_gamma = p->layers[l - 1].n2_w; g.r_beta = p->layers[l - 1].n2_b; }
                g.out_stats = w.stats_a; g.Cp = w.yP;          // raw y: fp32 (FF2's residual), planes (FF1's operand), statistics
            }
            if ((rc = launch_gemm_pp(g, EPI_BIAS_RES, np, s))) return rc;
            if (!pf && (rc = launch_layernorm_planes(y, lw.n1_w, lw.n1_b, M, D, np, w.yP, s))) return rc;
            g = PlaneGemmParams{};
            g.Ap = w.yP; g.Wp = lw.l1_wp; g.Cp = w.ffP; g.M = M; g.N = p->F; g.K = D; g.bias = lw.l1_b; g.acc_scale = wsc;
            if (pf) { g.Wp = lw.l1_wpf; consumer(g, w.stats_a, lw.l1_cf, lw.l1_df); }
            if ((rc = launch_gemm_pp(g, EPI_BIAS_GELU, np, s))) return rc;
            g = PlaneGemmParams{};
            g.Ap = w.ffP; g.Wp = lw.l2_wp; g.C = h; g.ldc = D; g.M = M; g.N = D; g.K = p->F;
            g.bias = lw.l2_b; g.R = y; g.ldr = D; g.acc_scale = wsc;
            if (pf) {
                g.ln_dim = D; g.ln_eps = 1e-5f;
                g.r_stats = w.stats_a; g.r_gamma = lw.n1_w; g.r_beta = lw.n1_b;      // the residual is LN1(raw y)
                g.out_stats = w.stats_b; g.Cp = w.hP;
            }
            if ((rc = launch_gemm_pp(g, EPI_BIAS_RES, np, s))) return rc;
            if (!pf && (rc = launch_layernorm_planes(h, lw.n2_w, lw.n2_b, M, D, np, w.hP, s))) return rc;
        }
        // the last norm2 has no GEMM of this kind behind it: one LayerNorm launch in front of the output head
        if (pf && (rc = launch_layernorm(h, p->layers[p->L - 1].n2_w, p->layers[p->L - 1].n2_b, M, D, s))) return rc;
    }
    const bool fold = p->ln_fold && !planes;
    // LayerNorm inside the producer GEMMs (out-projection, FF2): whole clips of 144 tokens, widths whose column tiles pair up
    // (not while the stream records a hipGraph: the exchange's per-launch tag would be replayed -- the GEMM + LayerNorm pair then)
    // (a stream that is recording a hipGraph keeps them: their tags come from the workspace's pass counter, a replay draws new ones)
    const bool lnf = p->ln_fused && !fold && !planes && gemm_ln_supported(M, D, D) && gemm_ln_supported(M, D, p->F);
    // tag of exchanging launch number `idx` of this pass (2 l, 2 l + 1: the LayerNorm GEMMs of layer l; 60: the output head)
    auto tag_launch = [&](GemmParams& g, int idx) { g.xln_epoch = p->salt + (unsigned)idx; };
    const int parts = D / 64;
    auto ln_operand = [&](GemmParams& g, const float* stats, const float* c) {
        g.ln_stats = stats; g.ln_parts = parts; g.ln_c = c; g.ln_dim = D; g.ln_eps = 1e-5f;
    };
    auto ln_residual = [&](GemmParams& g, const float* stats, const float* gamma, const float* beta) {
        g.r_stats = stats; g.r_parts = parts; g.r_gamma = gamma; g.r_beta = beta; g.ln_dim = D; g.ln_eps = 1e-5f;
    };
    // The four GEMMs between two attention launches as ONE launch (encoder_chain.hip): needs the in-kernel LayerNorm exchange (lnf) and
    // the released widths.  Layer l: [QKV of layer 0: its own launch] attention(l), chain(l) = out-proj + norm1, FF1, FF2 + norm2 and
    // the QKV projection of layer l + 1.  Tags: 4 l, 4 l + 1 (the chain's two LayerNorm exchanges and its flags); the head: 60.
    // From 32 clips on where whole rounds of its persistent workgroups fit the batch (encoder_chain_pays: 32, 48 .. 64, 122 .. 128, ...):
    // elsewhere the launch-per-GEMM path picks a tile width per GEMM and keeps more CUs busy (ROHM_POSENET_CHAIN_ANY=1 chains every
    // shape that has the form: tests).
    const bool chained = lnf && p->chain && encoder_chain_parts(M, D, p->F) != 0 && 4 * p->L + 2 <= 60 && ((B >= 32 && encoder_chain_pays(B)) || p->chain_any);
    const bool stacked = stacked_plan;
    if (stacked != (chained && p->chain == 2 && p->H == 4 && p->L <= 8 && S == 144 && D / p->H == 128)) {
        set_error("posenet: inconsistent launch plan");      // the two derivations of the plan must agree
        return ROHM_ERR_ARG;
    }
    if (stacked) {
        // ... and with attention inside, the layers looped in the kernel, the input embedding and layer 0's in-projection as leading
        // phases: ONE launch from the packed input to the encoder's output
        StackParams c{};
        if (front) {
            c.front = 1;
            c.apack = ge.A; c.lda_pack = ge.lda; c.w_embed = ge.W; c.ldw_embed = ge.ldw; c.k_embed = ge.K;
            c.S = S; c.tab = ge.tab; c.tab0 = ge.tab0; c.ldtab = ge.ldtab; c.ldtab0 = ge.ldtab0; c.tab_by_row = ge.tab_by_row;
        } else {      // ROHM_POSENET_STACK_FRONT=0: the round-5 first form, embed and QKV of layer 0 as their own launches
            const LayerW& l0 = p->layers[0];
            GemmParams g{};
            g.A = h; g.lda = D; g.W = l0.in_w; g.ldw = D; g.C = w.qkv; g.ldc = 3 * D; g.M = M; g.N = 3 * D; g.K = D;
            g.bias = l0.in_b; g.qcols = D; g.qscale = qscale_of(p);
            if ((rc = launch_gemm(g, EPI_QKV, s))) return rc;
        }
        c.h = h; c.y = y; c.ff = w.ff; c.qkv = w.qkv; c.ctx = w.ctx;
        c.M = M; c.D = D; c.F = p->F; c.L = p->L; c.n_head = p->H; c.qscale = qscale_of(p); c.ln_eps = 1e-5f;
        for (int l = 0; l < p->L; ++l) {
            const LayerW& lw = p->layers[l];
            c.layer[l] = StackLayerW{lw.out_w, lw.out_b, lw.n1_w, lw.n1_b, lw.l1_w, lw.l1_b, lw.l2_w, lw.l2_b, lw.n2_w, lw.n2_b, lw.in_w, lw.in_b};
        }
        {
            GemmParams b{};
            b.M = M;
            gemm_ln_bind(b, w.xln);
            c.xln_stats = b.xln_stats; c.xln_err = b.xln_err; c.xln_pass = b.xln_pass; c.xln_xcc = b.xln_xcc;
        }
        c.epoch = p->salt;
        c.flags = reinterpret_cast<unsigned long long*>(w.chain_flags);
        if (p->fault_left > 0) { --p->fault_left; c.fault = 1; }
        c.timeline = p->stack_timeline;
        {
            const int Gp = encoder_chain_parts(M, D, p->F), groups8 = (B + kNumXCD - 1) / kNumXCD * kNumXCD;
            (void)Gp; (void)groups8;
            if (tail && tail_ran && front && p->stack_tail && D == 512 && p->Cout == 272 && !fold) {
                c.tail = 1;
                c.t_out_w = p->out_w; c.t_out_b = p->out_b;
                c.t_x = tail->x; c.t_cond = tail->cond; c.t_noise = tail->sigma == 0.f ? nullptr : tail->noise;
                c.t_x0 = tail->x0; c.t_apack = tail->apack_next;
                c.t_c1 = tail->c1; c.t_c2 = tail->c2; c.t_sigma = tail->sigma;
                c.t_traj = p->traj; c.t_C = p->Cin; c.t_T = T; c.t_lda = p->KP; c.t_pass_ctr = tail->pass_ctr;
                c.pass_add = (unsigned)tail->step;
                *tail_ran = true;
            }
        }
        if ((rc = launch_encoder_stack(c, s))) return rc;
        if (c.tail) return ROHM_OK;      // the launch ended with x_prev in x (and the next step's pack): nothing left of the step
    }
    const float qscale = qscale_of(p);
    for (int l = 0; l < ((planes || !chained || stacked) ? 0 : p->L); ++l) {
        const LayerW& lw = p->layers[l];
        if (l == 0) {
            GemmParams g{};
            g.A = h; g.lda = D; g.W = lw.in_w; g.ldw = D; g.C = w.qkv; g.ldc = 3 * D; g.M = M; g.N = 3 * D; g.K = D;
            g.bias = lw.in_b; g.qcols = D; g.qscale = qscale;
            if ((rc = launch_gemm(g, EPI_QKV, s))) return rc;
        }
        if ((rc = launch_attention(w.qkv, w.ctx, B, p->H, S, D / p->H, s))) return rc;
        ChainParams c{};
        c.ctx = w.ctx; c.h = h; c.y = y; c.ff = w.ff; c.qkv = (l + 1 < p->L) ? w.qkv : nullptr;
        c.M = M; c.D = D; c.F = p->F;
        c.out_w = lw.out_w; c.out_b = lw.out_b; c.n1_w = lw.n1_w; c.n1_b = lw.n1_b; c.l1_w = lw.l1_w; c.l1_b = lw.l1_b;
        c.l2_w = lw.l2_w; c.l2_b = lw.l2_b; c.n2_w = lw.n2_w; c.n2_b = lw.n2_b;
        if (l + 1 < p->L) { c.in_w = p->layers[l + 1].in_w; c.in_b = p->layers[l + 1].in_b; }
        c.qscale = qscale; c.ln_eps = 1e-5f;
        {
            GemmParams b{};
            b.M = M;
            gemm_ln_bind(b, w.xln);
            c.xln_stats = b.xln_stats; c.xln_err = b.xln_err; c.xln_pass = b.xln_pass; c.xln_xcc = b.xln_xcc;
        }
        c.epoch = p->salt + (unsigned)(4 * l);
        c.flags = reinterpret_cast<unsigned long long*>(w.chain_flags);
        if (p->fault_left > 0) { --p->fault_left; c.fault = 1; }
        if ((rc = launch_encoder_chain(c, s))) return rc;
    }
    for (int l = 0; l < ((planes || chained) ? 0 : p->L); ++l) {
        const LayerW& lw = p->layers[l];
        // With folding, h holds the RAW (pre-norm2) output of the previous layer for l >= 1 and stats_b its row sums.
        GemmParams g{};
        g.A = h; g.lda = D; g.W = lw.in_w; g.ldw = D; g.C = w.qkv; g.ldc = 3 * D; g.M = M; g.N = 3 * D; g.K = D;
        g.bias = lw.in_b; g.qcols = D; g.qscale = 1.0f / sqrtf((float)(D / p->H));
        if (fold && l > 0) ln_operand(g, w.stats_b, lw.in_c);
        if ((rc = launch_gemm(g, EPI_QKV, s))) return rc;
        if ((rc = launch_attention(w.qkv, w.ctx, B, p->H, S, D / p->H, s))) return rc;
        g = GemmParams{};
        g.A = w.ctx; g.lda = D; g.W = lw.out_w; g.ldw = D; g.C = y; g.ldc = D; g.M = M; g.N = D; g.K = D;
        g.bias = lw.out_b; g.R = h; g.ldr = D;
        if (fold) {
            if (l > 0) ln_residual(g, w.stats_b, p->layers[l - 1].n2_w, p->layers[l - 1].n2_b);
            g.out_stats = w.stats_a; g.out_parts = parts;
        }
        if (lnf) {      // y = norm1(h + out_proj(ctx)) in one launch
            g.ln_gamma = lw.n1_w; g.ln_beta = lw.n1_b; g.ln_dim = D; g.ln_eps = 1e-5f;
            gemm_ln_bind(g, w.xln);
            tag_launch(g, 2 * l);
            if (p->fault_left > 0) { --p->fault_left; g.xln_fault = 1; }
            if ((rc = launch_gemm(g, EPI_BIAS_RES_LN, s))) return rc;
        } else {
            if ((rc = launch_gemm(g, EPI_BIAS_RES, s))) return rc;
            if (!fold && (rc = launch_layernorm(y, lw.n1_w, lw.n1_b, M, D, s))) return rc;
        }
        g = GemmParams{};
        g.A = y; g.lda = D; g.W = lw.l1_w; g.ldw = D; g.C = w.ff; g.ldc = p->F; g.M = M; g.N = p->F; g.K = D;
        g.bias = lw.l1_b;
        if (fold) ln_operand(g, w.stats_a, lw.l1_c);
        if ((rc = launch_gemm(g, EPI_BIAS_GELU, s))) return rc;
        g = GemmParams{};
        g.A = w.ff; g.lda = p->F; g.W = lw.l2_w; g.ldw = p->F; g.C = h; g.ldc = D; g.M = M; g.N = D; g.K = p->F;
        g.bias = lw.l2_b; g.R = y; g.ldr = D;
        if (fold) {
            ln_residual(g, w.stats_a, lw.n1_w, lw.n1_b);
            g.out_stats = w.stats_b; g.out_parts = parts;
        }
        if (lnf) {      // h = norm2(y + linear2(gelu(linear1(y))))
            g.ln_gamma = lw.n2_w; g.ln_beta = lw.n2_b; g.ln_dim = D; g.ln_eps = 1e-5f;
            gemm_ln_bind(g, w.xln);
            tag_launch(g, 2 * l + 1);
            if (p->fault_left > 0) { --p->fault_left; g.xln_fault = 1; }
            if ((rc = launch_gemm(g, EPI_BIAS_RES_LN, s))) return rc;
        } else {
            if ((rc = launch_gemm(g, EPI_BIAS_RES, s))) return rc;
            if (!fold && (rc = launch_layernorm(h, lw.n2_w, lw.n2_b, M, D, s))) return rc;
        }
    }
    {   // output head, transposed: rows = channels, cols = tokens
        GemmParams g{};
        g.A = p->out_w; g.lda = D; g.W = h; g.ldw = D; g.C = x0_out; g.M = p->Cout; g.N = M; g.K = D;
        g.bias = p->out_b; g.S = S; g.ch_off = p->Cin - p->Cout; g.C_total = p->Cin; g.T = T;
        if (fold) ln_operand(g, w.stats_b, p->out_c);
        // B = 64: 2 x 144 tiles on 256 CUs -- dealt out as (tile, K chunk) units instead of a second, 1/8-full round (common.h sk_*)
        if (p->head_sk) { gemm_sk_bind(g, w.sk, reinterpret_cast<unsigned*>(w.xln)); tag_launch(g, 60); }
        if ((rc = launch_gemm(g, EPI_OUT_T, s))) return rc;
    }
    return ROHM_OK;
}

static int launch_pack(const rohm_posenet* p, const float* src, float* apack, int B, int T, int which,
                       hipStream_t s, unsigned* pass_ctr = nullptr) {
    const int C = p->Cin, S = T + 1;
    const int zero_to = which == 0 ? C : (p->KP - C);   // second half also clears the K padding
    dim3 grid((zero_to + 31) / 32, (T + 31) / 32, B);
    prof::Scope ps("pack", 0.0, 8.0 * B * C * T, s);
    hipLaunchKernelGGL(pack_kernel, grid, dim3(256), 0, s, src, apack, C, T, S, p->KP, which * C, zero_to, pass_ctr);
    ROHM_LAUNCH_CHECK();
    return ROHM_OK;
}

static int launch_finish(float* x0, const float* cond, const float* x_t, const float* noise, float* x_prev,
                         float c1, float c2, float sigma, int traj, int C, int T, size_t n, hipStream_t s) {
    size_t blocks = (n + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    if (sigma == 0.f) noise = nullptr;
    prof::Scope ps("finish_ddpm", 0.0, 4.0 * n * (x_prev ? 4 : 1), s);
    hipLaunchKernelGGL(finish_kernel, dim3((unsigned)blocks), dim3(256), 0, s, x0, cond, x_t, noise, x_prev, c1, c2,
                       sigma, traj, C, T, n);
    ROHM_LAUNCH_CHECK();
    return ROHM_OK;
}

static int launch_finish_pack(const rohm_posenet* p, float* x0, const float* cond, float* x, const float* noise, float* apack, float c1,
                              float c2, float sigma, int B, int T, unsigned* pass_ctr, hipStream_t s) {
    const int C = p->Cin;
    if (sigma == 0.f) noise = nullptr;
    dim3 grid((C + 31) / 32, (T + 31) / 32, B);
    prof::Scope ps("finish_pack", 0.0, 4.0 * (double)B * C * T * (noise ? 6 : 5), s);
    hipLaunchKernelGGL(finish_pack_kernel, grid, dim3(256), 0, s, x0, cond, x, noise, apack, c1, c2, sigma, p->traj, C, T, T + 1, p->KP, pass_ctr);
    ROHM_LAUNCH_CHECK();
    return ROHM_OK;
}

}  // namespace rohm

using namespace rohm;

extern "C" {

int rohm_posenet_create(rohm_posenet_t** out, const rohm_posenet_weights* w, int d_model, int n_head, int d_ff,
                        int n_layer, int c_in, int c_out, int traj_dim, int device) {
    ROHM_ARG_CHECK(out && w && w->layers, "posenet_create: null argument");
    ROHM_ARG_CHECK(d_model % 256 == 0 && d_model <= 1024, "posenet_create: d_model must be 256/512/1024");
    ROHM_ARG_CHECK(n_head > 0 && d_model % n_head == 0 && (d_model / n_head == 128 || d_model / n_head == 64),
                   "posenet_create: head dim must be 64 or 128 (d_model=%d, n_head=%d)", d_model, n_head);
    ROHM_ARG_CHECK(d_ff % 64 == 0 && n_layer > 0, "posenet_create: bad d_ff/n_layer");
    ROHM_ARG_CHECK(c_out + traj_dim == c_in, "posenet_create: c_out + traj_dim must equal c_in");
    ROHM_ARG_CHECK(w->pe_len >= kMaxTok, "posenet_create: positional table too short");
    ROHM_HIP_CHECK(hipSetDevice(device));
    rohm_posenet* p = new rohm_posenet();
    p->D = d_model; p->H = n_head; p->F = d_ff; p->L = n_layer; p->Cin = c_in; p->Cout = c_out;
    p->traj = traj_dim; p->device = device; p->pe_len = w->pe_len;
    p->KP = (int)align_up((size_t)2 * c_in, 32);
    const size_t D = d_model, F = d_ff;
    size_t total = 0;
    auto cnt = [&](size_t n) { size_t o = total; total += align_up(n, 64); return o; };
    p->KX = (int)align_up((size_t)c_in, 32);
    const size_t o_embed = cnt(D * p->KP), o_tab = cnt((size_t)kMaxTok * D), o_pe = cnt((size_t)w->pe_len * D);
    const size_t o_embed_x = cnt(D * p->KX);
    const size_t o_tok = cnt((size_t)w->pe_len * D);
    const size_t o_w0 = cnt(D * D), o_b0 = cnt(D), o_w2 = cnt(D * D), o_b2 = cnt(D);
    const size_t o_ow = cnt((size_t)c_out * D), o_ob = cnt(c_out), o_oc = cnt(c_out);
    const size_t o_tmp = cnt(D * D);                      // staging for transposes / embed build
    const size_t o_tmp2 = cnt(2 * D * (size_t)c_in + 2 * D);
    const size_t per_layer = align_up(3 * D * D, 64) + align_up(3 * D, 64) + align_up(D * D, 64) + align_up(D, 64) +
                             align_up(F * D, 64) + align_up(F, 64) + align_up(D * F, 64) + align_up(D, 64) +
                             4 * align_up(D, 64) + align_up(3 * D, 64) + align_up(F, 64);
    const size_t o_layers = cnt(per_layer * n_layer);
    hipError_t e = hipMalloc(&p->arena, total * sizeof(float));
    if (e != hipSuccess) {
        delete p;
        set_error("posenet_create: hipMalloc(%zu) failed: %s", total * sizeof(float), hipGetErrorString(e));
        return ROHM_ERR_HIP;
    }
    float* a = p->arena;
    auto put = [&](float* dst, const float* src, size_t n) {
        return hipMemcpy(dst, src, n * sizeof(float), hipMemcpyDefault);
    };
#define PUT(dst, src, n)                                                                \
    do {                                                                                \
        hipError_t _e = put(dst, src, n);                                               \
        if (_e != hipSuccess) {                                                         \
            set_error("posenet_create: copy of %s failed: %s", #src, hipGetErrorString(_e)); \
            (void)hipFree(p->arena);                                                          \
            delete p;                                                                   \
            return ROHM_ERR_HIP;                                                        \
        }                                                                               \
    } while (0)
    p->w_embed = a + o_embed; p->w_embed_x = a + o_embed_x; p->tab = a + o_tab; p->pe = a + o_pe; p->tok_table = a + o_tok;
    p->t_w0T = a + o_w0; p->t_b0 = a + o_b0; p->t_w2T = a + o_w2; p->t_b2 = a + o_b2;
    p->out_w = a + o_ow; p->out_b = a + o_ob; p->out_c = a + o_oc;
    {
        // LayerNorm folded into the surrounding GEMMs (no LN launches, no extra pass over the residual stream): built,
        // parity-tested and MEASURED SLOWER than the separate 7.8 us kernel (B = 64: 19.38 vs 19.41 clips/s, B = 32:
        // 16.38 vs 16.73, B = 8: 5.59 vs 5.86 -- the statistics exchange and the longer epilogues sit on every
        // tile's critical path, the LN kernel overlaps nothing but costs little energy on a power-limited chip).
        // Opt-in for further tuning: ROHM_POSENET_LNFOLD=1.
        const char* e2 = getenv("ROHM_POSENET_LNFOLD");
        p->ln_fold = (e2 && atoi(e2) == 1) && d_model <= 512;       // 8 statistic slots of 64 columns
        // Round 4: LayerNorm inside the PRODUCER (out-projection / FF2) instead -- the column tiles of a row tile exchange their
        // row statistics through L2 while they run and store LN(x) once (gemm_f32.hip EPI_BIAS_RES_LN): 16 launches and one
        // write + read of the residual stream per layer less.  Default on; ROHM_POSENET_LN_FUSED=0 keeps the LayerNorm kernel.
        const char* e6 = getenv("ROHM_POSENET_LN_FUSED");
        p->ln_fused = !(e6 && e6[0] == '0');
        const char* e7 = getenv("ROHM_POSENET_HEAD_SK");
        p->head_sk = !(e7 && e7[0] == '0');
        // Both launch forms assume a whole MI355X (256 CUs free for one launch, block b on XCD b % 8); the tags leave 6 bits for the
        // launch index of a pass.  A device that does not look like that -- partitioned, CU-masked, shared -- gets the GEMM +
        // LayerNorm kernel pair and plain output-head tiles from the start (exchange.hip: properties + environment + a probe launch).
        const char* e9 = getenv("ROHM_POSENET_CHAIN");
        p->chain = 2;
        if (e9 && (e9[0] == '0')) p->chain = 0;
        else if (e9 && (!strcmp(e9, "layer") || !strcmp(e9, "1"))) p->chain = 1;
        else if (e9 && *e9 && strcmp(e9, "stack") && strcmp(e9, "2")) {
            set_error("posenet_create: ROHM_POSENET_CHAIN must be 0, layer or stack (got '%s')", e9);
            (void)hipFree(p->arena);
            delete p;
            return ROHM_ERR_ARG;
        }
        const char* e10 = getenv("ROHM_POSENET_CHAIN_ANY");
        p->chain_any = e10 && e10[0] == '1';
        const char* e11 = getenv("ROHM_POSENET_STACK_FRONT");
        p->stack_front = !(e11 && e11[0] == '0');
        const char* e12 = getenv("ROHM_POSENET_FINISH_PACK");
        p->finish_pack = !(e12 && e12[0] == '0');
        const char* e13 = getenv("ROHM_POSENET_STACK_TAIL");
        p->stack_tail = !(e13 && e13[0] == '0');
        p->ln_fused_env = p->ln_fused; p->head_sk_env = p->head_sk;
        p->exch_fallback = false;
        p->fault_left = 0;
        p->stack_timeline = nullptr;
        p->exch_reason = "not asked for";
        p->exch_allowed = (p->ln_fused || p->head_sk) && 2 * n_layer + 1 <= 60 && exchange_layout_ok(device, &p->exch_reason);
        if (!p->exch_allowed) p->ln_fused = p->head_sk = false;
        {      // per-handle salt of the launch tags: a recycled workspace that holds another handle's (or anybody's) old words is stale
            static std::atomic<unsigned> counter{0};
            unsigned v = (unsigned)(uintptr_t)p ^ (unsigned)((uintptr_t)p >> 32) ^ ((counter.fetch_add(1u) + 1u) * 0x9e3779b9u);
            v ^= v >> 16; v *= 0x7feb352du; v ^= v >> 15; v *= 0x846ca68bu; v ^= v >> 16;
            p->salt = v & ~63u;
        }
        // The cond half of the input embedding (InputProcess of batch['cond'], model/posenet.py:85-87) does not change over a sampling
        // loop: rohm_posenet_sample_loop computes it once per call and each step contracts the x_t half only (K 608 -> 320).
        const char* e8 = getenv("ROHM_POSENET_COND_HOIST");
        p->cond_hoist = !(e8 && e8[0] == '0');
        // Opt-in precision ladder: ROHM_GEMM_PRECISION=bf16x6 | bf16x3 | fp16x3 runs the four Linears of every
        // encoder layer as split-bf16 GEMMs on planes (gemm_pp.hip).  The default -- and every headline number -- is exact fp32.
        const char* e3 = getenv("ROHM_GEMM_PRECISION");
        p->nplane = 0;
        if (e3 && !strcmp(e3, "bf16x6")) p->nplane = 3;
        else if (e3 && !strcmp(e3, "bf16x3")) p->nplane = 2;
        else if (e3 && !strcmp(e3, "fp16x3")) p->nplane = kModeF16;
        else if (e3 && *e3 && strcmp(e3, "fp32")) {
            set_error("posenet_create: ROHM_GEMM_PRECISION must be fp32, bf16x6, bf16x3 or fp16x3 (got '%s')", e3);
            (void)hipFree(p->arena);
            delete p;
            return ROHM_ERR_ARG;
        }
        if (d_model / n_head != 128 || d_model % 64 || d_ff % 64) p->nplane = 0;   // shapes the plane kernels do not cover
        // ... and the widths their LayerNorm / fold forms exist for: launch_layernorm_planes knows D = 256 / 512 / 1024, and with the
        // fold the QKV GEMM's dynamic LDS request passes the 160 KiB of a CU from d_model = 1024 on.  Such a handle runs exact
        // fp32 (rohm_posenet_precision reports 0) instead of failing on every forward.
        if (p->nplane && d_model != 256 && d_model != 512) p->nplane = 0;
        if (p->nplane) p->ln_fold = false;
        p->wplanes = nullptr;
        p->foldvec = nullptr;
        p->wplanes_fold = nullptr;
        // LayerNorm folded into the plane GEMMs (fifteen of the sixteen LayerNorm launches of a step disappear): two-plane modes
        // only (the statistics tiles need the LDS a third plane occupies).  ROHM_PP_LNFOLD=0 switches it off.
        const char* e5 = getenv("ROHM_PP_LNFOLD");
        p->pp_fold = (p->nplane == 2 || p->nplane == kModeF16) && !(e5 && e5[0] == '0') && d_model % 128 == 0;
    }
    float* tmp = a + o_tmp;
    float* tmp2 = a + o_tmp2;
    PUT(p->pe, w->pe, (size_t)w->pe_len * D);
    PUT(p->t_b0, w->t_b0, D);
    PUT(p->t_b2, w->t_b2, D);
    PUT(p->out_w, w->out_w, (size_t)c_out * D);
    PUT(p->out_b, w->out_b, c_out);
    const int th = 256;
    PUT(tmp, w->t_w0, D * D);
    hipLaunchKernelGGL(transpose_kernel, dim3((unsigned)((D * D + th - 1) / th)), dim3(th), 0, 0, tmp, p->t_w0T, (int)D, (int)D);
    ROHM_HIP_CHECK(hipDeviceSynchronize());
    PUT(tmp, w->t_w2, D * D);
    hipLaunchKernelGGL(transpose_kernel, dim3((unsigned)((D * D + th - 1) / th)), dim3(th), 0, 0, tmp, p->t_w2T, (int)D, (int)D);
    ROHM_HIP_CHECK(hipDeviceSynchronize());
    float* wx = tmp2; float* wc = tmp2 + D * c_in; float* bx = wc + D * c_in; float* bc = bx + D;
    PUT(wx, w->in_x_w, D * c_in);
    PUT(wc, w->in_c_w, D * c_in);
    PUT(bx, w->in_x_b, D);
    PUT(bc, w->in_c_b, D);
    hipLaunchKernelGGL(build_embed_kernel, dim3((unsigned)((D * p->KP + th - 1) / th)), dim3(th), 0, 0, wx, wc, p->w_embed, (int)D, c_in, p->KP);
    hipLaunchKernelGGL(build_embed_kernel, dim3((unsigned)((D * p->KX + th - 1) / th)), dim3(th), 0, 0, wx, (const float*)nullptr, p->w_embed_x, (int)D, c_in, p->KX);
    hipLaunchKernelGGL(build_tab_kernel, dim3((unsigned)((kMaxTok * D + th - 1) / th)), dim3(th), 0, 0, p->pe, bx, bc, p->tab, kMaxTok, (int)D);
    // timestep tokens of every t in [0, pe_len)
    hipLaunchKernelGGL(timestep_token_kernel, dim3((unsigned)w->pe_len), dim3((unsigned)D), 2 * D * sizeof(float), 0, p->pe,
                       p->pe_len, (const int64_t*)nullptr, (int64_t)-1, p->t_w0T, p->t_b0, p->t_w2T, p->t_b2, p->tok_table, (int)D);
    ROHM_HIP_CHECK(hipDeviceSynchronize());
    float* lp = a + o_layers;
    p->layers.resize(n_layer);
    for (int l = 0; l < n_layer; ++l) {
        const rohm_posenet_layer_weights& s = w->layers[l];
        LayerW& d = p->layers[l];
        auto take = [&](size_t n) { float* r = lp; lp += align_up(n, 64); return r; };
        d.in_w = take(3 * D * D); d.in_b = take(3 * D); d.out_w = take(D * D); d.out_b = take(D);
        d.l1_w = take(F * D); d.l1_b = take(F); d.l2_w = take(D * F); d.l2_b = take(D);
        d.n1_w = take(D); d.n1_b = take(D); d.n2_w = take(D); d.n2_b = take(D);
        d.in_c = take(3 * D); d.l1_c = take(F);
        PUT(d.in_w, s.in_proj_w, 3 * D * D); PUT(d.in_b, s.in_proj_b, 3 * D);
        PUT(d.out_w, s.out_proj_w, D * D); PUT(d.out_b, s.out_proj_b, D);
        PUT(d.l1_w, s.lin1_w, F * D); PUT(d.l1_b, s.lin1_b, F);
        PUT(d.l2_w, s.lin2_w, D * F); PUT(d.l2_b, s.lin2_b, D);
        PUT(d.n1_w, s.norm1_w, D); PUT(d.n1_b, s.norm1_b, D);
        PUT(d.n2_w, s.norm2_w, D); PUT(d.n2_b, s.norm2_b, D);
    }
#undef PUT
    ROHM_HIP_CHECK(hipDeviceSynchronize());
    for (auto& d : p->layers) {
        d.in_wp = d.out_wp = d.l1_wp = d.l2_wp = d.l1_wpf = d.in_wpf = nullptr;
        d.l1_cf = d.l1_df = d.in_cf = d.in_df = nullptr;
    }
    if (p->nplane) {
        const size_t b_in = plane_tensor_bytes(3 * d_model, d_model, p->nplane), b_out = plane_tensor_bytes(d_model, d_model, p->nplane),
                     b_l1 = plane_tensor_bytes(d_ff, d_model, p->nplane), b_l2 = plane_tensor_bytes(d_model, d_ff, p->nplane);
        hipError_t e4 = hipMalloc(&p->wplanes, (b_in + b_out + b_l1 + b_l2) * n_layer);
        if (e4 != hipSuccess) {
            set_error("posenet_create: hipMalloc of the weight planes failed: %s", hipGetErrorString(e4));
            (void)hipFree(p->arena);
            delete p;
            return ROHM_ERR_HIP;
        }
        char* wp = p->wplanes;
        int rc4 = ROHM_OK;
        for (int l = 0; l < n_layer && rc4 == ROHM_OK; ++l) {
            LayerW& d = p->layers[l];
            d.in_wp = wp; wp += b_in; d.out_wp = wp; wp += b_out; d.l1_wp = wp; wp += b_l1; d.l2_wp = wp; wp += b_l2;
            // fp16 planes: the weights are cut from 2^8 w (Linear weights are ~1e-2: lifted clear of fp16's 6e-5 underflow
            // threshold, and |w| < 255 still fits); the GEMMs multiply their accumulators by 2^-8
            const float ws = (p->nplane == kModeF16) ? kF16WeightScale : 1.0f;
            rc4 = launch_plane_split(d.in_w, d_model, 3 * d_model, d_model, p->nplane, ws, d.in_wp, 0);
            if (!rc4) rc4 = launch_plane_split(d.out_w, d_model, d_model, d_model, p->nplane, ws, d.out_wp, 0);
            if (!rc4) rc4 = launch_plane_split(d.l1_w, d_model, d_ff, d_model, p->nplane, ws, d.l1_wp, 0);
            if (!rc4) rc4 = launch_plane_split(d.l2_w, d_ff, d_model, d_ff, p->nplane, ws, d.l2_wp, 0);
        }
        if (rc4 == ROHM_OK && p->pp_fold) {
            // gamma-scaled copies of W1 (every layer) and Win (layers >= 1, with the previous layer's norm2) are cut into their
            // own planes; c_n = sum_k gamma_k W_nk and d_n = b_n + sum_k beta_k W_nk come out of the same kernel (fp64 sums)
            const size_t b_l1f = plane_tensor_bytes(d_ff, d_model, p->nplane), b_inf = plane_tensor_bytes(3 * d_model, d_model, p->nplane);
            char* fp = nullptr;
            float *tmpw = nullptr;
            const size_t vec_floats = (size_t)n_layer * (2 * F + 2 * 3 * D);
            if (hipMalloc(&fp, (b_l1f + b_inf) * n_layer) != hipSuccess || hipMalloc(&p->foldvec, vec_floats * sizeof(float)) != hipSuccess ||
                hipMalloc(&tmpw, 3 * D * D * sizeof(float) > F * D * sizeof(float) ? 3 * D * D * sizeof(float) : F * D * sizeof(float)) != hipSuccess) {
                rc4 = ROHM_ERR_HIP;
                set_error("posenet_create: hipMalloc of the folded weight planes failed");
            }
            p->wplanes_fold = fp;
            float* fv = p->foldvec;
            const float ws = (p->nplane == kModeF16) ? kF16WeightScale : 1.0f;
            for (int l = 0; l < n_layer && rc4 == ROHM_OK; ++l) {
                LayerW& d = p->layers[l];
                d.l1_wpf = fp; fp += b_l1f; d.in_wpf = fp; fp += b_inf;
                d.l1_cf = fv; fv += F; d.l1_df = fv; fv += F; d.in_cf = fv; fv += 3 * D; d.in_df = fv; fv += 3 * D;
                if (hipMemcpy(tmpw, d.l1_w, F * D * sizeof(float), hipMemcpyDeviceToDevice) != hipSuccess ||
                    hipMemcpy(d.l1_df, d.l1_b, F * sizeof(float), hipMemcpyDeviceToDevice) != hipSuccess) { rc4 = ROHM_ERR_HIP; break; }
                hipLaunchKernelGGL(ln_fold_kernel, dim3((unsigned)F), dim3(256), 0, 0, tmpw, d.l1_df, d.n1_w, d.n1_b, d.l1_cf, (int)D);
                rc4 = launch_plane_split(tmpw, d_model, d_ff, d_model, p->nplane, ws, d.l1_wpf, 0);
                if (l > 0 && rc4 == ROHM_OK) {
                    const LayerW& pr = p->layers[l - 1];
                    if (hipDeviceSynchronize() != hipSuccess ||
                        hipMemcpy(tmpw, d.in_w, 3 * D * D * sizeof(float), hipMemcpyDeviceToDevice) != hipSuccess ||
                        hipMemcpy(d.in_df, d.in_b, 3 * D * sizeof(float), hipMemcpyDeviceToDevice) != hipSuccess) { rc4 = ROHM_ERR_HIP; break; }
                    hipLaunchKernelGGL(ln_fold_kernel, dim3((unsigned)(3 * D)), dim3(256), 0, 0, tmpw, d.in_df, pr.n2_w, pr.n2_b, d.in_cf, (int)D);
                    rc4 = launch_plane_split(tmpw, d_model, 3 * d_model, d_model, p->nplane, ws, d.in_wpf, 0);
                }
                if (hipDeviceSynchronize() != hipSuccess) rc4 = ROHM_ERR_HIP;
            }
            if (tmpw) (void)hipFree(tmpw);
            if (rc4 == ROHM_ERR_HIP) set_error("posenet_create: folding LayerNorm into the weight planes failed");
        }
        if (rc4 != ROHM_OK || hipDeviceSynchronize() != hipSuccess) {
            if (rc4 == ROHM_OK) set_error("posenet_create: cutting the weight planes failed");
            if (p->wplanes_fold) (void)hipFree(p->wplanes_fold);
            if (p->foldvec) (void)hipFree(p->foldvec);
            (void)hipFree(p->wplanes);
            (void)hipFree(p->arena);
            delete p;
            return rc4 != ROHM_OK ? rc4 : ROHM_ERR_HIP;
        }
    }
    if (p->ln_fold) {
        for (int l = 0; l < n_layer; ++l) {
            LayerW& d = p->layers[l];
            hipLaunchKernelGGL(ln_fold_kernel, dim3((unsigned)F), dim3(256), 0, 0, d.l1_w, d.l1_b, d.n1_w, d.n1_b, d.l1_c, (int)D);
            if (l > 0) {
                const LayerW& pr = p->layers[l - 1];
                hipLaunchKernelGGL(ln_fold_kernel, dim3((unsigned)(3 * D)), dim3(256), 0, 0, d.in_w, d.in_b, pr.n2_w, pr.n2_b,
                                   d.in_c, (int)D);
            }
        }
        const LayerW& last = p->layers[n_layer - 1];
        hipLaunchKernelGGL(ln_fold_kernel, dim3((unsigned)c_out), dim3(256), 0, 0, p->out_w, p->out_b, last.n2_w, last.n2_b,
                           p->out_c, (int)D);
        ROHM_HIP_CHECK(hipDeviceSynchronize());
    }
    *out = p;
    return ROHM_OK;
}

void rohm_posenet_destroy(rohm_posenet_t* h) {
    if (!h) return;
    if (h->arena) (void)hipFree(h->arena);
    if (h->wplanes) (void)hipFree(h->wplanes);
    if (h->wplanes_fold) (void)hipFree(h->wplanes_fold);
    if (h->foldvec) (void)hipFree(h->foldvec);
    delete h;
}

int rohm_posenet_precision(const rohm_posenet_t* h) { return h ? h->nplane : 0; }

int rohm_posenet_exchange_mode(const rohm_posenet_t* h) {
    if (!h) return 0;
    return (h->ln_fused ? 1 : 0) | (h->head_sk ? 2 : 0) | ((!h->exch_allowed && (h->ln_fused_env || h->head_sk_env)) ? 4 : 0) |
           (h->exch_fallback ? 8 : 0) | ((h->chain && h->ln_fused) ? 16 : 0) | ((h->chain == 2 && h->ln_fused) ? 32 : 0);
}

const char* rohm_posenet_exchange_guard(const rohm_posenet_t* h) { return h ? h->exch_reason : ""; }

int rohm_posenet_set_exchange(rohm_posenet_t* h, int on) {
    ROHM_ARG_CHECK(h != nullptr, "posenet_set_exchange: null handle");
    if (on) {      // back to what the environment asked for and the guard allows -- asked AGAIN: a tenant that made the probe at
                   // create fail may have gone, a handle that fell back may be on a device that is whole again.  The
                   // probe synchronises the device: this is a control call, never part of a launch path.
        if ((h->ln_fused_env || h->head_sk_env) && 2 * h->L + 1 <= 60 && (!h->exch_allowed || h->exch_fallback))
            h->exch_allowed = exchange_layout_ok(h->device, &h->exch_reason, true);
        h->exch_fallback = false;
        h->ln_fused = h->ln_fused_env && h->exch_allowed;
        h->head_sk = h->head_sk_env && h->exch_allowed;
    } else {
        h->exch_fallback = h->ln_fused || h->head_sk || h->exch_fallback;
        h->ln_fused = h->head_sk = false;
    }
    return ROHM_OK;
}

size_t rohm_posenet_stack_timeline_bytes(int B) {
    const int groups8 = (B + kNumXCD - 1) / kNumXCD * kNumXCD;
    return B > 0 ? (size_t)groups8 * 8 * kStackTimelineLayers * kStackTimelineStamps * sizeof(unsigned long long) : 0;
}

int rohm_posenet_set_stack_timeline(rohm_posenet_t* h, void* buf, size_t bytes, int B) {
    ROHM_ARG_CHECK(h != nullptr, "posenet_set_stack_timeline: null handle");
    ROHM_ARG_CHECK(buf == nullptr || (B > 0 && bytes >= rohm_posenet_stack_timeline_bytes(B) && (((uintptr_t)buf) & 7) == 0),
                   "posenet_set_stack_timeline: buffer too small for %d clips / misaligned", B);
    h->stack_timeline = static_cast<unsigned long long*>(buf);
    return ROHM_OK;
}

int rohm_posenet_inject_exchange_fault(rohm_posenet_t* h, int n_launches) {
    ROHM_ARG_CHECK(h != nullptr && n_launches >= 0, "posenet_inject_exchange_fault: bad argument");
    h->fault_left = n_launches;
    return ROHM_OK;
}

size_t rohm_posenet_workspace_bytes(const rohm_posenet_t* h, int B, int T) {
    if (!h || B <= 0 || T <= 0) return 0;
    return carve(h, B, T, nullptr).floats * sizeof(float);
}

int rohm_posenet_forward(const rohm_posenet_t* h, const float* x_t, const float* cond, const int64_t* t,
                         float* x0_out, int B, int T, void* ws, size_t ws_bytes, rohm_stream_t stream) {
    int rc = check_shape(h, B, T);
    if (rc) return rc;
    ROHM_ARG_CHECK(x_t && cond && t && x0_out && ws, "posenet_forward: null argument");
    ROHM_ARG_CHECK(((uintptr_t)ws % 256) == 0, "posenet_forward: workspace must be 256-byte aligned");
    hipStream_t s = (hipStream_t)stream;
    Workspace w = carve(h, B, T, (float*)ws);
    if (w.floats * sizeof(float) > ws_bytes) {
        set_error("posenet_forward: workspace too small (%zu < %zu)", ws_bytes, w.floats * sizeof(float));
        return ROHM_ERR_WORKSPACE;
    }
    if ((rc = arm_status(h, w, B, T, s))) return rc;
    if ((rc = launch_pack(h, x_t, w.apack, B, T, 0, s, pass_counter(w)))) return rc;
    if ((rc = launch_pack(h, cond, w.apack, B, T, 1, s))) return rc;
    if ((rc = run_network(h, w, t, 0, nullptr, x0_out, B, T, s))) return rc;
    const size_t n = (size_t)B * h->Cin * T;
    return launch_finish(x0_out, cond, nullptr, nullptr, nullptr, 0.f, 0.f, 0.f, h->traj, h->Cin, T, n, s);
}

size_t rohm_posenet_status_offset(const rohm_posenet_t* h, int B, int T) {
    if (!h || B <= 0 || T <= 0) return 0;
    // carve() on a null base yields null pointers: carve on a dummy base and take the distance (nothing is dereferenced)
    return (size_t)((char*)carve(h, B, T, (float*)256).xln - (char*)256);
}

int rohm_posenet_exchange_status(const rohm_posenet_t* h, int B, int T, void* ws, size_t ws_bytes, rohm_stream_t stream) {
    int rc = check_shape(h, B, T);
    if (rc) return rc;
    ROHM_ARG_CHECK(ws && ((uintptr_t)ws % 256) == 0, "posenet_exchange_status: null / misaligned workspace");
    Workspace w = carve(h, B, T, (float*)ws);
    ROHM_ARG_CHECK(w.floats * sizeof(float) <= ws_bytes, "posenet_exchange_status: workspace too small");
    hipStream_t s = (hipStream_t)stream;
    unsigned st[2] = {0u, 0u};
    ROHM_HIP_CHECK(hipMemcpyAsync(st, w.xln, sizeof(st), hipMemcpyDeviceToHost, s));
    ROHM_HIP_CHECK(hipStreamSynchronize(s));
    if (st[1] != kExchangeMagic || st[0] == 0u) return ROHM_OK;      // never used, or clean
    ROHM_HIP_CHECK(hipMemsetAsync(w.xln, 0, sizeof(unsigned), s));
    set_error("posenet: an in-kernel exchange failed since the last check (%s): the outputs computed on this workspace since then are "
              "not valid", st[0] == 2u ? "partner workgroups were placed on different XCDs" : "a wait for a partner workgroup ran into its bound");
    return ROHM_ERR_EXCHANGE;
}

int rohm_posenet_sample_loop(const rohm_posenet_t* h, float* x, const float* cond, const int64_t* t_model,
                             const float* coef, const float* noise, float* x0_last, float* x_in_last, int n_steps, int B,
                             int T, void* ws, size_t ws_bytes, rohm_stream_t stream) {
    int rc = check_shape(h, B, T);
    if (rc) return rc;
    ROHM_ARG_CHECK(x && cond && t_model && coef && ws, "posenet_sample_loop: null argument");
    ROHM_ARG_CHECK(n_steps >= 0, "posenet_sample_loop: negative step count");
    ROHM_ARG_CHECK(((uintptr_t)ws % 256) == 0, "posenet_sample_loop: workspace must be 256-byte aligned");
    hipStream_t s = (hipStream_t)stream;
    Workspace w = carve(h, B, T, (float*)ws);
    if (w.floats * sizeof(float) > ws_bytes) {
        set_error("posenet_sample_loop: workspace too small (%zu < %zu)", ws_bytes, w.floats * sizeof(float));
        return ROHM_ERR_WORKSPACE;
    }
    const size_t n = (size_t)B * h->Cin * T;
    ROHM_ARG_CHECK(n_steps <= kLoopChunk, "posenet_sample_loop: at most %d steps per call", kLoopChunk);
    if (n_steps == 0) return ROHM_OK;
    if ((rc = arm_status(h, w, B, T, s))) return rc;
    if ((rc = launch_pack(h, cond, w.apack, B, T, 1, s))) return rc;   // cond is constant over the loop
    // ... and so is its half of the input embedding: one [0 | cond] pass of the embed GEMM now, K = KX instead of KP in every step.
    // (The steps' A rows still carry cond in columns C .. KX: w_embed_x is zero there.)
    const bool hoist = h->cond_hoist && n_steps >= 4;
    if (hoist) {
        const int S = T + 1;
        ROHM_HIP_CHECK(hipMemset2DAsync(w.apack, (size_t)h->KP * sizeof(float), 0, (size_t)h->Cin * sizeof(float), (size_t)B * S, s));
        GemmParams g{};
        g.A = w.apack; g.lda = h->KP; g.W = h->w_embed; g.ldw = h->KP; g.C = w.econd; g.ldc = h->D;
        g.M = B * S; g.N = h->D; g.K = h->KP; g.S = S; g.tab = h->tab; g.tab0 = h->tok_table; g.ldtab = h->D; g.ldtab0 = 0;
        if ((rc = launch_gemm(g, EPI_EMBED, s))) return rc;
    }
    // timestep tokens come from the table built at create (the embedder depends on t only, heads.py:145-146)
    for (int i = 0; i < n_steps; ++i) {
        prof::set_step(i);
        const float c1 = coef[3 * i], c2 = coef[3 * i + 1], sigma = coef[3 * i + 2];
        ROHM_ARG_CHECK(sigma == 0.f || noise, "posenet_sample_loop: noise is required when sigma != 0");
        if (x_in_last && i == n_steps - 1)       // the reference keeps the input of the last step in batch['x_t']
            ROHM_HIP_CHECK(hipMemcpyAsync(x_in_last, x, n * sizeof(float), hipMemcpyDeviceToDevice, s));
        // x_t into the token-major pack: by its own kernel for the first step of the call, afterwards by the previous step's
        // finish_pack (which also advanced the pass counter)
        if ((i == 0 || !h->finish_pack) && (rc = launch_pack(h, x, w.apack, B, T, 0, s, pass_counter(w)))) return rc;
        float* x0 = (x0_last && i == n_steps - 1) ? x0_last : w.x0;
        const float* nz = noise ? noise + (size_t)i * n : nullptr;
        // one launch per step where the plan allows (run_network decides): the stack closes with head + update + the next step's pack
        TailArgs tail{x, cond, nz, (x0_last && i == n_steps - 1) ? x0_last : nullptr, (i + 1 < n_steps) ? w.apack : nullptr,
                      c1, c2, sigma, pass_counter(w), i};
        bool tail_ran = false;
        if ((rc = run_network(h, w, nullptr, t_model[i], nullptr, x0, B, T, s, hoist, h->finish_pack ? &tail : nullptr, &tail_ran))) return rc;
        if (tail_ran) continue;
        if (i + 1 < n_steps && h->finish_pack) {
            if ((rc = launch_finish_pack(h, x0, cond, x, nz, w.apack, c1, c2, sigma, B, T, pass_counter(w), s))) return rc;
        } else if ((rc = launch_finish(x0, cond, x, nz, x, c1, c2, sigma, h->traj, h->Cin, T, n, s))) {
            return rc;
        }
    }
    return ROHM_OK;
}

}  // extern "C"
