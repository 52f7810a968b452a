// Shared helpers for librohm_hip (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdarg.h>
#include "../../include/rohm_hip.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

namespace rohm {

void set_error(const char* fmt, ...);

#define ROHM_HIP_CHECK(expr)                                                             \
    do {                                                                                 \
        hipError_t _e = (expr);                                                          \
        if (_e != hipSuccess) {                                                          \
            rohm::set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), __FILE__, \
                            __LINE__);                                                   \
            return ROHM_ERR_HIP;                                                         \
        }                                                                                \
    } while (0)

#define ROHM_LAUNCH_CHECK()                                                              \
    do {                                                                                 \
        hipError_t _e = hipGetLastError();                                               \
        if (_e != hipSuccess) {                                                          \
            rohm::set_error("kernel launch failed: %s (%s:%d)", hipGetErrorString(_e),   \
                            __FILE__, __LINE__);                                         \
            return ROHM_ERR_HIP;                                                         \
        }                                                                                \
    } while (0)

#define ROHM_ARG_CHECK(cond, ...)                                                        \
    do {                                                                                 \
        if (!(cond)) {                                                                   \
            rohm::set_error(__VA_ARGS__);                                                \
            return ROHM_ERR_ARG;                                                         \
        }                                                                                \
    } while (0)

// ---- in-library launch profiler (HIP events on the launch stream; see rohm_profile_* in rohm_hip.h) --
namespace prof {
extern bool g_active;          // true while this launch should be bracketed
struct Scope {
    Scope(const char* label, double flops, double bytes, hipStream_t s);
    ~Scope();
    int idx;
    hipStream_t stream;
};
void set_step(int step);       // sample loops call this; activates every `stride`-th step
bool enabled();                // between rohm_profile_start and rohm_profile_stop
bool detail();                 // rohm_profile_detail(1): GEMM / GroupNorm launches are labelled with their shape
bool detail_requested();       // ... asked for, in a profiling session (whether or not the current step is a sampled one)
const char* intern(const char* s);
}  // namespace prof

constexpr int kNumXCD = 8;

// Bijective XCD-aware remap of a linear workgroup id (guide §5 "XCD swizzle must be
// bijective"): hardware places block b on XCD b % 8; we hand each XCD a contiguous run of
// logical tiles so neighbouring tiles (which share an A panel) share one L2.
__device__ __forceinline__ int xcd_remap(int bid, int nwg) {
    const int xcd = bid % kNumXCD;
    const int q = nwg / kNumXCD, r = nwg % kNumXCD;
    const int base = (xcd < r) ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
    return base + bid / kNumXCD;
}

// ---- GEMM (gemm_f32.hip) ---------------------------------------------------------------
enum GemmEpi {
    EPI_BIAS = 0,        // C = acc + bias[n]
    EPI_BIAS_GELU = 1,   // C = gelu_erf(acc + bias[n])
    EPI_BIAS_RES = 2,    // C = acc + bias[n] + R[m][n]
    EPI_QKV = 3,         // C = (acc + bias[n]) * (n < qcols ? qscale : 1)
    EPI_EMBED = 4,       // token-major embed: tok = m % S; tok==0 ? tab0[m/S][n] : acc + tab[tok][n]
    EPI_OUT_T = 5,       // transposed store of the output head (see posenet.hip)
    EPI_BIAS_RES_LN = 6, // C = LayerNorm(acc + bias[n] + R[m][n]) * gamma[n] + beta[n]: the row statistics are exchanged between the
                         // column tiles of a row tile through L2 while the kernel runs (xln_* below)
};

struct GemmParams {
    const float* A; int lda;
    const float* W; int ldw;
    float* C; int ldc;
    int M, N, K;
    const float* bias;
    const float* R; int ldr;
    // EPI_QKV
    int qcols; float qscale;
    // EPI_EMBED
    int S; const float* tab; const float* tab0; int ldtab; int ldtab0;
    int tab_by_row;              // 1: `tab` has one row per OUTPUT row m (a loop-invariant part of the sum computed once) instead of per token
    // EPI_OUT_T: rows = output channels (M = c_out), cols = tokens n -> (b = n / S, tok = n % S);
    // writes out[b][ch_off + m][tok-1] for tok >= 1 (T = S - 1 frames).
    int ch_off; int C_total; int T;
    // ---- conv-as-GEMM (trajnet.hip): the A operand is gathered from a channels-last activation
    // x[B, t_in, lda] with one K segment of cin_pad channels per tap:
    //   row m = b*conv_tq + tq,   k = j*cin_pad + ci   ->   x[b][tq*conv_stride + conv_off[j]][ci]
    // (zeros outside [0, conv_tin)); conv_taps == 0 means a plain GEMM.  The output row is
    // m*(orow_mul_m1 + 1) + orow_add (interleaved phases of a transposed convolution).
    int conv_taps, conv_cin_pad, conv_tin, conv_tq, conv_stride;
    int conv_off[5];             // tap offsets as the host states them; the kernel uses the progression below
    int conv_off0, conv_dstep;   // conv_off[j] == conv_off0 + j * conv_dstep (launch_gemm checks)
    const float* zero_page;      // >= 128 B of zeros (source of padded taps)
    int orow_mul_m1, orow_add;
    // columns >= ncol_split land ncol_jump floats further (0 = off): the two stride phases of a transposed convolution
    // computed as ONE GEMM with 2 C output columns, phase 1 going to the next output row (trajnet.hip upsample)
    int ncol_split, ncol_jump;
    // ---- split-K (few-tile, long-K problems: TrajNet's deep levels): workgroup (tile, split) accumulates the K
    // chunks of its split and stores the raw partial tile to partial[split][m][n] (row stride ld_partial); a second
    // kernel sums the splits in a fixed order, adds the bias and writes C.  ksplit <= 1: off.  EPI_BIAS only.
    int ksplit; float* partial; int ld_partial;
    int ksplit_defer;            // 1: leave the partial slabs un-reduced (the consumer sums them: trajnet.hip GroupNorm)
    int wg_per_cu;               // 2: let two workgroups share a CU (narrow tiles of latency-bound launches); else one is pinned
    // fused [conv5 | 1x1 residual] launch (trajnet.hip): columns >= res_col0 are the residual conv, whose weight is zero outside the
    // centre tap -- their tiles contract over the res_nk chunks starting at K offset res_k0 only, split res_ksplit ways (their slabs
    // are slabs 0 .. res_ksplit - 1 of `partial`; res_ksplit <= 1: they finish in the launch and store to C).  0 = off.
    int res_col0, res_ksplit, res_k0, res_nk;
    // ---- LayerNorm folded into the GEMMs around it (posenet.hip).  A producer (bias+residual epilogue) writes
    // per-row partial sums of its OUTPUT, one (sum, sum of squares) pair per column tile: out_stats[m][tile_n][2].
    // A consumer whose normalised operand is LN(x) = (x - mu) rstd gamma + beta runs on the RAW x with
    // gamma-scaled weights and finishes in the epilogue:  (acc - mu_m c_n) rstd_m + d_n,  c_n = sum_k gamma_k W_nk,
    // d_n = bias_n + sum_k beta_k W_nk  (passed as `ln_c` and `bias`).  For EPI_OUT_T the normalised operand is the
    // column side (tokens): (acc - mu_n c_m) rstd_n + d_m.  A residual that is itself LN(raw) is normalised on the
    // fly from `r_stats` / `r_gamma` / `r_beta`.  Stats cover ln_dim columns; parts = number of partial pairs.
    const float* ln_stats; int ln_parts; const float* ln_c;
    const float* r_stats; int r_parts; const float* r_gamma; const float* r_beta;
    float* out_stats; int out_parts;
    int ln_dim; float ln_eps;
    // ---- LayerNorm INSIDE the producer (EPI_BIAS_RES_LN; post-norm nn.TransformerEncoderLayer, model/posenet.py:63-69:
    // x = norm(x + sublayer(x))).  The N / BN column tiles of a row tile run at the same time on CUs of ONE XCD (the kernel's own
    // block -> tile map); each publishes its 144 per-row (mean, M2) pairs in xln_stats[row tile][column tile][144][4]
    // as (mean, tag, M2, tag) with ONE 16-byte store per row, polls the partner tiles' slots until they carry this
    // launch's tag (unique per launch, below), merges the pairs by Chan's update in a fixed tree over the tile index (every tile gets
    // bit-identical statistics), normalises its accumulators in registers and stores LN(x) once.  *xln_err is set if a wait ran
    // into its bound (1) or a partner published from another XCD (2) -- never on a whole, exclusively owned MI355X (the partner tiles
    // are co-resident by construction; the callers check that layout before they use these launches: exchange.hip).
    // Tag of a launch = xln_epoch (host: salt + launch index within a pass) + 64 x *xln_pass (device word, incremented by the first
    // kernel of every pass): recordable into a hipGraph, a replay draws fresh tags.  xln_fault != 0: test hook, column tile 0
    // publishes under a wrong tag so that its partners' waits expire.
    const float* ln_gamma; const float* ln_beta;
    float* xln_stats; unsigned* xln_err; unsigned xln_epoch; const unsigned* xln_pass; int xln_fault;
    unsigned* xln_xcc;      // [tiles_m][8]: XCD id + 1 of the workgroup that ran each tile (diagnostic: the tests assert the co-location)
    // ---- stream-K (EPI_OUT_T with 144 x 64 tiles: the output head, whose 288 tiles at B = 64 would take two rounds of 256 CUs with the
    // second one 1/8 full).  launch_gemm fills sk_units / sk_tiles8 when `sk_part` is bound and the shape qualifies (gemm_sk_plan):
    // 256 workgroups, each contracts sk_units consecutive (tile, K chunk) units of its XCD's sk_tiles8 tiles; a tile cut in two is
    // finished by the workgroup holding its tail, which adds the raw accumulators the other one left in sk_part[block] (flag =
    // (XCD id + 1) << 32 | tag in sk_flag[block]).  Same tag and the same caveats as the LayerNorm exchange: *xln_err reports a wait
    // that ran into its bound (1) or a partner on another XCD (2).
    float* sk_part; unsigned long long* sk_flag; int sk_units, sk_tiles8;
};
// stream-K scratch for launch_gemm(EPI_OUT_T): [flags: 256 x 8 B][slots: 256 x 36 KiB]; the error word is the caller's (the LayerNorm
// scratch's first word in posenet.hip).  gemm_sk_plan: does (M, N, K) run as stream-K, and with which split?
size_t gemm_sk_scratch_bytes();
void gemm_sk_bind(GemmParams& p, void* scratch, unsigned* err);
bool gemm_sk_plan(int M, int N, int K, int* units, int* tiles8);
// EPI_BIAS_RES_LN: can launch_gemm run (M, N, ...) with the in-kernel LayerNorm?  Scratch = stats + flags + error word.
bool gemm_ln_supported(int M, int N, int K);
size_t gemm_ln_scratch_bytes(int M, int N);
void gemm_ln_bind(GemmParams& p, void* scratch);  // fills xln_* from a scratch block (p.M must be set)

// ---- in-kernel exchanges: header words, first-use arming, layout guard (exchange.hip) -------------------------------------------
// An exchange scratch starts with a 64-byte header: [0] error word (0 fine, 1 a bounded wait expired, 2 partners on different XCDs),
// [1] kExchangeMagic once armed, [2] pass counter (the device part of the launch tags), [3] passes pending for it (StackParams::pass_add).  exchange_arm: on a scratch it sees for the
// first time (magic missing: torch.empty memory, a recycled block) it zeroes the header AND the slot regions `za` / `zb` / `zc` -- a tag is
// never 0 in its XCD field, so zeroed slots are stale by construction, whatever the block held before -- and, with `bump`, advances
// the pass counter (callers whose own first kernel does that pass bump = false).  Two tiny launches, no host synchronisation.
constexpr unsigned kExchangeMagic = 0x524f484du;
int exchange_arm(unsigned* header, void* za, size_t za_bytes, void* zb, size_t zb_bytes, bool bump, hipStream_t s, void* zc = nullptr,
                 size_t zc_bytes = 0);
// Does this device look like what the exchanging launches assume -- 256 CUs all available to one launch, 8 XCDs, block b on XCD
// b % 8 (a whole MI355X in SPX mode, no CU mask, nobody else's kernels resident)?  Queried once per device: properties, the CU-mask
// environment variables and a probe launch (256 one-per-CU workgroups that must all be resident at once and report their XCD).
// ROHM_EXCHANGE_GUARD=off trusts the device, =probe skips the environment shortcut.  `why` receives a static reason string.
// The probe allocates, launches on the null stream and synchronises the device: it belongs to handle creation / rohm_exchange_probe,
// never to a launch path that may be recording a hipGraph.  `reprobe` asks again even if a verdict is cached (after a fallback, or
// when a tenant that was there at first create has gone); a verdict that only says the probe could not run is never cached.
bool exchange_layout_ok(int device, const char** why, bool reprobe = false);
// The cached verdict without ever probing: 0 not probed yet, 1 fine, 2 refused.
int exchange_layout_state(int device, const char** why);
// The device that owns the allocation `ptr` points into (hipPointerGetAttributes); the current device if that cannot be told.
int device_of_pointer(const void* ptr);

int launch_gemm(const GemmParams& p, int epi, hipStream_t s);

// ---- the GEMM chain of one encoder layer as one launch (encoder_chain.hip) ---------------------------------------------------------
// y = norm1(h + out_proj(ctx)); ff = gelu(linear1(y)); h = norm2(y + linear2(ff)); qkv = in_proj_next(h) (absent when qkv == null).
// Token-major activations [M, ...], M = clips x 144; D = 512, F = 1024.  The workgroups of a clip exchange LayerNorm statistics and
// "my tile is stored" flags through L2: same scratch, tags and error word as EPI_BIAS_RES_LN (gemm_ln_bind fills xln_*), plus
// `flags` (encoder_chain_flag_bytes(M), zeroed at first use).  epoch: tags epoch, epoch + 1 (the two LayerNorm exchanges) and epoch
// (the flags) must be distinct from every other exchanging launch of the pass.
struct ChainParams {
    const float* ctx; float* h; float* y; float* ff; float* qkv;
    int M, D, F, tiles_m;
    const float *out_w, *out_b, *n1_w, *n1_b, *l1_w, *l1_b, *l2_w, *l2_b, *n2_w, *n2_b, *in_w, *in_b;
    float qscale, ln_eps;
    float* xln_stats; unsigned* xln_err; const unsigned* xln_pass; unsigned* xln_xcc; unsigned epoch;
    unsigned long long* flags;
    int fault;      // test hook: bit 0 / 1 sabotage the first / second LayerNorm exchange of the launch
};
// ---- ... and the whole encoder stack as one launch: per layer  attention (one head, or half of one, per workgroup of the clip) ->
// the chain above, the layers looped inside the kernel.  qkv holds layer 0's projection on entry (its own launch); h the embedded
// tokens; on return h = the encoder's output.  n_head must be 4.  flags: encoder_chain_flag_bytes(M).
struct StackLayerW { const float *out_w, *out_b, *n1_w, *n1_b, *l1_w, *l1_b, *l2_w, *l2_b, *n2_w, *n2_b, *in_w, *in_b; };
struct StackParams {
    float *h, *y, *ff, *qkv, *ctx;
    int M, D, F, tiles_m, L, n_head;
    float qscale, ln_eps;
    float* xln_stats; unsigned* xln_err; const unsigned* xln_pass; unsigned* xln_xcc; unsigned epoch;
    unsigned long long* flags;
    int fault;
    StackLayerW layer[8];      // in_w / in_b of layer l feed the phase that closes layer l - 1
    // front != 0: the launch starts from the packed input instead of from (h, qkv of layer 0): two leading phases per clip,
    //   h = InputProcess embedding (EPI_EMBED's arithmetic: A = apack [M, lda_pack] over k_embed columns, W = w_embed [D, ldw_embed],
    //   + positional / timestep table rows) and qkv = in_proj_0(h) -- model/posenet.py:85-92 up to the first encoder layer.
    int front;
    const float* apack; int lda_pack; const float* w_embed; int ldw_embed; int k_embed;
    int S; const float* tab; const float* tab0; int ldtab, ldtab0, tab_by_row;
    // Diagnostics (null on every product path): when set, lane 0 of every workgroup stamps the 100 MHz wall clock (s_memrealtime) at the
    // seams of its phases into timeline[(block * kStackTimelineLayers + layer) * kStackTimelineStamps + k]
    // (rohm_posenet_set_stack_timeline; scripts/stack_timeline.py turns the stamps into per-phase spans and the in-stack attention rate).
    unsigned long long* timeline;
    // tail != 0 (sampling loop, single-round launches only): the launch does not end with the encoder's output but carries on, per clip,
    // with OutputProcess (model/heads.py:171-176: x0[:, traj:] = h . Wout^T + bout, x0[:, :traj] = cond[:, :traj], model/posenet.py:94-96),
    // the ancestral update x_prev = c1 x0 + c2 x_t + sigma noise (gaussian_diffusion_posenet.py:212-234,426-434) written over x in place,
    // and the x_t half of the NEXT step's token-major pack -- one launch per denoising step.  The 17 x 9 (16 x 16) output blocks of a
    // clip are dealt to the 16 / 32 waves of its workgroups (<= 10 / 5 blocks per wave).
    int tail;
    const float* t_out_w; const float* t_out_b;      // [c_out, D], [c_out]
    float* t_x; const float* t_cond; const float* t_noise;      // [B, C, 1, T]; noise may be null (sigma == 0)
    float* t_x0; float* t_apack;                     // optional outputs: x0 [B, C, 1, T]; the next step's pack [M, t_lda] (x_t columns)
    float t_c1, t_c2, t_sigma;
    int t_traj, t_C, t_T, t_lda;
    unsigned* t_pass_ctr;                            // the exchange header's pass counter [0] and its pending word [1]
    // Tags of this launch: epoch + 64 x (*xln_pass + pass_add).  A launch with a tail is the only kernel of its step, so nobody
    // advances the counter between steps: the host numbers the steps of a call (pass_add = 0, 1, ...), the launch leaves pass_add + 1
    // in the pending word at its end, and the first kernel of the NEXT pass (the x_t pack) adds it to the counter.  The counter itself
    // is never written while workgroups of an exchanging launch may still be starting (a launch of more than one round of workgroups,
    // a CU mask): every workgroup of a launch reads the same value.
    unsigned pass_add;
};
// stamps of one layer: 0 layer entered, 1 qkv complete (attention starts), 2 attention done, 3 ctx complete (out-projection starts),
// 4 out-projection + norm1 done, 5 y complete, 6 linear1 + GELU done, 7 ff complete, 8 linear2 + norm2 done, 9 h complete,
// 10 next in-projection done (layer 7 with a tail: 9 h complete, 10 head + update + pack done); "layer" 8 = the leading phases: 0 entered, 1 embed done, 2 h complete, 3 in-projection of layer 0 done, 4 met
constexpr int kStackTimelineLayers = 9, kStackTimelineStamps = 12;
int launch_encoder_stack(const StackParams& p, hipStream_t s);
int encoder_chain_parts(int M, int D, int F);      // column tiles per clip (4 or 8), 0 = no chain form for this shape
bool encoder_chain_pays(int B);                    // whole rounds of persistent workgroups: is the chain / stack the faster plan for B clips?
size_t encoder_chain_flag_bytes(int M);
int launch_encoder_chain(const ChainParams& p, hipStream_t s);
// Is `s` recording a hipGraph?
bool stream_is_capturing(hipStream_t s);
#ifdef __HIPCC__
// All-lanes sum of a 64-wide wave, bit-identical to the xor butterfly  for (o = 32; o; o >>= 1) v += __shfl_xor(v, o)  -- but on the
// VALU: __shfl_xor compiles to ds_bpermute_b32 (the LDS crossbar, ~100 cycles and an lgkmcnt wait per step; twelve of them in a row
// were ~0.7 us of the LayerNorm kernel's latency chain).  Steps 32 / 16 exchange lane halves / rows with v_permlane32/16_swap (own +
// partner: addition commutes, so which of the two results is "own" does not matter); after them every lane of a column holds the
// same value, and a row rotation by 8 / 4 / 2 / 1 (DPP) delivers a value of exactly the class lane ^ 8 / 4 / 2 / 1 would.
__device__ __forceinline__ float wave_sum64(float v) {
    typedef unsigned rohm_u2 __attribute__((ext_vector_type(2)));
    rohm_u2 s = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    v = __uint_as_float(s[0]) + __uint_as_float(s[1]);
    s = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    v = __uint_as_float(s[0]) + __uint_as_float(s[1]);
    v += __uint_as_float(__builtin_amdgcn_update_dpp(0u, __float_as_uint(v), 0x128, 0xf, 0xf, true));      // row_ror:8
    v += __uint_as_float(__builtin_amdgcn_update_dpp(0u, __float_as_uint(v), 0x124, 0xf, 0xf, true));      // row_ror:4
    v += __uint_as_float(__builtin_amdgcn_update_dpp(0u, __float_as_uint(v), 0x122, 0xf, 0xf, true));      // row_ror:2
    v += __uint_as_float(__builtin_amdgcn_update_dpp(0u, __float_as_uint(v), 0x121, 0xf, 0xf, true));      // row_ror:1
    return v;
}

// erf-form GELU (activation="gelu", model/posenet.py:67; NOT the tanh approximation).  erf by Abramowitz-Stegun 7.1.26 (|abs err| <=
// 1.5e-7, i.e. at fp32 resolution of the 1 + erf term) -- branch-free, one rcp + one exp, ~3x cheaper than the
// library erff in a 72-element-per-lane epilogue.
__device__ __forceinline__ float gelu_erf(float x) {
    const float z = fabsf(x) * 0.70710678118654752440f;
    const float t = __frcp_rn(fmaf(0.3275911f, z, 1.0f));
    float p = fmaf(1.061405429f, t, -1.453152027f);
    p = fmaf(p, t, 1.421413741f);
    p = fmaf(p, t, -0.284496736f);
    p = fmaf(p, t, 0.254829592f);
    const float e = 1.0f - p * t * __expf(-z * z);      // erf(|x| / sqrt 2)
    return 0.5f * x * (1.0f + copysignf(e, x));
}

#endif

// ---- other kernels -----------------------------------------------------------------------
int launch_layernorm(float* x, const float* g, const float* b, int M, int D, hipStream_t s);
int launch_attention(const float* qkv, float* ctx, int n_seq, int n_head, int n_tok, int head_dim, hipStream_t s);
// Graph-replayable variants: per-step values come from device tables indexed by a device-side step counter.
int launch_ddpm_step_indexed(const float* x_t, const float* x0, const float* noise_base, const float* coef_tab,
                             const int* step_ctr, float* out, size_t n, hipStream_t s);
int launch_advance_counter(int* step_ctr, hipStream_t s);
int launch_ddpm_step(const float* x_t, const float* x0, const float* noise, const float* grad, float c1,
                     float c2, float sigma, float gscale, float* out, size_t n, hipStream_t s);

int launch_ddpm_step_table(const float* x_t, const float* x0, const float* noise, const float* ga, float wa,
                           const float* gb, float wb, const float* tables, const int64_t* t, int n_steps,
                           float* out, int B, size_t row_len, hipStream_t s);

}  // namespace rohm
