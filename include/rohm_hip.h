/*
 * rohm_hip.h -- C ABI of librohm_hip.so, the MI355X (gfx950) native library behind
 * RoHM's iterative-denoising hot path.
 *
 * The reference (sanweiliti/RoHM) is pure Python/PyTorch and has no FFI of its own
 * (SURVEY.md §8b); each entry point below names the reference function (file:line
 * under the RoHM tree) whose arithmetic it replaces.  The Python host side
 * (rohm_amd/) binds these through ctypes and mirrors the reference's classes.
 *
 * Conventions
 *   - every function returns 0 on success, <0 on error; rohm_last_error() returns a
 *     thread-local message.  Nothing throws or aborts across the boundary.
 *   - all tensor arguments are DEVICE pointers to contiguous fp32 (int64 for
 *     timesteps) owned by the caller; the library never allocates inside a
 *     forward/step call: scratch comes from the caller's workspace
 *     (*_workspace_bytes).  Weights are copied and re-laid-out at *_create time, so
 *     the caller may free its copies afterwards.
 *   - every launch goes on the caller-supplied hipStream_t (passed as void*);
 *     no hidden device synchronisation in forward / step / loop calls.  Set-up calls say so where they
 *     synchronise: *_create, rohm_exchange_probe, rohm_posenet_set_exchange(h, 1), the status read
 *     rohm_posenet_exchange_status; the stand-alone rohm_gemm_res_layernorm_f32 / rohm_output_process_f32
 *     probe an un-probed device on their FIRST call unless the stream records a graph (see there).
 *     One loop call does wait: rohm_trajnet_sample_loop in its clip-resident form (TrajNet's default, see rohm_trajnet_loop_mode)
 *     synchronises `stream` once at its end to read the in-kernel meetings' error word (ROHM_TRAJ_RESIDENT=0 keeps it wait-free).
 *   - handles are immutable after create (documented exceptions, all control calls that must not race with
 *     launches of the same handle: rohm_posenet_set_exchange, rohm_posenet_inject_exchange_fault,
 *     rohm_posenet_set_stack_timeline); calls are re-entrant across streams given distinct workspaces.
 *     One handle per device.
 */
#ifndef ROHM_HIP_H
#define ROHM_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define ROHM_OK 0
#define ROHM_ERR_ARG (-1)
#define ROHM_ERR_HIP (-2)
#define ROHM_ERR_WORKSPACE (-3)
#define ROHM_ERR_UNSUPPORTED (-4)
#define ROHM_ERR_EXCHANGE (-5)   /* an in-kernel exchange between workgroups failed: results since the last status check are invalid */

typedef void* rohm_stream_t; /* hipStream_t */

const char* rohm_last_error(void);
int rohm_version(void);

/* ------------------------------------------------------------------ launch profiler
 * Measurement aid for bench.py: while active, every instrumented kernel launch is bracketed by a
 * pair of HIP events recorded ON THE LAUNCH STREAM (so it sees the library's launches whatever
 * torch's current stream is).  Inside *_sample_loop only every `step_stride`-th denoising step is
 * bracketed, which keeps the timed region's perturbation negligible.  Not thread-safe.
 * rohm_profile_stop synchronises the recorded events and aggregates per kernel label:
 * launches, summed duration, summed algorithmic flops / bytes (as priced in DESIGN.md). */
typedef struct {
    char name[48];
    uint64_t launches;
    double total_ms;
    double flops;
    double bytes;
} rohm_profile_row;
/* The profiler is one process-wide recorder: start / stop / detail must not overlap launches issued by other host threads. */
int rohm_profile_start(int step_stride);
int rohm_profile_stop(rohm_profile_row* rows, int max_rows, int* n_rows);
/* on != 0: GEMM / conv-GEMM / GroupNorm launches are recorded under a label that carries their shape ("conv_gemm/64 M576
 * N512 K2560 S8"), one row per distinct shape -- per-launch-shape timing of the TrajNet step (scripts/bench_trajnet.py). */
int rohm_profile_detail(int on);

/* ------------------------------------------------------------------ building blocks
 * Exposed so that each kernel can be parity-tested and profiled on its own.        */

/* C[M,N] = epi(A[M,K] . W[N,K]^T): the fp32-MFMA GEMM every Linear of the path runs
 * on (nn.Linear in model/heads.py:154,169, nn.TransformerEncoderLayer in
 * model/posenet.py:63-69).  A, W row-major with K contiguous (lda/ldw in floats,
 * multiples of 4, 16-byte aligned); K a multiple of 32; M, N arbitrary.
 * epi: 0 = +bias, 1 = +bias, erf-form GELU (erf via A&S 7.1.26, abs err 1.5e-7), 2 = +bias +R[M,N](ldr).
 * bias may be NULL. */
int rohm_gemm_f32(const float* A, int lda, const float* W, int ldw, float* C, int ldc, int M, int N,
                  int K, const float* bias, const float* R, int ldr, int epi, rohm_stream_t stream);

/* In-place LayerNorm over the last dimension of x[M,D] (D == 512 or 256), eps 1e-5, biased
 * variance, affine -- nn.LayerNorm inside nn.TransformerEncoderLayer (model/posenet.py:63-69). */
int rohm_layernorm_f32(float* x, const float* gamma, const float* beta, int M, int D,
                       rohm_stream_t stream);

/* C = LayerNorm(A . W^T + bias + R) * gamma + beta in ONE launch -- the post-norm sublayer tail of nn.TransformerEncoderLayer
 * (model/posenet.py:63-69: x = norm1(x + out_proj(attn)), x = norm2(x + linear2(...)); `norm_first=False`), eps inside the root,
 * biased variance over the N columns.  The N / 64 or N / 128 column tiles of a 144-row tile exchange their per-row (mean, M2)
 * (merged pairwise by Chan's update: two-pass stability) through L2 while the kernel runs (they are dispatched back to back onto one XCD), so LN(x) is stored once and the raw
 * sum never reaches HBM.  Shapes: M % 144 == 0, K % 32 == 0, N / 64 or N / 128 in {1, 2, 4, 8} (else ROHM_ERR_UNSUPPORTED:
 * use rohm_gemm_f32(epi 2) + rohm_layernorm_f32).  `scratch`: rohm_gemm_res_layernorm_scratch_bytes(M, N) bytes, 64-byte aligned,
 * owned by the caller for the duration of the launch.  Its first 64 bytes are the exchange header: [0] the error word (0 = fine, 1 a
 * bounded wait expired, 2 partners on different XCDs; sticky -- the caller clears it), [1] a magic once armed, [2] a pass counter
 * this call advances on the device.  A scratch the library has not seen (no magic: uninitialised or recycled memory) is zeroed by the
 * call itself.  The slots are tagged with that device-side counter, so the launch may be recorded into a hipGraph: every replay
 * draws a fresh tag.  The partner tiles must be co-resident on one XCD: on a device that is not a whole MI355X (partitioned, CU
 * mask, probe launch failed -- see rohm_posenet_exchange_mode) the call returns ROHM_ERR_UNSUPPORTED.
 * The device in question is the one that owns `scratch`.  Its layout verdict comes from rohm_exchange_probe (below) or from an
 * earlier rohm_posenet_create on that device; on a device nobody has probed yet, the FIRST call of this function (and of
 * rohm_output_process_f32 with a scratch) runs the probe itself -- a hipMalloc, a null-stream launch and a device synchronisation,
 * once -- unless `stream` is recording a graph: then nothing is probed or cached, this call returns ROHM_ERR_UNSUPPORTED and
 * rohm_output_process_f32 uses plain tiles.  Call rohm_exchange_probe(device) before a capture (or before a latency-critical first
 * call) and these entry points never synchronise. */
/* Layout probe of `device`, always run afresh: properties, the CU-mask environment, 256 one-per-CU workgroups that must be resident
 * together with block b on XCD b % 8.  Returns 1 if the exchanging launches may be used there, 0 if not; `why` (optional) receives a
 * static string.  Allocates, launches on the null stream and synchronises the device: a set-up call.  A verdict about the device is
 * cached for the later launch calls; "the probe could not run" (set-up / launch failed) is returned but never cached.  Probes are
 * serialised across the processes of a host (advisory file lock), so the ranks of a node do not time each other out. */
int rohm_exchange_probe(int device, const char** why);
size_t rohm_gemm_res_layernorm_scratch_bytes(int M, int N);
int rohm_gemm_res_layernorm_f32(const float* A, int lda, const float* W, int ldw, float* C, int ldc, int M, int N, int K,
                                const float* bias, const float* R, int ldr, const float* gamma, const float* beta, float eps,
                                void* scratch, size_t scratch_bytes, rohm_stream_t stream);

/* Multi-head self-attention over n_tok tokens, head dim 64 or 128, for n_seq sequences:
 * qkv[n_seq*n_tok, 3*n_head*head_dim] (q | k | v blocks, q already scaled by head_dim^-1/2)
 * -> ctx[n_seq*n_tok, n_head*head_dim].  Replaces the scaled-dot-product inside
 * nn.MultiheadAttention (model/posenet.py:63-69).  n_tok = 144, head_dim = 128 (every released
 * configuration) runs the specialised kernel; other shapes the general one. */
int rohm_attention_f32(const float* qkv, float* ctx, int n_seq, int n_head, int n_tok, int head_dim,
                       rohm_stream_t stream);

/* ---- opt-in precision ladder: split GEMMs on 16-bit PLANES of the fp32 operands ----
 * The same Linears as rohm_gemm_f32 (model/posenet.py:63-69), each fp32 product a.w emulated by bf16 / fp16 MFMA products of
 * planes.  `nplane` is the MODE everywhere:  3 = bf16x6 (three bf16 planes cut by truncation, x = h + m + l exactly, six products,
 * fp32-class accuracy),  2 = bf16x3 (two bf16 planes, three products, ~2^-16),  16 = fp16x3 (two FP16 planes h = fp16(x),
 * l' = fp16((x - h) 2^11): x = h + 2^-11 l' to 2^-24 for 6e-5 <= |x| <= 65504; three products, the two cross terms in a second
 * accumulator of weight 2^-11: ~2^-22 per product).  A plane tensor of X[rows][K] (rows % 16 == 0, K % 32 == 0) is
 * rohm_planes_bytes(rows, K, mode) bytes in the fragment-major layout of rohm_amd/csrc/planes.h; producers write it (LayerNorm,
 * attention, the GELU GEMM) or rohm_planes_split cuts it from scale * X (scale: a power of two, 1 for activations; fp16 weight
 * planes are cut from 2^8 W and the GEMM is given acc_scale = 2^-8).  Never the default: rohm_posenet_create selects a mode only
 * under ROHM_GEMM_PRECISION=bf16x6 | bf16x3 | fp16x3. */
size_t rohm_planes_bytes(int rows, int K, int nplane);
int rohm_planes_split(const float* X, int ldx, int rows, int K, int nplane, float scale, void* planes, rohm_stream_t stream);
/* C[M,N] = epi(A . W^T) from the planes of A [M][K] and W [N][K]; M % 144 == 0, N % 64 == 0, K % 32 == 0.
 * epi: 0 = +bias, 1 = +bias, erf GELU, 2 = +bias +R, 3 = (+bias) * (n < qcols ? qscale : 1).  C (fp32, ldc) and / or
 * Cp (planes of the result over [M][N]) receive the result; the accumulator is multiplied by acc_scale first (0 = 1).
 * flags bit 0: plane output through 8-byte stores. */
int rohm_gemm_planes(const void* Ap, const void* Wp, float* C, int ldc, void* Cp, int M, int N, int K,
                     const float* bias, const float* R, int ldr, int qcols, float qscale, float acc_scale, int epi,
                     int nplane, int flags, rohm_stream_t stream);
/* rohm_gemm_planes with nn.LayerNorm (model/posenet.py:60-69 norm1 / norm2 of nn.TransformerEncoderLayer, post-norm) FOLDED into
 * the GEMMs around it (two-plane modes, K >= 192): the normalised tensor is never stored.  Row statistics travel as partial
 * (sum, sum of squares) pairs per 16 columns, row-block major: stats[row / 16][ln_dim / 16][row % 16][2] floats.
 *   out_stats (epi 2, ln_dim == N): receives the statistics of the result rows.
 *   ln_stats (epi 1 / 3, ln_dim == K): Ap are the planes of the RAW tensor x, Wp those of W[n][k] gamma[k], bias holds
 *     d[n] = b[n] + sum_k beta[k] W[n][k], ln_c holds c[n] = sum_k gamma[k] W[n][k]; the epilogue computes
 *     (acc - mu c) rstd + d = LN(x) W^T + b.
 *   r_stats (epi 2, ln_dim == N): R is raw as well; LN(R) = (R - mu) rstd r_gamma + r_beta is what is added.
 * Any of the three may be null. */
int rohm_gemm_planes_ln(const void* Ap, const void* Wp, float* C, int ldc, void* Cp, int M, int N, int K,
                        const float* bias, const float* R, int ldr, int qcols, float qscale, float acc_scale, int epi,
                        int nplane, const float* ln_stats, const float* ln_c, const float* r_stats, const float* r_gamma,
                        const float* r_beta, float* out_stats, int ln_dim, float ln_eps, rohm_stream_t stream);
/* rohm_layernorm_f32 that also writes the planes of its result (M % 16 == 0); the fp32 result is bit-identical. */
int rohm_layernorm_planes_f32(float* x, const float* gamma, const float* beta, int M, int D, int nplane,
                              void* planes, rohm_stream_t stream);
/* rohm_attention_f32 (n_tok = 144, head_dim = 128 only) writing the planes of ctx instead of fp32 ctx. */
int rohm_attention_planes_f32(const float* qkv, void* ctx_planes, int n_seq, int n_head, int nplane,
                              rohm_stream_t stream);

/* One DDPM ancestral update, elementwise over n floats:
 *   x_prev = c1*x0 + c2*x_t + guid_scale*guid_grad + sigma*noise
 * = q_posterior_mean_variance + p_sample[_with_grad]
 * (diffusion/gaussian_diffusion_posenet.py:212-234,426-434,466-479).  guid_grad may be NULL;
 * noise may be NULL when sigma == 0 (t == 0).  x_prev may alias x_t. */
int rohm_ddpm_step(const float* x_t, const float* x0, const float* noise, const float* guid_grad,
                   float c1, float c2, float sigma, float guid_scale, float* x_prev, size_t n,
                   rohm_stream_t stream);

/* Same update with per-sample timesteps and device-resident schedule tables (no host sync):
 *   tables [n_steps, 4] fp32 rows = {posterior_mean_coef1, posterior_mean_coef2,
 *                                     posterior_variance, posterior_log_variance_clipped}
 *   t      int64[B]   timestep of each sample (row of `tables`)
 *   x_prev[b] = c1*x0 + c2*x_t + var*(w_a*grad_a + w_b*grad_b) + [t_b != 0]*exp(0.5*logvar)*noise
 * over `row_len` floats per sample.  grad_a / grad_b may be NULL.  This is exactly
 * p_sample_with_grad (diffusion/gaussian_diffusion_posenet.py:436-480) after the network call. */
int rohm_ddpm_step_table(const float* x_t, const float* x0, const float* noise, const float* grad_a,
                         float w_a, const float* grad_b, float w_b, const float* tables,
                         const int64_t* t, int n_steps, float* x_prev, int B, size_t row_len,
                         rohm_stream_t stream);

/* ------------------------------------------------------------------------- PoseNet
 * model/posenet.py:12-96 + model/heads.py:112-176.                                   */
typedef struct rohm_posenet rohm_posenet_t;

typedef struct {
    const float *in_proj_w, *in_proj_b;   /* [3D, D], [3D]   self_attn.in_proj_*      */
    const float *out_proj_w, *out_proj_b; /* [D, D], [D]     self_attn.out_proj.*     */
    const float *lin1_w, *lin1_b;         /* [F, D], [F]     linear1.*                */
    const float *lin2_w, *lin2_b;         /* [D, F], [D]     linear2.*                */
    const float *norm1_w, *norm1_b;       /* [D]             norm1.*                  */
    const float *norm2_w, *norm2_b;       /* [D]             norm2.*                  */
} rohm_posenet_layer_weights;

typedef struct {
    const float *in_x_w, *in_x_b; /* [D, C_in], [D]  input_process.poseEmbedding.*         */
    const float *in_c_w, *in_c_b; /* [D, C_in], [D]  input_process_cond.poseEmbedding.*    */
    const float* pe;              /* [pe_len, D]     sequence_pos_encoder.pe (squeezed)    */
    int pe_len;
    const float *t_w0, *t_b0;     /* [D, D], [D]     embed_timestep.time_embed.0.*         */
    const float *t_w2, *t_b2;     /* [D, D], [D]     embed_timestep.time_embed.2.*         */
    const float *out_w, *out_b;   /* [C_out, D], [C_out]  output_process.poseFinal.*       */
    const rohm_posenet_layer_weights* layers; /* [n_layer] */
} rohm_posenet_weights;

/* OutputProcess.forward (model/heads.py:171-176): poseFinal Linear D -> C_out of every token, stored the way PoseNet.forward returns
 * it (model/posenet.py:94-96).  h [B * (T + 1), D] token-major (row b * (T + 1) + tok; the reference's [T + 1, B, D] with the two
 * leading axes exchanged; token 0 is the timestep token and has no output), w [C_out, D], b [C_out] ->
 * out[b][ch_off + c][0][tok - 1] of a [B, C_total, 1, T] tensor (the other channels are not touched).
 * `scratch` (optional, rohm_output_process_scratch_bytes() bytes, 256-byte aligned): lets shapes whose 144 x 64 tiles would need a
 * part-filled extra round of the 256 CUs (B = 64: 288 tiles) run as a stream-K launch -- the (tile, K chunk) units are dealt out
 * evenly, a tile cut in two is finished by the workgroup holding its tail.  It starts with the same exchange header as above ([0] the
 * error word, sticky).  NULL, or a device that fails the layout guard: plain tiling.  Recordable into a hipGraph.  D % 32 == 0. */
size_t rohm_output_process_scratch_bytes(void);
/* Host-only: the launch plan rohm_output_process_f32 (with scratch) uses for this shape.  Returns 1 for a stream-K launch -- 256
 * workgroups, 32 per XCD; XCD x owns tiles [x * tiles_per_xcd, (x + 1) * tiles_per_xcd) of the 144 x 64 tiling (row tiles of one column
 * tile adjacent), its j-th workgroup (block 8 j + x) the (tile, 32-wide K chunk) units [j u, (j + 1) u) of them in tile-major order,
 * u = units_per_workgroup -- and 0 for plain tiles (outputs set to 0).  tests/test_host_logic.py walks this schedule. */
int rohm_output_process_plan(int B, int T, int D, int C_out, int* units_per_workgroup, int* tiles_per_xcd);
int rohm_output_process_f32(const float* h, const float* w, const float* b, float* out, int B, int T, int D, int C_out,
                            int ch_off, int C_total, void* scratch, size_t scratch_bytes, rohm_stream_t stream);

/* Weight pointers may be host or device memory (copied with hipMemcpyDefault). */
int rohm_posenet_create(rohm_posenet_t** out, const rohm_posenet_weights* w, int d_model, int n_head,
                        int d_ff, int n_layer, int c_in, int c_out, int traj_dim, int device);
void rohm_posenet_destroy(rohm_posenet_t* h);
size_t rohm_posenet_workspace_bytes(const rohm_posenet_t* h, int B, int T);
/* 0: exact fp32 MFMA GEMMs (default); 3 / 2 / 16: the handle was created under ROHM_GEMM_PRECISION=bf16x6 / bf16x3 / fp16x3 and runs
 * the four Linears of every encoder layer (model/posenet.py:63-69) as split-bf16 GEMMs on planes. */
int rohm_posenet_precision(const rohm_posenet_t* h);

/* PoseNet.forward (model/posenet.py:75-96): x_t, cond [B, C_in, 1, T] contiguous, t int64[B]
 * -> x0_out [B, C_in, 1, T] (channels < traj_dim copied from cond, the C_out others predicted). */
int rohm_posenet_forward(const rohm_posenet_t* h, const float* x_t, const float* cond, const int64_t* t,
                         float* x0_out, int B, int T, void* ws, size_t ws_bytes, rohm_stream_t stream);

/* Status of the in-kernel exchanges of the forwards / loops run on workspace `ws` since the last call of this function.  Two of
 * PoseNet's kernels let workgroups of ONE launch hand data to each other through L2 (the LayerNorm inside the out-projection / FF2
 * GEMMs: rohm_gemm_res_layernorm_f32 above; the stream-K output head: rohm_output_process_f32 below).  Their waits are bounded; a
 * wait that runs into its bound (or partners found on different XCDs) sets a word in `ws` that stays set.  This call
 * synchronises `stream`, returns ROHM_ERR_EXCHANGE (and clears the word) if it is set, ROHM_OK otherwise.  Never expected on a
 * whole, exclusively owned MI355X -- the partner workgroups are co-resident by construction, and rohm_posenet_create checks the
 * device before it uses these launches -- but another tenant's long kernels can delay a partner past the bound, and a wrong result
 * must not pass silently: the Python loops call it after every fused chunk of steps (one synchronisation per <= 50 steps) and after
 * every step-wise forward, switch the handle to the exchange-free launches (rohm_posenet_set_exchange) and RE-RUN the chunk.
 * Direct callers of rohm_posenet_forward must call it before they trust the output. */
int rohm_posenet_exchange_status(const rohm_posenet_t* h, int B, int T, void* ws, size_t ws_bytes, rohm_stream_t stream);
/* Byte offset of the exchange header inside a workspace of this shape: [0] the error word (0 fine, 1 a bounded wait expired,
 * 2 partners on different XCDs), [1] 0x524f484d once a call has armed the workspace, [2] the pass counter (advanced on the device by
 * the first kernel of every network pass; the tags of a pass's exchanging launches derive from it).  Diagnostics and tests. */
size_t rohm_posenet_status_offset(const rohm_posenet_t* h, int B, int T);
/* Which launch forms this handle uses for the post-norm tails (model/posenet.py:63-69) and OutputProcess (model/heads.py:171-176):
 * bit 0 LayerNorm inside the out-projection / FF2 GEMMs, bit 1 stream-K output head; bit 2: the environment asked for them but the
 * layout guard refused at create (rohm_posenet_exchange_guard says why: < 256 CUs = a partitioned device, HSA_CU_MASK /
 * ROC_GLOBAL_CU_MASK set, or the probe launch -- 256 one-per-CU workgroups that must be resident together, block b on XCD b % 8 --
 * failed); bit 3: switched off after a failed exchange (rohm_posenet_set_exchange(h, 0)); bit 4: where the shape allows it (whole
 * 144-token clips, d_model 512, d_ff 1024) the four GEMMs between two attention launches -- out-projection + norm1, linear1 + GELU,
 * linear2 + norm2, the next layer's in-projection -- run as ONE launch whose workgroups hand tiles to each other per clip
 * (ROHM_POSENET_CHAIN=0: one launch per GEMM); bit 5: ... and attention too -- the whole encoder stack of a forward is one launch
 * (`encoder_stack_kernel`, the default from 32 clips on; ROHM_POSENET_CHAIN=layer keeps bit 4 without bit 5).  Bits 4 / 5 say what
 * the handle WOULD launch where the shape qualifies (whole clips, B >= 32 or ROHM_POSENET_CHAIN_ANY=1); they need bit 0.
 * ROHM_EXCHANGE_GUARD=off skips the guard, =probe skips its environment shortcut. */
int rohm_posenet_exchange_mode(const rohm_posenet_t* h);
const char* rohm_posenet_exchange_guard(const rohm_posenet_t* h);
/* on = 0: from now on this handle runs the exchange-free launches (GEMM + LayerNorm kernel pair, plain output-head tiles) -- what the
 * sampling loops do, before re-running the chunk, when rohm_posenet_exchange_status reports a failure.  on = 1: back to what the
 * environment asked for and the guard allows -- the guard is asked AGAIN if it had refused at create or the handle had fallen back
 * (a tenant that was resident then may have gone): that re-probe synchronises the device, so on = 1 is a control call between runs.
 * Not to be called while launches of this handle are being issued by another thread. */
int rohm_posenet_set_exchange(rohm_posenet_t* h, int on);
/* Test hook: the next `n_launches` LayerNorm-carrying GEMM launches of this handle publish one column tile's statistics under a
 * wrong tag, so its partners' waits expire (~0.2 s, once) and the error word is set -- a real failed exchange for the fallback tests. */
int rohm_posenet_inject_exchange_fault(rohm_posenet_t* h, int n_launches);
/* Diagnostics: phase timeline of the encoder stack (the one launch that carries nn.TransformerEncoder, model/posenet.py:63-69,92, from
 * 32 clips on).  With a device buffer of rohm_posenet_stack_timeline_bytes(B) bytes set, lane 0 of every workgroup of the following
 * stack launches of this handle (forwards / loop steps at batch size <= B) writes the 100 MHz wall clock at the seams of its phases:
 * buf[(block * 9 + layer) * 12 + k], k = 0 layer entered, 1 qkv of the clip complete (attention starts), 2 attention done, 3 ctx
 * complete, 4 out-projection + norm1 done, 5 met, 6 linear1 + GELU done, 7 met, 8 linear2 + norm2 done, 9 met, 10 next layer's
 * in-projection done; layer 8 = the leading phases (0 entered, 1 embedding done, 2 met, 3 in-projection of layer 0 done, 4 met).
 * Every launch overwrites the stamps.  buf = NULL switches it off (the default; the kernels then pay one scalar test per seam).
 * scripts/stack_timeline.py turns the stamps into per-phase spans and the in-stack attention rate (bench.py roofline.attention). */
size_t rohm_posenet_stack_timeline_bytes(int B);
int rohm_posenet_set_stack_timeline(rohm_posenet_t* h, void* buf, size_t bytes, int B);

/* Device-resident DDPM loop without guidance: p_sample_loop over `n_steps` descending timesteps
 * (diffusion/gaussian_diffusion_posenet.py:578-662, 388-434).
 *   x        [B, C_in, 1, T]  in: x_T, out: final sample (x_{-1})
 *   cond     [B, C_in, 1, T]
 *   t_model  int64[n_steps]   timestep fed to the network at loop step i (after timestep_map)
 *   coef     float[n_steps*3] per loop step: posterior_mean_coef1, posterior_mean_coef2,
 *                             sigma = exp(0.5*posterior_log_variance_clipped) (0 when t == 0)
 *   noise    [n_steps, B, C_in, 1, T] injected Gaussian noise (row i used at loop step i)
 *   x0_last  optional [B, C_in, 1, T]: pred_xstart of the last executed step (early_stop result)
 *   x_in_last optional [B, C_in, 1, T]: the INPUT x_t of the last executed step -- what the reference leaves in
 *            batch['x_t'] after a run (p_mean_variance, gaussian_diffusion_posenet.py:264)
 * All per-step scalars are host arrays (the loop is driven from the host, kernels stay async). */
int rohm_posenet_sample_loop(const rohm_posenet_t* h, float* x, const float* cond, const int64_t* t_model,
                             const float* coef, const float* noise, float* x0_last, float* x_in_last, int n_steps,
                             int B, int T, void* ws, size_t ws_bytes, rohm_stream_t stream);

/* ------------------------------------------------------------------------- TrajNet / TrajControl
 * model/trajnet.py:10-275 + model/heads.py:12-106: conv U-Net x0-predictor of the 13-channel trajectory,
 * optional ControlNet branch conditioned on PoseNet's 272-channel local pose. */
typedef struct rohm_trajnet rohm_trajnet_t;
typedef struct {
    const float* data; /* host or device */
    size_t numel;
} rohm_tensor_ref;
/* The parameter tensors in the reference's state_dict order (model/trajnet.py; `controlnet.*` first when
 * trajcontrol): every `weight` immediately followed by its `bias`; 186 tensors, +84 with TrajControl.
 * Conv weights are [C_out, C_in, k] (ConvTranspose1d: [C_in, C_out, k]) exactly as stored in a checkpoint. */
typedef struct {
    const rohm_tensor_ref* tensors;
    int n_tensors;
} rohm_trajnet_weights;

int rohm_trajnet_create(rohm_trajnet_t** out, const rohm_trajnet_weights* w, int mid_dim, int time_dim,
                        int c_traj, int c_ctrl, int trajcontrol, int device);
void rohm_trajnet_destroy(rohm_trajnet_t* h);
size_t rohm_trajnet_workspace_bytes(const rohm_trajnet_t* h, int B, int T);
/* Launch shape of TrajNet's latency-bound convolutions (process-wide; no counterpart in the reference, whose convs are
 * torch's): conv_wg_per_cu = workgroups of a conv GEMM that may share a CU (1 or 2; a 144x64 tile needs 60 KB of the 160 KB
 * LDS), split_min_chunks = fewest 32-wide K chunks a split-K slice may get, split_pow2 != 0 rounds split counts down to
 * powers of two (workgroup b computes split b % S and runs on XCD b % 8: with S a power of two an XCD's L2 holds only its
 * own K slices of the weights).  Results do not depend on them beyond the fp32 summation order of split-K.  The defaults are the measured optimum on MI355X. */
int rohm_trajnet_tune(int conv_wg_per_cu, int split_min_chunks, int split_pow2);

/* TrajNet.forward (model/trajnet.py:177-275): x_t, cond [B, T, c_traj], control_cond [B, T, c_ctrl] (NULL
 * without TrajControl), t int64[B] -> x0_out [B, T, c_traj].  T must be a multiple of 16. */
int rohm_trajnet_forward(const rohm_trajnet_t* h, const float* x_t, const float* cond, const float* control_cond,
                         const int64_t* t, float* x0_out, int B, int T, void* ws, size_t ws_bytes,
                         rohm_stream_t stream);

/* Device-resident DDPM loop (diffusion/gaussian_diffusion_trajnet.py:559-627, 440-466); arguments as
 * rohm_posenet_sample_loop with tensors of shape [B, T, c_traj] (x0_last / x_in_last optional, as there). */
int rohm_trajnet_sample_loop(const rohm_trajnet_t* h, float* x, const float* cond, const float* control_cond,
                             const int64_t* t_model, const float* coef, const float* noise, float* x0_last,
                             float* x_in_last, int n_steps, int B, int T, void* ws, size_t ws_bytes,
                             rohm_stream_t stream);

/* Which form the last rohm_trajnet_sample_loop of the calling host thread ran in: 0 one launch per layer, 1 the clip-resident step (one
 * launch per denoising step, an XCD's workgroups stay with its clips and meet through its L2 between the layers -- csrc/trajnet_resident.hip:
 * the default for TrajNet at B <= 64 on a device that passed the exchange probe, opt-in ROHM_TRAJ_RESIDENT=1 for TrajControl, off with
 * ROHM_TRAJ_RESIDENT=0; when it runs, rohm_trajnet_sample_loop waits for the stream once at its end to read the exchange's error word, and
 * a wait that expired hands the call back to form 0 with x restored), 2 the recorded step (opt-in ROHM_TRAJNET_GRAPH=1).  No counterpart in
 * the reference. */
int rohm_trajnet_loop_mode(void);

/* ------------------------------------------------------------------------- SMPL-X + guidance
 * Joints-only SMPL-X (third-party smplx==0.1.28 `SMPLX.forward` / `lbs`, called from
 * data_loaders/motion_representation.py:389) and the two test-time guidance gradients of
 * model/posenet.py:196-317.  The hot path reads only joints[:, 0:22]; those depend on
 * J_regressor.(v_template + shapedirs.beta) and the kinematic chain, never on vertices, so the
 * regressor is folded once at create time (fp64 accumulation) and a step is O(22) 3x3 products
 * per frame instead of full linear blend skinning. */
typedef struct rohm_smplx rohm_smplx_t;

/* v_template [V,3], shapedirs [V,3,n_shape_total] (first 10 = betas), J_regressor [J,V], parents int32[J];
 * pointers may be host or device memory. */
int rohm_smplx_create(rohm_smplx_t** out, const float* v_template, const float* shapedirs, int n_shape_total,
                      const float* J_regressor, const int32_t* parents, int V, int J, int device);
void rohm_smplx_destroy(rohm_smplx_t* h);

/* joints[N, n_out, 3] (n_out <= 22) from axis-angle pose [N, n_pose, 3] (global orient first; joints beyond
 * n_pose are unrotated), betas [N,10], transl [N,3]: Rodrigues (angle = |r + 1e-8|) + forward kinematics. */
int rohm_smplx_joints(const rohm_smplx_t* h, const float* pose, int n_pose, const float* betas,
                      const float* transl, int N, float* joints, int n_out, rohm_stream_t stream);

/* Dataset-side per-frame SMPL-X work, batched over the frames of a recording (data_loaders/dataloader_video.py:121-142,
 * :282-300 -- one smplx call, one cam2world transform and one update_globalRT_for_smplx (utils/other_utils.py:189-240)
 * PER FRAME there): axis-angle global_orient [N,3], body_pose [N,63], betas [N,10], transl [N,3] (device, float32),
 * rigid = cam2world [4,4] row-major (device float32) -> joints_world [N,22,3] (float32) and orient_transl_world [N,6]
 * (float64: new global_orient, new transl -- the two entries of the parameter dict the function rewrites). */
int rohm_smplx_frames_to_world(const rohm_smplx_t* h, const float* global_orient, const float* body_pose,
                               const float* betas, const float* transl, const float* rigid, int N, float* joints_world,
                               double* orient_transl_world, rohm_stream_t stream);

size_t rohm_guidance_workspace_bytes(int B, int T);

/* guide_skating_with_smpl (model/posenet.py:196-257, compute_grad='x_0'): x0 [B,294,1,T] normalised
 * prediction, mean294/std294 the dataset statistics -> grad_out [B,294,1,T] = d(-loss)/dx0 with channels
 * [0,22) and [290,294) zeroed.  counts2 (device float[2]) receives the two skating-mask counts
 * (abs-trajectory, SMPL-X recovery); both zero <=> the reference returns a 0-d zero, and grad_out is
 * then all zeros. */
int rohm_guidance_skating_grad(const rohm_smplx_t* h, const float* x0, const float* mean294, const float* std294,
                               int B, int T, float* grad_out, float* counts2, void* ws, size_t ws_bytes,
                               rohm_stream_t stream);

/* The two halves of rohm_guidance_skating_grad, for clip sharding with GLOBAL-batch semantics:
 * the loss normalises by mask counts over the whole batch (model/posenet.py:231,243), so each rank runs `prepare`
 * (local counts -> counts2), the host all-reduces the 2 floats (RCCL), and `apply` uses the summed counts.  `apply`
 * must follow `prepare` on the same x0 / workspace. */
int rohm_guidance_skating_prepare(const rohm_smplx_t* h, const float* x0, const float* mean294, const float* std294,
                                  int B, int T, float* counts2, void* ws, size_t ws_bytes, rohm_stream_t stream);
int rohm_guidance_skating_apply(const rohm_smplx_t* h, const float* x0, const float* mean294, const float* std294,
                                int B, int T, const float* counts2, float* grad_out, void* ws, size_t ws_bytes,
                                rohm_stream_t stream);

/* guide_2d_projection_with_smpl (model/posenet.py:260-317): transf_matrix [B,4,4] (affine, inverted on the
 * device), cam_R [3,3], cam_t [3], focal / center [B,2], kp2d [B, kp_frames, 22, 3] (u, v, confidence). */
int rohm_guidance_proj2d_grad(const rohm_smplx_t* h, const float* x0, const float* mean294, const float* std294,
                              const float* transf_matrix, const float* cam_R, const float* cam_t,
                              const float* focal, const float* center, const float* kp2d, int kp_frames, int B,
                              int T, float* grad_out, void* ws, size_t ws_bytes, rohm_stream_t stream);

/* Full linear blend skinning (smplx==0.1.28 `lbs`, as called with return_verts=True from
 * data_loaders/motion_representation.py:389-396; the post-loop meshes of test_amass_full.py:405-425).  Optional:
 * rohm_smplx_set_skinning uploads what the joints-only path does not need -- posedirs [(J-1)*9, V*3] (pose_feature
 * @ posedirs layout of smplx), lbs_weights [V, J] (+ v_template, shapedirs again) -- after which
 * rohm_smplx_forward produces joints [N, n_joints_out, 3] (may be NULL) and verts [N, V, 3] (may be NULL: joints only)
 * from poses pose [N, n_pose, 3] (pose_kind 0: axis-angle) or [N, n_pose, 6] (pose_kind 1: the interleaved 6-D vectors
 * of the motion representation, quaternion.py:482-501); global orient first; joints >= n_pose unrotated; expression = 0;
 * betas [N,10], transl [N,3].  ws: rohm_smplx_lbs_workspace_bytes(h, N), 256-byte aligned.  N <= 65535 per call.
 * The shape and pose blendshapes are one fp32-MFMA GEMM; the skinning blend T = W . A runs on the matrix core too when
 * lbs_weights is dense (mode 0), and over per-vertex ELL rows of the non-zero weights when >= 75 % of it is zero and no vertex
 * has more than 16 non-zero joints -- what a released SMPLX_*.npz looks like (mode 1).  rohm_smplx_skinning_mode reports the
 * choice made at rohm_smplx_set_skinning (-1: not set; 2: ELL rows of all joints, ROHM_LBS_SKIN=ell). */
int rohm_smplx_skinning_mode(const rohm_smplx_t* h);
int rohm_smplx_set_skinning(rohm_smplx_t* h, const float* v_template, const float* shapedirs, int n_shape_total,
                            const float* posedirs, int n_pose_feat, const float* lbs_weights);
size_t rohm_smplx_lbs_workspace_bytes(const rohm_smplx_t* h, int N);
int rohm_smplx_forward(const rohm_smplx_t* h, const float* pose, int n_pose, int pose_kind, const float* betas, const float* transl,
                       int N, float* joints, int n_joints_out, float* verts, void* ws, size_t ws_bytes,
                       rohm_stream_t stream);

/* recover_from_repr_smpl (data_loaders/motion_representation.py:332-398) straight from the 294-channel
 * representation: joints [B,T,22,3].  mode 0 = 'smplx_params' (:373-398, joints[:, 0:22] of the body model incl.
 * transl; h required), mode 1 = 'joint_abs_traj' (:349-371; h may be NULL), mode 2 = 'joint_rel_traj' (:312-329: root
 * angle / position as running sums of the per-frame velocities; h may be NULL).  repr is addressed with strides as in
 * rohm_traj_rederive; mean294/std294 de-normalise on the fly (both NULL = repr is already de-normalised). */
int rohm_repr_joints(const rohm_smplx_t* h, const float* repr, long long in_stride_b, long long in_stride_t,
                     long long in_stride_c, const float* mean294, const float* std294, int B, int T, int mode,
                     float* joints, rohm_stream_t stream);

/* Between-stage trajectory re-derivation: replaces the drivers' host round trip
 * test_amass_full.py:262-311 / test_prox_egobody.py:238-287 -- de-normalise TrajNet's representation,
 * recover_from_repr_smpl('smplx_params') (data_loaders/motion_representation.py:373-398), per-sequence
 * get_repr_smplx (:187-282), re-normalise, keep the 22 trajectory channels.
 * repr: element (b, t, c) at repr[b*in_stride_b + t*in_stride_t + c*in_stride_c] (floats), c < 294, t < T, normalised
 * with (mean_in, std_in); out: element (b, t, c), t < T-1, c < 22, at out[b*out_stride_b + t*out_stride_t +
 * c*out_stride_c], normalised with (mean_out, std_out) -- so the result can be written straight into channels 0..21
 * of PoseNet's `cond` in either layout.  2 <= T <= 800.  Degenerate facing directions reproduce the reference's NaN
 * handling (only the first NaN frame of a clip is patched with its predecessor, :212-215). */
int rohm_traj_rederive(const rohm_smplx_t* h, const float* repr, long long in_stride_b, long long in_stride_t,
                       long long in_stride_c, const float* mean_in, const float* std_in, const float* mean_out,
                       const float* std_out, int B, int T, float* out, long long out_stride_b,
                       long long out_stride_t, long long out_stride_c, rohm_stream_t stream);

/* AMASS evaluation metrics (eval_amass_full.py:67-147) as per-clip partial sums.  joints_clean / joints_rec
 * [B,T,22,3]; contact_* point at the 4 contact channels of the de-normalised clean / reconstructed representation
 * of frame (b, t) = contact[(b*T + t)*stride + k].  A (frame, joint) counts as occluded if bit `joint` of
 * occ_joint_mask is set (mask_scheme 'lower': joints 1,2,4,5,7,8,10,11, :76) or occ_start <= frame < occ_end
 * ('full': [65, 65 + int(ratio*145)), :84-87).  out [B,10] doubles:
 *   0 sum |clean - rec| over T*22        1 the same over occluded entries      2 number of occluded entries
 *   3 matching contact labels (of T*4)   4 skating frames, clean (of T-1)      5 skating frames, rec
 *   6 sum |accel_rec - accel_clean| over (T-2)*22    7 toe entries below -0.05 m (of T*2)
 *   8 sum of negative toe heights        9 min height of the clean clip (the ground reference, :105)
 * The script's numbers are sums over clips divided by the counts in brackets. */
int rohm_amass_metrics(const float* joints_clean, const float* joints_rec, const float* contact_clean,
                       long long contact_clean_stride, const float* contact_rec, long long contact_rec_stride,
                       unsigned occ_joint_mask, int occ_start, int occ_end, int B, int T, double* out,
                       rohm_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* ROHM_HIP_H */
