// TrajNet / TrajControl sample loop, CLIP-RESIDENT form: one launch per denoising step (model/trajnet.py:177-275, model/heads.py:12-106,
// diffusion/gaussian_diffusion_trajnet.py:440-466, 559-627).
//
// Why: the launch-per-layer loop (trajnet.hip) is 59 (TrajNet) / 86 (TrajControl) DEPENDENT kernels per step, each a few microseconds
// of work behind ~4 us of dispatch + drain -- 41 ms per 100 steps for one clip, 56 ms for 32 (profiles/r6_z_trajnet_loop.json).  A
// grid-wide barrier costs what a kernel boundary costs on this part (8 L2s, MI355X_MICROARCH.md "barrier-xcd" 4-5 us), so the step
// is NOT one grid-synchronised kernel.  Instead the dependency structure is used: clips never talk to each other, only the layers of
// ONE clip do.  An XCD owns a contiguous run of clips for the whole step; its 32 workgroups (block b runs on XCD b % 8) compute every
// layer of those clips and meet between layers through the flags of encoder_chain.hip's group_sync -- one L2, no cache maintenance:
// ~1.2 us per meeting.  Weights are read by every XCD (8 x 90 MB per step at the outside; one XCD streams at 1.3 TB/s, eight the same
// data at 6.9 TB/s: scripts/probes/xcd_stream_probe.hip); activations never leave the owner's L2.
//
// Work item of a layer = (a run of q clips of the XCD, one block of 16 output channels): the GEMM  rows = q x T_level conv positions,
// 16 columns, K = taps x C_in.  The INPUT rows of the item are staged once per 64-channel K chunk with a two-row zero halo per clip
// (LDS-DMA, sc1: partners wrote them in this launch), and the taps are row offsets into that image -- no gathered operand, the five
// taps of a conv re-use one staged tile.  The four waves split K (wave w takes channels 16 w .. 16 w + 15 of every chunk and tap), the
// partial accumulators meet in LDS, and the epilogue -- bias, GroupNorm(8) over (C/8 channels x T) per clip, Mish, time bias, residual,
// control residual (heads.py:43-54, 98-104) -- runs on the summed tile: a GroupNorm group is inside the item for C <= 128 and shared by
// 2 / 4 neighbouring items for C = 256 / 512, which exchange (mean, M2) granules through L2 and merge them in a fixed tree (Chan), the
// discipline of gemm_f32.hip's EPI_BIAS_RES_LN.  A ResidualTemporalBlock is two layers (its 1x1 residual conv rides along the first as
// a sixth "tap" on the centre row), the transposed conv two tap-pairs on one staged tile, the head + ancestral update the closing
// layer: 30 meetings per TrajNet step instead of 59 launches, 32 instead of 86 for TrajControl (its ControlNet branch runs in the same
// slots as the U-Net encoder, on other workgroups).
//
// Guards: exchange.hip's layout probe (whole device, block b on XCD b % 8), bounded waits that report into an error word; the host
// checks it at the end of the loop, restores x_T and hands the call to the launch-per-layer loop if a wait expired.
//
// Status: the default for TrajNet at B <= 64 (38.4 / 38.8 / 52.3 / 69.9 ms per 100 steps at B = 1 / 8 / 32 / 64 against 41.3 / 46.1 / 55.8 /
// 73.4 for the launch-per-layer loop; 100 launches instead of 5 946), opt-in for TrajControl (ROHM_TRAJ_RESIDENT=1), which is 20-25 %
// slower in this form.  What replaces 59 kernel boundaries at ~7 us is 30 layers at ~6 us of address set-up, K-split reduction, GroupNorm
// epilogue and meeting (NOTES.md section 12.4; the kernel prints its own per-layer timeline with ROHM_TRAJ_RESIDENT_TIMELINE=1).
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <algorithm>
#include <type_traits>
#include <vector>
#include "trajnet_priv.h"

namespace rohm {
namespace res {

constexpr int kWG = 32;                        // workgroups per XCD
constexpr int kMaxRB = 10;                     // 16-row blocks of conv positions per item
constexpr int kMaxRows = kMaxRB * 16;
constexpr int kMaxLdsRows = 176;               // staged input rows per item: clips x (T_in + 4), rounded up to the 16 rows of a DMA pass
constexpr int kAIters = kMaxLdsRows / 16;
constexpr int kKC = 64;                        // input channels per K chunk
constexpr int kMaxTaps = 5;
constexpr int kWRows = (kMaxTaps + 1) * 16;    // weight rows per chunk: taps x 16 columns, + 16 of the 1x1 residual conv
constexpr int kStage = 2 * (kMaxLdsRows + kWRows) * kKC;      // floats of the staging ring (two stages of the largest item; more, smaller ones
                                                            // for the deep levels); after the K loop the same floats hold the waves' partial tiles
constexpr int kRed = 4 * kMaxRows * 16;        // [wave][row][16]
static_assert(2 * kRed <= kStage, "the partial tiles of conv + residual must fit the staging buffers");
constexpr int kMaxSync = 64, kMaxItems = 256, kMaxOps = 64, kMaxQ = 16, kMaxStages = 8;

struct alignas(16) ROp {
    int kind;                                  // 0 conv layer, 1 head + update
    int cout, cin_pad, t_in, t_out, stride, ntaps;
    int off[kMaxTaps];
    int lda, ldw, omul, oadd, t_dst;
    int gn;                                    // GroupNorm group width in channels (0: none -> y = conv + bias)
    int tb_off;                                // time bias columns inside the step's row (-1: none)
    int ldres, ldadd2, lddst, lddst2, ld_res_out;
    int q, wg_off, sync_after;
    int rsplit;                                // > 1 (only with q == 1): a clip's conv positions are dealt to `rsplit` items of t_out / rsplit rows each
    int slot_base;                             // first statistics slot of this layer (layers between two meetings use disjoint slots)
    const float* A; const float* W; const float* bias;
    const float* Wres; const float* bres; float* res_out;      // fused 1x1 residual conv of the same input (null: none)
    const float* gamma; const float* beta;
    const float* res; const float* add2;
    float* dst; float* dst2;
};

static_assert(sizeof(ROp) % 4 == 0, "layer records are copied word by word");
constexpr int kOpsFloats = (int)(sizeof(ROp) / 4) * kMaxOps;      // the step's layer records, copied into LDS once per launch
constexpr int kLdsFloats = kStage + 1536 + kOpsFloats;            // + row statistics [160][4][2], pair statistics [16][2], layer records
static_assert(kLdsFloats * 4 <= 160 * 1024, "LDS budget");

struct RParams {
    const ROp* ops; int n_ops;
    int B, T, n_per_xcd;
    const float* tb_row;
    const float* zero_page;
    unsigned long long* flags;                 // [8 XCDs][kMaxSync][32]
    unsigned long long* slots;                 // [8 XCDs][kMaxItems][kMaxQ][2]
    unsigned* err;
    unsigned tag_base;                         // (step + 1) << 8: tags of this launch = tag_base + meeting / layer index
    // head + ancestral update (trajnet.hip traj_tail_kernel's arithmetic)
    const float* fin; int ldfin; const float* hw; int ldhw; int hcin; const float* hb; int ctraj;
    float* x; float* xin; int ldxin; float* x0_out; const float* noise; float c1, c2, sigma;
    // diagnostics (null on every product path): lane 0 of every workgroup stamps the 100 MHz clock at the seams of every layer into
    // timeline[(block * kMaxOps + layer) * 8 + k]: 0 layer entered, 1 operand addresses of its (last) item ready, 2 K loop done,
    // 3 partial tiles summed, 4 epilogue done, 5 met (ROHM_TRAJ_RESIDENT_TIMELINE=1 prints the last step's table)
    unsigned long long* timeline;
    int fault;                                 // test hook (ROHM_TRAJ_RESIDENT_FAULT=1): workgroup 0 of XCD 0 stays away from the fourth meeting
};

// x / d for 0 <= x < 65536, quotient <= 64, inv ~ 1 / d: exact (x + 0.5 stays >= 0.5 / d away from every multiple of d, the float error is < 1e-5)
__device__ __forceinline__ int idiv_small(int x, float inv) { return (int)(((float)x + 0.5f) * inv); }
// 1 / d for idiv_small (the hardware reciprocal, 1 ulp: far inside idiv_small's margin; a true division is a ten-instruction sequence, and an
// INTEGER division of two run-time values ~40 dependent ones -- four of them were 0.6 us of every item's set-up)
__device__ __forceinline__ float rcp_small(int d) { return __builtin_amdgcn_rcpf((float)d); }

__device__ __forceinline__ int lds_off64(int row, int slot) { return row * kKC + ((slot ^ (row & 15)) << 2); }

__device__ __forceinline__ f32x4 ld16_l2(const float* p) {      // 16 bytes another workgroup of the XCD may have written in this launch: past the L1
    const unsigned long long* q = reinterpret_cast<const unsigned long long*>(p);
    const unsigned long long a = __hip_atomic_load(q, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const unsigned long long b = __hip_atomic_load(q + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    return f32x4{__uint_as_float((unsigned)a), __uint_as_float((unsigned)(a >> 32)), __uint_as_float((unsigned)b), __uint_as_float((unsigned)(b >> 32))};
}

// error word: low byte 1 = a bounded wait expired, 2 = a partner published from another XCD; bits 8.. = where (meeting index, or
// 0x80 | layer index for a statistics exchange)
__device__ __forceinline__ bool wait_tag(const unsigned long long* p, unsigned tag, unsigned* err, unsigned where, unsigned long long* out) {
    for (int it = 0;; ++it) {
        const unsigned long long f = __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if ((unsigned)(f >> 32) == tag) { *out = f; return true; }
        __builtin_amdgcn_s_sleep(1);
        if ((it & 127) == 127 && __hip_atomic_load(err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u) return false;
        if (it > (1 << 19)) { __hip_atomic_store(err, 1u | (where << 8), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); return false; }
    }
}

// The 32 workgroups of an XCD meet: everybody's stores of the layer are in the XCD's L2 before anybody reads them (encoder_chain.hip
// group_sync with 32 partners; no acquire: every partner-written operand is read past the L1 -- LDS-DMA sc1, ld16_l2).
__device__ __forceinline__ void xcd_sync(unsigned long long* flags, int j, unsigned tag, unsigned xcc1, unsigned* err, unsigned where, int tid, bool absent) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    // the flag by a PLAIN store: it stays in the XCD's L2, where the partners' device-scope polls are served (an sc1 store writes through
    // to memory and drops the line: every poll then pays the trip to the memory side)
    if (tid == 0 && !absent) { flags[j] = ((unsigned long long)tag << 32) | xcc1; asm volatile("" ::: "memory"); }
    if (tid < kWG && tid != j) {
        unsigned long long f;
        if (wait_tag(flags + tid, tag, err, where, &f) && (unsigned)f != xcc1) __hip_atomic_store(err, 2u | (where << 8), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    __syncthreads();
}

// s_waitcnt vmcnt(n) for a wave-uniform run-time n (the instruction takes an immediate)
__device__ __forceinline__ void wait_vmcnt(int n) {
#define ROHM_VMC(k) case k: asm volatile("s_waitcnt vmcnt(" #k ")" ::: "memory"); break;
    switch (n) {
        ROHM_VMC(0) ROHM_VMC(1) ROHM_VMC(2) ROHM_VMC(3) ROHM_VMC(4) ROHM_VMC(5) ROHM_VMC(6) ROHM_VMC(7) ROHM_VMC(8) ROHM_VMC(9)
        ROHM_VMC(10) ROHM_VMC(11) ROHM_VMC(12) ROHM_VMC(13) ROHM_VMC(14) ROHM_VMC(15) ROHM_VMC(16) ROHM_VMC(17) ROHM_VMC(18) ROHM_VMC(19)
        ROHM_VMC(20) ROHM_VMC(21) ROHM_VMC(22) ROHM_VMC(23) ROHM_VMC(24) ROHM_VMC(25) ROHM_VMC(26) ROHM_VMC(27) ROHM_VMC(28) ROHM_VMC(29)
        ROHM_VMC(30) ROHM_VMC(31) ROHM_VMC(32) ROHM_VMC(33) ROHM_VMC(34) ROHM_VMC(35) ROHM_VMC(36) ROHM_VMC(37) ROHM_VMC(38) ROHM_VMC(39)
        ROHM_VMC(40) ROHM_VMC(41) ROHM_VMC(42) ROHM_VMC(43) ROHM_VMC(44) ROHM_VMC(45) ROHM_VMC(46) ROHM_VMC(47) ROHM_VMC(48) ROHM_VMC(49)
        ROHM_VMC(50) ROHM_VMC(51) ROHM_VMC(52) ROHM_VMC(53) ROHM_VMC(54) ROHM_VMC(55) ROHM_VMC(56) ROHM_VMC(57) ROHM_VMC(58) ROHM_VMC(59)
        ROHM_VMC(60) ROHM_VMC(61) ROHM_VMC(62) ROHM_VMC(63)
        default: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
    }
#undef ROHM_VMC
}

__device__ __forceinline__ float sum16(float v) {      // all-lanes sum of a 16-lane row
    v += __shfl_xor(v, 8);
    v += __shfl_xor(v, 4);
    v += __shfl_xor(v, 2);
    v += __shfl_xor(v, 1);
    return v;
}

// Shape of one (clip run, 16-column block) item and of its staging ring -- computed the same way by the item itself and by whoever
// requests its first weight chunks ahead of time.
struct Geom {
    int ncb, cb, clip0, qi, rows, nrb, rstride, lrows, a_iters, nt, ntt, nch, col0, per_chunk, stage_f, NS, pre;
    int rp, t_part, t0;                                // row part of the clip: conv positions t0 .. t0 + t_part
    bool has_res;
};
__device__ __forceinline__ Geom item_geom(const ROp& op, int id, int c_lo, int nx) {
    Geom g;
    g.ncb = op.cout >> 4;
    // item id = ((clip run * rsplit + row part) * column blocks + column block): the partners of a GroupNorm group are neighbours
    const int cr = idiv_small(id, rcp_small(g.ncb));
    g.cb = id - cr * g.ncb;
    const int cg = idiv_small(cr, rcp_small(op.rsplit));
    g.rp = cr - cg * op.rsplit;
    g.t_part = idiv_small(op.t_out, rcp_small(op.rsplit)); g.t0 = g.rp * g.t_part;
    const int k0 = cg * op.q;
    g.qi = min(op.q, nx - k0);
    g.clip0 = c_lo + k0;
    g.rows = g.qi * g.t_part; g.nrb = (g.rows + 15) >> 4;
    g.rstride = (op.rsplit > 1 ? g.t_part * op.stride : op.t_in) + 4;      // staged input rows per clip: the part's span + two rows either side
    g.lrows = g.qi * g.rstride; g.a_iters = (g.lrows + 15) >> 4;
    g.nt = op.ntaps; g.has_res = op.Wres != nullptr; g.ntt = g.nt + (g.has_res ? 1 : 0);
    g.nch = op.cin_pad >> 6; g.col0 = g.cb * 16;
    static_assert(kKC == 64, "chunk count by shift");
    // staging ring: a stage = the item's input rows (whole 16-row DMA passes) + its weight rows of ONE 64-channel chunk; as many
    // stages as the LDS holds (2 at level 0 .. 8 for a 1x1 conv at the deep levels): a deep-level layer (8 .. 16 chunks of < 1 us of
    // MFMAs each) lives on the chunks in flight
    g.per_chunk = g.a_iters + g.ntt;                   // DMA instructions per chunk and wave
    g.stage_f = g.per_chunk * 16 * kKC;
    int NS = min(kMaxStages, idiv_small(kStage, rcp_small(g.stage_f)));
    NS = min(NS, g.nch + 1);
    while (NS > 2 && (NS - 2) * g.per_chunk > 63) --NS;      // s_waitcnt vmcnt counts to 63
    g.NS = NS;
    g.pre = min(NS - 1, g.nch);                        // chunks in flight ahead of the one being multiplied
    return g;
}
// this lane's weight addresses of chunk 0: pass `tap` of a chunk = 16 columns x 64 channels of that tap (the 1x1 residual conv's last)
__device__ __forceinline__ void weight_src(const RParams& p, const ROp& op, const Geom& g, int tid, const float** w_src) {
    const int col = tid >> 4, slot = (tid & 15) ^ col;      // row of pass `tap` = tap * 16 + col: (row & 15) == col
    const unsigned wrow = (unsigned)((g.col0 + col) * op.ldw + slot * 4);
#pragma unroll
    for (int tp = 0; tp < kMaxTaps + 1; ++tp) {
        w_src[tp] = p.zero_page;
        if (tp < g.nt) w_src[tp] = op.W + (wrow + (unsigned)(tp * op.cin_pad));
        else if (tp == g.nt && g.has_res) w_src[tp] = op.Wres + (unsigned)((g.col0 + col) * op.cin_pad + slot * 4);
    }
}
__device__ __forceinline__ void dma_w(const Geom& g, const float* const* w_src, float* smem, int stage, int kc, int wave) {
    float* const Wb = smem + stage * g.stage_f + g.a_iters * 16 * kKC;
#pragma unroll
    for (int tp = 0; tp < kMaxTaps + 1; ++tp)
        if (tp < g.ntt)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(w_src[tp] + kc * kKC),
                                             (__attribute__((address_space(3))) void*)(Wb + (tp * 256 + wave * 64) * 4), 16, 0, 0);
}
// One (clip run, 16-column block) item of a conv layer.
__device__ __forceinline__ void conv_item(const RParams& p, const ROp& op, const int oi, const int id, const int xcd, const int c_lo, const int nx,
                                          float* smem, const int tid, unsigned long long* const tl) {
    auto stamp = [&](int k) __attribute__((always_inline)) { if (tl != nullptr && threadIdx.x == 0) tl[k] = __builtin_amdgcn_s_memrealtime(); };
    const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), li = lane & 15, lg = lane >> 4;
    const Geom g = item_geom(op, id, c_lo, nx);
    const int cb = g.cb, qi = g.qi, clip0 = g.clip0, rows = g.rows, nrb = g.nrb, RM = g.nrb * 16, rstride = g.rstride, lrows = g.lrows;
    const int a_iters = g.a_iters, nt = g.nt, ntt = g.ntt, nch = g.nch, col0 = g.col0, per_chunk = g.per_chunk, stage_f = g.stage_f, NS = g.NS, pre = g.pre;
    const bool has_res = g.has_res;
    const int t_in = op.t_in, t_out = g.t_part, stride = op.stride;      // t_out: conv positions per clip of THIS item
    const int in0 = g.t0 * stride;                     // first input row of the part (the staged image starts two rows earlier)
    // the tap offsets in scalar registers NOW: a read of the layer record inside the K loop would be a vector memory load, and waiting for
    // it (vmcnt counts in order) waits for every weight chunk in flight -- measured: 2 us per chunk, no overlap at all
    int shifts[kMaxTaps];
#pragma unroll
    for (int tp = 0; tp < kMaxTaps; ++tp) shifts[tp] = __builtin_amdgcn_readfirstlane(op.off[tp]);
    // ---- operand addresses of chunk 0 (16-byte units; unit u of a DMA pass lands at LDS position u: the global side is swizzled) ------
    const float inv_rstride = rcp_small(rstride);
    const float* a_src[kAIters];
    unsigned a_real = 0u;                              // bit it: this lane's unit of pass it is a real input row (else: halo / padding -> zeros)
#pragma unroll
    for (int it = 0; it < kAIters; ++it) {
        a_src[it] = p.zero_page;
        if (it < a_iters) {                            // (32-bit element offsets: an activation matrix has < 2^31 floats)
            const int u = it * 256 + tid, row = u >> 4, slot = (u & 15) ^ (row & 15);
            const int k = idiv_small(row, inv_rstride), t = in0 + row - k * rstride - 2;
            const bool real = row < lrows && t >= 0 && t < t_in;
            const unsigned off = (unsigned)(((clip0 + k) * t_in + t) * op.lda + slot * 4);
            a_src[it] = real ? op.A + off : p.zero_page + slot * 4;
            a_real |= real ? (1u << it) : 0u;
        }
    }
    const float* w_src[kMaxTaps + 1];
    weight_src(p, op, g, tid, w_src);
    auto dma_a = [&](int stage, int kc) __attribute__((always_inline)) {
        float* const Ab = smem + stage * stage_f;
#pragma unroll
        for (int it = 0; it < kAIters; ++it)
            if (it < a_iters)
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(a_src[it] + (((a_real >> it) & 1u) ? kc * kKC : 0)),
                                                 (__attribute__((address_space(3))) void*)(Ab + (it * 256 + wave * 64) * 4), 16, 0, 16);
    };
    auto dma = [&](int stage, int kc) __attribute__((always_inline)) { dma_a(stage, kc); dma_w(g, w_src, smem, stage, kc, wave); };

    // ---- fragment rows: conv position o = 16 r + li of the item -> staged row of its centre tap ------------------------------------
    int base[kMaxRB];
    {
        const float inv_t = rcp_small(t_out);
#pragma unroll
        for (int r = 0; r < kMaxRB; ++r) {
            base[r] = 2;
            if (r < nrb) {
                const int o = min(r * 16 + li, rows - 1);      // positions past the item's last repeat it (never stored, never counted)
                const int k = idiv_small(o, inv_t), tq = o - k * t_out;
                base[r] = k * rstride + 2 + tq * stride;
            }
        }
        // blocks beyond nrb that an instantiation with more row blocks still multiplies read block 0's rows again (never stored)
#pragma unroll
        for (int r = 1; r < kMaxRB; ++r)
            if (r >= nrb) base[r] = base[0];
    }
    f32x4 acc[kMaxRB], accr[kMaxRB];
#pragma unroll
    for (int r = 0; r < kMaxRB; ++r) acc[r] = accr[r] = f32x4{0.f, 0.f, 0.f, 0.f};

    stamp(1);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // (the previous item's stores: the counted waits below are about this item's DMAs only)
    __syncthreads();                                   // the previous item's partial tiles (same LDS) have been consumed
    for (int c = 0; c < pre; ++c) dma(c, c);
    const int kslot = 4 * wave + lg;                   // this lane's 16-byte slot of a 64-channel row: wave w contracts channels 16 w ..
    // The K loop, instantiated per (row blocks, taps, residual tap): with static trip counts the compiler can count the LDS reads in flight
    // (lgkmcnt) -- the fragments of tap t + 1 are requested before the MFMAs of tap t.  With run-time counts it emitted read - wait - 4 MFMAs
    // per block: ~1.1 us per chunk on top of the MFMAs (profiles/r6_r_*).  Row-block counts are rounded up to an instantiated one (the extra
    // blocks repeat the item's last row and are never stored).
    auto kl = [&](auto nrb_t, auto nt_t, auto res_t) __attribute__((always_inline)) {
        constexpr int NRB = decltype(nrb_t)::value, NT = decltype(nt_t)::value;
        constexpr bool RES = decltype(res_t)::value;
        constexpr int NTT = NT + (RES ? 1 : 0);
        int st = 0;                                    // stage of chunk kc
        for (int kc = 0; kc < nch; ++kc) {
            wait_vmcnt(min(pre - 1, nch - 1 - kc) * per_chunk);      // chunk kc has landed; the younger ones may still be on their way
            // ... for everybody, and everybody is done with chunk kc - 1 (its fragments were consumed by MFMAs).  A bare s_barrier:
            // __syncthreads() is a fence and waits for vmcnt(0) -- every chunk in flight -- first
            asm volatile("s_barrier" ::: "memory");
            const float* Ab = smem + st * stage_f;
            const float* Wb = Ab + a_iters * 16 * kKC;
            f32x4 fw[NTT], fa[2][NRB];
#pragma unroll
            for (int tp = 0; tp < NTT; ++tp) fw[tp] = *reinterpret_cast<const f32x4*>(Wb + lds_off64(tp * 16 + li, kslot));
#pragma unroll
            for (int r = 0; r < NRB; ++r) fa[0][r] = *reinterpret_cast<const f32x4*>(Ab + lds_off64(base[r] + (NT > 0 ? shifts[0] : 0), kslot));
            __builtin_amdgcn_sched_barrier(0);
            // the request for chunk kc + pre goes out BEHIND the first fragment reads: their LDS latency passes underneath its issue
            if (kc + pre < nch) { const int sn = st + pre; dma(sn >= NS ? sn - NS : sn, kc + pre); }      // pre == NS - 1: the stage of chunk kc - 1
            st = (st + 1 == NS) ? 0 : st + 1;
#pragma unroll
            for (int tp = 0; tp < NTT; ++tp) {
                if (tp + 1 < NTT) {
                    const int sh = (tp + 1 < NT) ? shifts[tp + 1] : 0;      // the residual 1x1 conv reads the centre row
#pragma unroll
                    for (int r = 0; r < NRB; ++r) fa[(tp + 1) & 1][r] = *reinterpret_cast<const f32x4*>(Ab + lds_off64(base[r] + sh, kslot));
                }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int jj = 0; jj < 4; ++jj)
#pragma unroll
                    for (int r = 0; r < NRB; ++r) {
                        if (RES && tp == NT) accr[r] = __builtin_amdgcn_mfma_f32_16x16x4f32(fw[tp][jj], fa[tp & 1][r][jj], accr[r], 0, 0, 0);
                        else acc[r] = __builtin_amdgcn_mfma_f32_16x16x4f32(fw[tp][jj], fa[tp & 1][r][jj], acc[r], 0, 0, 0);
                    }
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    };
    using I1 = std::integral_constant<int, 1>; using I2 = std::integral_constant<int, 2>; using I3 = std::integral_constant<int, 3>;
    using I5 = std::integral_constant<int, 5>; using I9 = std::integral_constant<int, 9>; using I10 = std::integral_constant<int, 10>;
    auto by_taps = [&](auto nrb_t) __attribute__((always_inline)) {
        if (nt == 5 && has_res) kl(nrb_t, I5{}, std::true_type{});
        else if (nt == 5) kl(nrb_t, I5{}, std::false_type{});
        else if (nt == 3) kl(nrb_t, I3{}, std::false_type{});
        else if (nt == 2) kl(nrb_t, I2{}, std::false_type{});
        else kl(nrb_t, I1{}, std::false_type{});
    };
    if (nrb <= 1) by_taps(I1{});
    else if (nrb == 2) by_taps(I2{});
    else if (nrb == 3) by_taps(I3{});
    else if (nrb <= 5) by_taps(I5{});
    else if (nrb <= 9) by_taps(I9{});
    else by_taps(I10{});
    __syncthreads();                                   // every wave is done with the staging buffers
    stamp(2);

    // ---- the four K slices meet: red[wave][row][16] ---------------------------------------------------------------------------------
    float* const red = smem;
    float* const redr = smem + kRed;
#pragma unroll
    for (int r = 0; r < kMaxRB; ++r)
        if (r < nrb) {
            *reinterpret_cast<f32x4*>(red + ((wave * RM + r * 16 + li) * 16 + lg * 4)) = acc[r];
            if (has_res) *reinterpret_cast<f32x4*>(redr + ((wave * RM + r * 16 + li) * 16 + lg * 4)) = accr[r];
        }
    __syncthreads();
    float* const rowst = smem + kStage;                // [160 rows][4 local groups][2]: (mean, M2) of the row's columns of the group
    float* const stat = rowst + kMaxRows * 8;          // [16 (clip, local group) pairs][2]: mean, M2
    constexpr int NU = 3;                              // 16-byte units per thread: rows x 4 <= 640
    const int c4 = (tid & 3) * 4;
    const bool gn = op.gn != 0;
    const float* const tb = (gn && op.tb_off >= 0) ? p.tb_row + op.tb_off + col0 + c4 : nullptr;
    const float inv_tout = rcp_small(t_out);
    int urow[NU], uclip[NU];
    bool uok[NU];
    // every operand of the epilogue is requested NOW (partner-written ones past the L1): they land underneath the sums and the statistics
    f32x4 r4[NU], a4[NU];
    const f32x4 zero4 = f32x4{0.f, 0.f, 0.f, 0.f};
    f32x4 g4 = zero4, be4 = zero4, t4 = zero4;
    const f32x4 b4 = *reinterpret_cast<const f32x4*>(op.bias + col0 + c4);
    f32x4 rb4 = zero4;
    if (has_res) rb4 = *reinterpret_cast<const f32x4*>(op.bres + col0 + c4);
    if (gn) {
        g4 = *reinterpret_cast<const f32x4*>(op.gamma + col0 + c4);
        be4 = *reinterpret_cast<const f32x4*>(op.beta + col0 + c4);
        if (tb) t4 = *reinterpret_cast<const f32x4*>(tb);
    }
#pragma unroll
    for (int i = 0; i < NU; ++i) {
        const int o = (tid + 256 * i) >> 2;
        uok[i] = o < rows;
        const int k = uok[i] ? idiv_small(o, inv_tout) : 0, tq = uok[i] ? o - k * t_out : 0;
        uclip[i] = k;
        urow[i] = (clip0 + k) * op.t_dst + (g.t0 + tq) * op.omul + op.oadd;
        r4[i] = a4[i] = zero4;
        if (uok[i] && gn) {
            if (op.res) r4[i] = ld16_l2(op.res + (size_t)urow[i] * op.ldres + col0 + c4);
            if (op.add2) a4[i] = ld16_l2(op.add2 + (size_t)urow[i] * op.ldadd2 + col0 + c4);
        }
    }
    f32x4 v[NU];
#pragma unroll
    for (int i = 0; i < NU; ++i) {
        const int o = (tid + 256 * i) >> 2;
        v[i] = zero4;
        if (uok[i]) {
            f32x4 s = *reinterpret_cast<const f32x4*>(red + ((0 * RM + o) * 16 + c4));
#pragma unroll
            for (int w = 1; w < 4; ++w) {
                const f32x4 t = *reinterpret_cast<const f32x4*>(red + ((w * RM + o) * 16 + c4));
#pragma unroll
                for (int x = 0; x < 4; ++x) s[x] += t[x];
            }
#pragma unroll
            for (int x = 0; x < 4; ++x) v[i][x] = s[x] + b4[x];
            if (has_res) {      // the 1x1 residual conv of the block's input, raw: the block's second layer adds it
                f32x4 sr = *reinterpret_cast<const f32x4*>(redr + ((0 * RM + o) * 16 + c4));
#pragma unroll
                for (int w = 1; w < 4; ++w) {
                    const f32x4 t = *reinterpret_cast<const f32x4*>(redr + ((w * RM + o) * 16 + c4));
#pragma unroll
                    for (int x = 0; x < 4; ++x) sr[x] += t[x];
                }
#pragma unroll
                for (int x = 0; x < 4; ++x) sr[x] += rb4[x];
                *reinterpret_cast<f32x4*>(op.res_out + (size_t)((clip0 + uclip[i]) * op.t_out + g.t0 + (o - uclip[i] * t_out)) * op.ld_res_out + col0 + c4) = sr;
            }
        }
    }

    stamp(3);
    if (!gn) {
#pragma unroll
        for (int i = 0; i < NU; ++i)
            if (uok[i]) *reinterpret_cast<f32x4*>(op.dst + (size_t)urow[i] * op.lddst + col0 + c4) = v[i];
        stamp(4);
        return;
    }

    // ---- GroupNorm over (group channels x T) per clip.  ONE pass, no cancellation: every unit's (mean, M2) of its 4 values, merged pairwise
    // (Chan) over the row's units of the group, the clip's rows, the 16 lanes of the pair -- each merge symmetric in its operands, so every lane
    // that holds a result holds the same bits, run after run
    const int gw = op.gn, gl = min(gw, 16), ngl = 16 / gl;       // local groups of this 16-column block: 4 / 2 / 1
    const int ugrp = (gl == 4) ? (tid & 3) : (gl == 8 ? ((tid & 3) >> 1) : 0);
    const int npairs = qi * ngl;                                 // <= 16: (clip, local group)
    auto merge_eq = [](float& m, float& q, float mb, float qb, float n_each) __attribute__((always_inline)) {
        const float d = mb - m;
        m = 0.5f * (m + mb);
        q = (q + qb) + 0.5f * n_each * d * d;
    };
#pragma unroll
    for (int i = 0; i < NU; ++i) {
        float m = 0.25f * ((v[i][0] + v[i][1]) + (v[i][2] + v[i][3])), q = 0.f;
#pragma unroll
        for (int x = 0; x < 4; ++x) { const float d = v[i][x] - m; q += d * d; }
        if (gl >= 8) { const float mb = __shfl_xor(m, 1), qb = __shfl_xor(q, 1); merge_eq(m, q, mb, qb, 4.f); }
        if (gl >= 16) { const float mb = __shfl_xor(m, 2), qb = __shfl_xor(q, 2); merge_eq(m, q, mb, qb, 8.f); }
        if (uok[i]) *reinterpret_cast<f32x2*>(rowst + (((tid + 256 * i) >> 2) * 4 + ugrp) * 2) = f32x2{m, q};
    }
    __syncthreads();
    {
        const int pi = tid >> 4, l16 = tid & 15;
        const float fgl = (float)gl;
        float r = 0.f, m = 0.f, q = 0.f;                         // rows merged so far, their mean, their M2
        if (pi < npairs) {
            const int k = idiv_small(pi, rcp_small(ngl)), g = pi - k * ngl;
            for (int t = l16; t < t_out; t += 16) {
                const f32x2 row = *reinterpret_cast<const f32x2*>(rowst + ((k * t_out + t) * 4 + g) * 2);
                const float d = row[0] - m, rn = r + 1.0f, inv = 1.0f / rn;
                m = (r * m + row[0]) * inv;
                q = (q + row[1]) + fgl * d * d * r * inv;
                r = rn;
            }
        }
#pragma unroll
        for (int sh = 8; sh >= 1; sh >>= 1) {
            const float rb = __shfl_xor(r, sh), mb = __shfl_xor(m, sh), qb = __shfl_xor(q, sh);
            const float n = r + rb, inv = n > 0.f ? 1.0f / n : 0.f, d = mb - m;
            m = (r * m + rb * mb) * inv;
            q = (q + qb) + fgl * d * d * (r * rb) * inv;
            r = n;
        }
        if (pi < npairs && l16 == 0) *reinterpret_cast<f32x2*>(stat + pi * 2) = f32x2{m, q};
    }
    __syncthreads();
    const int cparts = gw > 16 ? gw >> 4 : 1, parts = cparts * op.rsplit;
    if (parts > 1) {
        // the group spans `cparts` neighbouring column blocks and the clip `rsplit` row parts: parts = cparts x rsplit items (of the same
        // clip run, resident at the same time on other workgroups of the XCD) exchange (mean, M2) per (clip, local group) pair and merge them
        // in a fixed order -- every partner gets the same bits.  Granules {value, tag} by PLAIN 8-byte stores (they stay in the XCD's L2,
        // where the partners' device-scope loads find them)
        const int cpart = cb & (cparts - 1);
        const unsigned tag = p.tag_base + (unsigned)oi;
        if (tid < npairs) {
            unsigned long long* const mine = p.slots + (((size_t)xcd * kMaxItems + op.slot_base + id) * kMaxQ + tid) * 2;
            const float m = stat[tid * 2], q2 = stat[tid * 2 + 1];
            mine[0] = ((unsigned long long)tag << 32) | __float_as_uint(m);
            mine[1] = ((unsigned long long)tag << 32) | __float_as_uint(q2);
            asm volatile("" ::: "memory");
            const float n1 = (float)(t_out * gl);      // values behind one part's pair
            float mm = 0.f, qq = 0.f, nn = 0.f;          // merged so far: row parts outer, column parts inner, ascending
            for (int rp = 0; rp < op.rsplit; ++rp)
                for (int cp = 0; cp < cparts; ++cp) {
                    float mb = m, qb = q2;
                    if (!(rp == g.rp && cp == cpart)) {
                        const int pid = id + (rp - g.rp) * g.ncb + (cp - cpart);
                        const unsigned long long* theirs = p.slots + (((size_t)xcd * kMaxItems + op.slot_base + pid) * kMaxQ + tid) * 2;
                        unsigned long long a = 0ull, b = 0ull;
                        mb = 0.f; qb = 0.f;
                        if (wait_tag(theirs, tag, p.err, 0x80u | (unsigned)oi, &a) && wait_tag(theirs + 1, tag, p.err, 0x80u | (unsigned)oi, &b)) {
                            mb = __uint_as_float((unsigned)a); qb = __uint_as_float((unsigned)b);
                        }
                    }
                    const float nt2 = nn + n1, inv = 1.0f / nt2, d = mb - mm;      // Chan: (nn, mm, qq) + (n1, mb, qb)
                    mm = (nn * mm + n1 * mb) * inv;
                    qq = (qq + qb) + d * d * (nn * n1) * inv;
                    nn = nt2;
                }
            stat[tid * 2] = mm;
            stat[tid * 2 + 1] = qq;
        }
        __syncthreads();
    }
    const float inv_ng = 1.0f / (float)(op.t_out * gw);      // the whole group: all row parts, all its columns
#pragma unroll
    for (int i = 0; i < NU; ++i) {
        if (!uok[i]) continue;
        const int pr = (uclip[i] * ngl + ugrp) * 2;
        const float m = stat[pr], rstd = 1.0f / sqrtf(stat[pr + 1] * inv_ng + 1e-5f);
        f32x4 o;
#pragma unroll
        for (int x = 0; x < 4; ++x) o[x] = ((mishf((v[i][x] - m) * rstd * g4[x] + be4[x]) + t4[x]) + r4[i][x]) + a4[i][x];
        *reinterpret_cast<f32x4*>(op.dst + (size_t)urow[i] * op.lddst + col0 + c4) = o;
        if (op.dst2) *reinterpret_cast<f32x4*>(op.dst2 + (size_t)urow[i] * op.lddst2 + col0 + c4) = o;
    }
    stamp(4);
}

// Head Conv1d(32, c_traj, 1) + ancestral update + the padded copy of x_{t-1} for the next step (trajnet.py:158-161,
// gaussian_diffusion_trajnet.py:440-466) for the XCD's clips: one thread per (row, channel).
__device__ __forceinline__ void tail_op(const RParams& p, const int j, const int c_lo, const int nx, float* smem, const int tid) {
    __syncthreads();
    float* const ws = smem;      // [ctraj][hcin + 1]
    for (int i = tid; i < p.ctraj * p.hcin; i += 256) ws[(i / p.hcin) * (p.hcin + 1) + i % p.hcin] = p.hw[(size_t)(i / p.hcin) * p.ldhw + i % p.hcin];
    __syncthreads();
    const float* noise = (p.sigma == 0.f) ? nullptr : p.noise;
    const size_t m0 = (size_t)c_lo * p.T, n = (size_t)nx * p.T * p.ctraj;
    for (size_t e = (size_t)j * 256 + tid; e < n; e += (size_t)kWG * 256) {
        const size_t m = m0 + e / p.ctraj;
        const int c = (int)(e % p.ctraj);
        const size_t i = m * p.ctraj + c;
        const float* f = p.fin + m * p.ldfin;
        const float* wc = ws + c * (p.hcin + 1);
        float acc = 0.f;
        for (int k = 0; k < p.hcin; k += 4) {
            const f32x4 fv = ld16_l2(f + k);
            acc = fmaf(fv[0], wc[k], acc); acc = fmaf(fv[1], wc[k + 1], acc);
            acc = fmaf(fv[2], wc[k + 2], acc); acc = fmaf(fv[3], wc[k + 3], acc);
        }
        const float x0 = acc + p.hb[c];
        if (p.x0_out) p.x0_out[i] = x0;
        float vv = p.c1 * x0 + p.c2 * p.x[i];
        if (noise) vv += p.sigma * noise[i];
        p.x[i] = vv;
        p.xin[m * p.ldxin + c] = vv;
    }
}

__global__ __launch_bounds__(256) void traj_resident_kernel(RParams p) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int tid = threadIdx.x;
    const int xcd = blockIdx.x % kNumXCD, j = blockIdx.x / kNumXCD;
    const int c_lo = xcd * p.n_per_xcd, nx = min(p.B - c_lo, p.n_per_xcd);
    if (nx <= 0) return;
    if (__hip_atomic_load(p.err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u) return;      // an earlier step's wait expired: the host re-runs the loop
    const unsigned xcc1 = 1u + (unsigned)__builtin_amdgcn_s_getreg((3 << 11) | 20);
    // The layer records come from memory ONCE per launch, all of them in one round trip, into LDS: read one by one where they are used,
    // every layer started with ~1.4 us of scalar-load latency (a launch begins with cold caches; profiles/r6_u_*)
    unsigned* const lds_ops = reinterpret_cast<unsigned*>(smem + kStage + 1536);
    {
        const int words = p.n_ops * (int)(sizeof(ROp) / 4);
        const unsigned* src = reinterpret_cast<const unsigned*>(p.ops);
        for (int i = tid; i < words; i += 256) lds_ops[i] = src[i];
        __syncthreads();
    }
    constexpr int W = (int)(sizeof(ROp) / 4);
    static_assert(sizeof(ROp) % 16 == 0, "layer records are read 16 bytes at a time");
    typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
    auto load_op = [&](int oi) __attribute__((always_inline)) {      // a layer's record in scalar registers
        // all 16-byte reads first, then the broadcasts: word by word every read's LDS latency was exposed (~1.4 us per layer)
        constexpr int Q = W / 4;
        u32x4 q[Q];
#pragma unroll
        for (int i = 0; i < Q; ++i) q[i] = *reinterpret_cast<const u32x4*>(lds_ops + oi * W + i * 4);
        ROp o;
        unsigned* dst = reinterpret_cast<unsigned*>(&o);
#pragma unroll
        for (int i = 0; i < Q; ++i)
#pragma unroll
            for (int x = 0; x < 4; ++x) dst[i * 4 + x] = __builtin_amdgcn_readfirstlane(q[i][x]);
        return o;
    };
    int sync_idx = 0;
#pragma unroll 1
    for (int oi = 0; oi < p.n_ops; ++oi) {
        const ROp op = load_op(oi);
        unsigned long long* const tl = p.timeline ? p.timeline + ((size_t)blockIdx.x * kMaxOps + oi) * 8 : nullptr;
        if (tl != nullptr && threadIdx.x == 0) tl[0] = __builtin_amdgcn_s_memrealtime();
        if (op.kind == 1) {
            tail_op(p, j, c_lo, nx, smem, tid);
        } else {
            const int n_items = idiv_small(nx + op.q - 1, rcp_small(op.q)) * op.rsplit * (op.cout >> 4);
#pragma unroll 1
            for (int id = (j + kWG - op.wg_off) & (kWG - 1); id < n_items; id += kWG) {
                int t = tid;
                asm volatile("" : "+v"(t));      // per-item address arithmetic stays inside the item (no hoisting across the layer loop)
                conv_item(p, op, oi, id, xcd, c_lo, nx, smem, t, tl);
            }
        }
        if (op.sync_after) {
            xcd_sync(p.flags + ((size_t)xcd * kMaxSync + sync_idx) * kWG, j, p.tag_base + (unsigned)sync_idx, xcc1, p.err, (unsigned)sync_idx, tid,
                     p.fault != 0 && blockIdx.x == 0 && sync_idx == 3);
            ++sync_idx;
            if (tl != nullptr && threadIdx.x == 0) tl[5] = __builtin_amdgcn_s_memrealtime();
            if (__hip_atomic_load(p.err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u) return;
        }
    }
}

// ---- host: the step as a list of layers --------------------------------------------------------------------------------------------
struct Builder {
    const rohm_trajnet* h; const TWs& w; int B, T, n;
    std::vector<ROp> ops;
    bool ok = true;

    int pick_q(int t_in, int t_out, int ncb) const {
        int qmax = std::min(std::min(n, kMaxQ), std::min(kMaxRows / t_out, kMaxLdsRows / (t_in + 4)));
        if (qmax < 1) return 0;
        for (int q = 1; q <= qmax; ++q)
            if (((n + q - 1) / q) * ncb <= kWG) return q;      // fewest clips per item that keep the layer to one round of the XCD's workgroups
        const int groups = (n + qmax - 1) / qmax;
        return (n + groups - 1) / groups;
    }
    ROp conv(const ConvW& c, const float* A, int lda, int t_in, int t_out, int stride, int nt, const int* offs, float* dst, int lddst,
             int t_dst, int omul = 1, int oadd = 0) {
        ROp o;
        memset(&o, 0, sizeof(o));
        o.kind = 0; o.cout = c.cout; o.cin_pad = c.cin_pad; o.t_in = t_in; o.t_out = t_out; o.stride = stride; o.ntaps = nt;
        for (int j = 0; j < nt; ++j) o.off[j] = offs[j];
        o.lda = lda; o.ldw = (c.taps > 0 ? c.taps : 1) * c.cin_pad; o.omul = omul; o.oadd = oadd; o.t_dst = t_dst;
        o.tb_off = -1; o.A = A; o.W = c.w; o.bias = c.b; o.dst = dst; o.lddst = lddst; o.sync_after = 1;
        o.q = pick_q(t_in, t_out, c.cout / 16);
        o.rsplit = 1;
        if (o.q == 1) {      // one clip per item and workgroups to spare: deal the clip's conv positions to 2 .. 4 items (whole 16-row blocks where possible)
            const int items = n * (c.cout / 16);
            for (int rs : {3, 4, 2}) {
                if (items * rs > kWG || t_out % rs != 0 || t_out / rs < 16) continue;
                o.rsplit = rs;
                break;
            }
        }
        if (o.q < 1 || c.cout % 16 != 0 || c.cin_pad % kKC != 0 || ((n + o.q - 1) / o.q) * o.rsplit * (c.cout / 16) > kMaxItems) ok = false;
        return o;
    }
    // ResidualTemporalBlock (heads.py:43-54) as two layers
    void res_block(const ResW& r, const float* x, int ldx, int Tl, const float* add2, int ldadd2, float* dst, int lddst, float* dst2, int lddst2,
                   const Scratch& sc, std::vector<ROp>& out) {
        static const int offs5[5] = {-2, -1, 0, 1, 2};
        const int co = r.cout;
        ROp a = conv(r.b0.conv, x, ldx, Tl, Tl, 1, 5, offs5, sc.hb, co, Tl);
        a.gn = co / 8; a.gamma = r.b0.g; a.beta = r.b0.be; a.tb_off = r.tb_off;
        if (r.has_res) { a.Wres = r.res.w; a.bres = r.res.b; a.res_out = sc.rc; a.ld_res_out = co; }
        out.push_back(a);
        ROp b = conv(r.b1.conv, sc.hb, co, Tl, Tl, 1, 5, offs5, dst, lddst, Tl);
        b.gn = co / 8; b.gamma = r.b1.g; b.beta = r.b1.be;
        if (r.has_res) { b.res = sc.rc; b.ldres = co; } else { b.res = x; b.ldres = ldx; }
        b.add2 = add2; b.ldadd2 = ldadd2; b.dst2 = dst2; b.lddst2 = lddst2;
        out.push_back(b);
    }
    void down(const ConvW& c, const float* x, int ldx, int Tl, float* dst, int lddst, std::vector<ROp>& out) {      // Conv1d(k3, s2, p1): heads.py:72-78
        static const int offs[3] = {-1, 0, 1};
        out.push_back(conv(c, x, ldx, Tl, Tl / 2, 2, 3, offs, dst, lddst, Tl / 2));
    }
    void conv1(const ConvW& c, const float* x, int ldx, int Tl, float* dst, int lddst, std::vector<ROp>& out) {
        static const int offs[1] = {0};
        out.push_back(conv(c, x, ldx, Tl, Tl, 1, 1, offs, dst, lddst, Tl));
    }
    void up(const UpW& u, const float* x, int ldx, int Tq, float* dst, int lddst, std::vector<ROp>& out) {      // ConvTranspose1d(k4, s2, p1): heads.py:81-87
        static const int off_even[2] = {0, -1}, off_odd[2] = {1, 0};      // out[2t] = W1 x[t] + W3 x[t-1], out[2t+1] = W0 x[t+1] + W2 x[t]
        ROp e = conv(u.even, x, ldx, Tq, Tq, 1, 2, off_even, dst, lddst, 2 * Tq, 2, 0);
        e.sync_after = 0;
        out.push_back(e);
        out.push_back(conv(u.odd, x, ldx, Tq, Tq, 1, 2, off_odd, dst, lddst, 2 * Tq, 2, 1));
    }

    void build() {
        const int m = h->mid;
        const int ch[4] = {m / 8, m / 4, m / 2, m};
        std::vector<ROp> U, C;
        // U-Net encoder + first middle block (trajnet.py:211-237)
        const float* x = w.xin;
        int ldx = kPadC;
        for (int i = 0; i < 4; ++i) {
            const int Ti = T >> i;
            res_block(h->diff_enc[i], x, ldx, Ti, nullptr, 0, w.cat[i], 2 * ch[i], w.dcat[i] + ch[i], 2 * ch[i], w.sc, U);
            down(h->diff_down[i], w.cat[i], 2 * ch[i], Ti, w.ddn[i], 2 * ch[i], U);
            x = w.ddn[i]; ldx = 2 * ch[i];
        }
        const int T16 = T >> 4;
        res_block(h->mid_blk[0], w.ddn[3], 2 * m, T16, nullptr, 0, w.mid_a, m, nullptr, 0, w.sc, U);
        const size_t u_before_ctrl = U.size() + 1;      // + the first layer of the second middle block: its second layer adds ctrl_mid
        res_block(h->mid_blk[1], w.mid_a, m, T16, h->control ? w.ctrl_mid : nullptr, m, w.mid_b, m, nullptr, 0, w.sc, U);
        x = w.mid_b; ldx = m;
        for (int i = 3; i >= 0; --i) {
            const int Ti = T >> i;
            up(h->up[i], x, ldx, Ti / 2, w.dcat[i], 2 * ch[i], U);
            const int co = (i == 0) ? 32 : ch[i - 1];
            const int ldd = (i == 0) ? kPadC : co;
            res_block(h->dec[i], w.dcat[i], 2 * ch[i], Ti, h->control ? w.ctrl[i] : nullptr, co, w.d[i], ldd, nullptr, 0, w.sc, U);
            x = w.d[i]; ldx = ldd;
        }
        {   // head: Conv1dBlock(32, 32, k5), then Conv1d(32, c_traj, 1) + update (trajnet.py:158-161)
            static const int offs5[5] = {-2, -1, 0, 1, 2};
            ROp f = conv(h->final_blk.conv, w.d[0], kPadC, T, T, 1, 5, offs5, w.fin, kPadC, T);
            f.gn = 4; f.gamma = h->final_blk.g; f.beta = h->final_blk.be;
            U.push_back(f);
            ROp t;
            memset(&t, 0, sizeof(t));
            t.kind = 1; t.q = 1; t.cout = 16; t.rsplit = 1;
            U.push_back(t);
        }
        if (h->control) {      // ControlNet.forward (trajnet.py:43-75) behind control_zero_conv_0: depends on t and the conditions only
            const float* cx = w.cz;
            int ldc = kPadC;
            for (int i = 0; i < 4; ++i) {
                const int Ti = T >> i;
                res_block(h->c_enc[i], cx, ldc, Ti, nullptr, 0, w.ccat[i], 2 * ch[i], nullptr, 0, w.sc_ctl, C);
                conv1(h->c_zero[i], w.ccat[i], 2 * ch[i], Ti, w.ctrl[i], i == 0 ? 32 : ch[i - 1], C);
                C.back().sync_after = 0;
                down(h->c_down[i], w.ccat[i], 2 * ch[i], Ti, w.kdn[i], 2 * ch[i], C);
                cx = w.kdn[i]; ldc = 2 * ch[i];
            }
            res_block(h->c_mid[0], w.kdn[3], 2 * m, T16, nullptr, 0, w.kmid_a, m, nullptr, 0, w.sc_ctl, C);
            res_block(h->c_mid[1], w.kmid_a, m, T16, nullptr, 0, w.kmid_b, m, nullptr, 0, w.sc_ctl, C);
            conv1(h->c_zero_mid, w.kmid_b, m, T16, w.ctrl_mid, m, C);
        }
        // merge: slot k = [ControlNet slot k | U-Net slot k] while the U-Net does not need the control residuals yet
        size_t ui = 0, ci = 0;
        auto take_slot = [&](std::vector<ROp>& src, size_t& i, bool sync, int wg_off) {      // ops up to and including the next sync_after op
            int items = 0;
            while (i < src.size()) {
                ROp o = src[i++];
                const bool last = o.sync_after != 0;
                o.wg_off = (wg_off + items) & (kWG - 1) & ~3;
                o.slot_base = wg_off + items;
                items += o.kind == 0 ? ((n + o.q - 1) / o.q) * o.rsplit * (o.cout / 16) : 0;
                items = (items + 3) & ~3;
                if (wg_off + items > kMaxItems) ok = false;
                if (last) o.sync_after = sync ? 1 : 0;
                ops.push_back(o);
                if (last) break;
            }
            return items;
        };
        while (ci < C.size()) {
            const bool pair = ui < u_before_ctrl;
            const int used = take_slot(C, ci, !pair, 0);
            if (pair) {
                // count U ops consumed: a slot of U is one op here (the up-conv pairs come later)
                take_slot(U, ui, true, used);
            }
        }
        while (ui < U.size()) take_slot(U, ui, true, 0);
        int syncs = 0;
        for (const ROp& o : ops) syncs += o.sync_after;
        if ((int)ops.size() > kMaxOps || syncs > kMaxSync) ok = false;
    }
};

}  // namespace res

// resident scratch (floats): [ops][flags][slots][error word + pad][x checkpoint]
static size_t res_ops_floats() { return (sizeof(res::ROp) * res::kMaxOps + 255) / 256 * 64; }
static size_t res_flags_floats() { return (size_t)kNumXCD * res::kMaxSync * res::kWG * 2; }
static size_t res_slots_floats() { return (size_t)kNumXCD * res::kMaxItems * res::kMaxQ * 2 * 2; }

size_t resident_floats(int B, int T) {
    return res_ops_floats() + res_flags_floats() + res_slots_floats() + 64 + ((size_t)B * T * 32 + 63) / 64 * 64;
}

bool resident_ok(const rohm_trajnet* h, int B, int T, int n_steps, hipStream_t s) {
    // Default for TrajNet (no ControlNet branch), opt-in for TrajControl; ROHM_TRAJ_RESIDENT=0 / 1 switches it off / on for both.  Measured on
    // MI355X (profiles/r6_yd_*): 3e-6 from the launch-per-layer loop after 100 steps, the reference's goldens green; per 100-step TrajNet loop
    // 38.4 vs 41.3 ms at B = 1, 38.8 vs 46.1 at B = 8, 52.3 vs 55.8 at B = 32, 69.9 vs 73.4 at B = 64, 100 launches instead of 5 946.
    // TrajControl is 20-25 % SLOWER in this form (its ControlNet branch shares the U-Net's slots here and hides on a second stream there).
    const char* e = getenv("ROHM_TRAJ_RESIDENT");
    if (e && e[0] == '0') return false;
    if (!(e && e[0] == '1') && h->control) return false;
    if (n_steps < 1 || T % 16 != 0 || T > 160 || (T >> 4) < 1 || h->mid % 256 != 0 || h->final_conv.cin > 64 || h->final_conv.cin % 4 != 0) return false;
    const int n = (B + kNumXCD - 1) / kNumXCD;
    int bmax = 64;                     // clips per forward up to which the resident step is the faster plan (measured; ROHM_TRAJ_RESIDENT_MAX_B)
    if (const char* m = getenv("ROHM_TRAJ_RESIDENT_MAX_B")) bmax = atoi(m);
    if (B > bmax || n > res::kMaxQ) return false;
    // graph capture and the launch profiler's per-shape labels (rohm_profile_detail) belong to the launch-per-layer loop; the plain profiler
    // brackets the step's one launch (label conv_gemm/resident_step)
    if (prof::detail_requested() || stream_is_capturing(s)) return false;
    const char* why = nullptr;
    return exchange_layout_state(h->device, &why) == 1;
}

int resident_loop(const rohm_trajnet* h, const TWs& w, float* x, const float* noise, const int64_t* t_model, const float* coef,
                  float* x0_last, float* x_in_last, int n_steps, int B, int T, hipStream_t s) {
    using namespace res;
    float* base = w.resident;
    ROp* d_ops = reinterpret_cast<ROp*>(base);
    unsigned long long* flags = reinterpret_cast<unsigned long long*>(base + res_ops_floats());
    unsigned long long* slots = reinterpret_cast<unsigned long long*>(base + res_ops_floats() + res_flags_floats());
    unsigned* err = reinterpret_cast<unsigned*>(base + res_ops_floats() + res_flags_floats() + res_slots_floats());
    float* x_ckpt = base + res_ops_floats() + res_flags_floats() + res_slots_floats() + 64;
    const size_t M = (size_t)B * T, n = M * h->ctraj;

    Builder b{h, w, B, T, (B + kNumXCD - 1) / kNumXCD};
    b.build();
    if (!b.ok) return ROHM_ERR_UNSUPPORTED;
    // algorithmic work of a step for the launch profiler: the convolutions' multiply-adds; weights once + every activation written and read once
    double step_flops = 0.0, step_bytes = 0.0;
    for (const ROp& o : b.ops) {
        if (o.kind != 0) continue;
        const double rows = (double)B * o.t_out, k = (double)o.ntaps * o.cin_pad + (o.Wres ? o.cin_pad : 0);
        step_flops += 2.0 * rows * o.cout * k;
        step_bytes += 4.0 * (o.cout * k + (double)B * o.t_in * o.cin_pad + rows * o.cout);
    }
    // the meeting flags and statistics slots start from zero (a tag is never 0), the error word too; x_T is kept for a re-run
    ROHM_HIP_CHECK(hipMemsetAsync(flags, 0, (res_flags_floats() + res_slots_floats() + 64) * sizeof(float), s));
    ROHM_HIP_CHECK(hipMemcpyAsync(d_ops, b.ops.data(), b.ops.size() * sizeof(ROp), hipMemcpyHostToDevice, s));
    ROHM_HIP_CHECK(hipMemcpyAsync(x_ckpt, x, n * sizeof(float), hipMemcpyDeviceToDevice, s));
    // (the op list lives in pageable host memory: the copy above has been staged by the runtime when the call returns)

    static bool attr_set[64] = {};
    const size_t lds = (size_t)kLdsFloats * sizeof(float);
    if (h->device >= 0 && h->device < 64 && !attr_set[h->device]) {
        ROHM_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&traj_resident_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        attr_set[h->device] = true;
    }
    RParams p;
    memset(&p, 0, sizeof(p));
    p.ops = d_ops; p.n_ops = (int)b.ops.size(); p.B = B; p.T = T; p.n_per_xcd = b.n;
    p.zero_page = h->zero_page; p.flags = flags; p.slots = slots; p.err = err;
    p.fin = w.fin; p.ldfin = kPadC; p.hw = h->final_conv.w; p.ldhw = h->final_conv.cin_pad; p.hcin = h->final_conv.cin; p.hb = h->final_conv.b;
    p.ctraj = h->ctraj; p.x = x; p.xin = w.xin; p.ldxin = kPadC;
    { const char* f = getenv("ROHM_TRAJ_RESIDENT_FAULT"); p.fault = (f && f[0] == '1') ? 1 : 0; }
    const bool want_tl = getenv("ROHM_TRAJ_RESIDENT_TIMELINE") != nullptr;
    const size_t tl_words = (size_t)kNumXCD * kWG * kMaxOps * 8;
    unsigned long long* d_tl = nullptr;
    if (want_tl && hipMalloc(&d_tl, tl_words * 8) == hipSuccess) { (void)hipMemsetAsync(d_tl, 0, tl_words * 8, s); p.timeline = d_tl; }
    for (int i = 0; i < n_steps; ++i) {
        if (x_in_last && i == n_steps - 1) ROHM_HIP_CHECK(hipMemcpyAsync(x_in_last, x, n * sizeof(float), hipMemcpyDeviceToDevice, s));
        if (i % kTbSteps == 0) {
            const int run = (n_steps - i < kTbSteps) ? n_steps - i : kTbSteps;
            int rc = launch_time_path_steps(h, w, t_model + i, run, s);
            if (rc) return rc;
        }
        p.tb_row = w.tb_steps + (size_t)(i % kTbSteps) * h->tb_total;
        p.tag_base = (unsigned)(i + 1) << 8;
        p.c1 = coef[3 * i]; p.c2 = coef[3 * i + 1]; p.sigma = coef[3 * i + 2];
        p.noise = noise ? noise + (size_t)i * n : nullptr;
        p.x0_out = (x0_last && i == n_steps - 1) ? x0_last : nullptr;
        prof::set_step(i);
        {
            prof::Scope ps("conv_gemm/resident_step", step_flops, step_bytes, s);      // the step IS its convolutions: counted with the conv GEMMs
            hipLaunchKernelGGL(traj_resident_kernel, dim3(kNumXCD * kWG), dim3(256), lds, s, p);
        }
        ROHM_LAUNCH_CHECK();
    }
    // did every meeting complete?  (one host wait per loop call: 100 steps of work are behind it)
    unsigned host_err = 0u;
    ROHM_HIP_CHECK(hipMemcpyAsync(&host_err, err, sizeof(unsigned), hipMemcpyDeviceToHost, s));
    ROHM_HIP_CHECK(hipStreamSynchronize(s));
    if (d_tl) {      // the last step's stamps of XCD 0 (100 MHz ticks -> us): per layer, means over the workgroups that had an item
        std::vector<unsigned long long> tl(tl_words);
        (void)hipMemcpy(tl.data(), d_tl, tl_words * 8, hipMemcpyDeviceToHost);
        (void)hipFree(d_tl);
        unsigned long long t_first = ~0ull, t_last = 0ull;
        fprintf(stderr, "[rohm] resident TrajNet step, B = %d: layer  cout cin taps t_out items | setup  kloop  reduce  epi  | busy-wg span  meet(max)  layer span [us]\n", B);
        for (int oi = 0; oi < p.n_ops; ++oi) {
            const ROp& o = b.ops[oi];
            double su[4] = {0, 0, 0, 0}, meet_max = 0;
            int busy = 0;
            unsigned long long lo = ~0ull, hi = 0ull;
            for (int j = 0; j < kWG; ++j) {
                const unsigned long long* t = tl.data() + ((size_t)(j * kNumXCD) * kMaxOps + oi) * 8;      // block j * 8 = XCD 0, workgroup j
                if (t[0] == 0) continue;
                lo = std::min(lo, t[0]);
                const unsigned long long end = o.sync_after ? t[5] : (t[4] ? t[4] : t[0]);
                hi = std::max(hi, end);
                if (t[4] > t[0] && t[1] >= t[0]) {
                    ++busy;
                    su[0] += (double)(t[1] - t[0]); su[1] += (double)(t[2] - t[1]); su[2] += (double)(t[3] - t[2]); su[3] += (double)(t[4] - t[3]);
                }
                if (o.sync_after && t[5]) meet_max = std::max(meet_max, (double)(t[5] - std::max(t[4], t[0])));
            }
            if (lo == ~0ull) continue;
            t_first = std::min(t_first, lo); t_last = std::max(t_last, hi);
            const double d = busy ? 100.0 * busy : 1.0;
            const int items = o.kind == 0 ? ((b.n + o.q - 1) / o.q) * o.rsplit * (o.cout / 16) : 0;
            fprintf(stderr, "[rohm]   %2d %s %4d %4d %d %3d %3d | %5.2f %6.2f %6.2f %6.2f | %2d wgs  %6.2f  %6.2f%s\n", oi, o.kind ? "tail" : "conv", o.cout, o.cin_pad,
                    o.ntaps, o.t_out, items, su[0] / d, su[1] / d, su[2] / d, su[3] / d, busy, meet_max / 100.0, (double)(hi - lo) / 100.0, o.sync_after ? "" : "  (no meeting)");
        }
        fprintf(stderr, "[rohm]   step span %.2f us\n", (double)(t_last - t_first) / 100.0);
    }
    if (host_err != 0u) {
        ROHM_HIP_CHECK(hipMemcpyAsync(x, x_ckpt, n * sizeof(float), hipMemcpyDeviceToDevice, s));
        set_error("trajnet_sample_loop: a wait of the clip-resident step did not complete (%s, %s %u); the loop is re-run launch per layer",
                  (host_err & 0xffu) == 2u ? "partners on different XCDs" : "partner workgroups not co-resident or late",
                  ((host_err >> 8) & 0x80u) ? "statistics exchange of layer" : "meeting", (host_err >> 8) & 0x7fu);
        if (getenv("ROHM_TRAJ_RESIDENT_VERBOSE")) fprintf(stderr, "[rohm] resident TrajNet step fell back: error word 0x%x\n", host_err);
        return ROHM_ERR_UNSUPPORTED;
    }
    return ROHM_OK;
}

}  // namespace rohm
