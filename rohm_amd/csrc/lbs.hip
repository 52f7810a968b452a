// Full SMPL-X linear blend skinning: vertices [N, V, 3] (+ all J joints) of smplx.SMPLX.forward.
//
// Reference: third-party smplx==0.1.28 `lbs` (blend_shapes, vertices2joints, batch_rodrigues,
// batch_rigid_transform, skinning) as called from data_loaders/motion_representation.py:379-396 with
// `return_verts=True` (test_amass_full.py:283,405,415,425: the post-loop meshes).
// The denoising loops and the guidance never need vertices (smplx.hip); this is the mesh path.
//
//   lbs_pose_kernel   one thread per frame: Rodrigues, rest joints from the folded regressor, the 55-joint chain,
//                     relative transforms A[m][j] = [G_j | P_j - G_j J_j] and the pose feature (R_j - I, j >= 1).
//   gemm_f32_kernel   shape + pose blendshapes as ONE fp32-MFMA GEMM: v_posed[N, V*3] = [pose_feature(486) | beta(10) | 1] .
//                     [posedirs ; shapedirs ; v_template]  (K = 497 -> 512; 30.5 + 0.7 MFLOP per frame).
//   lbs_skin_mfma_kernel  DENSE skinning weights: the blend T = W[V, 55] . A[55, 12 F] is a dense contraction (13.8 MFLOP per
//                     frame, 660 fma per vertex-frame on the VALU before) -- on the matrix core: a workgroup keeps the 12 x 16
//                     transform rows of 16 frames in LDS, streams 144-vertex tiles of W through a double buffer, and applies
//                     v = T [p; 1] + transl in the epilogue.  Algorithmic traffic 12 B read + 12 B written per vertex-frame.
//   lbs_skin_ell_kernel   SPARSE skinning weights (a real SMPLX_NEUTRAL.npz: a handful of non-zero joints per vertex; chosen at
//                     rohm_smplx_set_skinning when >= 75 % of lbs_weights are zero): per vertex an ELL row (joint index, weight) of
//                     the widest vertex's length, one thread per vertex x 8 frames, transforms from LDS.
// Expression coefficients are taken as zero (every reference call site passes zeros, :383-388).
#include <stdlib.h>
#include <string.h>
#include <vector>
#include <type_traits>
#include "common.h"
#include "gemm_sched.h"
#include "smplx_fk.h"

namespace rohm {

constexpr int kMaxJ = 64;

__global__ __launch_bounds__(64) void lbs_pose_kernel(const float* __restrict__ pose, int n_pose, int pose_kind,
                                                      const float* __restrict__ betas, const float* __restrict__ Jt,
                                                      const float* __restrict__ Js, const int* __restrict__ parents,
                                                      int J, float* __restrict__ A, float* __restrict__ feat, int KP,
                                                      int N) {
    const int n = blockIdx.x * blockDim.x + threadIdx.x;
    if (n >= N) return;
    float beta[NBETA];
#pragma unroll
    for (int k = 0; k < NBETA; ++k) beta[k] = betas[(size_t)n * NBETA + k];
    float* An = A + (size_t)n * J * 12;          // pass 1: (G_j [9], P_j [3]); pass 2: P_j -> P_j - G_j J_j
    float* fn = feat + (size_t)n * KP;
    // columns (J-1)*9 .. +9: the shape coefficients, then a 1: the shape blend and v_template ride in the blendshape GEMM
    // (their rows follow posedirs in d_pdT)
    for (int k = (J - 1) * 9; k < KP; ++k) fn[k] = (k < (J - 1) * 9 + NBETA) ? beta[k - (J - 1) * 9] : (k == (J - 1) * 9 + NBETA ? 1.f : 0.f);
    auto rest = [&](int j, float* o) {
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            float v = Jt[j * 3 + c];
#pragma unroll
            for (int k = 0; k < NBETA; ++k) v = fmaf(Js[(j * 3 + c) * NBETA + k], beta[k], v);
            o[c] = v;
        }
    };
    for (int j = 0; j < J; ++j) {
        float R[9];
        if (j < n_pose && pose_kind == 1) {          // interleaved 6-D (quaternion.py:482-501), as the representation stores it
            float x6[6];
#pragma unroll
            for (int k = 0; k < 6; ++k) x6[k] = pose[((size_t)n * n_pose + j) * 6 + k];
            rot6d_fwd(x6, R);
        } else if (j < n_pose) {
            const float r[3] = {pose[((size_t)n * n_pose + j) * 3], pose[((size_t)n * n_pose + j) * 3 + 1],
                                pose[((size_t)n * n_pose + j) * 3 + 2]};
            rodrigues(r, R);
        } else {
#pragma unroll
            for (int i = 0; i < 9; ++i) R[i] = (i % 4 == 0) ? 1.f : 0.f;
        }
        if (j >= 1) {
#pragma unroll
            for (int i = 0; i < 9; ++i) fn[(j - 1) * 9 + i] = R[i] - ((i % 4 == 0) ? 1.f : 0.f);
        }
        float Jj[3];
        rest(j, Jj);
        float* o = An + j * 12;
        if (j == 0) {
#pragma unroll
            for (int i = 0; i < 9; ++i) o[i] = R[i];
#pragma unroll
            for (int c = 0; c < 3; ++c) o[9 + c] = Jj[c];
        } else {
            const int p = parents[j];
            const float* gp = An + p * 12;
            float Gp[9], Pp[3], Jp[3], G[9], w[3];
#pragma unroll
            for (int i = 0; i < 9; ++i) Gp[i] = gp[i];
#pragma unroll
            for (int c = 0; c < 3; ++c) Pp[c] = gp[9 + c];
            rest(p, Jp);
            mat_mul(Gp, R, G);
            const float off[3] = {Jj[0] - Jp[0], Jj[1] - Jp[1], Jj[2] - Jp[2]};
            mat_vec(Gp, off, w);
#pragma unroll
            for (int i = 0; i < 9; ++i) o[i] = G[i];
#pragma unroll
            for (int c = 0; c < 3; ++c) o[9 + c] = Pp[c] + w[c];
        }
    }
}

// joints out (posed + transl) and P_j -> t_j = P_j - G_j J_j; separate pass so children above read the parents' P_j
// Row of the skinning GEMM's transform operand that holds component `comp` (0..11) of frame `f`: groups of 16 frames = 192 rows; inside
// a group the row order makes one MFMA lane end up with all 12 components of ONE frame: wave w = f_in / 4 owns rows [48 w, 48 w + 48),
// column block c = comp / 4, lane group lg = f_in % 4, q = comp % 4  (see lbs_skin_mfma_kernel).
__host__ __device__ __forceinline__ size_t skin_row(int f, int comp) {
    const int g = f >> 4, fi = f & 15;
    return (size_t)g * 192 + (fi >> 2) * 48 + (comp >> 2) * 16 + (fi & 3) * 4 + (comp & 3);
}
constexpr int SKIN_K = 64;        // joints padded to two 32-wide K chunks (J <= 64)

__global__ __launch_bounds__(64) void lbs_finish_pose_kernel(const float* __restrict__ betas, const float* __restrict__ transl,
                                                             const float* __restrict__ Jt, const float* __restrict__ Js,
                                                             int J, float* __restrict__ A, float* __restrict__ Tm,
                                                             float* __restrict__ joints, int n_joints_out, int N) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= N * J) return;
    const int n = idx / J, j = idx % J;
    float* o = A + (size_t)idx * 12;
    float Jj[3];
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        float v = Jt[j * 3 + c];
#pragma unroll
        for (int k = 0; k < NBETA; ++k) v = fmaf(Js[(j * 3 + c) * NBETA + k], betas[(size_t)n * NBETA + k], v);
        Jj[c] = v;
    }
    if (joints && j < n_joints_out)
#pragma unroll
        for (int c = 0; c < 3; ++c) joints[((size_t)n * n_joints_out + j) * 3 + c] = o[9 + c] + transl[(size_t)n * 3 + c];
    float w[3];
    mat_vec(o, Jj, w);
#pragma unroll
    for (int c = 0; c < 3; ++c) o[9 + c] -= w[c];
    if (Tm) {      // the same 12 numbers as rows of the skinning GEMM's operand [frames x 12 (permuted), 64 joints]
#pragma unroll
        for (int c = 0; c < 12; ++c) Tm[skin_row(n, c) * SKIN_K + j] = o[c];
    }
}

// ---- skinning, dense weights: T = W . A on the matrix core ------------------------------------------------------------------
// C[v][(f, comp)] = sum_j W[v][j] A[f][j][comp]: M side = vertices (tiles of 144 rows), N side = 16 frames x 12 components = 192
// columns (wave w owns 48 of them = frames 4 w .. 4 w + 3), K = joints padded to 64 = two 32-wide chunks.  v_mfma_f32_16x16x4_f32
// with the weights on the MFMA "B" side, so lane (li = l & 15, lg = l >> 4) of wave w ends with
//     acc[r][c][q] = T_{4 c + q}(vertex m0 + 16 r + li, frame 16 g + 4 w + lg)        r = 0..8, c = 0..2, q = 0..3
// -- all twelve numbers of one (vertex, frame), thanks to the row order of the transform operand (skin_row) -- and applies
// v = T[0:9] p + T[9:12] + transl in registers.  A workgroup holds its frame group's transform rows in LDS (48 KB) for its
// whole life and streams vertex tiles of W (36 KB each, L2-resident: 2.7 MB in all) through two LDS buffers by LDS-DMA; the
// posed vertices p of the next tile are requested before the MFMAs of this one.  LDS images are [rows][32] per K chunk with the
// gemm_f32 swizzle (applied on the per-lane SOURCE address; the DMA writes lane-linear): conflict-free ds_read_b128 fragments.
constexpr int SKIN_BM = 144, SKIN_BN = 192;
__device__ __forceinline__ int skin_lds_off(int row, int slot) { return row * 32 + ((slot ^ ((row >> 1) & 7)) << 2); }

__global__ __launch_bounds__(256) void lbs_skin_mfma_kernel(const float* __restrict__ Wp, const float* __restrict__ Tm,
                                                            const float* __restrict__ vposed, int ldv,
                                                            const float* __restrict__ transl, int V, int N, int tiles_v,
                                                            int tiles_per_wg, float* __restrict__ verts) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* Bs = smem;                                   // [2 chunks][192 x 32]
    float* As = smem + 2 * SKIN_BN * 32;                // [2 buffers][2 chunks][144 x 32]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int li = lane & 15, lg = lane >> 4;
    const int wave_u = __builtin_amdgcn_readfirstlane(wave) * 64;
    const int g = blockIdx.x;                           // frame group
    const int t0 = blockIdx.y * tiles_per_wg;
    const int t1 = (t0 + tiles_per_wg < tiles_v) ? t0 + tiles_per_wg : tiles_v;
    if (t0 >= t1) return;
    // ---- LDS-DMA: unit u = 16 bytes; row = u / 8, slot = u % 8 XOR-ed on the source side
    auto dma_rows = [&](const float* base, int rows, float* dst_chunk0, int chunk_stride) {
        for (int u0 = 0; u0 < rows * 8; u0 += 256) {            // rows * 8 is a multiple of 256 for 192 rows; 144 rows: 4.5 passes
            const int u = u0 + tid;
            const int row = (u < rows * 8) ? (u >> 3) : 0;
            const int slot = (u & 7) ^ ((row >> 1) & 7);
            const bool live = (u0 + wave_u) < rows * 8;                                   // wave-uniform (rows * 8 % 64 == 0)
#pragma unroll
            for (int ch = 0; ch < 2; ++ch) {
                float* dst = dst_chunk0 + ch * chunk_stride + (u0 + wave_u) * 4;
                if (live)
                    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(base + (size_t)row * SKIN_K + ch * 32 + slot * 4),
                                                     (__attribute__((address_space(3))) void*)dst, 16, 0, 0);
            }
        }
    };
    dma_rows(Tm + (size_t)g * SKIN_BN * SKIN_K, SKIN_BN, Bs, SKIN_BN * 32);
    dma_rows(Wp + (size_t)t0 * SKIN_BM * SKIN_K, SKIN_BM, As, SKIN_BM * 32);
    const int f = g * 16 + wave * 4 + lg;               // this lane's frame
    const bool f_ok = f < N;
    float tr[3] = {0.f, 0.f, 0.f};
    if (f_ok) { tr[0] = transl[(size_t)f * 3]; tr[1] = transl[(size_t)f * 3 + 1]; tr[2] = transl[(size_t)f * 3 + 2]; }
    const float* vp_f = vposed + (size_t)(f_ok ? f : 0) * ldv;
    float* out_f = verts + (size_t)(f_ok ? f : 0) * V * 3;
    // posed vertices: a tile's 9 x 3 values per lane are requested at the top of the tile's iteration and used one iteration later (left to
    // the compiler the loads sink to their use and every tile pays their round trip in front of its stores)
    auto load_pv = [&](int t, float (&o)[9][3]) __attribute__((always_inline)) {
#pragma unroll
        for (int r = 0; r < 9; ++r) {
            const int v = t * SKIN_BM + r * 16 + li;
            const float* q = vp_f + (size_t)(v < V ? v : V - 1) * 3;
            o[r][0] = q[0]; o[r][1] = q[1]; o[r][2] = q[2];
        }
    };
    float pv[9][3];
    // Two accumulator sets: while tile t's 432 MFMAs run, tile t - 1's epilogue arithmetic (v = G p + t + transl: 135 VALU operations per
    // lane) is dealt out between them -- with one wave per SIMD the matrix pipe otherwise idles through every tile's epilogue (round 5:
    // pinning the fragment reads alone bought nothing, the kernel sat at 0.75 of its MFMA issue floor).  K = 64 = four k16 steps (2 chunks x
    // 2) of 12 fragment reads + 108 MFMAs; step s + 1's reads and a quarter of the previous tile's rows go underneath step s's MFMAs,
    // pinned with sched_group_barrier (one ds_read_b128 and <= 4 VALU per nine MFMAs); the quarter's (masked) stores follow the step.
    float fin[9][12];         // the previous tile's finished accumulators (copied out of the accumulator set: 108 moves per tile), as scalars
    float pv_prev[9][3];      // posed vertices of the previous tile (its epilogue runs now); `pv` = this tile's, landing under its MFMAs
    struct SkinFrag { f32x4 a[9], b[3]; };
    // rows [r_lo, r_hi) of a finished tile -> o3: the same expression order as the VALU kernels
    auto epi_rows = [&](const float (&a)[9][12], const float (&p)[9][3], int r_lo, int r_hi, float (&o3)[3][3]) __attribute__((always_inline)) {
#pragma unroll
        for (int r = 0; r < 9; ++r) {
            if (r < r_lo || r >= r_hi) continue;
#pragma unroll
            for (int c = 0; c < 3; ++c)      // T = a[r][0..11]: G row-major in [0, 9), t in [9, 12)
                o3[r - r_lo][c] = a[r][c * 3] * p[r][0] + a[r][c * 3 + 1] * p[r][1] + a[r][c * 3 + 2] * p[r][2] + a[r][9 + c] + tr[c];
        }
    };
    auto store_rows = [&](int t, int r_lo, int r_hi, const float (&o3)[3][3]) __attribute__((always_inline)) {
        if (!f_ok) return;
#pragma unroll
        for (int r = 0; r < 9; ++r) {
            if (r < r_lo || r >= r_hi) continue;
            const int v = t * SKIN_BM + r * 16 + li;
            if (v < V) {
                float* o = out_f + (size_t)v * 3;
                o[0] = o3[r - r_lo][0]; o[1] = o3[r - r_lo][1]; o[2] = o3[r - r_lo][2];
            }
        }
    };
    auto tile = [&](auto prev_tag, int t) __attribute__((always_inline)) {
        constexpr bool HAS_PREV = decltype(prev_tag)::value;   // is there a previous tile whose epilogue rides along?
        const int buf = (t - t0) & 1;
        // this tile's weights (DMA issued one iteration ago, or in the prologue) have landed; everyone is done with the other buffer
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (t + 1 < t1) dma_rows(Wp + (size_t)(t + 1) * SKIN_BM * SKIN_K, SKIN_BM, As + (buf ^ 1) * 2 * SKIN_BM * 32, SKIN_BM * 32);
        load_pv(t, pv);                                        // needed one tile from now
        f32x4 acc[9][3];
#pragma unroll
        for (int r = 0; r < 9; ++r)
#pragma unroll
            for (int c = 0; c < 3; ++c) acc[r][c] = f32x4{0.f, 0.f, 0.f, 0.f};
        const float* as = As + buf * 2 * SKIN_BM * 32;
        SkinFrag sf[2];
        auto skin_read = [&](SkinFrag& fr, int step) __attribute__((always_inline)) {
            const int ch = step >> 1, slot = (step & 1) * 4 + lg;
#pragma unroll
            for (int c = 0; c < 3; ++c) fr.b[c] = *reinterpret_cast<const f32x4*>(Bs + ch * SKIN_BN * 32 + skin_lds_off(wave * 48 + c * 16 + li, slot));
#pragma unroll
            for (int r = 0; r < 9; ++r) fr.a[r] = *reinterpret_cast<const f32x4*>(as + ch * SKIN_BM * 32 + skin_lds_off(r * 16 + li, slot));
        };
        auto skin_mma = [&](const SkinFrag& fr) __attribute__((always_inline)) {
#pragma unroll
            for (int r = 0; r < 9; ++r)
#pragma unroll
                for (int c = 0; c < 3; ++c)
#pragma unroll
                    for (int j = 0; j < 4; ++j) acc[r][c] = __builtin_amdgcn_mfma_f32_16x16x4f32(fr.b[c][j], fr.a[r][j], acc[r][c], 0, 0, 0);
        };
        skin_read(sf[0], 0);
        __builtin_amdgcn_sched_barrier(0);
        constexpr int kRowLo[5] = {0, 3, 5, 7, 9};             // the previous tile's rows whose epilogue rides under step 0 .. 3
#pragma unroll
        for (int step = 0; step < 4; ++step) {
            float o3[3][3];
            if (step + 1 < 4) skin_read(sf[(step + 1) & 1], step + 1);
            if constexpr (HAS_PREV) epi_rows(fin, pv_prev, kRowLo[step], kRowLo[step + 1], o3);
            skin_mma(sf[step & 1]);
            if (step + 1 < 4) SchedGroups<0, 12, 9, 12, 0, 0, HAS_PREV ? 4 : 0>::run();
            else SchedGroups<0, 12, 9, 0, 0, 1, HAS_PREV ? 4 : 0>::run();
            __builtin_amdgcn_sched_barrier(0);
            if constexpr (HAS_PREV) store_rows(t - 1, kRowLo[step], kRowLo[step + 1], o3);
            __builtin_amdgcn_sched_barrier(0);
        }
#pragma unroll
        for (int r = 0; r < 9; ++r)
#pragma unroll
            for (int c = 0; c < 3; ++c) {
#pragma unroll
                for (int q = 0; q < 4; ++q) fin[r][c * 4 + q] = acc[r][c][q];
                pv_prev[r][c] = pv[r][c];
            }
        // the next tile's DMA (issued above) landed under the MFMAs; the wait at the top of the next iteration covers it
    };
    tile(std::false_type{}, t0);
    for (int t = t0 + 1; t < t1; ++t) tile(std::true_type{}, t);
    // the last tile's epilogue has nobody to ride under
#pragma unroll
    for (int q = 0; q < 3; ++q) {
        float o3[3][3];
        epi_rows(fin, pv_prev, 3 * q, 3 * q + 3, o3);
        store_rows(t1 - 1, 3 * q, 3 * q + 3, o3);
    }
}

// ---- skinning, sparse weights (and the VALU reference form of the dense case) ------------------------------------------------
// One thread per vertex, SKIN_F consecutive frames per block: the vertex's ELL row (joint index, weight; `width` entries, zero
// weights as padding) is walked once, each weight blended into the SKIN_F transforms the thread accumulates; the frames' transforms
// A[n][j] are staged in LDS joint-major.  With width = J and identity indices this is the dense VALU kernel of round 3 (kept as
// ROHM_LBS_SKIN=ell for A/B runs and as the fallback for J > 64).  Sums run in ELL order with fmaf.
constexpr int SKIN_F = 8;
constexpr int SKIN_VPT = 4;       // vertex runs of 256 per block: the block's transform image is staged once for 1024 vertices

__global__ __launch_bounds__(256) void lbs_skin_ell_kernel(const int* __restrict__ ell_j, const float* __restrict__ ell_w, int width,
                                                           const float* __restrict__ vposed, int ldv,
                                                           const float* __restrict__ A, const float* __restrict__ transl, int J,
                                                           int V, int N, float* __restrict__ verts) {
    // [SKIN_F x 12 components][J]: lanes that ask for DIFFERENT joints (sparse rows) read different banks, lanes that ask for the
    // same one (dense rows walk j in step) get a broadcast; then transl [SKIN_F][3]
    extern __shared__ __attribute__((aligned(16))) float sA[];
    const int n0 = blockIdx.y * SKIN_F;
    const int nf = (N - n0 < SKIN_F) ? N - n0 : SKIN_F;
    float* st = sA + SKIN_F * 12 * J;
    for (int i = threadIdx.x; i < SKIN_F * J * 12; i += blockDim.x) {
        const int f = i / (J * 12), r = i % (J * 12), j = r / 12, c = r % 12;          // source order: coalesced reads of A[n][j][c]
        sA[(f * 12 + c) * J + j] = (f < nf) ? A[((size_t)(n0 + f) * J + j) * 12 + c] : 0.f;
    }
    for (int i = threadIdx.x; i < SKIN_F * 3; i += blockDim.x) st[i] = (i < nf * 3) ? transl[(size_t)n0 * 3 + i] : 0.f;
    __syncthreads();
    for (int run = 0; run < SKIN_VPT; ++run) {
        const int v = (blockIdx.x * SKIN_VPT + run) * blockDim.x + threadIdx.x;
        if (v >= V) break;
        float T[SKIN_F][12];
#pragma unroll
        for (int f = 0; f < SKIN_F; ++f)
#pragma unroll
            for (int i = 0; i < 12; ++i) T[f][i] = 0.f;
        for (int k = 0; k < width; ++k) {
            const float w = ell_w[(size_t)k * V + v];
            const float* a = sA + ell_j[(size_t)k * V + v];
#pragma unroll
            for (int f = 0; f < SKIN_F; ++f)
#pragma unroll
                for (int i = 0; i < 12; ++i) T[f][i] = fmaf(w, a[(f * 12 + i) * J], T[f][i]);
        }
#pragma unroll
        for (int f = 0; f < SKIN_F; ++f) {
            if (f < nf) {
                const int n = n0 + f;
                const float* q = vposed + (size_t)n * ldv + (size_t)v * 3;
                const float pq[3] = {q[0], q[1], q[2]};
#pragma unroll
                for (int c = 0; c < 3; ++c)
                    verts[((size_t)n * V + v) * 3 + c] =
                        T[f][c * 3] * pq[0] + T[f][c * 3 + 1] * pq[1] + T[f][c * 3 + 2] * pq[2] + T[f][9 + c] + st[f * 3 + c];
            }
        }
    }
}

// ELL rows from dense weights [V, J]: the non-zero joints of a vertex in ascending joint order, padded with (0, 0.0f) to `width`;
// stored entry-major ([width][V]) so a block's reads coalesce.  `width` = J with `dense` keeps every joint in place.
__global__ void lbs_build_ell_kernel(const float* __restrict__ w, int V, int J, int width, int dense, int* __restrict__ ell_j,
                                     float* __restrict__ ell_w) {
    const int v = blockIdx.x * blockDim.x + threadIdx.x;
    if (v >= V) return;
    int k = 0;
    for (int j = 0; j < J; ++j) {
        const float x = w[(size_t)v * J + j];
        if (dense || x != 0.f) {
            if (k < width) { ell_j[(size_t)k * V + v] = j; ell_w[(size_t)k * V + v] = x; }
            ++k;
        }
    }
    for (; k < width; ++k) { ell_j[(size_t)k * V + v] = 0; ell_w[(size_t)k * V + v] = 0.f; }
}
__global__ void lbs_pad_weights_kernel(const float* __restrict__ w, int V, int J, float* __restrict__ Wp, int rows) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;      // Wp [rows, 64]: zero beyond (V, J)
    if (i >= (long)rows * SKIN_K) return;
    const int v = (int)(i / SKIN_K), j = (int)(i % SKIN_K);
    Wp[i] = (v < V && j < J) ? w[(size_t)v * J + j] : 0.f;
}

// dst[c][r] = src[r][c] for r < R, c < Cc; dst rows padded to ldd (pre-zeroed)
__global__ void lbs_transpose_kernel(const float* __restrict__ src, int R, int Cc, float* __restrict__ dst, int ldd) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (long)R * Cc) return;
    const int r = (int)(i / Cc), c = (int)(i % Cc);
    dst[(size_t)c * ldd + r] = src[i];
}
__global__ void lbs_copy_shape_kernel(const float* __restrict__ src, int n_total, float* __restrict__ dst, long rows) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= rows * NBETA) return;
    dst[i] = src[(i / NBETA) * n_total + (i % NBETA)];
}

static inline size_t al64(size_t n) { return (n + 63) / 64 * 64; }

}  // namespace rohm

using namespace rohm;

extern "C" {

// d_pdT rows past posedirs: dst[n][P + k] = shapedirs[n][k] (k < 10), dst[n][P + 10] = v_template[n]  (n = v * 3 + c)
__global__ void lbs_shape_rows_kernel(const float* __restrict__ sd, const float* __restrict__ vt, long rows, float* __restrict__ dst,
                                      int ldd, int col0) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= rows * (NBETA + 1)) return;
    const long n = i / (NBETA + 1);
    const int k = (int)(i % (NBETA + 1));
    dst[(size_t)n * ldd + col0 + k] = (k < NBETA) ? sd[n * NBETA + k] : vt[n];
}

int rohm_smplx_set_skinning(rohm_smplx_t* h, const float* v_template, const float* shapedirs, int n_shape_total,
                            const float* posedirs, int n_pose_feat, const float* lbs_weights) {
    ROHM_ARG_CHECK(h && v_template && shapedirs && posedirs && lbs_weights, "smplx_set_skinning: null argument");
    ROHM_ARG_CHECK(n_pose_feat == (h->J - 1) * 9, "smplx_set_skinning: posedirs must have (J-1)*9 = %d rows (got %d)",
                   (h->J - 1) * 9, n_pose_feat);
    ROHM_ARG_CHECK(h->J <= kMaxJ && n_shape_total >= NBETA, "smplx_set_skinning: bad sizes");
    ROHM_HIP_CHECK(hipSetDevice(h->device));
    const int V = h->V, J = h->J;
    h->P = n_pose_feat;
    h->KP = (n_pose_feat + NBETA + 1 + 31) / 32 * 32;      // pose feature | betas | 1
    h->NP = (V * 3 + 383) / 384 * 384;
    h->tiles_v = (V + SKIN_BM - 1) / SKIN_BM;
    // sparse or dense skinning: counted on the host (the weights are given once)
    std::vector<float> hw((size_t)V * J);
    ROHM_HIP_CHECK(hipMemcpy(hw.data(), lbs_weights, hw.size() * sizeof(float), hipMemcpyDefault));
    size_t zeros = 0;
    int widest = 0;
    for (int v = 0; v < V; ++v) {
        int nz = 0;
        for (int j = 0; j < J; ++j) nz += hw[(size_t)v * J + j] != 0.f;
        zeros += (size_t)(J - nz);
        widest = nz > widest ? nz : widest;
    }
    const char* force = getenv("ROHM_LBS_SKIN");      // "mfma" | "ell" (the row keeps every joint) | "sparse": A/B runs and tests
    const bool sparse = force ? !strcmp(force, "sparse") : (zeros * 4 >= (size_t)V * J * 3 && widest <= 16);
    h->skin_mode = (force && !strcmp(force, "ell")) ? 2 : (sparse ? 1 : 0);      // 0 = MFMA, 1 = ELL of the non-zeros, 2 = ELL of all joints
    h->ell_width = h->skin_mode == 1 ? (widest + 3) / 4 * 4 : J;
    if (h->ell_width < 4) h->ell_width = 4;
    float *d_pd = nullptr, *d_w = nullptr, *d_sdin = nullptr;
    hipError_t e = hipSuccess;
    auto bad = [&](hipError_t err) {
        if (d_pd) (void)hipFree(d_pd);
        if (d_w) (void)hipFree(d_w);
        if (d_sdin) (void)hipFree(d_sdin);
        set_error("smplx_set_skinning: %s", hipGetErrorString(err));
        return ROHM_ERR_HIP;
    };
#define TRY(x) if ((e = (x)) != hipSuccess) return bad(e)
    if (!h->d_vt) TRY(hipMalloc(&h->d_vt, (size_t)V * 3 * sizeof(float)));
    if (!h->d_sd) TRY(hipMalloc(&h->d_sd, (size_t)V * 3 * NBETA * sizeof(float)));
    if (h->d_pdT) { (void)hipFree(h->d_pdT); h->d_pdT = nullptr; }
    TRY(hipMalloc(&h->d_pdT, (size_t)h->NP * h->KP * sizeof(float)));
    if (h->d_wT) { (void)hipFree(h->d_wT); h->d_wT = nullptr; }       // [tiles_v * 144, 64]: the MFMA kernel's weight operand
    TRY(hipMalloc(&h->d_wT, (size_t)h->tiles_v * SKIN_BM * SKIN_K * sizeof(float)));
    if (!h->d_zero_bias) {
        TRY(hipMalloc(&h->d_zero_bias, (size_t)h->NP * sizeof(float)));
        TRY(hipMemset(h->d_zero_bias, 0, (size_t)h->NP * sizeof(float)));
    }
    if (h->d_ell_j) { (void)hipFree(h->d_ell_j); h->d_ell_j = nullptr; }
    if (h->d_ell_w) { (void)hipFree(h->d_ell_w); h->d_ell_w = nullptr; }
    TRY(hipMalloc(&h->d_ell_j, (size_t)h->ell_width * V * sizeof(int)));
    TRY(hipMalloc(&h->d_ell_w, (size_t)h->ell_width * V * sizeof(float)));
    TRY(hipMalloc(&d_pd, (size_t)n_pose_feat * V * 3 * sizeof(float)));
    TRY(hipMalloc(&d_w, (size_t)V * J * sizeof(float)));
    TRY(hipMalloc(&d_sdin, (size_t)V * 3 * n_shape_total * sizeof(float)));
    TRY(hipMemcpy(h->d_vt, v_template, (size_t)V * 3 * sizeof(float), hipMemcpyDefault));
    TRY(hipMemcpy(d_sdin, shapedirs, (size_t)V * 3 * n_shape_total * sizeof(float), hipMemcpyDefault));
    TRY(hipMemcpy(d_pd, posedirs, (size_t)n_pose_feat * V * 3 * sizeof(float), hipMemcpyDefault));
    TRY(hipMemcpy(d_w, hw.data(), (size_t)V * J * sizeof(float), hipMemcpyDefault));
    TRY(hipMemset(h->d_pdT, 0, (size_t)h->NP * h->KP * sizeof(float)));
    {
        const long n = (long)V * 3 * NBETA;
        hipLaunchKernelGGL(lbs_copy_shape_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, 0, d_sdin, n_shape_total,
                           h->d_sd, (long)V * 3);
        const long n2 = (long)n_pose_feat * V * 3;     // posedirs [P, V*3] -> [V*3 (pad NP), KP]
        hipLaunchKernelGGL(lbs_transpose_kernel, dim3((unsigned)((n2 + 255) / 256)), dim3(256), 0, 0, d_pd, n_pose_feat,
                           V * 3, h->d_pdT, h->KP);
        const long n4 = (long)V * 3 * (NBETA + 1);     // + shapedirs and v_template as K rows P .. P + 10
        hipLaunchKernelGGL(lbs_shape_rows_kernel, dim3((unsigned)((n4 + 255) / 256)), dim3(256), 0, 0, h->d_sd, h->d_vt, (long)V * 3,
                           h->d_pdT, h->KP, n_pose_feat);
        const long n3 = (long)h->tiles_v * SKIN_BM * SKIN_K;
        hipLaunchKernelGGL(lbs_pad_weights_kernel, dim3((unsigned)((n3 + 255) / 256)), dim3(256), 0, 0, d_w, V, J, h->d_wT,
                           h->tiles_v * SKIN_BM);
        hipLaunchKernelGGL(lbs_build_ell_kernel, dim3((unsigned)((V + 255) / 256)), dim3(256), 0, 0, d_w, V, J, h->ell_width,
                           h->skin_mode == 1 ? 0 : 1, h->d_ell_j, h->d_ell_w);
    }
    TRY(hipDeviceSynchronize());
#undef TRY
    (void)hipFree(d_pd); (void)hipFree(d_w); (void)hipFree(d_sdin);
    return ROHM_OK;
}

int rohm_smplx_skinning_mode(const rohm_smplx_t* h) { return (h && h->d_pdT) ? h->skin_mode : -1; }

// workspace: pose feature [N, KP] | transforms A [N, J, 12] | posed vertices [N, NP] | transform rows of the skinning GEMM
// [ceil(N / 16) * 192, 64] (MFMA mode)
size_t rohm_smplx_lbs_workspace_bytes(const rohm_smplx_t* h, int N) {
    if (!h || N <= 0 || !h->d_pdT) return 0;
    const size_t Np = (size_t)(N + 143) / 144 * 144;      // the blendshape GEMM runs on whole 144-row tiles (its hot instantiation)
    return (al64(Np * h->KP) + al64((size_t)N * h->J * 12) + al64(Np * h->NP) +
            al64((size_t)((N + 15) / 16) * SKIN_BN * SKIN_K)) * sizeof(float);
}

int rohm_smplx_forward(const rohm_smplx_t* h, const float* pose, int n_pose, int pose_kind, const float* betas, const float* transl,
                       int N, float* joints, int n_joints_out, float* verts, void* ws, size_t ws_bytes,
                       rohm_stream_t stream) {
    ROHM_ARG_CHECK(h && pose && betas && transl && ws, "smplx_forward: null argument");
    ROHM_ARG_CHECK(h->d_pdT, "smplx_forward: call rohm_smplx_set_skinning first (posedirs / lbs_weights)");
    ROHM_ARG_CHECK(n_pose >= 1 && n_pose <= h->J, "smplx_forward: n_pose must be in [1, %d]", h->J);
    ROHM_ARG_CHECK(pose_kind == 0 || pose_kind == 1, "smplx_forward: pose_kind must be 0 (axis-angle) or 1 (6-D)");
    ROHM_ARG_CHECK(!joints || (n_joints_out >= 1 && n_joints_out <= h->J), "smplx_forward: n_joints_out must be in [1, %d]", h->J);
    ROHM_ARG_CHECK(((uintptr_t)ws % 256) == 0, "smplx_forward: workspace must be 256-byte aligned");
    if (N <= 0) return ROHM_OK;
    ROHM_ARG_CHECK(N <= 65535, "smplx_forward: at most 65535 frames per call (got %d)", N);
    ROHM_ARG_CHECK(ws_bytes >= rohm_smplx_lbs_workspace_bytes(h, N), "smplx_forward: workspace too small");
    hipStream_t s = (hipStream_t)stream;
    const size_t Np = (size_t)(N + 143) / 144 * 144;
    float* feat = (float*)ws;
    float* A = feat + al64(Np * h->KP);
    float* vposed = A + al64((size_t)N * h->J * 12);
    float* Tm = vposed + al64(Np * h->NP);
    const int groups = (N + 15) / 16;
    const bool mfma = verts && h->skin_mode == 0;
    {
        prof::Scope ps("lbs_pose", 0.0, 4.0 * N * (n_pose * 3 + 10 + h->J * 12 + h->KP), s);
        if (mfma)      // padded joints (columns J .. 63) and padded frames of the last group must be zero
            ROHM_HIP_CHECK(hipMemsetAsync(Tm, 0, (size_t)groups * SKIN_BN * SKIN_K * sizeof(float), s));
        hipLaunchKernelGGL(lbs_pose_kernel, dim3((N + 63) / 64), dim3(64), 0, s, pose, n_pose, pose_kind, betas, h->d_Jt, h->d_Js,
                           h->d_parents, h->J, A, feat, h->KP, N);
        const int nj = N * h->J;
        hipLaunchKernelGGL(lbs_finish_pose_kernel, dim3((nj + 63) / 64), dim3(64), 0, s, betas, transl, h->d_Jt, h->d_Js,
                           h->J, A, mfma ? Tm : nullptr, joints, n_joints_out, N);
        ROHM_LAUNCH_CHECK();
    }
    if (!verts) return ROHM_OK;
    GemmParams g{};      // v_posed = v_template + shapedirs . beta + posedirs . pose_feature
    g.A = feat; g.lda = h->KP; g.W = h->d_pdT; g.ldw = h->KP; g.C = vposed; g.ldc = h->NP;
    // whole tiles + a (zero) bias vector select the GEMM's hot instantiation (no edge masks, operands requested under the last K
    // chunk); the pad rows of `feat` hold whatever the workspace held: they only produce pad rows of v_posed, which nobody reads
    g.M = (int)Np; g.N = h->NP; g.K = h->KP; g.bias = h->d_zero_bias;
    int rc = launch_gemm(g, EPI_BIAS, s);
    if (rc) return rc;
    if (mfma) {
        prof::Scope ps("lbs_skin_mfma", 2.0 * N * h->V * (SKIN_K * 12 + 12), 24.0 * N * h->V, s);
        // >= ~1024 workgroups, each keeping its frame group's transforms for a run of vertex tiles
        // (a workgroup = 16 frames x a run of 144-vertex tiles: ~2300 of them keep the last, partial round of the 256 CUs short
        // while a run still amortises the 48 KB of transform rows it loads first; measured at 4576 frames: 1144 workgroups 837 us)
        int chunks = (2304 + groups - 1) / groups;
        chunks = chunks < 1 ? 1 : (chunks > h->tiles_v ? h->tiles_v : chunks);
        const int per = (h->tiles_v + chunks - 1) / chunks;
        chunks = (h->tiles_v + per - 1) / per;
        const size_t lds = (size_t)(2 * SKIN_BN * 32 + 4 * SKIN_BM * 32) * sizeof(float);      // 120 KB
        static bool attr_set[64] = {};
        int dev = 0;
        ROHM_HIP_CHECK(hipGetDevice(&dev));
        if (dev < 64 && !attr_set[dev]) {
            ROHM_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&lbs_skin_mfma_kernel),
                                               hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
            attr_set[dev] = true;
        }
        hipLaunchKernelGGL(lbs_skin_mfma_kernel, dim3(groups, chunks), dim3(256), lds, s, h->d_wT, Tm, vposed, h->NP, transl, h->V, N,
                           h->tiles_v, per, verts);
        ROHM_LAUNCH_CHECK();
        return ROHM_OK;
    }
    prof::Scope ps(h->skin_mode == 1 ? "lbs_skin_sparse" : "lbs_skin_ell", 2.0 * N * h->V * (h->ell_width * 12 + 12), 24.0 * N * h->V, s);
    const dim3 sgrid((h->V + 256 * SKIN_VPT - 1) / (256 * SKIN_VPT), (N + SKIN_F - 1) / SKIN_F);
    const size_t slds = (size_t)SKIN_F * (h->J * 12 + 3) * sizeof(float);
    hipLaunchKernelGGL(lbs_skin_ell_kernel, sgrid, dim3(256), slds, s, h->d_ell_j, h->d_ell_w, h->ell_width, vposed, h->NP, A, transl,
                       h->J, h->V, N, verts);
    ROHM_LAUNCH_CHECK();
    return ROHM_OK;
}

}  // extern "C"
