// Two helpers shared by the fp32-MFMA GEMM kernels (gemm_f32.hip, encoder_chain.hip): the bank-swizzled LDS image of a K chunk and
// the compile-time repetition of sched_group_barrier triples that pins their [LDS read | DMA | MFMA] interleave.
#pragma once
#include "common.h"

namespace rohm {

constexpr int kGemmBK = 32;      // K chunk

__device__ __forceinline__ int lds_off(int row, int slot) {   // float index inside a [rows][32] tile
    return row * kGemmBK + ((slot ^ ((row >> 1) & 7)) << 2);
}

// compile-time repetition of sched_group_barrier triples (the builtin needs literal arguments)
// VAL > 0 (conv gather): the per-chunk address arithmetic of the gathered operand (VALU) is dealt out between the MFMA
// groups as well -- left alone the scheduler hoists all of it in front of the half's first MFMA.
template <int I, int N, int MF, int DS_TOTAL, int VM_TOTAL, int ID, int VAL = 0>
struct SchedGroups {
    static __device__ __forceinline__ void run() {
        constexpr int kValu = 0x002, kMfma = 0x008, kVmem = 0x010, kDsRead = 0x100;
        __builtin_amdgcn_sched_group_barrier(kMfma, MF, ID);
        __builtin_amdgcn_sched_group_barrier(kDsRead, DS_TOTAL / N + (I < DS_TOTAL % N ? 1 : 0), ID);
        if constexpr (VAL > 0) __builtin_amdgcn_sched_group_barrier(kValu, VAL, ID);
        __builtin_amdgcn_sched_group_barrier(kVmem, VM_TOTAL / N + (I < VM_TOTAL % N ? 1 : 0), ID);
        SchedGroups<I + 1, N, MF, DS_TOTAL, VM_TOTAL, ID, VAL>::run();
    }
};
template <int N, int MF, int DS_TOTAL, int VM_TOTAL, int ID, int VAL>
struct SchedGroups<N, N, MF, DS_TOTAL, VM_TOTAL, ID, VAL> {
    static __device__ __forceinline__ void run() {}
};


}  // namespace rohm
