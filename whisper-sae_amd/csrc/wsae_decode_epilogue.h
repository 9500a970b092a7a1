// Block epilogue shared by the decode kernels: write this block's partial sums (loss, l0, and the [D] column sums
// of g) and let the LAST block to arrive reduce the loss / l0 partials in fixed order into the stats record.
// Hand-off form (cdna guide, Guideline 16 / microarch "valid forms"): the 4-byte partials are agent-scope atomic
// (sc1, write-through) stores, drained with vmcnt(0) before the ticket add, and read back with agent-scope atomic
// loads -- no fences, no separate launch.  The [D] column sums are consumed by a later kernel, so plain stores do
// for them.  `red` = 8 floats, `flag` = 1 int of LDS (the caller's: a static __shared__ here would shift the
// dynamic LDS base of kernels that need it 16-byte aligned).
#pragma once

#include "wsae_common.h"

// NW = waves per block (the per-wave column sums in dbd_s are [NW][D])
template <bool BWD, int NW = 4>
__device__ __forceinline__ void decode_block_epilogue(float loss_acc, int l0_acc, const float* dbd_s, int D, int B,
                                                      int loss_cols, float* red, int* flag, float* part_loss, float* part_l0,
                                                      float* part_dbd, int32_t* ticket, wsae_stats* stats) {
    const int lane = threadIdx.x & 63;
    const float bl = block_sum(loss_acc, red);
    const float b0 = block_sum(lane == 0 ? (float)l0_acc : 0.f, red);
    if (threadIdx.x == 0) {  // write-through (sc1) stores: visible to the last arriver without a release fence
        __hip_atomic_store(part_loss + blockIdx.x, bl, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(part_l0 + blockIdx.x, b0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    if (BWD) {
        for (int d = threadIdx.x; d < D; d += 64 * NW) {
            float a = (dbd_s[d] + dbd_s[D + d]) + (dbd_s[2 * D + d] + dbd_s[3 * D + d]);
            if constexpr (NW == 8) a += (dbd_s[4 * D + d] + dbd_s[5 * D + d]) + (dbd_s[6 * D + d] + dbd_s[7 * D + d]);
            part_dbd[(int64_t)blockIdx.x * D + d] = a;
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0) {
        unsigned unused;
        *flag = grid_ticket((unsigned long long*)ticket, 0u, &unused);
    }
    __syncthreads();
    if (!*flag) return;
    float a = 0.f, c = 0.f;
    for (int i = threadIdx.x; i < (int)gridDim.x; i += 64 * NW) {
        a += __hip_atomic_load(part_loss + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        c += __hip_atomic_load(part_l0 + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    const float ta = block_sum(a, red);
    const float tc = block_sum(c, red);
    if (threadIdx.x == 0) {
        stats->loss = ta / ((float)B * (float)loss_cols);  // mean over the REAL output columns (transcoders may pad D)
        stats->l0 = tc / (float)B;
    }
}
