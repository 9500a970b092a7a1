#include <hip/hip_runtime.h>
#include <cstdio>
// one-shot grid barrier: every block arrives once; generation g makes the counter monotonic (target = g * nblocks)
#define NGRP 32
// ctr[0..NGRP): first-level arrival counters (one 64-byte line each, index * 16), ctr[16*NGRP]: second level, ctr[16*NGRP+16]: flag
__device__ __forceinline__ bool grid_barrier(unsigned* ctr, unsigned gen) {
    __syncthreads();
    bool ok = true;
    if (threadIdx.x == 0) {
        const unsigned nblk = gridDim.x, bid = blockIdx.x;
        const unsigned grp = bid % NGRP, gsz = (nblk - grp + NGRP - 1) / NGRP;
        unsigned* flag = ctr + 16 * NGRP + 16;
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        const unsigned o1 = __hip_atomic_fetch_add(ctr + 16 * grp, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        bool last = false;
        if (o1 == gsz - 1) {
            __hip_atomic_store(ctr + 16 * grp, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const unsigned o2 = __hip_atomic_fetch_add(ctr + 16 * NGRP, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (o2 == NGRP - 1) {
                __hip_atomic_store(ctr + 16 * NGRP, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                __hip_atomic_store(flag, gen, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
                last = true;
            }
        }
        if (!last) {
            int spins = 0;
            while (__hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != gen) {
                __builtin_amdgcn_s_sleep(4);
                if (++spins > (1 << 21)) { ok = false; break; }
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    }
    __syncthreads();
    return ok;
}
__global__ void __launch_bounds__(256, 4) k(float* part, float* out, int n, unsigned* ctr, unsigned target) {
    const int b = blockIdx.x;
    if (threadIdx.x == 0) __hip_atomic_store(part + b, (float)(b + 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    grid_barrier(ctr, target);
    float s = 0.f;
    for (int i = threadIdx.x; i < n; i += 256) s += __hip_atomic_load(part + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __shared__ float red[256];
    red[threadIdx.x] = s; __syncthreads();
    if (threadIdx.x == 0) { float t = 0; for (int i = 0; i < 256; ++i) t += red[i]; out[b] = t; }
}
int main() {
    const int n = 808;
    float *part, *out; unsigned* ctr;
    (void)hipMalloc(&part, n * 4); (void)hipMalloc(&out, n * 4); (void)hipMalloc(&ctr, 4096); (void)hipMemset(ctr, 0, 4096);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    for (int it = 1; it <= 5; ++it) {
        (void)hipEventRecord(e0, 0);
        k<<<n, 256>>>(part, out, n, ctr, (unsigned)it);
        (void)hipEventRecord(e1, 0);
        (void)hipDeviceSynchronize();
        float ms; (void)hipEventElapsedTime(&ms, e0, e1);
        float h[808]; (void)hipMemcpy(h, out, n * 4, hipMemcpyDeviceToHost);
        printf("out0 %.0f out807 %.0f (expect %.0f)  %.1f us\n", h[0], h[807], 808.0 * 809 / 2, ms * 1e3);
    }
    return 0;
}
