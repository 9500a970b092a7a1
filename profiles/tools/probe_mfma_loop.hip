// How fast does the "24 x (ds_read_b128 ahead, counted wait, v_mfma_f32_32x32x16_bf16)" slab loop of the streaming /
// filter encoder experiments run when nothing else is in the kernel?  Two waves per SIMD (256 threads, 2 blocks per CU),
// operands resident in LDS, no DMA, no stores, no barrier.  Variants: MODE 0 = MFMAs only (B fragment in registers),
// 1 = asm reads 6 ahead + counted waits, 2 = plain C++ LDS reads (compiler-scheduled), 3 = as 1 plus a workgroup
// barrier per slab.   hipcc --offload-arch=gfx950 -O3 probe_mfma_loop.hip -o probe_mfma_loop && ./probe_mfma_loop
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

__device__ __forceinline__ bf16x8 lds_read16(uint32_t addr, int imm) {
    bf16x8 r;
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(r) : "v"(addr), "n"(imm));
    return r;
}
__device__ __forceinline__ void lds_wait_for(bf16x8& w, int newer) {
    switch (newer) {
        case 0: asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(w)); break;
        case 1: asm volatile("s_waitcnt lgkmcnt(1)" : "+v"(w)); break;
        case 2: asm volatile("s_waitcnt lgkmcnt(2)" : "+v"(w)); break;
        case 3: asm volatile("s_waitcnt lgkmcnt(3)" : "+v"(w)); break;
        case 4: asm volatile("s_waitcnt lgkmcnt(4)" : "+v"(w)); break;
        case 5: asm volatile("s_waitcnt lgkmcnt(5)" : "+v"(w)); break;
        default: asm volatile("s_waitcnt lgkmcnt(6)" : "+v"(w)); break;
    }
}

__device__ __forceinline__ void glds16s(const void* base, uint32_t voff, uint32_t lds_addr) {
    uint32_t keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(voff), "s"(base), "s"(lds_addr) : "memory");
}
template <int N>
__device__ __forceinline__ void vm_wait() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

// MODE 4: + the LDS-DMA ring (6 pieces per wave and slab from an L2-resident 2.4 MB matrix, counted vmcnt)
// MODE 5: as 4 but the DMA'd bytes come from ONE 24 KB slab (every workgroup re-reads the same lines)
// MODE 6: as 4 without the MFMAs / reads (DMA + barrier only)
template <int MODE>
__global__ void __launch_bounds__(256, 2) loop_kernel(float* out, int slabs, const char* W = nullptr) {
    constexpr int KS = 24, CPR = 48, SLAB = 32 * 384 * 2;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63, m = lane & 31, h = lane >> 5;
    for (int i = threadIdx.x; i < 3 * SLAB / 4; i += 256) ((uint32_t*)smem)[i] = 0x3c003c00u + (i & 0xff);
    __syncthreads();
    const uint32_t smem_lds = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) char*)smem;
    uint32_t a_addr[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) a_addr[j] = smem_lds + (uint32_t)m * (CPR * 16) + (uint32_t)(((2 * j) ^ ((h ^ m) & 15)) & 15) * 16u;
    bf16x8 xf[KS];
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) xf[ks] = *(const bf16x8*)(smem + ((m * CPR + ks) * 16) % (3 * SLAB));
    f32x16 tot;
#pragma unroll
    for (int r = 0; r < 16; ++r) tot[r] = 0.f;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    uint32_t dma_off[6];
#pragma unroll
    for (int j = 0; j < 6; ++j) {
        const int p = 64 * (wave + 4 * j) + lane;
        const int r = p / CPR, cpos = p % CPR;
        dma_off[j] = (uint32_t)(r * 768 + ((cpos & ~15) | ((cpos ^ r) & 15)) * 16);
    }
    auto dma = [&](int s) {
        const char* wb = W + (MODE == 5 ? 0 : (size_t)((s + blockIdx.x * 7) % 96) * SLAB);
        const uint32_t slot = smem_lds + (uint32_t)(s % 3) * SLAB;
#pragma unroll
        for (int j = 0; j < 6; ++j) glds16s(wb, dma_off[j], slot + (uint32_t)(wave + 4 * j) * 1024u);
    };
    if (MODE >= 4) { dma(0); dma(1); }
    for (int s = 0; s < slabs; ++s) {
        if (MODE == 3) __builtin_amdgcn_s_barrier();
        if (MODE >= 4) {
            vm_wait<6>();
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            dma(s + 2);
            if (MODE == 6) continue;
        }
        const int SO = 0;  // (all modes read slot 0: what is read does not matter, only when)
        f32x16 acc;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = 0.f;
        if constexpr (MODE == 0) {
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(xf[ks], xf[(ks + 1) % KS], acc, 0, 0, 0);
        } else if constexpr (MODE == 2) {
            const char* sl = smem + (s % 3) * SLAB;
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
                const bf16x8 w = *(const bf16x8*)(sl + (a_addr[ks & 7] - smem_lds) + (ks >> 3) * 256);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(xf[ks], w, acc, 0, 0, 0);
            }
        } else {
            constexpr int AH = 6, R = AH + 2;
            bf16x8 w[R];
#pragma unroll
            for (int ks = 0; ks < AH; ++ks) w[ks] = lds_read16(a_addr[ks & 7], SO + (ks >> 3) * 256);
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
                if (ks + AH < KS) w[(ks + AH) % R] = lds_read16(a_addr[(ks + AH) & 7], SO + ((ks + AH) >> 3) * 256);
                lds_wait_for(w[ks % R], (KS - 1 - ks) < AH ? (KS - 1 - ks) : AH);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(xf[ks], w[ks % R], acc, 0, 0, 0);
            }
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) tot[r] += acc[r];
    }
    float v = 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) v += tot[r];
    if (v == 12345.f) out[threadIdx.x] = v;
}

template <int MODE>
void run(const char* name) {
    float* out; hipMalloc(&out, 4096);
    char* W; hipMalloc(&W, 97 * 32 * 384 * 2); hipMemset(W, 0x3c, 97 * 32 * 384 * 2);
    const int slabs = 2000, blocks = 512;
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    loop_kernel<MODE><<<blocks, 256, 3 * 32 * 384 * 2>>>(out, 10, W);
    hipEventRecord(a);
    loop_kernel<MODE><<<blocks, 256, 3 * 32 * 384 * 2>>>(out, slabs, W);
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    // per SIMD: 2 waves x slabs x 24 MFMAs
    printf("%-44s %.3f ms: %.1f ns per slab pair (ideal 640 at 2.4 GHz), %.0f TFLOP/s\n", name, ms, ms * 1e6 / slabs,
           1.0 * blocks * 4 * slabs * 24 * 32768.0 / (ms * 1e-3) / 1e12);
}

int main() {
    run<0>("MFMAs only");
    run<1>("asm reads 6 ahead + counted waits");
    run<2>("plain LDS reads");
    run<3>("asm reads + barrier per slab");
    run<4>("... + LDS-DMA ring (2.4 MB matrix)");
    run<5>("... + LDS-DMA ring (one 24 KB slab)");
    run<6>("LDS-DMA ring + barrier only");
    return 0;
}
