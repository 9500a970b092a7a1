// Would ONE launch for "slab reduction + optimizer" beat the two launches of the step (grad_finish 24 us + update_rows 22.5 us)?
// Mock of the memory pattern at H = 3072, D = 384, 8 split-K slabs: one wave per feature h,
//   phase A: read the 8 slab copies of gradient rows h (dW_e) and H + h (dW_dT), sum, write the gradient rows, norm partial;
//            issue the loads of p, m, v of both rows;
//   grid barrier (ticket + epoch flag; every workgroup is resident: 256 workgroups of 12 waves on 256 CUs);
//   phase B: AdamW-shaped arithmetic on the registers, write p, m, v and a bf16 copy of p.
// Each launch is preceded by a 400 MB fill that evicts the working set from the Infinity Cache; a variant leaves the slabs
// "just written" (a kernel that writes them right before, as the contraction does in the step).
//   hipcc --offload-arch=gfx950 -O3 -o probe_fused_tail probe_fused_tail.hip && ./probe_fused_tail
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

constexpr int H = 3072, D = 384, NS = 8, NC = D / 4;  // NC float4 chunks per row

__device__ __forceinline__ float4 upd(float4 p, float4 g, float4& m, float4& v, float gs) {
    g.x *= gs; g.y *= gs; g.z *= gs; g.w *= gs;
    m.x += (g.x - m.x) * 0.1f; m.y += (g.y - m.y) * 0.1f; m.z += (g.z - m.z) * 0.1f; m.w += (g.w - m.w) * 0.1f;
    v.x = 0.999f * v.x + 0.001f * g.x * g.x; v.y = 0.999f * v.y + 0.001f * g.y * g.y;
    v.z = 0.999f * v.z + 0.001f * g.z * g.z; v.w = 0.999f * v.w + 0.001f * g.w * g.w;
    p.x -= 1e-4f * m.x * __builtin_amdgcn_rcpf(__builtin_amdgcn_sqrtf(v.x) + 1e-8f);
    p.y -= 1e-4f * m.y * __builtin_amdgcn_rcpf(__builtin_amdgcn_sqrtf(v.y) + 1e-8f);
    p.z -= 1e-4f * m.z * __builtin_amdgcn_rcpf(__builtin_amdgcn_sqrtf(v.z) + 1e-8f);
    p.w -= 1e-4f * m.w * __builtin_amdgcn_rcpf(__builtin_amdgcn_sqrtf(v.w) + 1e-8f);
    return p;
}

// slabs: [2H rows][NS][D]; G, P, M, V: [2H][D]; S: bf16 [2H][D]
template <bool BARRIER>
__global__ void __launch_bounds__(768) k_fused(const float4* __restrict__ slabs, float4* __restrict__ G, float4* __restrict__ P,
                                               float4* __restrict__ M, float4* __restrict__ V, bf16x4* __restrict__ S,
                                               float* __restrict__ part_sq, unsigned* __restrict__ ticket, unsigned* __restrict__ flag,
                                               unsigned epoch) {
    __shared__ float red[12];
    __shared__ float gs_s;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int h = blockIdx.x * 12 + wave;
    float4 g[2][2];
    float sq = 0.f;
    // ---- phase A
    {
        float4 v[2][2][NS];
#pragma unroll
        for (int r = 0; r < 2; ++r)
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const int c = min(lane + 64 * i, NC - 1);
                const long row = r ? H + h : h;
#pragma unroll
                for (int s = 0; s < NS; ++s) v[r][i][s] = slabs[(row * NS + s) * NC + c];
            }
#pragma unroll
        for (int r = 0; r < 2; ++r)
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                float4 a = v[r][i][0];
#pragma unroll
                for (int s = 1; s < NS; ++s) { a.x += v[r][i][s].x; a.y += v[r][i][s].y; a.z += v[r][i][s].z; a.w += v[r][i][s].w; }
                g[r][i] = a;
                const int c = lane + 64 * i;
                if (c < NC) {
                    G[(long)(r ? H + h : h) * NC + c] = a;
                    sq += a.x * a.x + a.y * a.y + a.z * a.z + a.w * a.w;
                }
            }
    }
    float4 p[2][2], m[2][2], vv[2][2];
#pragma unroll
    for (int r = 0; r < 2; ++r)
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const long o = (long)(r ? H + h : h) * NC + min(lane + 64 * i, NC - 1);
            p[r][i] = P[o]; m[r][i] = M[o]; vv[r][i] = V[o];
        }
    // block norm partial
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) sq += __shfl_xor(sq, o, 64);
    if (lane == 0) red[wave] = sq;
    __syncthreads();
    if (threadIdx.x == 0) {
        float t = 0.f;
        for (int w = 0; w < 12; ++w) t += red[w];
        __hip_atomic_store(part_sq + blockIdx.x, t, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (BARRIER) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            const unsigned o = __hip_atomic_fetch_add(ticket, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);
            if (o == gridDim.x - 1) {
                __hip_atomic_store(ticket, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                __hip_atomic_store(flag, epoch, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
            } else {
                int spins = 0;
                while (__hip_atomic_load(flag, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) != epoch && ++spins < (1 << 22)) __builtin_amdgcn_s_sleep(2);
            }
        }
    }
    __syncthreads();
    // ---- phase B: clip coefficient from the partials (every block), update, store
    {
        float s = 0.f;
        for (int i = threadIdx.x; i < (int)gridDim.x; i += 768) s += __hip_atomic_load(part_sq + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
        __syncthreads();
        if (lane == 0) red[wave] = s;
        __syncthreads();
        if (threadIdx.x == 0) {
            float t = 0.f;
            for (int w = 0; w < 12; ++w) t += red[w];
            gs_s = fminf(1.f, 1.f / (sqrtf(t) + 1e-6f));
        }
        __syncthreads();
    }
    const float gs = gs_s;
#pragma unroll
    for (int r = 0; r < 2; ++r)
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int c = lane + 64 * i;
            if (c >= NC) continue;
            const long o = (long)(r ? H + h : h) * NC + c;
            const float4 q = upd(p[r][i], g[r][i], m[r][i], vv[r][i], gs);
            P[o] = q; M[o] = m[r][i]; V[o] = vv[r][i];
            bf16x4 s; s[0] = (__bf16)q.x; s[1] = (__bf16)q.y; s[2] = (__bf16)q.z; s[3] = (__bf16)q.w;
            S[o] = s;
        }
}

__global__ void k_fill(float4* X, long n4) {
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long)gridDim.x * 256) X[i] = make_float4(1e-3f, 2e-3f, 3e-3f, 4e-3f);
}

int main() {
    const long rows = 2L * H, n4 = rows * NC;
    float4 *slabs, *G, *P, *M, *V, *junk;
    bf16x4* S;
    float* part;
    unsigned* sync;
    const long junk4 = 400L * 1024 * 1024 / 16;
    CK(hipMalloc(&slabs, n4 * NS * 16)); CK(hipMalloc(&G, n4 * 16)); CK(hipMalloc(&P, n4 * 16)); CK(hipMalloc(&M, n4 * 16));
    CK(hipMalloc(&V, n4 * 16)); CK(hipMalloc(&S, n4 * 8)); CK(hipMalloc(&junk, junk4 * 16)); CK(hipMalloc(&part, 4096));
    CK(hipMalloc(&sync, 64));
    CK(hipMemset(sync, 0, 64));
    k_fill<<<2048, 256>>>(slabs, n4 * NS); k_fill<<<2048, 256>>>(P, n4); k_fill<<<2048, 256>>>(M, n4); k_fill<<<2048, 256>>>(V, n4);
    CK(hipDeviceSynchronize());
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    unsigned epoch = 0;
    for (int variant = 0; variant < 4; ++variant) {
        const bool barrier = variant & 1, fresh_slabs = variant & 2;
        float best = 1e9f, sum = 0.f;
        const int reps = 12;
        for (int r = 0; r < reps; ++r) {
            k_fill<<<2048, 256>>>(junk, junk4);
            if (fresh_slabs) k_fill<<<2048, 256>>>(slabs, n4 * NS);
            CK(hipEventRecord(e0));
            ++epoch;
            if (barrier) {
                // (plain launch: 256 workgroups whose registers allow one per CU on a 256-CU device are all resident;
                // hipLaunchCooperativeKernel measured 35 us slower per launch)
                k_fused<true><<<H / 12, 768>>>(slabs, G, P, M, V, S, part, sync, sync + 8, epoch);
            } else {
                k_fused<false><<<H / 12, 768>>>(slabs, G, P, M, V, S, part, sync, sync + 8, epoch);
            }
            CK(hipEventRecord(e1));
            CK(hipEventSynchronize(e1));
            float ms;
            CK(hipEventElapsedTime(&ms, e0, e1));
            if (r >= 2) { best = fminf(best, ms); sum += ms; }
        }
        printf("fused tail, %s, slabs %s: best %.1f us, mean %.1f us\n", barrier ? "grid barrier (ticket + flag)" : "no barrier (mock: wrong clip)",
               fresh_slabs ? "just written" : "cold", best * 1e3f, sum / (reps - 2) * 1e3f);
    }
    const double mb = (n4 * NS * 16 + n4 * 16 * 3 + n4 * 16 * 4 + n4 * 8) / 1e6;
    printf("bytes per launch: %.1f MB (slabs %.1f, p/m/v read %.1f, grads + p/m/v + bf16 written %.1f)\n", mb, n4 * NS * 16 / 1e6,
           n4 * 48 / 1e6, (n4 * 64 + n4 * 8) / 1e6);
    return 0;
}
