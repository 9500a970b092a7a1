// Probe for the MFMA decode kernel's building blocks on one wave: LDS-DMA gather with the source swizzle, the
// transposed fragment read, and both 16x16x32 contractions, each against a host-side expectation.
//   hipcc --offload-arch=gfx950 -O2 -I whisper-sae_amd/csrc -I include profiles/tools/probe_decode_frag.hip -o /tmp/probe && /tmp/probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
#include "wsae_common.h"
#include "wsae_mfma.h"

typedef __attribute__((address_space(3))) bf16x4 lds_bf16x4;
__device__ __forceinline__ int dm_sw(int j) { return (j & 15) ^ ((j & 1) ? 12 : 0); }

__global__ void probe(const bf16_t* W /*[32][128]*/, const float* v /*[32]*/, const bf16_t* g /*[128]*/, float* recon /*[128]*/,
                      float* dots /*[32]*/, float* raw /*[64][8] tr frag of tile 0 as floats*/) {
    __shared__ __attribute__((aligned(256))) char slot[32 * 256];
    __shared__ __attribute__((aligned(16))) bf16_t grow[128];
    const int lane = threadIdx.x, n = lane & 15, grp = lane >> 4;
    const uint32_t slot_lds = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) char*)slot;
    for (int e = 0; e < 8; ++e) {
        const int j = 4 * e + grp;
        glds16(W + j * 128 + 8 * (n ^ dm_sw(j)), slot_lds + e * 1024);
    }
    for (int c = lane; c < 128; c += 64) grow[c] = g[c];
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    // A fragment of pass 1
    bf16x8 a1;
    for (int e = 0; e < 8; ++e) {
        const float x = v[8 * grp + e];
        const bf16_t hi = (bf16_t)x;
        const float r1 = x - (float)hi;
        const bf16_t lo = (bf16_t)r1;
        const bf16_t lo2 = (bf16_t)(r1 - (float)lo);
        a1[e] = n == 0 ? hi : n == 1 ? lo : n == 2 ? lo2 : (bf16_t)0.f;
    }
    for (int t = 0; t < 8; ++t) {
        const int i = n >> 2, p = n & 3;
        const int j0 = 8 * grp + i, j1 = j0 + 4;
        const int c = 2 * t + (p >> 1);
        const bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4*)(slot + j0 * 256 + ((c ^ dm_sw(j0)) << 4) + 8 * (p & 1)));
        const bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4*)(slot + j1 * 256 + ((c ^ dm_sw(j1)) << 4) + 8 * (p & 1)));
        const bf16x8 bf = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
        if (t == 0) for (int q = 0; q < 8; ++q) raw[lane * 8 + q] = (float)bf[q];
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
        acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a1, bf, acc, 0, 0, 0);
        if (lane < 16) recon[16 * t + n] = acc[0] + acc[1] + acc[2];
    }
    f32x4 acc2[2] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
    for (int kk = 0; kk < 4; ++kk) {
        const bf16x8 gf = *(const bf16x8*)(grow + 32 * kk + 8 * grp);
        for (int m = 0; m < 2; ++m) {
            const int j = 16 * m + n;
            const bf16x8 af = *(const bf16x8*)(slot + j * 256 + (((4 * kk + grp) ^ dm_sw(j)) << 4));
            acc2[m] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af, gf, acc2[m], 0, 0, 0);
        }
    }
    for (int m = 0; m < 2; ++m) {
        const int j = 16 * m + 4 * grp + (n & 3);
        const float d0 = (n & 2) ? ((n & 1) ? acc2[m][3] : acc2[m][2]) : ((n & 1) ? acc2[m][1] : acc2[m][0]);
        if (n < 4) dots[j] = d0;
    }
}

int main() {
    std::vector<bf16_t> W(32 * 128), g(128);
    std::vector<float> v(32);
    for (int j = 0; j < 32; ++j) for (int c = 0; c < 128; ++c) W[j * 128 + c] = (bf16_t)(float)(((j * 7 + c * 3) % 61) - 30);
    for (int j = 0; j < 32; ++j) v[j] = 0.37f * (j % 5) + 0.013f * j;
    for (int c = 0; c < 128; ++c) g[c] = (bf16_t)(float)((c % 9) - 4);
    bf16_t *dW, *dg; float *dv, *dr, *dd, *draw;
    hipMalloc(&dW, W.size() * 2); hipMalloc(&dg, 256); hipMalloc(&dv, 128); hipMalloc(&dr, 512); hipMalloc(&dd, 128); hipMalloc(&draw, 2048);
    hipMemcpy(dW, W.data(), W.size() * 2, hipMemcpyHostToDevice); hipMemcpy(dg, g.data(), 256, hipMemcpyHostToDevice);
    hipMemcpy(dv, v.data(), 128, hipMemcpyHostToDevice);
    probe<<<1, 64>>>(dW, dv, dg, dr, dd, draw);
    std::vector<float> r(128), d(32), raw(512);
    hipMemcpy(r.data(), dr, 512, hipMemcpyDeviceToHost); hipMemcpy(d.data(), dd, 128, hipMemcpyDeviceToHost);
    hipMemcpy(raw.data(), draw, 2048, hipMemcpyDeviceToHost);
    int bad_raw = 0;
    for (int l = 0; l < 64; ++l) for (int q = 0; q < 8; ++q) {
        const float want = (float)W[(8 * (l >> 4) + q) * 128 + (l & 15)];
        if (raw[l * 8 + q] != want) { if (bad_raw < 8) printf("raw lane %d elem %d got %g want %g\n", l, q, raw[l * 8 + q], want); ++bad_raw; }
    }
    printf("tr fragment mismatches: %d of 512\n", bad_raw);
    double e1 = 0, e2 = 0;
    for (int c = 0; c < 128; ++c) { double s = 0; for (int j = 0; j < 32; ++j) s += (double)v[j] * (float)W[j * 128 + c]; e1 = fmax(e1, fabs(s - r[c])); }
    for (int j = 0; j < 32; ++j) { double s = 0; for (int c = 0; c < 128; ++c) s += (double)(float)g[c] * (float)W[j * 128 + c]; e2 = fmax(e2, fabs(s - d[j])); }
    printf("pass 1 max err %g   pass 2 max err %g\n", e1, e2);
    return 0;
}
