// What gfx950 does with OCP fp8 (e4m3fn): v_cvt_pk_fp8_f32 (rounding, saturation) and the operand layout of
// v_mfma_f32_32x32x16_fp8_fp8.  hipcc --offload-arch=gfx950 probe_fp8.hip -o probe_fp8 && ./probe_fp8
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cmath>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));

__global__ void cvt_kernel(const float* in, uint8_t* out, int n) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) {
        int p = __builtin_amdgcn_cvt_pk_fp8_f32(in[i], 0.f, 0, false);
        out[i] = (uint8_t)(p & 0xff);
    }
}
__global__ void back_kernel(const uint8_t* in, float* out, int n) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = __builtin_amdgcn_cvt_f32_fp8((int)in[i], 0);
}
// A [32][16], B [16][32] as fp8 bytes; assumed layout: lane l supplies A[l & 31][8 (l >> 5) + j], B[8 (l >> 5) + j][l & 31], byte j
__global__ void mfma_kernel(const uint8_t* A, const uint8_t* B, float* C) {
    const int l = threadIdx.x;
    uint64_t a = 0, b = 0;
    for (int j = 0; j < 8; ++j) {
        a |= (uint64_t)A[(l & 31) * 16 + 8 * (l >> 5) + j] << (8 * j);
        b |= (uint64_t)B[(8 * (l >> 5) + j) * 32 + (l & 31)] << (8 * j);
    }
    f32x16 acc;
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    acc = __builtin_amdgcn_mfma_f32_32x32x16_fp8_fp8((long)a, (long)b, acc, 0, 0, 0);
    for (int r = 0; r < 16; ++r) C[((r & 3) + 8 * (r >> 2) + 4 * (l >> 5)) * 32 + (l & 31)] = acc[r];
}

int main() {
    const float tests[] = {0.f, 1.f, -1.f, 0.0625f, 448.f, 449.f, 464.f, 480.f, 1000.f, 1e30f, 17.f, 18.f, 19.f, 0.001953125f, 0.0009765625f, 0.0005f, 3.3f, -447.9f, 208.1f, 240.f, 1.0625f, 1.1875f};
    const int n = sizeof(tests) / sizeof(float);
    float* din; uint8_t* dq; float* dback;
    hipMalloc(&din, n * 4); hipMalloc(&dq, n); hipMalloc(&dback, n * 4);
    hipMemcpy(din, tests, n * 4, hipMemcpyHostToDevice);
    cvt_kernel<<<1, 64>>>(din, dq, n);
    back_kernel<<<1, 64>>>(dq, dback, n);
    std::vector<uint8_t> q(n); std::vector<float> back(n);
    hipMemcpy(q.data(), dq, n, hipMemcpyDeviceToHost); hipMemcpy(back.data(), dback, n * 4, hipMemcpyDeviceToHost);
    for (int i = 0; i < n; ++i) printf("cvt %12g -> 0x%02x -> %g\n", tests[i], q[i], back[i]);
    // all 256 codes
    std::vector<uint8_t> codes(256); for (int i = 0; i < 256; ++i) codes[i] = (uint8_t)i;
    uint8_t* dc; float* dv; hipMalloc(&dc, 256); hipMalloc(&dv, 1024);
    hipMemcpy(dc, codes.data(), 256, hipMemcpyHostToDevice);
    back_kernel<<<1, 256>>>(dc, dv, 256);
    std::vector<float> vals(256); hipMemcpy(vals.data(), dv, 1024, hipMemcpyDeviceToHost);
    printf("codes 0x7e %g 0x7f %g 0x08 %g 0x01 %g 0xff %g 0x78 %g\n", vals[0x7e], vals[0x7f], vals[0x08], vals[0x01], vals[0xff], vals[0x78]);
    // mfma layout
    std::vector<uint8_t> A(32 * 16), B(16 * 32);
    for (int i = 0; i < 32 * 16; ++i) A[i] = (uint8_t)(0x20 + (i * 7) % 48);   // positive normal codes
    for (int i = 0; i < 16 * 32; ++i) B[i] = (uint8_t)(0x28 + (i * 5) % 40) | ((i % 3 == 0) ? 0x80 : 0);
    uint8_t *dA, *dB; float* dC; hipMalloc(&dA, 512); hipMalloc(&dB, 512); hipMalloc(&dC, 4096);
    hipMemcpy(dA, A.data(), 512, hipMemcpyHostToDevice); hipMemcpy(dB, B.data(), 512, hipMemcpyHostToDevice);
    mfma_kernel<<<1, 64>>>(dA, dB, dC);
    std::vector<float> C(1024); hipMemcpy(C.data(), dC, 4096, hipMemcpyDeviceToHost);
    double worst = 0;
    for (int m = 0; m < 32; ++m) for (int nn = 0; nn < 32; ++nn) {
        double s = 0;
        for (int k = 0; k < 16; ++k) s += (double)vals[A[m * 16 + k]] * (double)vals[B[k * 32 + nn]];
        worst = fmax(worst, fabs(s - C[m * 32 + nn]) / (fabs(s) + 1e-6));
    }
    printf("mfma_f32_32x32x16_fp8_fp8 with the assumed layout: worst relative error %g\n", worst);
    return 0;
}
