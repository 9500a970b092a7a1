// Operand layout of v_mfma_scale_f32_32x32x64_f8f6f4 with e4m3 operands and unit block scales (E8M0 0x7F), checked
// against a host product on exact small integers.  Hypotheses for lane l (r = l & 31, h = l >> 5), byte j = 0..31 of the
// 32-byte operand:   H1: k = 32 h + j        H2: k = 16 h + (j & 15) + 32 (j >> 4)
//   hipcc --offload-arch=gfx950 -O3 -o probe_mxfp8 probe_mxfp8.hip && ./probe_mxfp8
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
typedef int v8i __attribute__((ext_vector_type(8)));
typedef float v16f __attribute__((ext_vector_type(16)));

__global__ void k(const v8i* a, const v8i* b, v16f* c, int scale) {
    v16f acc = {};
    acc = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a[threadIdx.x], b[threadIdx.x], acc, 0, 0, 0, scale, 0, scale);
    c[threadIdx.x] = acc;
}

static uint8_t e4m3(int v) {  // exact for |v| <= 8: sign, exponent (bias 7), 3 mantissa bits
    if (v == 0) return 0;
    const uint8_t s = v < 0 ? 0x80 : 0;
    int m = abs(v), e = 0;
    while ((1 << (e + 1)) <= m) ++e;
    const int frac = ((m << 3) >> e) & 7;
    return s | (uint8_t)((e + 7) << 3) | (uint8_t)frac;
}

int main() {
    int A[32][64], B[64][32];
    srand(5);
    for (int i = 0; i < 32; ++i) for (int kk = 0; kk < 64; ++kk) A[i][kk] = rand() % 9 - 4;
    for (int kk = 0; kk < 64; ++kk) for (int j = 0; j < 32; ++j) B[kk][j] = rand() % 9 - 4;
    double ref[32][32];
    for (int i = 0; i < 32; ++i) for (int j = 0; j < 32; ++j) { double s = 0; for (int kk = 0; kk < 64; ++kk) s += A[i][kk] * B[kk][j]; ref[i][j] = s; }
    for (int hyp = 1; hyp <= 2; ++hyp) {
        uint8_t ha[64][32], hb[64][32];
        for (int l = 0; l < 64; ++l)
            for (int j = 0; j < 32; ++j) {
                const int r = l & 31, h = l >> 5;
                const int kk = hyp == 1 ? 32 * h + j : 16 * h + (j & 15) + 32 * (j >> 4);
                ha[l][j] = e4m3(A[r][kk]);
                hb[l][j] = e4m3(B[kk][r]);
            }
        void *da, *db, *dc;
        hipMalloc(&da, sizeof(ha)); hipMalloc(&db, sizeof(hb)); hipMalloc(&dc, 64 * 64);
        hipMemcpy(da, ha, sizeof(ha), hipMemcpyHostToDevice);
        hipMemcpy(db, hb, sizeof(hb), hipMemcpyHostToDevice);
        for (int sc : {(int)0x7F7F7F7F, (int)0x80808080u}) {
            k<<<1, 64>>>((const v8i*)da, (const v8i*)db, (v16f*)dc, sc);
            float out[64][16];
            hipMemcpy(out, dc, sizeof(out), hipMemcpyDeviceToHost);
            int bad = 0;
            double ratio = 0;
            for (int l = 0; l < 64; ++l)
                for (int rg = 0; rg < 16; ++rg) {
                    const int row = (rg & 3) + 8 * (rg >> 2) + 4 * (l >> 5), col = l & 31;
                    const double want = ref[row][col] * (sc == 0x7F7F7F7F ? 1.0 : 4.0);  // both scales 2^1: product x 4
                    if (out[l][rg] != (float)want) ++bad;
                    if (ref[row][col] != 0) ratio = out[l][rg] / ref[row][col];
                }
            printf("hypothesis H%d, scale bytes 0x%02X: %d of 1024 outputs differ (last ratio %.3f)\n", hyp, sc & 0xFF, bad, ratio);
        }
    }
    return 0;
}
