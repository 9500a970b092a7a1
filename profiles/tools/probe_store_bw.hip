#include <hip/hip_runtime.h>
#include <cstdio>
// store-bandwidth probe: 16384 x 3072 floats (201 MB) written with different per-instruction footprints
// mode 0: contiguous (each wave instruction 1 KB contiguous, a block streams a contiguous range)
// mode 1: tile pattern of the encoder GEMM: block = 256x256 tile, wave = 128 rows x 64 cols, instruction = 4 rows x 256 B
// mode 2: tile pattern with 512-byte row segments: instruction = 2 rows x 512 B (wave 64 rows x 128 cols)
// mode 3: instruction = 1 row x 1 KB (wave 32 rows x 256 cols)
__global__ void __launch_bounds__(512) wk(float* __restrict__ out, int B, int H, int mode) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const float4 v = make_float4(1.f, 2.f, 3.f, (float)tid);
    const int ntn = H / 256, ntiles = ntn * (B / 256);
    for (int t = blockIdx.x; t < ntiles; t += gridDim.x) {
        const int m0 = (t / ntn) * 256, n0 = (t % ntn) * 256;
        if (mode == 0) {
            float4* base = (float4*)out + (size_t)t * (256 * 256 / 4);
            for (int i = 0; i < 32; ++i) base[(size_t)(wave * 32 + i) * 64 + lane] = v;
        } else if (mode == 1) {
            const int wm = wave >> 2, wn = wave & 3, pr = lane >> 4, pc = (lane & 15) * 4;
            for (int i = 0; i < 32; ++i) {
                const int row = m0 + wm * 128 + 4 * i + pr;
                *(float4*)(out + (size_t)row * H + n0 + wn * 64 + pc) = v;
            }
        } else if (mode == 2) {
            const int wm = wave >> 1, wn = wave & 1, pr = lane >> 5, pc = (lane & 31) * 4;
            for (int i = 0; i < 32; ++i) {
                const int row = m0 + wm * 64 + 2 * i + pr;
                *(float4*)(out + (size_t)row * H + n0 + wn * 128 + pc) = v;
            }
        } else {
            for (int i = 0; i < 32; ++i) {
                const int row = m0 + wave * 32 + i;
                *(float4*)(out + (size_t)row * H + n0 + lane * 4) = v;
            }
        }
    }
}
int main() {
    const int B = 16384, H = 3072;
    float* out; (void)hipMalloc(&out, (size_t)B * H * 4);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    for (int mode = 0; mode < 4; ++mode)
        for (int rep = 0; rep < 3; ++rep) {
            (void)hipEventRecord(e0, 0);
            for (int it = 0; it < 10; ++it) wk<<<256, 512>>>(out, B, H, mode);
            (void)hipEventRecord(e1, 0);
            (void)hipDeviceSynchronize();
            float ms; (void)hipEventElapsedTime(&ms, e0, e1);
            if (rep == 2) printf("mode %d: %.1f us per 201 MB = %.2f TB/s\n", mode, ms * 100, (double)B * H * 4 / (ms / 10 * 1e-3) / 1e12);
        }
    return 0;
}
