// Prints what the gfx950 cross-lane VALU forms used by decode_fast_kernel deliver, lane by lane.
//   hipcc --offload-arch=gfx950 -O2 profiles/tools/lane_ops_probe.hip -o /tmp/lane_probe && /tmp/lane_probe
#include <hip/hip_runtime.h>
#include <cstdio>
template <int CTRL>
__device__ int dpp(int x) { return __builtin_amdgcn_update_dpp(-1, x, CTRL, 0xF, 0xF, false); }
__global__ void k(int* out) {
    const int l = threadIdx.x;
    const int a = l, b = 100 + l;
    auto s16 = __builtin_amdgcn_permlane16_swap(a, b, false, false);
    auto s32 = __builtin_amdgcn_permlane32_swap(a, b, false, false);
    out[l] = s16[0]; out[64 + l] = s16[1]; out[128 + l] = s32[0]; out[192 + l] = s32[1];
    out[256 + l] = dpp<0x128>(a); out[320 + l] = dpp<0x104>(a); out[384 + l] = dpp<0x114>(a);
    out[448 + l] = dpp<0xB1>(a); out[512 + l] = dpp<0x4E>(a);
}
int main() {
    int* d; hipMalloc(&d, 576 * 4);
    k<<<1, 64>>>(d);
    int h[576]; hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    const char* names[9] = {"swap16.dst(a=l,b=100+l)", "swap16.src", "swap32.dst", "swap32.src", "row_ror:8", "row_shl:4", "row_shr:4", "quad[1,0,3,2]", "quad[2,3,0,1]"};
    for (int r = 0; r < 9; ++r) { printf("%-24s", names[r]); for (int l = 0; l < 64; ++l) printf(" %d", h[r * 64 + l]); printf("\n"); }
    return 0;
}
