// Encode path: batch staging, pre-activation GEMM on MFMA, per-row TopK with wavefront reductions.
//   reference: TopKSAE.encode, src/whisper_sae/sae/model.py:98-118
#include <stdlib.h>

#include "wsae_common.h"
#include "wsae_mfma.h"

__device__ __forceinline__ uint32_t f32_ord(float f) {
    const uint32_t u = __float_as_uint(f);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ __forceinline__ float ord_f32(uint32_t o) {
    return __uint_as_float((o & 0x80000000u) ? (o & 0x7FFFFFFFu) : ~o);
}
// value of lane (l ^ M) without the LDS pipe (__shfl_xor = ds_bpermute_b32, ~100+ cycles of latency each, and a
// bitonic sort of 64 keys chains 21 of them per 32-bit half): DPP for M = 1, 2, 4, 8, v_permlane16/32_swap for
// M = 16, 32 (profiles/tools/lane_ops_probe.hip prints what each control delivers).  Every DPP move runs with
// all lanes active and the select comes after it: a DPP source lane that is masked off reads as invalid.
template <int M>
__device__ __forceinline__ uint32_t lane_xor_u32(uint32_t x, int lane) {
    if constexpr (M == 1) return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0xB1, 0xF, 0xF, false);        // quad_perm [1,0,3,2]
    else if constexpr (M == 2) return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x4E, 0xF, 0xF, false);   // quad_perm [2,3,0,1]
    else if constexpr (M == 4) {
        const uint32_t up = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x104, 0xF, 0xF, false);  // row_shl:4 = lane l + 4
        const uint32_t dn = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x114, 0xF, 0xF, false);  // row_shr:4 = lane l - 4
        return (lane & 4) ? dn : up;
    } else if constexpr (M == 8) return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x128, 0xF, 0xF, false);  // row_ror:8
    else if constexpr (M == 16) {
        // swap16(a = x, b = x): a' = rows [x0, x0, x2, x2], b' = rows [x1, x1, x3, x3]
        const auto r = __builtin_amdgcn_permlane16_swap((int)x, (int)x, false, false);
        return (uint32_t)((lane & 16) ? r[0] : r[1]);
    } else {
        static_assert(M == 32, "lane_xor_u32: M must be a power of two <= 32");
        // swap32(a = x, b = x): a' = [x.lo, x.lo], b' = [x.hi, x.hi]
        const auto r = __builtin_amdgcn_permlane32_swap((int)x, (int)x, false, false);
        return (uint32_t)((lane & 32) ? r[0] : r[1]);
    }
}

template <int M>
__device__ __forceinline__ uint64_t lane_xor_u64(uint64_t v, int lane) {
    return ((uint64_t)lane_xor_u32<M>((uint32_t)(v >> 32), lane) << 32) | lane_xor_u32<M>((uint32_t)v, lane);
}


// ------------------------------------------------------------------------------------------------
// stage_batch: gather + convert the batch once.
//   xb [B][D]      compute dtype: bf16(x) (BF16 mode) or x - b_pre in f32 (FP32 mode, model.py:108)
//   xT [D][ldT]    its transpose (Bt operand of the dW_e contraction), zero beyond column B
// 64 x 64 tiles through LDS so both the read (along d) and the write (along b) are coalesced.
// ------------------------------------------------------------------------------------------------
template <typename T>
__device__ __forceinline__ void store4(T* p, const float (&v)[4]) {
    if (sizeof(T) == 2) {
        bf16x4 o;
#pragma unroll
        for (int i = 0; i < 4; ++i) o[i] = (bf16_t)v[i];
        *(bf16x4*)p = o;
    } else {
        *(float4*)p = make_float4(v[0], v[1], v[2], v[3]);
    }
}

template <int XDT>
__device__ __forceinline__ void load4(const void* x, int64_t i, float (&v)[4]) {
    if (XDT == WSAE_DT_BF16) {
        const bf16x4 t = *(const bf16x4*)((const bf16_t*)x + i);
#pragma unroll
        for (int q = 0; q < 4; ++q) v[q] = (float)t[q];
    } else {
        const float4 t = *(const float4*)((const float*)x + i);
        v[0] = t.x; v[1] = t.y; v[2] = t.z; v[3] = t.w;
    }
}

template <int XDT, typename T>
__global__ void __launch_bounds__(256) stage_batch_kernel(const void* __restrict__ x, const int32_t* __restrict__ rows,
                                                          const float* __restrict__ bpre, T* __restrict__ xb,
                                                          T* __restrict__ xT, int B, int D, int ldT) {
    __shared__ float tile[64][65];
    const int b0 = blockIdx.x * 64, d0 = blockIdx.y * 64;
    const int q = threadIdx.x & 15, r16 = threadIdx.x >> 4;  // 16 threads x 4 elements span the tile width
#pragma unroll
    for (int p = 0; p < 4; ++p) {
        const int bl = r16 + 16 * p, b = b0 + bl, d = d0 + 4 * q;
        float v[4] = {0.f, 0.f, 0.f, 0.f};
        if (b < B && d < D) {
            const int64_t src = rows ? (int64_t)rows[b] : (int64_t)b;
            load4<XDT>(x, src * D + d, v);
            if (sizeof(T) == 4) {
                const float4 bp = *(const float4*)(bpre + d);
                v[0] -= bp.x; v[1] -= bp.y; v[2] -= bp.z; v[3] -= bp.w;
            }
            store4<T>(xb + (int64_t)b * D + d, v);
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) tile[bl][4 * q + i] = v[i];
    }
    __syncthreads();
#pragma unroll
    for (int p = 0; p < 4; ++p) {
        const int dl = r16 + 16 * p, d = d0 + dl, b = b0 + 4 * q;
        if (d < D && b < ldT) {
            float v[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) v[i] = (T)tile[4 * q + i][dl];
            store4<T>(xT + (int64_t)d * ldT + b, v);
        }
    }
}

// ------------------------------------------------------------------------------------------------
// pre[b][h] = sum_d xb[b][d] * W[h][d] + bias[h]          (model.py:111)
//
// One kernel, three uses (template MODE):
//   GEMM_DENSE  : write the [B][ldp] pre-activation matrix (API path, small shapes, sample pass, fallback)
//   GEMM_FILTER : do NOT write pre; keep only elements >= thr[b] as 64-bit keys in a private
//                 (row, feature-tile) slot group -- the fused-TopK path.  thr[b] comes from a sample
//                 pass (every wstride-th feature), so the [B,H] matrix never goes to HBM.
// wstride > 1 samples features h = n * wstride (W row stride and bias index scale with it).
// arows / n_dev (nullable): batch-row indirection and device-side row count, used by the fallback
// launch over the rows the filter could not settle; blocks beyond the count exit immediately.
// ------------------------------------------------------------------------------------------------
#define GEMM_DENSE 0
#define GEMM_FILTER 1
#define CAND_SLOTS 24  // candidate slots per (row, 128-feature tile)

template <typename T, int MODE>
__global__ void __launch_bounds__(256)
encode_gemm_kernel(const T* __restrict__ xb, const T* __restrict__ W, const float* __restrict__ bias,
                   float* __restrict__ pre, int ldp, int B, int H, int D, int wstride, const int32_t* __restrict__ arows,
                   const int32_t* __restrict__ n_dev, const float* __restrict__ thr, int thr_stride,
                   uint64_t* __restrict__ cand, int32_t* __restrict__ cand_cnt, int32_t* __restrict__ ovf) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* As = smem;
    char* Bs = smem + TILE_LDS_BYTES;
    int* cnt_s = (int*)(smem + 2 * TILE_LDS_BYTES);  // [128] (FILTER)
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int m0 = blockIdx.y * TILE_M, n0 = blockIdx.x * TILE_N;
    if (n_dev) B = min(B, *n_dev);
    if (m0 >= B) return;
    constexpr int KT = Mfma<T>::KT;
    if (MODE == GEMM_FILTER && tid < 128) cnt_s[tid] = 0;

    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    // A rows may be indirect (fallback): resolve this thread's four slab rows once
    int arow[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int row = m0 + ((tid + 256 * i) >> 3);
        arow[i] = row < B ? (arows ? arows[row] : row) : -1;
    }
    constexpr int EPC = 16 / (int)sizeof(T);
    auto chunk_a = [&](int i, int k0) {
        const int k = k0 + (tid & 7) * EPC;
        return (arow[i] >= 0 && k < D) ? *(const uint4*)(xb + (int64_t)arow[i] * D + k) : make_uint4(0, 0, 0, 0);
    };
    auto load_a = [&](SlabRegs<T>& r, int k0) {
        r.v0 = chunk_a(0, k0);
        r.v1 = chunk_a(1, k0);
        r.v2 = chunk_a(2, k0);
        r.v3 = chunk_a(3, k0);
    };
    const int64_t ldw = (int64_t)D * wstride;
    SlabRegs<T> ra, rb;
    load_a(ra, 0);
    slab_load<T>(rb, W, ldw, n0, H, 0, D, tid);
    const int nk = (D + KT - 1) / KT;
    for (int kt = 0; kt < nk; ++kt) {
        slab_store<T>(ra, As, tid);
        slab_store<T>(rb, Bs, tid);
        __syncthreads();
        if (kt + 1 < nk) {
            load_a(ra, (kt + 1) * KT);
            slab_load<T>(rb, W, ldw, n0, H, (kt + 1) * KT, D, tid);
        }
        Mfma<T>::slab(As, Bs, wm * 64, wn * 64, lane, acc);
        __syncthreads();
    }
    const int col = lane & 31, rq = lane >> 5;
    if (MODE == GEMM_DENSE) {
#pragma unroll
        for (int mi = 0; mi < 2; ++mi)
#pragma unroll
            for (int ni = 0; ni < 2; ++ni) {
                const int h = n0 + wn * 64 + ni * 32 + col;
                if (h >= H) continue;
                const float bv = bias[(int64_t)h * wstride];
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int b = m0 + wm * 64 + mi * 32 + (r & 3) + 8 * (r >> 2) + 4 * rq;
                    if (b < B) pre[(int64_t)b * ldp + h] = acc[mi][ni][r] + bv;
                }
            }
    } else {
        const int ntile = gridDim.x;
#pragma unroll
        for (int mi = 0; mi < 2; ++mi) {
            float tv[16];
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int b = m0 + wm * 64 + mi * 32 + (r & 3) + 8 * (r >> 2) + 4 * rq;
                tv[r] = b < B ? thr[(int64_t)b * thr_stride] : INFINITY;
            }
#pragma unroll
            for (int ni = 0; ni < 2; ++ni) {
                const int h = n0 + wn * 64 + ni * 32 + col;
                const bool hin = h < H;
                const float bv = hin ? bias[h] : 0.f;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const float v = acc[mi][ni][r] + bv;
                    const bool pass = hin && v >= tv[r];
                    if (__ballot(pass)) {  // wave-uniform: most registers hold no candidate at all
                        if (pass) {
                            const int rl = wm * 64 + mi * 32 + (r & 3) + 8 * (r >> 2) + 4 * rq;
                            const int slot = atomicAdd(&cnt_s[rl], 1);
                            if (slot < CAND_SLOTS)
                                cand[((int64_t)(m0 + rl) * ntile + blockIdx.x) * CAND_SLOTS + slot] =
                                    ((uint64_t)f32_ord(v) << 32) | (uint32_t)(~(uint32_t)h);
                        }
                    }
                }
            }
        }
        __syncthreads();
        if (tid < 128 && m0 + tid < B) {
            const int c = cnt_s[tid];
            cand_cnt[(int64_t)(m0 + tid) * ntile + blockIdx.x] = min(c, CAND_SLOTS);
            if (c > CAND_SLOTS) ovf[m0 + tid] = 1;  // slot group overflowed: the row goes to the exact fallback
        }
    }
}

// ------------------------------------------------------------------------------------------------
// encode_gemm256_kernel: 256 x 256 output tile, 8 waves (4 x 2, each 128 x 64), LDS double buffer,
// one barrier per K slab, next slab's operands in flight during the MFMAs.  DENSE only: the
// pre-activation matrix for the standalone TopK kernel.  Requires D % KT == 0.
// ------------------------------------------------------------------------------------------------
#define T256_LDS (256 * LDS_ROW_BYTES)

template <typename T, int MODE>
__global__ void __launch_bounds__(512)
encode_gemm256_kernel(const T* __restrict__ xb, const T* __restrict__ W, const float* __restrict__ bias,
                      float* __restrict__ pre, int ldp, int B, int H, int D, const float* __restrict__ thr,
                      int thr_stride, uint64_t* __restrict__ cand, int32_t* __restrict__ cand_cnt,
                      int32_t* __restrict__ ovf, int cap) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    int* cnt_s = (int*)(smem + 4 * T256_LDS);   // [256] candidates per row   (FILTER)
    float* thr_s = (float*)(cnt_s + 256);        // [256] row thresholds
    constexpr int KT = Mfma<T>::KT;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 2, wn = wave & 3;  // 2 x 4 waves: 128-row halves x 64-column quarters
    const int m0 = blockIdx.y * 256, n0 = blockIdx.x * 256;
    const int t2 = tid & 255, half = tid >> 8;  // each 256-thread half stages 128 rows of A and of W
    if (MODE == GEMM_FILTER && tid < 256) {
        cnt_s[tid] = 0;
        thr_s[tid] = (m0 + tid < B) ? thr[(int64_t)(m0 + tid) * thr_stride] : INFINITY;
    }

    f32x16 acc[4][2];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    SlabRegs<T> ra, rb;
    slab_load_fast<T>(ra, xb, D, m0 + 128 * half, B - 1, 0, t2);
    slab_load_fast<T>(rb, W, D, n0 + 128 * half, H - 1, 0, t2);
    const int nk = D / KT;
    for (int kt = 0; kt < nk; ++kt) {
        char* As = smem + (kt & 1) * 2 * T256_LDS;
        char* Bs = As + T256_LDS;
        slab_store<T>(ra, As + 128 * half * LDS_ROW_BYTES, t2);
        slab_store<T>(rb, Bs + 128 * half * LDS_ROW_BYTES, t2);
        __syncthreads();
        if (kt + 1 < nk) {
            slab_load_fast<T>(ra, xb, D, m0 + 128 * half, B - 1, (kt + 1) * KT, t2);
            slab_load_fast<T>(rb, W, D, n0 + 128 * half, H - 1, (kt + 1) * KT, t2);
        }
        Mfma256<T>::slab(As, Bs, wm * 128, wn * 64, lane, acc);
    }
    const int col = lane & 31, rq = lane >> 5;
    if (MODE == GEMM_FILTER) {
        // keep only elements >= the row threshold: 64-bit keys into the (row, column-tile) list
        // (write-through stores: the candidate stream must not evict W_e from L2)
#pragma unroll
        for (int mi = 0; mi < 4; ++mi) {
            float tv[16];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const float4 t4 = *(const float4*)(thr_s + wm * 128 + mi * 32 + 8 * q + 4 * rq);
                tv[4 * q] = t4.x; tv[4 * q + 1] = t4.y; tv[4 * q + 2] = t4.z; tv[4 * q + 3] = t4.w;
            }
#pragma unroll
            for (int ni = 0; ni < 2; ++ni) {
                const int h = n0 + wn * 64 + ni * 32 + col;
                const bool hin = h < H;
                const float bv = hin ? bias[h] : 0.f;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const float v = acc[mi][ni][r] + bv;
                    const bool pass = hin && v >= tv[r];
                    if (__ballot(pass)) {
                        if (pass) {
                            const int rl = wm * 128 + mi * 32 + (r & 3) + 8 * (r >> 2) + 4 * rq;
                            const int slot = atomicAdd(&cnt_s[rl], 1);
                            if (slot < cap)
                                __hip_atomic_store(cand + ((int64_t)(m0 + rl) * gridDim.x + blockIdx.x) * cap + slot,
                                                   ((uint64_t)f32_ord(v) << 32) | (uint32_t)(~(uint32_t)h),
                                                   __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        }
                    }
                }
            }
        }
        __syncthreads();
        if (tid < 256 && m0 + tid < B) {
            const int c = cnt_s[tid];
            cand_cnt[(int64_t)(m0 + tid) * gridDim.x + blockIdx.x] = min(c, cap);
            if (c > cap) ovf[m0 + tid] = 1;
        }
        return;
    }
    // ---- dense epilogue: 16-byte row stores.  Each wave transposes its accumulators through a private
    // 32 x 64 LDS patch (the operand buffers are free now), so a store instruction writes four 256-byte
    // row pieces instead of two 128-byte ones and there are 32 of them per lane instead of 128: the tail
    // of this kernel is store-issue bound, not bandwidth bound.
    __syncthreads();  // every wave is done reading the operand buffers
    constexpr int PS = 68;  // patch row stride in floats (64 + 4: the two half-waves land on disjoint banks)
    float* patch = (float*)smem + wave * 32 * PS;
    const int pr = lane >> 4, pc = (lane & 15) * 4;  // read-back: 16 lanes per row, float4 each
    const int hcol = n0 + wn * 64 + pc;
    float4 bv4 = make_float4(0.f, 0.f, 0.f, 0.f);
    if (hcol < H) bv4 = *(const float4*)(bias + hcol);
#pragma unroll
    for (int mi = 0; mi < 4; ++mi) {
#pragma unroll
        for (int ni = 0; ni < 2; ++ni)
#pragma unroll
            for (int r = 0; r < 16; ++r)
                patch[((r & 3) + 8 * (r >> 2) + 4 * rq) * PS + ni * 32 + col] = acc[mi][ni][r];
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int rl = pr + 4 * i;
            const int b = m0 + wm * 128 + mi * 32 + rl;
            float4 v = *(const float4*)(patch + rl * PS + pc);
            v.x += bv4.x; v.y += bv4.y; v.z += bv4.z; v.w += bv4.w;
            if (b < B && hcol < H) *(float4*)(pre + (int64_t)b * ldp + hcol) = v;
        }
        __builtin_amdgcn_wave_barrier();
    }
}

// ------------------------------------------------------------------------------------------------
// encode_gemm256p_kernel: the DENSE 256 x 256 GEMM as a persistent kernel - one workgroup per CU walks
// tiles blockIdx.x, + gridDim.x, ...  What the walk buys over one workgroup per tile (768 tiles = three
// rounds on 256 CUs, each paying its own start-up and drain): the first operand slabs of the NEXT tile are
// requested before the epilogue of the current one, and the epilogue's 256 KB of stores drain
// underneath the next tile's MFMAs instead of holding the CU until the workgroup retires.
// ------------------------------------------------------------------------------------------------
template <typename T>
__global__ void __launch_bounds__(512)
encode_gemm256p_kernel(const T* __restrict__ xb, const T* __restrict__ W, const float* __restrict__ bias,
                       float* __restrict__ pre, int ldp, int B, int H, int D, int ntn, int ntiles,
                       float* __restrict__ smax) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int KT = Mfma<T>::KT;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 2, wn = wave & 3;  // 2 x 4 waves: 128-row halves x 64-column quarters
    const int t2 = tid & 255, half = tid >> 8;  // each 256-thread half stages 128 rows of A and of W
    const int nk = D / KT;
    const int col = lane & 31, rq = lane >> 5;
    constexpr int PS = 68;  // epilogue patch row stride in floats (see encode_gemm256_kernel)
    float* patch = (float*)smem + wave * 32 * PS;
    const int pr = lane >> 4, pc = (lane & 15) * 4;

    // XCD-aware walk: workgroup w runs on XCD w % 8.  When the batch tiles divide evenly over the XCDs
    // every XCD gets whole batch tiles (all ntn feature tiles of each), so its L2 reads a batch tile once
    // instead of once per feature tile on every XCD (111 MB -> the 15 MB of operands; profiles/README.md).
    // Walk index i (this workgroup's i-th tile) -> global tile id:
    const int ntm = ntiles / ntn;
    const bool xcd_walk = (gridDim.x % 8 == 0) && (ntm % 8 == 0);
    const int xcd = blockIdx.x & 7, local = blockIdx.x >> 3, per_xcd = gridDim.x >> 3;
    auto tile_at = [&](int i) -> int {
        if (!xcd_walk) return (int)blockIdx.x + i * (int)gridDim.x;
        const int tl = local + i * per_xcd;                 // index among this XCD's (ntm / 8) * ntn tiles
        if (tl >= (ntm >> 3) * ntn) return ntiles;          // past the end
        return (xcd * (ntm >> 3) + tl / ntn) * ntn + tl % ntn;
    };
    SlabRegs<T> ra, rb;
    int it = 0;
    int tile = tile_at(0);
    if (tile < ntiles) {
        slab_load_fast<T>(ra, xb, D, (tile / ntn) * 256 + 128 * half, B - 1, 0, t2);
        slab_load_fast<T>(rb, W, D, (tile % ntn) * 256 + 128 * half, H - 1, 0, t2);
    }
    for (; tile < ntiles; tile = tile_at(++it)) {
        const int m0 = (tile / ntn) * 256, n0 = (tile % ntn) * 256;
        f32x16 acc[4][2];
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
        for (int kt = 0; kt < nk; ++kt) {
            char* As = smem + (kt & 1) * 2 * T256_LDS;
            char* Bs = As + T256_LDS;
            slab_store<T>(ra, As + 128 * half * LDS_ROW_BYTES, t2);
            slab_store<T>(rb, Bs + 128 * half * LDS_ROW_BYTES, t2);
            __syncthreads();
            if (kt + 1 < nk) {
                slab_load_fast<T>(ra, xb, D, m0 + 128 * half, B - 1, (kt + 1) * KT, t2);
                slab_load_fast<T>(rb, W, D, n0 + 128 * half, H - 1, (kt + 1) * KT, t2);
            }
            Mfma256<T>::slab(As, Bs, wm * 128, wn * 64, lane, acc);
        }
        // next tile's first slabs fly during the epilogue (past the last tile: clamped re-read, unused)
        {
            const int nt = min(tile_at(it + 1), ntiles - 1);
            slab_load_fast<T>(ra, xb, D, (nt / ntn) * 256 + 128 * half, B - 1, 0, t2);
            slab_load_fast<T>(rb, W, D, (nt % ntn) * 256 + 128 * half, H - 1, 0, t2);
        }
        __syncthreads();  // every wave is done reading the operand buffers: they become the transpose patches
        const int hcol = n0 + wn * 64 + pc;
        float4 bv4 = make_float4(0.f, 0.f, 0.f, 0.f);
        if (hcol < H) bv4 = *(const float4*)(bias + hcol);
#pragma unroll
        for (int mi = 0; mi < 4; ++mi) {
#pragma unroll
            for (int ni = 0; ni < 2; ++ni)
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    patch[((r & 3) + 8 * (r >> 2) + 4 * rq) * PS + ni * 32 + col] = acc[mi][ni][r];
            __builtin_amdgcn_wave_barrier();
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const int rl = pr + 4 * i;
                const int b = m0 + wm * 128 + mi * 32 + rl;
                float4 v = *(const float4*)(patch + rl * PS + pc);
                v.x += bv4.x; v.y += bv4.y; v.z += bv4.z; v.w += bv4.w;
                if (b < B && hcol < H) *(float4*)(pre + (int64_t)b * ldp + hcol) = v;
                if (smax) {
                    // maximum of the 16-column strip this quad of lanes covers (two DPP quad permutes): the TopK
                    // kernel reads these 4 bytes per 64 instead of the strip unless the strip can hold a winner
                    float m = fmaxf(fmaxf(v.x, v.y), fmaxf(v.z, v.w));
                    m = fmaxf(m, __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(m), 0xB1, 0xF, 0xF, false)));
                    m = fmaxf(m, __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(m), 0x4E, 0xF, 0xF, false)));
                    if ((lane & 3) == 0 && b < B && hcol < H) smax[(int64_t)b * (H >> 4) + (hcol >> 4)] = m;
                }
            }
            __builtin_amdgcn_wave_barrier();
        }
        __syncthreads();  // patches read: the buffers may take the next tile's operands
    }
}

// ------------------------------------------------------------------------------------------------
// encode_gemm256d_kernel: encode_gemm256p_kernel with both operands staged by LDS-DMA into the swizzled,
// unpadded image (wsae_mfma.h): no staging registers, no ds_write pass, 8 one-KB pieces per wave and K step,
// issued before the MFMAs of the previous step.  LDS: stage 0 at [0, 64 KB), stage 1 at [72 KB, 136 KB); the
// epilogue's transpose patches (69.6 KB) take stage 0's place, so the next tile's first slabs can already be
// landing in stage 1 while the current tile is stored: with an even number of K steps (checked by the launcher)
// every tile starts in stage 1, ends its MFMAs in stage 0, and finds the next tile's first slab in stage 1.
// ------------------------------------------------------------------------------------------------
#define G256D_STAGE (2 * 256 * SWZ_ROW_BYTES)           // A tile + W tile = 64 KB
#define G256D_STAGE1 (72 * 1024)
#define G256D_LDS (G256D_STAGE1 + G256D_STAGE)          // 136 KB

// Generic form: C[M][N] (+ z * cz) = A[M][K range z] . Bt[N][K range z]^T (+ bias[n] on z = 0); the encoder GEMM is
// (A = xb, Bt = W_e, K = D, one K range); the ReLU SAE's dense GEMMs use the split-K form (wsae_relu.hip).
// B, H, D below are M, N and the K extent of ONE range (kper); lda / ldb / ldp the leading dimensions.
template <typename T>
__global__ void __launch_bounds__(512)
encode_gemm256d_kernel(const T* __restrict__ xb, int64_t lda, const T* __restrict__ W, int64_t ldb,
                       const float* __restrict__ bias, float* __restrict__ pre, int64_t ldp, int B, int H, int D, int ntn,
                       int ntiles_mn, int nsplit, int64_t cz, float* __restrict__ smax) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int KT = SWZ_ROW_BYTES / (int)sizeof(T);
    constexpr int EPC = 16 / (int)sizeof(T);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 2, wn = wave & 3;
    const int nk = D / KT;
    const int col = lane & 31, rq = lane >> 5;
    constexpr int PS = 68;
    float* patch = (float*)smem + wave * 32 * PS;  // inside stage 0's region
    const int pr = lane >> 4, pc = (lane & 15) * 4;
    const uint32_t smem_lds = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) char*)smem;
    const int dma_r = lane >> 3, dma_s = lane & 7;

    const int ntiles = ntiles_mn * nsplit;  // work items: (K range z, tile)
    const int ntm = ntiles_mn / ntn;
    const bool xcd_walk = nsplit == 1 && (gridDim.x % 8 == 0) && (ntm % 8 == 0);
    const int xcd = blockIdx.x & 7, local = blockIdx.x >> 3, per_xcd = gridDim.x >> 3;
    auto tile_at = [&](int i) -> int {
        if (!xcd_walk) return (int)blockIdx.x + i * (int)gridDim.x;
        const int tl = local + i * per_xcd;
        if (tl >= (ntm >> 3) * ntn) return ntiles;
        return (xcd * (ntm >> 3) + tl / ntn) * ntn + tl % ntn;
    };
    // K slab k0 (absolute element offset) of tile (m0, n0) into stage st: pieces 0..31 = A rows, 32..63 = Bt rows
    auto dma = [&](int m0, int n0, int64_t k0, int st) {
        const uint32_t base = smem_lds + (st ? G256D_STAGE1 : 0);
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int piece = wave + 8 * j;            // 0..63
            const int row = (piece & 31) * 8 + dma_r;  // row inside the 256-row operand tile
            const int c = dma_s ^ ((row >> 1) & 7);
            const T* src = piece < 32 ? xb + (int64_t)min(m0 + row, B - 1) * lda + k0 + c * EPC
                                      : W + (int64_t)min(n0 + row, H - 1) * ldb + k0 + c * EPC;
            glds16(src, base + piece * 1024);
        }
    };
    auto m_of = [&](int t) { return ((t % ntiles_mn) / ntn) * 256; };
    auto n_of = [&](int t) { return ((t % ntiles_mn) % ntn) * 256; };
    auto k_of = [&](int t) { return (int64_t)(t / ntiles_mn) * D; };  // first element of the tile's K range

    int it = 0, st = 1;
    int tile = tile_at(0);
    if (tile < ntiles) dma(m_of(tile), n_of(tile), k_of(tile), st);
    for (; tile < ntiles; tile = tile_at(++it)) {
        const int m0 = m_of(tile), n0 = n_of(tile);
        const int64_t kbase = k_of(tile);
        const int z = tile / ntiles_mn;
        f32x16 acc[4][2];
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
        const int nt = min(tile_at(it + 1), ntiles - 1);
        for (int kt = 0; kt < nk; ++kt) {
            dma_wait();       // slab kt (issued one step ago) has landed
            __syncthreads();  // ... for every wave; and everybody is done reading the other stage
            // request the next slab into the other stage: the next K step of this tile, or the first of the next tile
            if (kt + 1 < nk) dma(m0, n0, kbase + (int64_t)(kt + 1) * KT, st ^ 1);
            else dma(m_of(nt), n_of(nt), k_of(nt), st ^ 1);
            const char* As = smem + (st ? G256D_STAGE1 : 0);
            Mfma256s<T>::slab(As, As + 256 * SWZ_ROW_BYTES, wm * 128, wn * 64, lane, acc);
            st ^= 1;
        }
        // the slab in flight targets stage 1 (st == 1 again); the patches take stage 0's place
        __syncthreads();  // every wave is done reading stage 0
        const int hcol = n0 + wn * 64 + pc;
        float4 bv4 = make_float4(0.f, 0.f, 0.f, 0.f);
        if (bias && z == 0 && hcol < H) bv4 = *(const float4*)(bias + hcol);
        float* Cz = pre + (int64_t)z * cz;
#pragma unroll
        for (int mi = 0; mi < 4; ++mi) {
#pragma unroll
            for (int ni = 0; ni < 2; ++ni)
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    patch[((r & 3) + 8 * (r >> 2) + 4 * rq) * PS + ni * 32 + col] = acc[mi][ni][r];
            __builtin_amdgcn_wave_barrier();
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const int rl = pr + 4 * i;
                const int b = m0 + wm * 128 + mi * 32 + rl;
                float4 v = *(const float4*)(patch + rl * PS + pc);
                v.x += bv4.x; v.y += bv4.y; v.z += bv4.z; v.w += bv4.w;
                if (b < B && hcol < H) *(float4*)(Cz + (int64_t)b * ldp + hcol) = v;
                if (smax) {
                    float m = fmaxf(fmaxf(v.x, v.y), fmaxf(v.z, v.w));
                    m = fmaxf(m, __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(m), 0xB1, 0xF, 0xF, false)));
                    m = fmaxf(m, __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(m), 0x4E, 0xF, 0xF, false)));
                    if ((lane & 3) == 0 && b < B && hcol < H) smax[(int64_t)b * (H >> 4) + (hcol >> 4)] = m;
                }
            }
            __builtin_amdgcn_wave_barrier();
        }
        // (the loop top's barrier separates these patch reads from the next slab landing in stage 0)
    }
    dma_wait();
}

// ------------------------------------------------------------------------------------------------
// TopK: one wave per row.
//   key = (orderable(value) << 32) | ~index : descending key order == (value desc, index asc).
//   1. per-lane maximum over the lane's share of the row;
//   2. T = K-th largest of the 64 lane maxima (64 distinct elements >= T, so at least K elements of
//      the row are >= T: a safe threshold, tight to roughly the 1.4*K-th largest);
//   3. compact the elements with value >= T (ballot + mbcnt prefix) into an LDS list;
//   4. bitonic sort of the list across the wave (1, 2 or 4 keys per lane), emit the first K.
//   Rows with more than 256 candidates (heavy ties, adversarial layouts) or K > 64 take the exact
//   path: bisection on the 64-bit key for the K-th largest key, then the same compaction + sort.
// ------------------------------------------------------------------------------------------------
// partner key at lane distance stride / NPL (the loops around the call are fully unrolled, so the switch folds)
template <int M>
__device__ __forceinline__ uint64_t lane_xor_key(uint64_t k, int lane) { return lane_xor_u64<M>(k, lane); }
template <int M>
__device__ __forceinline__ uint32_t lane_xor_key(uint32_t k, int lane) { return lane_xor_u32<M>(k, lane); }

template <int NPL, typename KT>
__device__ __forceinline__ KT xor_partner(KT k, int stride, int lane) {
    switch (stride / NPL) {
        case 1: return lane_xor_key<1>(k, lane);
        case 2: return lane_xor_key<2>(k, lane);
        case 4: return lane_xor_key<4>(k, lane);
        case 8: return lane_xor_key<8>(k, lane);
        case 16: return lane_xor_key<16>(k, lane);
        default: return lane_xor_key<32>(k, lane);
    }
}

// (KT = uint32_t when only the value decides - the threshold sorts - halves the work of the 64-bit (value, index) keys)
template <int NPL, typename KT = uint64_t>
__device__ __forceinline__ void wave_sort_desc(KT (&key)[NPL], int lane) {
#pragma unroll
    for (int size = 2; size <= 64 * NPL; size <<= 1) {
#pragma unroll
        for (int stride = size >> 1; stride > 0; stride >>= 1) {
            if (stride >= NPL) {
#pragma unroll
                for (int i = 0; i < NPL; ++i) {
                    const KT other = xor_partner<NPL>(key[i], stride, lane);
                    const int p = lane * NPL + i;
                    const bool desc = (p & size) == 0;
                    const bool lower = (p & stride) == 0;
                    const bool keep_max = (lower == desc);
                    const KT mx = key[i] > other ? key[i] : other;
                    const KT mn = key[i] > other ? other : key[i];
                    key[i] = keep_max ? mx : mn;
                }
            } else {
#pragma unroll
                for (int i = 0; i < NPL; ++i) {
                    if ((i & stride) == 0) {
                        const int j = i | stride;
                        const int p = lane * NPL + i;
                        const bool desc = (p & size) == 0;
                        const KT a = key[i], b = key[j];
                        const KT mx = a > b ? a : b, mn = a > b ? b : a;
                        key[i] = desc ? mx : mn;
                        key[j] = desc ? mn : mx;
                    }
                }
            }
        }
    }
}

#define TOPK_CAP 256

template <int NPL>
__device__ __forceinline__ void topk_emit(const uint64_t* list, int count, int K, int lane, float* vrow, int32_t* irow) {
    uint64_t key[NPL];
#pragma unroll
    for (int i = 0; i < NPL; ++i) {
        const int p = lane * NPL + i;
        key[i] = p < count ? list[p] : 0ull;
    }
    wave_sort_desc<NPL>(key, lane);
#pragma unroll
    for (int i = 0; i < NPL; ++i) {
        const int p = lane * NPL + i;
        if (p < K) {
            vrow[p] = ord_f32((uint32_t)(key[i] >> 32));
            irow[p] = (int32_t)(~(uint32_t)key[i]);
        }
    }
}

// compact every element with key >= kmin into list (wave-private LDS); returns the count
// (wave-uniform).  Stops storing beyond TOPK_CAP but keeps counting.
__device__ __forceinline__ int topk_compact(const float* __restrict__ row, int H, uint64_t kmin, uint64_t* list,
                                            int lane) {
    int base = 0;
    for (int e0 = 0; e0 < H; e0 += 256) {
        const int e = e0 + lane * 4;
        float4 v = make_float4(0, 0, 0, 0);
        const bool in = e < H;
        if (in) v = *(const float4*)(row + e);
        const float vv[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const uint64_t key = ((uint64_t)f32_ord(vv[c]) << 32) | (uint32_t)(~(uint32_t)(e + c));
            const bool pass = in && key >= kmin;
            const unsigned long long m = __ballot(pass);
            if (m) {
                const int pos = base + __popcll(m & ((1ull << lane) - 1ull));
                if (pass && pos < TOPK_CAP) list[pos] = key;
                base += __popcll(m);
            }
        }
    }
    return base;
}

__device__ __forceinline__ int topk_count_ge(const float* __restrict__ row, int H, uint64_t kmin, int lane) {
    int cnt = 0;
    for (int e0 = 0; e0 < H; e0 += 256) {
        const int e = e0 + lane * 4;
        if (e < H) {
            const float4 v = *(const float4*)(row + e);
            const float vv[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                const uint64_t key = ((uint64_t)f32_ord(vv[c]) << 32) | (uint32_t)(~(uint32_t)(e + c));
                cnt += key >= kmin ? 1 : 0;
            }
        }
    }
    return wave_sum_i(cnt);
}

__global__ void __launch_bounds__(256) topk_kernel(const float* __restrict__ pre, int B, int H, int K,
                                                   float* __restrict__ vals, int32_t* __restrict__ idx,
                                                   int64_t* __restrict__ step_count, int32_t* __restrict__ fallback_rows,
                                                   const int32_t* __restrict__ out_rows,
                                                   const int32_t* __restrict__ n_dev) {
    __shared__ uint64_t lists[4][TOPK_CAP];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (blockIdx.x == 0 && threadIdx.x == 0 && step_count) *step_count += 1;  // model.py:175
    const int b = blockIdx.x * 4 + wave;
    if (n_dev) B = min(B, *n_dev);
    if (b >= B) return;
    const float* row = pre + (int64_t)b * H;
    uint64_t* list = lists[wave];

    uint64_t kmin = 0;
    if (K <= 64) {
        float m = -INFINITY;
        bool any = false;
        for (int e = lane * 4; e < H; e += 256) {
            const float4 v = *(const float4*)(row + e);
            m = fmaxf(fmaxf(m, fmaxf(v.x, v.y)), fmaxf(v.z, v.w));
            any = true;
        }
        uint64_t mk[1] = {any ? ((uint64_t)f32_ord(m) << 32) : 0ull};
        wave_sort_desc<1>(mk, lane);
        // K-th largest lane maximum; lanes without elements carry key 0 (below every value)
        const uint32_t thi = __shfl((uint32_t)(mk[0] >> 32), K - 1, 64);
        kmin = (uint64_t)thi << 32;  // value >= T, any index
    }
    int count = topk_compact(row, H, kmin, list, lane);
    if (count > TOPK_CAP || count < K) {
        // exact path: the K-th largest 64-bit key by bisection from the top bit down
        if (lane == 0) atomicAdd(fallback_rows, 1);
        uint64_t prefix = 0;
        for (int bit = 63; bit >= 0; --bit) {
            const uint64_t cand = prefix | (1ull << bit);
            if (topk_count_ge(row, H, cand, lane) >= K) prefix = cand;
        }
        count = topk_compact(row, H, prefix, list, lane);  // == K exactly (keys are distinct)
    }
    const int ob = out_rows ? out_rows[b] : b;  // fallback launches write back to the original row
    float* vrow = vals + (int64_t)ob * K;
    int32_t* irow = idx + (int64_t)ob * K;
    if (count <= 64)
        topk_emit<1>(list, count, K, lane, vrow, irow);
    else if (count <= 128)
        topk_emit<2>(list, count, K, lane, vrow, irow);
    else
        topk_emit<4>(list, count, K, lane, vrow, irow);
}

// ------------------------------------------------------------------------------------------------
// topk_rows_kernel: the common case of topk_kernel (K <= 64, H <= 256*VPL <= 4096) with the row held
// in registers (VPL float4 per lane), read from HBM exactly once:
//   lane maxima -> K-th largest lane maximum T (bitonic sort over the 64 lanes) -> every lane files
//   its own elements >= T into a private LDS strip (no ballots, no atomics) -> wave prefix sum of the
//   strip lengths -> dense list -> bitonic sort -> the K best.  A lane with more than TOPK_STRIP
//   survivors, or more than TOPK_CAP in total (ties, adversarial layouts), hands the row to the
//   generic kernel's exact bisection path (same launch, same outputs).
// ------------------------------------------------------------------------------------------------
#define TOPK_STRIP 8

__device__ void topk_row_generic(const float* row, int H, int K, uint64_t* list, int lane, float* vrow, int32_t* irow,
                                 int32_t* fallback_rows);

template <int VPL>
__global__ void __launch_bounds__(256)
topk_rows_kernel(const float* __restrict__ pre, int B, int H, int K, float* __restrict__ vals,
                 int32_t* __restrict__ idx, int64_t* __restrict__ step_count, int32_t* __restrict__ fallback_rows) {
    __shared__ uint64_t lists[4][TOPK_CAP];
    __shared__ uint64_t strips[4][64 * TOPK_STRIP];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (blockIdx.x == 0 && threadIdx.x == 0 && step_count) *step_count += 1;  // model.py:175
    const int b = blockIdx.x * 4 + wave;
    if (b >= B) return;
    const float* row = pre + (int64_t)b * H;
    uint64_t* list = lists[wave];
    uint64_t* strip = strips[wave] + lane * TOPK_STRIP;

    float4 v[VPL];
    float m = -INFINITY;
#pragma unroll
    for (int i = 0; i < VPL; ++i) {
        const int e = lane * 4 + 256 * i;
        v[i] = e < H ? *(const float4*)(row + e) : make_float4(-INFINITY, -INFINITY, -INFINITY, -INFINITY);
        m = fmaxf(fmaxf(m, fmaxf(v[i].x, v[i].y)), fmaxf(v[i].z, v[i].w));
    }
    uint32_t mk[1] = {f32_ord(m)};
    wave_sort_desc<1, uint32_t>(mk, lane);
    const uint32_t thi = __shfl(mk[0], K - 1, 64);  // ord(T): 64 distinct elements are >= T

    int cnt = 0;
#pragma unroll
    for (int i = 0; i < VPL; ++i) {
        const float vv[4] = {v[i].x, v[i].y, v[i].z, v[i].w};
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const uint32_t o = f32_ord(vv[c]);
            if (o >= thi && lane * 4 + 256 * i + c < H) {
                if (cnt < TOPK_STRIP) strip[cnt] = ((uint64_t)o << 32) | (uint32_t)(~(uint32_t)(lane * 4 + 256 * i + c));
                ++cnt;
            }
        }
    }
    int incl = cnt;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const int n = __shfl_up(incl, o, 64);
        if (lane >= o) incl += n;
    }
    const int total = __shfl(incl, 63, 64);
    float* vrow = vals + (int64_t)b * K;
    int32_t* irow = idx + (int64_t)b * K;
    if (__any(cnt > TOPK_STRIP) || total > TOPK_CAP || total < K) {
        topk_row_generic(row, H, K, list, lane, vrow, irow, fallback_rows);
        return;
    }
    const int off = incl - cnt;
    for (int s_ = 0; s_ < cnt; ++s_) list[off + s_] = strip[s_];
    __builtin_amdgcn_wave_barrier();
    if (total <= 64)
        topk_emit<1>(list, total, K, lane, vrow, irow);
    else if (total <= 128)
        topk_emit<2>(list, total, K, lane, vrow, irow);
    else
        topk_emit<4>(list, total, K, lane, vrow, irow);
}

// exact path shared with topk_kernel: bisection on the 64-bit key for the K-th largest key
__device__ void topk_row_generic(const float* row, int H, int K, uint64_t* list, int lane, float* vrow, int32_t* irow,
                                 int32_t* fallback_rows) {
    if (lane == 0) atomicAdd(fallback_rows, 1);
    uint64_t prefix = 0;
    for (int bit = 63; bit >= 0; --bit) {
        const uint64_t cand = prefix | (1ull << bit);
        if (topk_count_ge(row, H, cand, lane) >= K) prefix = cand;
    }
    const int count = topk_compact(row, H, prefix, list, lane);  // == K exactly (keys are distinct)
    if (count <= 64)
        topk_emit<1>(list, count, K, lane, vrow, irow);
    else
        topk_emit<2>(list, count, K, lane, vrow, irow);
}

// ------------------------------------------------------------------------------------------------
// topk_strips_kernel: TopK guided by the strip maxima the GEMM epilogue leaves (smax [B][H/16]).
//   T = K-th largest of the 64 lane maxima of the row's strip maxima: at least K strips - hence at least K
//   distinct elements - are >= T, and every element >= T lives in a strip whose maximum is >= T.  So only
//   those strips (typically K .. 1.5 K of the H/16) are read from the [B,H] matrix: 4 lanes x 16 bytes per
//   strip, 16 strips per load instruction.  ~3 KB per row instead of 12 KB at H = 3072 - the same rows,
//   the same keys, the same sort as topk_rows_kernel; anything unusual (more than TS_MAX_STRIPS candidate
//   strips or TOPK_CAP candidates) goes to the exact full-row path.
// ------------------------------------------------------------------------------------------------
#define TS_MAX_STRIPS 128

// NTOP = strip maxima each lane contributes to the threshold: 1 -> T = K-th largest of 64 (K <= 32 in practice:
// at K = 64 that would be the smallest lane maximum and nearly every strip would qualify), 2 -> K-th largest of
// the 128 values "largest and second largest strip maximum of every lane" (distinct strips, so still >= K
// distinct elements >= T).
template <int SPL, int NTOP>  // SPL = strip maxima per lane: H / 16 <= 64 * SPL
__global__ void __launch_bounds__(256)
topk_strips_kernel(const float* __restrict__ pre, const float* __restrict__ smax, int B, int H, int K,
                   float* __restrict__ vals, int32_t* __restrict__ idx, int64_t* __restrict__ step_count,
                   int32_t* __restrict__ fallback_rows) {
    __shared__ uint64_t lists[4][TOPK_CAP];
    __shared__ int strips[4][TS_MAX_STRIPS];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (blockIdx.x == 0 && threadIdx.x == 0 && step_count) *step_count += 1;  // model.py:175
    const int b = blockIdx.x * 4 + wave;
    if (b >= B) return;
    const int ns = H >> 4;
    const float* row = pre + (int64_t)b * H;
    const float* srow = smax + (int64_t)b * ns;
    uint64_t* list = lists[wave];
    int* slist = strips[wave];
    float* vrow = vals + (int64_t)b * K;
    int32_t* irow = idx + (int64_t)b * K;

    float sm[SPL];
    float m = -INFINITY, m2 = -INFINITY;
#pragma unroll
    for (int i = 0; i < SPL; ++i) {
        const int s = lane + 64 * i;
        sm[i] = s < ns ? srow[s] : -INFINITY;
        m2 = fmaxf(m2, fminf(m, sm[i]));
        m = fmaxf(m, sm[i]);
    }
    uint32_t thi;  // ord(T)
    if constexpr (NTOP == 1) {
        uint32_t mk[1] = {f32_ord(m)};
        wave_sort_desc<1, uint32_t>(mk, lane);
        thi = __shfl(mk[0], K - 1, 64);
    } else {
        uint32_t mk[2] = {f32_ord(m), f32_ord(m2)};
        wave_sort_desc<2, uint32_t>(mk, lane);  // position p of the descending order sits in lane p / 2, slot p % 2
        const uint32_t lo = __shfl(mk[0], (K - 1) >> 1, 64), hi = __shfl(mk[1], (K - 1) >> 1, 64);
        thi = ((K - 1) & 1) ? hi : lo;
    }

    // candidate strips -> wave-private list (ballot prefix per i)
    int nstr = 0;
#pragma unroll
    for (int i = 0; i < SPL; ++i) {
        const bool pass = lane + 64 * i < ns && f32_ord(sm[i]) >= thi;
        const unsigned long long mask = __ballot(pass);
        const int pos = nstr + __popcll(mask & ((1ull << lane) - 1ull));
        if (pass && pos < TS_MAX_STRIPS) slist[pos] = lane + 64 * i;
        nstr += __popcll(mask);
    }
    if (nstr > TS_MAX_STRIPS) {
        topk_row_generic(row, H, K, list, lane, vrow, irow, fallback_rows);
        return;
    }
    __builtin_amdgcn_wave_barrier();
    // read the candidate strips, 16 per pass: lane l -> strip slist[base + l / 4], float4 number l & 3
    int total = 0;
    for (int base = 0; base < nstr; base += 16) {
        const int si = base + (lane >> 2);
        const bool in = si < nstr;
        const int s = in ? slist[si] : 0;
        const int e = s * 16 + (lane & 3) * 4;
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (in) v = *(const float4*)(row + e);
        const float vv[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const uint32_t o = f32_ord(vv[c]);
            const bool pass = in && o >= thi;
            const unsigned long long mask = __ballot(pass);
            if (mask) {
                const int pos = total + __popcll(mask & ((1ull << lane) - 1ull));
                if (pass && pos < TOPK_CAP) list[pos] = ((uint64_t)o << 32) | (uint32_t)(~(uint32_t)(e + c));
                total += __popcll(mask);
            }
        }
    }
    if (total > TOPK_CAP || total < K) {
        topk_row_generic(row, H, K, list, lane, vrow, irow, fallback_rows);
        return;
    }
    __builtin_amdgcn_wave_barrier();
    if (total <= 64)
        topk_emit<1>(list, total, K, lane, vrow, irow);
    else if (total <= 128)
        topk_emit<2>(list, total, K, lane, vrow, irow);
    else
        topk_emit<4>(list, total, K, lane, vrow, irow);
}

// ------------------------------------------------------------------------------------------------
// select_kernel: final TopK of the fused path.  One wave per row gathers the row's candidates from its
// per-tile slot groups (lanes over tiles, ballot-free prefix via wave scan), and if they are a
// complete answer (>= K candidates, no slot group overflowed, <= TOPK_CAP in total) sorts them and
// emits the K best.  Otherwise the row is appended to flag_rows for the exact fallback launch.
// ------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256)
select_kernel(const uint64_t* __restrict__ cand, const int32_t* __restrict__ cand_cnt, int32_t* __restrict__ ovf, int B,
              int ntile, int cap, int K, float* __restrict__ vals, int32_t* __restrict__ idx,
              int32_t* __restrict__ flag_rows, int32_t* __restrict__ n_flag, int64_t* __restrict__ step_count) {
    __shared__ uint64_t lists[4][TOPK_CAP];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (blockIdx.x == 0 && threadIdx.x == 0 && step_count) *step_count += 1;  // model.py:175
    const int b = blockIdx.x * 4 + wave;
    if (b >= B) return;
    uint64_t* list = lists[wave];
    int total = 0;
    bool bad = ovf[b] != 0;
    if (bad && lane == 0) ovf[b] = 0;
    for (int t0 = 0; t0 < ntile && !bad; t0 += 64) {
        const int t = t0 + lane;
        const int c = t < ntile ? cand_cnt[(int64_t)b * ntile + t] : 0;
        int incl = c;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const int n = __shfl_up(incl, o, 64);
            if (lane >= o) incl += n;
        }
        const int off = total + incl - c;
        const int sum = __shfl(incl, 63, 64);
        if (total + sum > TOPK_CAP) {
            bad = true;
            break;
        }
        if (ntile <= 8) {
            // few long lists (row-owner kernel): all lanes copy each list in turn
            for (int g = 0; g < ntile; ++g) {
                const int cg = __shfl(c, g, 64), og = __shfl(off, g, 64);
                const uint64_t* src = cand + ((int64_t)b * ntile + g) * cap;
                for (int s_ = lane; s_ < cg; s_ += 64) list[og + s_] = src[s_];
            }
        } else {
            const uint64_t* src = cand + ((int64_t)b * ntile + t) * cap;
            for (int s_ = 0; s_ < c; ++s_) list[off + s_] = src[s_];
        }
        total += sum;
    }
    if (bad || total < K) {
        if (lane == 0) flag_rows[atomicAdd(n_flag, 1)] = b;
        return;
    }
    __builtin_amdgcn_wave_barrier();
    float* vrow = vals + (int64_t)b * K;
    int32_t* irow = idx + (int64_t)b * K;
    if (total <= 64)
        topk_emit<1>(list, total, K, lane, vrow, irow);
    else if (total <= 128)
        topk_emit<2>(list, total, K, lane, vrow, irow);
    else
        topk_emit<4>(list, total, K, lane, vrow, irow);
}

// hidden = zeros; hidden[b][idx] = relu(val)        (model.py:115-116)
__global__ void __launch_bounds__(256) densify_kernel(const float* __restrict__ vals, const int32_t* __restrict__ idx,
                                                      int B, int H, int K, float* __restrict__ hidden) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= (int64_t)B * K) return;
    const int b = (int)(i / K);
    const float v = vals[i];
    hidden[(int64_t)b * H + idx[i]] = v > 0.f ? v : 0.f;
}

// fold the number of fallback rows into the stats counter and clear it for the next call
__global__ void flag_reset_kernel(int32_t* __restrict__ n_flag, int32_t* __restrict__ fb) {
    if (threadIdx.x == 0) {
        const int n = *n_flag;
        if (n) *fb += n;
        *n_flag = 0;
    }
}

// ------------------------------------------------------------------------------------------------
// host launchers
// ------------------------------------------------------------------------------------------------
static int round_up(int a, int b) { return (a + b - 1) / b * b; }

template <typename T>
static int stage_batch(wsae_ctx* c, const float* params, const void* x, int x_dtype, const int32_t* rows, int B,
                       hipStream_t st) {
    const int D = c->D;
    const int ldT = round_up(B, 128);
    dim3 sg(ceil_div(ldT, 64), ceil_div(D, 64));
    WSAE_PROF_BEGIN(c, WSAE_K_STAGE_BATCH, st);
    if (x_dtype == WSAE_DT_F32)
        stage_batch_kernel<WSAE_DT_F32, T><<<sg, 256, 0, st>>>(x, rows, params + c->off[4], (T*)c->xb, (T*)c->xT, B, D, ldT);
    else
        stage_batch_kernel<WSAE_DT_BF16, T><<<sg, 256, 0, st>>>(x, rows, params + c->off[4], (T*)c->xb, (T*)c->xT, B, D, ldT);
    WSAE_PROF_END(c, WSAE_K_STAGE_BATCH, st);
    WSAE_LAUNCH_CHECK();
    return WSAE_OK;
}

template <typename T>
static int gemm_dense(wsae_ctx* c, const float* params, int B, int nfeat, int wstride, float* pre, int ldp,
                      const int32_t* arows, const int32_t* n_dev, hipStream_t st) {
    const T* W = sizeof(T) == 2 ? (const T*)c->We_bf16 : (const T*)(params + c->off[0]);
    const float* bias = sizeof(T) == 2 ? c->c_fold : params + c->off[2];
    dim3 gg(ceil_div(nfeat, TILE_N), ceil_div(B, TILE_M));
    WSAE_PROF_BEGIN(c, WSAE_K_ENCODE_GEMM, st);
    if (wstride == 1 && !arows && !n_dev && B >= 2048 && nfeat % 256 == 0 && c->D % Mfma<T>::KT == 0) {
        const int ntn = nfeat / 256, ntiles = ntn * ceil_div(B, 256);
        static int cus = 0;
        if (!cus && (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, c->device) != hipSuccess || cus < 1))
            cus = 256;
        // strip maxima for the strip-guided TopK kernel when this is the full pre-activation matrix of the ctx
        float* smax = (pre == c->pre && nfeat == c->H && ldp == c->H) ? c->smax : nullptr;
        static const bool no_dma = getenv("WSAE_GEMM_REGSTAGE") != nullptr;  // A/B: the register-staged kernel (+6 us at cfg 2)
        if (!no_dma && (c->D / Mfma<T>::KT) % 2 == 0)
            encode_gemm256d_kernel<T><<<min(ntiles, cus), 512, G256D_LDS, st>>>((const T*)c->xb, c->D, W, c->D, bias, pre, ldp, B,
                                                                               nfeat, c->D, ntn, ntiles, 1, 0, smax);
        else
        encode_gemm256p_kernel<T><<<min(ntiles, cus), 512, 4 * T256_LDS, st>>>((const T*)c->xb, W, bias, pre, ldp, B, nfeat,
                                                                            c->D, ntn, ntiles, smax);
        if (smax) c->smax_valid = 1;
    } else {
        if (pre == c->pre) c->smax_valid = 0;
        encode_gemm_kernel<T, GEMM_DENSE><<<gg, 256, 2 * TILE_LDS_BYTES, st>>>(
            (const T*)c->xb, W, bias, pre, ldp, B, nfeat, c->D, wstride, arows, n_dev, nullptr, 0, nullptr, nullptr,
            nullptr);
    }
    WSAE_PROF_END(c, WSAE_K_ENCODE_GEMM, st);
    WSAE_LAUNCH_CHECK();
    return WSAE_OK;
}

template <typename T>
static int stage_and_gemm(wsae_ctx* c, const float* params, const void* x, int x_dtype, const int32_t* rows, int B,
                          float* pre, hipStream_t st) {
    int rc = stage_batch<T>(c, params, x, x_dtype, rows, B, st);
    if (rc) return rc;
    return gemm_dense<T>(c, params, B, c->H, 1, pre, c->H, nullptr, nullptr, st);
}

// Fused-TopK path: sample pass -> per-row threshold -> filtering GEMM -> select -> exact fallback
// for the (rare) rows the filter could not settle.  Never writes the [B,H] pre-activation matrix.
//   sample: every 8th feature, S = H/8 of them; thr[b] = KS-th largest sample pre-activation with
//   KS = ceil(3*K/8): the expected rank of thr[b] in the full row is ~3K, so ~3K candidates survive
//   the filter and a row falls short of K candidates with probability ~1e-3.  Correctness never
//   depends on these odds: a row with < K candidates (or an overflowed slot group) is recomputed exactly.
template <typename T>
static int encode_topk_fused(wsae_ctx* c, const float* params, int B, float* vals, int32_t* idx, int64_t* step_count,
                             int32_t* fb, hipStream_t st) {
    const int H = c->H, K = c->K, D = c->D;
    const int S = H / 8, KS = max(4, (3 * K + 7) / 8);
    int rc = gemm_dense<T>(c, params, B, S, 8, c->pre, S, nullptr, nullptr, st);
    if (rc) return rc;
    WSAE_PROF_BEGIN(c, WSAE_K_TOPK, st);
    topk_kernel<<<ceil_div(B, 4), 256, 0, st>>>(c->pre, B, S, KS, c->thr_vals, c->thr_idx, nullptr, fb, nullptr, nullptr);
    WSAE_PROF_END(c, WSAE_K_TOPK, st);
    WSAE_LAUNCH_CHECK();
    const T* W = sizeof(T) == 2 ? (const T*)c->We_bf16 : (const T*)(params + c->off[0]);
    const float* bias = sizeof(T) == 2 ? c->c_fold : params + c->off[2];
    const int ntile = ceil_div(H, TILE_N);
    int ngroup = ntile, cap = CAND_SLOTS;  // candidate lists: [row][ngroup][cap]
    WSAE_PROF_BEGIN(c, WSAE_K_ENCODE_FILTER, st);
    if (H % 256 == 0 && D % Mfma<T>::KT == 0 && B >= 2048) {
        ngroup = H / 256;
        cap = 2 * CAND_SLOTS;
        dim3 g2(ngroup, ceil_div(B, 256));
        encode_gemm256_kernel<T, GEMM_FILTER><<<g2, 512, 4 * T256_LDS + 2048, st>>>(
            (const T*)c->xb, W, bias, nullptr, 0, B, H, D, c->thr_vals + (KS - 1), KS, c->cand, c->cand_cnt, c->cand_ovf,
            cap);
    } else {
        dim3 gg(ntile, ceil_div(B, TILE_M));
        encode_gemm_kernel<T, GEMM_FILTER><<<gg, 256, 2 * TILE_LDS_BYTES + 512, st>>>(
            (const T*)c->xb, W, bias, nullptr, 0, B, H, D, 1, nullptr, nullptr, c->thr_vals + (KS - 1), KS, c->cand,
            c->cand_cnt, c->cand_ovf);
    }
    WSAE_PROF_END(c, WSAE_K_ENCODE_FILTER, st);
    WSAE_LAUNCH_CHECK();
    int32_t* n_flag = c->counters + 8;
    WSAE_PROF_BEGIN(c, WSAE_K_SELECT, st);
    select_kernel<<<ceil_div(B, 4), 256, 0, st>>>(c->cand, c->cand_cnt, c->cand_ovf, B, ngroup, cap, K, vals, idx,
                                                  c->flag_rows, n_flag, step_count);
    WSAE_PROF_END(c, WSAE_K_SELECT, st);
    WSAE_LAUNCH_CHECK();
    // exact fallback over the flagged rows (device-side count; blocks beyond it exit at once)
    rc = gemm_dense<T>(c, params, B, H, 1, c->pre, H, c->flag_rows, n_flag, st);
    if (rc) return rc;
    topk_kernel<<<ceil_div(B, 4), 256, 0, st>>>(c->pre, B, H, K, vals, idx, nullptr, fb, c->flag_rows, n_flag);
    WSAE_LAUNCH_CHECK();
    flag_reset_kernel<<<1, 64, 0, st>>>(n_flag, fb);
    WSAE_LAUNCH_CHECK();
    return WSAE_OK;
}

static int check_batch(const wsae_ctx* c, const void* x, int x_dtype, int B, const char* who) {
    WSAE_REQUIRE(c && x, "%s: null argument", who);
    WSAE_REQUIRE(B >= 1 && B <= c->maxB, "%s: batch %d outside [1, max_batch=%d]", who, B, c->maxB);
    WSAE_REQUIRE(x_dtype == WSAE_DT_F32 || x_dtype == WSAE_DT_BF16, "%s: unknown activation dtype %d", who, x_dtype);
    return WSAE_OK;
}

// Persistent LDS-DMA GEMM for other callers (wsae_relu.hip): C[M][N] (+ z * cz) = A . Bt^T over nsplit K ranges of
// K / nsplit elements each.  Returns false when the shape does not qualify (caller falls back to its own kernel).
template <typename T>
static bool gemm256d_try(wsae_ctx* c, const void* A, int64_t lda, const void* Bt, int64_t ldb, const float* bias, float* C,
                         int64_t ldc, int M, int N, int K, int nsplit, int64_t cz, hipStream_t st) {
    constexpr int KT = SWZ_ROW_BYTES / (int)sizeof(T);
    if (M < 512 || N < 128 || N % 4 || K % nsplit || (K / nsplit) % (2 * KT) || lda % (16 / (int)sizeof(T)) ||
        ldb % (16 / (int)sizeof(T)) || ldc % 4)
        return false;
    static int cus = 0;
    if (!cus && (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, c->device) != hipSuccess || cus < 1)) cus = 256;
    const int ntn = ceil_div(N, 256), ntiles_mn = ntn * ceil_div(M, 256);
    encode_gemm256d_kernel<T><<<min(ntiles_mn * nsplit, cus), 512, G256D_LDS, st>>>((const T*)A, lda, (const T*)Bt, ldb, bias, C, ldc,
                                                                                   M, N, K / nsplit, ntn, ntiles_mn, nsplit, cz,
                                                                                   nullptr);
    return true;
}

bool wsae_internal_gemm256d(wsae_ctx* c, const void* A, int64_t lda, const void* Bt, int64_t ldb, const float* bias, float* C,
                            int64_t ldc, int M, int N, int K, int nsplit, int64_t cz, hipStream_t st) {
    return c->prec == WSAE_PREC_BF16 ? gemm256d_try<bf16_t>(c, A, lda, Bt, ldb, bias, C, ldc, M, N, K, nsplit, cz, st)
                                     : gemm256d_try<float>(c, A, lda, Bt, ldb, bias, C, ldc, M, N, K, nsplit, cz, st);
}

int wsae_internal_stage_and_gemm(wsae_ctx* ctx, const float* params, const void* x, int x_dtype, const int32_t* rows,
                                 int B, float* pre, hipStream_t st) {
    return ctx->prec == WSAE_PREC_BF16 ? stage_and_gemm<bf16_t>(ctx, params, x, x_dtype, rows, B, pre, st)
                                       : stage_and_gemm<float>(ctx, params, x, x_dtype, rows, B, pre, st);
}

extern "C" int wsae_encode_dense(wsae_ctx* ctx, const float* params, const void* x, int32_t x_dtype,
                                 const int32_t* rows, int32_t B, float* pre, void* stream) {
    int rc = check_batch(ctx, x, x_dtype, B, "wsae_encode_dense");
    if (rc) return rc;
    WSAE_REQUIRE(params && pre, "wsae_encode_dense: null argument");
    hipStream_t st = (hipStream_t)stream;
    return ctx->prec == WSAE_PREC_BF16 ? stage_and_gemm<bf16_t>(ctx, params, x, x_dtype, rows, B, pre, st)
                                       : stage_and_gemm<float>(ctx, params, x, x_dtype, rows, B, pre, st);
}

extern "C" int wsae_encode_topk(wsae_ctx* ctx, const float* params, const void* x, int32_t x_dtype,
                                const int32_t* rows, int32_t B, float* vals, int32_t* idx, int64_t* step_count,
                                wsae_stats* stats, void* stream) {
    int rc = check_batch(ctx, x, x_dtype, B, "wsae_encode_topk");
    if (rc) return rc;
    WSAE_REQUIRE(params && vals && idx, "wsae_encode_topk: null argument");
    hipStream_t st = (hipStream_t)stream;
    int32_t* fb = stats ? &stats->topk_fallback_rows : ctx->counters;
    // fused TopK pays once the [B,H] round trip dominates; small shapes keep the dense two-kernel path
    const bool fused = ctx->H >= 1024 && ctx->H % 128 == 0 && ctx->K <= 64 && B >= 512 && ctx->fused_topk;
    if (fused) {
        rc = ctx->prec == WSAE_PREC_BF16 ? stage_batch<bf16_t>(ctx, params, x, x_dtype, rows, B, st)
                                         : stage_batch<float>(ctx, params, x, x_dtype, rows, B, st);
        if (rc) return rc;
        return ctx->prec == WSAE_PREC_BF16 ? encode_topk_fused<bf16_t>(ctx, params, B, vals, idx, step_count, fb, st)
                                           : encode_topk_fused<float>(ctx, params, B, vals, idx, step_count, fb, st);
    }
    rc = ctx->prec == WSAE_PREC_BF16 ? stage_and_gemm<bf16_t>(ctx, params, x, x_dtype, rows, B, ctx->pre, st)
                                     : stage_and_gemm<float>(ctx, params, x, x_dtype, rows, B, ctx->pre, st);
    if (rc) return rc;
    WSAE_PROF_BEGIN(ctx, WSAE_K_TOPK, st);
    const int vpl = ceil_div(ctx->H, 256);
    static const bool no_strips = getenv("WSAE_TOPK_ROWS") != nullptr;  // A/B runs
    const int ns = ctx->H / 16;
    if (ctx->smax_valid && !no_strips && ctx->K <= 64 && ctx->H % 16 == 0 && ns >= 128 && ns <= 1024) {
#define TS_ARGS ctx->pre, ctx->smax, B, ctx->H, ctx->K, vals, idx, step_count, fb
        const dim3 tg(ceil_div(B, 4));
        if (ctx->K <= 32) {
            if (ns <= 192) topk_strips_kernel<3, 1><<<tg, 256, 0, st>>>(TS_ARGS);
            else if (ns <= 512) topk_strips_kernel<8, 1><<<tg, 256, 0, st>>>(TS_ARGS);
            else topk_strips_kernel<16, 1><<<tg, 256, 0, st>>>(TS_ARGS);
        } else {
            if (ns <= 192) topk_strips_kernel<3, 2><<<tg, 256, 0, st>>>(TS_ARGS);
            else if (ns <= 512) topk_strips_kernel<8, 2><<<tg, 256, 0, st>>>(TS_ARGS);
            else topk_strips_kernel<16, 2><<<tg, 256, 0, st>>>(TS_ARGS);
        }
#undef TS_ARGS
    } else if (ctx->K <= 64 && vpl <= 4)
        topk_rows_kernel<4><<<ceil_div(B, 4), 256, 0, st>>>(ctx->pre, B, ctx->H, ctx->K, vals, idx, step_count, fb);
    else if (ctx->K <= 64 && vpl <= 12)
        topk_rows_kernel<12><<<ceil_div(B, 4), 256, 0, st>>>(ctx->pre, B, ctx->H, ctx->K, vals, idx, step_count, fb);
    else if (ctx->K <= 64 && vpl <= 16)
        topk_rows_kernel<16><<<ceil_div(B, 4), 256, 0, st>>>(ctx->pre, B, ctx->H, ctx->K, vals, idx, step_count, fb);
    else
        topk_kernel<<<ceil_div(B, 4), 256, 0, st>>>(ctx->pre, B, ctx->H, ctx->K, vals, idx, step_count, fb, nullptr,
                                                    nullptr);
    WSAE_PROF_END(ctx, WSAE_K_TOPK, st);
    WSAE_LAUNCH_CHECK();
    return WSAE_OK;
}

extern "C" int wsae_densify(wsae_ctx* ctx, const float* vals, const int32_t* idx, int32_t B, float* hidden,
                            void* stream) {
    WSAE_REQUIRE(ctx && vals && idx && hidden && B >= 1, "wsae_densify: bad argument");
    hipStream_t st = (hipStream_t)stream;
    WSAE_HIP_CHECK(hipMemsetAsync(hidden, 0, (size_t)B * ctx->H * 4, st));
    const int64_t n = (int64_t)B * ctx->K;
    densify_kernel<<<(unsigned)ceil_div64(n, 256), 256, 0, st>>>(vals, idx, B, ctx->H, ctx->K, hidden);
    WSAE_LAUNCH_CHECK();
    return WSAE_OK;
}
