// Encode path: batch staging, pre-activation GEMM on MFMA, per-row TopK with wavefront reductions.
//   reference: TopKSAE.encode, src/whisper_sae/sae/model.py:98-118
#include "wsae_common.h"
#include "wsae_mfma.h"
#include "wsae_topk.h"

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
                                                          T* __restrict__ xT, int B, int D, int ldT,
                                                          int64_t* __restrict__ step_count) {
    __shared__ float tile[64][65];
    if (step_count && blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0) *step_count += 1;  // model.py:175
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
    if (!xT) return;  // (callers that read the staged rows row-major only: the ReLU path's row-major-GEMM flow)
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

// bf16 rows -> the contiguous bf16 batch, nothing else (no transposed copy, no pre-bias: the row-major-GEMM flow of the ReLU
// path).  One 16-byte chunk per thread and pass, two passes in flight: a row-list entry and one load deep, where the tile
// kernel above walks four dependent (row index, 8-byte load) pairs per thread (1-2 us of the ReLU step at B = 16384, D = 384).
__global__ void __launch_bounds__(256) stage_rows_copy_kernel(const uint4* __restrict__ x, const int32_t* __restrict__ rows,
                                                              uint4* __restrict__ xb, int64_t nchunks, int cpr,
                                                              int64_t* __restrict__ step_count) {
    if (step_count && blockIdx.x == 0 && threadIdx.x == 0) *step_count += 1;  // model.py:175
    const int64_t i0 = (int64_t)blockIdx.x * 512 + threadIdx.x, i1 = i0 + 256;
    const int64_t j0 = min(i0, nchunks - 1), j1 = min(i1, nchunks - 1);
    const int b0 = (int)(j0 / cpr), b1 = (int)(j1 / cpr);
    const int64_t r0 = rows ? (int64_t)rows[b0] : (int64_t)b0, r1 = rows ? (int64_t)rows[b1] : (int64_t)b1;
    // (plain loads: the rows are read again by the residual pass)
    const uint4 v0 = x[r0 * cpr + (j0 - (int64_t)b0 * cpr)];
    const uint4 v1 = x[r1 * cpr + (j1 - (int64_t)b1 * cpr)];
    if (i0 < nchunks) xb[i0] = v0;
    if (i1 < nchunks) xb[i1] = v1;
}

// ------------------------------------------------------------------------------------------------
// encode_gemm256d_kernel: encode_gemm256p_kernel with both operands staged by LDS-DMA into the swizzled,
// unpadded image (wsae_mfma.h): no staging registers, no ds_write pass, 8 one-KB pieces per wave and K step,
// issued before the MFMAs of the previous step.  LDS: stage 0 at [0, 64 KB), stage 1 at [72 KB, 136 KB); the
// epilogue's transpose patches (69.6 KB) take stage 0's place, so the next tile's first slabs can already be
// landing in stage 1 while the current tile is stored: with an even number of K steps (checked by the launcher)
// every tile starts in stage 1, ends its MFMAs in stage 0, and finds the next tile's first slab in stage 1.
// ------------------------------------------------------------------------------------------------
// Diagnostic build only (-DWSAE_ENC_STAMPS, never the product library; profiles/tools/stamps_encode.py): per-phase cycle
// sums of the persistent encoder GEMM, summed over all waves.
#ifdef WSAE_ENC_STAMPS
__device__ unsigned long long enc_stamp_sum[8];
#define EN_T(i)                                                                                     \
    {                                                                                               \
        __builtin_amdgcn_sched_barrier(0);                                                          \
        unsigned long long t_;                                                                      \
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_)::"memory");                  \
        __builtin_amdgcn_sched_barrier(0);                                                          \
        en_acc[i] += t_ - en_last;                                                                  \
        en_last = t_;                                                                               \
    }
extern "C" int wsae_debug_enc_stamps(double* out, int reset) {
    unsigned long long h[8];
    if (hipMemcpyFromSymbol(h, HIP_SYMBOL(enc_stamp_sum), sizeof(h)) != hipSuccess) return -1;
    for (int i = 0; i < 8; ++i) out[i] = (double)h[i];
    if (reset) {
        unsigned long long z[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        (void)hipMemcpyToSymbol(HIP_SYMBOL(enc_stamp_sum), z, sizeof(z));
    }
    return 0;
}
#else
#define EN_T(i)
#endif
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
                       int ntiles_mn, int nsplit, int64_t cz, float* __restrict__ smax, const int32_t* __restrict__ arows,
                       int64_t* __restrict__ step_count, const float* __restrict__ rscale = nullptr,
                       const float* __restrict__ cscale = nullptr, uint32_t* __restrict__ tmin = nullptr, int tmin_cur = 0,
                       float tg_fixed_s = 0.f) {
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
    // (the encoder forward gathers its batch rows straight from the activation ring: A row m is xb[arows[m]]; it is
    // then also the first kernel of the step and advances the dead-feature clock of model.py:175)
    if (step_count && blockIdx.x == 0 && tid == 0) *step_count += 1;
    // strip store threshold (wsae_topk.h): from the smallest row thresholds the last two TopK launches saw; this launch
    // re-arms the slot its own TopK launch will fill (which nobody reads meanwhile)
    // (wave 0 works it out - its loads overlap the first slab's DMA - and the others pick it up after the K loop's barriers)
    __shared__ float tg_sh;
    if (wave == 0) {
        float s_now = 0.f;
        const float t = tmin ? strip_store_threshold(tmin, tmin_cur, lane, tg_fixed_s, &s_now) : -INFINITY;
        if (lane == 0) tg_sh = t;
        if (tmin && blockIdx.x == 0) {
            uint32_t* grp = tmin + tmin_cur * TG_GROUP_WORDS;
            grp[lane * TG_SLOT_STRIDE] = 0xFFFFFFFFu;
            if (lane == 0) {
                grp[TG_HDR_S] = __float_as_uint(s_now);
                grp[TG_HDR_MISSES] = 0u;
                __builtin_nontemporal_store(t, (float*)tmin + TG_USED);  // what the TopK launch checks rows against
            }
        }
    }

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
    // source rows of this lane's four A pieces (j = 0..3: piece = wave + 8 j < 32) for the tile at m0
    auto a_rows = [&](int m0, int (&ar)[4]) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int m = min(m0 + (wave + 8 * j) * 8 + dma_r, B - 1);
            ar[j] = arows ? arows[m] : m;
        }
    };
    // K slab k0 (absolute element offset) of tile (rows ar, n0) into stage st: pieces 0..31 = A rows, 32..63 = Bt rows
    auto dma = [&](const int (&ar)[4], int n0, int64_t k0, int st) {
        const uint32_t base = smem_lds + (st ? G256D_STAGE1 : 0);
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int piece = wave + 8 * j;            // 0..63
            const int row = (piece & 31) * 8 + dma_r;  // row inside the 256-row operand tile
            const int c = dma_s ^ ((row >> 1) & 7);
            const T* src = j < 4 ? xb + (int64_t)ar[j & 3] * lda + k0 + c * EPC
                                 : W + (int64_t)min(n0 + row, H - 1) * ldb + k0 + c * EPC;
            glds16(src, base + piece * 1024);
        }
    };
    auto m_of = [&](int t) { return ((t % ntiles_mn) / ntn) * 256; };
    auto n_of = [&](int t) { return ((t % ntiles_mn) % ntn) * 256; };
    auto k_of = [&](int t) { return (int64_t)(t / ntiles_mn) * D; };  // first element of the tile's K range

#ifdef WSAE_ENC_STAMPS
    unsigned long long en_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0}, en_last = __builtin_amdgcn_s_memtime();
#endif
    int it = 0, st = 1;
    int tile = tile_at(0);
    int ar[4] = {0, 0, 0, 0}, ar_n[4];
    if (tile < ntiles) {
        a_rows(m_of(tile), ar);
        dma(ar, n_of(tile), k_of(tile), st);
    }
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
        a_rows(m_of(nt), ar_n);  // (fetched a whole tile ahead of their use)
        EN_T(0)
        for (int kt = 0; kt < nk; ++kt) {
            dma_wait();       // slab kt (issued one step ago) has landed
            EN_T(1)
            __syncthreads();  // ... for every wave; and everybody is done reading the other stage
            EN_T(2)
            // request the next slab into the other stage: the next K step of this tile, or the first of the next tile
            if (kt + 1 < nk) dma(ar, n0, kbase + (int64_t)(kt + 1) * KT, st ^ 1);
            else dma(ar_n, n_of(nt), k_of(nt), st ^ 1);
            EN_T(3)
            const char* As = smem + (st ? G256D_STAGE1 : 0);
            Mfma256s<T>::slab(As, As + 256 * SWZ_ROW_BYTES, wm * 128, wn * 64, lane, acc);
            st ^= 1;
            EN_T(4)
        }
        // the slab in flight targets stage 1 (st == 1 again); the patches take stage 0's place
        __syncthreads();  // every wave is done reading stage 0
        EN_T(5)
        const int hcol = n0 + wn * 64 + pc;
        float4 bv4 = make_float4(0.f, 0.f, 0.f, 0.f);
        if (bias && z == 0 && hcol < H) bv4 = *(const float4*)(bias + hcol);
        // Interior tiles of the encoder forward (every tile at the benchmarked shapes): no per-element predicates,
        // buffer stores addressed by one per-lane byte offset + a scalar row offset, strip maxima on v_max3 /
        // v_max_f32_dpp (fmaxf would canonicalise every operand first): ~130 vector instructions per row group
        // against ~250 vector + ~150 scalar ones in the general form below.
        const float tg = tg_sh;
        const bool fast = !rscale && smax && z == 0 && m0 + 256 <= B && n0 + 256 <= H && (int64_t)B * ldp < (1ll << 29);
        if (fast) {
            typedef int i32x4 __attribute__((ext_vector_type(4)));
            const __amdgpu_buffer_rsrc_t r_pre = __builtin_amdgcn_make_buffer_rsrc(pre, 0, 0x7fffffff, 0x00020000);
            const __amdgpu_buffer_rsrc_t r_sm = __builtin_amdgcn_make_buffer_rsrc(smax, 0, 0x7fffffff, 0x00020000);
            const int hs = H >> 4;
            const int b_lane = m0 + wm * 128 + pr;
            const int off_pre = (int)(((int64_t)b_lane * ldp + hcol) * 4);
            const int off_sm = (b_lane * hs + (hcol >> 4)) * 4;
            const int row_pre = (int)ldp * 16, row_sm = hs * 16;  // bytes per 4 rows
#pragma unroll
            for (int mi = 0; mi < 4; ++mi) {
#pragma unroll
                for (int ni = 0; ni < 2; ++ni)
#pragma unroll
                    for (int r = 0; r < 16; ++r)
                        patch[((r & 3) + 8 * (r >> 2) + 4 * rq) * PS + ni * 32 + col] = acc[mi][ni][r];
                __builtin_amdgcn_wave_barrier();
                EN_T(6)
#pragma unroll
                for (int i = 0; i < 8; i += 2) {
                    f32x4 va = *(const f32x4*)(patch + (pr + 4 * i) * PS + pc);
                    f32x4 vb = *(const f32x4*)(patch + (pr + 4 * i + 4) * PS + pc);
                    va.x += bv4.x; va.y += bv4.y; va.z += bv4.z; va.w += bv4.w;
                    vb.x += bv4.x; vb.y += bv4.y; vb.z += bv4.z; vb.w += bv4.w;
                    const int q = mi * 8 + i;  // 4-row group inside the wave's 128 rows
                    float ma, mb, ta, tb;  // (two rows per block: the second chain fills the first one's DPP wait states)
                    asm volatile(
                        "v_max3_f32 %0, %4, %5, %6\n\t"
                        "v_max3_f32 %1, %8, %9, %10\n\t"
                        "v_max_f32 %0, %0, %7\n\t"
                        "v_max_f32 %1, %1, %11\n\t"
                        "s_nop 0\n\t"
                        "v_max_f32_dpp %2, %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
                        "v_max_f32_dpp %3, %1, %1 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
                        "s_nop 0\n\t"
                        "v_max_f32_dpp %0, %2, %2 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n\t"
                        "v_max_f32_dpp %1, %3, %3 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf"
                        : "=&v"(ma), "=&v"(mb), "=&v"(ta), "=&v"(tb)
                        : "v"(va.x), "v"(va.y), "v"(va.z), "v"(va.w), "v"(vb.x), "v"(vb.y), "v"(vb.z), "v"(vb.w));
                    // (all four lanes of a quad hold the strip maximum and store it to the same word: a store costs the
                    // memory pipe the same with 16 lanes as with 64, and no exec switching is needed)
                    // (NOT non-temporal: these 16-byte pieces stop being merged into lines on the way - the GEMM takes 12 us longer)
                    __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(int, ma), r_sm, off_sm, q * row_sm, 0);
                    __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(int, mb), r_sm, off_sm, (q + 1) * row_sm, 0);
                    // the strip itself only when its maximum reaches the store threshold (wsae_topk.h, "strip store
                    // threshold"): the TopK reads nothing below it, and recomputes the strip in the rare row where it must
                    // (non-temporal, aux bit 1: the strips are read once, by the TopK launch right behind, and must not push the
                    // optimizer state out of the Infinity Cache - 3.5 us of the step, half of them in update_rows)
                    if (!(ma < tg)) __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(i32x4, va), r_pre, off_pre, q * row_pre, 2);
                    if (!(mb < tg)) __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(i32x4, vb), r_pre, off_pre, (q + 1) * row_pre, 2);
                }
                __builtin_amdgcn_wave_barrier();
                EN_T(7)
            }
        } else {
            // quantised operands (fp8): C = rscale[m] * cscale[n] * acc (+ bias)
            float4 cs4 = make_float4(1.f, 1.f, 1.f, 1.f);
            if (cscale && hcol < H) cs4 = *(const float4*)(cscale + hcol);
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
                    if (rscale) {
                        const float rs = rscale[min(b, B - 1)];
                        v.x *= rs * cs4.x; v.y *= rs * cs4.y; v.z *= rs * cs4.z; v.w *= rs * cs4.w;
                    }
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
        }
        // (the loop top's barrier separates these patch reads from the next slab landing in stage 0)
#pragma unroll
        for (int j = 0; j < 4; ++j) ar[j] = ar_n[j];
    }
    dma_wait();
#ifdef WSAE_ENC_STAMPS
    if (lane == 0)
        for (int i = 0; i < 8; ++i) atomicAdd(&enc_stamp_sum[i], en_acc[i]);
#endif
}
// ------------------------------------------------------------------------------------------------
// encode_gemm_kernel: pre[b][h] = sum_d xb[b][d] * W[h][d] + bias[h]          (model.py:111)
// 128 x 128 tiles, four waves, register-staged prefetch.  Serves the shapes the persistent kernel does not:
// small batches, H not a multiple of 256, an odd number of K slabs.
// ------------------------------------------------------------------------------------------------
// A rows through a row list (the batch's rows of the activation ring, gathered by the GEMM itself): the slab_load of
// wsae_mfma.h with row m -> arows[m]
template <typename T>
__device__ __forceinline__ void slab_load_rows(SlabRegs<T>& r, const T* __restrict__ base, int64_t ld, const int32_t* __restrict__ arows,
                                               int row0, int n_rows, int k0, int k_total, int tid) {
    constexpr int EPC = 16 / (int)sizeof(T);
    auto chunk = [&](int c) -> uint4 {
        const int row = row0 + (c >> 3), k = k0 + (c & 7) * EPC;
        if (row < n_rows && k < k_total) return *(const uint4*)(base + (int64_t)arows[row] * ld + k);
        return make_uint4(0, 0, 0, 0);
    };
    r.v0 = chunk(tid); r.v1 = chunk(tid + 256); r.v2 = chunk(tid + 512); r.v3 = chunk(tid + 768);
}

template <typename T>
__global__ void __launch_bounds__(256)
encode_gemm_kernel(const T* __restrict__ xb, const T* __restrict__ W, const float* __restrict__ bias,
                   float* __restrict__ pre, int ldp, int B, int H, int D, const int32_t* __restrict__ arows,
                   int64_t* __restrict__ step_count) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* As = smem;
    char* Bs = smem + TILE_LDS_BYTES;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int m0 = blockIdx.y * TILE_M, n0 = blockIdx.x * TILE_N;
    constexpr int KT = Mfma<T>::KT;
    // (when it gathers the batch rows itself this is the first kernel of the step: it advances the clock of model.py:175)
    if (step_count && blockIdx.x == 0 && blockIdx.y == 0 && tid == 0) *step_count += 1;

    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    SlabRegs<T> ra, rb;
    if (arows) slab_load_rows<T>(ra, xb, D, arows, m0, B, 0, D, tid);
    else slab_load<T>(ra, xb, D, m0, B, 0, D, tid);
    slab_load<T>(rb, W, D, n0, H, 0, D, tid);
    const int nk = (D + KT - 1) / KT;
    for (int kt = 0; kt < nk; ++kt) {
        slab_store<T>(ra, As, tid);
        slab_store<T>(rb, Bs, tid);
        __syncthreads();
        if (kt + 1 < nk) {
            if (arows) slab_load_rows<T>(ra, xb, D, arows, m0, B, (kt + 1) * KT, D, tid);
            else slab_load<T>(ra, xb, D, m0, B, (kt + 1) * KT, D, tid);
            slab_load<T>(rb, W, D, n0, H, (kt + 1) * KT, D, tid);
        }
        Mfma<T>::slab(As, Bs, wm * 64, wn * 64, lane, acc);
        __syncthreads();
    }
    const int col = lane & 31, rq = lane >> 5;
#pragma unroll
    for (int mi = 0; mi < 2; ++mi)
#pragma unroll
        for (int ni = 0; ni < 2; ++ni) {
            const int h = n0 + wn * 64 + ni * 32 + col;
            if (h >= H) continue;
            const float bv = bias[h];
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int b = m0 + wm * 64 + mi * 32 + (r & 3) + 8 * (r >> 2) + 4 * rq;
                if (b < B) pre[(int64_t)b * ldp + h] = acc[mi][ni][r] + bv;
            }
        }
}

// ------------------------------------------------------------------------------------------------
// encode_direct_kernel (bf16, small batches - the reference YAMLs' own 64 / 128 rows): no LDS, no barrier.  A wave owns one
// 32 x 32 tile of pre (32 batch rows x 32 features); both MFMA operands are K-contiguous in memory, so every fragment is one
// 16-byte global load per lane - the x rows through the step's row list, the W_e rows from L2 - issued 8 K steps at a time.
// Grid (H / 32, ceil(B / 128)): 96 workgroups at 384 -> 3072 / B = 128, where the 128 x 128 LDS kernel has 24.
// ------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256)
encode_direct_kernel(const bf16_t* __restrict__ xa, const int32_t* __restrict__ arows, const bf16_t* __restrict__ W,
                     const float* __restrict__ bias, float* __restrict__ pre, int ldp, int B, int H, int D,
                     int64_t* __restrict__ step_count) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (step_count && blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0) *step_count += 1;  // model.py:175
    const int n0 = blockIdx.x * 32, m0 = blockIdx.y * 128 + wave * 32;
    if (m0 >= B) return;
    const int r = lane & 31, h = lane >> 5;
    const int mrow = min(m0 + r, B - 1);  // rows past the batch repeat its last row: never stored
    const int64_t arow = arows ? (int64_t)arows[mrow] : (int64_t)mrow;
    const bf16_t* ap = xa + arow * D + 8 * h;
    const bf16_t* bp = W + (int64_t)min(n0 + r, H - 1) * D + 8 * h;
    f32x16 acc;
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] = 0.f;
    const int nk = D / 16;
    for (int k0 = 0; k0 < nk; k0 += 8) {
        bf16x8 a[8], b[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int kk = min(k0 + j, nk - 1);  // (clamped: unconditional loads; the surplus steps are not multiplied)
            a[j] = *(const bf16x8*)(ap + 16 * kk);
            b[j] = *(const bf16x8*)(bp + 16 * kk);
        }
#pragma unroll
        for (int j = 0; j < 8; ++j)
            if (k0 + j < nk) acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[j], b[j], acc, 0, 0, 0);
    }
    const int n = n0 + r;
    if (n >= H) return;
    const float bv = bias[n];
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        const int m = m0 + (i & 3) + 8 * (i >> 2) + 4 * h;
        if (m < B) pre[(int64_t)m * ldp + n] = acc[i] + bv;
    }
}

// ------------------------------------------------------------------------------------------------
// TopK kernels: one wave per row (wsae_topk.h).
//   topk_kernel        any H, K <= 128: lane maxima -> safe threshold (K <= 64) -> compaction -> sort; exact
//                      bisection when the threshold leaves too many / too few candidates;
//   topk_rows_kernel   K <= 64, H <= 256 * VPL: the row held in registers, read from HBM exactly once;
//   topk_strips_kernel batches served by the persistent GEMM: guided by the strip maxima it leaves.
// ------------------------------------------------------------------------------------------------
#define TOPK_CAP 256

__global__ void __launch_bounds__(256) topk_kernel(const float* __restrict__ pre, int B, int H, int K,
                                                   float* __restrict__ vals, int32_t* __restrict__ idx,
                                                   int32_t* __restrict__ fallback_rows) {
    __shared__ uint64_t lists[4][TOPK_CAP];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int b = blockIdx.x * 4 + wave;
    if (b >= B) return;
    const float* row = pre + (int64_t)b * H;
    uint64_t* list = lists[wave];
    float* vrow = vals + (int64_t)b * K;
    int32_t* irow = idx + (int64_t)b * K;

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
    const int count = topk_compact<TOPK_CAP>(row, H, kmin, list, lane);
    if (count > TOPK_CAP || count < K) {
        topk_row_generic<TOPK_CAP>(row, H, K, list, lane, vrow, irow, fallback_rows);
        return;
    }
    topk_emit_any<TOPK_CAP>(list, count, K, lane, vrow, irow);
}

// lane maxima -> K-th largest lane maximum T (bitonic sort over the 64 lanes) -> every lane files its own
// elements >= T into a private LDS strip (no ballots, no atomics) -> wave prefix sum of the strip lengths ->
// dense list -> bitonic sort -> the K best.  A lane with more than TOPK_STRIP survivors, or more than TOPK_CAP
// in total (ties, adversarial layouts), hands the row to the exact bisection path.
#define TOPK_STRIP 8

template <int VPL>
__global__ void __launch_bounds__(256)
topk_rows_kernel(const float* __restrict__ pre, int B, int H, int K, float* __restrict__ vals,
                 int32_t* __restrict__ idx, int32_t* __restrict__ fallback_rows) {
    __shared__ uint64_t lists[4][TOPK_CAP];
    __shared__ uint64_t strips[4][64 * TOPK_STRIP];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
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
        topk_row_generic<TOPK_CAP>(row, H, K, list, lane, vrow, irow, fallback_rows);
        return;
    }
    const int off = incl - cnt;
    for (int s_ = 0; s_ < cnt; ++s_) list[off + s_] = strip[s_];
    __builtin_amdgcn_wave_barrier();
    topk_emit_any<TOPK_CAP>(list, total, K, lane, vrow, irow);
}

template <int SPL, int NTOP>
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(8, 8)))  // (the refill path may spill; the common path fits)
topk_strips_kernel(const float* pre, const float* __restrict__ smax, int B, int H, int K,
                   float* __restrict__ vals, int32_t* __restrict__ idx, int32_t* __restrict__ fallback_rows, StripFix fix,
                   const uint32_t* tmin) {
    __shared__ uint64_t lists[4][TOPK_CAP];
    __shared__ int strips[4][TS_MAX_STRIPS];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int b = blockIdx.x * 4 + wave;
    if (b >= B) return;
    // (tmin: the encoder GEMM of this batch stored strips selectively and left its threshold there; null: it stored every strip)
    fix.tg = tmin ? *(const float*)(tmin + TG_USED) : -INFINITY;
    topk_strips_row<SPL, NTOP, TOPK_CAP>(pre + (int64_t)b * H, smax + (int64_t)b * (H >> 4), H, K, lane, lists[wave],
                                         strips[wave], vals + (int64_t)b * K, idx + (int64_t)b * K, fallback_rows, nullptr,
                                         nullptr, tmin != nullptr, fix, b);
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

// ------------------------------------------------------------------------------------------------
// host launchers
// ------------------------------------------------------------------------------------------------
static int round_up(int a, int b) { return (a + b - 1) / b * b; }

// step_count (nullable): the dead-feature clock of model.py:175 advances by one in this launch, i.e. BEFORE any
// kernel of the forward reads it (the decode launch stamps last_activated with the advanced value)
template <typename T>
static int stage_batch(wsae_ctx* c, const float* params, const void* x, int x_dtype, const int32_t* rows, int B,
                       int64_t* step_count, hipStream_t st, bool want_xT = true) {
    const int D = c->D;
    const int ldT = round_up(B, 128);
    dim3 sg(ceil_div(ldT, 64), ceil_div(D, 64));
    T* xT = want_xT ? (T*)c->xT : nullptr;
    WSAE_PROF_BEGIN(c, WSAE_K_STAGE_BATCH, st);
    if (!want_xT && sizeof(T) == 2 && x_dtype == WSAE_DT_BF16 && D % 8 == 0) {
        const int64_t nchunks = (int64_t)B * (D / 8);
        stage_rows_copy_kernel<<<(unsigned)ceil_div64(nchunks, 512), 256, 0, st>>>((const uint4*)x, rows, (uint4*)c->xb, nchunks, D / 8,
                                                                                 step_count);
    } else if (x_dtype == WSAE_DT_F32)
        stage_batch_kernel<WSAE_DT_F32, T><<<sg, 256, 0, st>>>(x, rows, params + c->off[4], (T*)c->xb, xT, B, D, ldT, step_count);
    else
        stage_batch_kernel<WSAE_DT_BF16, T><<<sg, 256, 0, st>>>(x, rows, params + c->off[4], (T*)c->xb, xT, B, D, ldT, step_count);
    WSAE_PROF_END(c, WSAE_K_STAGE_BATCH, st);
    WSAE_LAUNCH_CHECK();
    return WSAE_OK;
}

// pre [B][ldp] = xb . W_e^T + bias.  The persistent LDS-DMA kernel takes batches of >= 2048 rows when the
// feature count is a multiple of 256 and K walks in an even number of slabs; it also leaves the strip maxima
// when `pre` is the ctx's own scratch matrix.
// does the persistent LDS-DMA kernel serve a batch of B rows on this ctx?
template <typename T>
static bool persistent_ok(const wsae_ctx* c, int B) {
    constexpr int KT = SWZ_ROW_BYTES / (int)sizeof(T);
    return B >= 2048 && c->H % 256 == 0 && c->D % (2 * KT) == 0;
}

// xa / arows: the A operand - the staged batch ctx->xb (arows null), or the caller's activation buffer with the batch's
// row indices (the ring gather done by the GEMM itself; then step_count, if given, is advanced by this launch)
template <typename T>
static int gemm_dense(wsae_ctx* c, const float* params, int B, float* pre, int ldp, const T* xa, const int32_t* arows,
                      int64_t* step_count, hipStream_t st, bool predicate = false) {
    const T* W = sizeof(T) == 2 ? (const T*)c->We_bf16 : (const T*)(params + c->off[0]);
    const float* bias = sizeof(T) == 2 ? c->c_fold : params + c->off[2];
    const int H = c->H;
    WSAE_PROF_BEGIN(c, WSAE_K_ENCODE_GEMM, st);
    if (persistent_ok<T>(c, B)) {
        const int ntn = H / 256, ntiles = ntn * ceil_div(B, 256);
        float* smax = (pre == c->pre && ldp == H) ? c->smax : nullptr;
        // selective strip stores (wsae_topk.h): only for the strip-guided TopK launch that follows in encode_topk
        uint32_t* tmin = nullptr;
        if (pre == c->pre) c->pred_valid = 0;
        if (predicate && smax && sizeof(T) == 2) {
            c->tmin_cur = (c->tmin_cur + 1) % 3;
            tmin = c->tmin;
            c->pred_valid = 1;
            c->pred_x = xa;
            c->pred_rows = arows;
        }
        encode_gemm256d_kernel<T><<<min(ntiles, c->cus), 512, G256D_LDS, st>>>(xa, c->D, W, c->D, bias, pre, ldp, B, H, c->D, ntn,
                                                                              ntiles, 1, 0, smax, arows, step_count, nullptr,
                                                                              nullptr, tmin, c->tmin_cur, c->tg_fixed_s);
        if (pre == c->pre) c->smax_valid = smax ? 1 : 0;
    } else if (sizeof(T) == 2 && B <= 1024 && c->D % 16 == 0) {
        if (pre == c->pre) c->smax_valid = c->pred_valid = 0;
        encode_direct_kernel<<<dim3(ceil_div(H, 32), ceil_div(B, 128)), 256, 0, st>>>((const bf16_t*)xa, arows, (const bf16_t*)W, bias, pre,
                                                                                     ldp, B, H, c->D, step_count);
    } else {
        if (pre == c->pre) c->smax_valid = c->pred_valid = 0;
        dim3 gg(ceil_div(H, TILE_N), ceil_div(B, TILE_M));
        encode_gemm_kernel<T><<<gg, 256, 2 * TILE_LDS_BYTES, st>>>(xa, W, bias, pre, ldp, B, H, c->D, arows, step_count);
    }
    WSAE_PROF_END(c, WSAE_K_ENCODE_GEMM, st);
    WSAE_LAUNCH_CHECK();
    return WSAE_OK;
}

// direct != 0 (the encoder forward of a bf16 batch in BF16 mode on the persistent kernel): no staging launch - the
// GEMM gathers the batch rows from the caller's buffer itself and wsae_weight_grads transposes x in its own bucket
// launch (ctx->xT_valid = 0 tells it so).  Everything else stages xb / xT first.
template <typename T>
static int stage_and_gemm(wsae_ctx* c, const float* params, const void* x, int x_dtype, const int32_t* rows, int B,
                          float* pre, int64_t* step_count, int direct, hipStream_t st, bool predicate = false) {
    // (the persistent kernel for large batches; below its minimum the 128 x 128 kernel takes the row list the same way: at the
    // reference's own batch sizes the staging launch was 5 of the step's ~80 us of kernels)
    if (direct && sizeof(T) == 2 && x_dtype == WSAE_DT_BF16) {
        c->xT_valid = 0;
        return gemm_dense<T>(c, params, B, pre, c->H, (const T*)x, rows, step_count, st, predicate);
    }
    int rc = stage_batch<T>(c, params, x, x_dtype, rows, B, step_count, st);
    if (rc) return rc;
    c->xT_valid = 1;
    return gemm_dense<T>(c, params, B, pre, c->H, (const T*)c->xb, nullptr, nullptr, st, predicate);
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
    const int ntn = ceil_div(N, 256), ntiles_mn = ntn * ceil_div(M, 256);
    encode_gemm256d_kernel<T><<<min(ntiles_mn * nsplit, c->cus), 512, G256D_LDS, st>>>((const T*)A, lda, (const T*)Bt, ldb, bias, C, ldc,
                                                                                      M, N, K / nsplit, ntn, ntiles_mn, nsplit, cz,
                                                                                      nullptr, nullptr, nullptr);
    return true;
}

// fp8 e4m3 operands with a dequantisation scale per row of A and per row of Bt (the ReLU SAE's fp8 forward)
bool wsae_internal_gemm256d_fp8(wsae_ctx* c, const void* A, int64_t lda, const void* Bt, int64_t ldb, const float* rscale,
                                const float* cscale, const float* bias, float* C, int64_t ldc, int M, int N, int K,
                                hipStream_t st) {
    constexpr int KT = SWZ_ROW_BYTES;  // 128 one-byte elements per slab
    if (M < 512 || N < 128 || N % 4 || K % (2 * KT) || lda % 16 || ldb % 16 || ldc % 4 || !rscale || !cscale) return false;
    const int ntn = ceil_div(N, 256), ntiles_mn = ntn * ceil_div(M, 256);
    encode_gemm256d_kernel<fp8_t><<<min(ntiles_mn, c->cus), 512, G256D_LDS, st>>>((const fp8_t*)A, lda, (const fp8_t*)Bt, ldb, bias, C, ldc,
                                                                                 M, N, K, ntn, ntiles_mn, 1, 0, nullptr, nullptr,
                                                                                 nullptr, rscale, cscale);
    return true;
}

bool wsae_internal_gemm256d(wsae_ctx* c, const void* A, int64_t lda, const void* Bt, int64_t ldb, const float* bias, float* C,
                            int64_t ldc, int M, int N, int K, int nsplit, int64_t cz, hipStream_t st) {
    return c->prec == WSAE_PREC_BF16 ? gemm256d_try<bf16_t>(c, A, lda, Bt, ldb, bias, C, ldc, M, N, K, nsplit, cz, st)
                                     : gemm256d_try<float>(c, A, lda, Bt, ldb, bias, C, ldc, M, N, K, nsplit, cz, st);
}

// staging alone (xb, xT): for callers that run their own encoder GEMM (the ReLU SAE's fp8 forward)
int wsae_internal_stage(wsae_ctx* ctx, const float* params, const void* x, int x_dtype, const int32_t* rows, int B, hipStream_t st) {
    const int rc = ctx->prec == WSAE_PREC_BF16 ? stage_batch<bf16_t>(ctx, params, x, x_dtype, rows, B, nullptr, st)
                                                : stage_batch<float>(ctx, params, x, x_dtype, rows, B, nullptr, st);
    if (rc == WSAE_OK) ctx->xT_valid = 1;
    return rc;
}

// xb alone (no transposed copy): for callers that read the staged rows row-major (wsae_relu.hip, row-major-GEMM flow)
int wsae_internal_stage_rows(wsae_ctx* ctx, const float* params, const void* x, int x_dtype, const int32_t* rows, int B, hipStream_t st) {
    ctx->xT_valid = 0;
    return ctx->prec == WSAE_PREC_BF16 ? stage_batch<bf16_t>(ctx, params, x, x_dtype, rows, B, nullptr, st, false)
                                       : stage_batch<float>(ctx, params, x, x_dtype, rows, B, nullptr, st, false);
}

int wsae_internal_stage_and_gemm(wsae_ctx* ctx, const float* params, const void* x, int x_dtype, const int32_t* rows,
                                 int B, float* pre, int64_t* step_count, int direct, hipStream_t st, bool predicate) {
    return ctx->prec == WSAE_PREC_BF16 ? stage_and_gemm<bf16_t>(ctx, params, x, x_dtype, rows, B, pre, step_count, direct, st, predicate)
                                       : stage_and_gemm<float>(ctx, params, x, x_dtype, rows, B, pre, step_count, direct, st, false);
}

extern "C" int wsae_encode_dense(wsae_ctx* ctx, const float* params, const void* x, int32_t x_dtype,
                                 const int32_t* rows, int32_t B, float* pre, void* stream) {
    int rc = check_batch(ctx, x, x_dtype, B, "wsae_encode_dense");
    if (rc) return rc;
    WSAE_REQUIRE(params && pre, "wsae_encode_dense: null argument");
    return wsae_internal_stage_and_gemm(ctx, params, x, x_dtype, rows, B, pre, nullptr, 0, (hipStream_t)stream);
}

// strip-guided TopK is possible when the persistent GEMM just left the strip maxima of ctx->pre
bool wsae_internal_strips_ok(const wsae_ctx* ctx) {
    const int ns = ctx->H / 16;
    return ctx->smax_valid && ctx->K <= 64 && ctx->H % 16 == 0 && ns >= 128 && ns <= 1024;
}

int wsae_internal_topk(wsae_ctx* ctx, int B, float* vals, int32_t* idx, int32_t* fb, hipStream_t st) {
    WSAE_PROF_BEGIN(ctx, WSAE_K_TOPK, st);
    const int vpl = ceil_div(ctx->H, 256);
    const int ns = ctx->H / 16;
    const dim3 tg(ceil_div(B, 4));
    if (wsae_internal_strips_ok(ctx)) {
        // the GEMM stored this batch's strips selectively: this launch checks every row against the same threshold,
        // fills in what a row is missing and leaves its own minimum for the next GEMM (wsae_topk.h)
        StripFix fix = {};
        const uint32_t* tmin = nullptr;
        if (ctx->pred_valid) {
            fix.x = (const bf16_t*)ctx->pred_x;
            fix.arows = ctx->pred_rows;
            fix.We = ctx->We_bf16;
            fix.bias = ctx->c_fold;
            fix.D = ctx->D;
            fix.tmin_group = ctx->tmin + ctx->tmin_cur * TG_GROUP_WORDS;
            fix.miss_rows = (int32_t*)(ctx->tmin + TG_REFILLED);
            tmin = ctx->tmin;
            ctx->pred_valid = 0;  // (pre is consumed: a second TopK over it would find the slots moved on)
        }
#define TS_ARGS ctx->pre, ctx->smax, B, ctx->H, ctx->K, vals, idx, fb, fix, tmin
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
        topk_rows_kernel<4><<<tg, 256, 0, st>>>(ctx->pre, B, ctx->H, ctx->K, vals, idx, fb);
    else if (ctx->K <= 64 && vpl <= 12)
        topk_rows_kernel<12><<<tg, 256, 0, st>>>(ctx->pre, B, ctx->H, ctx->K, vals, idx, fb);
    else if (ctx->K <= 64 && vpl <= 16)
        topk_rows_kernel<16><<<tg, 256, 0, st>>>(ctx->pre, B, ctx->H, ctx->K, vals, idx, fb);
    else
        topk_kernel<<<tg, 256, 0, st>>>(ctx->pre, B, ctx->H, ctx->K, vals, idx, fb);
    WSAE_PROF_END(ctx, WSAE_K_TOPK, st);
    WSAE_LAUNCH_CHECK();
    return WSAE_OK;
}

// encoder + TopK of one batch: the dense GEMM into ctx->pre followed by a TopK launch
int wsae_internal_encode_topk(wsae_ctx* ctx, const float* params, const void* x, int x_dtype, const int32_t* rows, int B,
                              float* vals, int32_t* idx, int64_t* step_count, int32_t* fb, hipStream_t st) {
    // (selective strip stores need the strip-guided TopK right behind the GEMM: the conditions of wsae_internal_strips_ok
    // that do not depend on the launch)
    const int ns = ctx->H / 16;
    const bool predicate = ctx->strip_predict && ctx->K <= 64 && ctx->H % 16 == 0 && ns >= 128 && ns <= 1024;
    int rc = wsae_internal_stage_and_gemm(ctx, params, x, x_dtype, rows, B, ctx->pre, step_count, 1, st, predicate);
    if (rc) return rc;
    WSAE_REQUIRE(!ctx->pred_valid || wsae_internal_strips_ok(ctx), "encode_topk: selective strip stores without the strip-guided TopK");
    return wsae_internal_topk(ctx, B, vals, idx, fb, st);
}

extern "C" int wsae_encode_topk(wsae_ctx* ctx, const float* params, const void* x, int32_t x_dtype,
                                const int32_t* rows, int32_t B, float* vals, int32_t* idx, int64_t* step_count,
                                wsae_stats* stats, void* stream) {
    int rc = check_batch(ctx, x, x_dtype, B, "wsae_encode_topk");
    if (rc) return rc;
    WSAE_REQUIRE(params && vals && idx, "wsae_encode_topk: null argument");
    hipStream_t st = (hipStream_t)stream;
    int32_t* fb = stats ? &stats->topk_fallback_rows : ctx->counters;
    return wsae_internal_encode_topk(ctx, params, x, x_dtype, rows, B, vals, idx, step_count, fb, st);
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
