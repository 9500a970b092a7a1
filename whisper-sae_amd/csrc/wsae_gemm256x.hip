// Persistent 256 x 256 bf16 GEMM with operand-layout and epilogue variants (the ReLU SAE's dense path, wsae_relu.hip;
// reference ReLUSAE, model.py:260-322).
//
//   C[M][N] (+ z * cz) = A . B over K range z, fp32 accumulate, where each operand is given either
//     K-contiguous ("NT": A[m][k], Bt[n][k] - one ds_read_b128 per MFMA fragment), or
//     ROW-MAJOR IN K ("RM": Ak[k][m], Bk[k][n] - the layout activations and their gradients already have in memory),
//       staged as [column block of 64][64 k rows][128 bytes] by LDS-DMA and read with ds_read_b64_tr_b16 (two per
//       fragment; conflict-free under rm_swz) - so no kernel has to write a transposed copy of a [B, H] matrix.
//   The five GEMMs of a ReLU-SAE step then read x, hidden, g, dpre and the decoder shadow exactly as they lie:
//     hidden = relu(x W_e^T + b_e)      NT/NT, epilogue RELU : bf16 hidden (+ the fp32 API copy on request), L1 / l0 partials
//     recon  = hidden W_d^T             NT/RM (W_dT [H][D] is W_d row-major in K = h), split-K 2 into two slabs
//     dh     = g W_d -> dpre            NT/NT, epilogue DPRE : dpre = (dh + l1 w) * [hidden > 0] as bf16, db_e column partials
//     dW_e   = dpre^T x, dW_dT = hidden^T g      RM/RM, split-K over the batch into the slabs
//   Same skeleton as encode_gemm256d_kernel (wsae_encode.hip): one workgroup of 8 waves per CU walks its tiles, two 64 KB
//   LDS stages filled by LDS-DMA a K slab ahead (the next tile's first slab lands while the current tile is stored), wave
//   tile 128 x 64 = 4 x 2 MFMA tiles of 32x32x16, the epilogue goes through per-wave LDS patches so that every lane owns 4
//   consecutive columns of a row.  bf16 only.
#include "wsae_common.h"
#include "wsae_mfma.h"

#define GX_STAGE (2 * 256 * SWZ_ROW_BYTES)  // A tile + B tile = 64 KB
#define GX_STAGE1 (72 * 1024)
#define GX_LDS (GX_STAGE1 + GX_STAGE)       // 136 KB

__device__ __forceinline__ int gx_rm_swz(int r) { return (((r >> 1) & 1) << 2) | ((r >> 2) & 3); }

typedef __attribute__((address_space(3))) bf16x4 gx_lds_bf16x4;

// FP8 (NT/NT only): one-byte OCP e4m3 operands - a 128-byte image row holds 128 K elements - multiplied by
// v_mfma_scale_f32_32x32x64_f8f6f4 with unit block scales (E8M0 0x7F): twice the bf16 form's flops per cycle; the per-row
// dequantisation scales of both operands are applied to the accumulator in the epilogue.  Lane l (row l & 31, half l >> 5)
// supplies the 32 bytes k = 32 (l >> 5) .. + 31 of its row for each 64-deep step (profiles/tools/probe_mxfp8.hip: with unit
// scales any assignment of a half's 32 k to the lane's bytes is equivalent as long as both operands use the same one).
template <bool ARM, bool BRM, int EPI, bool FP8 = false>
__global__ void __launch_bounds__(512)
gemm256x_kernel(const bf16_t* __restrict__ A, int64_t lda, const bf16_t* __restrict__ Bm, int64_t ldb, int M, int N, int Kr,
                int ntn, int ntiles_mn, int nsplit, GxEpi e) {
    static_assert(!FP8 || (!ARM && !BRM), "fp8 operands are K-contiguous");
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int ESZ = FP8 ? 1 : 2;                 // bytes per operand element (A / Bm / lda / ldb / Kr count ELEMENTS)
    constexpr int KT = 128 / ESZ, EPC = 16 / ESZ;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 2, wn = wave & 3;
    const int nk = Kr / KT;
    const int col = lane & 31, rq = lane >> 5;
    constexpr int PS = 68;
    float* patch = (float*)smem + wave * 32 * PS;  // inside stage 0's region
    const int pr = lane >> 4, pc = (lane & 15) * 4;
    const uint32_t smem_lds = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) char*)smem;
    const int dma_r = lane >> 3, dma_s = lane & 7;
    const int ntiles = ntiles_mn * nsplit;

    // K slab k0 of tile (m0, n0) into stage st: pieces 0..31 = the A operand, 32..63 = the B operand.  Source address of a
    // piece = a wave-uniform base (operand + the tile's first row / the slab's first k: an SGPR pair) + a 32-bit per-lane byte
    // offset inside the tile (row clamps included), so eight pieces cost eight VGPRs, not eight address pairs.
    auto dma = [&](int m0, int n0, int64_t k0, int st) {
        const uint32_t base = smem_lds + (st ? GX_STAGE1 : 0);
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int piece = wave + 8 * j;  // 0..63
            const int pi = piece & 31;
            const bool isA = j < 4;
            const char* op = (const char*)(isA ? A : Bm);
            const int64_t ld = isA ? lda : ldb;
            const int t0 = isA ? m0 : n0, T = isA ? M : N;
            const char* sb;
            uint32_t voff;
            if (isA ? ARM : BRM) {
                const int krow = (pi & 7) * 8 + dma_r;                              // k row inside the slab
                const int cbk = (pi >> 3) * 64 + (dma_s ^ gx_rm_swz(krow)) * EPC;   // column inside the 256-wide tile
                sb = op + k0 * ld * ESZ;                                            // the slab's first k row
                voff = (uint32_t)(((int64_t)krow * ld + min(t0 + cbk, T - EPC)) * ESZ);
            } else {
                const int row = pi * 8 + dma_r;                                     // operand row inside the 256-row tile
                const int c = dma_s ^ ((row >> 1) & 7);
                sb = op + ((int64_t)t0 * ld + k0) * ESZ;                            // the tile's first row at the slab's first k
                voff = (uint32_t)(((int64_t)(min(t0 + row, T - 1) - t0) * ld + c * EPC) * ESZ);
            }
            glds16_s(sb, voff, base + piece * 1024);
        }
    };
    auto m_of = [&](int t) { return ((t % ntiles_mn) / ntn) * 256; };
    auto n_of = [&](int t) { return ((t % ntiles_mn) % ntn) * 256; };
    auto k_of = [&](int t) { return (int64_t)(t / ntiles_mn) * Kr; };

    // lane-constant fragment addresses (offsets inside an operand's 32 KB image)
    const int r31 = lane & 31, h = lane >> 5;
    const int sw = (r31 >> 1) & 7;
    int a_nt = (wm * 128 + r31) * SWZ_ROW_BYTES, b_nt = (wn * 64 + r31) * SWZ_ROW_BYTES;
    int a_rm[4][2], b_rm[2][2];
    {
        const int l15 = lane & 15, g1 = (lane >> 4) & 1;
        auto rm_addr = [&](int col0, int q2) {
            const int c0 = col0 + g1 * 16 + 4 * (l15 & 3);
            const int row = 8 * h + 4 * q2 + (l15 >> 2);  // + 16 kk: does not change the swizzle
            return (c0 >> 6) * 8192 + row * 128 + ((((c0 & 63) >> 3) ^ gx_rm_swz(row)) << 4) + 8 * (l15 & 1);
        };
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int q2 = 0; q2 < 2; ++q2) a_rm[i][q2] = rm_addr(wm * 128 + 32 * i, q2);
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int q2 = 0; q2 < 2; ++q2) b_rm[i][q2] = rm_addr(wn * 64 + 32 * i, q2);
    }
    auto rm_frag = [&](const char* img, const int (&ad)[2], int kk) -> bf16x8 {
        const bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((gx_lds_bf16x4*)(img + ad[0] + kk * 2048));
        const bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((gx_lds_bf16x4*)(img + ad[1] + kk * 2048));
        return __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
    };

    int it = 0, st = 1;
    int tile = (int)blockIdx.x;
    if (tile < ntiles) dma(m_of(tile), n_of(tile), k_of(tile), st);
    for (; tile < ntiles; tile = (int)blockIdx.x + (++it) * (int)gridDim.x) {
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
        const int nt = min(tile + (int)gridDim.x, ntiles - 1);
        for (int kt = 0; kt < nk; ++kt) {
            dma_wait();       // slab kt (issued one step ago) has landed
            __syncthreads();  // ... for every wave; and everybody is done reading the other stage
            if (kt + 1 < nk) dma(m0, n0, kbase + (int64_t)(kt + 1) * KT, st ^ 1);
            else dma(m_of(nt), n_of(nt), k_of(nt), st ^ 1);
            const char* As = smem + (st ? GX_STAGE1 : 0);
            const char* Bs = As + 256 * SWZ_ROW_BYTES;
            if constexpr (FP8) {
                typedef int v8i __attribute__((ext_vector_type(8)));
                typedef int v4i __attribute__((ext_vector_type(4)));
#pragma unroll
                for (int s2 = 0; s2 < 2; ++s2) {  // two 64-deep steps per 128-byte row
                    const int c0 = 4 * s2 + 2 * h;
                    const int o0 = (c0 ^ sw) << 4, o1 = ((c0 + 1) ^ sw) << 4;
                    v8i b[2];
#pragma unroll
                    for (int i = 0; i < 2; ++i) {
                        const v4i lo = *(const v4i*)(Bs + b_nt + i * 32 * SWZ_ROW_BYTES + o0);
                        const v4i hi = *(const v4i*)(Bs + b_nt + i * 32 * SWZ_ROW_BYTES + o1);
                        b[i] = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
                    }
#pragma unroll
                    for (int mh = 0; mh < 2; ++mh) {  // (the A fragments two at a time: 8 registers each)
                        v8i a[2];
#pragma unroll
                        for (int i = 0; i < 2; ++i) {
                            const v4i lo = *(const v4i*)(As + a_nt + (2 * mh + i) * 32 * SWZ_ROW_BYTES + o0);
                            const v4i hi = *(const v4i*)(As + a_nt + (2 * mh + i) * 32 * SWZ_ROW_BYTES + o1);
                            a[i] = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
                        }
#pragma unroll
                        for (int i = 0; i < 2; ++i)
#pragma unroll
                            for (int ni = 0; ni < 2; ++ni)
                                acc[2 * mh + i][ni] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a[i], b[ni], acc[2 * mh + i][ni], 0, 0, 0,
                                                                                                      0x7F7F7F7F, 0, 0x7F7F7F7F);
                    }
                }
            } else
#pragma unroll
            for (int kk = 0; kk < 4; ++kk) {
                const int co = ((kk * 2 + h) ^ sw) << 4;
                bf16x8 a[4], b[2];
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    if constexpr (ARM) a[i] = rm_frag(As, a_rm[i], kk);
                    else a[i] = *(const bf16x8*)(As + a_nt + i * 32 * SWZ_ROW_BYTES + co);
                }
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    if constexpr (BRM) b[i] = rm_frag(Bs, b_rm[i], kk);
                    else b[i] = *(const bf16x8*)(Bs + b_nt + i * 32 * SWZ_ROW_BYTES + co);
                }
#pragma unroll
                for (int mi = 0; mi < 4; ++mi)
#pragma unroll
                    for (int ni = 0; ni < 2; ++ni)
                        acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[mi], b[ni], acc[mi][ni], 0, 0, 0);
            }
            st ^= 1;
        }
        // the slab in flight targets stage 1 (st == 1 again: nk is even); the patches take stage 0's place
        __syncthreads();
        const int ncol = n0 + wn * 64 + pc;  // this lane's 4 columns
        const bool col_ok = ncol < N;        // (N is a multiple of 4)
        // (the per-column constants - bias, L1 weights, fp8 column scales - are (re)loaded inside the row-group loop below, once per
        // 32 rows: held across the whole epilogue they sat on top of the 128 live accumulator registers and pushed the
        // kernel into scratch, which cost the two epilogue variants 11 us per launch)
        float s_l1 = 0.f, s_cnt = 0.f;                 // RELU partials
        float4 csum = make_float4(0.f, 0.f, 0.f, 0.f);  // DPRE column sums (this lane's rows)
        const int64_t widx = (n0 + wn * 64) >> 6;        // this wave's word of a row's activity bits
        // DPRE: the activity bits of the wave tile's 128 rows, two rows per lane, requested before the patches are touched
        uint64_t mw0 = 0, mw1 = 0;
        if constexpr (EPI == GX_EPI_DPRE) {
            const int ra = min(m0 + wm * 128 + lane, M - 1), rb = min(m0 + wm * 128 + 64 + lane, M - 1);
            if (n0 + wn * 64 < N) {  // (column-major words: a wave's 128 rows of one word column are 1 KB contiguous)
                mw0 = e.bits[widx * e.ldbits + ra];
                mw1 = e.bits[widx * e.ldbits + rb];
            }
        }
#pragma unroll
        for (int mi = 0; mi < 4; ++mi) {
#pragma unroll
            for (int ni = 0; ni < 2; ++ni)
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    patch[((r & 3) + 8 * (r >> 2) + 4 * rq) * PS + ni * 32 + col] = acc[mi][ni][r];
            __builtin_amdgcn_wave_barrier();
            unsigned long long keep = 0;  // RELU: lane (pr, j = i) keeps the activity word of row pr + 4 i; ONE store per 32 rows
            float keepmax = 0.f;
            float4 bv4 = make_float4(0.f, 0.f, 0.f, 0.f), w4 = make_float4(1.f, 1.f, 1.f, 1.f), cs4 = make_float4(1.f, 1.f, 1.f, 1.f);
            if (EPI != GX_EPI_DPRE && e.bias && z == 0 && col_ok) bv4 = *(const float4*)(e.bias + ncol);
            if (EPI != GX_EPI_PLAIN && e.colw && col_ok) w4 = *(const float4*)(e.colw + ncol);
            if constexpr (FP8) {
                if (col_ok) cs4 = *(const float4*)(e.cscale + ncol);  // fp8 operands: column (Bt row) dequantisation scales
            }
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const int rl = pr + 4 * i;
                const int m = m0 + wm * 128 + mi * 32 + rl;
                float4 v = *(const float4*)(patch + rl * PS + pc);
                const bool ok = m < M && col_ok;
                if constexpr (FP8) {
                    const float rs = e.rscale[min(m, M - 1)];
                    v.x *= rs * cs4.x; v.y *= rs * cs4.y; v.z *= rs * cs4.z; v.w *= rs * cs4.w;
                }
                if constexpr (EPI == GX_EPI_PLAIN) {
                    v.x += bv4.x; v.y += bv4.y; v.z += bv4.z; v.w += bv4.w;
                    if (ok) *(float4*)(e.c + (int64_t)z * e.cz + (int64_t)m * e.ldc + ncol) = v;
                } else if constexpr (EPI == GX_EPI_RELU) {
                    v.x = fmaxf(v.x + bv4.x, 0.f); v.y = fmaxf(v.y + bv4.y, 0.f);
                    v.z = fmaxf(v.z + bv4.z, 0.f); v.w = fmaxf(v.w + bv4.w, 0.f);
                    if (ok) {
                        bf16x4 o;
                        o[0] = (bf16_t)v.x; o[1] = (bf16_t)v.y; o[2] = (bf16_t)v.z; o[3] = (bf16_t)v.w;
                        __builtin_nontemporal_store(o, (bf16x4*)(e.out16 + (int64_t)m * e.ld16 + ncol));
                        if (e.c) *(float4*)(e.c + (int64_t)m * e.ldc + ncol) = v;
                        s_l1 += (v.x * w4.x + v.y * w4.y) + (v.z * w4.z + v.w * w4.w);
                        s_cnt += (float)((v.x > 0.f) + (v.y > 0.f) + (v.z > 0.f) + (v.w > 0.f));
                    }
                    // activity bits of this instruction's 4 rows (every lane takes part in the ballots; rows / columns past the
                    // matrix vote 0); the first lane of a row's 16 writes the row's word.  (bf16 keeps fp32's exponent range: a
                    // positive hidden value stays positive when it is rounded, so these are also the bits of bf16(hidden) > 0.)
                    {
                        // (32-bit field extracts: the 64-bit vector shifts of the obvious form cost this epilogue 14 us per launch)
                        const unsigned long long bx = __ballot(ok && v.x > 0.f), by = __ballot(ok && v.y > 0.f);
                        const unsigned long long bz = __ballot(ok && v.z > 0.f), bw = __ballot(ok && v.w > 0.f);
                        const bool uh = pr >= 2;         // rows 2, 3 of the instruction sit in the ballots' upper words
                        const int sh = 16 * (pr & 1);
                        const uint32_t fx = __builtin_amdgcn_ubfe(uh ? (uint32_t)(bx >> 32) : (uint32_t)bx, sh, 16);
                        const uint32_t fy = __builtin_amdgcn_ubfe(uh ? (uint32_t)(by >> 32) : (uint32_t)by, sh, 16);
                        const uint32_t fz = __builtin_amdgcn_ubfe(uh ? (uint32_t)(bz >> 32) : (uint32_t)bz, sh, 16);
                        const uint32_t fw = __builtin_amdgcn_ubfe(uh ? (uint32_t)(bw >> 32) : (uint32_t)bw, sh, 16);
                        const unsigned long long word = (unsigned long long)(fx | (fy << 16)) | ((unsigned long long)(fz | (fw << 16)) << 32);
                        if ((lane & 15) == i) keep = word;
                    }
                    if constexpr (FP8) {  // maximum of this row's 64-column span (relu >= 0): the 16 lanes of the row meet by lane exchanges
                        // (of the bf16-ROUNDED values: what the quantisation pass and the oracle's "fp8" mode take the maximum of)
                        float mx = fmaxf(fmaxf((float)(bf16_t)v.x, (float)(bf16_t)v.y), fmaxf((float)(bf16_t)v.z, (float)(bf16_t)v.w));
                        mx = ok ? mx : 0.f;
                        mx = fmaxf(mx, __uint_as_float(lane_xor_u32<1>(__float_as_uint(mx), lane)));
                        mx = fmaxf(mx, __uint_as_float(lane_xor_u32<2>(__float_as_uint(mx), lane)));
                        mx = fmaxf(mx, __uint_as_float(lane_xor_u32<4>(__float_as_uint(mx), lane)));
                        mx = fmaxf(mx, __uint_as_float(lane_xor_u32<8>(__float_as_uint(mx), lane)));
                        if ((lane & 15) == i) keepmax = mx;
                    }
                } else {
                    float4 d = make_float4(0.f, 0.f, 0.f, 0.f);
                    // this row's word sits in lane (32 mi + 4 i + pr) & 63 of mw0 (rows 0..63 of the wave tile) or mw1
                    const int srcl = (32 * mi + 4 * i + pr) & 63;
                    const unsigned long long mw = mi < 2 ? mw0 : mw1;
                    const uint32_t lo = (uint32_t)__shfl((int)(uint32_t)mw, srcl, 64), hi = (uint32_t)__shfl((int)(uint32_t)(mw >> 32), srcl, 64);
                    const int j = lane & 15;
                    if (ok) {
                        d.x = ((lo >> j) & 1u) ? v.x + e.l1 * w4.x : 0.f;
                        d.y = ((lo >> (16 + j)) & 1u) ? v.y + e.l1 * w4.y : 0.f;
                        d.z = ((hi >> j) & 1u) ? v.z + e.l1 * w4.z : 0.f;
                        d.w = ((hi >> (16 + j)) & 1u) ? v.w + e.l1 * w4.w : 0.f;
                        bf16x4 o;
                        o[0] = (bf16_t)d.x; o[1] = (bf16_t)d.y; o[2] = (bf16_t)d.z; o[3] = (bf16_t)d.w;
                        __builtin_nontemporal_store(o, (bf16x4*)(e.out16 + (int64_t)m * e.ld16 + ncol));  // (read once, by the contraction behind)
                    }
                    csum.x += d.x; csum.y += d.y; csum.z += d.z; csum.w += d.w;
                }
            }
            if constexpr (EPI == GX_EPI_RELU) {
                // rows pr + 4 j (j = lane & 15 < 8) of this 32-row group: 32 consecutive rows of one word column = 256 contiguous bytes
                const int j = lane & 15;
                const int m = m0 + wm * 128 + mi * 32 + pr + 4 * j;
                if (j < 8 && m < M && n0 + wn * 64 < N) {
                    e.bits[widx * e.ldbits + m] = keep;
                    if constexpr (FP8) e.rowmax[widx * e.ldbits + m] = keepmax;
                }
            }
            __builtin_amdgcn_wave_barrier();
        }
        if constexpr (EPI == GX_EPI_RELU) {
            // tile partials in fixed order: wave sums -> LDS -> thread 0 (the patches are free: every wave passed its last wave barrier
            // only for ITS OWN patch, so the scratch below lives behind the patches)
            float* red = (float*)smem + 8 * 32 * PS;  // 16 floats behind the eight patches (inside stage 0: 69.6 KB < 72 KB)
            const float a = wave_sum(s_l1), c = wave_sum(s_cnt);
            if (lane == 0) { red[wave] = a; red[8 + wave] = c; }
            __syncthreads();
            if (tid == 0) {
                float ta = 0.f, tc = 0.f;
#pragma unroll
                for (int w = 0; w < 8; ++w) { ta += red[w]; tc += red[8 + w]; }
                e.part[tile] = ta;
                e.part[e.nslots + tile] = tc;
            }
        }
        if constexpr (EPI == GX_EPI_DPRE) {
            // lanes pc, pc + 16, pc + 32, pc + 48 hold the four row classes of the same 4 columns
#pragma unroll
            for (int o = 16; o <= 32; o <<= 1) {
                csum.x += __shfl_xor(csum.x, o, 64); csum.y += __shfl_xor(csum.y, o, 64);
                csum.z += __shfl_xor(csum.z, o, 64); csum.w += __shfl_xor(csum.w, o, 64);
            }
            if (lane < 16 && col_ok && m0 + wm * 128 < M) *(float4*)(e.colpart + (int64_t)(m0 / 128 + wm) * N + ncol) = csum;
        }
        // (the loop top's barrier separates these patch reads from the next slab landing in stage 0)
    }
    dma_wait();
}

// host side --------------------------------------------------------------------------------------------------------------
// Shape rules: M, N multiples of 8 (N of 4 for fp32 C), Kr = K / nsplit a multiple of 128 (an even number of 64-deep slabs),
// leading dimensions multiples of 8 elements.  Returns false when the shape does not qualify (the caller keeps its old path).
bool wsae_internal_gemm256x(wsae_ctx* c, int a_rm, int b_rm, int epi, const void* A, int64_t lda, const void* Bm, int64_t ldb,
                            int M, int N, int K, int nsplit, const GxEpi& e, hipStream_t st, int fp8) {
    // (an even number of K slabs per range: 64 elements per slab in bf16, 128 in fp8; 16-byte aligned rows)
    const int slab2 = fp8 ? 256 : 128, al = fp8 ? 16 : 8;
    if (M < 256 || N < 128 || M % 8 || N % 8 || nsplit < 1 || K % nsplit || (K / nsplit) % slab2 || lda % al || ldb % al) return false;
    if (fp8 && (a_rm || b_rm || epi == GX_EPI_DPRE || !e.rscale || !e.cscale || (epi == GX_EPI_RELU && !e.rowmax))) return false;
    if (epi != GX_EPI_PLAIN && nsplit != 1) return false;
    const int ntn = ceil_div(N, 256), ntiles_mn = ntn * ceil_div(M, 256);
    if (epi == GX_EPI_RELU && ntiles_mn > e.nslots) return false;
    const int grid = min(ntiles_mn * nsplit, c->cus);
    const bf16_t* a = (const bf16_t*)A;
    const bf16_t* b = (const bf16_t*)Bm;
#define GX_LAUNCH(AR, BR, EP) gemm256x_kernel<AR, BR, EP><<<grid, 512, GX_LDS, st>>>(a, lda, b, ldb, M, N, K / nsplit, ntn, ntiles_mn, nsplit, e)
    if (fp8) {
        if (epi == GX_EPI_RELU) gemm256x_kernel<false, false, GX_EPI_RELU, true><<<grid, 512, GX_LDS, st>>>(a, lda, b, ldb, M, N, K / nsplit, ntn, ntiles_mn, nsplit, e);
        else gemm256x_kernel<false, false, GX_EPI_PLAIN, true><<<grid, 512, GX_LDS, st>>>(a, lda, b, ldb, M, N, K / nsplit, ntn, ntiles_mn, nsplit, e);
        return true;
    }
    if (!a_rm && !b_rm && epi == GX_EPI_RELU) GX_LAUNCH(false, false, GX_EPI_RELU);
    else if (!a_rm && !b_rm && epi == GX_EPI_DPRE) GX_LAUNCH(false, false, GX_EPI_DPRE);
    else if (!a_rm && b_rm && epi == GX_EPI_PLAIN) GX_LAUNCH(false, true, GX_EPI_PLAIN);
    else if (a_rm && b_rm && epi == GX_EPI_PLAIN) GX_LAUNCH(true, true, GX_EPI_PLAIN);
    else return false;
#undef GX_LAUNCH
    return true;
}
