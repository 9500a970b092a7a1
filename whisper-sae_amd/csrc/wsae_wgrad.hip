// Weight gradients (second half of backward).
//   reference: autograd of nn.Linear encoder/decoder, model.py:111,129:
//     dW_d = g^T hidden          -> here dW_dT[h,:] = sum_b hidden[b,h] * g[b,:]
//     dW_e = dpre^T (x - b_pre)  -> here dW_e[h,:]  = sum_b dpre[b,h]   * x_c[b,:]
//     db_e = sum_b dpre ; db_d = sum_b g ; db_pre = db_d - W_e^T db_e
//
// Both contractions are [H x B] x [B x D] with a left operand that is 99% zeros and only exists
// as the compact TopK code (vals, idx)[B,k].  A scatter-add of B*k rows would be atomic-bound
// (2 x 805 MB of float atomics at cfg2/B=16384 against ~1.3 TB/s); instead each workgroup rebuilds
// its 128-feature x KT-row slice of the left operand in LDS from the compact code (zero fill +
// scatter of the entries whose feature falls in the tile) and runs it through the same NT MFMA
// slab code as the encode GEMM.  The right operands are the transposed batch operands xT / gT
// left in the ctx by the encode and decode launches.
//
// Split-K over the batch: workgroup id = tile * nsplit + split, so with nsplit = 8 the round-robin
// workgroup->XCD dealing puts every tile of one batch range on one XCD, whose 4 MB L2 then holds
// that range's xT/gT columns and compact code (2048 rows at cfg2 = 3.6 MB): each XCD streams its
// share of the batch from HBM once and the 144 tiles re-read it from L2.  Every split writes a
// private fp32 slab; a second kernel sums the slabs in fixed order (deterministic, no float atomics).
#include "wsae_common.h"
#include "wsae_mfma.h"

// ------------------------------------------------------------------------------------------------
// bucket_kernel, blocks [0, nchunks): per KT-row chunk, counting-sort the chunk's KT*K compact entries by tw-feature tile
// (tw = 128 for wgrad_kernel, 192 for wgrad2_kernel).
//   ent_off[chunk][t] .. ent_off[chunk][t+1] : positions (in the flat sorted arrays) of the entries of
//   tile t;  ent_pos = (feature - t * tw) << 16 | row-in-chunk;  ent_hid = relu(value), ent_dpre = dpre.
// A (tile, chunk) workgroup of the contraction then touches only its own ~KT*K*128/H entries
// instead of scanning all KT*K of them for a 128/H hit rate.
// ------------------------------------------------------------------------------------------------
#define BUCKET_MAX_TILES 512

template <typename T>
__global__ void __launch_bounds__(256)
bucket_kernel(const float* __restrict__ vals, const int32_t* __restrict__ idx, const float* __restrict__ dpre, int B,
              int K, int ntiles, int tw, uint32_t* __restrict__ ent_pos, T* __restrict__ ent_hid, T* __restrict__ ent_dpre,
              int32_t* __restrict__ ent_off, int nchunks, const float* __restrict__ g, const bf16_t* __restrict__ gb,
              T* __restrict__ gT, int D, int ldT, const bf16_t* __restrict__ xsrc, const int32_t* __restrict__ xrows,
              T* __restrict__ xT) {
    __shared__ __attribute__((aligned(16))) char sm[64 * 65 * 4];
    const int tid = threadIdx.x;
    if ((int)blockIdx.x >= nchunks) {
        // ---- second kind of block: g [B][D] (f32, or the bf16 copy the MFMA decode kernel leaves: gb != null) ->
        // gT [D][ldT] in the contraction dtype, zero-padded beyond column B.  64 x 64 tiles through LDS; 8/16-byte
        // reads along d, 8/16-byte writes along b.
        // A third kind (blocks past the g tiles, only when the encoder GEMM gathered its rows itself and no staging
        // launch ran): the batch rows of x (bf16, gathered through xrows) -> xT, the same way.
        float (*tile)[65] = (float (*)[65])sm;
        int t = blockIdx.x - nchunks;
        const int ntb = ldT / 64;
        const int ntr = ntb * ((D + 63) / 64);
        const bool is_x = t >= ntr;
        if (is_x) {
            t -= ntr;
            gT = xT;
        }
        const int b0 = (t % ntb) * 64, d0 = (t / ntb) * 64;
        if (sizeof(T) == 2 && (is_x || gb) && d0 + 64 <= D) {
            // bf16 -> bf16 (the benchmarked mode): 16-byte reads along d and 16-byte writes along b, two passes each
            const bf16_t* src = is_x ? xsrc : gb;
            const int q8 = tid & 7, r32 = tid >> 3;
#pragma unroll
            for (int p = 0; p < 2; ++p) {
                const int bl = r32 + 32 * p, b = b0 + bl;
                bf16x8 v = {(bf16_t)0.f, (bf16_t)0.f, (bf16_t)0.f, (bf16_t)0.f, (bf16_t)0.f, (bf16_t)0.f, (bf16_t)0.f, (bf16_t)0.f};
                if (b < B) {
                    const int64_t row = (is_x && xrows) ? (int64_t)xrows[b] : (int64_t)b;
                    v = *(const bf16x8*)(src + row * D + d0 + 8 * q8);
                }
#pragma unroll
                for (int i = 0; i < 8; ++i) tile[bl][8 * q8 + i] = (float)v[i];
            }
            __syncthreads();
#pragma unroll
            for (int p = 0; p < 2; ++p) {
                const int dl = r32 + 32 * p, b = b0 + 8 * q8;
                if (b < ldT) {
                    bf16x8 o;
#pragma unroll
                    for (int i = 0; i < 8; ++i) o[i] = (bf16_t)tile[8 * q8 + i][dl];
                    *(bf16x8*)((bf16_t*)gT + (int64_t)(d0 + dl) * ldT + b) = o;
                }
            }
            return;
        }
        const int q = tid & 15, r16 = tid >> 4;
#pragma unroll
        for (int p = 0; p < 4; ++p) {
            const int bl = r16 + 16 * p, b = b0 + bl, d = d0 + 4 * q;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (b < B && d < D) {
                if (is_x) {
                    const int64_t src = xrows ? (int64_t)xrows[b] : (int64_t)b;
                    const bf16x4 t4 = *(const bf16x4*)(xsrc + src * D + d);
                    v = make_float4((float)t4[0], (float)t4[1], (float)t4[2], (float)t4[3]);
                } else if (gb) {
                    const bf16x4 t4 = *(const bf16x4*)(gb + (int64_t)b * D + d);
                    v = make_float4((float)t4[0], (float)t4[1], (float)t4[2], (float)t4[3]);
                } else {
                    v = *(const float4*)(g + (int64_t)b * D + d);
                }
            }
            tile[bl][4 * q] = v.x; tile[bl][4 * q + 1] = v.y; tile[bl][4 * q + 2] = v.z; tile[bl][4 * q + 3] = v.w;
        }
        __syncthreads();
#pragma unroll
        for (int p = 0; p < 4; ++p) {
            const int dl = r16 + 16 * p, d = d0 + dl, b = b0 + 4 * q;
            if (d < D && b < ldT) {
                if (sizeof(T) == 2) {
                    bf16x4 o;
#pragma unroll
                    for (int i = 0; i < 4; ++i) o[i] = (bf16_t)tile[4 * q + i][dl];
                    *(bf16x4*)(gT + (int64_t)d * ldT + b) = o;
                } else {
                    *(float4*)(gT + (int64_t)d * ldT + b) =
                        make_float4(tile[4 * q][dl], tile[4 * q + 1][dl], tile[4 * q + 2][dl], tile[4 * q + 3][dl]);
                }
            }
        }
        return;
    }
    int* cnt = (int*)sm;
    int* cur = cnt + BUCKET_MAX_TILES;
    constexpr int KT = Mfma<T>::KT;
    const int chunk = blockIdx.x;
    const int b0 = chunk * KT;
    const int nent = min(KT, B - b0) * K;
    const int64_t base = (int64_t)b0 * K;
    // Entries are fetched in batches of EB per thread before anything depends on them: a loop with one dependent
    // load -> LDS atomic -> store chain per trip ran 8 trips of full memory latency (12 us for this block kind alone).
    constexpr int EB = 8;
    for (int t = tid; t < ntiles; t += 256) cnt[t] = 0;
    __syncthreads();
    for (int e0 = 0; e0 < nent; e0 += 256 * EB) {
        int f[EB];
#pragma unroll
        for (int j = 0; j < EB; ++j) f[j] = idx[base + min(e0 + tid + 256 * j, nent - 1)];
#pragma unroll
        for (int j = 0; j < EB; ++j)
            if (e0 + tid + 256 * j < nent) atomicAdd(&cnt[f[j] / tw], 1);
    }
    __syncthreads();
    if (tid < 64) {  // exclusive scan over tiles by one wave
        int carry = 0;
        for (int t0 = 0; t0 < ntiles; t0 += 64) {
            const int t = t0 + tid;
            const int c = t < ntiles ? cnt[t] : 0;
            int incl = c;
#pragma unroll
            for (int o = 1; o < 64; o <<= 1) {
                const int n = __shfl_up(incl, o, 64);
                if (tid >= o) incl += n;
            }
            if (t < ntiles) {
                cur[t] = carry + incl - c;
                ent_off[(int64_t)chunk * (ntiles + 1) + t] = (int)base + carry + incl - c;
            }
            carry += __shfl(incl, 63, 64);
        }
        if (tid == 0) ent_off[(int64_t)chunk * (ntiles + 1) + ntiles] = (int)base + carry;
    }
    __syncthreads();
    for (int e0 = 0; e0 < nent; e0 += 256 * EB) {
        int f[EB], p[EB];
        float v[EB], dp[EB];
#pragma unroll
        for (int j = 0; j < EB; ++j) {
            const int e = min(e0 + tid + 256 * j, nent - 1);
            f[j] = idx[base + e];
            v[j] = vals[base + e];
            dp[j] = dpre[base + e];
        }
#pragma unroll
        for (int j = 0; j < EB; ++j) {
            const int t = f[j] / tw;
            p[j] = (e0 + tid + 256 * j < nent) ? atomicAdd(&cur[t], 1) : -1;
            f[j] -= t * tw;
        }
#pragma unroll
        for (int j = 0; j < EB; ++j) {
            if (p[j] < 0) continue;
            const int e = e0 + tid + 256 * j;
            ent_pos[base + p[j]] = ((uint32_t)f[j] << 16) | (uint32_t)(e / K);
            ent_hid[base + p[j]] = (T)(v[j] > 0.f ? v[j] : 0.f);
            ent_dpre[base + p[j]] = (T)dp[j];
        }
    }
}

// ------------------------------------------------------------------------------------------------
// bucket_sort_kernel: the counting sort of bucket_kernel's first block kind alone, for the row-major contraction (no
// transposition blocks in the launch).  1024 threads per 64-row chunk, every entry (feature, value, dpre) fetched ONCE,
// up front, into registers: the count pass and the scatter pass both run from there, so a block pays one global round trip
// instead of two, with 16 waves per CU in flight instead of 4.
// ------------------------------------------------------------------------------------------------
template <typename T>
__global__ void __launch_bounds__(1024)
bucket_sort_kernel(const float* __restrict__ vals, const int32_t* __restrict__ idx, const float* __restrict__ dpre, int B,
                   int K, int ntiles, int tw, uint32_t* __restrict__ ent_pos, T* __restrict__ ent_hid, T* __restrict__ ent_dpre,
                   int32_t* __restrict__ ent_off) {
    __shared__ int cnt[BUCKET_MAX_TILES];
    __shared__ int cur[BUCKET_MAX_TILES];
    constexpr int KT = Mfma<T>::KT;
    constexpr int EB = 4;  // entries per thread: KT * K <= 64 * 64 = 4096
    const int tid = threadIdx.x;
    const int chunk = blockIdx.x;
    const int b0 = chunk * KT;
    const int nent = min(KT, B - b0) * K;
    const int64_t base = (int64_t)b0 * K;
    int f[EB];
    float v[EB], dp[EB];
#pragma unroll
    for (int j = 0; j < EB; ++j) {
        const int e = min(tid + 1024 * j, nent - 1);
        f[j] = idx[base + e];
        v[j] = vals[base + e];
        dp[j] = dpre[base + e];
    }
    for (int t = tid; t < ntiles; t += 1024) cnt[t] = 0;
    __syncthreads();
    int tl[EB];
#pragma unroll
    for (int j = 0; j < EB; ++j) {
        tl[j] = f[j] / tw;
        if (tid + 1024 * j < nent) atomicAdd(&cnt[tl[j]], 1);
    }
    __syncthreads();
    if (tid < 64) {  // exclusive scan over tiles by one wave
        int carry = 0;
        for (int t0 = 0; t0 < ntiles; t0 += 64) {
            const int t = t0 + tid;
            const int c = t < ntiles ? cnt[t] : 0;
            int incl = c;
#pragma unroll
            for (int o = 1; o < 64; o <<= 1) {
                const int n = __shfl_up(incl, o, 64);
                if (tid >= o) incl += n;
            }
            if (t < ntiles) {
                cur[t] = carry + incl - c;
                ent_off[(int64_t)chunk * (ntiles + 1) + t] = (int)base + carry + incl - c;
            }
            carry += __shfl(incl, 63, 64);
        }
        if (tid == 0) ent_off[(int64_t)chunk * (ntiles + 1) + ntiles] = (int)base + carry;
    }
    __syncthreads();
#pragma unroll
    for (int j = 0; j < EB; ++j) {
        const int e = tid + 1024 * j;
        if (e >= nent) continue;
        const int p = atomicAdd(&cur[tl[j]], 1);
        ent_pos[base + p] = ((uint32_t)(f[j] - tl[j] * tw) << 16) | (uint32_t)(e / K);
        ent_hid[base + p] = (T)(v[j] > 0.f ? v[j] : 0.f);
        ent_dpre[base + p] = (T)dp[j];
    }
}

// ------------------------------------------------------------------------------------------------
// wgrad_kernel: one 128-feature x 128-column tile of dW_dT (which = 0) or dW_e (which = 1) over one
// batch split.  Per KT-row chunk: zero the sparse slice A[buf], scatter the tile's bucketed entries
// into it, stage the dense slab Bt[buf], MFMA.  A and Bt are double-buffered so that two barriers
// per chunk suffice: zero(k) may start as soon as every wave has passed the second barrier of chunk
// k-1 (all reads of buffer k&1 belong to chunk k-2), and the operands of chunk k+1 are fetched into
// registers while the MFMAs of chunk k run.
// ------------------------------------------------------------------------------------------------
// NT = 128-column groups per workgroup: the workgroup has 4*NT waves, wave group g works on columns
// [d0 + 128 g, d0 + 128 (g+1)) against the SAME sparse slice, so one zero+scatter feeds NT times the MFMA work.
template <typename T, int NT>
__global__ void __launch_bounds__(256 * NT)
wgrad_kernel(const uint32_t* __restrict__ ent_pos, const T* __restrict__ ent_hid, const T* __restrict__ ent_dpre,
             const int32_t* __restrict__ ent_off, const T* __restrict__ xT, const T* __restrict__ gT, int B, int ldT,
             int H, int D, int nsplit, int ntm, int ntn, float* __restrict__ out, int64_t slab_stride,
             float* __restrict__ dbe_slab) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int NTHR = 256 * NT;
    constexpr int STAGE = (1 + NT) * TILE_LDS_BYTES;  // A slice + NT dense slabs
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int grp = wave >> 2, gtid = tid & 255;      // column group, thread index inside it
    const int wm = (wave >> 1) & 1, wn = wave & 1;
    const int split = blockIdx.x % nsplit;
    int tile = blockIdx.x / nsplit;
    const int which = tile / (ntm * ntn);  // 0: dW_dT (hidden, g)   1: dW_e (dpre, x_c)
    tile -= which * ntm * ntn;
    const int tm = tile / ntn;
    const int f0 = tm * TILE_M, d0 = (tile % ntn) * TILE_N * NT + grp * TILE_N;
    constexpr int KT = Mfma<T>::KT;
    const T* Bt = which == 0 ? gT : xT;
    const T* sv = which == 0 ? ent_hid : ent_dpre;
    const bool do_dbe = (which == 1) && (tile % ntn == 0);

    const int nchunks = (B + KT - 1) / KT;
    const int per = (nchunks + nsplit - 1) / nsplit;
    const int c_begin = split * per, c_end = min(nchunks, c_begin + per);

    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    // db_e = row sums of the dpre slice: the waves of column group 0 / wn 0 run one extra MFMA against a
    // ones operand per A fragment (fixed summation order, no LDS re-read, no atomics)
    const bool rs_wave = do_dbe && grp == 0 && wn == 0;
    f32x16 rs[2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) rs[i][r] = 0.f;

    SlabRegs<T> rb;
    int e_lo = 0, e_n = 0;   // entry range of the chunk whose operands sit in registers
    int n_lo = 0, n_n = 0;   // ... and of the chunk after it (offsets run one chunk ahead of the entries)
    uint32_t e_pos = 0;      // first NTHR entries: one per thread
    T e_val = (T)0.f;
    // All loads below are unpredicated (indices clamped instead): predicated loads sit behind exec
    // branches, and hipcc then drains every outstanding load (vmcnt(0)) at the next use.
    auto offsets = [&](int ck) {
        const int32_t* o = ent_off + (int64_t)min(ck, nchunks - 1) * (ntm + 1) + tm;
        n_lo = o[0];
        n_n = o[1] - n_lo;
    };
    auto fetch = [&](int ck) {  // entry range of ck must already sit in (n_lo, n_n)
        slab_load_fast<T>(rb, Bt, ldT, d0, D - 1, ck * KT, gtid);  // rows past D repeat row D-1: never stored
        e_lo = n_lo;
        e_n = n_n;
        const int ei = e_lo + min(tid, max(e_n - 1, 0));  // lanes past the bucket re-read its last entry
        e_pos = ent_pos[ei];
        e_val = sv[ei];
        offsets(ck + 1);
    };
    if (c_begin < c_end) {
        offsets(c_begin);
        fetch(c_begin);
    }
    for (int ck = c_begin; ck < c_end; ++ck) {
        const int buf = (ck - c_begin) & 1;
        char* As = smem + buf * STAGE;
        char* Bs = As + (1 + grp) * TILE_LDS_BYTES;
        for (int c = tid; c < TILE_LDS_BYTES / 16; c += NTHR) *(uint4*)(As + c * 16) = make_uint4(0, 0, 0, 0);
        slab_store<T>(rb, Bs, gtid);
        __syncthreads();
        const int n = e_n, lo = e_lo;
        if (tid < n) *(T*)(As + (e_pos >> 16) * LDS_ROW_BYTES + (e_pos & 0xFFFFu) * (int)sizeof(T)) = e_val;
        for (int e = tid + NTHR; e < n; e += NTHR) {  // buckets beyond one entry per thread (rare)
            const uint32_t p = ent_pos[lo + e];
            *(T*)(As + (p >> 16) * LDS_ROW_BYTES + (p & 0xFFFFu) * (int)sizeof(T)) = sv[lo + e];
        }
        if (ck + 1 < c_end) fetch(ck + 1);  // next chunk's operands fly during the MFMAs
        __syncthreads();
        if (rs_wave)
            Mfma<T>::template slab<true>(As, Bs, wm * 64, wn * 64, lane, acc, rs);
        else
            Mfma<T>::template slab<false>(As, Bs, wm * 64, wn * 64, lane, acc);
    }

    // slab layout [row of the 2H x D matrix][split][D]: the 8 partial rows grad_finish sums are one contiguous 8 D run
    const int64_t rstr = (int64_t)WSAE_WGRAD_MAX_SPLIT * D;
    float* dst = out + (which == 0 ? (int64_t)H * rstr : 0) + (int64_t)split * D;
    const int col = lane & 31, rq = lane >> 5;
#pragma unroll
    for (int mi = 0; mi < 2; ++mi)
#pragma unroll
        for (int ni = 0; ni < 2; ++ni) {
            const int d = d0 + wn * 64 + ni * 32 + col;
            if (d >= D) continue;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int f = f0 + wm * 64 + mi * 32 + (r & 3) + 8 * (r >> 2) + 4 * rq;
                if (f < H) dst[(int64_t)f * rstr + d] = acc[mi][ni][r];
            }
        }
    if (rs_wave && col == 0) {  // column 0 of the ones product = the row sums
#pragma unroll
        for (int mi = 0; mi < 2; ++mi)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int f = f0 + wm * 64 + mi * 32 + (r & 3) + 8 * (r >> 2) + 4 * rq;
                if (f < H) dbe_slab[(int64_t)split * H + f] = rs[mi][r];
            }
    }
}

// ------------------------------------------------------------------------------------------------
// wgrad2_kernel: the same contraction on a 192-feature x 384-column workgroup tile (D > 256).
//   * 8 waves as 2 (features) x 4 (columns), wave tile 96 x 96: 6 fragment reads per 9 MFMAs
//     (wgrad_kernel: 4 per 4), so the MFMA phase is bound by the matrix cores, not by LDS reads;
//   * 128-byte swizzled LDS rows (wsae_mfma.h): A slice 24 KB + dense slab 48 KB per stage, two
//     stages = 144 KB, one workgroup per CU;
//   * the dense slab of chunk k+1 is written by LDS-DMA (global_load_lds_dwordx4, 6 per wave) while
//     the MFMAs of chunk k run: no staging registers, no ds_write pass;
//   * ceil(H/192) * ceil(D/384) * 2 tiles * nsplit workgroups: 256 at cfg2 with nsplit = 8, one
//     round on the 256 CUs with every tile of one batch range on one XCD.
// Per chunk: [entries(k+1) -> registers, DMA(k+1), zero A(k+1)] -> MFMA(k) -> barrier -> scatter(k+1)
// -> barrier.  Every VMEM operation issued at the top of an iteration has the whole MFMA phase to land,
// so the vmcnt(0) that __syncthreads() implies costs nothing.
// ------------------------------------------------------------------------------------------------
#define W2_M 192
#define W2_N 384
#define W2_A_BYTES (W2_M * SWZ_ROW_BYTES)
#define W2_B_BYTES (W2_N * SWZ_ROW_BYTES)
#define W2_STAGE (W2_A_BYTES + W2_B_BYTES)
#define W2_MI 3  // MFMA row tiles per wave: 2 -> 12 waves (3 x 4), 3 -> 8 waves (2 x 4)
#define W2_WAVES (W2_M / (32 * W2_MI) * 4)
#define W2_THREADS (64 * W2_WAVES)
#define W2_PIECES (W2_N / 8 / W2_WAVES)  // 1 KB LDS-DMA pieces per wave and chunk

// RM (bf16 only): the dense operands are read ROW-MAJOR as they already lie in memory - g as the decode launch leaves it
// (gb [B][D]), x as the batch's rows of the activation ring (through the step's row list) or the staged xb [B][D] - so
// no launch has to write the K-contiguous transposes xT / gT (2 x 12.6 MB out and back in per step).  Image of a stage's
// dense operand: [6 column blocks of 64][64 batch rows][128 bytes]; one LDS-DMA piece = 8 batch rows x 128 bytes of
// one column block (a wave's six pieces are the six column blocks of the same 8 rows: one row pointer per lane and
// chunk); the MFMA B fragments are taken with ds_read_b64_tr_b16 (cdna guide T10): per 16-lane group 4 batch rows x
// 16 columns, delivered column-major = 4 consecutive k of one column per lane, two reads per fragment.  Chunk c of
// row r sits at slot c ^ rm_swz(r): a 32-lane half reads 4 rows x 64 bytes, rows r and r + 2 share a bank half and land in
// different 64-byte halves of it (bit 2 of the slot), so the transposed reads are conflict-free.
__device__ __forceinline__ int rm_swz(int r) { return (((r >> 1) & 1) << 2) | ((r >> 2) & 3); }

template <typename T, bool RM>
__global__ void __launch_bounds__(W2_THREADS)
wgrad2_kernel(const uint32_t* __restrict__ ent_pos, const T* __restrict__ ent_hid, const T* __restrict__ ent_dpre,
              const int32_t* __restrict__ ent_off, const T* __restrict__ xT, const T* __restrict__ gT, int B, int ldT,
              int H, int D, int nsplit, int ntm, int ntn, float* __restrict__ out, int64_t slab_stride,
              float* __restrict__ dbe_slab, const int32_t* __restrict__ xrows, int which0, int nslots, int dec_row_base) {
    static_assert(!RM || sizeof(T) == 2, "the row-major operand path is the bf16 one (16-bit transposed LDS reads)");
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int KT = SWZ_ROW_BYTES / (int)sizeof(T);
    constexpr int EPC = 16 / (int)sizeof(T);  // elements per 16-byte chunk
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 2, wn = wave & 3;
    const int split = blockIdx.x % nsplit;
    int tile = blockIdx.x / nsplit;
    const int which = which0 + tile / (ntm * ntn);  // 0: dW_dT (hidden, g)   1: dW_e (dpre, x_c)
    tile -= (which - which0) * ntm * ntn;
    const int tm = tile / ntn, tn = tile % ntn;
    const int f0 = tm * W2_M, d0 = tn * W2_N;
    const T* Bt = which == 0 ? gT : xT;  // RM: [B][D] row-major (x: indexed through xrows when given)
    const T* sv = which == 0 ? ent_hid : ent_dpre;
    const bool do_dbe = (which == 1) && (tn == 0);

    const int nchunks = (B + KT - 1) / KT;
    const int per = (nchunks + nsplit - 1) / nsplit;
    const int c_begin = split * per, c_end = min(nchunks, c_begin + per);

    f32x16 acc[W2_MI][3];
#pragma unroll
    for (int i = 0; i < W2_MI; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    // db_e: wave w sums the 16-row tiles w and w + W2_WAVES (12 tiles) with 16x16 MFMAs against ones
    f32x4 rs0 = {0.f, 0.f, 0.f, 0.f}, rs1 = {0.f, 0.f, 0.f, 0.f};

    // per-lane constants of the DMA: piece p = wave + 8 j covers rows 8 p .. 8 p + 7 of the dense slab
    const int dma_r = lane >> 3, dma_s = lane & 7;
    const uint32_t smem_lds = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) char*)smem;
    // dense slab of chunk ck into stage 0 / 1: half of this wave's pieces per call (j0 = 0 or W2_PIECES / 2)
    // RM: source row of this lane's pieces for the chunk the next dma3 calls fetch (set by rm_row, a chunk ahead)
    // (rm_row only LOADS the row-list entry; rm_use turns it into the pointer at the top of the next iteration, so the
    // wait hipcc puts in front of the first use of a loaded value falls behind that iteration's dma_wait, not into the MFMA
    // phase the load was issued from - the hardware counts the asm DMA pieces in the same in-order vmcnt)
    const T* rm_src = Bt;
    int rm_b = 0;
    const int rm_c = (dma_s ^ rm_swz(wave * 8 + dma_r)) * EPC;  // source chunk of this lane's LDS slot
    auto rm_row = [&](int ck) {
        const int b = min(ck * KT + wave * 8 + dma_r, B - 1);  // rows past the batch repeat its last row (their A columns are zero)
        rm_b = (which == 1 && xrows) ? xrows[b] : b;
    };
    auto rm_use = [&]() { rm_src = Bt + (int64_t)rm_b * D; };
    auto dma3 = [&](int ck, int stage, int j0) {
#pragma unroll
        for (int j = j0; j < j0 + W2_PIECES / 2; ++j) {
            if constexpr (RM) {
                // piece j of this wave: column block j (64 columns) of batch rows 8 wave .. 8 wave + 7
                static_assert(!RM || (W2_PIECES == W2_N / 64 && W2_WAVES * 8 == 64), "one wave = 8 batch rows of every column block");
                const int dcol = min(d0 + j * 64 + rm_c, D - EPC);  // column blocks past D repeat the last chunk: never stored
                glds16(rm_src + dcol, smem_lds + stage * W2_STAGE + W2_A_BYTES + j * 8192 + wave * 1024);
            } else {
                const int piece = wave + W2_WAVES * j;
                const int row = piece * 8 + dma_r;
                const int c = dma_s ^ ((row >> 1) & 7);
                const int d = min(d0 + row, D - 1);  // columns past D repeat column D-1: never stored
                const T* src = Bt + (int64_t)d * ldT + (int64_t)ck * KT + c * EPC;
                glds16(src, smem_lds + stage * W2_STAGE + W2_A_BYTES + piece * 1024);
            }
        }
    };
    auto put = [&](char* As, uint32_t p, T v) {
        const int f = (int)(p >> 16), bb = (int)(p & 0xFFFFu) * (int)sizeof(T);
        *(T*)(As + swz_off(f, bb >> 4) + (bb & 15)) = v;
    };

    // The sparse slice is never re-zeroed: after the MFMAs of a chunk its entries are overwritten with
    // zeros again (un-scatter, <= 1 two-byte store per thread) instead of clearing 24 KB per chunk.
    int e_lo = 0, e_n = 0;   // entry range of the chunk whose entries sit in registers (the NEXT chunk)
    int p_lo = 0, p_n = 0;   // ... of the chunk the MFMAs are working on (for the un-scatter)
    int n_lo = 0, n_n = 0;   // ... and of the chunk after e (offsets run one chunk ahead of the entries)
    uint32_t e_pos = 0, p_pos = 0;
    T e_val = (T)0.f;
    auto offsets = [&](int ck) {
        const int32_t* o = ent_off + (int64_t)min(ck, nchunks - 1) * (ntm + 1) + tm;
        n_lo = o[0];
        n_n = o[1] - n_lo;
    };
    auto entries = [&](int ck) {  // entry range of ck must already sit in (n_lo, n_n)
        p_lo = e_lo; p_n = e_n; p_pos = e_pos;
        e_lo = n_lo;
        e_n = n_n;
        const int ei = e_lo + min(tid, max(e_n - 1, 0));  // lanes past the bucket re-read its last entry
        e_pos = ent_pos[ei];
        e_val = sv[ei];
        offsets(ck + 1);
    };
    auto scatter = [&](char* As) {
        const int n = e_n, lo = e_lo;
        if (tid < n) put(As, e_pos, e_val);
        for (int e = tid + W2_THREADS; e < n; e += W2_THREADS) put(As, ent_pos[lo + e], sv[lo + e]);
    };
    auto unscatter = [&](char* As, int lo, int n, uint32_t pos) {
        if (tid < n) put(As, pos, (T)0.f);
        for (int e = tid + W2_THREADS; e < n; e += W2_THREADS) put(As, ent_pos[lo + e], (T)0.f);
    };

    if (c_begin < c_end) {
        offsets(c_begin);
        entries(c_begin);
        if constexpr (RM) {
            rm_row(c_begin);
            rm_use();
        }
        dma3(c_begin, 0, 0);
        dma3(c_begin, 0, W2_PIECES / 2);
        if constexpr (RM) rm_row(min(c_begin + 1, c_end - 1));
#pragma unroll
        for (int st = 0; st < 2; ++st)
#pragma unroll
            for (int c = 0; c < W2_A_BYTES / 16 / W2_THREADS; ++c)
                *(uint4*)(smem + st * W2_STAGE + (tid + c * W2_THREADS) * 16) = make_uint4(0, 0, 0, 0);
        __syncthreads();
        scatter(smem);
        dma_wait();
        __syncthreads();
    }
    for (int ck = c_begin; ck < c_end; ++ck) {
        const int buf = (ck - c_begin) & 1;
        char* cur = smem + buf * W2_STAGE;
        char* nxt = smem + (buf ^ 1) * W2_STAGE;
        const bool more = ck + 1 < c_end;
        entries(min(ck + 1, c_end - 1));  // (the last iteration re-reads its own chunk: keeps p_* = this chunk)
        if constexpr (RM) rm_use();       // row pointer of chunk ck + 1 (its list entry was requested a chunk ago)
        // (one call site: two copies of the MFMA phase made hipcc double the accumulators and spill)
        Mfma96<T, W2_MI>::template slab<RM, false, RM>(cur, cur + W2_A_BYTES, wm * 32 * W2_MI, wn * 96, lane, acc, [&](int kk) {
            if (more && kk < 2) dma3(ck + 1, buf ^ 1, kk * (W2_PIECES / 2));  // >= half a phase to land
            if constexpr (RM) {
                if (kk == 2) rm_row(min(ck + 2, c_end - 1));  // the row list entry of the chunk after next (an L2 hit by then)
            }
        });
        if (do_dbe) {
            Mfma96<T, W2_MI>::rowsum16(cur, wave * 16, lane, rs0);
            if (wave + W2_WAVES < W2_M / 16) Mfma96<T, W2_MI>::rowsum16(cur, (wave + W2_WAVES) * 16, lane, rs1);
        }
        dma_wait();  // issued during the MFMA phase of a 2 us chunk: landed
        __syncthreads();
        if (more) {
            unscatter(cur, p_lo, p_n, p_pos);
            scatter(nxt);
        }
        __syncthreads();
    }

    // slab layout [row][slot][D] (nslots slots per row: the partial rows grad_finish sums are one contiguous run); rows
    // [0, H) = dW_e, the dW_dT rows start at dec_row_base (H when one launch holds both matrices, 0 when it holds one)
    const int64_t rstr = (int64_t)nslots * D;
    float* dst = out + (which == 0 ? (int64_t)dec_row_base * rstr : 0) + (int64_t)split * D;
    const int col = lane & 31, rq = lane >> 5;
    if (f0 + W2_M <= H && d0 + W2_N <= D) {
        // Interior tiles: the accumulators go through an LDS patch (the stages are free now) and leave as 16-byte stores,
        // 36 store instructions per wave instead of 144 four-byte ones.  The tail of this kernel is bound by the number
        // of store instructions, not by their bytes (bf16 slabs - half the bytes, same instructions - did not move it).
        constexpr int PSW = 100;  // floats per patch row: 96 + 4 (16-byte aligned rows, conflict-free column writes)
        float* patch = (float*)smem + wave * 32 * PSW;
        float* drow = dst + (int64_t)(f0 + wm * 32 * W2_MI) * rstr + d0 + wn * 96;
#pragma unroll
        for (int mi = 0; mi < W2_MI; ++mi) {
#pragma unroll
            for (int ni = 0; ni < 3; ++ni)
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    patch[((r & 3) + 8 * (r >> 2) + 4 * rq) * PSW + ni * 32 + col] = acc[mi][ni][r];
            __builtin_amdgcn_wave_barrier();
#pragma unroll
            for (int i = 0; i < 12; ++i) {
                const int idx = lane + 64 * i;      // float4 number inside the 32 x 96 patch
                const int row = idx / 24, c4 = idx - row * 24;
                const float4 v = *(const float4*)(patch + row * PSW + c4 * 4);
                *(float4*)(drow + (int64_t)(mi * 32 + row) * rstr + c4 * 4) = v;
            }
            __builtin_amdgcn_wave_barrier();
        }
    } else {
#pragma unroll
        for (int mi = 0; mi < W2_MI; ++mi)
#pragma unroll
            for (int ni = 0; ni < 3; ++ni) {
                const int d = d0 + wn * 96 + ni * 32 + col;
                if (d >= D) continue;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int f = f0 + wm * 32 * W2_MI + mi * 32 + (r & 3) + 8 * (r >> 2) + 4 * rq;
                    if (f < H) dst[(int64_t)f * rstr + d] = acc[mi][ni][r];
                }
            }
    }
    if (do_dbe && (lane & 15) == 0) {  // column 0 of the ones product = the row sums
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int fa = f0 + wave * 16 + 4 * (lane >> 4) + j;
            if (fa < H) dbe_slab[(int64_t)split * H + fa] = rs0[j];
            const int fb = f0 + (wave + W2_WAVES) * 16 + 4 * (lane >> 4) + j;
            if (wave + W2_WAVES < W2_M / 16 && fb < H) dbe_slab[(int64_t)split * H + fb] = rs1[j];
        }
    }
}

// ------------------------------------------------------------------------------------------------
// wgrad2d_kernel: wgrad2_kernel's geometry (192 x 384 tiles, 8 waves as 2 x 4, wave tile 96 x 96, split-K over the batch,
// both matrices in one launch: 32 tiles x 8 splits = one workgroup per CU at 384 -> 3072) with a DENSE left operand: the ReLU
// SAE's contractions dW_dT = hidden^T g and dW_e = dpre^T x (wsae_relu.hip), where hidden and dpre are bf16 [B][H] matrices
// read row-major like the right operands.  Per 64-row chunk a wave DMA-gathers its 8 batch rows of three 64-feature blocks
// (left) and six 64-column blocks (right); both fragment kinds are transposed LDS reads.  No code, no scatter: ONE barrier per
// chunk.  (The 256 x 256 tiles of gemm256x_kernel put a quarter of their MFMA work outside a 384-wide matrix and fill 192 of
// 256 CUs: 2 x 64 us against this launch's single pass.)  B must be a multiple of 64 (no zero padding of a dense operand).
// ------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(W2_THREADS)
wgrad2d_kernel(const bf16_t* __restrict__ hid, const bf16_t* __restrict__ dpre, const bf16_t* __restrict__ xb,
               const bf16_t* __restrict__ gb, int B, int H, int D, int nsplit, int ntm, int ntn, float* __restrict__ out,
               int nslots) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int KT = 64, EPC = 8;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 2, wn = wave & 3;
    const int split = blockIdx.x % nsplit;
    int tile = blockIdx.x / nsplit;
    const int which = tile / (ntm * ntn);  // 0: dW_dT (hidden, g)   1: dW_e (dpre, x)
    tile -= which * ntm * ntn;
    const int tm = tile / ntn, tn = tile % ntn;
    const int f0 = tm * W2_M, d0 = tn * W2_N;
    const bf16_t* Am = which == 0 ? hid : dpre;  // [B][H]
    const bf16_t* Bm = which == 0 ? gb : xb;     // [B][D]
    const int nchunks = B / KT;
    const int per = (nchunks + nsplit - 1) / nsplit;
    const int c_begin = split * per, c_end = min(nchunks, c_begin + per);

    f32x16 acc[W2_MI][3];
#pragma unroll
    for (int i = 0; i < W2_MI; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    const int dma_r = lane >> 3, dma_s = lane & 7;
    const uint32_t smem_lds = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) char*)smem;
    const int rm_c = (dma_s ^ rm_swz(wave * 8 + dma_r)) * EPC;  // source chunk of this lane's LDS slot
    // chunk ck into stage `stage`: this wave's 8 batch rows x (3 feature blocks | 6 column blocks); part = 0 / 1 / 2 issues
    // (A blocks) / (B blocks 0-2) / (B blocks 3-5) so that the pieces can be spread over the MFMA phase
    auto dma = [&](int ck, int stage, int part) {
        const int64_t b = (int64_t)ck * KT + wave * 8 + dma_r;
        const uint32_t base = smem_lds + stage * W2_STAGE;
        if (part == 0) {
#pragma unroll
            for (int j = 0; j < 3; ++j)
                glds16_nt(Am + b * H + min(f0 + j * 64 + rm_c, H - EPC), base + j * 8192 + wave * 1024);  // streamed once: nt
        } else {
#pragma unroll
            for (int j = 3 * (part - 1); j < 3 * part; ++j)
                glds16(Bm + b * D + min(d0 + j * 64 + rm_c, D - EPC), base + W2_A_BYTES + j * 8192 + wave * 1024);
        }
    };
    if (c_begin < c_end) {
        dma(c_begin, 0, 0); dma(c_begin, 0, 1); dma(c_begin, 0, 2);
        dma_wait();
        __syncthreads();
    }
    for (int ck = c_begin; ck < c_end; ++ck) {
        const int buf = (ck - c_begin) & 1;
        char* cur = smem + buf * W2_STAGE;
        const bool more = ck + 1 < c_end;
        Mfma96<bf16_t, W2_MI>::template slab<true, true>(  // (the pipelined form gains nothing here: the launch waits for its left operands from HBM)
            cur, cur + W2_A_BYTES, wm * 32 * W2_MI, wn * 96, lane, acc, [&](int kk) {
            if (more) dma(ck + 1, buf ^ 1, kk);  // (kk = 0, 1, 2: the three parts)
        });
        dma_wait();       // issued during this chunk's MFMA phase: landed
        __syncthreads();  // ... for every wave; and everybody is done reading `cur` (the chunk after next lands there)
    }

    const int64_t rstr = (int64_t)nslots * D;
    float* dst = out + (which == 0 ? (int64_t)H * rstr : 0) + (int64_t)split * D;
    const int col = lane & 31, rq = lane >> 5;
    if (f0 + W2_M <= H && d0 + W2_N <= D) {  // interior tiles: 16-byte stores through an LDS patch (as wgrad2_kernel)
        constexpr int PSW = 100;
        float* patch = (float*)smem + wave * 32 * PSW;
        float* drow = dst + (int64_t)(f0 + wm * 32 * W2_MI) * rstr + d0 + wn * 96;
#pragma unroll
        for (int mi = 0; mi < W2_MI; ++mi) {
#pragma unroll
            for (int ni = 0; ni < 3; ++ni)
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    patch[((r & 3) + 8 * (r >> 2) + 4 * rq) * PSW + ni * 32 + col] = acc[mi][ni][r];
            __builtin_amdgcn_wave_barrier();
#pragma unroll
            for (int i = 0; i < 12; ++i) {
                const int idx = lane + 64 * i;
                const int row = idx / 24, c4 = idx - row * 24;
                const float4 v = *(const float4*)(patch + row * PSW + c4 * 4);
                *(float4*)(drow + (int64_t)(mi * 32 + row) * rstr + c4 * 4) = v;
            }
            __builtin_amdgcn_wave_barrier();
        }
    } else {
#pragma unroll
        for (int mi = 0; mi < W2_MI; ++mi)
#pragma unroll
            for (int ni = 0; ni < 3; ++ni) {
                const int d = d0 + wn * 96 + ni * 32 + col;
                if (d >= D) continue;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int f = f0 + wm * 32 * W2_MI + mi * 32 + (r & 3) + 8 * (r >> 2) + 4 * rq;
                    if (f < H) dst[(int64_t)f * rstr + d] = acc[mi][ni][r];
                }
            }
    }
}

// ------------------------------------------------------------------------------------------------
// grad_finish_kernel: everything between the split-K contraction and the optimizer (or the data-parallel exchange), in
// ONE launch of three kinds of blocks (they share nothing, so they share the launch instead of queueing behind each
// other's tails):
//   [0, nrb)               out[W rows] = sum over splits of the slabs (fixed order) for the rows [row0, row0 + nrows) of the
//                          [2 H][D] gradient matrix (rows < H: dW_e, the rest dW_dT); db_e likewise; in BF16 mode the rank-1
//                          correction of the folded pre-bias on the way:
//                              dW_e[h,:] -= db_e[h] * b_pre   (the MFMA contraction saw raw x, not x - b_pre)
//                          and the block's share of the global-norm partials;
//   [nrb, nrb + nblk_h)    db_pre partials  part[blk][d] = sum_{h in blk's 128 rows} db_e[h] * W_e[h][d]
//                          (db_e re-summed from the split slabs, so these blocks do not wait for the first kind);
//   [.., + DBD_L1)         level-1 reduction of the decode launch's column sums of g,
//                          dbd2[j][d] = sum over decode blocks i = j, j + DBD_L1, ... of part_dbd[i][d].
// The last block to arrive (two-level ticket) finishes the biases: db_d = sum of the level-1 rows,
// db_pre = db_d - sum of the partial rows, and their norm partial.  Hand-off as in the decode epilogue:
// the partial rows are agent-scope (write-through) atomic stores drained with vmcnt(0) before the ticket
// and read back with agent-scope atomic loads.  No float atomics anywhere: every sum has a fixed order.
// Loads are issued in batches: a one-at-a-time loop here is pure L2 latency.
//
// The OUTPUT is addressed through five offsets (GfOut) and an element type OT, so the same kernel writes
//   * the fp32 gradient buffer in pack order [W_e | W_dT | b_e | b_d | b_pre] (single process), or
//   * the data-parallel WIRE buffer [W_dT | W_e | b_e | b_d | b_pre | fired] in fp32 or bf16 (include/wsae.h): the decoder
//     matrix first, so that the launch that reduces the decoder half (bias kinds absent: do_bias = 0) fills ONE contiguous
//     range whose all-reduce can start while the encoder half is still being contracted; the encoder launch fills the
//     rest, copying the fired indicators behind the biases.
// NS = slab slots per row (8: both matrices in one contraction launch; 16: one matrix per launch, split-K 16).
// ------------------------------------------------------------------------------------------------
#define DBPRE_ROWS 128
#define DBD_L1 16
struct GfOut {
    int64_t oWe, oWd, oBe, oBd, oBp, oFired;
};

// the step's (loss, l0) as exact-summable digits behind the fired indicators (include/wsae.h, WSAE_WIRE_METRIC_SLOTS):
// element i of the 24: loss digits 0..9 (40-bit fixed point, 2^-24), l0 digits 10..17 (32-bit, 2^-16), 18 = non-finite flag
__device__ __forceinline__ float wire_metric_digit(const float* __restrict__ m, int i) {
    if (i >= 19 || !m) return 0.f;
    const float loss = m[0], l0 = m[1];
    const bool bad = !(fabsf(loss) <= 3.0e38f) || !(fabsf(l0) <= 3.0e38f);
    if (i == 18) return bad ? 1.f : 0.f;
    if (bad) return 0.f;
    if (i < 10) {
        const double q = fmin(fmax((double)loss, 0.0) * 16777216.0 + 0.5, 1099511627775.0);
        return (float)(((unsigned long long)q >> (4 * i)) & 15ull);
    }
    const double q = fmin(fmax((double)l0, 0.0) * 65536.0 + 0.5, 4294967295.0);
    return (float)(((unsigned long long)q >> (4 * (i - 10))) & 15ull);
}

template <typename OT>
__device__ __forceinline__ void gf_store4(OT* p, const float4& a) {
    if constexpr (sizeof(OT) == 2) {
        bf16x4 o;
        o[0] = (bf16_t)a.x; o[1] = (bf16_t)a.y; o[2] = (bf16_t)a.z; o[3] = (bf16_t)a.w;
        *(bf16x4*)p = o;
    } else {
        *(float4*)p = a;
    }
}

template <typename TW, bool FOLD, typename OT, int NS>
__global__ void __launch_bounds__(256)
grad_finish_kernel(const float* __restrict__ slabs, const float* __restrict__ dbe_slab, int nsplit,
                   const float* __restrict__ bpre, OT* __restrict__ out, GfOut o, int H, int D, float* __restrict__ part_sq,
                   int nrb, int row0, int nrows, int slab_row0, const TW* __restrict__ W, float* __restrict__ part,
                   int nblk_h, const float* __restrict__ part_dbd, int n_dec, float* __restrict__ dbd2,
                   unsigned long long* __restrict__ ticket, int do_bias, const float* __restrict__ fired_src, int with_dbe,
                   const float* __restrict__ metrics_src = nullptr) {
    __shared__ float red[8];
    __shared__ float e_s[DBPRE_ROWS];
    __shared__ __attribute__((aligned(16))) float p_s[1024];  // row-part sums of a db_pre block (D <= 512: parts x D <= 1024)
    __shared__ int last_s;
    // the few latency-bound blocks (kinds two and three) take the first block ids so that they start
    // first and run under the streaming blocks instead of forming the tail of the launch
    const int tid = threadIdx.x;
    const int nlat = do_bias ? nblk_h + DBD_L1 : 0;
    const int bid = (int)blockIdx.x >= nlat ? (int)blockIdx.x - nlat : nrb + (int)blockIdx.x;
    if (bid < nrb) {
        // A wave owns `rpw` consecutive rows of its launch's row range and walks them two at a time as ONE contiguous range
        // of float4 chunks: every slab load of up to TR full trips (NS splits x TR independent 16-byte loads per lane) is in
        // flight before anything is summed.  (Row by row, a 96-chunk row is one full trip and one half-empty trip, each a
        // separate round of memory latency: 22 us for 75 MB.)
        static_assert(NS == 8 || NS == 16, "the db_e gather packs (row, split) into the lanes: 8 x 8 or 4 x 16");
        constexpr int TR = NS == 8 ? 3 : 2;
        const int lane = tid & 63, wave = tid >> 6;
        const int nc = D >> 2;  // float4 chunks per row
        const int rpw = (nrows + nrb * 4 - 1) / (nrb * 4);
        float sq = 0.f;  // sum of squares of everything this block writes (global-norm partial)
        for (int q = 0; q < rpw; q += 2) {
            const int r0 = row0 + (bid * 4 + wave) * rpw + q;
            if (r0 >= row0 + nrows) break;
            const int nr = min(2, min(rpw - q, row0 + nrows - r0));
            // db_e[h] = sum over splits, in split order (the db_pre blocks below use the same order)
            float be[2] = {0.f, 0.f};
            if (with_dbe && r0 < H) {  // (only dW_e rows have one; the dense ReLU contraction leaves none: with_dbe = 0)
                const int j = lane / NS, sp = lane % NS;
                const float v = (j < nr && sp < nsplit && r0 + j < H) ? dbe_slab[(int64_t)sp * H + r0 + j] : 0.f;
#pragma unroll
                for (int s2 = 0; s2 < NS; ++s2) {
                    be[0] += __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), s2));
                    be[1] += __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), NS + s2));
                }
                if (lane == 0) {
#pragma unroll
                    for (int j2 = 0; j2 < 2; ++j2)
                        if (j2 < nr && r0 + j2 < H) {
                            out[o.oBe + r0 + j2] = (OT)be[j2];
                            sq = fmaf(be[j2], be[j2], sq);
                        }
                }
            }
            const int total = nr * nc;
            const float4* base = (const float4*)slabs + (int64_t)(r0 - slab_row0) * NS * nc;  // [row][slot][D]
            OT* orow[2];
#pragma unroll
            for (int j2 = 0; j2 < 2; ++j2) {
                const int r = r0 + j2;
                orow[j2] = out + (r < H ? o.oWe + (int64_t)r * D : o.oWd + (int64_t)(r - H) * D);
            }
            for (int c0 = 0; c0 < total; c0 += 64 * TR) {
                float4 v[TR][NS];
#pragma unroll
                for (int t = 0; t < TR; ++t) {
                    const int idx = min(c0 + lane + 64 * t, total - 1);  // clamped: unconditional loads
                    const int jr = idx >= nc ? 1 : 0, cc = idx - jr * nc;
                    if (nsplit == 1) {  // (small batches: one slab - one load, not NS copies of it; a uniform branch)
                        v[t][0] = nt_load4(base + (jr * NS) * nc + cc);
#pragma unroll
                        for (int sp = 1; sp < NS; ++sp) v[t][sp] = make_float4(0.f, 0.f, 0.f, 0.f);
                    } else {
#pragma unroll
                        for (int sp = 0; sp < NS; ++sp)  // splits past nsplit re-read the last one (weight 0)
                            v[t][sp] = nt_load4(base + (jr * NS + min(sp, nsplit - 1)) * nc + cc);  // (last use of the slabs)
                    }
                }
#pragma unroll
                for (int t = 0; t < TR; ++t) {
                    const int idx = c0 + lane + 64 * t;
                    if (idx >= total) continue;
                    float4 a = v[t][0];
#pragma unroll
                    for (int sp = 1; sp < NS; ++sp) {
                        const float w = sp < nsplit ? 1.f : 0.f;
                        a.x = fmaf(w, v[t][sp].x, a.x); a.y = fmaf(w, v[t][sp].y, a.y);
                        a.z = fmaf(w, v[t][sp].z, a.z); a.w = fmaf(w, v[t][sp].w, a.w);
                    }
                    const int j = idx >= nc ? 1 : 0, c = idx - j * nc;
                    if (FOLD && r0 + j < H) {
                        const float4 bp = *(const float4*)(bpre + 4 * c);
                        const float b1 = j ? be[1] : be[0];
                        a.x -= b1 * bp.x; a.y -= b1 * bp.y; a.z -= b1 * bp.z; a.w -= b1 * bp.w;
                    }
                    gf_store4<OT>((j ? orow[1] : orow[0]) + 4 * c, a);
                    sq += a.x * a.x + a.y * a.y + a.z * a.z + a.w * a.w;
                }
            }
        }
        // data parallel: the fired indicators ride behind the biases on the wire (sums of at most world ones: exact in bf16)
        if (fired_src) {
            const int per = (H + nrb - 1) / nrb;
            for (int i = tid; i < per; i += 256) {
                const int h = bid * per + i;
                if (h < H) out[o.oFired + h] = (OT)fired_src[h];
            }
            if (bid == 0 && tid < WSAE_WIRE_METRIC_SLOTS) out[o.oFired + H + tid] = (OT)wire_metric_digit(metrics_src, tid);
        }
        const float t = block_sum(sq, red);
        if (tid == 0) part_sq[bid] = t;
        if (!do_bias) return;  // (no ticket in a launch without the bias kinds: nobody finishes anything)
    } else if (bid < nrb + nblk_h) {
        // db_pre partial of DBPRE_ROWS feature rows.  Work item = (row part, group of 4 columns): every thread streams
        // its rows with 8/16-byte loads, 16 in flight; with D <= 512 the block's rows are split over 256 / (D / 4) parts
        // whose sums meet in LDS.  (One thread per column walking all 128 rows with 2-byte loads was the tail of the
        // launch.)
        const int blk = bid - nrb;
        const int h0 = blk * DBPRE_ROWS;
        if (tid < DBPRE_ROWS) {
            float be = 0.f;
            if (h0 + tid < H)
                for (int s = 0; s < nsplit; ++s) be += dbe_slab[(int64_t)s * H + h0 + tid];  // same order as above
            e_s[tid] = be;  // (0 for rows past H)
        }
        __syncthreads();
        const int ncg = D >> 2;
        int nrp = 1;                                  // row parts: the largest power of two <= min(8, 256 / groups)
        while (nrp < 8 && 2 * nrp * ncg <= 256) nrp *= 2;
        const int rpp = DBPRE_ROWS / nrp;             // rows per part: a multiple of the 16-row load batch
        for (int c = tid; c < ncg * nrp; c += 256) {
            const int cg = c % ncg, rp = c / ncg;
            float4 a = make_float4(0.f, 0.f, 0.f, 0.f);
            for (int r0 = rp * rpp; r0 < (rp + 1) * rpp; r0 += 16) {
                float4 w[16];
#pragma unroll
                for (int i = 0; i < 16; ++i) {
                    const TW* src = W + (int64_t)min(h0 + r0 + i, H - 1) * D + 4 * cg;
                    if constexpr (sizeof(TW) == 2) {
                        const bf16x4 t4 = *(const bf16x4*)src;
                        w[i] = make_float4((float)t4[0], (float)t4[1], (float)t4[2], (float)t4[3]);
                    } else {
                        w[i] = *(const float4*)src;
                    }
                }
#pragma unroll
                for (int i = 0; i < 16; ++i) {
                    const float e = e_s[r0 + i];
                    a.x = fmaf(e, w[i].x, a.x); a.y = fmaf(e, w[i].y, a.y);
                    a.z = fmaf(e, w[i].z, a.z); a.w = fmaf(e, w[i].w, a.w);
                }
            }
            if (nrp == 1) {
                float* dst = part + (int64_t)blk * D + 4 * cg;
                __hip_atomic_store(dst, a.x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                __hip_atomic_store(dst + 1, a.y, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                __hip_atomic_store(dst + 2, a.z, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                __hip_atomic_store(dst + 3, a.w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            } else {
                *(float4*)(p_s + rp * D + 4 * cg) = a;
            }
        }
        if (nrp > 1) {
            __syncthreads();
            for (int d = tid; d < D; d += 256) {
                float a = 0.f;
                for (int rp = 0; rp < nrp; ++rp) a += p_s[rp * D + d];
                __hip_atomic_store(part + (int64_t)blk * D + d, a, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        }
    } else {
        // level-1 sums of the decode launch's per-block column sums: unconditional (clamped) loads, 16 in flight,
        // rows past n_dec weighted 0 (a load under a per-element condition is a branch and a full wait each)
        const int j = bid - nrb - nblk_h;
        const int nterm = (n_dec + DBD_L1 - 1) / DBD_L1;
        for (int d = tid; d < D; d += 256) {
            float a = 0.f;
            for (int t0 = 0; t0 < nterm; t0 += 16) {
                float v[16];
#pragma unroll
                for (int t = 0; t < 16; ++t) v[t] = part_dbd[(int64_t)min(j + DBD_L1 * (t0 + t), n_dec - 1) * D + d];
#pragma unroll
                for (int t = 0; t < 16; ++t) a = fmaf(j + DBD_L1 * (t0 + t) < n_dec ? 1.f : 0.f, v[t], a);
            }
            __hip_atomic_store(dbd2 + (int64_t)j * D + d, a, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (tid == 0) {
        unsigned unused;
        last_s = grid_ticket(ticket, 0u, &unused);
    }
    __syncthreads();
    if (!last_s) return;
    // the last block to arrive finishes db_d and db_pre: every partial of a column is requested before any is summed
    float sq = 0.f;
    for (int d = tid; d < D; d += 256) {
        float sd = 0.f, sp = 0.f;
        float v[DBD_L1];
#pragma unroll
        for (int i = 0; i < DBD_L1; ++i)
            v[i] = __hip_atomic_load(dbd2 + (int64_t)i * D + d, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        for (int b0 = 0; b0 < nblk_h; b0 += 32) {
            float u[32];
#pragma unroll
            for (int i = 0; i < 32; ++i)
                u[i] = __hip_atomic_load(part + (int64_t)min(b0 + i, nblk_h - 1) * D + d, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#pragma unroll
            for (int i = 0; i < 32; ++i) sp = fmaf(b0 + i < nblk_h ? 1.f : 0.f, u[i], sp);
        }
#pragma unroll
        for (int i = 0; i < DBD_L1; ++i) sd += v[i];
        const float pp = sd - sp;
        out[o.oBd + d] = (OT)sd;
        out[o.oBp + d] = (OT)pp;
        sq += sd * sd + pp * pp;
    }
    const float t = block_sum(sq, red);
    if (tid == 0) part_sq[nrb] = t;
}

// Split-K factor.  A workgroup walks ceil(nchunks / nsplit) batch chunks and the grid runs in
// ceil(tiles * nsplit / resident workgroups) rounds, so the kernel time goes like rounds * chunks per
// split; every split also costs one slab written and re-read (the small per-split term).  At
// H = 3072, D = 384, B = 16384 (wgrad2_kernel: 32 tiles, one workgroup per CU) this picks 8: 256 workgroups,
// one round of 32 chunks; one matrix per launch (16 tiles, max_split 16) gets 16.
static int pick_nsplit(wsae_ctx* ctx, int tiles, int nchunks, int nt, int max_split, int reserve_cus = 0) {
    // (reserve_cus: compute units left without a workgroup of this launch - the contraction's workgroups fill a CU's register
    // file, so a collective that is meant to run UNDER the launch needs CUs of its own: wsae_ctx_set_comm_reserve)
    const int resident = max(1, ctx->cus - reserve_cus) * (nt == 1 ? 2 : 1);  // LDS: 72 KB per workgroup at NT = 1, 108 / 144 KB above
    int best = 1;
    int64_t best_cost = INT64_MAX;
    for (int ns = 1; ns <= max_split; ++ns) {
        if (ns > 1 && nchunks / ns < 4) break;  // >= 4 chunks per split
        const int64_t rounds = ceil_div64(tiles * ns, resident);
        const int64_t cost = rounds * ceil_div(nchunks, ns) * 16 + 24 * ns;  // 16ths of a chunk time; a slab ~ 1.5 chunks
        if (cost < best_cost) { best_cost = cost; best = ns; }
    }
    return best;
}

// geometry of one wsae_weight_grads call
struct WgPlan {
    int ldT, nchunks, nt, ntm, ntn, nsplit, nslots, nwhich, which0, dec_row_base;
    bool v2, rm;
};

// part: WSAE_PART_ALL = both contractions in one launch (split-K <= 8, slab rows [0, 2 H)); WSAE_PART_DECODER / _ENCODER = one
// matrix (split-K <= 16, slab rows [0, H)): the halves of a data-parallel step (include/wsae.h)
template <typename T>
static WgPlan plan_wgrad(wsae_ctx* ctx, int B, int part) {
    WgPlan p;
    const int D = ctx->D, H = ctx->H;
    p.ldT = (B + 127) / 128 * 128;
    p.nchunks = ceil_div(B, Mfma<T>::KT);
    const int ncol = ceil_div(D, TILE_N);  // column groups per workgroup: all of D in one workgroup when D is 2 or 3 tiles wide
    p.v2 = D > 256;                        // wgrad2_kernel (192 x 384 tiles) for wide inputs, the 128-feature kernel serves D <= 256
    p.nt = p.v2 ? 0 : (ncol % 3 == 0) ? 3 : (ncol % 2 == 0) ? 2 : 1;
    p.ntm = p.v2 ? ceil_div(H, W2_M) : ceil_div(H, TILE_M);
    p.ntn = p.v2 ? ceil_div(D, W2_N) : ncol / p.nt;
    p.nwhich = part == WSAE_PART_ALL ? 2 : 1;
    p.which0 = part == WSAE_PART_ENCODER ? 1 : 0;
    p.nslots = part == WSAE_PART_ALL ? WSAE_WGRAD_MAX_SPLIT : 2 * WSAE_WGRAD_MAX_SPLIT;
    p.dec_row_base = part == WSAE_PART_ALL ? H : 0;
    // the encoder half of a data-parallel backward runs beside the all-reduce of the decoder half
    p.nsplit = pick_nsplit(ctx, p.ntm * p.ntn * p.nwhich, p.nchunks, p.nt, p.nslots, part == WSAE_PART_ENCODER ? ctx->comm_reserve : 0);
    p.rm = sizeof(T) == 2 && p.nt == 0 && ctx->g_is_bf16 && D % 8 == 0;
    return p;
}

template <typename T>
static void launch_wgrad(wsae_ctx* ctx, const WgPlan& p, hipStream_t st, const float* vals, const int32_t* idx,
                         const float* dpre, int B, float* out, const void* x, const int32_t* rows, bool sort_code) {
    constexpr int KT = Mfma<T>::KT;
    const int nchunks = p.nchunks, ldT = p.ldT, ntm = p.ntm, nt = p.nt;
    const int64_t slab_stride = 2 * (int64_t)ctx->H * ctx->D;
    // Row-major dense operands (wgrad2_kernel<bf16, RM>): g as the MFMA decode launch left it (gb), x as the staged batch
    // xb or - when the encoder GEMM gathered its rows itself - the caller's bf16 rows through the row list.  Then the
    // bucket launch is the counting sort alone: no g -> gT / x -> xT blocks.
    const bool rm = p.rm;
    // (the decode launch of this batch may have sorted the code itself - chunked form, wsae_decode_mfma.hip: then nothing is launched)
    const bool presorted = rm && ctx->ent_valid && ctx->ent_vals == vals && ctx->ent_B == B;
    if (sort_code && !presorted) {
        WSAE_PROF_BEGIN(ctx, WSAE_K_BUCKET, st);
        // otherwise: + the transposition of g (left row-major by the decode launch) as a second kind of block in the same launch
        // ... and, when no staging launch left xT (the encoder GEMM gathered the batch rows itself), of x as a third kind
        const int ntr = rm ? 0 : (ldT / 64) * ceil_div(ctx->D, 64);
        const int nxt = ctx->xT_valid ? 0 : ntr;
        if (rm && KT * ctx->K <= 4096)
            bucket_sort_kernel<T><<<nchunks, 1024, 0, st>>>(vals, idx, dpre, B, ctx->K, ntm, W2_M, ctx->ent_pos, (T*)ctx->ent_hid,
                                                            (T*)ctx->ent_dpre, ctx->ent_off);
        else
            bucket_kernel<T><<<nchunks + ntr + nxt, 256, 0, st>>>(vals, idx, dpre, B, ctx->K, ntm, nt == 0 ? W2_M : TILE_M, ctx->ent_pos,
                                                                  (T*)ctx->ent_hid, (T*)ctx->ent_dpre, ctx->ent_off, nchunks, ctx->g,
                                                                  ctx->g_is_bf16 ? ctx->gb : nullptr, (T*)ctx->gT, ctx->D, ldT,
                                                                  (const bf16_t*)x, rows, (T*)ctx->xT);
        WSAE_PROF_END(ctx, WSAE_K_BUCKET, st);
    }
    const dim3 grid(p.ntm * p.ntn * p.nwhich * p.nsplit);
    WSAE_PROF_BEGIN(ctx, WSAE_K_WGRAD, st);
#define WG_ARGS ctx->ent_pos, (const T*)ctx->ent_hid, (const T*)ctx->ent_dpre, ctx->ent_off, (const T*)ctx->xT, \
                (const T*)ctx->gT, B, ldT, ctx->H, ctx->D, p.nsplit, ntm, p.ntn, out, slab_stride, ctx->dbe_slab
    if (nt == 0) {
        bool done = false;
        if constexpr (sizeof(T) == 2) {
            if (rm) {
                const T* xsrc = ctx->xT_valid ? (const T*)ctx->xb : (const T*)x;
                wgrad2_kernel<T, true><<<grid, W2_THREADS, 2 * W2_STAGE, st>>>(
                    ctx->ent_pos, (const T*)ctx->ent_hid, (const T*)ctx->ent_dpre, ctx->ent_off, xsrc, (const T*)ctx->gb, B, ldT,
                    ctx->H, ctx->D, p.nsplit, ntm, p.ntn, out, slab_stride, ctx->dbe_slab, ctx->xT_valid ? nullptr : rows,
                    p.which0, p.nslots, p.dec_row_base);
                done = true;
            }
        }
        if (!done) wgrad2_kernel<T, false><<<grid, W2_THREADS, 2 * W2_STAGE, st>>>(WG_ARGS, nullptr, p.which0, p.nslots, p.dec_row_base);
    }
    else if (nt == 3) wgrad_kernel<T, 3><<<grid, 768, 2 * 4 * TILE_LDS_BYTES, st>>>(WG_ARGS);
    else if (nt == 2) wgrad_kernel<T, 2><<<grid, 512, 2 * 3 * TILE_LDS_BYTES, st>>>(WG_ARGS);
    else wgrad_kernel<T, 1><<<grid, 256, 2 * 2 * TILE_LDS_BYTES, st>>>(WG_ARGS);
#undef WG_ARGS
    WSAE_PROF_END(ctx, WSAE_K_WGRAD, st);
}

// feature-tile width and count of the contraction wsae_weight_grads will run on this ctx (the chunked decode kernel sorts
// the code for it)
void wsae_internal_wgrad_tiling(const wsae_ctx* c, int* tile_width, int* ntiles) {
    const bool v2 = c->D > 256;
    *tile_width = v2 ? W2_M : TILE_M;
    *ntiles = ceil_div(c->H, *tile_width);
}

template <typename TW, bool FOLD, typename OT>
static void launch_grad_finish(wsae_ctx* ctx, const WgPlan& p, hipStream_t st, const float* bpre, OT* out, const GfOut& o,
                               const TW* W, int part, const float* fired_src) {
    const int H = ctx->H, D = ctx->D;
    const int nrows = part == WSAE_PART_ALL ? 2 * H : H;
    const int row0 = part == WSAE_PART_DECODER ? H : 0;
    const int slab_row0 = part == WSAE_PART_DECODER ? H : 0;  // (one matrix per launch: its slab rows start at 0)
    const int do_bias = part != WSAE_PART_DECODER;
    // reduction blocks: 8 rows each (two per wave), more when nrows / 8 would exceed the number of norm-partial slots
    const int nrb = (int)min((int64_t)WSAE_MAX_PARTIALS - 1, ceil_div64(nrows, 8));
    const int nblk = ceil_div(H, DBPRE_ROWS);
    unsigned long long* ticket = (unsigned long long*)(ctx->counters + 16 + 4 * TICKET_WORDS);
    const int grid = nrb + (do_bias ? nblk + DBD_L1 : 0);
#define GF_ARGS ctx->wg_slabs, ctx->dbe_slab, p.nsplit, bpre, out, o, H, D, ctx->part_sq, nrb, row0, nrows, slab_row0, W, \
                ctx->dbpre_part, nblk, ctx->part_dbd, ctx->n_dec_blocks, ctx->dbd2, ticket, do_bias, fired_src, 1, \
                (fired_src ? ctx->wire_metrics : nullptr)
    if (p.nslots == WSAE_WGRAD_MAX_SPLIT) grad_finish_kernel<TW, FOLD, OT, WSAE_WGRAD_MAX_SPLIT><<<grid, 256, 0, st>>>(GF_ARGS);
    else grad_finish_kernel<TW, FOLD, OT, 2 * WSAE_WGRAD_MAX_SPLIT><<<grid, 256, 0, st>>>(GF_ARGS);
#undef GF_ARGS
    ctx->n_sq_parts = part == WSAE_PART_ALL ? nrb + 1 : 0;  // wsae_adamw_step(norm_from_wgrad = 1) sums these (halves: the wire unpack leaves the norm)
}

static int weight_grads_impl(wsae_ctx* ctx, const float* params, const void* x, int32_t x_dtype, const int32_t* rows,
                             const float* vals, const int32_t* idx, const float* dpre, int32_t B, int part, float* grads,
                             void* wire, int wire_dtype, hipStream_t st) {
    WSAE_REQUIRE(ctx && params && vals && idx && dpre, "wsae_weight_grads: null argument");
    // x is read only when the forward skipped the staging launch (then it must be the same bf16 batch)
    WSAE_REQUIRE(ctx->xT_valid || (x && x_dtype == WSAE_DT_BF16 && ctx->prec == WSAE_PREC_BF16),
                 "wsae_weight_grads: the forward of this batch left no staged x^T and x is not a bf16 buffer");
    WSAE_REQUIRE(B >= 1 && B <= ctx->maxB, "wsae_weight_grads: batch %d outside [1, %d]", B, ctx->maxB);
    const int H = ctx->H;
    const bool bf = ctx->prec == WSAE_PREC_BF16;
    const WgPlan p = bf ? plan_wgrad<bf16_t>(ctx, B, part) : plan_wgrad<float>(ctx, B, part);
    WSAE_REQUIRE(p.ntm <= BUCKET_MAX_TILES, "hidden_dim %d too large for the bucket pass (max %d)", H, BUCKET_MAX_TILES * 128);
    WSAE_REQUIRE(part == WSAE_PART_ALL || p.nt == 0, "wsae_weight_grads_wire: one matrix per launch needs input_dim > 256 (ask "
                 "wsae_wgrad_parts_supported)");
    // the code is sorted once per batch: by the first launch that needs it (the decoder half, or the single launch)
    const bool sort_code = part != WSAE_PART_ENCODER;
    WSAE_REQUIRE(part != WSAE_PART_ENCODER || (ctx->wire_dec_vals == vals && ctx->wire_dec_B == B),
                 "wsae_weight_grads_wire: WSAE_PART_ENCODER must follow WSAE_PART_DECODER of the same batch (it reads the code that call sorted)");
    ctx->wire_dec_vals = part == WSAE_PART_DECODER ? vals : nullptr;
    ctx->wire_dec_B = part == WSAE_PART_DECODER ? B : 0;
    float* out = ctx->wg_slabs;
    if (bf) launch_wgrad<bf16_t>(ctx, p, st, vals, idx, dpre, B, out, x, rows, sort_code);
    else launch_wgrad<float>(ctx, p, st, vals, idx, dpre, B, out, x, rows, sort_code);
    if (part != WSAE_PART_DECODER) ctx->ent_valid = 0;  // (the encoder half of the same batch still reads the sorted code)
    WSAE_LAUNCH_CHECK();

    const float* bpre = params + ctx->off[4];
    const int64_t HD = (int64_t)H * ctx->D;
    WSAE_PROF_BEGIN(ctx, WSAE_K_WGRAD_REDUCE, st);
    if (!wire) {
        const GfOut o = {ctx->off[0], ctx->off[1], ctx->off[2], ctx->off[3], ctx->off[4], 0};
        if (bf) launch_grad_finish<bf16_t, true, float>(ctx, p, st, bpre, grads, o, (const bf16_t*)ctx->We_bf16, part, nullptr);
        else launch_grad_finish<float, false, float>(ctx, p, st, bpre, grads, o, params + ctx->off[0], part, nullptr);
    } else {
        // wire order: [W_dT | W_e | b_e | b_d | b_pre | fired]
        const GfOut o = {HD, 0, 2 * HD, 2 * HD + H, 2 * HD + H + ctx->D, ctx->P};
        const float* fired = part != WSAE_PART_DECODER ? ctx->fired : nullptr;
        if (wire_dtype == WSAE_DT_BF16) {
            if (bf) launch_grad_finish<bf16_t, true, bf16_t>(ctx, p, st, bpre, (bf16_t*)wire, o, (const bf16_t*)ctx->We_bf16, part, fired);
            else launch_grad_finish<float, false, bf16_t>(ctx, p, st, bpre, (bf16_t*)wire, o, params + ctx->off[0], part, fired);
        } else {
            if (bf) launch_grad_finish<bf16_t, true, float>(ctx, p, st, bpre, (float*)wire, o, (const bf16_t*)ctx->We_bf16, part, fired);
            else launch_grad_finish<float, false, float>(ctx, p, st, bpre, (float*)wire, o, params + ctx->off[0], part, fired);
        }
        ctx->n_sq_parts = 0;  // the norm is taken from the summed wire (wsae_grads_unpack_wire)
    }
    WSAE_PROF_END(ctx, WSAE_K_WGRAD_REDUCE, st);
    WSAE_LAUNCH_CHECK();
    return WSAE_OK;
}

extern "C" int wsae_weight_grads(wsae_ctx* ctx, const float* params, const void* x, int32_t x_dtype,
                                 const int32_t* rows, const float* vals, const int32_t* idx, const float* dpre,
                                 int32_t B, float* grads, void* stream) {
    WSAE_REQUIRE(grads, "wsae_weight_grads: null gradient buffer");
    return weight_grads_impl(ctx, params, x, x_dtype, rows, vals, idx, dpre, B, WSAE_PART_ALL, grads, nullptr, 0, (hipStream_t)stream);
}

extern "C" int wsae_wgrad_parts_supported(const wsae_ctx* ctx) { return (ctx && ctx->D > 256) ? 1 : 0; }

extern "C" int wsae_ctx_set_comm_reserve(wsae_ctx* ctx, int32_t n_cus) {
    WSAE_REQUIRE(ctx && n_cus >= 0 && n_cus < ctx->cus, "wsae_ctx_set_comm_reserve: %d compute units of %d", n_cus, ctx ? ctx->cus : 0);
    ctx->comm_reserve = n_cus;
    return WSAE_OK;
}

extern "C" int wsae_weight_grads_wire(wsae_ctx* ctx, const float* params, const void* x, int32_t x_dtype,
                                      const int32_t* rows, const float* vals, const int32_t* idx, const float* dpre,
                                      int32_t B, int32_t part, void* wire, int32_t wire_dtype, void* stream) {
    WSAE_REQUIRE(wire && (wire_dtype == WSAE_DT_F32 || wire_dtype == WSAE_DT_BF16), "wsae_weight_grads_wire: bad wire buffer / dtype");
    WSAE_REQUIRE(part == WSAE_PART_ALL || part == WSAE_PART_DECODER || part == WSAE_PART_ENCODER,
                 "wsae_weight_grads_wire: part must be WSAE_PART_ALL, _DECODER or _ENCODER");
    WSAE_REQUIRE(ctx && (part == WSAE_PART_DECODER || ctx->fired), "wsae_weight_grads_wire: set the fired buffer first (wsae_ctx_set_fired)");
    return weight_grads_impl(ctx, params, x, x_dtype, rows, vals, idx, dpre, B, part, nullptr, wire, wire_dtype, (hipStream_t)stream);
}

// The ReLU SAE's two weight-gradient contractions (wsae_relu.hip, row-major-GEMM flow) on wgrad2d_kernel + the slab
// reduction of grad_finish_kernel (matrix rows and their norm partials only: this module's bias gradients come from its own
// column sums and it has no pre-bias).  hid / dpre: bf16 [B][H]; xb / gb: bf16 [B][D]; grads: the fp32 pack.  Returns the
// number of norm partials left in ctx->part_sq (slots [0, n)), or 0 when the shape is not served (the caller keeps its GEMMs).
int wsae_internal_relu_wgrad(wsae_ctx* ctx, const void* hid, const void* dpre, const void* xb, const void* gb, int B, float* grads,
                             hipStream_t st) {
    const int H = ctx->H, D = ctx->D;
    if (ctx->prec != WSAE_PREC_BF16 || D <= 256 || D % 8 || H % 8 || B % 64 || B < 64) return 0;
    const int ntm = ceil_div(H, W2_M), ntn = ceil_div(D, W2_N);
    const int nchunks = B / 64;
    const int nsplit = pick_nsplit(ctx, ntm * ntn * 2, nchunks, 0, WSAE_WGRAD_MAX_SPLIT);
    wgrad2d_kernel<<<ntm * ntn * 2 * nsplit, W2_THREADS, 2 * W2_STAGE, st>>>((const bf16_t*)hid, (const bf16_t*)dpre, (const bf16_t*)xb,
                                                                             (const bf16_t*)gb, B, H, D, nsplit, ntm, ntn, ctx->wg_slabs,
                                                                             WSAE_WGRAD_MAX_SPLIT);
    const GfOut o = {ctx->off[0], ctx->off[1], ctx->off[2], ctx->off[3], ctx->off[4], 0};
    const int nrb = (int)min((int64_t)WSAE_MAX_PARTIALS - 2, ceil_div64(2 * (int64_t)H, 8));
    grad_finish_kernel<bf16_t, false, float, WSAE_WGRAD_MAX_SPLIT><<<nrb, 256, 0, st>>>(
        ctx->wg_slabs, ctx->dbe_slab, nsplit, nullptr, grads, o, H, D, ctx->part_sq, nrb, 0, 2 * H, 0, (const bf16_t*)nullptr,
        ctx->dbpre_part, 0, ctx->part_dbd, 0, ctx->dbd2, (unsigned long long*)(ctx->counters + 16 + 4 * TICKET_WORDS), 0, nullptr, 0);
    return nrb;
}
