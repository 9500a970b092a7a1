// Sparse decode + MSE residual + dpre on the matrix cores (BF16 mode).
//   reference: TopKSAE.decode / forward / _update_dead_features, model.py:120-181, and the autograd of them
//   (SURVEY.md row A6): g = 2(recon-x)/(BD), dh = g W_d, dpre = dh * 1[v>0].
//
// One wave per batch row.  The k selected decoder rows (bf16 shadow of W_dT, 2 D bytes each) are GATHERED INTO
// LDS by LDS-DMA, 16 bytes per lane (the kernel this replaces pulled 8 bytes per lane into 96 packed VGPRs and ran
// at 0.4 of its L2-gather floor: waves waited 49 % of their cycles and the registers capped occupancy).  A row is
// fetched in pieces of 256 bytes = 128 columns ("sub-steps"): one DMA wave-instruction moves 4 rows x 256 B, so a
// piece of k = 32 rows is 8 instructions and 8 KB of LDS.  Two such slots per wave form a ring over the stream of
// pieces (row r, sub-step s): piece p lands in slot p & 1 and piece p + 2 is issued as soon as piece p has been
// consumed - across row boundaries too (the next row's code is fetched at the top of the current row), so the L2
// latency of a gather sits under the arithmetic of the piece before it.  Both contractions of a piece run on
// v_mfma_f32_16x16x32_bf16 out of the same slot:
//   pass 1  recon[c] = sum_j relu(v_j) W[idx_j][c]:   A = (v_hi, v_lo, v_lo2) as three bf16 rows (their sum is v
//           exactly, so the product is exact to fp32 accumulation), B = the slot read TRANSPOSED with
//           ds_read_b64_tr_b16 (k = feature slot, n = column); 8 MFMAs per piece;
//   (residual, loss, g = 2 r / (B D), bf16(g) -> a 2 D byte LDS row)
//   pass 2  dh_j += sum_c bf16(g[c]) W[idx_j][c]:     A = the slot rows as they lie (k = column), B = bf16(g)
//           broadcast; 8 MFMAs per piece, accumulators carried across the pieces of a row.
// LDS image of a slot: [row j][16 chunks of 16 B], chunk c of row j at position c ^ sw(j), sw(j) = j ^ (j&1 ? 12 : 0)
// (low four bits): conflict-free for the ds_read_b128 fragments of pass 2 AND the transposed reads of pass 1
// (checked exhaustively over the lane groups of MI355X_MICROARCH.md, LDS; profiles/tools/probe_decode_frag.hip runs
// both passes of one piece against a host sum).  The DMA writes lane-linear, so the swizzle is applied to the
// per-lane SOURCE address (cdna guide, rule 21).
//
// Folding the per-row TopK into this kernel (its VALU work under the gather latency) was built and measured: with
// the two waves per SIMD that 79 KB of LDS per workgroup allow, the TopK's own dependent HBM round trips are as
// exposed as the gathers (85.6 us fused against 24.7 + 53.2 us separate at cfg 2), so the TopK stays its own launch.
#include "wsae_common.h"
#include "wsae_decode_epilogue.h"
#include "wsae_mfma.h"

// Diagnostic build only (-DWSAE_DM_STAMPS, never the product library; profiles/tools/stamps_decode.py): per-phase
// cycle sums of the row loop.
#ifdef WSAE_DM_STAMPS
__device__ unsigned long long dm_stamp_sum[8];
#define DM_T(i)                                                                                     \
    {                                                                                               \
        __builtin_amdgcn_sched_barrier(0);                                                          \
        unsigned long long t_;                                                                      \
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_)::"memory");                  \
        __builtin_amdgcn_sched_barrier(0);                                                          \
        dm_acc[i] += t_ - dm_last;                                                                  \
        dm_last = t_;                                                                               \
    }
extern "C" int wsae_debug_stamps(double* out, int reset) {
    unsigned long long h[8];
    if (hipMemcpyFromSymbol(h, HIP_SYMBOL(dm_stamp_sum), sizeof(h)) != hipSuccess) return -1;
    for (int i = 0; i < 8; ++i) out[i] = (double)h[i];
    if (reset) {
        unsigned long long z[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        (void)hipMemcpyToSymbol(HIP_SYMBOL(dm_stamp_sum), z, sizeof(z));
    }
    return 0;
}
#else
// Product build: the phase boundaries stay scheduling fences.  Without them hipcc interleaves the phases of a row
// (gather issue, code fetch, the two MFMA passes) so that their waits serialise: 76 us against 25 us at cfg 2.
#define DM_T(i) __builtin_amdgcn_sched_barrier(0);
#endif

typedef __attribute__((address_space(3))) bf16x4 lds_bf16x4;

__device__ __forceinline__ int dm_sw(int j) { return (j & 15) ^ ((j & 1) ? 12 : 0); }

// 8 k-values (two blocks of 4 LDS rows) x this lane's column, as an MFMA B fragment
__device__ __forceinline__ bf16x8 tr_frag(const char* a0, const char* a1) {
    // (the v4i16 form + per-element bit casts to bf16 miscompiles on ROCm 7.2: every element became element 0)
    const bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4*)a0);
    const bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4*)a1);
    return __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
}

// One piece = N LDS-DMA wave-instructions of 1 KB to consecutive LDS addresses from `lds` on: lane sources are
// base + off[e] bytes (SGPR base + 32-bit VGPR offset form: no 64-bit address arithmetic per instruction), M0 is
// saved once, advanced by 1 KB per instruction and restored once.  Same caveats as glds16 (wsae_mfma.h): the loads
// are invisible to hipcc's wait counting, the caller waits with vm_wait.
template <int N>
__device__ __forceinline__ void dma_piece(const void* base, const uint32_t (&off)[N], uint32_t lds) {
    static_assert(N == 8 || N == 16, "a piece is 32 or 64 rows");
    const uint32_t l0 = __builtin_amdgcn_readfirstlane(lds);
    uint32_t keep;
#define DP1(i) "global_load_lds_dwordx4 %" #i ", %[b]\n\ts_add_u32 m0, m0, 0x400\n\ts_nop 0\n\t"
    if constexpr (N == 8) {
        asm volatile("s_mov_b32 %[k], m0\n\ts_mov_b32 m0, %[l]\n\ts_nop 0\n\t" DP1(2) DP1(3) DP1(4) DP1(5) DP1(6) DP1(7) DP1(8) DP1(9)
                     "s_mov_b32 m0, %[k]"
                     : [k] "=&s"(keep)
                     : [l] "s"(l0), "v"(off[0]), "v"(off[1]), "v"(off[2]), "v"(off[3]), "v"(off[4]), "v"(off[5]), "v"(off[6]),
                       "v"(off[7]), [b] "s"(base)
                     : "memory");
    } else {
        asm volatile("s_mov_b32 %[k], m0\n\ts_mov_b32 m0, %[l]\n\ts_nop 0\n\t" DP1(2) DP1(3) DP1(4) DP1(5) DP1(6) DP1(7) DP1(8) DP1(9)
                     DP1(10) DP1(11) DP1(12) DP1(13) DP1(14) DP1(15) DP1(16) DP1(17) "s_mov_b32 m0, %[k]"
                     : [k] "=&s"(keep)
                     : [l] "s"(l0), "v"(off[0]), "v"(off[1]), "v"(off[2]), "v"(off[3]), "v"(off[4]), "v"(off[5]), "v"(off[6]),
                       "v"(off[7]), "v"(off[8]), "v"(off[9]), "v"(off[10]), "v"(off[11]), "v"(off[12]), "v"(off[13]),
                       "v"(off[14]), "v"(off[15]), [b] "s"(base)
                     : "memory");
    }
#undef DP1
}

template <int N>
__device__ __forceinline__ void vm_wait() {
    if constexpr (N == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    else if constexpr (N == 8) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    else {
        static_assert(N == 16, "vm_wait: one piece of 32 or 64 rows in flight");
        asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
    }
}

// bytes of LDS one wave needs: two slots, two code buffers (values | indices), the bf16 g row, the 128-float
// staging row through which a piece's sums go from the MFMA layout (16 lanes x 8 tiles) to two columns per lane
__host__ __device__ constexpr int dm_wave_bytes(int KS, int D) {
    return 2 * (32 * KS) * 256 + 2 * 2 * (32 * KS) * 4 + D * 2 + 128 * 4;
}

// KS = MFMA k-steps of 32 feature slots (k <= 32: 1, k <= 64: 2), NSUB = D / 128, XDT = element type of x.
// Two workgroups per CU (two waves per SIMD, <= 256 VGPRs) while their LDS allows it: k <= 32 and D <= 384.
// NW = 4: one resident round of 256-thread blocks, rows dealt round-robin over all waves of the grid.
// NW = 8 ("chunked", backward only): one 512-thread block per 64-row chunk of the batch - the chunk the weight-gradient
// contraction walks (wsae_wgrad.hip) - each wave takes 8 of its rows; when the rows are done the block counting-sorts the
// chunk's K * 64 code entries by feature tile (what bucket_sort_kernel would do in a launch of its own: ent_pos / ent_hid /
// ent_dpre / ent_off), reading back the dpre values its own waves have just stored (same CU, behind a barrier).
template <int KS, int NSUB, int XDT, bool BWD, int NW>
__global__ void __launch_bounds__(64 * NW, (KS == 1 && NSUB <= 3) ? 2 : 1)
decode_mfma_kernel(const bf16_t* __restrict__ WdT, const float* __restrict__ bd, const float* __restrict__ bpre,
                   const void* __restrict__ x, const int32_t* __restrict__ rows, const float* __restrict__ vals,
                   const int32_t* __restrict__ idx, int B, int K, float* __restrict__ recon_out, float* __restrict__ dpre,
                   float* __restrict__ g32_out, bf16_t* __restrict__ gb_out, int64_t* __restrict__ last_activated,
                   float* __restrict__ fired, const int64_t* __restrict__ step_count, float* __restrict__ part_loss,
                   float* __restrict__ part_l0, float* __restrict__ part_dbd, int32_t* __restrict__ ticket,
                   wsae_stats* __restrict__ stats, int loss_cols, int ntiles, int tw, uint32_t* __restrict__ ent_pos,
                   bf16_t* __restrict__ ent_hid, bf16_t* __restrict__ ent_dpre, int32_t* __restrict__ ent_off) {
    static_assert(NW == 4 || (NW == 8 && BWD), "the chunked form exists for the training backward only");
    constexpr int D = 128 * NSUB;
    constexpr int KP = 32 * KS;      // feature slots per row (K padded)
    constexpr int E = KP / 4;        // DMA wave-instructions per piece
    constexpr int SLOT = KP * 256;   // bytes
    constexpr int WAVE_BYTES = dm_wave_bytes(KS, D);
    static_assert(WAVE_BYTES % 16 == 0, "per-wave LDS regions stay 16-byte aligned");
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int n = lane & 15, grp = lane >> 4;
    // which of the three bf16 pieces of a value this lane's A row carries (rows 0..2 of the 16-row tile; the rest zero)
    const uint32_t piece_mask[3] = {n == 0 ? 0xFFFFFFFFu : 0u, n == 1 ? 0xFFFFFFFFu : 0u, n == 2 ? 0xFFFFFFFFu : 0u};
    char* wbase = smem + wave * WAVE_BYTES;
    char* slot0 = wbase;
    char* code = wbase + 2 * SLOT;                            // [2 buffers][vals KP f32 | idx KP i32]
    bf16_t* grow = (bf16_t*)(code + 2 * 2 * KP * 4);          // [D] bf16(g) of the current row
    float* stage = (float*)(grow + D);                        // [128] sums of the current piece
    float* red = (float*)(smem + NW * WAVE_BYTES);            // [8]
    int* flag_s = (int*)(red + 8);                            // [4]
    float* dbd_s = (float*)smem;                              // [NW][D] at the end (aliases the slots)
    const uint32_t slot_lds = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) char*)slot0;

    for (int i = lane; i < 2 * 2 * KP; i += 64) ((int*)code)[i] = 0;  // padded slots: value 0, feature 0
    // after the MFMAs of a piece every lane owns TWO columns, 128 s + 2 lane + {0, 1}: bias, x, residual, g, the
    // column sums of g and every store are per-lane pairs (8-byte accesses, 512 contiguous bytes per wave)
    f32x2 bsum2[NSUB];
#pragma unroll
    for (int s = 0; s < NSUB; ++s) {
        const float2 a = *(const float2*)(bd + 128 * s + 2 * lane), p = *(const float2*)(bpre + 128 * s + 2 * lane);
        bsum2[s] = f32x2{a.x + p.x, a.y + p.y};
    }

    const float scale = 2.0f / ((float)B * (float)loss_cols);
    const int64_t step = (last_activated && step_count) ? *step_count : 0;
    constexpr bool x_bf16 = XDT == WSAE_DT_BF16;
    // lane-constant LDS offsets (the slot and the tile / k-step enter as uniform or compile-time terms):
    //   pass 1, B fragment of tile t: block rows 32 q + 8 grp + {0..3} / {4..7}; lane 4 i + p of a 16-lane group addresses
    //   block row i, columns 16 t + 4 p .. + 3 = chunk 2 t + (p >> 1), byte 8 (p & 1): offset = row * 256 + 8 (p & 1) +
    //   ((32 t) ^ (16 (sw(row) ^ (p >> 1))))   [2 t is even, so the chunk's low bit folds into the lane constant]
    //   pass 2, A fragment of k-step kk: row 16 m + n, chunk 4 kk + grp: offset = row * 256 + ((64 kk) ^ (16 (sw(row) ^ grp)))
    int tr_off[KS][2], tr_swz[2], a2_off[2 * KS], a2_swz[2 * KS];
    {
        const int i = n >> 2, p = n & 3;
#pragma unroll
        for (int h2 = 0; h2 < 2; ++h2) {
            const int jr = 8 * grp + i + 4 * h2;  // (row 32 q + jr has the same swizzle for every q)
            tr_swz[h2] = (dm_sw(jr) ^ (p >> 1)) << 4;
#pragma unroll
            for (int q = 0; q < KS; ++q) tr_off[q][h2] = (32 * q + jr) * 256 + 8 * (p & 1);
        }
#pragma unroll
        for (int m = 0; m < 2 * KS; ++m) {
            a2_off[m] = (16 * m + n) * 256;
            a2_swz[m] = (dm_sw(16 * m + n) ^ grp) << 4;
        }
    }
    float loss_acc = 0.f;
    int l0_acc = 0;
    f32x2 dbd[NSUB];
#pragma unroll
    for (int s = 0; s < NSUB; ++s) dbd[s] = f32x2{0.f, 0.f};

    // the code (values, indices) of batch row bb -> code buffer `cb`
    auto get_code = [&](int bb, int cb) {
        float* vs = (float*)(code + cb * 2 * KP * 4);
        int32_t* is = (int32_t*)(vs + KP);
        if (lane < K) {
            vs[lane] = vals[(int64_t)bb * K + lane];
            is[lane] = idx[(int64_t)bb * K + lane];
        }
    };
    // gather sources of a row: DMA instruction e of a piece covers slot rows 4 e + grp; this lane fills position n
    // (kept as 32-bit element offsets into W_dT: H * D < 2^31 for every supported shape)
    auto sources = [&](int cb, uint32_t (&rp)[E]) {
        const int32_t* is = (const int32_t*)(code + cb * 2 * KP * 4) + KP;
#pragma unroll
        for (int e = 0; e < E; ++e) {
            const int j = 4 * e + grp;
            rp[e] = (uint32_t)is[j] * (uint32_t)D + 8u * (uint32_t)(n ^ dm_sw(j));
        }
    };
    auto issue = [&](const uint32_t (&rp)[E], int s, int slot) {  // every read of that slot has been waited for
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        uint32_t off[E];
#pragma unroll
        for (int e = 0; e < E; ++e) off[e] = 2u * rp[e] + 256u * (uint32_t)s;  // bytes
        dma_piece<E>(WdT, off, slot_lds + slot * SLOT);
    };

    const int b_first = NW == 8 ? blockIdx.x * 64 + wave : blockIdx.x * 4 + wave;
    const int b_step = NW == 8 ? 8 : gridDim.x * 4;
    const int b_end = NW == 8 ? min(B, (int)blockIdx.x * 64 + 64) : B;  // chunked: this block's rows end with its chunk
    int buf = 0, par = 0;  // code buffer of the current row; slot of its first piece
    int64_t src_next = 0;  // source row of the next batch row (ring gather), fetched one row ahead
    uint32_t rowp[E];
#ifdef WSAE_DM_STAMPS
    unsigned long long dm_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0}, dm_last = __builtin_amdgcn_s_memtime();
#endif
    if (b_first < b_end) {
        src_next = rows ? (int64_t)rows[b_first] : (int64_t)b_first;
        get_code(b_first, 0);
        sources(0, rowp);
        issue(rowp, 0, 0);
        if (NSUB > 1) issue(rowp, 1, 1);
    }
    DM_T(0)
    for (int b = b_first; b < b_end; b += b_step) {
        const float* vs = (const float*)(code + buf * 2 * KP * 4);
        const int32_t* is = (const int32_t*)(vs + KP);
        const bool more = b + b_step < b_end;
        // ---- dead-feature clock, l0 (model.py:148, :178-181): lane j looks at feature slot j ----
        {
            const float v = lane < KP ? vs[lane < KP ? lane : 0] : 0.f;
            const bool on = lane < K && v > 0.f;
            l0_acc += __popcll(__ballot(on));
            if (on && last_activated) {
                const int f = is[lane];
                last_activated[f] = step;  // same value from every writer
                if (fired) fired[f] = 1.f;
            }
        }
        // ---- the row of x: this lane's two columns of every piece (one 4- or 8-byte load per piece, 256 / 512
        // contiguous bytes per wave), kept raw until the residual uses them
        const int64_t src = src_next;
        if (more) src_next = rows ? (int64_t)rows[b + b_step] : (int64_t)(b + b_step);
        uint32_t xraw[NSUB][x_bf16 ? 1 : 2];
#pragma unroll
        for (int s = 0; s < NSUB; ++s) {
            if constexpr (x_bf16) {
                xraw[s][0] = *(const uint32_t*)((const uint16_t*)x + src * D + 128 * s + 2 * lane);
            } else {
                const uint2 t2 = *(const uint2*)((const uint32_t*)x + src * D + 128 * s + 2 * lane);
                xraw[s][0] = t2.x;
                xraw[s][1] = t2.y;
            }
        }
        // ---- A fragments of pass 1: rows 0..2 = the three bf16 pieces of relu(v), k = feature slot ----
        // (branch-free: the three pieces are computed in packed pairs by every lane and picked with lane-constant bit
        // masks - written as n == 0 ? hi : n == 1 ? lo : .. hipcc nested two divergent branches around every element)
        bf16x8 a1[KS];
#pragma unroll
        for (int q = 0; q < KS; ++q) {
            const float4 v0 = *(const float4*)(vs + 32 * q + 8 * grp), v1 = *(const float4*)(vs + 32 * q + 8 * grp + 4);
            const float vv[8] = {v0.x, v0.y, v0.z, v0.w, v1.x, v1.y, v1.z, v1.w};
            uint32_t w4[4];
#pragma unroll
            for (int p2 = 0; p2 < 4; ++p2) {
                // negative winners decode as zero (model.py:116): clamp to [0, inf) in one instruction
                const float a = __builtin_amdgcn_fmed3f(vv[2 * p2], 0.f, INFINITY);
                const float b = __builtin_amdgcn_fmed3f(vv[2 * p2 + 1], 0.f, INFINITY);
                bf16x2 h2;
                h2[0] = (bf16_t)a; h2[1] = (bf16_t)b;
                const uint32_t hb = __builtin_bit_cast(uint32_t, h2);
                const float ra = a - __uint_as_float(hb << 16), rb = b - __uint_as_float(hb & 0xFFFF0000u);
                bf16x2 l2;
                l2[0] = (bf16_t)ra; l2[1] = (bf16_t)rb;
                const uint32_t lb = __builtin_bit_cast(uint32_t, l2);
                const float sa = ra - __uint_as_float(lb << 16), sb = rb - __uint_as_float(lb & 0xFFFF0000u);
                bf16x2 t2;
                t2[0] = (bf16_t)sa; t2[1] = (bf16_t)sb;
                const uint32_t tb = __builtin_bit_cast(uint32_t, t2);
                w4[p2] = (hb & piece_mask[0]) | (lb & piece_mask[1]) | (tb & piece_mask[2]);
            }
            typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
            const u32x4 wv = {w4[0], w4[1], w4[2], w4[3]};
            a1[q] = __builtin_bit_cast(bf16x8, wv);
        }
        DM_T(1)
        // ---- the next row's code and gather sources (its first pieces are issued from inside this row's loop) ----
        uint32_t rowp_n[E];
        if (more) {
            get_code(b + b_step, buf ^ 1);
            sources(buf ^ 1, rowp_n);
        } else {
#pragma unroll
            for (int e = 0; e < E; ++e) rowp_n[e] = rowp[e];
        }
        if (NSUB == 1 && more) issue(rowp_n, 0, par ^ 1);  // (one piece per row: the other slot is free already)
        DM_T(2)
        if (NSUB == 1 && more) vm_wait<E>(); else vm_wait<0>();  // this row's first pieces, its x row, the next code: landed
        DM_T(3)

        f32x4 acc2[2 * KS];
#pragma unroll
        for (int m = 0; m < 2 * KS; ++m) acc2[m] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int s = 0; s < NSUB; ++s) {
            // piece (row, s) sits in slot (par + s) & 1; the only younger gather is the piece issued after the previous
            // one was consumed: this row's s + 1, or the next row's first
            if (s >= 2 || (s == 1 && NSUB == 2)) {
                if (s + 1 < NSUB || more) vm_wait<E>(); else vm_wait<0>();
            }
            DM_T(4)
            const int sl = (par + s) & 1;
            const char* slot = slot0 + sl * SLOT;
            // pass 1: 8 column tiles of 16.  Lanes 0..15 of a tile's accumulator hold its rows 0..2 (the hi / lo / lo2
            // parts of the sum); they go through the staging row so that the arithmetic after them runs once per piece on
            // two columns per lane with every lane busy, instead of eight times on a quarter of the lanes.
#pragma unroll
            for (int th = 0; th < 2; ++th) {
                f32x4 acc[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const int t = 4 * th + u;
                    acc[u] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                    for (int q = 0; q < KS; ++q) {
                        const bf16x8 bf = tr_frag(slot + tr_off[q][0] + ((32 * t) ^ tr_swz[0]), slot + tr_off[q][1] + ((32 * t) ^ tr_swz[1]));
                        acc[u] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a1[q], bf, acc[u], 0, 0, 0);
                    }
                }
                if (lane < 16) {
#pragma unroll
                    for (int u = 0; u < 4; ++u) stage[16 * (4 * th + u) + n] = (acc[u][0] + acc[u][1]) + acc[u][2];
                }
            }
            {
                const float2 sm = *(const float2*)(stage + 2 * lane);
                const float rec0 = sm.x + bsum2[s][0], rec1 = sm.y + bsum2[s][1];  // + b_d + b_pre
                float x0, x1;
                if constexpr (x_bf16) {
                    x0 = __uint_as_float(xraw[s][0] << 16);
                    x1 = __uint_as_float(xraw[s][0] & 0xFFFF0000u);
                } else {
                    x0 = __uint_as_float(xraw[s][0]);
                    x1 = __uint_as_float(xraw[s][1]);
                }
                const float r0 = rec0 - x0, r1 = rec1 - x1;
                const float g0 = r0 * scale, g1 = r1 * scale;
                loss_acc = fmaf(r0, r0, fmaf(r1, r1, loss_acc));
                dbd[s][0] += g0;
                dbd[s][1] += g1;
                bf16x2 gb2;
                gb2[0] = (bf16_t)g0;
                gb2[1] = (bf16_t)g1;
                const int col = 128 * s + 2 * lane;
                *(bf16x2*)(grow + col) = gb2;
                if (recon_out) *(float2*)(recon_out + (int64_t)b * D + col) = make_float2(rec0, rec1);
                if (BWD) {
                    *(bf16x2*)(gb_out + (int64_t)b * D + col) = gb2;
                    if (g32_out) *(float2*)(g32_out + (int64_t)b * D + col) = make_float2(g0, g1);
                }
            }
            DM_T(5)
            // pass 2: dh_j += sum over this piece's 128 columns of bf16(g) * W[idx_j]
            if (BWD) {
#pragma unroll
                for (int kk = 0; kk < 4; ++kk) {
                    const bf16x8 gf = *(const bf16x8*)(grow + 128 * s + 32 * kk + 8 * grp);
#pragma unroll
                    for (int m = 0; m < 2 * KS; ++m) {
                        const bf16x8 af = *(const bf16x8*)(slot + a2_off[m] + ((64 * kk) ^ a2_swz[m]));
                        acc2[m] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af, gf, acc2[m], 0, 0, 0);
                    }
                }
            }
            // the slot is free: piece p + 2 of the stream goes there
            if (NSUB >= 2) {
                if (s + 2 < NSUB) issue(rowp, s + 2, sl);
                else if (more) issue(rowp_n, s + 2 - NSUB, sl);
            }
            DM_T(6)
        }
        // ---- outputs of the row ----
        if (BWD) {
            // every column of the pass-2 tile holds the same 16 dots; lane (grp, n < 4) writes slot 16 m + 4 grp + n
#pragma unroll
            for (int m = 0; m < 2 * KS; ++m) {
                const int j = 16 * m + 4 * grp + (n & 3);
                const float d0 = (n & 2) ? ((n & 1) ? acc2[m][3] : acc2[m][2]) : ((n & 1) ? acc2[m][1] : acc2[m][0]);
                float out = vs[j] > 0.f ? d0 : 0.f;
                out = (float)(bf16_t)out;  // what the weight-gradient MFMAs are fed (oracle "amp": dpre rounded once, here)
                if (n < 4 && j < K) dpre[(int64_t)b * K + j] = out;
            }
        }
        par = (par + NSUB) & 1;
        buf ^= 1;
#pragma unroll
        for (int e = 0; e < E; ++e) rowp[e] = rowp_n[e];
        DM_T(7)
    }
#ifdef WSAE_DM_STAMPS
    if (lane == 0)
        for (int i = 0; i < 8; ++i) atomicAdd(&dm_stamp_sum[i], dm_acc[i]);
#endif

    vm_wait<0>();
    __syncthreads();  // every wave is done with its slots: they become the per-wave column sums of g
    if constexpr (NW == 8) {
        // ---- counting sort of this chunk's code by feature tile (the bucket launch of wsae_wgrad.hip, fused) ----
        // entries are fetched once, up front; the dpre values are this block's own stores (same CU's L1, behind the barrier)
        int* cnt = (int*)smem;              // [ntiles <= 512]   (the slots are free now)
        int* cur = cnt + 512;
        constexpr int EB = 8;               // 512 threads x 8 >= 64 rows x K <= 64
        const int tid = threadIdx.x;
        const int c0 = (int)blockIdx.x * 64;
        const int nent = (b_end - c0) * K;
        const int64_t base = (int64_t)c0 * K;
        int f[EB], tl[EB];
        float v[EB], dp[EB];
#pragma unroll
        for (int j = 0; j < EB; ++j) {
            const int e = min(tid + 512 * j, nent - 1);
            f[j] = idx[base + e];
            v[j] = vals[base + e];
            dp[j] = dpre[base + e];
        }
        for (int t = tid; t < ntiles; t += 512) cnt[t] = 0;
        __syncthreads();
#pragma unroll
        for (int j = 0; j < EB; ++j) {
            tl[j] = f[j] / tw;
            if (tid + 512 * j < nent) atomicAdd(&cnt[tl[j]], 1);
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
                    const int nn = __shfl_up(incl, o, 64);
                    if (tid >= o) incl += nn;
                }
                if (t < ntiles) {
                    cur[t] = carry + incl - c;
                    ent_off[(int64_t)blockIdx.x * (ntiles + 1) + t] = (int)base + carry + incl - c;
                }
                carry += __shfl(incl, 63, 64);
            }
            if (tid == 0) ent_off[(int64_t)blockIdx.x * (ntiles + 1) + ntiles] = (int)base + carry;
        }
        __syncthreads();
#pragma unroll
        for (int j = 0; j < EB; ++j) {
            const int e = tid + 512 * j;
            if (e >= nent) continue;
            const int p = atomicAdd(&cur[tl[j]], 1);
            ent_pos[base + p] = ((uint32_t)(f[j] - tl[j] * tw) << 16) | (uint32_t)(e / K);
            ent_hid[base + p] = (bf16_t)(v[j] > 0.f ? v[j] : 0.f);
            ent_dpre[base + p] = (bf16_t)dp[j];
        }
        __syncthreads();  // cnt / cur alias the column-sum rows written next
    }
    if (BWD) {
#pragma unroll
        for (int s = 0; s < NSUB; ++s) *(float2*)(dbd_s + wave * D + 128 * s + 2 * lane) = make_float2(dbd[s][0], dbd[s][1]);
    }
    __syncthreads();
    decode_block_epilogue<BWD, NW>(loss_acc, l0_acc, dbd_s, D, B, loss_cols, red, flag_s, part_loss, part_l0, part_dbd, ticket, stats);
}

// ------------------------------------------------------------------------------------------------
// workgroups of `kernel` one CU holds (registers and LDS both count); the grid is one resident round of them
template <typename F>
static int resident_per_cu(F kernel, size_t sh) {
    int n = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, kernel, 256, sh) != hipSuccess || n < 1) n = 1;
    return n;
}

// tiling of the weight-gradient contraction that will consume the bucketed code (wsae_wgrad.hip)
void wsae_internal_wgrad_tiling(const wsae_ctx* c, int* tile_width, int* ntiles);

template <int KS, int NSUB, int XDT>
static int dm_launch(wsae_ctx* c, const float* params, const void* x, const int32_t* rows, const float* vals,
                     const int32_t* idx, int B, float* recon, int want_bwd, float* dpre, float* g32, int64_t* last_activated,
                     const int64_t* step_count, wsae_stats* stats, hipStream_t st) {
    const size_t sh = 4 * (size_t)dm_wave_bytes(KS, 128 * NSUB) + 12 * 4;
    static const int per_cu[2] = {resident_per_cu(decode_mfma_kernel<KS, NSUB, XDT, false, 4>, sh),
                                  resident_per_cu(decode_mfma_kernel<KS, NSUB, XDT, true, 4>, sh)};
    int tw = 0, ntiles = 0;
    wsae_internal_wgrad_tiling(c, &tw, &ntiles);
#define DM_ARGS c->WdT_bf16, params + c->off[3], params + c->off[4], x, rows, vals, idx, B, c->K, recon, dpre, g32, \
                c->gb, last_activated, c->fired, step_count, c->part_loss, c->part_l0, c->part_dbd,                       \
                c->counters + 16 + 2 * TICKET_WORDS, stats, c->loss_cols, ntiles, tw, c->ent_pos, (bf16_t*)c->ent_hid,   \
                (bf16_t*)c->ent_dpre, c->ent_off
    // chunked form: training backward on batches that give every CU at least one 64-row chunk (below that the one-round
    // form with rows dealt over all waves keeps more CUs busy) and at most one partial slot per chunk
    const int nchunks = ceil_div(B, 64);
    c->ent_valid = 0;
    if constexpr (KS == 1 && NSUB <= 3) {
        if (want_bwd && dpre && NSUB == 3 && nchunks >= c->cus && nchunks <= WSAE_MAX_PARTIALS && ntiles <= 512 &&
            c->K * 64 <= 4096) {  // (D > 256: the contraction that reads the code row-major, wsae_wgrad.hip)
            const size_t sh8 = 8 * (size_t)dm_wave_bytes(KS, 128 * NSUB) + 12 * 4;
            decode_mfma_kernel<KS, NSUB, XDT, true, 8><<<nchunks, 512, sh8, st>>>(DM_ARGS);
            c->ent_valid = 1;
            c->ent_vals = vals;
            c->ent_B = B;
            return nchunks;
        }
    }
    const int nblk = min(ceil_div(B, 4), min(per_cu[want_bwd ? 1 : 0] * c->cus, WSAE_MAX_PARTIALS));
    if (want_bwd) decode_mfma_kernel<KS, NSUB, XDT, true, 4><<<nblk, 256, sh, st>>>(DM_ARGS);
    else decode_mfma_kernel<KS, NSUB, XDT, false, 4><<<nblk, 256, sh, st>>>(DM_ARGS);
#undef DM_ARGS
    return nblk;
}

// does the MFMA decode kernel serve this ctx?  (BF16 mode, D = 128 / 256 / 384 / 768 / 1280, k <= 64; k > 32 only
// from D = 384 up)
bool wsae_internal_decode_mfma_ok(const wsae_ctx* c) {
    if (c->prec != WSAE_PREC_BF16 || c->D % 128 || c->K > 64) return false;
    const int ns = c->D / 128;
    if (c->K > 32) return ns == 3 || ns == 6 || ns == 10;
    return ns == 1 || ns == 2 || ns == 3 || ns == 6 || ns == 10;
}

int wsae_internal_decode_mfma(wsae_ctx* c, const float* params, const void* x, int x_dtype, const int32_t* rows,
                              const float* vals, const int32_t* idx, int B, float* recon, int want_bwd, float* dpre,
                              int want_g32, int64_t* last_activated, const int64_t* step_count, wsae_stats* stats,
                              hipStream_t st) {
    const int ns = c->D / 128, ks = c->K <= 32 ? 1 : 2;
    int nblk = 0;
    float* g32 = want_g32 ? c->g : nullptr;
#define DM_CASE(KS_, NS_)                                                                                              \
    nblk = x_dtype == WSAE_DT_BF16                                                                                     \
               ? dm_launch<KS_, NS_, WSAE_DT_BF16>(c, params, x, rows, vals, idx, B, recon, want_bwd, dpre, g32,        \
                                                   last_activated, step_count, stats, st)                               \
               : dm_launch<KS_, NS_, WSAE_DT_F32>(c, params, x, rows, vals, idx, B, recon, want_bwd, dpre, g32,         \
                                                  last_activated, step_count, stats, st);                               \
    break
    WSAE_PROF_BEGIN(c, WSAE_K_DECODE, st);
    if (ks == 1) {
        switch (ns) {
            case 1: DM_CASE(1, 1);
            case 2: DM_CASE(1, 2);
            case 3: DM_CASE(1, 3);
            case 6: DM_CASE(1, 6);
            default: DM_CASE(1, 10);
        }
    } else {
        switch (ns) {
            case 3: DM_CASE(2, 3);
            case 6: DM_CASE(2, 6);
            default: DM_CASE(2, 10);
        }
    }
    WSAE_PROF_END(c, WSAE_K_DECODE, st);
#undef DM_CASE
    WSAE_LAUNCH_CHECK();
    c->n_dec_blocks = nblk;
    c->g_is_bf16 = 1;
    c->g32_valid = want_g32 ? 1 : 0;
    return WSAE_OK;
}
