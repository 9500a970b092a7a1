// Encoder GEMM + TopK without the [B,H] pre-activation matrix in HBM (BF16 mode, bf16 batch, D <= 384).
//   reference: pre = encoder(x - b_pre); values, indices = torch.topk(pre, k)   src/whisper_sae/sae/model.py:108-114
//
// encode_filter_kernel   a workgroup owns 128 batch rows x one part of the features (H / NQ of them, in 32-feature
//                        slabs).  Each of its 4 waves keeps its 32 rows of x in registers as MFMA A fragments for the
//                        whole kernel; the slabs of W_e stream through a 3-deep LDS ring by LDS-DMA (a 32 x D bf16
//                        slab, XOR-swizzled on the source address) and are the B operand: a lane holds ONE feature of
//                        the slab (lane & 31) for 16 of the wave's rows - register r of the lower / upper half-wave
//                        is row m(r) / m(r) + 4.  Nothing dense is written.  The first 4 slabs stay in registers and
//                        calibrate a threshold T per row: the smallest of the maxima of its eight 16-value groups
//                        (4 lanes x 4 slabs), near the row's 14 % quantile and above its k-th largest value only if
//                        all eight groups hold one of the row's k largest (3e-7 at k = 32 of 3072).  From then on
//                        every value >= T is filed: one compare per register gives the 64-lane mask of a row PAIR,
//                        v_mbcnt turns it into positions behind the pair's counter (an SGPR) and the passing lanes
//                        append (value, feature, half) to the pair's list with one store - consecutive entries, so
//                        the store is one or two short segments instead of 64 scattered rows.  A slab's values are
//                        filed one slab late, after the next barrier, so that the counted vmcnt that retires the DMA
//                        only meets stores a whole MFMA phase old.
// select_filtered_kernel one wave per row pair: with Tmax = the largest threshold any part used for the row, every
//                        element >= Tmax is on file; if at least K of them are, the K largest are the row's TopK,
//                        exactly (threshold from the lane maxima, compaction, one 64-key sort: wsae_topk.h).
//                        Otherwise - or when a list overflowed - the row goes to the fallback list.
// encode_fallback_kernel recomputes listed rows densely (one workgroup per 256 features) and runs the exact TopK on
//                        them; it reads the count from device memory and normally finds zero.
#include "wsae_common.h"
#include "wsae_mfma.h"
#include "wsae_topk.h"

#include <type_traits>

#define FZ_ROWS 128
#define FZ_SAMPLE 4
#define FZ_RING 3

namespace {

template <int N>
__device__ __forceinline__ void vm_wait() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

// 16 bytes from LDS byte address addr + imm (asm: see slab_mfma)
__device__ __forceinline__ bf16x8 lds_read16(uint32_t addr, int imm) {  // imm: a constant once the caller is unrolled
    bf16x8 r;
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(r) : "v"(addr), "n"(imm));
    return r;
}
// wait until at most `newer` LDS operations are outstanding, then release w to its consumer
__device__ __forceinline__ void lds_wait_for(bf16x8& w, int newer) {
    switch (newer) {
        case 0: asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(w)); break;
        case 1: asm volatile("s_waitcnt lgkmcnt(1)" : "+v"(w)); break;
        case 2: asm volatile("s_waitcnt lgkmcnt(2)" : "+v"(w)); break;
        case 3: asm volatile("s_waitcnt lgkmcnt(3)" : "+v"(w)); break;
        case 4: asm volatile("s_waitcnt lgkmcnt(4)" : "+v"(w)); break;
        case 5: asm volatile("s_waitcnt lgkmcnt(5)" : "+v"(w)); break;
        default: asm volatile("s_waitcnt lgkmcnt(6)" : "+v"(w)); break;
    }
}

// batch row (inside the wave's 32) of accumulator register r in the lower half-wave; the upper half holds row + 4
__host__ __device__ __forceinline__ constexpr int fz_row_of(int r) { return (r & 3) + 8 * (r >> 2); }

// one 1 KiB LDS-DMA piece: lane l fetches 16 bytes at base + voff into LDS byte lds_addr + 16 l.  SGPR base + 32-bit
// lane offset (no 64-bit lane address to compute); M0 is compiler-reserved, saved and restored.
__device__ __forceinline__ void glds16s(const void* base, uint32_t voff, uint32_t lds_addr) {
    uint32_t keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep)
                 : "v"(voff), "s"(base), "s"(lds_addr)
                 : "memory");
}

template <int D>
__global__ void __launch_bounds__(256, 2)
encode_filter_kernel(const bf16_t* __restrict__ x, const int32_t* __restrict__ rows, const bf16_t* __restrict__ W,
                     const float* __restrict__ bias, int B, int NQ, int SP, uint2* __restrict__ cand,
                     int32_t* __restrict__ cand_cnt, float* __restrict__ cand_T, int32_t* __restrict__ n_fail,
                     int64_t* __restrict__ step_count) {
    constexpr int KS = D / 16;             // MFMA K steps per slab
    constexpr int CPR = D / 8;             // 16-byte chunks per row of a slab
    constexpr int SLAB = 32 * D * 2;       // bytes of one 32-row slab
    constexpr int PW = D / 64;             // LDS-DMA instructions per wave and slab (D / 16 per slab)
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);  // scalar: the DMA's LDS addresses stay in SGPRs
    const int m = lane & 31, h = lane >> 5;
    // consecutive workgroup ids go round the 8 XCDs: give each XCD a contiguous run of (tile, part) pairs, so that the
    // NQ parts of a row tile share one L2 (their x rows are fetched from HBM once)
    int bid = blockIdx.x;
    if ((gridDim.x & 7) == 0) bid = (bid & 7) * (gridDim.x >> 3) + (bid >> 3);
    const int tile = bid / NQ, part = bid % NQ;
    const uint32_t smem_lds = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) char*)smem;
    float* bias_s = (float*)(smem + FZ_RING * SLAB);
    if (blockIdx.x == 0 && tid == 0) {
        if (step_count) *step_count += 1;  // the dead-feature clock of model.py:175 (this is the step's first kernel)
        *n_fail = 0;
    }
    const int row0 = tile * FZ_ROWS + wave * 32;

    // LDS image of a slab (32 rows of D bf16): chunk c of row r sits at position r * CPR + ((c & ~15) | ((c ^ r) & 15)).
    // A fragment read (row m, chunk 2 ks + h) then takes 16 distinct 16-byte slots per 16-lane group: conflict-free.
    // With base = (h ^ m) & 15 the chunk's position is ((2 ks & 15) ^ base) + (chunk & ~15): eight lane addresses,
    // everything else (K step / 8, ring slot) is an immediate.
    uint32_t a_addr[8];
#pragma unroll
    for (int j = 0; j < 8; ++j)
        a_addr[j] = smem_lds + (uint32_t)m * (CPR * 16) + (uint32_t)(((2 * j) ^ ((h ^ m) & 15)) & 15) * 16u;

    // ---- the wave's 32 batch rows -> A fragments.  Loaded in whole 128-byte lines (8 lanes per row, 8 rows per
    // instruction) and turned into the fragment layout through a ring slot: loading the fragments directly
    // (2 x 16 bytes of 32 different rows per instruction) re-fetches every line four times through a thrashing L1 ----
    bf16x8 xf[KS];
    {
        const int r8 = lane >> 3, c8 = lane & 7;
        bf16x8 xq[4][PW];
#pragma unroll
        for (int it = 0; it < 4; ++it) {
            const int rr = min(row0 + r8 + 8 * it, B - 1);
            const bf16_t* xr = x + (rows ? (int64_t)rows[rr] : (int64_t)rr) * D + c8 * 8;
#pragma unroll
            for (int jj = 0; jj < PW; ++jj) xq[it][jj] = *(const bf16x8*)(xr + jj * 64);
        }
        for (int i = tid; i < SP * 32; i += 256) bias_s[i] = bias[part * SP * 32 + i];
        // waves 0..2 use slots 0..2, wave 3 follows in slot 0
#pragma unroll
        for (int round = 0; round < 2; ++round) {
            if ((wave < 3) == (round == 0)) {
                char* slot = smem + (wave % 3) * SLAB;
#pragma unroll
                for (int it = 0; it < 4; ++it) {
                    const int r = r8 + 8 * it;
#pragma unroll
                    for (int jj = 0; jj < PW; ++jj) {
                        const int c = jj * 8 + c8;
                        *(bf16x8*)(slot + (r * CPR + ((c & ~15) | ((c ^ r) & 15))) * 16) = xq[it][jj];
                    }
                }
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                __builtin_amdgcn_wave_barrier();
                const char* sl = smem + (wave % 3) * SLAB;
#pragma unroll
                for (int ks = 0; ks < KS; ++ks) {
                    const int c = 2 * ks + h;
                    xf[ks] = *(const bf16x8*)(sl + (m * CPR + ((c & ~15) | ((c ^ m) & 15))) * 16);
                }
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            }
            __syncthreads();
        }
    }

    // ---- W_e slabs by LDS-DMA: one instruction fills 64 consecutive 16-byte positions of the image, so lane l of
    // instruction i fetches the chunk that belongs at position 64 i + l ----
    uint32_t dma_off[PW];
#pragma unroll
    for (int j = 0; j < PW; ++j) {
        const int p = 64 * (wave + 4 * j) + lane;
        const int r = p / CPR, cpos = p % CPR;
        const int c = (cpos & ~15) | ((cpos ^ r) & 15);
        dma_off[j] = (uint32_t)(r * D * 2 + c * 16);
    }
    const char* wpart = (const char*)W + (int64_t)part * SP * SLAB;
    auto dma = [&](int s) {
        const char* wb = wpart + (int64_t)s * SLAB;
        const uint32_t slot = smem_lds + (uint32_t)(s % FZ_RING) * SLAB;
#pragma unroll
        for (int j = 0; j < PW; ++j) glds16s(wb, dma_off[j], slot + (uint32_t)(wave + 4 * j) * 1024u);
    };
    dma(0);
    dma(1);

    // the wave's 16 pair lists in this part's region: pair r = rows (fz_row_of(r), + 4) of the wave
    const int PCAP = 2 * WSAE_FZ_ROW_CAP / NQ;
    const int64_t region = ((int64_t)(tile * NQ + part) * 4 + wave) * 16;
    uint2* lists = cand + region * PCAP;
    int cnt[16];  // r * PCAP + entries filed so far: the next free slot of pair r counted from the region's start
#pragma unroll
    for (int r = 0; r < 16; ++r) cnt[r] = r * PCAP;
    float T[16];

    // file register r of a slab: entry = {feature << 1 | half, value bits} for every lane with v >= T[r], appended
    // behind the pair's counter.  5 vector + 5 scalar instructions and one store under the mask (EXEC switched around
    // it): no branch.  A list with less than a wave's worth of room left keeps writing over its last 64 slots - still
    // inside the list - and its count, which keeps growing, tells the select kernel that it overflowed.
    auto file_one = [&](int r, float v, uint32_t tag) {
        unsigned long long mask;
        asm volatile("v_cmp_ge_f32_e64 %0, %1, %2" : "=s"(mask) : "v"(v), "v"(T[r]));  // (asm: stays in its MFMA gap)
        const uint32_t p = __builtin_amdgcn_mbcnt_hi((uint32_t)(mask >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mask, 0u));
        const uint32_t off = (p + (uint32_t)min(cnt[r], r * PCAP + PCAP - 64)) << 3;
        cnt[r] += __popcll(mask);
        const uint64_t data = ((uint64_t)__float_as_uint(v) << 32) | tag;
        // lane 63 always stores: the counted waits of slab_begin rely on every one of these stores being issued (one
        // under an empty mask is not).  When lane 63 does not pass, its entry lands in the slot right behind the
        // passing lanes' - the list's next free slot, not counted, overwritten by the next entry.
        asm volatile("s_mov_b64 exec, %0\n\ts_bitset1_b64 exec, 63\n\tglobal_store_dwordx2 %1, %2, %3\n\ts_mov_b64 exec, -1"
                     :: "s"(mask), "v"(off), "v"(data), "s"(lists) : "memory");
    };
    auto tag_of = [&](int s) { return (uint32_t)(((part * SP + s) * 32 + m) << 1 | h); };
    auto file = [&](int s, const f32x16& v) {
        const uint32_t tag = tag_of(s);
#pragma unroll
        for (int r = 0; r < 16; ++r) file_one(r, v[r], tag);
    };

    // one slab: 24 MFMAs with their B fragments (feature row m of the slab) FZ_AHEAD K steps ahead - an LDS read takes
    // ~130 cycles to return, four MFMAs' worth; left to hipcc the reads sit two ahead.  The reads and their counted
    // waits are asm so that they stay where they are put; each wait hands its fragment to the MFMA through an in/out
    // operand.  The ring has two more registers than reads in flight: a register is refilled two MFMAs after the one
    // that read it.  SLOT is a compile-time constant: ring slot and K step become immediate offsets.
    // FILE: the previous slab's 16 registers are filed in the gaps behind the first 16 MFMAs.  The two workgroups of
    // a CU run in step, so a separate filing phase does not hide under the other workgroup's MFMAs - both file, then
    // both multiply; interleaved in one instruction stream the vector work sits in the MFMAs' shadow.
    auto slab_mfma = [&](auto slot_c, auto file_c, int s, f32x16& acc, const f32x16& prev) {
        constexpr int SO = decltype(slot_c)::value * SLAB;
        constexpr bool FILE = decltype(file_c)::value;
        constexpr int FZ_AHEAD = 6, R = FZ_AHEAD + 2;
        const uint32_t tag = tag_of(s - 1);
        bf16x8 w[R];
#pragma unroll
        for (int ks = 0; ks < FZ_AHEAD; ++ks) w[ks] = lds_read16(a_addr[ks & 7], SO + (ks >> 3) * 256);
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            if (ks + FZ_AHEAD < KS) w[(ks + FZ_AHEAD) % R] = lds_read16(a_addr[(ks + FZ_AHEAD) & 7], SO + ((ks + FZ_AHEAD) >> 3) * 256);
            lds_wait_for(w[ks % R], (KS - 1 - ks) < FZ_AHEAD ? (KS - 1 - ks) : FZ_AHEAD);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(xf[ks], w[ks % R], acc, 0, 0, 0);
            if constexpr (FILE) {
                if (KS >= 16) {
                    if (ks < 16) file_one(ks, prev[ks], tag);
                } else {  // fewer MFMAs than registers: two per gap
                    if (2 * ks < 16) file_one(2 * ks, prev[2 * ks], tag);
                    if (2 * ks + 1 < 16) file_one(2 * ks + 1, prev[2 * ks + 1], tag);
                }
            }
        }
        const float b = bias_s[s * 32 + m];
        const f32x2 b2 = {b, b};
#pragma unroll
        for (int i = 0; i < 8; ++i) {  // (two adds per instruction)
            f32x2 t = {acc[2 * i], acc[2 * i + 1]};
            t += b2;
            acc[2 * i] = t[0];
            acc[2 * i + 1] = t[1];
        }
    };
    auto slab_any = [&](auto file_c, int s, f32x16& acc, const f32x16& prev) {
        switch (s % FZ_RING) {
            case 0: slab_mfma(std::integral_constant<int, 0>{}, file_c, s, acc, prev); break;
            case 1: slab_mfma(std::integral_constant<int, 1>{}, file_c, s, acc, prev); break;
            default: slab_mfma(std::integral_constant<int, 2>{}, file_c, s, acc, prev); break;
        }
    };
    // Retire slab s's DMA and meet the other waves.  Vector memory operations retire in issue order, so
    // s_waitcnt vmcnt(N) with N = the operations issued AFTER slab s's pieces leaves exactly those in flight: the next
    // slab's PW pieces, plus the 16 stores filed in the MFMA gaps since (the 64 of the calibration slabs right after
    // them: capped at the counter's 63).  Counting fewer than were issued only waits longer.
    auto slab_begin = [&](int s) {
        const bool more = s + 1 < SP;
        if (s < FZ_SAMPLE) { if (more) vm_wait<PW>(); else vm_wait<0>(); }
        else if (s < FZ_SAMPLE + 2) vm_wait<63>();
        else if (more) vm_wait<PW + 16>();
        else vm_wait<16>();
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
    };

    // ---- calibration: the first FZ_SAMPLE slabs stay in registers ----
    f32x16 keep[FZ_SAMPLE];
#pragma unroll
    for (int s = 0; s < FZ_SAMPLE; ++s) {
        slab_begin(s);
        if (s + 2 < SP) dma(s + 2);
        slab_any(std::false_type{}, s, keep[s], keep[s]);
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        float g = fmaxf(fmaxf(keep[0][r], keep[1][r]), fmaxf(keep[2][r], keep[3][r]));
        // maximum of each group of 4 lanes, then the minimum over the half-wave's 8 groups
        g = fmaxf(g, __uint_as_float(lane_xor_u32<1>(__float_as_uint(g), lane)));
        g = fmaxf(g, __uint_as_float(lane_xor_u32<2>(__float_as_uint(g), lane)));
        g = fminf(g, __uint_as_float(lane_xor_u32<4>(__float_as_uint(g), lane)));
        g = fminf(g, __uint_as_float(lane_xor_u32<8>(__float_as_uint(g), lane)));
        g = fminf(g, __uint_as_float(lane_xor_u32<16>(__float_as_uint(g), lane)));
        const int row = row0 + fz_row_of(r) + 4 * h;
        T[r] = row < B ? g : INFINITY;
        if (m == 0 && row < B) cand_T[(int64_t)row * WSAE_FZ_MAX_PARTS + part] = g;
    }
#pragma unroll
    for (int s = 0; s < FZ_SAMPLE; ++s) file(s, keep[s]);

    // ---- the rest of the part: slab s's values are filed after the barrier of slab s + 1 ----
    f32x16 prev, cur;
    if (FZ_SAMPLE < SP) {
        slab_begin(FZ_SAMPLE);
        if (FZ_SAMPLE + 2 < SP) dma(FZ_SAMPLE + 2);
        slab_any(std::false_type{}, FZ_SAMPLE, prev, prev);
    }
    for (int s = FZ_SAMPLE + 1; s < SP; ++s) {
        slab_begin(s);
        if (s + 2 < SP) dma(s + 2);
        slab_any(std::true_type{}, s, cur, prev);
        prev = cur;
    }
    if (SP > FZ_SAMPLE) file(SP - 1, prev);
    int mine = 0;
#pragma unroll
    for (int r = 0; r < 16; ++r) mine = lane == r ? cnt[r] - r * PCAP : mine;
    if (lane < 16) cand_cnt[region + lane] = mine;
}

// ------------------------------------------------------------------------------------------------
#define FZ_LIST 256

// One wave per row pair.  All of the pair's entries are requested up front - LPP = PCAP / 128 loads per lane and
// part, half of a list's capacity, the requests independent of each other - and stay in registers: walking the lists
// with one dependent load per 64 entries was latency-bound (53 us for 16384 rows).  A list longer than that sends the
// pair to the fallback.
template <int NQ>
__global__ void __launch_bounds__(256)
select_filtered_kernel(const uint2* __restrict__ cand, const int32_t* __restrict__ cand_cnt, const float* __restrict__ cand_T,
                       int B, int K, float* __restrict__ vals, int32_t* __restrict__ idx,
                       int32_t* __restrict__ n_fail, int32_t* __restrict__ fail_list) {
    constexpr int PCAP = 2 * WSAE_FZ_ROW_CAP / NQ;
    constexpr int LPP = PCAP / 128;  // loads per lane and part
    __shared__ uint64_t lists[4][2][FZ_LIST];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int p = blockIdx.x * 4 + wave;                // pair index: ((tile * 4 + w) * 16 + r)
    const int tile = p >> 6, w = (p >> 4) & 3, r = p & 15;
    const int rowA = tile * FZ_ROWS + w * 32 + fz_row_of(r), rowB = rowA + 4;
    if (rowA >= B) return;
    const bool hasB = rowB < B;
    auto region = [&](int q) { return ((int64_t)(tile * NQ + q) * 4 + w) * 16 + r; };  // list q of this pair
    const int n_l = lane < NQ ? cand_cnt[region(lane)] : 0;
    const bool fail = __any(n_l > PCAP - 64);  // (a list that got this far may have overwritten entries)
    // batches of 64 * LPP entries per part: one for all but a few pairs per thousand
    int nmax = n_l;
#pragma unroll
    for (int o = 8; o > 0; o >>= 1) nmax = max(nmax, __shfl_xor(nmax, o, 64));
    nmax = __shfl(nmax, 0, 64);
    const int nb = fail ? 0 : (nmax + 64 * LPP - 1) / (64 * LPP);
    const float tA = lane < NQ ? cand_T[(int64_t)rowA * WSAE_FZ_MAX_PARTS + lane] : -INFINITY;
    const float tB = lane < NQ && hasB ? cand_T[(int64_t)rowB * WSAE_FZ_MAX_PARTS + lane] : -INFINITY;
    const uint32_t tmax[2] = {f32_ord(wave_max(tA)), f32_ord(wave_max(tB))};
    uint2 e[NQ][LPP];
    uint32_t o[NQ][LPP];  // orderable values; 0 (below every real value's key) where there is no entry
    auto load = [&](int bt) {
#pragma unroll
        for (int q = 0; q < NQ; ++q) {
            const int n = __shfl(n_l, q, 64);
            const uint2* lq = cand + region(q) * PCAP;
#pragma unroll
            for (int t = 0; t < LPP; ++t) {
                // every load is issued, whatever the list's length (a load behind `if (j < n)` is compiled as a branch
                // and a full wait per element: 32 dependent round trips); slots past the end get o = 0 below
                e[q][t] = lq[min(lane + 64 * (t + LPP * bt), PCAP - 1)];
            }
        }
#pragma unroll
        for (int q = 0; q < NQ; ++q) {
            const int n = __shfl(n_l, q, 64);
#pragma unroll
            for (int t = 0; t < LPP; ++t) o[q][t] = lane + 64 * (t + LPP * bt) < n ? f32_ord(__uint_as_float(e[q][t].y)) : 0u;
        }
    };
    uint32_t mx[2] = {0u, 0u};
    int nok[2] = {0, 0};
    for (int bt = 0; bt < nb; ++bt) {
        load(bt);
#pragma unroll
        for (int q = 0; q < NQ; ++q) {
#pragma unroll
            for (int t = 0; t < LPP; ++t) {
                const bool hb = e[q][t].x & 1;
                const bool okA = !hb && o[q][t] >= tmax[0] && o[q][t] != 0u, okB = hb && o[q][t] >= tmax[1] && o[q][t] != 0u;
                mx[0] = okA ? max(mx[0], o[q][t]) : mx[0];
                mx[1] = okB ? max(mx[1], o[q][t]) : mx[1];
                nok[0] += okA;
                nok[1] += okB;
            }
        }
    }
    bool bad[2];
    uint32_t thr[2];
#pragma unroll
    for (int t = 0; t < 2; ++t) {
        bad[t] = fail || wave_sum_i(nok[t]) < K;
        thr[t] = tmax[t];
        if (K <= 64) {
            uint32_t mk[1] = {mx[t]};
            wave_sort_desc<1, uint32_t>(mk, lane);
            thr[t] = max(thr[t], __shfl(mk[0], K - 1, 64));  // fewer than K lanes with a maximum: everything >= Tmax passes
        }
    }
    // compact the keys >= the row's threshold (a single batch is still in registers)
    int total[2] = {0, 0};
    for (int bt = 0; bt < nb; ++bt) {
        if (nb > 1) load(bt);
#pragma unroll
        for (int q = 0; q < NQ; ++q) {
#pragma unroll
            for (int t = 0; t < LPP; ++t) {
                const bool hb = e[q][t].x & 1;
                const uint64_t key = ((uint64_t)o[q][t] << 32) | (uint32_t)(~(e[q][t].x >> 1));
#pragma unroll
                for (int u = 0; u < 2; ++u) {
                    const bool pass = o[q][t] != 0u && hb == (bool)u && o[q][t] >= thr[u];
                    const unsigned long long mask = __ballot(pass);
                    if (mask) {
                        const int pos = total[u] + __popcll(mask & ((1ull << lane) - 1ull));
                        if (pass && pos < FZ_LIST) lists[wave][u][pos] = key;
                        total[u] += __popcll(mask);
                    }
                }
            }
        }
    }
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int t = 0; t < 2; ++t) {
        const int b = t ? rowB : rowA;
        if (t && !hasB) break;
        if (bad[t] || total[t] > FZ_LIST || total[t] < K) {
            if (lane == 0) fail_list[atomicAdd(n_fail, 1)] = b;
        } else {
            topk_emit_any<FZ_LIST>(lists[wave][t], total[t], K, lane, vals + (int64_t)b * K, idx + (int64_t)b * K);
        }
    }
}

// ------------------------------------------------------------------------------------------------
// TopK of one full row (K <= 64): K-th largest of the 64 lane maxima as threshold, compaction, sort; anything
// unusual (ties filling the list) goes to the bisection
template <int CAP>
__device__ __forceinline__ void topk_row_lanemax(const float* row, int H, int K, uint64_t* list, int lane, float* vrow,
                                                 int32_t* irow, int32_t* fallback_rows) {
    float mxv = -INFINITY;
    bool any = false;
    for (int e0 = lane * 4; e0 < H; e0 += 256) {
        const float4 v = *(const float4*)(row + e0);
        mxv = fmaxf(fmaxf(mxv, fmaxf(v.x, v.y)), fmaxf(v.z, v.w));
        any = true;
    }
    uint32_t mk[1] = {any ? f32_ord(mxv) : 0u};
    wave_sort_desc<1, uint32_t>(mk, lane);
    const uint64_t kmin = (uint64_t)__shfl(mk[0], K - 1, 64) << 32;
    const int count = topk_compact<CAP>(row, H, kmin, list, lane);
    if (count > CAP || count < K) {
        topk_row_generic<CAP>(row, H, K, list, lane, vrow, irow, fallback_rows);
        return;
    }
    if (lane == 0) atomicAdd(fallback_rows, 1);
    __builtin_amdgcn_wave_barrier();
    topk_emit_any<CAP>(list, count, K, lane, vrow, irow);
}

// rows the filter could not settle: pre[f] = x . W[f] + bias[f] by plain FMAs, one workgroup per (row, 256 features);
// the last of a row's workgroups to arrive copies the finished row to LDS and runs the exact TopK there
#define FZ_FB_MAXH 16384
__global__ void __launch_bounds__(256)
encode_fallback_kernel(const bf16_t* __restrict__ x, const int32_t* __restrict__ rows, const bf16_t* __restrict__ W,
                       const float* __restrict__ bias, int D, int H, int K, const int32_t* __restrict__ n_fail,
                       const int32_t* __restrict__ fail_list, float* __restrict__ pre, int32_t* __restrict__ tickets,
                       float* __restrict__ vals, int32_t* __restrict__ idx, int32_t* __restrict__ fallback_rows) {
    extern __shared__ __attribute__((aligned(16))) char fb_smem[];
    float* xs = (float*)fb_smem;                       // [D <= 2048]
    uint64_t* list = (uint64_t*)(fb_smem + 8192);      // [256]
    float* rowbuf = (float*)(fb_smem + 8192 + 2048);   // [H] when H <= FZ_FB_MAXH
    __shared__ int last_s;
    const int n = *n_fail;
    const int chunks = (H + 255) / 256;
    const int tid = threadIdx.x;
    for (int item = blockIdx.x; item < n * chunks; item += gridDim.x) {
        const int i = item / chunks, ch = item % chunks;
        const int b = fail_list[i];
        const bf16_t* xr = x + (rows ? (int64_t)rows[b] : (int64_t)b) * D;
        __syncthreads();  // xs / last_s / rowbuf of the previous item are no longer read
        for (int d = tid; d < D; d += 256) xs[d] = (float)xr[d];
        __syncthreads();
        float* prow = pre + (int64_t)i * H;
        // a wave per feature: the feature's W row read as one coalesced run (8 bf16 per lane), the products summed
        // in a fixed lane order
        const int wv = tid >> 6, ln = tid & 63;
        for (int fo = wv; fo < 256; fo += 4) {
            const int f = ch * 256 + fo;
            if (f >= H) break;
            float acc = 0.f;
            for (int d8 = ln; d8 < D / 8; d8 += 64) {
                const bf16x8 w = *(const bf16x8*)(W + (int64_t)f * D + d8 * 8);
#pragma unroll
                for (int e = 0; e < 8; ++e) acc = fmaf((float)w[e], xs[8 * d8 + e], acc);
            }
            acc = wave_sum(acc);
            if (ln == 0) prow[f] = acc + bias[f];
        }
        __threadfence();
        __syncthreads();
        if (tid == 0) {
            const int t = atomicAdd(&tickets[i], 1);
            last_s = t == chunks - 1;
            if (last_s) tickets[i] = 0;
        }
        __syncthreads();
        if (last_s) {
            __threadfence();
            const float* src = prow;
            if (H <= FZ_FB_MAXH) {  // the bisection re-reads the row 64 times: from LDS, not from L2
                for (int e = tid; e < H; e += 256) rowbuf[e] = __builtin_nontemporal_load(prow + e);
                src = rowbuf;
            }
            __syncthreads();
            if (tid < 64) {
                if (K <= 64) topk_row_lanemax<256>(src, H, K, list, tid, vals + (int64_t)b * K, idx + (int64_t)b * K, fallback_rows);
                else topk_row_generic<256>(src, H, K, list, tid, vals + (int64_t)b * K, idx + (int64_t)b * K, fallback_rows);
            }
        }
    }
}

template <int D>
static int launch_filter(wsae_ctx* c, const bf16_t* x, const int32_t* rows, int B, int NQ, int SP, int64_t* step_count,
                         hipStream_t st) {
    const int lds = FZ_RING * 32 * D * 2 + SP * 32 * 4;
    encode_filter_kernel<D><<<ceil_div(B, FZ_ROWS) * NQ, 256, lds, st>>>(x, rows, c->We_bf16, c->c_fold, B, NQ, SP, (uint2*)c->fz_cand,
                                                                        c->fz_cnt, c->fz_T, c->fz_state, step_count);
    return WSAE_OK;
}

}  // namespace

// parts per row tile for this batch: enough workgroups for two per CU, at least FZ_SAMPLE + 2 slabs per part
static int fz_parts(const wsae_ctx* c, int B) {
    const int slabs = c->H / 32, tiles = ceil_div(B, FZ_ROWS);
    int nq = 1;
    while (nq < WSAE_FZ_MAX_PARTS && tiles * nq < 2 * c->cus && slabs % (2 * nq) == 0 && slabs / (2 * nq) >= FZ_SAMPLE + 2) nq *= 2;
    return nq;
}

bool wsae_internal_fused_ok(const wsae_ctx* c, int x_dtype, int B) {
    if (!c->fz_cand || c->prec != WSAE_PREC_BF16 || x_dtype != WSAE_DT_BF16) return false;
    if (c->D != 128 && c->D != 256 && c->D != 384) return false;
    // the threshold sits near the row's 14 % quantile: keep K well below that share of the features
    if (c->K > 64 || c->H < 64 * c->K || B < 1024) return false;
    const int nq = fz_parts(c, B);
    const int sp = c->H / 32 / nq;
    return (c->H / 32) % nq == 0 && sp >= FZ_SAMPLE + 1 && sp * 32 * 4 <= 8 * 1024;
}

extern "C" int wsae_ctx_encode_path(const wsae_ctx* ctx, int32_t x_dtype, int32_t B) {
    if (!ctx || B < 1 || B > ctx->maxB) return -1;
    return wsae_internal_fused_ok(ctx, x_dtype, B) ? 1 : 0;
}

int wsae_internal_encode_select(wsae_ctx* c, const void* x, const int32_t* rows, int B, float* vals, int32_t* idx,
                                int64_t* step_count, int32_t* fb, hipStream_t st) {
    const int NQ = fz_parts(c, B), SP = c->H / 32 / NQ;
    const bf16_t* xb = (const bf16_t*)x;
    int rc;
    WSAE_PROF_BEGIN(c, WSAE_K_ENCODE_GEMM, st);
    if (c->D == 128) rc = launch_filter<128>(c, xb, rows, B, NQ, SP, step_count, st);
    else if (c->D == 256) rc = launch_filter<256>(c, xb, rows, B, NQ, SP, step_count, st);
    else rc = launch_filter<384>(c, xb, rows, B, NQ, SP, step_count, st);
    WSAE_PROF_END(c, WSAE_K_ENCODE_GEMM, st);
    if (rc) return rc;
    WSAE_LAUNCH_CHECK();
    WSAE_PROF_BEGIN(c, WSAE_K_TOPK, st);
    const int pairs = ceil_div(B, FZ_ROWS) * 64;
#define FZ_SELECT(N) select_filtered_kernel<N><<<pairs / 4, 256, 0, st>>>((const uint2*)c->fz_cand, c->fz_cnt, c->fz_T, B, c->K, vals, idx, \
                                                                       c->fz_state, c->fz_fail)
    switch (NQ) {
        case 1: FZ_SELECT(1); break;
        case 2: FZ_SELECT(2); break;
        case 4: FZ_SELECT(4); break;
        case 8: FZ_SELECT(8); break;
        default: FZ_SELECT(16); break;
    }
#undef FZ_SELECT
    const int fb_lds = 8192 + 2048 + (c->H <= FZ_FB_MAXH ? c->H * 4 : 0);
    encode_fallback_kernel<<<min(c->cus, 256), 256, fb_lds, st>>>(xb, rows, c->We_bf16, c->c_fold, c->D, c->H, c->K, c->fz_state,
                                                                 c->fz_fail, c->pre, c->fz_tickets, vals, idx, fb);
    WSAE_PROF_END(c, WSAE_K_TOPK, st);
    WSAE_LAUNCH_CHECK();
    c->xT_valid = 0;     // nothing was staged: wsae_weight_grads transposes x itself
    c->smax_valid = 0;
    return WSAE_OK;
}
