// Encoder GEMM for narrow inputs (BF16 mode, D <= 384): pre [B][H] = x . W_e^T + c, plus the strip maxima the TopK reads.
//   reference: pre = encoder(x - b_pre)   src/whisper_sae/sae/model.py:108-111
//
// At K = D = 384 a 256 x 256 tile is only six K slabs deep: the persistent tile kernel (encode_gemm256d_kernel) spends
// its time waiting for the next slab's DMA behind a barrier and transposing its accumulators through LDS (56 us without
// its 201 MB store, 15 us at the MFMA rate).  This kernel turns the loop round:
//   * a workgroup owns 256 batch rows x one part of the features (one workgroup per CU: its 8 waves share every W_e slab,
//     so a slab crosses the CU's vector-memory path once per 256 rows); each wave keeps ITS 32 rows of x in
//     registers as MFMA A fragments for the whole kernel (96 VGPRs at D = 384) - x is read once, in whole 128-byte lines,
//     and turned into the fragment layout through a ring slot;
//   * W_e streams through a 3-deep LDS ring in 32-feature slabs (32 x D bf16, LDS-DMA, XOR swizzle on the source
//     address) and is the B operand; the slab after next is requested right after the barrier that frees its slot, the
//     counted vmcnt at the next barrier leaves exactly the newer operations in flight (stores included: vector memory
//     operations retire in issue order);
//   * the accumulators come out with ONE feature per lane (lane & 31) and 16 batch rows per lane (register r = row
//     m(r) in the lower half-wave, m(r) + 4 in the upper): register r of a half-wave is 32 consecutive floats of one
//     row, so it is stored as it stands - two 128-byte segments per store, no transpose through LDS;
//   * the maximum of every 16-feature strip is four DPP row shifts per register, stored by the strip's last lane;
//   * the stores and maxima of slab s are issued in the MFMA gaps of slab s + 1 (the two workgroups of a CU run in step:
//     a separate epilogue phase hides under nobody's MFMAs).
// Same MFMA instruction, same K order, bias added last: bit-identical to encode_gemm256d_kernel.
// The skeleton comes from the fused GEMM + TopK filter experiment (profiles/experiments/): what made that slow was the
// filter (12 instructions per value against one store here) and its select kernel, not the streaming loop.
#include "wsae_common.h"
#include "wsae_mfma.h"

#include <type_traits>

#define ES_ROWS 256
#define ES_WAVES 8
#define ES_RING 3

namespace {

template <int N>
__device__ __forceinline__ void vm_wait() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

// 16 bytes from LDS byte address addr + imm.  asm: hipcc sinks plain LDS reads to just before the MFMA that uses them
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
__host__ __device__ __forceinline__ constexpr int es_row_of(int r) { return (r & 3) + 8 * (r >> 2); }

// one 1 KiB LDS-DMA piece: lane l fetches 16 bytes at base + voff into LDS byte lds_addr + 16 l.  SGPR base + 32-bit
// lane offset; M0 is compiler-reserved, saved and restored.
__device__ __forceinline__ void glds16s(const void* base, uint32_t voff, uint32_t lds_addr) {
    uint32_t keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep)
                 : "v"(voff), "s"(base), "s"(lds_addr)
                 : "memory");
}

template <int D>
__global__ void __launch_bounds__(64 * ES_WAVES, 1)
encode_stream_kernel(const bf16_t* __restrict__ x, const int32_t* __restrict__ rows, const bf16_t* __restrict__ W,
                     const float* __restrict__ bias, int B, int H, int NQ, int SP, float* __restrict__ pre,
                     float* __restrict__ smax, int64_t* __restrict__ step_count) {
    constexpr int KS = D / 16;             // MFMA K steps per slab
    constexpr int CPR = D / 8;             // 16-byte chunks per row of a slab
    constexpr int SLAB = 32 * D * 2;       // bytes of one 32-row slab
    constexpr int PW = D / 16 / ES_WAVES;  // LDS-DMA instructions per wave and slab (D / 16 per slab)
    constexpr int XP = D / 64;             // 128-byte column blocks of an x row
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
    float* bias_s = (float*)(smem + ES_RING * SLAB);
    if (step_count && blockIdx.x == 0 && tid == 0) *step_count += 1;  // the dead-feature clock of model.py:175
    const int row0 = tile * ES_ROWS + wave * 32;

    // LDS image of a slab (32 rows of D bf16): chunk c of row r sits at position r * CPR + ((c & ~15) | ((c ^ r) & 15)).
    // A fragment read (row m, chunk 2 ks + h) then takes 16 distinct 16-byte slots per 16-lane group: conflict-free.
    // With base = (h ^ m) & 15 the chunk's position is ((2 ks & 15) ^ base) + (chunk & ~15): eight lane addresses,
    // everything else (K step / 8, ring slot) is an immediate.
    uint32_t a_addr[8];
#pragma unroll
    for (int j = 0; j < 8; ++j)
        a_addr[j] = smem_lds + (uint32_t)m * (CPR * 16) + (uint32_t)(((2 * j) ^ ((h ^ m) & 15)) & 15) * 16u;

    // ---- the wave's 32 batch rows -> A fragments: whole 128-byte lines (8 lanes per row, 8 rows per instruction)
    // through a ring slot (waves 0..2 use slots 0..2, wave 3 follows in slot 0) ----
    bf16x8 xf[KS];
    {
        const int r8 = lane >> 3, c8 = lane & 7;
        bf16x8 xq[4][XP];
#pragma unroll
        for (int it = 0; it < 4; ++it) {
            const int rr = row0 + r8 + 8 * it;  // (< B: the launcher takes whole row tiles only)
            const bf16_t* xr = x + (rows ? (int64_t)rows[rr] : (int64_t)rr) * D + c8 * 8;
#pragma unroll
            for (int jj = 0; jj < XP; ++jj) xq[it][jj] = *(const bf16x8*)(xr + jj * 64);
        }
        for (int i = tid; i < SP * 32; i += 64 * ES_WAVES) bias_s[i] = bias[part * SP * 32 + i];
#pragma unroll
        for (int round = 0; round < (ES_WAVES + 2) / 3; ++round) {
            if (wave / 3 == round) {
                char* slot = smem + (wave % 3) * SLAB;
#pragma unroll
                for (int it = 0; it < 4; ++it) {
                    const int r = r8 + 8 * it;
#pragma unroll
                    for (int jj = 0; jj < XP; ++jj) {
                        const int c = jj * 8 + c8;
                        *(bf16x8*)(slot + (r * CPR + ((c & ~15) | ((c ^ r) & 15))) * 16) = xq[it][jj];
                    }
                }
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                __builtin_amdgcn_wave_barrier();
#pragma unroll
                for (int ks = 0; ks < KS; ++ks) {
                    const int c = 2 * ks + h;
                    xf[ks] = *(const bf16x8*)(slot + (m * CPR + ((c & ~15) | ((c ^ m) & 15))) * 16);
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
        const int p = 64 * (wave + ES_WAVES * j) + lane;
        const int r = p / CPR, cpos = p % CPR;
        const int c = (cpos & ~15) | ((cpos ^ r) & 15);
        dma_off[j] = (uint32_t)(r * D * 2 + c * 16);
    }
    const char* wpart = (const char*)W + (int64_t)part * SP * SLAB;
    auto dma = [&](int s) {
        const char* wb = wpart + (int64_t)s * SLAB;
        const uint32_t slot = smem_lds + (uint32_t)(s % ES_RING) * SLAB;
#pragma unroll
        for (int j = 0; j < PW; ++j) glds16s(wb, dma_off[j], slot + (uint32_t)(wave + ES_WAVES * j) * 1024u);
    };
    dma(0);
    dma(1);

    // output addressing: register r of this lane is pre[row0 + es_row_of(r) + 4 h][part's features + 32 s + m]; the
    // row part is a per-lane byte offset (B * H * 4 < 4 GB: the launcher checks), the slab part an SGPR base
    uint32_t o_pre[16], o_smax[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const uint32_t row = (uint32_t)(row0 + es_row_of(r) + 4 * h);
        o_pre[r] = (row * (uint32_t)H + (uint32_t)m) * 4u;
        o_smax[r] = (row * (uint32_t)(H >> 4) + (uint32_t)(m >> 4)) * 4u;
    }
    const char* pre_part = (const char*)pre + (int64_t)part * SP * 32 * 4;
    const char* smax_part = (const char*)smax + (int64_t)part * SP * 2 * 4;

    // Emit a finished slab in groups of four registers (one group per MFMA gap):
    //   * the values as they stand: register r of a half-wave is 32 consecutive floats of one row, two 128-byte segments
    //     per store;
    //   * their 16-lane strip maxima: v_max_f32 with a DPP row shift of 1, 2, 4, 8 leaves the maximum in the strip's last
    //     lane (lanes without a source keep their value); the four registers' chains are interleaved, so a shifted
    //     read never follows the write of its source by less than three instructions;
    //   * the maxima go to a 64-float LDS row of the wave (lanes 15, 31, 47, 63 of register r -> slots 4 r + 0..3) and
    //     leave as ONE store per slab: a store instruction costs the memory pipe the same with 4 lanes as with 64.
    const uint32_t smax_row_lds = smem_lds + ES_RING * SLAB + 8 * 1024 + (uint32_t)wave * 256u;
    // lane l of the slab's one strip-maximum store: register l >> 2, source lane 15 + 16 (l & 3) = half (l & 3) >> 1, strip l & 1
    const uint32_t o_smax_lane = ((uint32_t)(row0 + es_row_of(lane >> 2) + 4 * ((lane & 3) >> 1)) * (uint32_t)(H >> 4) + (uint32_t)(lane & 1)) * 4u;
    const uint32_t my_slot = smax_row_lds + (uint32_t)(lane >> 4) * 4u;  // (for lanes 15, 31, 47, 63: + 16 r added per register)
    const unsigned long long strip_last = 0x8000800080008000ull;          // lanes 15, 31, 47, 63
    auto emit_four = [&](int r0, const f32x16& v, const char* pbase) {
        float m0 = v[r0], m1 = v[r0 + 1], m2 = v[r0 + 2], m3 = v[r0 + 3];
        asm volatile("global_store_dword %4, %0, %8\n\t"
                     "global_store_dword %5, %1, %8\n\t"
                     "global_store_dword %6, %2, %8\n\t"
                     "global_store_dword %7, %3, %8"
                     :: "v"(m0), "v"(m1), "v"(m2), "v"(m3), "v"(o_pre[r0]), "v"(o_pre[r0 + 1]), "v"(o_pre[r0 + 2]), "v"(o_pre[r0 + 3]), "s"(pbase)
                     : "memory");
#define ES_DPP(sh)                                                                   \
        asm volatile("s_nop 1\n\tv_max_f32_dpp %0, %0, %0 row_shr:" #sh " row_mask:0xf bank_mask:0xf\n\t" \
                     "v_max_f32_dpp %1, %1, %1 row_shr:" #sh " row_mask:0xf bank_mask:0xf\n\t" \
                     "v_max_f32_dpp %2, %2, %2 row_shr:" #sh " row_mask:0xf bank_mask:0xf\n\t" \
                     "v_max_f32_dpp %3, %3, %3 row_shr:" #sh " row_mask:0xf bank_mask:0xf"      \
                     : "+v"(m0), "+v"(m1), "+v"(m2), "+v"(m3))
        ES_DPP(1);
        ES_DPP(2);
        ES_DPP(4);
        ES_DPP(8);
#undef ES_DPP
        asm volatile("s_mov_b64 exec, %5\n\t"
                     "ds_write_b32 %4, %0 offset:%6\n\t"
                     "ds_write_b32 %4, %1 offset:%7\n\t"
                     "ds_write_b32 %4, %2 offset:%8\n\t"
                     "ds_write_b32 %4, %3 offset:%9\n\t"
                     "s_mov_b64 exec, -1"
                     :: "v"(m0), "v"(m1), "v"(m2), "v"(m3), "v"(my_slot), "s"(strip_last), "n"(16 * r0), "n"(16 * r0 + 16),
                        "n"(16 * r0 + 32), "n"(16 * r0 + 48)
                     : "memory");
    };
    // after the 16th register: the wave's 64 maxima as one store
    auto emit_smax = [&](const char* sbase) {
        float mv;
        asm volatile("s_waitcnt lgkmcnt(0)\n\tds_read_b32 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(mv) : "v"(smax_row_lds + (uint32_t)lane * 4u) : "memory");
        asm volatile("global_store_dword %0, %1, %2" ::"v"(o_smax_lane), "v"(mv), "s"(sbase) : "memory");
    };

    // one slab: KS MFMAs with their B fragments (feature row m of the slab) ES_AHEAD K steps ahead - an LDS read takes
    // ~130 cycles to return, four MFMAs' worth.  The reads and their counted waits are asm so that they stay where they
    // are put; each wait hands its fragment to the MFMA through an in/out operand.  The ring has two more registers
    // than reads in flight: a register is refilled two MFMAs after the one that read it.  SLOT is a compile-time
    // constant: ring slot and K step are immediate offsets.  EMIT: the previous slab's 16 registers go out in the gaps.
    auto slab_mfma = [&](auto slot_c, auto emit_c, int s, f32x16& acc, const f32x16& prev) {
        constexpr int SO = decltype(slot_c)::value * SLAB;
        constexpr bool EMIT = decltype(emit_c)::value;
        constexpr int ES_AHEAD = 6, R = ES_AHEAD + 2;
        const char* pbase = pre_part + (int64_t)(s - 1) * 32 * 4;
        const char* sbase = smax_part + (int64_t)(s - 1) * 2 * 4;
        bf16x8 w[R];
#pragma unroll
        for (int ks = 0; ks < ES_AHEAD; ++ks) w[ks] = lds_read16(a_addr[ks & 7], SO + (ks >> 3) * 256);
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            if (ks + ES_AHEAD < KS) w[(ks + ES_AHEAD) % R] = lds_read16(a_addr[(ks + ES_AHEAD) & 7], SO + ((ks + ES_AHEAD) >> 3) * 256);
            lds_wait_for(w[ks % R], (KS - 1 - ks) < ES_AHEAD ? (KS - 1 - ks) : ES_AHEAD);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(xf[ks], w[ks % R], acc, 0, 0, 0);
            if constexpr (EMIT) {
                if (ks < 4) emit_four(4 * ks, prev, pbase);
                else if (ks == 4) emit_smax(sbase);
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
    auto slab_any = [&](auto emit_c, int s, f32x16& acc, const f32x16& prev) {
        switch (s % ES_RING) {
            case 0: slab_mfma(std::integral_constant<int, 0>{}, emit_c, s, acc, prev); break;
            case 1: slab_mfma(std::integral_constant<int, 1>{}, emit_c, s, acc, prev); break;
            default: slab_mfma(std::integral_constant<int, 2>{}, emit_c, s, acc, prev); break;
        }
    };
    // Retire slab s's DMA and meet the other waves.  s_waitcnt vmcnt(N) with N = the operations issued AFTER slab s's
    // pieces leaves exactly those in flight: the next slab's PW pieces and the 17 stores emitted in the MFMA gaps since
    // (every one of them is issued: none sits under an empty mask).  Counting fewer than were issued only waits longer.
    auto slab_begin = [&](int s) {
        const bool more = s + 1 < SP;
        if (s < 2) { if (more) vm_wait<PW>(); else vm_wait<0>(); }
        else if (more) vm_wait<PW + 17>();
        else vm_wait<17>();
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
    };

    f32x16 prev, cur;
    slab_begin(0);
    if (2 < SP) dma(2);
    slab_any(std::false_type{}, 0, prev, prev);
    int s = 1;
    for (; s + 1 < SP; s += 2) {  // (two slabs per trip: the accumulators swap roles instead of being copied)
        slab_begin(s);
        if (s + 2 < SP) dma(s + 2);
        slab_any(std::true_type{}, s, cur, prev);
        slab_begin(s + 1);
        if (s + 3 < SP) dma(s + 3);
        slab_any(std::true_type{}, s + 1, prev, cur);
    }
    const char* pbase = pre_part + (int64_t)(SP - 1) * 32 * 4;
    const char* sbase = smax_part + (int64_t)(SP - 1) * 2 * 4;
    if (s < SP) {
        slab_begin(s);
        slab_any(std::true_type{}, s, cur, prev);
#pragma unroll
        for (int r0 = 0; r0 < 16; r0 += 4) emit_four(r0, cur, pbase);
    } else {
#pragma unroll
        for (int r0 = 0; r0 < 16; r0 += 4) emit_four(r0, prev, pbase);
    }
    emit_smax(sbase);
}

template <int D>
static void launch_stream(wsae_ctx* c, const bf16_t* x, const int32_t* rows, int B, int NQ, int SP, float* pre, float* smax,
                          int64_t* step_count, hipStream_t st) {
    const int lds = ES_RING * 32 * D * 2 + 8 * 1024 + ES_WAVES * 256;  // ring | bias (<= 8 KB) | one 64-float row per wave
    encode_stream_kernel<D><<<(B / ES_ROWS) * NQ, 64 * ES_WAVES, lds, st>>>(x, rows, c->We_bf16, c->c_fold, B, c->H, NQ, SP, pre, smax,
                                                                 step_count);
}

}  // namespace

// parts per row tile for this batch: enough workgroups for one per CU, at least 4 slabs per part
static int es_parts(const wsae_ctx* c, int B) {
    const int slabs = c->H / 32, tiles = B / ES_ROWS;
    int nq = 1;
    while (nq < 16 && tiles * nq < c->cus && slabs % (2 * nq) == 0 && slabs / (2 * nq) >= 4) nq *= 2;
    return nq;
}

// does the streaming kernel serve this batch?  BF16 mode, input_dim 128 / 256 / 384, whole 128-row tiles and 16-feature
// strips, the [B, H] matrix and its strip maxima addressable with 32-bit byte offsets
bool wsae_internal_stream_ok(const wsae_ctx* c, int B) {
    if (c->prec != WSAE_PREC_BF16 || (c->D != 128 && c->D != 256 && c->D != 384)) return false;
    if (B < 1024 || B % ES_ROWS || c->H % 256 || (int64_t)B * c->H * 4 >= (1ll << 32)) return false;
    const int sp = c->H / 32 / es_parts(c, B);
    return sp >= 3 && sp * 32 * 4 <= 8 * 1024;
}

// pre (= ctx->pre, leading dimension H) and ctx->smax from bf16 rows x (gathered through `rows` when given)
int wsae_internal_encode_stream(wsae_ctx* c, const void* x, const int32_t* rows, int B, int64_t* step_count, hipStream_t st) {
    const int NQ = es_parts(c, B), SP = c->H / 32 / NQ;
    const bf16_t* xb = (const bf16_t*)x;
    if (c->D == 128) launch_stream<128>(c, xb, rows, B, NQ, SP, c->pre, c->smax, step_count, st);
    else if (c->D == 256) launch_stream<256>(c, xb, rows, B, NQ, SP, c->pre, c->smax, step_count, st);
    else launch_stream<384>(c, xb, rows, B, NQ, SP, c->pre, c->smax, step_count, st);
    WSAE_LAUNCH_CHECK();
    c->smax_valid = 1;
    return WSAE_OK;
}
