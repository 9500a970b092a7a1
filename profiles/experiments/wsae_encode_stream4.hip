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

    // ---- output.  A finished slab (register r of this lane = pre[row0 + es_row_of(r) + 4 h][32 s + m]) goes through a
    // 32 x 32 patch of the wave in LDS and leaves as 16-byte stores: 4 store instructions per slab (8 rows x 128 bytes
    // each) + 4 for the strip maxima, instead of 16 four-byte ones + 1.  The output phases of these kernels are bound by
    // the NUMBER of store instructions, not by their bytes.  Spread over the next slab's K steps:
    //   ks = 0: 16 ds_write_b32 (immediate row offsets);  ks = 2: 4 ds_read_b128;  ks = 4 + i: row group i - store,
    //   strip maxima (v_max3 / v_max / two quad DPP steps: a quad of lanes = one 16-column strip), their store.
    constexpr int PSTR = 36;  // floats per patch row (16-byte aligned rows)
    const uint32_t patch_lds = smem_lds + ES_RING * SLAB + 8 * 1024 + (uint32_t)wave * (32 * PSTR * 4);
    const uint32_t pw_addr = patch_lds + (uint32_t)((4 * h) * PSTR + m) * 4u;
    const uint32_t pr_addr = patch_lds + (uint32_t)((lane >> 3) * PSTR + (lane & 7) * 4) * 4u;
    uint32_t o_pre4[4], o_sm4[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const uint32_t row = (uint32_t)(row0 + (lane >> 3) + 8 * i);
        o_pre4[i] = (row * (uint32_t)H + (uint32_t)(lane & 7) * 4u) * 4u;
        o_sm4[i] = (row * (uint32_t)(H >> 4) + (uint32_t)((lane & 7) >> 2)) * 4u;
    }
    const char* pre_part = (const char*)pre + (int64_t)part * SP * 32 * 4;
    const char* smax_part = (const char*)smax + (int64_t)part * SP * 2 * 4;
    auto emit_write = [&](const f32x16& v) {
#define ES_W(r) asm volatile("ds_write_b32 %0, %1 offset:%2" ::"v"(pw_addr), "v"(v[r]), "n"(es_row_of(r) * PSTR * 4) : "memory")
        ES_W(0); ES_W(1); ES_W(2); ES_W(3); ES_W(4); ES_W(5); ES_W(6); ES_W(7);
        ES_W(8); ES_W(9); ES_W(10); ES_W(11); ES_W(12); ES_W(13); ES_W(14); ES_W(15);
#undef ES_W
    };
    f32x4 pv[4];
    auto emit_read = [&]() {
#define ES_R(i) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(pv[i]) : "v"(pr_addr), "n"((i) * 8 * PSTR * 4) : "memory")
        ES_R(0); ES_R(1); ES_R(2); ES_R(3);
#undef ES_R
    };
    // one row group (rows (lane >> 3) + 8 i): `newer` = LDS operations issued after its ds_read_b128 that may still be out
    auto emit_store = [&](f32x4& p, uint32_t op, uint32_t os, int newer, const char* pbase, const char* sbase) {
        switch (newer) {
            case 0: asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(p)); break;
            case 1: asm volatile("s_waitcnt lgkmcnt(1)" : "+v"(p)); break;
            case 2: asm volatile("s_waitcnt lgkmcnt(2)" : "+v"(p)); break;
            case 3: asm volatile("s_waitcnt lgkmcnt(3)" : "+v"(p)); break;
            default: asm volatile("s_waitcnt lgkmcnt(5)" : "+v"(p)); break;
        }
        float mx, t;
        asm volatile("global_store_dwordx4 %2, %3, %4\n\t"
                     "v_max3_f32 %0, %5, %6, %7\n\t"
                     "v_max_f32 %0, %0, %8\n\t"
                     "s_nop 1\n\t"
                     "v_max_f32_dpp %1, %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
                     "s_nop 1\n\t"
                     "v_max_f32_dpp %0, %1, %1 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n\t"
                     "s_nop 0\n\t"
                     "global_store_dword %9, %0, %10"
                     : "=&v"(mx), "=&v"(t)
                     : "v"(op), "v"(p), "s"(pbase), "v"(p[0]), "v"(p[1]), "v"(p[2]), "v"(p[3]), "v"(os), "s"(sbase)
                     : "memory");
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
                // (5 = the patch reads after row group i's + the B fragment reads issued since: 3 - i + 2 + i)
                if (ks == 0) emit_write(prev);
                else if (ks == 2) emit_read();
                else if (ks >= 4 && ks < 8) emit_store(pv[ks - 4], o_pre4[ks - 4], o_sm4[ks - 4], 5, pbase, sbase);
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
    // pieces leaves exactly those in flight: the next slab's PW pieces and the 8 stores emitted in the MFMA gaps of the slab before this one
    // (every one of them is issued: none sits under an empty mask).  Counting fewer than were issued only waits longer.
    auto slab_begin = [&](int s) {
        const bool more = s + 1 < SP;
        if (s < 2) { if (more) vm_wait<PW>(); else vm_wait<0>(); }
        else if (more) vm_wait<PW + 8>();
        else vm_wait<8>();
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
    auto emit_all = [&](const f32x16& v) {
        emit_write(v);
        emit_read();
#pragma unroll
        for (int i = 0; i < 4; ++i) emit_store(pv[i], o_pre4[i], o_sm4[i], 3 - i, pbase, sbase);
    };
    if (s < SP) {
        slab_begin(s);
        slab_any(std::true_type{}, s, cur, prev);
        emit_all(cur);
    } else {
        emit_all(prev);
    }
}

template <int D>
static void launch_stream(wsae_ctx* c, const bf16_t* x, const int32_t* rows, int B, int NQ, int SP, float* pre, float* smax,
                          int64_t* step_count, hipStream_t st) {
    const int lds = ES_RING * 32 * D * 2 + 8 * 1024 + ES_WAVES * 32 * 36 * 4;  // ring | bias (<= 8 KB) | one 32 x 36 float patch per wave
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
