// MFMA tile machinery shared by the encode GEMM and the weight-gradient GEMMs.
//
// Both contractions are "NT": C[m][n] = sum_k A[m][k] * Bt[n][k], with K contiguous in both
// operands, so every MFMA fragment is one contiguous LDS read.  Workgroup tile 128 x 128, four
// waves in a 2 x 2 arrangement, each wave 64 x 64 = 2 x 2 MFMA tiles of 32 x 32 (64 accumulator
// VGPRs).  K is walked in slabs of 128 bytes per row (64 bf16 or 32 f32); LDS rows are padded to
// 144 bytes, which makes the 16-byte fragment reads of any 16 consecutive rows hit 16 distinct
// 4-bank slots (bank = (addr/4) % 64, rows 36 dwords apart: cdna guide, LDS section).
//
// Fragment maps (gfx950, cdna guide section 3):
//   v_mfma_f32_32x32x16_bf16: lane l holds A[row l&31][k = 8*(l>>5)+j], B[k = 8*(l>>5)+j][col l&31]
//   v_mfma_f32_32x32x2_f32  : lane l holds A[row l&31][k = l>>5],        B[k = l>>5][col l&31]
//   C/D (both)              : reg r of lane l is C[(r&3) + 8*(r>>2) + 4*(l>>5)][l&31]
#pragma once

#include "wsae_common.h"

#define TILE_M 128
#define TILE_N 128
#define LDS_ROW_BYTES 144  // 128 data + 16 pad
#define TILE_LDS_BYTES (128 * LDS_ROW_BYTES)

template <typename T>
struct Mfma;

template <>
struct Mfma<bf16_t> {
    static constexpr int KT = 64;  // elements per K slab
    // acc[mi][ni] += A(64 rows at a_row0) x Bt(64 rows at b_row0) over the whole slab
    // ROWSUM: additionally rs[mi] += A(mi) x ones, i.e. every column of rs[mi] holds the row sums of
    // the A slice (exact fp32 accumulation in fixed order; used for db_e = sum_b dpre)
    template <bool ROWSUM = false>
    static __device__ __forceinline__ void slab(const char* As, const char* Bs, int a_row0, int b_row0, int lane,
                                                f32x16 (&acc)[2][2], f32x16* rs = nullptr) {
        const int r = lane & 31, h = lane >> 5;
        bf16x8 ones;
#pragma unroll
        for (int i = 0; i < 8; ++i) ones[i] = (bf16_t)1.0f;
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
            bf16x8 a[2], b[2];
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                a[i] = *(const bf16x8*)(As + (a_row0 + i * 32 + r) * LDS_ROW_BYTES + kk * 32 + h * 16);
                b[i] = *(const bf16x8*)(Bs + (b_row0 + i * 32 + r) * LDS_ROW_BYTES + kk * 32 + h * 16);
            }
#pragma unroll
            for (int mi = 0; mi < 2; ++mi)
#pragma unroll
                for (int ni = 0; ni < 2; ++ni)
                    acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[mi], b[ni], acc[mi][ni], 0, 0, 0);
            if (ROWSUM) {
#pragma unroll
                for (int mi = 0; mi < 2; ++mi)
                    rs[mi] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[mi], ones, rs[mi], 0, 0, 0);
            }
        }
    }
};

template <>
struct Mfma<float> {
    static constexpr int KT = 32;
    template <bool ROWSUM = false>
    static __device__ __forceinline__ void slab(const char* As, const char* Bs, int a_row0, int b_row0, int lane,
                                                f32x16 (&acc)[2][2], f32x16* rs = nullptr) {
        const int r = lane & 31, h = lane >> 5;
#pragma unroll
        for (int kk = 0; kk < 16; ++kk) {
            float a[2], b[2];
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                a[i] = *(const float*)(As + (a_row0 + i * 32 + r) * LDS_ROW_BYTES + (kk * 2 + h) * 4);
                b[i] = *(const float*)(Bs + (b_row0 + i * 32 + r) * LDS_ROW_BYTES + (kk * 2 + h) * 4);
            }
#pragma unroll
            for (int mi = 0; mi < 2; ++mi)
#pragma unroll
                for (int ni = 0; ni < 2; ++ni)
                    acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[mi], b[ni], acc[mi][ni], 0, 0, 0);
            if (ROWSUM) {
#pragma unroll
                for (int mi = 0; mi < 2; ++mi)
                    rs[mi] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[mi], 1.0f, rs[mi], 0, 0, 0);
            }
        }
    }
};

// 8-wave / 256 x 256 workgroup tiles: a wave owns 128 rows x 64 columns = 4 x 2 MFMA tiles (128
// accumulator VGPRs); per 16-deep K step it reads 4 A fragments + 2 B fragments for 8 MFMAs.
// (Per flop this moves half the operand bytes of the 128 x 128 tile: at K = 384 the 128 x 128 tile needs
// 64 B/clk/CU of L1 bandwidth at the MFMA peak, i.e. all of it.)
template <typename T>
struct Mfma256;

template <>
struct Mfma256<bf16_t> {
    static __device__ __forceinline__ void slab(const char* As, const char* Bs, int a_row0, int b_row0, int lane,
                                                f32x16 (&acc)[4][2]) {
        const int r = lane & 31, h = lane >> 5;
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
            bf16x8 a[4], b[2];
#pragma unroll
            for (int i = 0; i < 4; ++i)
                a[i] = *(const bf16x8*)(As + (a_row0 + i * 32 + r) * LDS_ROW_BYTES + kk * 32 + h * 16);
#pragma unroll
            for (int i = 0; i < 2; ++i)
                b[i] = *(const bf16x8*)(Bs + (b_row0 + i * 32 + r) * LDS_ROW_BYTES + kk * 32 + h * 16);
#pragma unroll
            for (int mi = 0; mi < 4; ++mi)
#pragma unroll
                for (int ni = 0; ni < 2; ++ni)
                    acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[mi], b[ni], acc[mi][ni], 0, 0, 0);
        }
    }
};

template <>
struct Mfma256<float> {
    static __device__ __forceinline__ void slab(const char* As, const char* Bs, int a_row0, int b_row0, int lane,
                                                f32x16 (&acc)[4][2]) {
        const int r = lane & 31, h = lane >> 5;
#pragma unroll
        for (int kk = 0; kk < 16; ++kk) {
            float a[4], b[2];
#pragma unroll
            for (int i = 0; i < 4; ++i)
                a[i] = *(const float*)(As + (a_row0 + i * 32 + r) * LDS_ROW_BYTES + (kk * 2 + h) * 4);
#pragma unroll
            for (int i = 0; i < 2; ++i)
                b[i] = *(const float*)(Bs + (b_row0 + i * 32 + r) * LDS_ROW_BYTES + (kk * 2 + h) * 4);
#pragma unroll
            for (int mi = 0; mi < 4; ++mi)
#pragma unroll
                for (int ni = 0; ni < 2; ++ni)
                    acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[mi], b[ni], acc[mi][ni], 0, 0, 0);
        }
    }
};

// Dense operand slab: rows [row0, row0+128) x K elements [k0, k0+KT) of a row-major matrix with
// leading dimension ld (elements).  Rows >= n_rows and K chunks >= k_total read as zero.
// 1024 16-byte chunks per slab, 4 per thread; 8 consecutive threads cover one 128-byte row.
// (Four named members, not an array: an array member indexed in an unrolled loop kept the whole
// struct in scratch memory in the larger kernels.)
template <typename T>
struct SlabRegs {
    uint4 v0, v1, v2, v3;
};

template <typename T>
__device__ __forceinline__ uint4 slab_chunk(const T* __restrict__ base, int64_t ld, int row0, int n_rows, int k0,
                                            int k_total, int c) {
    constexpr int EPC = 16 / (int)sizeof(T);  // elements per chunk
    const int row = row0 + (c >> 3);
    const int k = k0 + (c & 7) * EPC;
    if (row < n_rows && k < k_total) return *(const uint4*)(base + (int64_t)row * ld + k);
    return make_uint4(0, 0, 0, 0);
}

template <typename T>
__device__ __forceinline__ void slab_load(SlabRegs<T>& r, const T* __restrict__ base, int64_t ld, int row0,
                                          int n_rows, int k0, int k_total, int tid) {
    r.v0 = slab_chunk<T>(base, ld, row0, n_rows, k0, k_total, tid);
    r.v1 = slab_chunk<T>(base, ld, row0, n_rows, k0, k_total, tid + 256);
    r.v2 = slab_chunk<T>(base, ld, row0, n_rows, k0, k_total, tid + 512);
    r.v3 = slab_chunk<T>(base, ld, row0, n_rows, k0, k_total, tid + 768);
}

// Same, for callers that guarantee the whole slab is in bounds (rows clamped by the caller): no
// predicates, so the four loads are straight-line code and hipcc can wait for them with a COUNTED
// vmcnt instead of draining everything (predicated loads end up behind exec branches and force
// vmcnt(0), which serialises a two-deep prefetch on the L2 latency).
template <typename T>
__device__ __forceinline__ uint4 slab_chunk_fast(const T* __restrict__ base, int64_t ld, int row0, int row_max, int k0,
                                                 int c) {
    constexpr int EPC = 16 / (int)sizeof(T);
    const int row = min(row0 + (c >> 3), row_max);
    return *(const uint4*)(base + (int64_t)row * ld + k0 + (c & 7) * EPC);
}

template <typename T>
__device__ __forceinline__ void slab_load_fast(SlabRegs<T>& r, const T* __restrict__ base, int64_t ld, int row0,
                                               int row_max, int k0, int tid) {
    r.v0 = slab_chunk_fast<T>(base, ld, row0, row_max, k0, tid);
    r.v1 = slab_chunk_fast<T>(base, ld, row0, row_max, k0, tid + 256);
    r.v2 = slab_chunk_fast<T>(base, ld, row0, row_max, k0, tid + 512);
    r.v3 = slab_chunk_fast<T>(base, ld, row0, row_max, k0, tid + 768);
}

template <typename T>
__device__ __forceinline__ void slab_store(const SlabRegs<T>& r, char* lds, int tid) {
    *(uint4*)(lds + ((tid) >> 3) * LDS_ROW_BYTES + (tid & 7) * 16) = r.v0;
    *(uint4*)(lds + ((tid + 256) >> 3) * LDS_ROW_BYTES + (tid & 7) * 16) = r.v1;
    *(uint4*)(lds + ((tid + 512) >> 3) * LDS_ROW_BYTES + (tid & 7) * 16) = r.v2;
    *(uint4*)(lds + ((tid + 768) >> 3) * LDS_ROW_BYTES + (tid & 7) * 16) = r.v3;
}

// ------------------------------------------------------------------------------------------------
// Swizzled, unpadded operand image (weight-gradient kernel v2).  Rows are exactly 128 bytes (64 bf16
// or 32 f32 along K); the 16-byte chunk c of row r lives at slot c ^ ((r >> 1) & 7).  ds_read_b128
// serves 16-lane groups against 64 banks = two rows per bank line, and every group's rows
// ({0-3,12-15,20-27}, {4-11,16-19,28-31}, ...: MI355X_MICROARCH.md, LDS) split into 8 even + 8 odd
// rows whose (r >> 1) & 7 are all different, so a fragment read (lanes = rows, same chunk) is
// conflict-free without the 16 pad bytes - which is what lets a global_load_lds_dwordx4 fill the
// image directly: one wave-instruction writes 1 KB = 8 whole rows, lane i -> row i>>3, slot i&7,
// and the swizzle is applied to the per-lane SOURCE address instead (cdna guide, rule 21).
// ------------------------------------------------------------------------------------------------
#define SWZ_ROW_BYTES 128

__device__ __forceinline__ int swz_off(int row, int chunk) {
    return row * SWZ_ROW_BYTES + ((chunk ^ ((row >> 1) & 7)) << 4);
}

typedef __attribute__((address_space(3))) void lds_void_t;
typedef const __attribute__((address_space(1))) void glb_void_t;

// One LDS-DMA piece: 64 lanes x 16 bytes from per-lane global addresses to 1 KB of LDS at the
// wave-uniform address dst.  Inline asm rather than __builtin_amdgcn_global_load_lds: hipcc orders
// every LDS access that follows the builtin behind s_waitcnt vmcnt(0), which would land the whole
// slab before the MFMA phase instead of underneath it.  The asm load is invisible to hipcc's wait
// counting, so the kernel waits for it explicitly (dma_wait) before the barrier that publishes it.
// M0 (the DMA's LDS base) is compiler-reserved: saved and restored around the instruction.
__device__ __forceinline__ void glds16(const void* src, uint32_t lds_byte) {
    const uint32_t lds_addr = __builtin_amdgcn_readfirstlane(lds_byte);
    uint32_t keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep)
                 : "v"(src), "s"(lds_addr)
                 : "memory");
}
// SGPR base + 32-bit per-lane byte offset form: one VGPR per piece instead of a 64-bit address pair (the persistent GEMMs hold
// eight pieces' addresses across their whole tile loop: with pairs they ran out of registers)
__device__ __forceinline__ void glds16_s(const void* sbase, uint32_t voff, uint32_t lds_byte) {
    const uint32_t lds_addr = __builtin_amdgcn_readfirstlane(lds_byte);
    uint32_t keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep)
                 : "v"(voff), "s"(sbase), "s"(lds_addr)
                 : "memory");
}
// the same with the non-temporal policy: for operands every byte of which ONE workgroup reads once (a streamed [B, H] matrix)
__device__ __forceinline__ void glds16_nt(const void* src, uint32_t lds_byte) {
    const uint32_t lds_addr = __builtin_amdgcn_readfirstlane(lds_byte);
    uint32_t keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off nt\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep)
                 : "v"(src), "s"(lds_addr)
                 : "memory");
}
__device__ __forceinline__ void dma_wait() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }

// wave tile (32 MI) x 96 = MI x 3 MFMA tiles: MI = 3 is 144 accumulator VGPRs and 6 fragment reads
// per 9 MFMAs (two waves per SIMD); MI = 2 is 96 VGPRs and 5 reads per 6 MFMAs (three waves per SIMD)
template <typename T, int MI>
struct Mfma96;

template <int MI>
struct Mfma96<bf16_t, MI> {
    // between(kk) runs after the MFMAs of K step kk have been issued (kk = 0..2): the caller's
    // LDS-DMA pieces for the next chunk go there, in the shadow of the matrix pipe
    // RM = false: Bs is the K-contiguous image [column][64 k] (one ds_read_b128 per B fragment);
    // RM = true : Bs is the row-major image [column block of 64][64 k rows][128 bytes], chunk c of row r at slot
    //             c ^ rm_swz(r) (wsae_wgrad.hip): a B fragment = two ds_read_b64_tr_b16, each 4 consecutive k of this
    //             lane's column (lane 4 q + p of a 16-lane group addresses block row q, columns 4 p .. 4 p + 3).
    // ARM = true: As too is a row-major image [feature block of 64][64 k rows][128 bytes] (the dense left operand of the
    //             ReLU SAE's contractions, wgrad2d_kernel): A fragments by transposed reads as well.
    // PIPE = true : the four K steps as a software pipeline (below)
    template <bool RM = false, bool ARM = false, bool PIPE = false, typename F>
    static __device__ __forceinline__ void slab(const char* As, const char* Bs, int a_row0, int b_row0, int lane,
                                                f32x16 (&acc)[MI][3], F&& between) {
        const int r = lane & 31, h = lane >> 5;
        const int sw = (r >> 1) & 7;  // a_row0, b_row0 and the 32-row steps are multiples of 32: they do not change the swizzle
        const char* ap = As + (a_row0 + r) * SWZ_ROW_BYTES;
        const char* bp = Bs + (b_row0 + r) * SWZ_ROW_BYTES;
        const char* tp[3][2];   // RM: per column tile and 4-row block, the address for k step 0 (k steps are 2 KB apart)
        const char* tpa[MI][2];  // ARM: the same for the left operand's feature tiles
        typedef __attribute__((address_space(3))) bf16x4 lds_bf16x4_t;
        auto rm_addr = [&](const char* img, int col0, int q2) -> const char* {
            const int i = lane & 15, g1 = (lane >> 4) & 1;
            const int c0 = col0 + g1 * 16 + 4 * (i & 3);          // first of this lane's 4 address columns
            const int cidx = (c0 & 63) >> 3;                      // 16-byte chunk inside the 128-byte row
            const int row = 8 * h + 4 * q2 + (i >> 2);            // + 16 kk: does not change the swizzle
            const int swz = (((row >> 1) & 1) << 2) | ((row >> 2) & 3);
            return img + (c0 >> 6) * 8192 + row * 128 + ((cidx ^ swz) << 4) + 8 * (i & 1);
        };
        if constexpr (RM) {
#pragma unroll
            for (int ni = 0; ni < 3; ++ni)
#pragma unroll
                for (int q2 = 0; q2 < 2; ++q2) tp[ni][q2] = rm_addr(Bs, b_row0 + ni * 32, q2);
        }
        if constexpr (ARM) {
#pragma unroll
            for (int mi = 0; mi < MI; ++mi)
#pragma unroll
                for (int q2 = 0; q2 < 2; ++q2) tpa[mi][q2] = rm_addr(As, a_row0 + mi * 32, q2);
        }
        auto load_a = [&](int kk, int i) -> bf16x8 {
            if constexpr (ARM) {
                const bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4_t*)(tpa[i][0] + kk * 2048));
                const bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4_t*)(tpa[i][1] + kk * 2048));
                return __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
            } else {
                return *(const bf16x8*)(ap + i * 32 * SWZ_ROW_BYTES + (((kk * 2 + h) ^ sw) << 4));
            }
        };
        auto load_b = [&](int kk, int i) -> bf16x8 {
            if constexpr (RM) {
                const bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4_t*)(tp[i][0] + kk * 2048));
                const bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4_t*)(tp[i][1] + kk * 2048));
                return __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
            } else {
                return *(const bf16x8*)(bp + i * 32 * SWZ_ROW_BYTES + (((kk * 2 + h) ^ sw) << 4));
            }
        };
        if constexpr (PIPE) {
        // Software pipeline over the four K steps: the column fragments roll (b[ni] is reloaded for the next step as soon as its
        // MI MFMAs of this step are issued), the row fragments of the next step are requested behind the last column group -
        // all of it in front of between(kk), whose asm would otherwise keep every read of step kk + 1 behind it
        bf16x8 a[MI], b[3];
#pragma unroll
        for (int i = 0; i < MI; ++i) a[i] = load_a(0, i);
#pragma unroll
        for (int i = 0; i < 3; ++i) b[i] = load_b(0, i);
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
#pragma unroll
            for (int ni = 0; ni < 3; ++ni) {
#pragma unroll
                for (int mi = 0; mi < MI; ++mi)
                    acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[mi], b[ni], acc[mi][ni], 0, 0, 0);
                if (kk < 3) b[ni] = load_b(kk + 1, ni);
            }
            if (kk < 3) {
#pragma unroll
                for (int i = 0; i < MI; ++i) a[i] = load_a(kk + 1, i);
                between(kk);
            }
        }
        } else {
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
            bf16x8 a[MI], b[3];
#pragma unroll
            for (int i = 0; i < MI; ++i) a[i] = load_a(kk, i);
#pragma unroll
            for (int i = 0; i < 3; ++i) b[i] = load_b(kk, i);
#pragma unroll
            for (int mi = 0; mi < MI; ++mi)
#pragma unroll
                for (int ni = 0; ni < 3; ++ni)
                    acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[mi], b[ni], acc[mi][ni], 0, 0, 0);
            if (kk < 3) between(kk);
        }
        }
    }
    // row sums of the 16-row tile at row0 (+= over the chunk), every column of rs holds them
    static __device__ __forceinline__ void rowsum16(const char* As, int row0, int lane, f32x4& rs) {
        const int r = row0 + (lane & 15), q = lane >> 4;
        bf16x8 ones;
#pragma unroll
        for (int i = 0; i < 8; ++i) ones[i] = (bf16_t)1.0f;
#pragma unroll
        for (int k2 = 0; k2 < 2; ++k2) {
            const bf16x8 a = *(const bf16x8*)(As + swz_off(r, k2 * 4 + q));
            rs = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, ones, rs, 0, 0, 0);
        }
    }
};

// f32: a lane's 16-byte read holds 4 consecutive k; MFMA step j of read cc pairs element j of the
// h = 0 lanes (k = 8 cc + j) with element j of the h = 1 lanes (k = 8 cc + 4 + j).  A and B use the
// same k permutation, so the contraction is unchanged.
template <int MI>
struct Mfma96<float, MI> {
    template <bool RM = false, bool ARM = false, bool PIPE = false, typename F>  // (PIPE: accepted, not used)
    static __device__ __forceinline__ void slab(const char* As, const char* Bs, int a_row0, int b_row0, int lane,
                                                f32x16 (&acc)[MI][3], F&& between) {
        static_assert(!RM && !ARM, "row-major dense operands need 16-bit transposed LDS reads: bf16 only");
        const int r = lane & 31, h = lane >> 5;
        const int sw = (r >> 1) & 7;
        const char* ap = As + (a_row0 + r) * SWZ_ROW_BYTES;
        const char* bp = Bs + (b_row0 + r) * SWZ_ROW_BYTES;
#pragma unroll
        for (int cc = 0; cc < 4; ++cc) {
            const int co = ((cc * 2 + h) ^ sw) << 4;
            f32x4 a[MI], b[3];
#pragma unroll
            for (int i = 0; i < MI; ++i) a[i] = *(const f32x4*)(ap + i * 32 * SWZ_ROW_BYTES + co);
#pragma unroll
            for (int i = 0; i < 3; ++i) b[i] = *(const f32x4*)(bp + i * 32 * SWZ_ROW_BYTES + co);
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int mi = 0; mi < MI; ++mi)
#pragma unroll
                    for (int ni = 0; ni < 3; ++ni)
                        acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[mi][j], b[ni][j], acc[mi][ni], 0, 0, 0);
            if (cc < 3) between(cc);
        }
    }
    static __device__ __forceinline__ void rowsum16(const char* As, int row0, int lane, f32x4& rs) {
        const int r = row0 + (lane & 15), q = lane >> 4;
#pragma unroll
        for (int k2 = 0; k2 < 2; ++k2) {
            const f32x4 a = *(const f32x4*)(As + swz_off(r, k2 * 4 + q));
#pragma unroll
            for (int j = 0; j < 4; ++j) rs = __builtin_amdgcn_mfma_f32_16x16x4f32(a[j], 1.0f, rs, 0, 0, 0);
        }
    }
};

// 256 x 256 workgroup tile on the swizzled image: a wave owns 128 rows x 64 columns = 4 x 2 MFMA tiles (as Mfma256)
template <typename T>
struct Mfma256s;

template <>
struct Mfma256s<bf16_t> {
    static __device__ __forceinline__ void slab(const char* As, const char* Bs, int a_row0, int b_row0, int lane,
                                                f32x16 (&acc)[4][2]) {
        const int r = lane & 31, h = lane >> 5;
        const int sw = (r >> 1) & 7;
        const char* ap = As + (a_row0 + r) * SWZ_ROW_BYTES;
        const char* bp = Bs + (b_row0 + r) * SWZ_ROW_BYTES;
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
            const int co = ((kk * 2 + h) ^ sw) << 4;
            bf16x8 a[4], b[2];
#pragma unroll
            for (int i = 0; i < 4; ++i) a[i] = *(const bf16x8*)(ap + i * 32 * SWZ_ROW_BYTES + co);
#pragma unroll
            for (int i = 0; i < 2; ++i) b[i] = *(const bf16x8*)(bp + i * 32 * SWZ_ROW_BYTES + co);
#pragma unroll
            for (int mi = 0; mi < 4; ++mi)
#pragma unroll
                for (int ni = 0; ni < 2; ++ni)
                    acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[mi], b[ni], acc[mi][ni], 0, 0, 0);
        }
    }
};

template <>
struct Mfma256s<float> {
    static __device__ __forceinline__ void slab(const char* As, const char* Bs, int a_row0, int b_row0, int lane,
                                                f32x16 (&acc)[4][2]) {
        const int r = lane & 31, h = lane >> 5;
        const int sw = (r >> 1) & 7;
        const char* ap = As + (a_row0 + r) * SWZ_ROW_BYTES;
        const char* bp = Bs + (b_row0 + r) * SWZ_ROW_BYTES;
#pragma unroll
        for (int cc = 0; cc < 4; ++cc) {  // same k permutation in both operands (see Mfma96<float>)
            const int co = ((cc * 2 + h) ^ sw) << 4;
            f32x4 a[4], b[2];
#pragma unroll
            for (int i = 0; i < 4; ++i) a[i] = *(const f32x4*)(ap + i * 32 * SWZ_ROW_BYTES + co);
#pragma unroll
            for (int i = 0; i < 2; ++i) b[i] = *(const f32x4*)(bp + i * 32 * SWZ_ROW_BYTES + co);
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int mi = 0; mi < 4; ++mi)
#pragma unroll
                    for (int ni = 0; ni < 2; ++ni)
                        acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[mi][j], b[ni][j], acc[mi][ni], 0, 0, 0);
        }
    }
};

// OCP fp8 e4m3 operands (gfx950: v_mfma_f32_32x32x16_fp8_fp8, the bf16 form's rate on half the operand bytes).
// A 128-byte image row holds 128 K elements = 8 MFMA K steps; lane l supplies A[l & 31][8 (l >> 5) + j] as byte j of a
// 64-bit operand (profiles/tools/probe_fp8.hip), i.e. half (l >> 5) of 16-byte chunk kk.
struct fp8_t {
    uint8_t bits;
};

template <>
struct Mfma256s<fp8_t> {
    static __device__ __forceinline__ void slab(const char* As, const char* Bs, int a_row0, int b_row0, int lane,
                                                f32x16 (&acc)[4][2]) {
        const int r = lane & 31, h = lane >> 5;
        const int sw = (r >> 1) & 7;
        const char* ap = As + (a_row0 + r) * SWZ_ROW_BYTES + 8 * h;
        const char* bp = Bs + (b_row0 + r) * SWZ_ROW_BYTES + 8 * h;
#pragma unroll
        for (int kk = 0; kk < 8; ++kk) {
            const int co = (kk ^ sw) << 4;
            long a[4], b[2];
#pragma unroll
            for (int i = 0; i < 4; ++i) a[i] = *(const long*)(ap + i * 32 * SWZ_ROW_BYTES + co);
#pragma unroll
            for (int i = 0; i < 2; ++i) b[i] = *(const long*)(bp + i * 32 * SWZ_ROW_BYTES + co);
#pragma unroll
            for (int mi = 0; mi < 4; ++mi)
#pragma unroll
                for (int ni = 0; ni < 2; ++ni)
                    acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_fp8_fp8(a[mi], b[ni], acc[mi][ni], 0, 0, 0);
        }
    }
};

