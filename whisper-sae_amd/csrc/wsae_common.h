// Shared device/host helpers for libwsae_hip (gfx950 only).
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

#include "wsae.h"

// ------------------------------------------------------------------------------------------------
// host side: error plumbing + ctx definition
// ------------------------------------------------------------------------------------------------
void wsae_set_error(const char* fmt, ...);

#define WSAE_HIP_CHECK(expr)                                                              \
    do {                                                                                  \
        hipError_t e_ = (expr);                                                           \
        if (e_ != hipSuccess) {                                                           \
            wsae_set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), __FILE__, \
                           __LINE__);                                                     \
            return WSAE_ERR_HIP;                                                          \
        }                                                                                 \
    } while (0)

#define WSAE_REQUIRE(cond, ...)          \
    do {                                 \
        if (!(cond)) {                   \
            wsae_set_error(__VA_ARGS__); \
            return WSAE_ERR_INVALID;     \
        }                                \
    } while (0)

#define WSAE_LAUNCH_CHECK()                                                                  \
    do {                                                                                     \
        hipError_t e_ = hipGetLastError();                                                   \
        if (e_ != hipSuccess) {                                                              \
            wsae_set_error("kernel launch failed: %s (%s:%d)", hipGetErrorString(e_), __FILE__, \
                           __LINE__);                                                        \
            return WSAE_ERR_HIP;                                                             \
        }                                                                                    \
    } while (0)

typedef __bf16 bf16_t;
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));

static inline int64_t ceil_div64(int64_t a, int64_t b) { return (a + b - 1) / b; }
static inline int ceil_div(int a, int b) { return (a + b - 1) / b; }

// number of partial-sum slots every reduction in the library uses (one per producing block,
// blocks beyond it fold round-robin): fixed so that the reduction order is deterministic.
#define WSAE_MAX_PARTIALS 1024
// split-K factor of the weight-gradient contractions: one batch range per XCD
#define WSAE_WGRAD_MAX_SPLIT 8

struct wsae_prof {
    unsigned mask;                      // bit per kernel id
    int max_samples;
    int count[WSAE_K_COUNT];
    hipEvent_t* ev[WSAE_K_COUNT];       // 2*max_samples events per enabled kernel
};

struct wsae_ctx {
    wsae_prof prof;
    int D, H, K, maxB, prec, device;
    const float* relu_l1w;  // per-feature weights of the ReLU path's L1 term (nullable; caller-owned [H] floats)
    int loss_cols;        // columns the MSE averages over (= D; transcoders with a narrower output pad D and set this)
    int cus;              // compute units of `device` (persistent-kernel grid size)
    int comm_reserve;     // CUs the encoder half of a data-parallel backward leaves free for the collective running beside it
    int64_t P;            // flat pack element count
    int64_t off[5];       // W_e, W_dT, b_e, b_d, b_pre
    // ---- derived shadows -------------------------------------------------------------------
    bf16_t* We_bf16;      // [H][D]  (BF16 mode)
    bf16_t* WdT_bf16;     // [H][D]  bf16 shadow of W_dT (BF16 mode: decode / dpre gathers)
    float* c_fold;        // [H] folded encoder bias (BF16 mode)
    // ---- per-batch workspace ---------------------------------------------------------------
    void* xb;             // [maxB][D] staged batch in compute dtype (bf16 | f32 = x - b_pre)
    void* xT;             // [D][maxB] its transpose (B operand of the dW_e contraction)
    void* gT;             // [D][maxB] g = 2(recon-x)/(BD) transposed, compute dtype
    float* g;             // [maxB][D] fp32 g (dx path; the contraction operand in FP32 mode)
    bf16_t* gb;           // [maxB][D] bf16 g (BF16 mode: what the dW_d contraction and dh are fed)
    int g_is_bf16;        // 1 when the last decode launch left its g in gb (bucket transposes from there)
    int xT_valid;         // 1 when stage_batch left xT for this batch; 0: wsae_weight_grads transposes x itself
    int g32_valid;        // 1 when the last decode launch (also) left the fp32 g that wsae_input_grad reads
    float* pre;           // [maxB][H] pre-activation scratch (TopK input)
    float* smax;          // [maxB][H/16] maxima of the 16-column strips of pre (written by the persistent GEMM)
    int smax_valid;       // 1 when smax matches the pre the last dense GEMM wrote
    // selective strip stores (wsae_topk.h, "strip store threshold")
    uint32_t* tmin;       // three groups of minimum slots (wsae_topk.h, TG_GROUP_WORDS each) in rotation, then the count of rows that recomputed strips
    int tmin_cur;         // slot the current batch's TopK launch fills (the GEMM read the other two)
    int strip_predict;    // 1 (default): the encoder GEMM of encode_topk stores strips selectively
    float tg_fixed_s;     // > 0: fixed margin s of the store threshold instead of the adaptive one (WSAE_STRIP_SAFETY, experiments)
    int pred_valid;       // 1: pre holds only the strips above the threshold; the TopK launch must check rows against it
    const void* pred_x;   // that GEMM's A operand and row list (what a row recomputes its missing strips from)
    const int32_t* pred_rows;
    float* part_loss;     // [WSAE_MAX_PARTIALS]
    float* part_l0;       // [WSAE_MAX_PARTIALS]
    float* part_dbd;      // [WSAE_MAX_PARTIALS][D]
    float* part_sq;       // [WSAE_MAX_PARTIALS]
    float* colnorm;       // [H] decoder column sum of squares scratch
    float* wg_slabs;      // [WSAE_WGRAD_MAX_SPLIT][2*H*D] split-K partial weight gradients
    float* dbe_slab;      // [WSAE_WGRAD_MAX_SPLIT][H]
    float* dbpre_part;    // [ceil(H/32)][D]
    uint32_t* ent_pos;    // [maxB*K] bucketed compact code: (feature & 127) << 16 | row-in-chunk
    void* ent_hid;        // [maxB*K] relu(value) in the contraction dtype, bucket order
    void* ent_dpre;       // [maxB*K] dpre, bucket order
    int32_t* ent_off;     // [ceil(maxB/32)][ceil(H/128)+1] bucket boundaries
    int ent_valid;        // 1: the last decode launch left the bucketed code of (ent_vals, ent_B) itself (chunked form)
    const float* ent_vals;
    int ent_B;
    const float* wire_dec_vals;  // batch whose decoder half wsae_weight_grads_wire has just run (the encoder half must follow it)
    int wire_dec_B;
    int32_t* counters;    // small int scratch (fallback rows, resample cursors; [16..) = arrival tickets, 8-byte aligned)
    int32_t* dead_list;   // [H] compacted dead feature indices (resample)
    int32_t* row_order;   // [maxB] rows sorted by error (resample)
    int relu_fp8;         // 1: the ReLU SAE's two forward GEMMs run on fp8 e4m3 operands (wsae_ctx_set_relu_fp8)
    void* relu_ws;        // ReLU-SAE workspace (wsae_relu.hip), allocated by wsae_ctx_reserve_relu
    int relu_x_B;         // batch size of the last ReLU forward that ran the row-major-GEMM flow (its bf16 hidden is in relu_ws); 0 = none
    int relu_g_B;         // ... and whose residual pass left g (gb) and the db_d partials for the backward; 0 = none
    float* fired;         // caller-owned [H] indicator buffer for the DDP dead-feature clock, or null
    const float* wire_metrics;  // caller-owned (loss, l0) of the step, encoded behind the indicators on the wire, or null
    int n_dec_blocks;     // blocks used by the last decode launch (partials to reduce)
    int n_sq_parts;       // global-norm partials left in part_sq by the last wsae_weight_grads
    float* dbd2;          // [64][D] level-1 reduction of part_dbd
    size_t ws_bytes;
};

// bracket one kernel launch with events when profiling is enabled for its id
#define WSAE_PROF_BEGIN(ctx, id, st)                                                          \
    const bool prof_on_##id = ((ctx)->prof.mask >> (id)) & 1u &&                               \
                              (ctx)->prof.count[id] < (ctx)->prof.max_samples;                 \
    if (prof_on_##id) (void)hipEventRecord((ctx)->prof.ev[id][2 * (ctx)->prof.count[id]], st)
#define WSAE_PROF_END(ctx, id, st)                                                            \
    if (prof_on_##id) {                                                                       \
        (void)hipEventRecord((ctx)->prof.ev[id][2 * (ctx)->prof.count[id] + 1], st);           \
        (ctx)->prof.count[id]++;                                                              \
    }

// ------------------------------------------------------------------------------------------------
// device helpers
// ------------------------------------------------------------------------------------------------
#ifdef __HIPCC__

#define WSAE_WAVE 64

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

// butterfly reductions over the 64 lanes (every lane gets the result; pair order 32, 16, .., 1 - the order the
// __shfl_xor form had, so the sums are bit-identical to it) on the lane exchanges above: six ds_bpermute round trips
// of ~100+ cycles each were most of a wave's tail in the one-round kernels (update_rows, grad_finish, the decode epilogue)
__device__ __forceinline__ float wave_sum(float v) {
    const int lane = threadIdx.x & 63;
    v += __uint_as_float(lane_xor_u32<32>(__float_as_uint(v), lane));
    v += __uint_as_float(lane_xor_u32<16>(__float_as_uint(v), lane));
    v += __uint_as_float(lane_xor_u32<8>(__float_as_uint(v), lane));
    v += __uint_as_float(lane_xor_u32<4>(__float_as_uint(v), lane));
    v += __uint_as_float(lane_xor_u32<2>(__float_as_uint(v), lane));
    v += __uint_as_float(lane_xor_u32<1>(__float_as_uint(v), lane));
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
    const int lane = threadIdx.x & 63;
    v = fmaxf(v, __uint_as_float(lane_xor_u32<32>(__float_as_uint(v), lane)));
    v = fmaxf(v, __uint_as_float(lane_xor_u32<16>(__float_as_uint(v), lane)));
    v = fmaxf(v, __uint_as_float(lane_xor_u32<8>(__float_as_uint(v), lane)));
    v = fmaxf(v, __uint_as_float(lane_xor_u32<4>(__float_as_uint(v), lane)));
    v = fmaxf(v, __uint_as_float(lane_xor_u32<2>(__float_as_uint(v), lane)));
    v = fmaxf(v, __uint_as_float(lane_xor_u32<1>(__float_as_uint(v), lane)));
    return v;
}
__device__ __forceinline__ int wave_sum_i(int v) {
    const int lane = threadIdx.x & 63;
    v += (int)lane_xor_u32<32>((uint32_t)v, lane);
    v += (int)lane_xor_u32<16>((uint32_t)v, lane);
    v += (int)lane_xor_u32<8>((uint32_t)v, lane);
    v += (int)lane_xor_u32<4>((uint32_t)v, lane);
    v += (int)lane_xor_u32<2>((uint32_t)v, lane);
    v += (int)lane_xor_u32<1>((uint32_t)v, lane);
    return v;
}

// deterministic block reduction of one float per thread (blockDim.x multiple of 64, <= 1024);
// result valid in thread 0.  `scratch` must hold blockDim.x/64 floats.
__device__ __forceinline__ float block_sum(float v, float* scratch) {
    v = wave_sum(v);
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, nw = blockDim.x >> 6;
    __syncthreads();
    if (lane == 0) scratch[w] = v;
    __syncthreads();
    float r = 0.f;
    if (threadIdx.x == 0)
        for (int i = 0; i < nw; ++i) r += scratch[i];
    return r;
}

// load one activation element as float
// Two-level arrival ticket.  Exactly one caller per grid - the last block to arrive - gets true;
// `payload` values (their grid sum must stay below 2^32) are added up on the way and handed to that
// block in *total.  One thread per block calls this after the block's own work is complete.
// 768 blocks hitting ONE address serialise in the memory-side atomic unit (~9 ns each: 7 us of a
// 25 us kernel); here a block bumps one of TICKET_GROUPS group words and only the last arriver of a
// group bumps the grid word, so no address sees more than max(grid / groups, groups) atomics.
// t: 1 + TICKET_GROUPS zeroed 64-bit words, left zeroed again for the next launch.
#define TICKET_GROUPS 32
#define TICKET_WORDS (1 + TICKET_GROUPS)
__device__ __forceinline__ bool grid_ticket(unsigned long long* t, unsigned payload, unsigned* total) {
    const unsigned nblk = gridDim.x * gridDim.y;
    const unsigned bid = blockIdx.y * gridDim.x + blockIdx.x;
    const unsigned ngrp = nblk < TICKET_GROUPS ? nblk : TICKET_GROUPS;
    const unsigned grp = bid % ngrp;
    const unsigned gsz = (nblk - grp + ngrp - 1) / ngrp;
    const unsigned long long add = (1ull << 32) | payload;
    const unsigned long long o1 = __hip_atomic_fetch_add(t + 1 + grp, add, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if ((unsigned)(o1 >> 32) != gsz - 1) return false;
    __hip_atomic_store(t + 1 + grp, 0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const unsigned long long add2 = (1ull << 32) | (unsigned)((o1 + add) & 0xFFFFFFFFull);
    const unsigned long long o2 = __hip_atomic_fetch_add(t, add2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if ((unsigned)(o2 >> 32) != ngrp - 1) return false;
    __hip_atomic_store(t, 0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    *total = (unsigned)((o2 + add2) & 0xFFFFFFFFull);
    return true;
}

// 16-byte load / store with the non-temporal hint: data used once, which should leave the Infinity Cache to what is used again
__device__ __forceinline__ float4 nt_load4(const float4* p) {
    const f32x4 v = __builtin_nontemporal_load((const f32x4*)p);
    return make_float4(v[0], v[1], v[2], v[3]);
}
__device__ __forceinline__ void nt_store4(float4* p, const float4& a) {
    f32x4 v;
    v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w;
    __builtin_nontemporal_store(v, (f32x4*)p);
}

template <int DT>
__device__ __forceinline__ float load_act(const void* p, int64_t i) {
    if (DT == WSAE_DT_BF16) return (float)((const bf16_t*)p)[i];
    return ((const float*)p)[i];
}

#endif  // __HIPCC__

// internal (wsae_encode.hip): stage the batch (xb, xT) and run the dense encoder GEMM into pre [B][H]
int wsae_internal_stage(wsae_ctx* ctx, const float* params, const void* x, int x_dtype, const int32_t* rows, int B, hipStream_t st);
int wsae_internal_stage_rows(wsae_ctx* ctx, const float* params, const void* x, int x_dtype, const int32_t* rows, int B, hipStream_t st);
int wsae_internal_stage_and_gemm(wsae_ctx* ctx, const float* params, const void* x, int x_dtype, const int32_t* rows,
                                 int B, float* pre, int64_t* step_count, int direct, hipStream_t st, bool predicate = false);
// internal (wsae_encode.hip): the standalone TopK launch over ctx->pre; whether the strip-guided form applies
int wsae_internal_topk(wsae_ctx* ctx, int B, float* vals, int32_t* idx, int32_t* fb, hipStream_t st);
int wsae_internal_encode_topk(wsae_ctx* ctx, const float* params, const void* x, int x_dtype, const int32_t* rows, int B,
                              float* vals, int32_t* idx, int64_t* step_count, int32_t* fb, hipStream_t st);
bool wsae_internal_strips_ok(const wsae_ctx* ctx);
// internal (wsae_decode_mfma.hip): the MFMA decode kernel (BF16 mode)
bool wsae_internal_decode_mfma_ok(const wsae_ctx* c);
int wsae_internal_decode_mfma(wsae_ctx* c, const float* params, const void* x, int x_dtype, const int32_t* rows,
                              const float* vals, const int32_t* idx, int B, float* recon, int want_bwd, float* dpre,
                              int want_g32, int64_t* last_activated, const int64_t* step_count, wsae_stats* stats,
                              hipStream_t st);
// internal (wsae_encode.hip): the persistent LDS-DMA NT GEMM for other dense contractions; false = shape not supported
bool wsae_internal_gemm256d_fp8(wsae_ctx* c, const void* A, int64_t lda, const void* Bt, int64_t ldb, const float* rscale,
                                const float* cscale, const float* bias, float* C, int64_t ldc, int M, int N, int K,
                                hipStream_t st);
bool wsae_internal_gemm256d(wsae_ctx* c, const void* A, int64_t lda, const void* Bt, int64_t ldb, const float* bias, float* C,
                            int64_t ldc, int M, int N, int K, int nsplit, int64_t cz, hipStream_t st);

// internal (wsae_gemm256x.hip): persistent bf16 GEMM with row-major-in-K operands and fused epilogues (the ReLU SAE's dense path)
enum { GX_EPI_PLAIN = 0, GX_EPI_RELU = 1, GX_EPI_DPRE = 2 };
struct GxEpi {
    const float* bias;     // PLAIN / RELU: + bias[n] (nullable; z = 0 only)
    float* c;              // PLAIN: fp32 C [M][ldc] (+ z * cz);  RELU: optional fp32 copy of hidden (nullable)
    int64_t ldc, cz;
    bf16_t* out16;         // RELU: hidden bf16 [M][ld16];  DPRE: dpre bf16 [M][ld16]
    uint64_t* bits;        // RELU (out) / DPRE (in): [N / 64][ldbits] words (column-major: ldbits >= M rows per word column), one
                           // per (row, 64-column span of a wave tile): bit 16 c + j = (hidden > 0) at column 64 w + 4 j + c - the
                           // dpre mask at 1/16 of the bytes of re-reading bf16 hidden
    int64_t ldbits;
    int64_t ld16;
    const float* colw;     // RELU / DPRE: per-column weights of the L1 term (nullable = 1)
    float l1;              // DPRE: sparsity_weight / (B H)
    float* part;           // RELU: part[tile] = sum relu * w, part[nslots + tile] = count(relu > 0)
    int nslots;
    float* colpart;        // DPRE: colpart[(m0 / 128 + wm)][n] = column sums of dpre over 128 rows
    const float* rscale;   // fp8 operands: C = rscale[m] * cscale[n] * acc (+ bias) - the per-row dequantisation scales (nullable)
    const float* cscale;
    float* rowmax;         // RELU (optional): [N / 64][ldbits] maxima of relu over each row's 64-column span (the next GEMM's
                           // per-row quantisation scale comes from them without another pass over hidden)
};
bool wsae_internal_gemm256x(wsae_ctx* c, int a_rm, int b_rm, int epi, const void* A, int64_t lda, const void* Bm, int64_t ldb,
                            int M, int N, int K, int nsplit, const GxEpi& e, hipStream_t st, int fp8 = 0);
// internal (wsae_wgrad.hip): the ReLU SAE's two contractions on the 192 x 384 geometry + slab reduction; 0 = shape not served
int wsae_internal_relu_wgrad(wsae_ctx* ctx, const void* hid, const void* dpre, const void* xb, const void* gb, int B, float* grads,
                             hipStream_t st);
