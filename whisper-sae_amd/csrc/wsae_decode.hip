// Sparse decode + MSE residual + first half of backward, from the compact TopK code.
//   reference: TopKSAE.decode / forward / _update_dead_features, model.py:120-181, and the
//   autograd of them: g = 2(recon-x)/(BD), dh = g W_d, dpre = dh * 1[v>0].
//
// One wave per batch row.  A row of W_dT (one decoder column) is D contiguous floats, so the
// k gathers per row are coalesced; lane l owns elements l, l+64, ... of the row.
#include "wsae_common.h"
#include "wsae_decode_epilogue.h"

// Generic shapes (any D <= 2048 multiple of 4, any K <= 128): lane l owns the 4-element chunks
// (l + 64 c) * 4 .. + 3, c < NCH, of a row (8- or 16-byte loads, 512 / 1024 contiguous bytes per wave
// instruction), and the selected feature rows are gathered DEC_JB at a time so that DEC_JB * NCH loads are in
// flight before the first FMA waits (the first version of this kernel gathered one row at a time with 2-byte
// loads: 1.8 ms at 768 -> 12288, k = 64, B = 8192).
#define DEC_JB 8

template <typename TW>
__device__ __forceinline__ float4 load_chunk4(const TW* p) {
    if constexpr (sizeof(TW) == 2) {
        const bf16x4 t = *(const bf16x4*)p;
        return make_float4((float)t[0], (float)t[1], (float)t[2], (float)t[3]);
    } else {
        return *(const float4*)p;
    }
}

template <typename TW, int NCH, int XDT, bool BWD, bool ROUND_DPRE>
__global__ void __launch_bounds__(256)
decode_kernel(const TW* __restrict__ WdT, const float* __restrict__ bd, const float* __restrict__ bpre,
              const void* __restrict__ x, const int32_t* __restrict__ rows, const float* __restrict__ vals,
              const int32_t* __restrict__ idx, int B, int D, int K, float* __restrict__ recon_out,
              float* __restrict__ dpre, float* __restrict__ g_out, int64_t* __restrict__ last_activated, float* __restrict__ fired,
              const int64_t* __restrict__ step_count, float* __restrict__ part_loss, float* __restrict__ part_l0,
              float* __restrict__ part_dbd, int32_t* __restrict__ ticket, wsae_stats* __restrict__ stats, int loss_cols) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* red = (float*)smem;          // [8] block reduction scratch
    float* dbd_s = (float*)smem + 8;    // [4][D]
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const float scale = 2.0f / ((float)B * (float)loss_cols);
    const int64_t step = (last_activated && step_count) ? *step_count : 0;

    // chunk c of this lane starts at column dcol[c]; chunks past D are clamped for loads and masked for results
    int dcol[NCH];
    bool dok[NCH];
    float4 bsum[NCH], dbd[NCH];
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
        const int d = (lane + 64 * c) * 4;
        dok[c] = d < D;
        dcol[c] = dok[c] ? d : 0;
        const float4 a = *(const float4*)(bd + dcol[c]), p = *(const float4*)(bpre + dcol[c]);
        bsum[c] = dok[c] ? make_float4(a.x + p.x, a.y + p.y, a.z + p.z, a.w + p.w) : make_float4(0.f, 0.f, 0.f, 0.f);
        dbd[c] = make_float4(0.f, 0.f, 0.f, 0.f);
    }
    float loss_acc = 0.f;
    int l0_acc = 0;

    for (int b = blockIdx.x * 4 + wave; b < B; b += gridDim.x * 4) {
        float4 rec[NCH];
#pragma unroll
        for (int c = 0; c < NCH; ++c) rec[c] = bsum[c];
        const int64_t code = (int64_t)b * K;
        // ---- decode: recon = sum_j relu(v_j) * W_dT[idx_j, :] ----
        for (int jb = 0; jb < K; jb += 64) {
            const int nj = min(64, K - jb);
            const float v = (lane < nj) ? vals[code + jb + lane] : 0.f;
            const int f = (lane < nj) ? idx[code + jb + lane] : 0;
            const bool on = v > 0.f;
            l0_acc += __popcll(__ballot(on));
            if (on && last_activated) {
                last_activated[f] = step;  // model.py:178-181 (same value from every writer)
                if (fired) fired[f] = 1.f;
            }
            const float vr = on ? v : 0.f;  // negative winners decode as zero (model.py:116)
            for (int j0 = 0; j0 < nj; j0 += DEC_JB) {
                float4 w[DEC_JB][NCH];
                float vj[DEC_JB];
#pragma unroll
                for (int t = 0; t < DEC_JB; ++t) {
                    const int j = min(j0 + t, nj - 1);
                    vj[t] = (j0 + t < nj) ? __uint_as_float(__builtin_amdgcn_readlane(__float_as_uint(vr), j)) : 0.f;
                    const TW* row = WdT + (int64_t)__builtin_amdgcn_readlane(f, j) * D;
#pragma unroll
                    for (int c = 0; c < NCH; ++c) w[t][c] = load_chunk4<TW>(row + dcol[c]);
                }
#pragma unroll
                for (int t = 0; t < DEC_JB; ++t)
#pragma unroll
                    for (int c = 0; c < NCH; ++c) {
                        rec[c].x = fmaf(vj[t], w[t][c].x, rec[c].x);
                        rec[c].y = fmaf(vj[t], w[t][c].y, rec[c].y);
                        rec[c].z = fmaf(vj[t], w[t][c].z, rec[c].z);
                        rec[c].w = fmaf(vj[t], w[t][c].w, rec[c].w);
                    }
            }
        }
        // ---- residual, loss, g ----
        const int64_t src = rows ? (int64_t)rows[b] : (int64_t)b;
        float4 g[NCH];
#pragma unroll
        for (int c = 0; c < NCH; ++c) {
            g[c] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (dok[c]) {
                const int64_t o = src * D + dcol[c];
                const float r0 = rec[c].x - load_act<XDT>(x, o), r1 = rec[c].y - load_act<XDT>(x, o + 1);
                const float r2 = rec[c].z - load_act<XDT>(x, o + 2), r3 = rec[c].w - load_act<XDT>(x, o + 3);
                loss_acc += (r0 * r0 + r1 * r1) + (r2 * r2 + r3 * r3);
                g[c] = make_float4(r0 * scale, r1 * scale, r2 * scale, r3 * scale);
                dbd[c].x += g[c].x; dbd[c].y += g[c].y; dbd[c].z += g[c].z; dbd[c].w += g[c].w;
                if (recon_out) *(float4*)(recon_out + (int64_t)b * D + dcol[c]) = rec[c];
                if (BWD) *(float4*)(g_out + (int64_t)b * D + dcol[c]) = g[c];
                if (BWD && sizeof(TW) == 2) {  // bf16 mode: dh is taken with bf16(g), like decode_fast_kernel and the oracle's "amp" mode
                    g[c].x = (float)(bf16_t)g[c].x; g[c].y = (float)(bf16_t)g[c].y;
                    g[c].z = (float)(bf16_t)g[c].z; g[c].w = (float)(bf16_t)g[c].w;
                }
            }
        }
        // ---- dpre_j = (v_j > 0) ? g . W_dT[idx_j, :] : 0 ----
        if (BWD) {
            for (int jb = 0; jb < K; jb += 64) {
                const int nj = min(64, K - jb);
                const float v = (lane < nj) ? vals[code + jb + lane] : 0.f;
                const int f = (lane < nj) ? idx[code + jb + lane] : 0;
                float mine = 0.f;
                for (int j0 = 0; j0 < nj; j0 += DEC_JB) {
                    float4 w[DEC_JB][NCH];
#pragma unroll
                    for (int t = 0; t < DEC_JB; ++t) {
                        const TW* row = WdT + (int64_t)__builtin_amdgcn_readlane(f, min(j0 + t, nj - 1)) * D;
#pragma unroll
                        for (int c = 0; c < NCH; ++c) w[t][c] = load_chunk4<TW>(row + dcol[c]);
                    }
#pragma unroll
                    for (int t = 0; t < DEC_JB; ++t) {
                        float dot = 0.f;
#pragma unroll
                        for (int c = 0; c < NCH; ++c) {  // (chunks past D carry g = 0)
                            dot = fmaf(g[c].x, w[t][c].x, dot); dot = fmaf(g[c].y, w[t][c].y, dot);
                            dot = fmaf(g[c].z, w[t][c].z, dot); dot = fmaf(g[c].w, w[t][c].w, dot);
                        }
                        dot = wave_sum(dot);
                        if (lane == j0 + t) mine = dot;
                    }
                }
                if (!(v > 0.f)) mine = 0.f;
                if (ROUND_DPRE) mine = (float)(bf16_t)mine;  // what the MFMA contraction will be fed
                if (lane < nj) dpre[code + jb + lane] = mine;
            }
        }
    }

    if (BWD) {
        __syncthreads();
#pragma unroll
        for (int c = 0; c < NCH; ++c)
            if (dok[c]) *(float4*)(dbd_s + wave * D + dcol[c]) = dbd[c];
        __syncthreads();
    }
    __shared__ int last_flag;
    decode_block_epilogue<BWD>(loss_acc, l0_acc, dbd_s, D, B, loss_cols, red, &last_flag, part_loss, part_l0, part_dbd, ticket, stats);
}

// ------------------------------------------------------------------------------------------------
// Fast path (K even <= 64, D = 32*EPL): still one wave per batch row, but
//   * the two 32-lane halves of the wave work on two different selected features at once; lane li
//     of a half owns the EPL contiguous elements [EPL*li, EPL*(li+1)) of a decoder row, fetched as
//     8/16-byte loads (a whole 768-byte bf16 row of W_dT per half-wave instruction group);
//   * the K gathered rows stay in registers (packed bf16) between the decode pass and the dpre
//     pass, so every W_dT row is read once per (batch row, selected feature);
//   * the K/2 per-half dot products are reduced with a transposing butterfly (15 shuffles for 16
//     values instead of 16 x 5).
// W_dT is the bf16 shadow in BF16 mode (2.36 MB at cfg2: resident in every XCD's 4 MB L2) and the
// fp32 master in FP32 mode.
// ------------------------------------------------------------------------------------------------
// A lane's EPL elements of a D = 32*EPL row are taken CHUNK-INTERLEAVED: load c of lane li covers
// bytes [CS*(32*c + li), +CS), so every load instruction of a half-wave reads 32*CS contiguous bytes
// (whole 128-byte lines).  (A lane-contiguous mapping -- 24 bytes per lane as 3 x 8 -- made each
// instruction span all six lines of the row: 3x the L2 requests, and the kernel was L2-request bound.)
template <typename TW, int EPL>
struct RowSeg {
    static constexpr int NB = EPL * (int)sizeof(TW);                       // bytes per lane
    static constexpr int CS = (NB % 16 == 0) ? 16 : (NB % 8 == 0) ? 8 : 4;  // load width
    static constexpr int NW = NB / 4;                                       // dwords per lane
    static constexpr int EPC = CS / (int)sizeof(TW);                        // elements per chunk
    uint32_t w[NW];
    // element index (within the row) of this lane's e-th element
    static __device__ __forceinline__ int elem(int li, int e) { return (32 * (e / EPC) + li) * EPC + (e % EPC); }
    __device__ __forceinline__ void load(const TW* row, int li) {
        const char* p = (const char*)row + CS * li;
#pragma unroll
        for (int c = 0; c < NB / CS; ++c) {
            if (CS == 16) {
                const uint4 v = *(const uint4*)(p + c * 32 * CS);
                w[4 * c] = v.x; w[4 * c + 1] = v.y; w[4 * c + 2] = v.z; w[4 * c + 3] = v.w;
            } else if (CS == 8) {
                const uint2 v = *(const uint2*)(p + c * 32 * CS);
                w[2 * c] = v.x; w[2 * c + 1] = v.y;
            } else {
                w[c] = *(const uint32_t*)(p + c * 32 * CS);
            }
        }
    }
    __device__ __forceinline__ float get(int e) const {
        if (sizeof(TW) == 4) return __uint_as_float(w[e]);
        const uint32_t u = w[e >> 1];
        return __uint_as_float((e & 1) ? (u & 0xFFFF0000u) : (u << 16));
    }
};

// Cross-lane exchanges without the LDS pipe (__shfl / __shfl_xor compile to ds_bpermute_b32; this kernel
// issued 85 of them per row).  gfx950 has VALU forms for every exchange needed here
// (profiles/tools/lane_ops_probe.hip prints what each delivers):
//   xor 32 / xor 16 : v_permlane32_swap / v_permlane16_swap on the pair (a, b) give a' + b' =
//                     (a + a^M) on the lanes with bit M clear and (b + b^M) on the others - exactly one
//                     butterfly step, no selects;
//   xor 8           : DPP row_ror:8;   xor 4: DPP row_shl:4 / row_shr:4 by lane parity;
//   xor 2 / xor 1   : DPP quad_perm.
// They cost no time by themselves (the kernel is latency bound) but 28 fewer VGPRs (216 -> 188).
template <int CTRL>
__device__ __forceinline__ float dpp_mov(float x) {
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), CTRL, 0xF, 0xF, false));
}

template <int M>
__device__ __forceinline__ float lane_xor(float x, int li) {  // x of lane (l ^ M), M in {1, 2, 4, 8}
    if constexpr (M == 1) return dpp_mov<0xB1>(x);        // quad_perm [1,0,3,2]
    else if constexpr (M == 2) return dpp_mov<0x4E>(x);   // quad_perm [2,3,0,1]
    else if constexpr (M == 4) {
        // both DPP moves run with every lane active, THEN the select: inside a ternary each would run under a
        // partial EXEC mask, and a DPP source lane that is masked off reads as invalid
        const float up = dpp_mov<0x104>(x), dn = dpp_mov<0x114>(x);  // row_shl:4 (lane l + 4), row_shr:4 (lane l - 4)
        return (li & 4) ? dn : up;
    } else return dpp_mov<0x128>(x);                       // row_ror:8
}

// one butterfly step on the pair (a, b): lanes with bit M clear get a + a^M, the others b + b^M
template <int M>
__device__ __forceinline__ float pair_step(float a, float b, int li) {
    if constexpr (M == 32) {
        const auto r = __builtin_amdgcn_permlane32_swap(__float_as_int(a), __float_as_int(b), false, false);
        return __int_as_float(r[0]) + __int_as_float(r[1]);
    } else if constexpr (M == 16) {
        const auto r = __builtin_amdgcn_permlane16_swap(__float_as_int(a), __float_as_int(b), false, false);
        return __int_as_float(r[0]) + __int_as_float(r[1]);
    } else {
        const float sa = a + lane_xor<M>(a, li), sb = b + lane_xor<M>(b, li);
        return (li & M) ? sb : sa;
    }
}

template <int KJ, int N, int M>
__device__ __forceinline__ void bfly(float (&pd)[KJ], int li) {
    if constexpr (M >= 1) {
        if constexpr (N >= 1) {
#pragma unroll
            for (int i = 0; i < N; ++i) pd[i] = pair_step<M>(pd[i], pd[i + N], li);
            bfly<KJ, N / 2, M / 2>(pd, li);
        } else {
            pd[0] = pair_step<M>(pd[0], pd[0], li);  // plain all-reduce over the remaining masks
            bfly<KJ, 0, M / 2>(pd, li);
        }
    }
}

template <typename TW, int EPL, int KJ, int XDT, bool BWD, bool ROUND_DPRE>
__global__ void __launch_bounds__(256, 2)
decode_fast_kernel(const TW* __restrict__ WdT, const float* __restrict__ bd, const float* __restrict__ bpre,
                   const void* __restrict__ x, const int32_t* __restrict__ rows, const float* __restrict__ vals,
                   const int32_t* __restrict__ idx, int B, int K, float* __restrict__ recon_out,
                   float* __restrict__ dpre, float* __restrict__ g_out, int64_t* __restrict__ last_activated, float* __restrict__ fired,
                   const int64_t* __restrict__ step_count, float* __restrict__ part_loss,
                   float* __restrict__ part_l0, float* __restrict__ part_dbd, int32_t* __restrict__ ticket,
                   wsae_stats* __restrict__ stats, int loss_cols) {
    constexpr int D = 32 * EPL;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* red = (float*)smem;        // [8]
    float* dbd_s = (float*)smem + 8;  // [4][D]
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int half = lane >> 5, li = lane & 31;
    const float scale = 2.0f / ((float)B * (float)loss_cols);
    const int64_t step = (last_activated && step_count) ? *step_count : 0;

    // b_d + b_pre and the per-wave column sums of g live in LDS, not in registers: the gathered rows
    // (96 packed registers at 384/k=32) need the room
    float* bsum_s = dbd_s + 4 * D;  // [D]
    for (int d = threadIdx.x; d < D; d += 256) {
        bsum_s[d] = bd[d] + bpre[d];
        dbd_s[d] = 0.f; dbd_s[D + d] = 0.f; dbd_s[2 * D + d] = 0.f; dbd_s[3 * D + d] = 0.f;
    }
    __syncthreads();
    float loss_acc = 0.f;
    int l0_acc = 0;

    // software pipeline over the wave's rows: the compact code and the x row of the NEXT row are requested
    // before the current row's gathers, so a row costs one exposed L2 round trip (the gathers), not three
    using Seg = RowSeg<TW, EPL>;
    constexpr int NCH = Seg::NB / Seg::CS;  // chunks per lane; a chunk is Seg::EPC consecutive elements
    // the next row of x is requested a phase ahead and kept RAW (packed bf16 when the input is bf16): converting at
    // load time would put the wait for this load - two dependent round trips when rows[] is used - right behind its issue
    constexpr int XW = (XDT == WSAE_DT_BF16) ? EPL / 2 : EPL;  // dwords per lane
    constexpr bool XVEC = Seg::EPC == 4;                        // 8-byte (bf16) / 16-byte (f32) chunk loads
    auto load_x = [&](int bb, uint32_t (&xw)[XW]) {
        const int64_t src = rows ? (int64_t)rows[bb] : (int64_t)bb;
#pragma unroll
        for (int c = 0; c < NCH; ++c) {
            const int64_t o = src * D + (32 * c + li) * Seg::EPC;
            if constexpr (XDT == WSAE_DT_BF16 && XVEC) {
                const uint2 t = *(const uint2*)((const bf16_t*)x + o);
                xw[2 * c] = t.x; xw[2 * c + 1] = t.y;
            } else if constexpr (XDT == WSAE_DT_F32 && XVEC) {
                const uint4 t = *(const uint4*)((const float*)x + o);
                xw[4 * c] = t.x; xw[4 * c + 1] = t.y; xw[4 * c + 2] = t.z; xw[4 * c + 3] = t.w;
            } else if constexpr (XDT == WSAE_DT_BF16) {  // EPC == 2: one dword = two bf16
                static_assert(Seg::EPC == 2 || XVEC, "bf16 rows come in chunks of 2 or 4 elements");
                xw[c] = *(const uint32_t*)((const bf16_t*)x + o);
            } else {
#pragma unroll
                for (int q = 0; q < Seg::EPC; ++q) xw[c * Seg::EPC + q] = *(const uint32_t*)((const float*)x + o + q);
            }
        }
    };
    auto x_at = [&](const uint32_t (&xw)[XW], int e) -> float {  // element e of this lane's share of the row
        if constexpr (XDT == WSAE_DT_BF16) {
            const uint32_t u = xw[e >> 1];
            return __uint_as_float((e & 1) ? (u & 0xFFFF0000u) : (u << 16));
        } else {
            return __uint_as_float(xw[e]);
        }
    };
    const int b_first = blockIdx.x * 4 + wave, b_step = gridDim.x * 4;
    float v_n = 0.f;
    uint32_t xr[XW];
    int f_n = 0;
    if (b_first < B) {
        v_n = (lane < K) ? vals[(int64_t)b_first * K + lane] : 0.f;
        f_n = (lane < K) ? idx[(int64_t)b_first * K + lane] : 0;
        load_x(b_first, xr);
    }
    for (int b = b_first; b < B; b += b_step) {
        const int64_t code = (int64_t)b * K;
        const float v_l = v_n;
        const int f_l = f_n;
        const bool on = v_l > 0.f;
        l0_acc += __popcll(__ballot(on));
        if (on && last_activated) {
            last_activated[f_l] = step;  // model.py:178-181
            if (fired) fired[f_l] = 1.f;
        }
        const float vr_l = on ? v_l : 0.f;

        Seg seg[KJ];
        float acc[EPL];
#pragma unroll
        for (int e = 0; e < EPL; ++e) acc[e] = 0.f;
        // ---- gathers of this row first, then the requests for the next row ----
        // feature j = 2 jj + half of this row: v_readlane of lanes 2 jj and 2 jj + 1, picked per half-wave
        auto val_of = [&](int jj) {
            const float va = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(vr_l), 2 * jj));
            const float vb = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(vr_l), 2 * jj + 1));
            return half ? vb : va;
        };
#pragma unroll
        for (int jj = 0; jj < KJ; ++jj) {
            const int fa = __builtin_amdgcn_readlane(f_l, 2 * jj), fb = __builtin_amdgcn_readlane(f_l, 2 * jj + 1);
            const int fj = half ? fb : fa;  // lanes >= K carry feature 0, value 0
            seg[jj].load(WdT + (int64_t)fj * D, li);
        }
        const bool more = b + b_step < B;
        if (more) {
            v_n = (lane < K) ? vals[(int64_t)(b + b_step) * K + lane] : 0.f;
            f_n = (lane < K) ? idx[(int64_t)(b + b_step) * K + lane] : 0;
        }
        // ---- decode: this half accumulates features j = 2*jj + half ----
        if constexpr (sizeof(TW) == 2) {
            // bf16 rows: two elements per packed word -> one v_pk_fma_f32 per word
            f32x2 acc2[EPL / 2];
#pragma unroll
            for (int q = 0; q < EPL / 2; ++q) acc2[q] = f32x2{0.f, 0.f};
#pragma unroll
            for (int jj = 0; jj < KJ; ++jj) {
                const float vjj = val_of(jj);
                const f32x2 v2 = {vjj, vjj};
#pragma unroll
                for (int q = 0; q < EPL / 2; ++q) {
                    const uint32_t u = seg[jj].w[q];
                    const f32x2 w2 = {__uint_as_float(u << 16), __uint_as_float(u & 0xFFFF0000u)};
                    acc2[q] = __builtin_elementwise_fma(w2, v2, acc2[q]);
                }
            }
#pragma unroll
            for (int q = 0; q < EPL / 2; ++q) {
                acc[2 * q] = acc2[q].x;
                acc[2 * q + 1] = acc2[q].y;
            }
        } else {
#pragma unroll
            for (int jj = 0; jj < KJ; ++jj) {
#pragma unroll
                for (int e = 0; e < EPL; ++e) acc[e] = fmaf(val_of(jj), seg[jj].get(e), acc[e]);
            }
        }
        // ---- residual, loss, g (both halves end up with the full sum) ----
        float g[EPL];
#pragma unroll
        for (int c = 0; c < NCH; ++c) {
            float rec[Seg::EPC];
#pragma unroll
            for (int q = 0; q < Seg::EPC; ++q) {
                const int e = c * Seg::EPC + q;
                const int d = (32 * c + li) * Seg::EPC + q;
                rec[q] = pair_step<32>(acc[e], acc[e], lane) + bsum_s[d];  // both halves: the sum over all features
                const float r = rec[q] - x_at(xr, e);
                g[e] = r * scale;
                if (half == 0) {
                    loss_acc = fmaf(r, r, loss_acc);
                    dbd_s[wave * D + d] += g[e];  // this wave's private row of the LDS accumulator
                }
            }
            if (half == 0) {
                const int64_t o = (int64_t)b * D + (32 * c + li) * Seg::EPC;
                if (Seg::EPC == 4) {
                    if (recon_out) *(float4*)(recon_out + o) = make_float4(rec[0], rec[1], rec[2], rec[3]);
                    if (BWD) *(float4*)(g_out + o) = make_float4(g[c * 4], g[c * 4 + 1], g[c * 4 + 2], g[c * 4 + 3]);
                } else {
#pragma unroll
                    for (int q = 0; q < Seg::EPC; ++q) {
                        if (recon_out) recon_out[o + q] = rec[q];
                        if (BWD) g_out[o + q] = g[c * Seg::EPC + q];
                    }
                }
            }
        }
        if (more) load_x(b + b_step, xr);  // lands under the dpre pass
        // ---- dpre_j = (v_j > 0) ? g . W_dT[idx_j, :] : 0 ----
        if (BWD) {
            // dot of g with feature row jj of this half (rows stay PACKED across the two passes: without the
            // asm barrier the compiler carries the 16 x 12 unpacked floats of the decode pass over - 354 registers)
            bf16x2 gq[(EPL + 1) / 2];
            if constexpr (sizeof(TW) == 2) {
#pragma unroll
                for (int q = 0; q < EPL / 2; ++q) {
                    gq[q][0] = (bf16_t)g[2 * q];
                    gq[q][1] = (bf16_t)g[2 * q + 1];
                }
            }
            auto row_dot = [&](int jj) {
#pragma unroll
                for (int q = 0; q < RowSeg<TW, EPL>::NW; ++q) asm volatile("" : "+v"(seg[jj].w[q]));
                float dot = 0.f;
                if constexpr (sizeof(TW) == 2) {
                    // bf16 rows against bf16(g): v_dot2c_f32_bf16, exact products, fp32 accumulate
#pragma unroll
                    for (int q = 0; q < EPL / 2; ++q) {
                        const uint32_t u = seg[jj].w[q];
                        dot = __builtin_amdgcn_fdot2_f32_bf16(*(const bf16x2*)&u, gq[q], dot, false);
                    }
                } else {
#pragma unroll
                    for (int e = 0; e < EPL; ++e) dot = fmaf(g[e], seg[jj].get(e), dot);
                }
                return dot;
            };
            // transposing butterfly over the 32 lanes of each half: a stage with lane mask m halves the number of
            // live values (lanes with the bit set keep the upper half); once one value is left the remaining masks
            // are a plain all-reduce.  Lane li ends up with the total of jj = top log2(KJ) bits of li.  The first
            // stage (mask 16, pairs jj / jj + KJ/2) is taken as soon as both dots of a pair exist: KJ/2 live values.
            float pd[KJ / 2];
#pragma unroll
            for (int i = 0; i < KJ / 2; ++i) {
                const float a = row_dot(i), c2 = row_dot(i + KJ / 2);
                pd[i] = pair_step<16>(a, c2, li);
            }
            bfly<KJ / 2, KJ / 4, 8>(pd, li);
            // which jj does this lane hold?  the selection bits came from li's top log2(KJ) bits
            int sh = 0;
#pragma unroll
            for (int t = KJ; t > 1; t >>= 1) ++sh;
            const int jj_mine = (li >> (5 - sh)) & (KJ - 1);
            const int j_mine = 2 * jj_mine + half;
            const float vj = __shfl(v_l, j_mine, 64);
            float out = vj > 0.f ? pd[0] : 0.f;
            if (ROUND_DPRE) out = (float)(bf16_t)out;
            const bool writer = (li & ((32 >> sh) - 1)) == 0 && j_mine < K;
            if (writer) dpre[code + j_mine] = out;
        }
    }

    __syncthreads();
    __shared__ int last_flag;
    decode_block_epilogue<BWD>(loss_acc, l0_acc, dbd_s, D, B, loss_cols, red, &last_flag, part_loss, part_l0, part_dbd, ticket, stats);
}

// ------------------------------------------------------------------------------------------------
template <typename TW, int EPL, int XDT>
static void launch_decode(wsae_ctx* c, const TW* WdT, const float* params, const void* x, const int32_t* rows,
                          const float* vals, const int32_t* idx, int B, float* recon, int want_bwd, float* dpre,
                          int64_t* last_activated, const int64_t* step_count, int nblk, wsae_stats* stats, hipStream_t st) {
    const float* bd = params + c->off[3];
    const float* bpre = params + c->off[4];
    const size_t sh = (8 + 4 * (size_t)c->D) * sizeof(float);
#define DEC_ARGS WdT, bd, bpre, x, rows, vals, idx, B, c->D, c->K, recon, dpre, c->g, last_activated, c->fired, step_count, \
                 c->part_loss, c->part_l0, c->part_dbd, c->counters + 16 + 2 * TICKET_WORDS, stats, c->loss_cols
    if (!want_bwd)
        decode_kernel<TW, EPL, XDT, false, false><<<nblk, 256, sh, st>>>(DEC_ARGS);
    else if (c->prec == WSAE_PREC_BF16)
        decode_kernel<TW, EPL, XDT, true, true><<<nblk, 256, sh, st>>>(DEC_ARGS);
    else
        decode_kernel<TW, EPL, XDT, true, false><<<nblk, 256, sh, st>>>(DEC_ARGS);
#undef DEC_ARGS
}

template <typename TW, int EPL, int KJ, int XDT>
static void launch_decode_fast(wsae_ctx* c, const TW* WdT, const float* params, const void* x, const int32_t* rows,
                               const float* vals, const int32_t* idx, int B, float* recon, int want_bwd, float* dpre,
                               int64_t* last_activated, const int64_t* step_count, int nblk, wsae_stats* stats, hipStream_t st) {
    const float* bd = params + c->off[3];
    const float* bpre = params + c->off[4];
    const size_t sh = (8 + 5 * (size_t)c->D) * sizeof(float);
#define DEC_ARGS WdT, bd, bpre, x, rows, vals, idx, B, c->K, recon, dpre, c->g, last_activated, c->fired, step_count, \
                 c->part_loss, c->part_l0, c->part_dbd, c->counters + 16 + 2 * TICKET_WORDS, stats, c->loss_cols
    if (!want_bwd)
        decode_fast_kernel<TW, EPL, KJ, XDT, false, false><<<nblk, 256, sh, st>>>(DEC_ARGS);
    else if (c->prec == WSAE_PREC_BF16)
        decode_fast_kernel<TW, EPL, KJ, XDT, true, true><<<nblk, 256, sh, st>>>(DEC_ARGS);
    else
        decode_fast_kernel<TW, EPL, KJ, XDT, true, false><<<nblk, 256, sh, st>>>(DEC_ARGS);
#undef DEC_ARGS
}

template <typename TW, int XDT>
static int dispatch_decode(wsae_ctx* c, const TW* WdT, const float* params, const void* x, const int32_t* rows,
                           const float* vals, const int32_t* idx, int B, float* recon, int want_bwd, float* dpre,
                           int64_t* last_activated, const int64_t* step_count, int nblk, wsae_stats* stats, hipStream_t st) {
    // fast path: (D = 32*EPL, K = 2*KJ) shapes with the gathered rows held in registers
#define FAST_CASE(E, J)                                                                                            \
    if (c->D == 32 * E && c->K == 2 * J) {                                                                         \
        launch_decode_fast<TW, E, J, XDT>(c, WdT, params, x, rows, vals, idx, B, recon, want_bwd, dpre,            \
                                          last_activated, step_count, nblk, stats, st);                                   \
        return WSAE_OK;                                                                                            \
    }
    FAST_CASE(12, 16)  // 384, k = 32 (whisper-tiny, cfg 1-3)
    FAST_CASE(2, 4)    // 64, k = 8   (the reference's small test shape)
    FAST_CASE(4, 8)    // 128, k = 16
#undef FAST_CASE
    const int nch = ceil_div(c->D, 256);  // 4-element chunks per lane
#define DEC_CASE(N)                                                                                                \
    if (nch <= N) {                                                                                                \
        launch_decode<TW, N, XDT>(c, WdT, params, x, rows, vals, idx, B, recon, want_bwd, dpre, last_activated,    \
                                  step_count, nblk, stats, st);                                                           \
        return WSAE_OK;                                                                                            \
    }
    DEC_CASE(1) DEC_CASE(2) DEC_CASE(3) DEC_CASE(4) DEC_CASE(6) DEC_CASE(8)
#undef DEC_CASE
    wsae_set_error("decode: input_dim %d too large", c->D);
    return WSAE_ERR_INVALID;
}

template <int XDT>
static int dispatch_decode_prec(wsae_ctx* c, const float* params, const void* x, const int32_t* rows, const float* vals,
                                const int32_t* idx, int B, float* recon, int want_bwd, float* dpre,
                                int64_t* last_activated, const int64_t* step_count, int nblk, wsae_stats* stats, hipStream_t st) {
    if (c->prec == WSAE_PREC_BF16)
        return dispatch_decode<bf16_t, XDT>(c, c->WdT_bf16, params, x, rows, vals, idx, B, recon, want_bwd, dpre,
                                            last_activated, step_count, nblk, stats, st);
    return dispatch_decode<float, XDT>(c, params + c->off[1], params, x, rows, vals, idx, B, recon, want_bwd, dpre,
                                       last_activated, step_count, nblk, stats, st);
}

static int decode_launch(wsae_ctx* ctx, const float* params, const void* x, int32_t x_dtype, const int32_t* rows,
                         float* vals, int32_t* idx, int32_t B, float* recon, int32_t want_bwd, float* dpre,
                         int64_t* last_activated, const int64_t* step_count, wsae_stats* stats, hipStream_t st) {
    const int bwd = want_bwd & 1, want_g32 = (want_bwd >> 1) & 1;
    ctx->relu_g_B = 0;  // (gb / part_dbd are about to be this launch's)
    if (wsae_internal_decode_mfma_ok(ctx))
        return wsae_internal_decode_mfma(ctx, params, x, x_dtype, rows, vals, idx, B, recon, bwd, dpre, want_g32,
                                         last_activated, step_count, stats, st);
    // other shapes, FP32 mode: one round of resident blocks of the VALU kernels (2 per CU)
    const int nblk = min(ceil_div(B, 4), 2 * ctx->cus);
    int rc;
    WSAE_PROF_BEGIN(ctx, WSAE_K_DECODE, st);
    if (x_dtype == WSAE_DT_F32)
        rc = dispatch_decode_prec<WSAE_DT_F32>(ctx, params, x, rows, vals, idx, B, recon, bwd, dpre, last_activated,
                                               step_count, nblk, stats, st);
    else
        rc = dispatch_decode_prec<WSAE_DT_BF16>(ctx, params, x, rows, vals, idx, B, recon, bwd, dpre, last_activated,
                                                step_count, nblk, stats, st);
    if (rc) return rc;
    WSAE_PROF_END(ctx, WSAE_K_DECODE, st);
    WSAE_LAUNCH_CHECK();
    ctx->n_dec_blocks = nblk;
    ctx->g_is_bf16 = 0;  // these kernels leave the fp32 g (wsae_weight_grads transposes it in its bucket launch)
    ctx->g32_valid = bwd;
    return WSAE_OK;
}

static int check_decode_args(wsae_ctx* ctx, const float* params, const void* x, int32_t x_dtype, const float* vals,
                             const int32_t* idx, int32_t B, int32_t want_bwd, float* dpre, int64_t* last_activated,
                             const int64_t* step_count, wsae_stats* stats, const char* who) {
    WSAE_REQUIRE(ctx && params && x && vals && idx && stats, "%s: null argument", who);
    WSAE_REQUIRE(B >= 1 && B <= ctx->maxB, "%s: batch %d outside [1, %d]", who, B, ctx->maxB);
    WSAE_REQUIRE(x_dtype == WSAE_DT_F32 || x_dtype == WSAE_DT_BF16, "%s: unknown activation dtype %d", who, x_dtype);
    WSAE_REQUIRE(!(want_bwd & 1) || dpre, "%s: want_bwd needs a dpre buffer", who);
    WSAE_REQUIRE(!last_activated || step_count, "%s: last_activated needs step_count", who);
    return WSAE_OK;
}

extern "C" int wsae_decode_loss(wsae_ctx* ctx, const float* params, const void* x, int32_t x_dtype,
                                const int32_t* rows, const float* vals, const int32_t* idx, int32_t B, float* recon,
                                int32_t want_bwd, float* dpre, int64_t* last_activated, const int64_t* step_count,
                                wsae_stats* stats, void* stream) {
    int rc = check_decode_args(ctx, params, x, x_dtype, vals, idx, B, want_bwd, dpre, last_activated, step_count, stats,
                               "wsae_decode_loss");
    if (rc) return rc;
    return decode_launch(ctx, params, x, x_dtype, rows, (float*)vals, (int32_t*)idx, B, recon, want_bwd, dpre, last_activated,
                         step_count, stats, (hipStream_t)stream);
}

extern "C" int wsae_encode_decode(wsae_ctx* ctx, const float* params, const void* x, int32_t x_dtype,
                                  const int32_t* rows, int32_t B, float* vals, int32_t* idx, int64_t* step_count,
                                  float* recon, int32_t want_bwd, float* dpre, int64_t* last_activated, wsae_stats* stats,
                                  void* stream) {
    int rc = check_decode_args(ctx, params, x, x_dtype, vals, idx, B, want_bwd, dpre, last_activated, step_count, stats,
                               "wsae_encode_decode");
    if (rc) return rc;
    hipStream_t st = (hipStream_t)stream;
    rc = wsae_internal_encode_topk(ctx, params, x, x_dtype, rows, B, vals, idx, step_count, &stats->topk_fallback_rows, st);
    if (rc) return rc;
    return decode_launch(ctx, params, x, x_dtype, rows, vals, idx, B, recon, want_bwd, dpre, last_activated, step_count, stats,
                         st);
}

// g = 2 (recon - target) / (B cols) of the last decode launch, fp32 [B, D] (kept when want_bwd had bit 1 set): the
// gradient of the loss with respect to the reconstruction; a transcoder's skip path trains through it.
extern "C" int wsae_last_residual_grad(wsae_ctx* ctx, int32_t B, float* g_out, void* stream) {
    WSAE_REQUIRE(ctx && g_out && B >= 1 && B <= ctx->maxB, "wsae_last_residual_grad: bad argument");
    WSAE_REQUIRE(ctx->g32_valid, "wsae_last_residual_grad: the preceding decode did not keep the fp32 g (pass want_bwd = 3)");
    WSAE_HIP_CHECK(hipMemcpyAsync(g_out, ctx->g, (size_t)B * ctx->D * sizeof(float), hipMemcpyDeviceToDevice,
                                  (hipStream_t)stream));
    return WSAE_OK;
}

// ------------------------------------------------------------------------------------------------
// dL/dx = dpre W_e - g      (autograd API path only; model.py:108,145)
// ------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) input_grad_kernel(const float* __restrict__ We, const float* __restrict__ g,
                                                         const int32_t* __restrict__ idx, const float* __restrict__ dpre,
                                                         int B, int D, int K, float* __restrict__ dx) {
    const int lane = threadIdx.x & 63;
    const int b = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (b >= B) return;
    for (int d = lane; d < D; d += 64) {
        float a = g ? -g[(int64_t)b * D + d] : 0.f;  // (g null: the target is not the input - transcoders)
        for (int j = 0; j < K; ++j) {
            const float dp = dpre[(int64_t)b * K + j];
            if (dp != 0.f) a = fmaf(dp, We[(int64_t)idx[(int64_t)b * K + j] * D + d], a);
        }
        dx[(int64_t)b * D + d] = a;
    }
}

extern "C" int wsae_input_grad(wsae_ctx* ctx, const float* params, const int32_t* idx, const float* dpre, int32_t B,
                               float* dx, int32_t subtract_g, void* stream) {
    WSAE_REQUIRE(ctx && params && idx && dpre && dx && B >= 1 && B <= ctx->maxB, "wsae_input_grad: bad argument");
    WSAE_REQUIRE(!subtract_g || ctx->g32_valid, "wsae_input_grad: the preceding decode did not keep the fp32 g (pass want_bwd = 3)");
    input_grad_kernel<<<ceil_div(B, 4), 256, 0, (hipStream_t)stream>>>(params + ctx->off[0], subtract_g ? ctx->g : nullptr, idx, dpre,
                                                                       B, ctx->D, ctx->K, dx);
    WSAE_LAUNCH_CHECK();
    return WSAE_OK;
}
