// Per-row TopK on one wavefront: orderable keys, VALU lane exchanges, bitonic sorts, candidate compaction, the
// exact bisection path and the strip-guided selection.  Shared by the standalone TopK kernels (wsae_encode.hip)
// and the fused TopK + decode kernel (wsae_decode_mfma.hip).
//   reference: torch.topk(pre, k) in TopKSAE.encode, src/whisper_sae/sae/model.py:114
//
//   key = (orderable(value) << 32) | ~index : descending key order == (value desc, index asc).
//   Tie rule (the reference leaves it to torch.topk): value descending, then lowest index first.  The orderable
//   map is a total order on bit patterns: +0.0 ranks above -0.0 and NaNs sort by payload (above +inf when
//   positive) - torch.topk treats the zeros as equal and NaN as the largest value; finite, non-tied data (every
//   fixture) cannot tell the difference.
#pragma once

#include "wsae_common.h"

__device__ __forceinline__ uint32_t f32_ord(float f) {
    const uint32_t u = __float_as_uint(f);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ __forceinline__ float ord_f32(uint32_t o) {
    return __uint_as_float((o & 0x80000000u) ? (o & 0x7FFFFFFFu) : ~o);
}
// (lane_xor_u32<M>: wsae_common.h)
template <int M>
__device__ __forceinline__ uint64_t lane_xor_u64(uint64_t v, int lane) {
    return ((uint64_t)lane_xor_u32<M>((uint32_t)(v >> 32), lane) << 32) | lane_xor_u32<M>((uint32_t)v, lane);
}

// partner key at lane distance stride / NPL (the loops around the call are fully unrolled, so the switch folds)
template <int M>
__device__ __forceinline__ uint64_t lane_xor_key(uint64_t k, int lane) { return lane_xor_u64<M>(k, lane); }
template <int M>
__device__ __forceinline__ uint32_t lane_xor_key(uint32_t k, int lane) { return lane_xor_u32<M>(k, lane); }

template <int NPL, typename KT>
__device__ __forceinline__ KT xor_partner(KT k, int stride, int lane) {
    switch (stride / NPL) {
        case 1: return lane_xor_key<1>(k, lane);
        case 2: return lane_xor_key<2>(k, lane);
        case 4: return lane_xor_key<4>(k, lane);
        case 8: return lane_xor_key<8>(k, lane);
        case 16: return lane_xor_key<16>(k, lane);
        default: return lane_xor_key<32>(k, lane);
    }
}

// bitonic sort, descending, of 64 * NPL keys held NPL per lane (position p = lane * NPL + i)
// (KT = uint32_t when only the value decides - the threshold sorts - halves the work of the 64-bit (value, index) keys)
template <int NPL, typename KT = uint64_t>
__device__ __forceinline__ void wave_sort_desc(KT (&key)[NPL], int lane) {
#pragma unroll
    for (int size = 2; size <= 64 * NPL; size <<= 1) {
#pragma unroll
        for (int stride = size >> 1; stride > 0; stride >>= 1) {
            if (stride >= NPL) {
#pragma unroll
                for (int i = 0; i < NPL; ++i) {
                    const KT other = xor_partner<NPL>(key[i], stride, lane);
                    const int p = lane * NPL + i;
                    const bool desc = (p & size) == 0;
                    const bool lower = (p & stride) == 0;
                    const bool keep_max = (lower == desc);
                    // one select per exchange: keep the own key when (own > other) says what this position wants, else
                    // take the partner's (max and min computed separately and selected afterwards cost three selects)
                    key[i] = ((key[i] > other) == keep_max) ? key[i] : other;
                }
            } else {
#pragma unroll
                for (int i = 0; i < NPL; ++i) {
                    if ((i & stride) == 0) {
                        const int j = i | stride;
                        const int p = lane * NPL + i;
                        const bool desc = (p & size) == 0;
                        const KT a = key[i], b = key[j];
                        const bool swap = (a > b) != desc;
                        key[i] = swap ? b : a;
                        key[j] = swap ? a : b;
                    }
                }
            }
        }
    }
}

// sort the `count` keys of `list`, write the K best (value, index) pairs to vrow / irow (global) and, when given,
// to vs / is (wave-private LDS rows the fused decode continues from)
template <int NPL>
__device__ __forceinline__ void topk_emit(const uint64_t* list, int count, int K, int lane, float* vrow, int32_t* irow,
                                          float* vs = nullptr, int32_t* is = nullptr) {
    uint64_t key[NPL];
#pragma unroll
    for (int i = 0; i < NPL; ++i) {
        const int p = lane * NPL + i;
        key[i] = p < count ? list[p] : 0ull;
    }
    wave_sort_desc<NPL>(key, lane);
#pragma unroll
    for (int i = 0; i < NPL; ++i) {
        const int p = lane * NPL + i;
        if (p < K) {
            const float v = ord_f32((uint32_t)(key[i] >> 32));
            const int32_t f = (int32_t)(~(uint32_t)key[i]);
            vrow[p] = v;
            irow[p] = f;
            if (vs) {
                vs[p] = v;
                is[p] = f;
            }
        }
    }
}

template <int CAP>
__device__ __forceinline__ void topk_emit_any(const uint64_t* list, int count, int K, int lane, float* vrow,
                                              int32_t* irow, float* vs = nullptr, int32_t* is = nullptr) {
    if (count <= 64) topk_emit<1>(list, count, K, lane, vrow, irow, vs, is);
    else if (CAP <= 128 || count <= 128) topk_emit<2>(list, count, K, lane, vrow, irow, vs, is);
    else topk_emit<4>(list, count, K, lane, vrow, irow, vs, is);
}

// compact every element with key >= kmin into list (wave-private LDS, CAP entries); returns the count
// (wave-uniform).  Stops storing beyond CAP but keeps counting.
template <int CAP>
__device__ __forceinline__ int topk_compact(const float* __restrict__ row, int H, uint64_t kmin, uint64_t* list,
                                            int lane) {
    int base = 0;
    for (int e0 = 0; e0 < H; e0 += 256) {
        const int e = e0 + lane * 4;
        float4 v = make_float4(0, 0, 0, 0);
        const bool in = e < H;
        if (in) v = *(const float4*)(row + e);
        const float vv[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const uint64_t key = ((uint64_t)f32_ord(vv[c]) << 32) | (uint32_t)(~(uint32_t)(e + c));
            const bool pass = in && key >= kmin;
            const unsigned long long m = __ballot(pass);
            if (m) {
                const int pos = base + __popcll(m & ((1ull << lane) - 1ull));
                if (pass && pos < CAP) list[pos] = key;
                base += __popcll(m);
            }
        }
    }
    return base;
}

__device__ __forceinline__ int topk_count_ge(const float* __restrict__ row, int H, uint64_t kmin, int lane) {
    int cnt = 0;
    for (int e0 = 0; e0 < H; e0 += 256) {
        const int e = e0 + lane * 4;
        if (e < H) {
            const float4 v = *(const float4*)(row + e);
            const float vv[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                const uint64_t key = ((uint64_t)f32_ord(vv[c]) << 32) | (uint32_t)(~(uint32_t)(e + c));
                cnt += key >= kmin ? 1 : 0;
            }
        }
    }
    return wave_sum_i(cnt);
}

// exact path: bisection on the 64-bit key for the K-th largest key (K <= 128), then compaction + sort
template <int CAP>
__device__ __noinline__ void topk_row_generic(const float* row, int H, int K, uint64_t* list, int lane, float* vrow,
                                              int32_t* irow, int32_t* fallback_rows, float* vs = nullptr,
                                              int32_t* is = nullptr) {
    if (lane == 0) atomicAdd(fallback_rows, 1);
    uint64_t prefix = 0;
    for (int bit = 63; bit >= 0; --bit) {
        const uint64_t cand = prefix | (1ull << bit);
        if (topk_count_ge(row, H, cand, lane) >= K) prefix = cand;
    }
    const int count = topk_compact<CAP>(row, H, prefix, list, lane);  // == K exactly (keys are distinct)
    if (count <= 64) topk_emit<1>(list, count, K, lane, vrow, irow, vs, is);
    else topk_emit<2>(list, count, K, lane, vrow, irow, vs, is);
}

// ------------------------------------------------------------------------------------------------
// Strip-guided TopK of one row (one wave): smax holds the maxima of the row's 16-column strips (left by the
// encoder GEMM's epilogue).
//   T = K-th largest of the 64 lane maxima of the strip maxima: at least K strips - hence at least K distinct
//   elements - are >= T, and every element >= T lives in a strip whose maximum is >= T.  So only those strips
//   (typically K .. 1.5 K of the H/16) are read from the [B,H] matrix: 4 lanes x 16 bytes per strip, 16 strips
//   per load instruction.  Anything unusual (more than TS_MAX_STRIPS candidate strips or CAP candidates) goes
//   to the exact full-row path.
// NTOP = strip maxima each lane contributes to the threshold: 1 -> T = K-th largest of 64 (K <= 32 in practice:
// at K = 64 that would be the smallest lane maximum and nearly every strip would qualify), 2 -> K-th largest of
// the 128 values "largest and second largest strip maximum of every lane" (distinct strips, so still >= K
// distinct elements >= T).  SPL = strip maxima per lane: H / 16 <= 64 * SPL.
// ------------------------------------------------------------------------------------------------
#define TS_MAX_STRIPS 128

template <int SPL, int NTOP, int CAP>
__device__ __forceinline__ void topk_strips_row(const float* __restrict__ row, const float* __restrict__ srow, int H,
                                                int K, int lane, uint64_t* list, int* slist, float* vrow,
                                                int32_t* irow, int32_t* fallback_rows, float* vs = nullptr,
                                                int32_t* is = nullptr) {
    const int ns = H >> 4;
    float sm[SPL];
    float m = -INFINITY, m2 = -INFINITY;
#pragma unroll
    for (int i = 0; i < SPL; ++i) {
        const int s = lane + 64 * i;
        sm[i] = s < ns ? srow[s] : -INFINITY;
        m2 = fmaxf(m2, fminf(m, sm[i]));
        m = fmaxf(m, sm[i]);
    }
    uint32_t thi;  // ord(T)
    if constexpr (NTOP == 1) {
        uint32_t mk[1] = {f32_ord(m)};
        wave_sort_desc<1, uint32_t>(mk, lane);
        thi = (uint32_t)__builtin_amdgcn_readlane((int)mk[0], K - 1);  // (K is wave-uniform: v_readlane, not an LDS round trip)
    } else {
        uint32_t mk[2] = {f32_ord(m), f32_ord(m2)};
        wave_sort_desc<2, uint32_t>(mk, lane);  // position p of the descending order sits in lane p / 2, slot p % 2
        const uint32_t lo = (uint32_t)__builtin_amdgcn_readlane((int)mk[0], (K - 1) >> 1);
        const uint32_t hi = (uint32_t)__builtin_amdgcn_readlane((int)mk[1], (K - 1) >> 1);
        thi = ((K - 1) & 1) ? hi : lo;
    }

    // candidate strips -> wave-private list (ballot prefix per i)
    int nstr = 0;
#pragma unroll
    for (int i = 0; i < SPL; ++i) {
        const bool pass = lane + 64 * i < ns && f32_ord(sm[i]) >= thi;
        const unsigned long long mask = __ballot(pass);
        const int pos = nstr + __popcll(mask & ((1ull << lane) - 1ull));
        if (pass && pos < TS_MAX_STRIPS) slist[pos] = lane + 64 * i;
        nstr += __popcll(mask);
    }
    if (nstr > TS_MAX_STRIPS) {
        topk_row_generic<CAP>(row, H, K, list, lane, vrow, irow, fallback_rows, vs, is);
        return;
    }
    __builtin_amdgcn_wave_barrier();
    // read the candidate strips, 16 per pass: lane l -> strip slist[base + l / 4], float4 number l & 3
    int total = 0;
    for (int base = 0; base < nstr; base += 16) {
        const int si = base + (lane >> 2);
        const bool in = si < nstr;
        const int s = in ? slist[si] : 0;
        const int e = s * 16 + (lane & 3) * 4;
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (in) v = *(const float4*)(row + e);
        const float vv[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const uint32_t o = f32_ord(vv[c]);
            const bool pass = in && o >= thi;
            const unsigned long long mask = __ballot(pass);
            if (mask) {
                const int pos = total + __popcll(mask & ((1ull << lane) - 1ull));
                if (pass && pos < CAP) list[pos] = ((uint64_t)o << 32) | (uint32_t)(~(uint32_t)(e + c));
                total += __popcll(mask);
            }
        }
    }
    if (total > CAP || total < K) {
        topk_row_generic<CAP>(row, H, K, list, lane, vrow, irow, fallback_rows, vs, is);
        return;
    }
    __builtin_amdgcn_wave_barrier();
    topk_emit_any<CAP>(list, total, K, lane, vrow, irow, vs, is);
}
