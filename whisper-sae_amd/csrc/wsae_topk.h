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

// ------------------------------------------------------------------------------------------------
// Strip store threshold.  The selection above reads a strip only when its maximum is >= the row's T, so the encoder
// GEMM need not write the strips below T to HBM at all - but it finishes a row's strips long before the row's T
// exists.  It therefore predicts: every TopK launch leaves the smallest T of (a sample of) its batch behind, and the
// NEXT encoder launch stores only the strips whose maximum reaches
//     tg = s * (the smaller of the last TWO launches' minima - callers that alternate between two kinds of batches,
//          training and validation say, are then predicted from the right kind as well)
// (everything when there is no history or that minimum is <= 0).
// The prediction is checked, not trusted: a row whose own T turns out below tg first tries the selection at tg itself
// (every element >= tg is in a stored strip: with K of them nothing is missing after all) and otherwise - or when it
// needs the exact full-row path - recomputes the strips it is missing right here: the same MFMA, the same K order,
// the same bias add as the encoder GEMM's (encode_gemm256d_kernel / Mfma256s<bf16_t>), so the values are the ones
// the GEMM would have stored, bit for bit, written into the row before it is read.  Results never depend on the
// prediction; only the time does (measured: DESIGN.md section 4.1).
// A row that recomputes strips holds its TopK launch up by tens of microseconds, so the margin s looks after itself:
// it starts at TG_S_START, loses a tenth whenever the previous launch had such a row and otherwise creeps up by
// TG_S_UP per launch to at most TG_S_MAX - on data whose row thresholds scatter widely it settles low (at TG_S_MIN
// hardly a strip is skipped and hardly a row can miss), on well-behaved data just below the batch minimum.
// ------------------------------------------------------------------------------------------------
#define TG_S_START 0.95f
#define TG_S_MAX 1.0f
#define TG_S_MIN 0.25f
#define TG_S_DOWN 0.9f
#define TG_S_UP 0.005f

// slots: three groups used in rotation; `cur` is the group of the current batch (its GEMM launch re-arms it, its TopK
// launch fills it), the other two belong to the two batches before it.  A group: TG_SLOTS minimum words, TG_SLOT_STRIDE
// apart (one L2 channel each, so that the atomic minima of a launch do not queue on one address; 0 / 0xFFFFFFFF: never
// filled), and in the gap after the first slot the margin s its GEMM launch used and the number of rows that recomputed
// strips in its TopK launch.  strip_store_threshold is called by whole waves - of the GEMM only (128 cache lines per
// call): block 0 leaves the threshold in the word TG_USED for the TopK launch to read.
#define TG_SLOTS 64
#define TG_SLOT_STRIDE 64   // words
#define TG_GROUP_WORDS (TG_SLOTS * TG_SLOT_STRIDE)
#define TG_HDR_S 1          // float, word offset inside the group
#define TG_HDR_MISSES 2     // int
#define TG_USED (3 * TG_GROUP_WORDS)        // float: the store threshold of the last predicated GEMM launch
#define TG_REFILLED (3 * TG_GROUP_WORDS + 1)  // int: rows that recomputed strips, cumulative
#define TG_WORDS (3 * TG_GROUP_WORDS + 16)

// fixed_s > 0: use this margin instead of the adaptive one (experiments).  *s_out: the margin of this launch.
__device__ __forceinline__ float strip_store_threshold(const uint32_t* slots, int cur, int lane, float fixed_s, float* s_out) {
    const uint32_t* prev = slots + ((cur + 2) % 3) * TG_GROUP_WORDS;
    const uint32_t* prev2 = slots + ((cur + 1) % 3) * TG_GROUP_WORDS;
    uint32_t a = __builtin_nontemporal_load(prev + lane * TG_SLOT_STRIDE);
    uint32_t b = __builtin_nontemporal_load(prev2 + lane * TG_SLOT_STRIDE);
    const float s_prev = __uint_as_float(__builtin_nontemporal_load(prev + TG_HDR_S));
    const int misses = (int)__builtin_nontemporal_load(prev + TG_HDR_MISSES);
    float s = !(s_prev > 0.f) ? TG_S_START : misses > 0 ? fmaxf(s_prev * TG_S_DOWN, TG_S_MIN) : fminf(s_prev + TG_S_UP, TG_S_MAX);
    if (fixed_s > 0.f) s = fixed_s;
    *s_out = s;
    if (a == 0u) a = 0xFFFFFFFFu;
    if (b == 0u) b = 0xFFFFFFFFu;
    uint32_t o = a < b ? a : b;
    o = min(o, lane_xor_u32<1>(o, lane));
    o = min(o, lane_xor_u32<2>(o, lane));
    o = min(o, lane_xor_u32<4>(o, lane));
    o = min(o, lane_xor_u32<8>(o, lane));
    o = min(o, lane_xor_u32<16>(o, lane));
    o = min(o, lane_xor_u32<32>(o, lane));
    o = (uint32_t)__builtin_amdgcn_readfirstlane((int)o);
    if (o == 0xFFFFFFFFu) return -INFINITY;  // no history: store everything
    const float t = ord_f32(o);
    return t > 0.f ? s * t : -INFINITY;  // (NaN -> -inf as well)
}

struct StripFix {
    const bf16_t* x;        // the encoder GEMM's A operand (bf16 rows of D elements) ...
    const int32_t* arows;   // ... through its row list (nullable)
    const bf16_t* We;       // [H][D] bf16
    const float* bias;      // the folded encoder bias the GEMM added
    int D;
    float tg;               // the threshold the GEMM stored with (-inf: everything was stored)
    uint32_t* tmin_group;   // this launch's group (nullable)
    int32_t* miss_rows;     // rows that had to recompute strips, cumulative (statistics)
};

// this row's T: keep the batch minimum for the next launches' prediction.  One row in four reports (a minimum over a
// quarter of the batch predicts as well, and the check below does not rely on it).
__device__ __forceinline__ void strip_fix_note(const StripFix& f, uint32_t thi, int b, int lane) {
    if (f.tmin_group && (b & 3) == 0 && lane == 0) atomicMin(f.tmin_group + ((b >> 2) & (TG_SLOTS - 1)) * TG_SLOT_STRIDE, thi);
}

// Recompute, into prow, the strips of one row named by `masks` (wave-private LDS: bit s & 31 of word s >> 5 = strip s is
// wanted; nwords words).  xs: wave-private LDS for the row's x (2 D bytes), or null (D too large: x is read from memory).
// Output row 0 of a 32 x 32 MFMA tile whose A rows all carry x[b]: lane (r, h) supplies k = 16 s + 8 h + j of both
// operands at step s, as the GEMM's lanes do (Mfma256s<bf16_t>::slab, slabs and steps ascending in k), so the sums are the
// GEMM's own.  W_e comes from L2 eight K steps at a time.
typedef const __attribute__((address_space(1))) bf16x8* gptr_bf16x8;
typedef __attribute__((address_space(3))) char* lds_bytes;

template <bool XS>
__device__ __noinline__ void strip_fix_fill(const bf16_t* xrow, const bf16_t* We, const float* bias, int D, int32_t* miss_rows,
                                            int32_t* group_misses, const int* masks, int nwords, lds_bytes xs, float* prow, int lane) {
    if (lane == 0) {
        atomicAdd(miss_rows, 1);
        atomicAdd(group_misses, 1);
    }
    const int r = lane & 31, h = lane >> 5;
    gptr_bf16x8 ap = (gptr_bf16x8)(uintptr_t)(xrow + 8 * h);
    if constexpr (XS) {
        for (int c = lane; c < (D >> 3); c += 64) *(__attribute__((address_space(3))) bf16x8*)(xs + 16 * c) = *(gptr_bf16x8)(uintptr_t)(xrow + 8 * c);
        __builtin_amdgcn_wave_barrier();
    }
    constexpr int NB = XS ? 8 : 4;  // K steps in flight (D is a multiple of 128 on the persistent GEMM's shapes)
    for (int w = 0; w < nwords; ++w) {
        const uint32_t m = (uint32_t)__builtin_amdgcn_readfirstlane(masks[w]);
        uint32_t pairs = (m | (m >> 1)) & 0x55555555u;  // 32-column blocks with a wanted strip
        while (pairs) {
            const int p = __builtin_ctz(pairs);
            pairs &= pairs - 1;
            const int blk = (32 * w + p) >> 1;
            const bool n0 = (m >> p) & 1u, n1 = (m >> (p + 1)) & 1u;
            gptr_bf16x8 bp = (gptr_bf16x8)(uintptr_t)(We + (int64_t)(blk * 32 + r) * D + 8 * h);
            f32x16 acc;
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[i] = 0.f;
            for (int s0 = 0; s0 < (D >> 4); s0 += NB) {
                bf16x8 bv[NB], av[XS ? 1 : NB];
#pragma unroll
                for (int j = 0; j < NB; ++j) {
                    bv[j] = bp[2 * (s0 + j)];
                    if constexpr (!XS) av[j] = ap[2 * (s0 + j)];
                }
#pragma unroll
                for (int j = 0; j < NB; ++j) {
                    if constexpr (XS) av[0] = *(const __attribute__((address_space(3))) bf16x8*)(xs + 32 * (s0 + j) + 16 * h);
                    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av[XS ? 0 : j], bv[j], acc, 0, 0, 0);
                }
            }
            const float v = acc[0] + bias[blk * 32 + r];
            if (h == 0 && (r < 16 ? n0 : n1)) prow[blk * 32 + r] = v;
        }
    }
    __threadfence();  // the row is read back by this wave (and must not come from a stale L1 line)
}

// candidate strips (maximum ranks >= th) -> slist, their elements ranking >= th -> list; returns the number of such
// elements (may exceed CAP: only CAP are stored), or -1 when more than TS_MAX_STRIPS strips qualify
template <int SPL, int CAP>
__device__ __forceinline__ int strips_collect(const float* row, const float (&sm)[SPL], int ns, uint32_t th, int lane,
                                              uint64_t* list, int* slist) {
    int nstr = 0;
#pragma unroll
    for (int i = 0; i < SPL; ++i) {
        const bool pass = lane + 64 * i < ns && f32_ord(sm[i]) >= th;
        const unsigned long long mask = __ballot(pass);
        const int pos = nstr + __popcll(mask & ((1ull << lane) - 1ull));
        if (pass && pos < TS_MAX_STRIPS) slist[pos] = lane + 64 * i;
        nstr += __popcll(mask);
    }
    if (nstr > TS_MAX_STRIPS) return -1;
    __builtin_amdgcn_wave_barrier();
    // read the candidate strips, 16 per pass: lane l -> strip slist[base + l / 4], float4 number l & 3
    int total = 0;
    for (int base = 0; base < nstr; base += 16) {
        const int si = base + (lane >> 2);
        const bool in = si < nstr;
        const int s = in ? slist[si] : 0;
        const int e = s * 16 + (lane & 3) * 4;
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (in) v = nt_load4((const float4*)(row + e));  // (read once)
        const float vv[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const uint32_t o = f32_ord(vv[c]);
            const bool pass = in && o >= th;
            const unsigned long long mask = __ballot(pass);
            if (mask) {
                const int pos = total + __popcll(mask & ((1ull << lane) - 1ull));
                if (pass && pos < CAP) list[pos] = ((uint64_t)o << 32) | (uint32_t)(~(uint32_t)(e + c));
                total += __popcll(mask);
            }
        }
    }
    __builtin_amdgcn_wave_barrier();
    return total;
}

// rebuild the strips of row b that the GEMM did not store (maximum < tg) and whose maximum ranks >= lo
template <int SPL, int CAP>
__device__ __forceinline__ void strips_refill(const StripFix& fix, int b, const float (&sm)[SPL], int ns, uint32_t lo, int lane,
                                              uint64_t* list, int* slist, float* row) {
    if (!(fix.tg > -INFINITY)) return;
#pragma unroll
    for (int i = 0; i < SPL; ++i) {
        const bool want = lane + 64 * i < ns && sm[i] < fix.tg && f32_ord(sm[i]) >= lo;
        const unsigned long long m = __ballot(want);
        if (lane == 0) {
            slist[2 * i] = (int)(uint32_t)m;
            slist[2 * i + 1] = (int)(uint32_t)(m >> 32);
        }
    }
    __builtin_amdgcn_wave_barrier();
    const bf16_t* xrow = fix.x + (int64_t)(fix.arows ? fix.arows[b] : b) * fix.D;
    if (fix.D <= 4 * CAP)  // the row's x fits the wave's candidate list (rebuilt afterwards)
        strip_fix_fill<true>(xrow, fix.We, fix.bias, fix.D, fix.miss_rows, (int32_t*)fix.tmin_group + TG_HDR_MISSES, slist, 2 * SPL, (lds_bytes)(char*)list, row, lane);
    else
        strip_fix_fill<false>(xrow, fix.We, fix.bias, fix.D, fix.miss_rows, (int32_t*)fix.tmin_group + TG_HDR_MISSES, slist, 2 * SPL, nullptr, row, lane);
    __builtin_amdgcn_wave_barrier();
}

// (rare paths of topk_strips_row, out of line so that they cost the common path no registers)
// The row's T is below the store threshold tg.  First select at tg itself: every element >= tg lives in a stored strip, so
// when at least K of them turn up the K best are among them and nothing is missing after all (T is only a lower bound of
// the K-th largest element - about 5 % below it on Gaussian rows).  Otherwise rebuild the strips with T <= maximum < tg.
// full != 0: the exact path is wanted straight away (too many candidate strips or candidates at T).
template <int SPL, int CAP>
__device__ __noinline__ void topk_strips_row_rare(const float* row, const float* srow, int H, int K, int lane, uint64_t* list,
                                                  int* slist, float* vrow, int32_t* irow, int32_t* fallback_rows, float* vs,
                                                  int32_t* is, StripFix fix, int b, uint32_t thi, int full) {
    const int ns = H >> 4;
    float sm[SPL];
#pragma unroll
    for (int i = 0; i < SPL; ++i) sm[i] = lane + 64 * i < ns ? srow[lane + 64 * i] : -INFINITY;
    if (!full) {
        int total = strips_collect<SPL, CAP>(row, sm, ns, f32_ord(fix.tg), lane, list, slist);
        if (total >= 0 && total < K) {
            strips_refill<SPL, CAP>(fix, b, sm, ns, thi, lane, list, slist, (float*)row);
            total = strips_collect<SPL, CAP>(row, sm, ns, thi, lane, list, slist);
        }
        if (total >= K && total <= CAP) {
            topk_emit_any<CAP>(list, total, K, lane, vrow, irow, vs, is);
            return;
        }
    }
    strips_refill<SPL, CAP>(fix, b, sm, ns, 0u, lane, list, slist, (float*)row);  // the exact path reads the whole row
    topk_row_generic<CAP>(row, H, K, list, lane, vrow, irow, fallback_rows, vs, is);
}

// pred: the encoder GEMM stored this batch's strips selectively (see above), fix says how to rebuild one
template <int SPL, int NTOP, int CAP>
__device__ __forceinline__ void topk_strips_row(const float* row, const float* __restrict__ srow, int H,
                                                int K, int lane, uint64_t* list, int* slist, float* vrow,
                                                int32_t* irow, int32_t* fallback_rows, float* vs = nullptr,
                                                int32_t* is = nullptr, bool pred = false, const StripFix& fix = StripFix(), int b = 0) {
    const int ns = H >> 4;
    float sm[SPL];
    float m = -INFINITY, m2 = -INFINITY;
#pragma unroll
    for (int i = 0; i < SPL; ++i) {
        const int s = lane + 64 * i;
        sm[i] = s < ns ? __builtin_nontemporal_load(srow + s) : -INFINITY;
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
    if (pred) {
        strip_fix_note(fix, thi, b, lane);
        if (!(ord_f32(thi) >= fix.tg)) {
            topk_strips_row_rare<SPL, CAP>(row, srow, H, K, lane, list, slist, vrow, irow, fallback_rows, vs, is, fix, b, thi, 0);
            return;
        }
    }
    const int total = strips_collect<SPL, CAP>(row, sm, ns, thi, lane, list, slist);
    if (total < 0 || total > CAP || total < K) {
        if (pred) topk_strips_row_rare<SPL, CAP>(row, srow, H, K, lane, list, slist, vrow, irow, fallback_rows, vs, is, fix, b, thi, 1);
        else topk_row_generic<CAP>(row, H, K, list, lane, vrow, irow, fallback_rows, vs, is);
        return;
    }
    topk_emit_any<CAP>(list, total, K, lane, vrow, irow, vs, is);
}
