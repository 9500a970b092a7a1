// Row N4: the consumer right after the path -- for every feature keep the `keep` strongest
// activations seen so far, on the device, fed by the compact code (vals, idx) the TopK kernel emits (or by a dense
// [rows][H] activation matrix).  Replaces the reference's Python triple loop + one heap per feature
// (/root/reference/src/whisper_sae/analysis/feature_viz.py:94-158).
//
// An activation is identified by its arrival ordinal (ord_base + row): the host maps ordinals back to
// (sample, position).  Order inside a feature's list: value descending, then ordinal ascending -- the reference keeps
// the earlier of two equal values at the boundary (its `act_value > heap[0][0]` is strict) and the list is
// deterministic because (value, ordinal) is unique inside one feature.
//
// Four launches per update, all HBM-bound integer/byte work (12 B per entry read twice + 8 B written once):
//   count    histogram of the positive entries per feature (+ the total)
//   scan     exclusive scan of the histogram (one block)
//   scatter  entries grouped by feature (order inside a group is arbitrary; the merge is order-independent)
//   merge    one wave per feature: the list lives one element per lane, candidates that beat the current minimum
//            are inserted by a lane shift
#include "wsae_common.h"

namespace {

constexpr int FT_BLOCK = 256;

template <bool DENSE>
__device__ __forceinline__ int ft_feature(const int32_t* idx, int64_t e, int width) {
    return DENSE ? (int)(e % width) : idx[e];
}

template <bool DENSE>
__global__ __launch_bounds__(FT_BLOCK) void ft_count_kernel(const float* __restrict__ vals, const int32_t* __restrict__ idx,
                                                            int64_t n, int width, int H, int32_t* __restrict__ cnt,
                                                            unsigned long long* __restrict__ total) {
    unsigned long long mine = 0;
    for (int64_t e = (int64_t)blockIdx.x * FT_BLOCK + threadIdx.x; e < n; e += (int64_t)gridDim.x * FT_BLOCK) {
        const float v = vals[e];
        if (v > 0.f) {
            const int f = ft_feature<DENSE>(idx, e, width);
            if ((unsigned)f < (unsigned)H) {
                atomicAdd(&cnt[f], 1);
                ++mine;
            }
        }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) mine += __shfl_xor(mine, o, 64);
    if ((threadIdx.x & 63) == 0 && mine) atomicAdd(total, mine);
}

// offs[f] = sum of cnt[<f]; cursor[f] = offs[f]
__global__ __launch_bounds__(1024) void ft_scan_kernel(const int32_t* __restrict__ cnt, int32_t* __restrict__ offs,
                                                       int32_t* __restrict__ cursor, int H) {
    __shared__ int32_t wsum[16];
    __shared__ int32_t carry;
    const int t = threadIdx.x, lane = t & 63, w = t >> 6;
    if (t == 0) carry = 0;
    __syncthreads();
    for (int base = 0; base < H; base += 1024) {
        const int f = base + t;
        const int32_t c = f < H ? cnt[f] : 0;
        int32_t s = c;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const int32_t u = __shfl_up(s, o, 64);
            if (lane >= o) s += u;
        }
        if (lane == 63) wsum[w] = s;
        __syncthreads();
        int32_t before = carry;
        for (int i = 0; i < w; ++i) before += wsum[i];
        if (f < H) {
            offs[f] = before + s - c;
            cursor[f] = before + s - c;
        }
        __syncthreads();
        if (t == 1023) carry = before + s;
        __syncthreads();
    }
    if (t == 0) offs[H] = carry;
}

template <bool DENSE>
__global__ __launch_bounds__(FT_BLOCK) void ft_scatter_kernel(const float* __restrict__ vals, const int32_t* __restrict__ idx,
                                                              int64_t n, int width, int H, int32_t* __restrict__ cursor,
                                                              float* __restrict__ ent_val, uint32_t* __restrict__ ent_row) {
    for (int64_t e = (int64_t)blockIdx.x * FT_BLOCK + threadIdx.x; e < n; e += (int64_t)gridDim.x * FT_BLOCK) {
        const float v = vals[e];
        if (v > 0.f) {
            const int f = ft_feature<DENSE>(idx, e, width);
            if ((unsigned)f < (unsigned)H) {
                const int p = atomicAdd(&cursor[f], 1);
                ent_val[p] = v;
                ent_row[p] = (uint32_t)(e / width);
            }
        }
    }
}

__device__ __forceinline__ bool ft_beats(float v, int64_t o, float lv, int64_t lo) { return v > lv || (v == lv && o < lo); }

// one wave per feature; lane l holds list element l (keep <= 64)
__global__ __launch_bounds__(FT_BLOCK) void ft_merge_kernel(const int32_t* __restrict__ offs, const float* __restrict__ ent_val,
                                                            const uint32_t* __restrict__ ent_row, int H, int keep,
                                                            int64_t ord_base, float* __restrict__ top_val,
                                                            int64_t* __restrict__ top_ord, int32_t* __restrict__ top_cnt) {
    const int lane = threadIdx.x & 63;
    const int f = blockIdx.x * (FT_BLOCK / 64) + (threadIdx.x >> 6);
    if (f >= H) return;
    const int beg = offs[f], end = offs[f + 1];
    if (beg == end) return;
    int c = top_cnt[f];
    float lv = -1.f;
    int64_t lo = 0x7fffffffffffffffll;
    if (lane < c) {
        lv = top_val[(int64_t)f * keep + lane];
        lo = top_ord[(int64_t)f * keep + lane];
    }
    for (int s = beg; s < end; s += 64) {
        float cv = -1.f;
        int64_t co = 0;
        if (s + lane < end) {
            cv = ent_val[s + lane];
            co = ord_base + ent_row[s + lane];
        }
        // candidates that cannot enter a full list are dropped before the serial part
        const float minv = __shfl(lv, keep - 1, 64);
        const int64_t mino = __shfl(lo, keep - 1, 64);
        const bool want = cv > 0.f && (c < keep || ft_beats(cv, co, minv, mino));
        unsigned long long m = __ballot(want);
        while (m) {
            const int j = __ffsll((long long)m) - 1;
            m &= m - 1;
            const float v = __shfl(cv, j, 64);
            const int64_t o = __shfl(co, j, 64);
            // p = how many list elements stay ahead of the candidate (the list is sorted, so they are lanes 0..p-1)
            const int p = __popcll(__ballot(lane < c && ft_beats(lv, lo, v, o)));
            if (p >= keep) continue;
            const float uv = __shfl_up(lv, 1, 64);
            const int64_t uo = __shfl_up(lo, 1, 64);
            if (lane > p) {
                lv = uv;
                lo = uo;
            } else if (lane == p) {
                lv = v;
                lo = o;
            }
            if (c < keep) ++c;
            if (lane >= c) {  // lanes past the list (and past `keep`) stay empty
                lv = -1.f;
                lo = 0x7fffffffffffffffll;
            }
        }
    }
    if (lane < c) {
        top_val[(int64_t)f * keep + lane] = lv;
        top_ord[(int64_t)f * keep + lane] = lo;
    }
    if (lane == 0) top_cnt[f] = c;
}

}  // namespace

extern "C" int64_t wsae_feature_topk_workspace_bytes(int64_t max_entries, int32_t H) {
    if (max_entries < 0 || H < 1) return -1;
    // cnt[H] | offs[H+1] | cursor[H] (int32, rounded up to 256 B) | ent_val[max_entries] | ent_row[max_entries]
    const int64_t head = ((3 * (int64_t)H + 1) * 4 + 255) / 256 * 256;
    return head + 8 * max_entries;
}

extern "C" int wsae_feature_topk_update(const float* vals, const int32_t* idx, int64_t rows, int32_t width, int32_t H,
                                        int32_t keep, int64_t ord_base, float* top_vals, int64_t* top_ord,
                                        int32_t* top_cnt, int64_t* total_active, void* workspace, int64_t workspace_bytes,
                                        void* stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    WSAE_REQUIRE(vals && top_vals && top_ord && top_cnt && total_active && workspace, "wsae_feature_topk_update: null pointer");
    WSAE_REQUIRE(rows >= 0 && width >= 1 && H >= 1 && keep >= 1 && keep <= 64,
                 "wsae_feature_topk_update: need rows >= 0, width >= 1, 1 <= keep <= 64 (got rows %lld width %d keep %d)",
                 (long long)rows, width, keep);
    WSAE_REQUIRE(idx || width == H, "wsae_feature_topk_update: a dense matrix (idx == null) must be [rows][H]");
    WSAE_REQUIRE(rows < (1ll << 32), "wsae_feature_topk_update: at most 2^32 - 1 rows per call");
    const int64_t n = rows * width;
    WSAE_REQUIRE(workspace_bytes >= wsae_feature_topk_workspace_bytes(n, H),
                 "wsae_feature_topk_update: workspace too small (%lld < %lld)", (long long)workspace_bytes,
                 (long long)wsae_feature_topk_workspace_bytes(n, H));
    if (n == 0) return WSAE_OK;
    int32_t* cnt = (int32_t*)workspace;
    int32_t* offs = cnt + H;
    int32_t* cursor = offs + H + 1;
    const int64_t head = ((3 * (int64_t)H + 1) * 4 + 255) / 256 * 256;
    float* ent_val = (float*)((char*)workspace + head);
    uint32_t* ent_row = (uint32_t*)(ent_val + n);
    const int grid = (int)(ceil_div64(n, FT_BLOCK) < 8192 ? ceil_div64(n, FT_BLOCK) : 8192);
    WSAE_HIP_CHECK(hipMemsetAsync(cnt, 0, sizeof(int32_t) * H, stream));
    if (idx) {
        ft_count_kernel<false><<<grid, FT_BLOCK, 0, stream>>>(vals, idx, n, width, H, cnt, (unsigned long long*)total_active);
    } else {
        ft_count_kernel<true><<<grid, FT_BLOCK, 0, stream>>>(vals, idx, n, width, H, cnt, (unsigned long long*)total_active);
    }
    WSAE_LAUNCH_CHECK();
    ft_scan_kernel<<<1, 1024, 0, stream>>>(cnt, offs, cursor, H);
    WSAE_LAUNCH_CHECK();
    if (idx) {
        ft_scatter_kernel<false><<<grid, FT_BLOCK, 0, stream>>>(vals, idx, n, width, H, cursor, ent_val, ent_row);
    } else {
        ft_scatter_kernel<true><<<grid, FT_BLOCK, 0, stream>>>(vals, idx, n, width, H, cursor, ent_val, ent_row);
    }
    WSAE_LAUNCH_CHECK();
    ft_merge_kernel<<<ceil_div(H, FT_BLOCK / 64), FT_BLOCK, 0, stream>>>(offs, ent_val, ent_row, H, keep, ord_base, top_vals,
                                                                         top_ord, top_cnt);
    WSAE_LAUNCH_CHECK();
    return WSAE_OK;
}
