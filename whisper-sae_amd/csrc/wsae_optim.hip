// Optimizer tail and weight maintenance.
//   reference: training.py:186-198 (clip_grad_norm_ -> AdamW.step -> normalize_decoder_weights),
//              model.py:91-96 (F.normalize(decoder.weight, dim=0)), model.py:183-257 (dead features)
#include "wsae_common.h"

int wsae_prepare_launch(wsae_ctx* ctx, const float* params, hipStream_t st);  // wsae_ctx.hip

// ---- global gradient norm: per-block partial sums of (scale*g)^2, fixed reduction order ----------
__global__ void __launch_bounds__(256) sqnorm_kernel(const float* __restrict__ g, int64_t n4, float scale,
                                                     float* __restrict__ part) {
    __shared__ float red[8];
    float a = 0.f;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (int64_t)gridDim.x * 256) {
        const float4 v = ((const float4*)g)[i];
        const float x = v.x * scale, y = v.y * scale, z = v.z * scale, w = v.w * scale;
        a += x * x + y * y + z * z + w * w;
    }
    const float t = block_sum(a, red);
    if (threadIdx.x == 0) part[blockIdx.x] = t;
}

// ---- data parallel (whisper_sae/distributed.py): the summed wire [W_dT | W_e | b_e | b_d | b_pre | fired] (fp32 or bf16)
// becomes the fp32 buffer the optimizer reads, [pack order: W_e | W_dT | b_e | b_d | b_pre | fired], and in the same pass
// the gradient part's sum of squares is left in the norm partials - the separate 9.45 MB norm pass of a DDP step is gone
// (wsae_adamw_step with norm_from_wgrad = 1 takes them).  8 values per thread and trip; fixed reduction order.  The two
// matrices swap places between the layouts (hd8 = H D / 8 groups each), the tail keeps its position.
template <typename WT>
__global__ void __launch_bounds__(256) wire_unpack_kernel(const WT* __restrict__ wire, int64_t n8, int64_t p8, int64_t hd8,
                                                          float* __restrict__ out, float* __restrict__ part,
                                                          const float* metrics_sum, float inv_world, wsae_stats* stats,
                                                          int metric_digits) {
    __shared__ float red[8];
    float a = 0.f;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n8; i += (int64_t)gridDim.x * 256) {
        float f[8];
        if constexpr (sizeof(WT) == 2) {
            const bf16x8 v = ((const bf16x8*)wire)[i];
#pragma unroll
            for (int e = 0; e < 8; ++e) f[e] = (float)v[e];
        } else {
            const float4 v0 = ((const float4*)wire)[2 * i], v1 = ((const float4*)wire)[2 * i + 1];
            f[0] = v0.x; f[1] = v0.y; f[2] = v0.z; f[3] = v0.w; f[4] = v1.x; f[5] = v1.y; f[6] = v1.z; f[7] = v1.w;
        }
        const int64_t o = i < hd8 ? i + hd8 : (i < 2 * hd8 ? i - hd8 : i);  // wire group -> pack group
        ((float4*)out)[2 * o] = make_float4(f[0], f[1], f[2], f[3]);
        ((float4*)out)[2 * o + 1] = make_float4(f[4], f[5], f[6], f[7]);
        if (i < p8) {
#pragma unroll
            for (int e = 0; e < 8; ++e) a = fmaf(f[e], f[e], a);
        }
    }
    const float t = block_sum(a, red);
    if (threadIdx.x == 0) part[blockIdx.x] = t;
    if (metrics_sum && stats && blockIdx.x == 0 && threadIdx.x == 0) {  // mean over the ranks of the per-rank batch means
        const float s0 = metrics_sum[0], s1 = metrics_sum[1];  // (may BE the record's loss / l0 words: summed in place)
        stats->loss = s0 * inv_world;
        stats->l0 = s1 * inv_world;
    }
    if (metric_digits && stats && blockIdx.x == 0 && threadIdx.x == 0) {
        // the digit sums behind the fired indicators (include/wsae.h): exact integers below 256 in either wire dtype
        const WT* d = wire + 8 * n8;
        double ql = 0.0, q0 = 0.0;
        for (int i = 9; i >= 0; --i) ql = ql * 16.0 + (double)(float)d[i];
        for (int i = 17; i >= 10; --i) q0 = q0 * 16.0 + (double)(float)d[i];
        const bool bad = (float)d[18] > 0.f;
        stats->loss = bad ? NAN : (float)(ql / 16777216.0 * (double)inv_world);
        stats->l0 = bad ? NAN : (float)(q0 / 65536.0 * (double)inv_world);
    }
}

extern "C" int wsae_grads_unpack_wire(wsae_ctx* ctx, const void* wire, int32_t wire_dtype, float* grads_ext,
                                      const float* metrics_sum, int32_t world, wsae_stats* stats, void* stream) {
    WSAE_REQUIRE(ctx && wire && grads_ext, "wsae_grads_unpack_wire: null argument");
    WSAE_REQUIRE(wire_dtype == WSAE_DT_F32 || wire_dtype == WSAE_DT_BF16, "wsae_grads_unpack_wire: unknown wire dtype %d", wire_dtype);
    WSAE_REQUIRE(world >= 1, "wsae_grads_unpack_wire: world size %d", world);
    const int64_t n_total = ctx->P + ctx->H, hd = (int64_t)ctx->H * ctx->D;
    // (P = 2 H D + H + 2 D with H, D multiples of 32: every boundary of the layout is a multiple of 8)
    WSAE_REQUIRE(n_total % 8 == 0 && ctx->P % 8 == 0 && hd % 8 == 0, "wsae_grads_unpack_wire: pack %lld not in groups of 8", (long long)ctx->P);
    const int64_t n8 = n_total / 8;
    const int nparts = (int)min((int64_t)WSAE_MAX_PARTIALS, ceil_div64(n8, 256));
    hipStream_t st = (hipStream_t)stream;
    const int digits = (!metrics_sum && ctx->wire_metrics) ? 1 : 0;  // (loss, l0) came over the wire itself
    WSAE_REQUIRE(!digits || world <= 16, "wsae_grads_unpack_wire: the wire's metric digits are exact for at most 16 ranks (%d)", world);
    if (wire_dtype == WSAE_DT_BF16)
        wire_unpack_kernel<bf16_t><<<nparts, 256, 0, st>>>((const bf16_t*)wire, n8, ctx->P / 8, hd / 8, grads_ext, ctx->part_sq,
                                                            metrics_sum, 1.f / (float)world, stats, digits);
    else
        wire_unpack_kernel<float><<<nparts, 256, 0, st>>>((const float*)wire, n8, ctx->P / 8, hd / 8, grads_ext, ctx->part_sq,
                                                           metrics_sum, 1.f / (float)world, stats, digits);
    WSAE_LAUNCH_CHECK();
    ctx->n_sq_parts = nparts;
    return WSAE_OK;
}

// ---- per-feature-row maintenance, one wave per feature row h ---------------------------------------
//   NORMALIZE: W_dT[h,:] /= max(||W_dT[h,:]||_2, 1e-12)   == F.normalize(decoder.weight, dim=0), column h
//   SHADOW   : bf16 shadows of W_e[h,:] and W_dT[h,:], folded bias c[h] = b_e[h] - bf16(W_e)[h,:] . b_pre
//   dead scan: (step_count - last_activated[h]) > threshold counted into the stats record by the last
//              block to arrive (integer atomics: deterministic)        model.py:183-195
#define REFRESH_ROWS 4  // feature rows per block: one per wave
template <bool NORMALIZE, bool SHADOW>
__global__ void __launch_bounds__(256)
refresh_kernel(const float* __restrict__ We, float* __restrict__ WdT, const float* __restrict__ be,
               const float* __restrict__ bpre, bf16_t* __restrict__ We16, bf16_t* __restrict__ WdT16,
               float* __restrict__ cfold, int H, int D, const int64_t* __restrict__ last, const int64_t* __restrict__ step_count,
               int64_t thr, int32_t* __restrict__ counters, wsae_stats* __restrict__ stats) {
    __shared__ int dead_s[4];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    int dead = 0;
    constexpr int RPW = REFRESH_ROWS / 4;
    const int hbase = blockIdx.x * REFRESH_ROWS + wave * RPW;
    // both rows of the wave are fetched up front (<= 8 float4 per lane per row covers D <= 2048);
    // every store is 8 or 16 bytes per lane (2-byte bf16 stores made this kernel store-issue bound)
    float4 wd[RPW][8], we[RPW][8];
#pragma unroll
    for (int r = 0; r < RPW; ++r) {
        const int h = min(hbase + r, H - 1);
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int d = lane * 4 + 256 * i;
            if (d < D) {
                wd[r][i] = *(const float4*)(WdT + (int64_t)h * D + d);
                if (SHADOW) we[r][i] = *(const float4*)(We + (int64_t)h * D + d);
            }
        }
    }
#pragma unroll
    for (int r = 0; r < RPW; ++r) {
        const int h = hbase + r;
        if (h >= H) break;
        float inv = 1.f;
        if (NORMALIZE) {
            float s = 0.f;
#pragma unroll
            for (int i = 0; i < 8; ++i)
                if (lane * 4 + 256 * i < D) {
                    const float4 w = wd[r][i];
                    s += w.x * w.x + w.y * w.y + w.z * w.z + w.w * w.w;
                }
            s = wave_sum(s);
            inv = 1.f / fmaxf(sqrtf(s), 1e-12f);
        }
        float dot = 0.f;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int d = lane * 4 + 256 * i;
            if (d < D) {
                float4 w = wd[r][i];
                if (NORMALIZE) {
                    w.x *= inv; w.y *= inv; w.z *= inv; w.w *= inv;
                    *(float4*)(WdT + (int64_t)h * D + d) = w;
                }
                if (SHADOW) {
                    bf16x4 o, e;
                    o[0] = (bf16_t)w.x; o[1] = (bf16_t)w.y; o[2] = (bf16_t)w.z; o[3] = (bf16_t)w.w;
                    *(bf16x4*)(WdT16 + (int64_t)h * D + d) = o;
                    const float4 ev = we[r][i];
                    e[0] = (bf16_t)ev.x; e[1] = (bf16_t)ev.y; e[2] = (bf16_t)ev.z; e[3] = (bf16_t)ev.w;
                    *(bf16x4*)(We16 + (int64_t)h * D + d) = e;
                    const float4 bp = *(const float4*)(bpre + d);
                    dot = fmaf((float)e[0], bp.x, dot);
                    dot = fmaf((float)e[1], bp.y, dot);
                    dot = fmaf((float)e[2], bp.z, dot);
                    dot = fmaf((float)e[3], bp.w, dot);
                }
            }
        }
        if (SHADOW) {
            dot = wave_sum(dot);
            if (lane == 0) cfold[h] = be[h] - dot;
        }
        if (last) dead += ((*step_count - last[h]) > thr) ? 1 : 0;
    }
    if (!last) return;
    if (lane == 0) dead_s[wave] = dead;
    __syncthreads();
    if (threadIdx.x == 0) {
        unsigned tot;  // the ticket also sums the per-block dead counts
        if (grid_ticket((unsigned long long*)(counters + 16), (unsigned)(dead_s[0] + dead_s[1] + dead_s[2] + dead_s[3]), &tot)) {
            stats->dead_count = (int)tot;
            stats->dead_ratio = (float)tot / (float)H;
        }
    }
}

static int launch_refresh(wsae_ctx* ctx, float* params, bool normalize, const int64_t* last, const int64_t* step_count,
                          int64_t thr, wsae_stats* stats, hipStream_t st) {
    const bool shadow = ctx->prec == WSAE_PREC_BF16;
    if (!normalize && !shadow && !last) return WSAE_OK;
    const int nb = ceil_div(ctx->H, REFRESH_ROWS);
#define RF_ARGS params + ctx->off[0], params + ctx->off[1], params + ctx->off[2], params + ctx->off[4], ctx->We_bf16, \
                ctx->WdT_bf16, ctx->c_fold, ctx->H, ctx->D, last, step_count, thr, ctx->counters, stats
    WSAE_PROF_BEGIN(ctx, WSAE_K_ROWNORM, st);
    if (normalize && shadow) refresh_kernel<true, true><<<nb, 256, 0, st>>>(RF_ARGS);
    else if (normalize) refresh_kernel<true, false><<<nb, 256, 0, st>>>(RF_ARGS);
    else if (shadow) refresh_kernel<false, true><<<nb, 256, 0, st>>>(RF_ARGS);
    else refresh_kernel<false, false><<<nb, 256, 0, st>>>(RF_ARGS);
    WSAE_PROF_END(ctx, WSAE_K_ROWNORM, st);
#undef RF_ARGS
    WSAE_LAUNCH_CHECK();
    return WSAE_OK;
}

// ------------------------------------------------------------------------------------------------
// update_rows_kernel: the whole optimizer tail in ONE launch, organised by feature row h.
//   clip coefficient from the norm partials -> AdamW on W_e[h,:], W_dT[h,:], b_e[h] -> unit-norm
//   W_dT[h,:] -> bf16 shadows + folded bias -> dead-feature count.  b_d and b_pre (2D values) are
//   updated redundantly by every block in registers (the folded bias needs the NEW b_pre) and written
//   back by block 0 only.  One pass: 28 B/param of traffic, no separate AdamW / renorm / refresh passes.
// ------------------------------------------------------------------------------------------------
struct AdamArgs {
    float max_norm, grad_scale, part_scale, decay, beta1, beta2, eps, step_size, bc2_sqrt, inv_bc2_sqrt;
};

// torch.optim.AdamW, single step t: p *= 1 - lr*wd; m = lerp(m, g, 1-b1); v = b2 v + (1-b2) g g;
// denom = sqrt(v)/sqrt(1-b2^t) + eps; p -= (lr/(1-b1^t)) * m/denom.   g is first scaled by gs = clip coefficient *
// grad_scale (1/world under DDP); decay = 1 - lr*wd, step_size = lr/(1-b1^t), bc2_sqrt = sqrt(1-b2^t) come from the host.
__device__ __forceinline__ float adam1(float p, float g, float& m, float& v, const AdamArgs& a, float gs) {
    const float gc = g * gs;
    m = m + (gc - m) * (1.f - a.beta1);
    v = a.beta2 * v + (1.f - a.beta2) * gc * gc;
    // v_sqrt_f32 / v_rcp_f32 (1 ulp each) and a multiplication by 1/sqrt(1-b2^t) instead of the correctly rounded
    // sqrt and two divisions: 40 % fewer vector instructions in the optimizer tail (-1.5 us at cfg 2); the update term is
    // lr-sized, so the parameter moves by < 1e-10 relative against the exact form - inside every pin.
    return p * a.decay - a.step_size * (m * __builtin_amdgcn_rcpf(__builtin_amdgcn_sqrtf(v) * a.inv_bc2_sqrt + a.eps));
}

// Memory-level parallelism: a wave owns ONE feature row and issues every load of it (p, g, m, v of
// the W_e and W_dT rows, the b_e scalars: 16 float4 per lane at D = 384) before anything else; the
// per-block prologue (norm partials -> clip coefficient, new b_pre) then runs underneath those loads.
// NI = float4 per lane and row (D <= 256 NI).  Addresses past the row are clamped instead of
// predicated (predicated loads sit behind exec branches and hipcc drains vmcnt(0) at each of them).
template <int NI>
struct RowRegs {
    float4 p[NI], g[NI], m[NI], v[NI];
};

template <int NI>
__device__ __forceinline__ void row_load(RowRegs<NI>& r, const float* __restrict__ P, const float* __restrict__ G,
                                         const float* __restrict__ M, const float* __restrict__ V, int64_t base, int D,
                                         int lane) {
#pragma unroll
    for (int i = 0; i < NI; ++i) {
        const int64_t o = base + min(lane * 4 + 256 * i, D - 4);
        r.p[i] = *(const float4*)(P + o);
        r.g[i] = nt_load4((const float4*)(G + o));  // (last use of the gradient)
        r.m[i] = *(const float4*)(M + o);
        r.v[i] = *(const float4*)(V + o);
    }
}

template <int NI>
__device__ __forceinline__ void row_adam(RowRegs<NI>& r, const AdamArgs& a, float gs) {
#pragma unroll
    for (int i = 0; i < NI; ++i) {
        r.p[i].x = adam1(r.p[i].x, r.g[i].x, r.m[i].x, r.v[i].x, a, gs);
        r.p[i].y = adam1(r.p[i].y, r.g[i].y, r.m[i].y, r.v[i].y, a, gs);
        r.p[i].z = adam1(r.p[i].z, r.g[i].z, r.m[i].z, r.v[i].z, a, gs);
        r.p[i].w = adam1(r.p[i].w, r.g[i].w, r.m[i].w, r.v[i].w, a, gs);
    }
}

template <bool NORMALIZE, bool SHADOW, int NI>
__global__ void __launch_bounds__(256)
update_rows_kernel(float* __restrict__ P, const float* __restrict__ G, float* __restrict__ M, float* __restrict__ V,
                   int64_t oWe, int64_t oWd, int64_t oBe, int64_t oBd, int64_t oBp, int H, int D,
                   const float* __restrict__ part_sq, int nparts, AdamArgs a, bf16_t* __restrict__ We16,
                   bf16_t* __restrict__ WdT16, float* __restrict__ cfold, int64_t* __restrict__ last,
                   const int64_t* __restrict__ step_count, int64_t thr, int32_t* __restrict__ counters,
                   wsae_stats* __restrict__ stats, float* __restrict__ fired) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* bp_s = (float*)smem;  // [D] updated b_pre
    __shared__ float red[8];
    __shared__ float gs_s;
    __shared__ int dead_s[4];
    __shared__ int last_blk;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    static_assert(REFRESH_ROWS == 4, "one row per wave");
    const int h = blockIdx.x * REFRESH_ROWS + wave;
    const int hc = min(h, H - 1);
    constexpr bool BOTH = NI <= 3;  // both rows fit in registers at once (8 NI float4 per lane)

    RowRegs<NI> re, rd;
    row_load<NI>(re, P, G, M, V, oWe + (int64_t)hc * D, D, lane);
    if (BOTH) row_load<NI>(rd, P, G, M, V, oWd + (int64_t)hc * D, D, lane);
    float be_p = P[oBe + hc], be_g = G[oBe + hc], be_m = M[oBe + hc], be_v = V[oBe + hc];
    int64_t la = 0, sc = 0;
    float fr = 0.f;
    if (last) {
        la = last[hc];
        sc = *step_count;
        if (fired) fr = fired[hc];  // summed over the ranks: > 0 when the feature fired anywhere in this step
    }

    // ---- per-block prologue: clip coefficient, the NEW b_pre (every block computes it from the old
    // state with identical arithmetic; the state itself is written once, by the last block to arrive)
    float sq = 0.f;
    for (int i = threadIdx.x; i < nparts; i += 256) sq += part_sq[i];
    const float tot = block_sum(sq, red);
    if (threadIdx.x == 0) {
        const float nrm = sqrtf(tot) * a.part_scale;
        float coef = 1.f;
        if (a.max_norm > 0.f) coef = fminf(1.f, a.max_norm / (nrm + 1e-6f));
        gs_s = coef * a.grad_scale;
        if (blockIdx.x == 0 && stats) {
            stats->grad_norm = nrm;
            stats->clip_coef = coef;
        }
    }
    __syncthreads();
    const float gs = gs_s;
    for (int d = threadIdx.x; d < D; d += 256) {
        float m = M[oBp + d], v = V[oBp + d];
        bp_s[d] = adam1(P[oBp + d], G[oBp + d], m, v, a, gs);
    }
    __syncthreads();

    int dead = 0;
    if (h < H) {
        // ---- W_e row: AdamW, shadow, folded-bias dot ----
        row_adam<NI>(re, a, gs);
        float dot = 0.f;
#pragma unroll
        for (int i = 0; i < NI; ++i) {
            const int d = lane * 4 + 256 * i;
            if (d < D) {
                const int64_t o = oWe + (int64_t)h * D + d;
                *(float4*)(P + o) = re.p[i]; *(float4*)(M + o) = re.m[i]; *(float4*)(V + o) = re.v[i];
                if (SHADOW) {
                    bf16x4 e;
                    e[0] = (bf16_t)re.p[i].x; e[1] = (bf16_t)re.p[i].y; e[2] = (bf16_t)re.p[i].z; e[3] = (bf16_t)re.p[i].w;
                    *(bf16x4*)(We16 + (int64_t)h * D + d) = e;
                    const float4 bp = *(const float4*)(bp_s + d);
                    dot = fmaf((float)e[0], bp.x, dot); dot = fmaf((float)e[1], bp.y, dot);
                    dot = fmaf((float)e[2], bp.z, dot); dot = fmaf((float)e[3], bp.w, dot);
                }
            }
        }
        // ---- b_e[h] ----
        const float nbe = adam1(be_p, be_g, be_m, be_v, a, gs);
        if (lane == 0) {
            P[oBe + h] = nbe; M[oBe + h] = be_m; V[oBe + h] = be_v;
        }
        if (SHADOW) {
            dot = wave_sum(dot);
            if (lane == 0) cfold[h] = nbe - dot;
        }
        // ---- W_dT row: AdamW, unit norm, shadow ----
        if (!BOTH) row_load<NI>(rd, P, G, M, V, oWd + (int64_t)h * D, D, lane);
        row_adam<NI>(rd, a, gs);
        float s2 = 0.f;
#pragma unroll
        for (int i = 0; i < NI; ++i)
            if (lane * 4 + 256 * i < D) {
                const float4 p = rd.p[i];
                s2 += p.x * p.x + p.y * p.y + p.z * p.z + p.w * p.w;
            }
        float inv = 1.f;
        if (NORMALIZE) {
            s2 = wave_sum(s2);
            inv = 1.f / fmaxf(sqrtf(s2), 1e-12f);
        }
#pragma unroll
        for (int i = 0; i < NI; ++i) {
            const int d = lane * 4 + 256 * i;
            if (d < D) {
                const int64_t o = oWd + (int64_t)h * D + d;
                float4 p = rd.p[i];
                p.x *= inv; p.y *= inv; p.z *= inv; p.w *= inv;
                *(float4*)(P + o) = p; *(float4*)(M + o) = rd.m[i]; *(float4*)(V + o) = rd.v[i];
                if (SHADOW) {
                    bf16x4 o16;
                    o16[0] = (bf16_t)p.x; o16[1] = (bf16_t)p.y; o16[2] = (bf16_t)p.z; o16[3] = (bf16_t)p.w;
                    *(bf16x4*)(WdT16 + (int64_t)h * D + d) = o16;
                }
            }
        }
        if (last) {
            if (fired && lane == 0) {  // DDP clock merge (include/wsae.h, wsae_ctx_set_fired)
                if (fr > 0.f) last[h] = sc;
                fired[h] = 0.f;
            }
            if (fr > 0.f) la = sc;
            dead = ((sc - la) > thr) ? 1 : 0;
        }
    }
    // arrival ticket (low word also sums the dead-feature count); every block's reads of the old
    // b_pre / b_d state precede its ticket, so the last arriver may overwrite that state
    if (lane == 0) dead_s[wave] = dead;
    __syncthreads();
    if (threadIdx.x == 0) {
        unsigned t = 0;
        last_blk = grid_ticket((unsigned long long*)(counters + 16), (unsigned)(dead_s[0] + dead_s[1] + dead_s[2] + dead_s[3]), &t);
        if (last_blk && last && stats) {
            stats->dead_count = (int)t;
            stats->dead_ratio = (float)t / (float)H;
        }
    }
    __syncthreads();
    if (!last_blk) return;
    for (int d = threadIdx.x; d < D; d += 256) {
        float m = M[oBp + d], v = V[oBp + d];
        const float np = adam1(P[oBp + d], G[oBp + d], m, v, a, gs);
        P[oBp + d] = np; M[oBp + d] = m; V[oBp + d] = v;
        float md = M[oBd + d], vd = V[oBd + d];
        const float nd = adam1(P[oBd + d], G[oBd + d], md, vd, a, gs);
        P[oBd + d] = nd; M[oBd + d] = md; V[oBd + d] = vd;
    }
}

template <int NI>
static void launch_update_rows(bool normalize, bool shadow, int nb, size_t sh, hipStream_t st, float* P, const float* G,
                               float* M, float* V, wsae_ctx* ctx, int nparts, const AdamArgs& a, int64_t* last,
                               const int64_t* step_count, int64_t thr, wsae_stats* stats) {
#define UP_ARGS P, G, M, V, ctx->off[0], ctx->off[1], ctx->off[2], ctx->off[3], ctx->off[4], ctx->H, ctx->D, ctx->part_sq, \
                nparts, a, ctx->We_bf16, ctx->WdT_bf16, ctx->c_fold, last, step_count, thr, ctx->counters, stats, ctx->fired
    if (normalize && shadow) update_rows_kernel<true, true, NI><<<nb, 256, sh, st>>>(UP_ARGS);
    else if (normalize) update_rows_kernel<true, false, NI><<<nb, 256, sh, st>>>(UP_ARGS);
    else if (shadow) update_rows_kernel<false, true, NI><<<nb, 256, sh, st>>>(UP_ARGS);
    else update_rows_kernel<false, false, NI><<<nb, 256, sh, st>>>(UP_ARGS);
#undef UP_ARGS
}

extern "C" int wsae_normalize_decoder(wsae_ctx* ctx, float* params, void* stream) {
    WSAE_REQUIRE(ctx && params, "wsae_normalize_decoder: null argument");
    return launch_refresh(ctx, params, true, nullptr, nullptr, 0, nullptr, (hipStream_t)stream);
}

extern "C" int wsae_adamw_step(wsae_ctx* ctx, float* params, const float* grads, float* exp_avg, float* exp_avg_sq,
                               float lr, float beta1, float beta2, float eps, float weight_decay, int32_t step,
                               float max_norm, float grad_scale, int32_t normalize_decoder, int32_t norm_from_wgrad,
                               int64_t* last_activated, const int64_t* step_count, int64_t dead_threshold,
                               wsae_stats* stats, void* stream) {
    WSAE_REQUIRE(ctx && params && grads && exp_avg && exp_avg_sq, "wsae_adamw_step: null argument");
    WSAE_REQUIRE(step >= 1, "wsae_adamw_step: step is the 1-based update count, got %d", step);
    WSAE_REQUIRE(ctx->P % 4 == 0, "wsae_adamw_step: pack size %lld not a multiple of 4", (long long)ctx->P);
    WSAE_REQUIRE(!last_activated || (step_count && stats), "wsae_adamw_step: the dead scan needs step_count and stats");
    hipStream_t st = (hipStream_t)stream;
    const int64_t n4 = ctx->P / 4;
    int nparts;
    float part_scale;
    if (norm_from_wgrad == 2) norm_from_wgrad = ctx->n_sq_parts > 0 ? 1 : 0;  // "if the backward left them" (the ReLU path's two flows)
    if (norm_from_wgrad) {  // partial sums of squares were left by wsae_weight_grads (grads untouched since)
        WSAE_REQUIRE(ctx->n_sq_parts > 0, "wsae_adamw_step: norm_from_wgrad without a preceding wsae_weight_grads");
        nparts = ctx->n_sq_parts;
        part_scale = grad_scale;
    } else {
        nparts = (int)min((int64_t)WSAE_MAX_PARTIALS, ceil_div64(n4, 256 * 2));
        part_scale = 1.f;
        WSAE_PROF_BEGIN(ctx, WSAE_K_SQNORM, st);
        sqnorm_kernel<<<nparts, 256, 0, st>>>(grads, n4, grad_scale, ctx->part_sq);
        WSAE_PROF_END(ctx, WSAE_K_SQNORM, st);
        WSAE_LAUNCH_CHECK();
    }
    ctx->n_sq_parts = 0;
    const double bc1 = 1.0 - pow((double)beta1, (double)step);
    const double bc2 = 1.0 - pow((double)beta2, (double)step);
    const float step_size = (float)((double)lr / bc1);
    const float bc2_sqrt = (float)sqrt(bc2);
    AdamArgs a;
    a.max_norm = max_norm; a.grad_scale = grad_scale; a.part_scale = part_scale; a.decay = 1.f - lr * weight_decay;
    a.beta1 = beta1; a.beta2 = beta2; a.eps = eps; a.step_size = step_size; a.bc2_sqrt = bc2_sqrt;
    a.inv_bc2_sqrt = (float)(1.0 / sqrt(bc2));
    const int nb = ceil_div(ctx->H, REFRESH_ROWS);
    const size_t sh = (size_t)ctx->D * sizeof(float);
    const bool shadow = ctx->prec == WSAE_PREC_BF16;
    WSAE_REQUIRE(ctx->D % 4 == 0 && ctx->D >= 4 && ctx->D <= 2048, "wsae_adamw_step: input_dim %d not in [4, 2048] / 4", ctx->D);
    const int ni = ceil_div(ctx->D, 256);
    WSAE_PROF_BEGIN(ctx, WSAE_K_ADAMW, st);
#define UP_CALL(NI_) launch_update_rows<NI_>(normalize_decoder != 0, shadow, nb, sh, st, params, grads, exp_avg, exp_avg_sq, \
                                            ctx, nparts, a, last_activated, step_count, dead_threshold, stats)
    if (ni <= 1) UP_CALL(1);
    else if (ni == 2) UP_CALL(2);
    else if (ni == 3) UP_CALL(3);
    else if (ni == 4) UP_CALL(4);
    else UP_CALL(8);
#undef UP_CALL
    WSAE_PROF_END(ctx, WSAE_K_ADAMW, st);
    WSAE_LAUNCH_CHECK();
    return WSAE_OK;
}

// ---- dead features -------------------------------------------------------------------------------
// mask[h] = (step_count - last_activated[h]) > threshold   (model.py:183-190, strict)
__global__ void __launch_bounds__(1024)
dead_scan_kernel(const int64_t* __restrict__ last, const int64_t* __restrict__ step_count, int64_t thr, int H,
                 uint8_t* __restrict__ mask, wsae_stats* __restrict__ stats) {
    __shared__ float red[16];
    const int64_t step = *step_count;
    int cnt = 0;
    for (int h = threadIdx.x; h < H; h += 1024) {
        const bool dead = (step - last[h]) > thr;
        if (mask) mask[h] = dead ? 1 : 0;
        cnt += dead ? 1 : 0;
    }
    const float t = block_sum((float)cnt, red);  // exact for H < 2^24
    if (threadIdx.x == 0 && stats) {
        stats->dead_count = (int32_t)t;
        stats->dead_ratio = t / (float)H;
    }
}

extern "C" int wsae_dead_scan(wsae_ctx* ctx, const int64_t* last_activated, const int64_t* step_count,
                              int64_t threshold, uint8_t* mask, wsae_stats* stats, void* stream) {
    WSAE_REQUIRE(ctx && last_activated && step_count, "wsae_dead_scan: null argument");
    hipStream_t st = (hipStream_t)stream;
    WSAE_PROF_BEGIN(ctx, WSAE_K_DEAD_SCAN, st);
    dead_scan_kernel<<<1, 1024, 0, st>>>(last_activated, step_count, threshold, ctx->H, mask, stats);
    WSAE_PROF_END(ctx, WSAE_K_DEAD_SCAN, st);
    WSAE_LAUNCH_CHECK();
    return WSAE_OK;
}

// row_err[b] = sum_d (x[b,d] - recon[b,d])^2      (model.py:230-231)
template <int XDT>
__global__ void __launch_bounds__(256) row_err_kernel(const void* __restrict__ x, const int32_t* __restrict__ rows,
                                                      const float* __restrict__ recon, int B, int D,
                                                      float* __restrict__ row_err, float* __restrict__ resid) {
    const int lane = threadIdx.x & 63;
    const int b = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (b >= B) return;
    const int64_t src = rows ? (int64_t)rows[b] : (int64_t)b;
    float s = 0.f;
    for (int d = lane; d < D; d += 64) {
        const float r = load_act<XDT>(x, src * D + d) - recon[(int64_t)b * D + d];
        s = fmaf(r, r, s);
        if (resid) resid[(int64_t)b * D + d] = r;  // target - predicted (transcoder.py:236)
    }
    s = wave_sum(s);
    if (lane == 0) row_err[b] = s;
}

extern "C" int wsae_row_errors(wsae_ctx* ctx, const void* x, int32_t x_dtype, const int32_t* rows, const float* recon,
                               int32_t B, float* row_err, float* resid, void* stream) {
    WSAE_REQUIRE(ctx && x && recon && row_err && B >= 1, "wsae_row_errors: bad argument");
    hipStream_t st = (hipStream_t)stream;
    if (x_dtype == WSAE_DT_F32)
        row_err_kernel<WSAE_DT_F32><<<ceil_div(B, 4), 256, 0, st>>>(x, rows, recon, B, ctx->D, row_err, resid);
    else
        row_err_kernel<WSAE_DT_BF16><<<ceil_div(B, 4), 256, 0, st>>>(x, rows, recon, B, ctx->D, row_err, resid);
    WSAE_LAUNCH_CHECK();
    return WSAE_OK;
}

// dead list: ascending indices of mask != 0, at most cap (single block, ordered compaction)
__global__ void __launch_bounds__(1024) dead_list_kernel(const uint8_t* __restrict__ mask, int H, int cap,
                                                         int32_t* __restrict__ list, int32_t* __restrict__ n_out) {
    __shared__ int wave_cnt[16];
    __shared__ int base_s;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (threadIdx.x == 0) base_s = 0;
    __syncthreads();
    for (int h0 = 0; h0 < H; h0 += 1024) {
        const int h = h0 + threadIdx.x;
        const bool on = h < H && mask[h] != 0;
        const unsigned long long bm = __ballot(on);
        if (lane == 0) wave_cnt[wave] = __popcll(bm);
        __syncthreads();
        int off = base_s;
        for (int w = 0; w < wave; ++w) off += wave_cnt[w];
        const int pos = off + __popcll(bm & ((1ull << lane) - 1ull));
        if (on && pos < cap) list[pos] = h;
        __syncthreads();
        if (threadIdx.x == 0) {
            int t = 0;
            for (int w = 0; w < 16; ++w) t += wave_cnt[w];
            base_s += t;
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) *n_out = min(base_s, cap);
}

// rows by error descending (ties: lower row first): single-block bitonic sort of 64-bit keys in LDS
__global__ void __launch_bounds__(1024) sort_rows_kernel(const float* __restrict__ row_err, int Br, int npow2,
                                                         int32_t* __restrict__ order) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    uint64_t* key = (uint64_t*)smem;
    for (int i = threadIdx.x; i < npow2; i += 1024) {
        uint64_t k = 0;
        if (i < Br) {
            const uint32_t u = __float_as_uint(row_err[i]);
            const uint32_t o = (u & 0x80000000u) ? ~u : (u | 0x80000000u);
            k = ((uint64_t)o << 32) | (uint32_t)(~(uint32_t)i);
        }
        key[i] = k;
    }
    __syncthreads();
    for (int size = 2; size <= npow2; size <<= 1)
        for (int stride = size >> 1; stride > 0; stride >>= 1) {
            for (int i = threadIdx.x; i < npow2; i += 1024) {
                const int j = i ^ stride;
                if (j > i) {
                    const bool desc = (i & size) == 0;
                    const uint64_t a = key[i], b = key[j];
                    if ((a < b) == desc) {
                        key[i] = b;
                        key[j] = a;
                    }
                }
            }
            __syncthreads();
        }
    for (int i = threadIdx.x; i < Br; i += 1024) order[i] = (int32_t)(~(uint32_t)key[i]);
}

// rewrite dead feature i with the i-th highest-error input row, L2-normalised (model.py:237-255)
template <int XDT>
__global__ void __launch_bounds__(256)
resample_write_kernel(const void* __restrict__ inputs, const int32_t* __restrict__ rows, int Br, int D,
                      const int32_t* __restrict__ dead_list, const int32_t* __restrict__ n_dead,
                      const int32_t* __restrict__ order, float* __restrict__ We, float* __restrict__ WdT,
                      float* __restrict__ be, int64_t* __restrict__ last, const int64_t* __restrict__ step_count,
                      const float* __restrict__ dec_src) {
    const int lane = threadIdx.x & 63;
    const int i = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int n = min(*n_dead, Br);
    if (i >= n) return;
    const int f = dead_list[i];
    const int r = order[i];
    const int64_t src = rows ? (int64_t)rows[r] : (int64_t)r;
    float s = 0.f;
    for (int d = lane; d < D; d += 64) {
        const float v = load_act<XDT>(inputs, src * D + d);
        s = fmaf(v, v, s);
    }
    s = wave_sum(s);
    const float inv = 1.f / fmaxf(sqrtf(s), 1e-12f);
    // decoder column: the same direction (SAE, model.py:251), or the L2-normalised row r of dec_src (transcoders write
    // the normalised residual there, transcoder.py:247-249)
    float invd = inv;
    if (dec_src) {
        float sd = 0.f;
        for (int d = lane; d < D; d += 64) {
            const float v = dec_src[(int64_t)r * D + d];
            sd = fmaf(v, v, sd);
        }
        sd = wave_sum(sd);
        invd = 1.f / fmaxf(sqrtf(sd), 1e-12f);
    }
    for (int d = lane; d < D; d += 64) {
        const float v = load_act<XDT>(inputs, src * D + d);
        We[(int64_t)f * D + d] = v * inv;
        WdT[(int64_t)f * D + d] = dec_src ? dec_src[(int64_t)r * D + d] * invd : v * inv;
    }
    if (lane == 0) {
        be[f] = 0.f;
        last[f] = *step_count;
    }
}

extern "C" int wsae_resample_dead(wsae_ctx* ctx, float* params, const void* inputs, int32_t x_dtype,
                                  const int32_t* rows, int32_t Br, const float* row_err, const uint8_t* dead_mask,
                                  int64_t* last_activated, const int64_t* step_count, int32_t num_cap,
                                  int32_t* n_dead_out, const float* dec_src, void* stream) {
    WSAE_REQUIRE(ctx && params && inputs && row_err && dead_mask && last_activated && step_count && n_dead_out,
                 "wsae_resample_dead: null argument");
    WSAE_REQUIRE(Br >= 1, "wsae_resample_dead: empty resample batch");
    hipStream_t st = (hipStream_t)stream;
    const int H = ctx->H, D = ctx->D;
    int npow2 = 1;
    while (npow2 < Br) npow2 <<= 1;
    WSAE_REQUIRE((size_t)npow2 * 8 <= 128 * 1024, "resample batch %d too large (max 16384 rows)", Br);
    WSAE_REQUIRE(Br <= ctx->maxB, "resample batch %d exceeds max_batch %d", Br, ctx->maxB);
    const int cap = num_cap >= 0 ? min(num_cap, H) : H;
    dead_list_kernel<<<1, 1024, 0, st>>>(dead_mask, H, cap, ctx->dead_list, n_dead_out);
    WSAE_LAUNCH_CHECK();
    sort_rows_kernel<<<1, 1024, (size_t)npow2 * 8, st>>>(row_err, Br, npow2, ctx->row_order);
    WSAE_LAUNCH_CHECK();
    const int nwork = min(cap, Br);
    if (nwork > 0) {
        float* We = params + ctx->off[0];
        float* WdT = params + ctx->off[1];
        float* be = params + ctx->off[2];
        if (x_dtype == WSAE_DT_F32)
            resample_write_kernel<WSAE_DT_F32><<<ceil_div(nwork, 4), 256, 0, st>>>(
                inputs, rows, Br, D, ctx->dead_list, n_dead_out, ctx->row_order, We, WdT, be, last_activated, step_count, dec_src);
        else
            resample_write_kernel<WSAE_DT_BF16><<<ceil_div(nwork, 4), 256, 0, st>>>(
                inputs, rows, Br, D, ctx->dead_list, n_dead_out, ctx->row_order, We, WdT, be, last_activated, step_count, dec_src);
        WSAE_LAUNCH_CHECK();
    }
    return wsae_prepare_launch(ctx, params, st);
}
