// ctx lifecycle, error reporting, flat-pack geometry, derived-weight refresh (wsae_prepare).
#include <stdarg.h>
#include <stdlib.h>
#include <string.h>

#include <new>

#include "wsae_common.h"
#include "wsae_topk.h"
#include <cstdlib>
#include <vector>

static thread_local char g_err[512] = "";
static void prof_free(wsae_ctx* c);

void wsae_set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

extern "C" const char* wsae_last_error(void) { return g_err; }
extern "C" int wsae_version(void) { return WSAE_VERSION; }

extern "C" int64_t wsae_param_count(int32_t D, int32_t H) {
    return 2 * (int64_t)D * H + H + 2 * (int64_t)D;
}

extern "C" int wsae_param_offsets(int32_t D, int32_t H, int64_t off[5]) {
    WSAE_REQUIRE(off != nullptr && D > 0 && H > 0, "wsae_param_offsets: bad arguments");
    off[0] = 0;
    off[1] = (int64_t)D * H;
    off[2] = 2 * (int64_t)D * H;
    off[3] = off[2] + H;
    off[4] = off[3] + D;
    return WSAE_OK;
}

namespace {
// switch to `device` for the lifetime of the object, then back to whatever was current
struct DeviceGuard {
    int prev = -1;
    bool good = false;
    explicit DeviceGuard(int device) {
        if (hipGetDevice(&prev) != hipSuccess) prev = -1;
        good = hipSetDevice(device) == hipSuccess;
    }
    ~DeviceGuard() {
        if (prev >= 0) (void)hipSetDevice(prev);
    }
    bool ok() const { return good; }
};

struct Carver {
    size_t total = 0;
    size_t take(size_t bytes) {
        size_t at = total;
        total += (bytes + 255) & ~(size_t)255;
        return at;
    }
};
}  // namespace

extern "C" int wsae_ctx_create(const wsae_config* cfg, wsae_ctx** out) {
    WSAE_REQUIRE(cfg && out, "wsae_ctx_create: null argument");
    const int D = cfg->input_dim, H = cfg->hidden_dim, K = cfg->k, maxB = cfg->max_batch;
    WSAE_REQUIRE(D >= 32 && D % 32 == 0 && D <= 2048, "input_dim must be a multiple of 32 in [32,2048], got %d", D);
    WSAE_REQUIRE(H >= 32 && H % 32 == 0, "hidden_dim must be a positive multiple of 32, got %d", H);
    WSAE_REQUIRE(K >= 1 && K <= 128 && K <= H, "k must be in [1, min(128, hidden_dim)], got %d", K);
    WSAE_REQUIRE(maxB >= 1, "max_batch must be >= 1, got %d", maxB);
    WSAE_REQUIRE(cfg->precision == WSAE_PREC_BF16 || cfg->precision == WSAE_PREC_FP32, "unknown precision %d",
                 cfg->precision);
    DeviceGuard guard(cfg->device);  // the caller's current device is restored on every return path
    if (!guard.ok()) {
        wsae_set_error("hipSetDevice(%d) failed", cfg->device);
        return WSAE_ERR_HIP;
    }

    wsae_ctx* c = new (std::nothrow) wsae_ctx();
    if (!c) {
        wsae_set_error("out of host memory");
        return WSAE_ERR_NOMEM;
    }
    memset(c, 0, sizeof(*c));
    c->loss_cols = D;
    c->D = D; c->H = H; c->K = K; c->maxB = maxB; c->prec = cfg->precision; c->device = cfg->device;
    if (hipDeviceGetAttribute(&c->cus, hipDeviceAttributeMultiprocessorCount, cfg->device) != hipSuccess || c->cus < 1)
        c->cus = 256;
    c->P = wsae_param_count(D, H);
    wsae_param_offsets(D, H, c->off);

    const size_t esz = (c->prec == WSAE_PREC_BF16) ? 2 : 4;
    const size_t maxBp = ((size_t)maxB + 127) / 128 * 128;  // transposed operands are padded to 128 columns
    Carver cv;
    const size_t o_we = cv.take((size_t)H * D * 2);
    const size_t o_wd = cv.take((size_t)H * D * 2);
    const size_t o_cf = cv.take((size_t)H * 4);
    const size_t o_xb = cv.take((size_t)maxB * D * esz);
    const size_t o_xT = cv.take(maxBp * D * esz);
    const size_t o_gT = cv.take(maxBp * D * esz);
    const size_t o_g = cv.take((size_t)maxB * D * 4);
    const size_t o_gb = cv.take((size_t)maxB * D * 2);
    const size_t o_pre = cv.take((size_t)maxB * H * 4);
    const size_t o_sm = cv.take((size_t)maxB * ((H + 15) / 16) * 4);
    const size_t o_pl = cv.take(WSAE_MAX_PARTIALS * 4);
    const size_t o_p0 = cv.take(WSAE_MAX_PARTIALS * 4);
    const size_t o_pd = cv.take((size_t)WSAE_MAX_PARTIALS * D * 4);
    const size_t o_ps = cv.take((WSAE_MAX_PARTIALS + 64) * 4);
    const size_t o_d2 = cv.take((size_t)64 * D * 4);
    const size_t o_cn = cv.take((size_t)H * 4);
    const size_t o_ct = cv.take((16 + 3 * 2 * TICKET_WORDS) * 4);  // 16 ints + three tickets (optimizer, decode, grad finish)
    const size_t o_ws = cv.take((size_t)WSAE_WGRAD_MAX_SPLIT * 2 * H * D * 4);
    const size_t o_ds = cv.take((size_t)2 * WSAE_WGRAD_MAX_SPLIT * H * 4);  // (16 splits when one matrix is contracted per launch)
    const size_t o_dp = cv.take((size_t)((H + 31) / 32) * D * 4);
    const size_t o_ep = cv.take((size_t)maxB * K * 4);
    const size_t o_eh = cv.take((size_t)maxB * K * 4);
    const size_t o_ed = cv.take((size_t)maxB * K * 4);
    const size_t o_eo = cv.take((size_t)((maxB + 31) / 32) * ((H + 127) / 128 + 1) * 4);
    const size_t o_dl = cv.take((size_t)H * 4);
    const size_t o_ro = cv.take((size_t)maxB * 4);
    const size_t o_tm = cv.take(TG_WORDS * 4);
    char* base = nullptr;
    hipError_t e = hipMalloc((void**)&base, cv.total);
    if (e != hipSuccess) {
        wsae_set_error("hipMalloc(%zu bytes of workspace) failed: %s", cv.total, hipGetErrorString(e));
        delete c;
        return WSAE_ERR_NOMEM;
    }
    e = hipMemset(base, 0, cv.total);
    if (e != hipSuccess) {
        wsae_set_error("hipMemset failed: %s", hipGetErrorString(e));
        (void)hipFree(base);
        delete c;
        return WSAE_ERR_HIP;
    }
    c->We_bf16 = (bf16_t*)(base + o_we);
    c->WdT_bf16 = (bf16_t*)(base + o_wd);
    c->c_fold = (float*)(base + o_cf);
    c->xb = base + o_xb;
    c->xT = base + o_xT;
    c->gT = base + o_gT;
    c->g = (float*)(base + o_g);
    c->gb = (bf16_t*)(base + o_gb);
    c->pre = (float*)(base + o_pre);
    c->smax = (float*)(base + o_sm);
    c->part_loss = (float*)(base + o_pl);
    c->part_l0 = (float*)(base + o_p0);
    c->part_dbd = (float*)(base + o_pd);
    c->part_sq = (float*)(base + o_ps);
    c->dbd2 = (float*)(base + o_d2);
    c->colnorm = (float*)(base + o_cn);
    c->counters = (int32_t*)(base + o_ct);
    c->wg_slabs = (float*)(base + o_ws);
    c->dbe_slab = (float*)(base + o_ds);
    c->dbpre_part = (float*)(base + o_dp);
    c->ent_pos = (uint32_t*)(base + o_ep);
    c->ent_hid = base + o_eh;
    c->ent_dpre = base + o_ed;
    c->ent_off = (int32_t*)(base + o_eo);
    c->dead_list = (int32_t*)(base + o_dl);
    c->row_order = (int32_t*)(base + o_ro);
    c->tmin = (uint32_t*)(base + o_tm);
    const char* sp = getenv("WSAE_STRIP_PREDICT");  // (A/B runs: "0" creates contexts with the selective strip stores off)
    c->strip_predict = (sp && sp[0] == '0') ? 0 : 1;
    const char* ss = getenv("WSAE_STRIP_SAFETY");
    c->tg_fixed_s = ss ? (float)atof(ss) : 0.f;
    c->ws_bytes = cv.total;
    *out = c;
    return WSAE_OK;
}

extern "C" int wsae_ctx_destroy(wsae_ctx* ctx) {
    if (!ctx) return WSAE_OK;
    prof_free(ctx);
    if (ctx->relu_ws) (void)hipFree(ctx->relu_ws);
    if (ctx->We_bf16) (void)hipFree((void*)ctx->We_bf16);  // base of the single allocation
    delete ctx;
    return WSAE_OK;
}

// ---- kernel timing -------------------------------------------------------------------------------
static const char* const k_names[WSAE_K_COUNT] = {"stage_batch", "encode_gemm", "topk", "decode", "bucket", "wgrad",
                                                  "wgrad_reduce", "sqnorm", "adamw", "rownorm", "prepare", "dead_scan"};

extern "C" const char* wsae_kernel_name(int32_t id) { return (id >= 0 && id < WSAE_K_COUNT) ? k_names[id] : "?"; }

static void prof_free(wsae_ctx* c) {
    for (int k = 0; k < WSAE_K_COUNT; ++k) {
        if (c->prof.ev[k]) {
            for (int i = 0; i < 2 * c->prof.max_samples; ++i) (void)hipEventDestroy(c->prof.ev[k][i]);
            delete[] c->prof.ev[k];
            c->prof.ev[k] = nullptr;
        }
        c->prof.count[k] = 0;
    }
    c->prof.mask = 0;
    c->prof.max_samples = 0;
}

extern "C" int wsae_profile_disable(wsae_ctx* ctx) {
    WSAE_REQUIRE(ctx, "wsae_profile_disable: null ctx");
    prof_free(ctx);
    return WSAE_OK;
}

extern "C" int wsae_profile_enable(wsae_ctx* ctx, int32_t kernel_id, int32_t max_samples) {
    WSAE_REQUIRE(ctx && kernel_id >= -1 && kernel_id < WSAE_K_COUNT && max_samples >= 1 && max_samples <= 65536,
                 "wsae_profile_enable: bad argument");
    prof_free(ctx);
    ctx->prof.max_samples = max_samples;
    for (int k = 0; k < WSAE_K_COUNT; ++k) {
        if (kernel_id != -1 && kernel_id != k) continue;
        ctx->prof.ev[k] = new (std::nothrow) hipEvent_t[2 * (size_t)max_samples];
        if (!ctx->prof.ev[k]) {
            wsae_set_error("out of host memory");
            return WSAE_ERR_NOMEM;
        }
        for (int i = 0; i < 2 * max_samples; ++i) WSAE_HIP_CHECK(hipEventCreate(&ctx->prof.ev[k][i]));
        ctx->prof.mask |= 1u << k;
    }
    return WSAE_OK;
}

extern "C" int wsae_profile_read(wsae_ctx* ctx, int32_t kernel_id, int32_t* n_launches, double* total_ms) {
    WSAE_REQUIRE(ctx && kernel_id >= 0 && kernel_id < WSAE_K_COUNT && n_launches && total_ms,
                 "wsae_profile_read: bad argument");
    *n_launches = 0;
    *total_ms = 0.0;
    if (!ctx->prof.ev[kernel_id]) return WSAE_OK;
    const int n = ctx->prof.count[kernel_id];
    double tot = 0.0;
    for (int i = 0; i < n; ++i) {
        WSAE_HIP_CHECK(hipEventSynchronize(ctx->prof.ev[kernel_id][2 * i + 1]));
        float ms = 0.f;
        WSAE_HIP_CHECK(hipEventElapsedTime(&ms, ctx->prof.ev[kernel_id][2 * i], ctx->prof.ev[kernel_id][2 * i + 1]));
        tot += ms;
    }
    *n_launches = n;
    *total_ms = tot;
    return WSAE_OK;
}

extern "C" int wsae_ctx_set_fired(wsae_ctx* ctx, float* fired) {
    WSAE_REQUIRE(ctx, "wsae_ctx_set_fired: null ctx");
    ctx->fired = fired;
    return WSAE_OK;
}

// Selective strip stores of the encoder GEMM (wsae_topk.h, "strip store threshold"; include/wsae.h).
static uint32_t host_f32_ord(float f) {
    uint32_t u;
    memcpy(&u, &f, 4);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
extern "C" int wsae_ctx_set_strip_predict(wsae_ctx* ctx, int32_t on, float assume_store_threshold) {
    WSAE_REQUIRE(ctx, "wsae_ctx_set_strip_predict: null ctx");
    ctx->strip_predict = on ? 1 : 0;
    ctx->pred_valid = 0;
    // the next encoder launch moves on to group tmin_cur + 1 and reads the other two.
    // NaN: forget the history (the next launch stores every strip, the margin starts over); otherwise make the next launch
    // store with exactly this threshold (a predecessor whose minimum was this value, margin 1, no misses; none before it)
    WSAE_HIP_CHECK(hipDeviceSynchronize());
    const bool forget = assume_store_threshold != assume_store_threshold;
    const float one = 1.0f;
    uint32_t hdr[3] = {forget ? 0u : host_f32_ord(assume_store_threshold), 0u, 0u};  // minimum slot 0, margin, misses
    if (!forget) memcpy(&hdr[1], &one, 4);
    uint32_t* prev = ctx->tmin + ctx->tmin_cur * TG_GROUP_WORDS;  // the group the next launch sees as its predecessor
    WSAE_HIP_CHECK(hipMemset(prev, 0, TG_GROUP_WORDS * 4));
    WSAE_HIP_CHECK(hipMemset(ctx->tmin + ((ctx->tmin_cur + 2) % 3) * TG_GROUP_WORDS, 0, TG_GROUP_WORDS * 4));
    WSAE_HIP_CHECK(hipMemcpy(prev, hdr, sizeof(hdr), hipMemcpyHostToDevice));
    return WSAE_OK;
}
extern "C" int wsae_ctx_strip_stats(wsae_ctx* ctx, int64_t* refilled_rows, float* last_min_threshold, float* margin) {
    WSAE_REQUIRE(ctx && refilled_rows && last_min_threshold && margin, "wsae_ctx_strip_stats: null argument");
    std::vector<uint32_t> h(TG_WORDS);
    WSAE_HIP_CHECK(hipDeviceSynchronize());
    WSAE_HIP_CHECK(hipMemcpy(h.data(), ctx->tmin, h.size() * 4, hipMemcpyDeviceToHost));
    *refilled_rows = (int64_t)h[TG_REFILLED];
    uint32_t o = 0xFFFFFFFFu;
    for (int i = 0; i < TG_SLOTS; ++i) {
        const uint32_t s = h[ctx->tmin_cur * TG_GROUP_WORDS + i * TG_SLOT_STRIDE];
        if (s != 0u && s < o) o = s;
    }
    float f = NAN;
    if (o != 0u && o != 0xFFFFFFFFu) {
        const uint32_t u = (o & 0x80000000u) ? (o & 0x7FFFFFFFu) : ~o;
        memcpy(&f, &u, 4);
    }
    *last_min_threshold = f;
    memcpy(margin, &h[ctx->tmin_cur * TG_GROUP_WORDS + TG_HDR_S], 4);
    return WSAE_OK;
}

extern "C" int wsae_ctx_set_wire_metrics(wsae_ctx* ctx, const float* loss_l0) {
    WSAE_REQUIRE(ctx, "wsae_ctx_set_wire_metrics: null ctx");
    ctx->wire_metrics = loss_l0;
    return WSAE_OK;
}

extern "C" int wsae_ctx_set_loss_cols(wsae_ctx* ctx, int32_t cols) {
    WSAE_REQUIRE(ctx && cols >= 1 && cols <= ctx->D, "wsae_ctx_set_loss_cols: columns must be in [1, input_dim]");
    ctx->loss_cols = cols;
    return WSAE_OK;
}

extern "C" size_t wsae_ctx_workspace_bytes(const wsae_ctx* ctx) { return ctx ? ctx->ws_bytes : 0; }

// ------------------------------------------------------------------------------------------------
// wsae_prepare: bf16 shadow of W_e + folded bias.  One wave per feature row.
//   c[h] = b_e[h] - sum_d bf16(W_e[h,d]) * b_pre[d]      (oracle: pre_activation, mode "amp")
// ------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) prepare_kernel(const float* __restrict__ We, const float* __restrict__ WdT,
                                                      const float* __restrict__ be, const float* __restrict__ bpre,
                                                      bf16_t* __restrict__ We16, bf16_t* __restrict__ WdT16,
                                                      float* __restrict__ cfold, int H, int D) {
    const int lane = threadIdx.x & 63;
    const int h = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (h >= H) return;
    const float* row = We + (int64_t)h * D;
    bf16_t* orow = We16 + (int64_t)h * D;
    const float* drow = WdT + (int64_t)h * D;
    bf16_t* odrow = WdT16 + (int64_t)h * D;
    float dot = 0.f;
    for (int d = lane * 4; d < D; d += 256) {
        const float4 wd = *(const float4*)(drow + d);
        bf16x4 o, e;
        o[0] = (bf16_t)wd.x; o[1] = (bf16_t)wd.y; o[2] = (bf16_t)wd.z; o[3] = (bf16_t)wd.w;
        *(bf16x4*)(odrow + d) = o;
        const float4 w = *(const float4*)(row + d);
        e[0] = (bf16_t)w.x; e[1] = (bf16_t)w.y; e[2] = (bf16_t)w.z; e[3] = (bf16_t)w.w;
        *(bf16x4*)(orow + d) = e;
        const float4 bp = *(const float4*)(bpre + d);
        dot = fmaf((float)e[0], bp.x, dot);
        dot = fmaf((float)e[1], bp.y, dot);
        dot = fmaf((float)e[2], bp.z, dot);
        dot = fmaf((float)e[3], bp.w, dot);
    }
    dot = wave_sum(dot);
    if (lane == 0) cfold[h] = be[h] - dot;
}

int wsae_prepare_launch(wsae_ctx* ctx, const float* params, hipStream_t st) {
    if (ctx->prec != WSAE_PREC_BF16) return WSAE_OK;
    WSAE_PROF_BEGIN(ctx, WSAE_K_PREPARE, st);
    prepare_kernel<<<ceil_div(ctx->H, 4), 256, 0, st>>>(params + ctx->off[0], params + ctx->off[1],
                                                        params + ctx->off[2], params + ctx->off[4], ctx->We_bf16,
                                                        ctx->WdT_bf16, ctx->c_fold, ctx->H, ctx->D);
    WSAE_PROF_END(ctx, WSAE_K_PREPARE, st);
    WSAE_LAUNCH_CHECK();
    return WSAE_OK;
}

extern "C" int wsae_prepare(wsae_ctx* ctx, const float* params, void* stream) {
    WSAE_REQUIRE(ctx && params, "wsae_prepare: null argument");
    return wsae_prepare_launch(ctx, params, (hipStream_t)stream);
}
