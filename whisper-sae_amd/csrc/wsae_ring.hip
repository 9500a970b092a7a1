// On-device activation ring buffer + dense-code decode.
//   reference: the feed half of src/whisper_sae/data/feature_cache.py:169-197
//   (torch.load -> TensorDataset -> DataLoader(shuffle=True, pin_memory)) is replaced by rows resident
//   in HBM and batches described as row-index lists the kernels gather from.
#include <new>

#include "wsae_common.h"

struct wsae_ring {
    int device, dim, dtype;
    int64_t cap, size, head;
    void* data;
};

extern "C" int wsae_ring_create(int32_t device, int64_t capacity_rows, int32_t dim, int32_t dtype, wsae_ring** out) {
    WSAE_REQUIRE(out && capacity_rows >= 1 && dim >= 1, "wsae_ring_create: bad argument");
    WSAE_REQUIRE(dtype == WSAE_DT_F32 || dtype == WSAE_DT_BF16, "wsae_ring_create: unknown dtype %d", dtype);
    int prev = -1;
    (void)hipGetDevice(&prev);
    struct Restore {  // the caller's current device comes back on every return path
        int d;
        ~Restore() { if (d >= 0) (void)hipSetDevice(d); }
    } restore{prev};
    WSAE_HIP_CHECK(hipSetDevice(device));
    wsae_ring* r = new (std::nothrow) wsae_ring();
    if (!r) {
        wsae_set_error("out of host memory");
        return WSAE_ERR_NOMEM;
    }
    r->device = device; r->dim = dim; r->dtype = dtype; r->cap = capacity_rows; r->size = 0; r->head = 0;
    const size_t bytes = (size_t)capacity_rows * dim * (dtype == WSAE_DT_BF16 ? 2 : 4);
    hipError_t e = hipMalloc(&r->data, bytes);
    if (e != hipSuccess) {
        wsae_set_error("hipMalloc(%zu bytes for the activation ring) failed: %s", bytes, hipGetErrorString(e));
        delete r;
        return WSAE_ERR_NOMEM;
    }
    *out = r;
    return WSAE_OK;
}

extern "C" int wsae_ring_destroy(wsae_ring* ring) {
    if (!ring) return WSAE_OK;
    if (ring->data) (void)hipFree(ring->data);
    delete ring;
    return WSAE_OK;
}

extern "C" void* wsae_ring_data(wsae_ring* ring) { return ring ? ring->data : nullptr; }
extern "C" int64_t wsae_ring_size(const wsae_ring* ring) { return ring ? ring->size : 0; }

template <int SDT, int DDT>
__global__ void __launch_bounds__(256) ring_push_kernel(const void* __restrict__ src, void* __restrict__ dst,
                                                        int64_t n_rows, int dim, int64_t head, int64_t cap) {
    const int64_t n = n_rows * dim;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        const int64_t r = i / dim;
        const int d = (int)(i - r * dim);
        const int64_t slot = (head + r) % cap;
        const float v = load_act<SDT>(src, i);
        if (DDT == WSAE_DT_BF16)
            ((bf16_t*)dst)[slot * dim + d] = (bf16_t)v;
        else
            ((float*)dst)[slot * dim + d] = v;
    }
}

extern "C" int wsae_ring_push(wsae_ring* ring, const void* src, int32_t src_dtype, int64_t n_rows, void* stream) {
    WSAE_REQUIRE(ring && src && n_rows >= 0, "wsae_ring_push: bad argument");
    WSAE_REQUIRE(src_dtype == WSAE_DT_F32 || src_dtype == WSAE_DT_BF16, "wsae_ring_push: unknown dtype %d", src_dtype);
    if (n_rows == 0) return WSAE_OK;
    if (n_rows > ring->cap) {  // only the newest cap rows can survive
        const int64_t skip = n_rows - ring->cap;
        src = (const char*)src + (size_t)skip * ring->dim * (src_dtype == WSAE_DT_BF16 ? 2 : 4);
        ring->head = (ring->head + skip) % ring->cap;
        n_rows = ring->cap;
    }
    hipStream_t st = (hipStream_t)stream;
    const int nb = (int)min((int64_t)4096, ceil_div64(n_rows * ring->dim, 256));
#define PUSH(S, D_) ring_push_kernel<S, D_><<<nb, 256, 0, st>>>(src, ring->data, n_rows, ring->dim, ring->head, ring->cap)
    if (src_dtype == WSAE_DT_F32 && ring->dtype == WSAE_DT_F32) PUSH(WSAE_DT_F32, WSAE_DT_F32);
    else if (src_dtype == WSAE_DT_F32) PUSH(WSAE_DT_F32, WSAE_DT_BF16);
    else if (ring->dtype == WSAE_DT_F32) PUSH(WSAE_DT_BF16, WSAE_DT_F32);
    else PUSH(WSAE_DT_BF16, WSAE_DT_BF16);
#undef PUSH
    WSAE_LAUNCH_CHECK();
    ring->head = (ring->head + n_rows) % ring->cap;
    ring->size = min(ring->cap, ring->size + n_rows);
    return WSAE_OK;
}

// ---- producer side: the final LayerNorm of the Whisper stack applied to a block of hidden
// states on their way into the ring - one wave per row: mean and variance by wave reductions (two passes over the
// row held in registers), then gamma * (v - mean) * rstd + beta written in the ring's dtype.  Replaces
// layer_norm(activation).cpu() -> list -> torch.cat -> disk of /root/reference/src/whisper_sae/sae/hooks.py:86-92.
template <int SRC, int DST, int VPL>
__global__ void __launch_bounds__(256) ring_push_ln_kernel(const void* __restrict__ src, void* __restrict__ dst,
                                                           int64_t n_rows, int dim, int64_t head, int64_t cap,
                                                           const float* __restrict__ gamma,
                                                           const float* __restrict__ beta, float eps) {
    const int lane = threadIdx.x & 63;
    const int64_t r = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (r >= n_rows) return;
    float v[VPL];
    float sum = 0.f;
#pragma unroll
    for (int i = 0; i < VPL; ++i) {
        const int d = lane + 64 * i;
        v[i] = 0.f;
        if (d < dim) v[i] = SRC == WSAE_DT_F32 ? ((const float*)src)[r * dim + d] : (float)((const bf16_t*)src)[r * dim + d];
        sum += v[i];
    }
    const float mean = wave_sum(sum) / (float)dim;
    float sq = 0.f;
#pragma unroll
    for (int i = 0; i < VPL; ++i) {
        const float c = lane + 64 * i < dim ? v[i] - mean : 0.f;
        sq = fmaf(c, c, sq);
    }
    const float rstd = rsqrtf(wave_sum(sq) / (float)dim + eps);  // biased variance, as torch.nn.LayerNorm
    const int64_t slot = (head + r) % cap;
#pragma unroll
    for (int i = 0; i < VPL; ++i) {
        const int d = lane + 64 * i;
        if (d < dim) {
            const float y = (v[i] - mean) * rstd * gamma[d] + beta[d];
            if (DST == WSAE_DT_F32) ((float*)dst)[slot * dim + d] = y;
            else ((bf16_t*)dst)[slot * dim + d] = (bf16_t)y;
        }
    }
}

extern "C" int wsae_ring_push_layernorm(wsae_ring* ring, const void* src, int32_t src_dtype, int64_t n_rows,
                                        const float* gamma, const float* beta, float eps, void* stream) {
    WSAE_REQUIRE(ring && src && gamma && beta && n_rows >= 0, "wsae_ring_push_layernorm: bad argument");
    WSAE_REQUIRE(src_dtype == WSAE_DT_F32 || src_dtype == WSAE_DT_BF16, "wsae_ring_push_layernorm: unknown dtype %d", src_dtype);
    WSAE_REQUIRE(ring->dim <= 2048, "wsae_ring_push_layernorm: row width %d > 2048", ring->dim);
    if (n_rows == 0) return WSAE_OK;
    if (n_rows > ring->cap) {  // only the newest cap rows can survive
        const int64_t skip = n_rows - ring->cap;
        src = (const char*)src + (size_t)skip * ring->dim * (src_dtype == WSAE_DT_BF16 ? 2 : 4);
        ring->head = (ring->head + skip) % ring->cap;
        n_rows = ring->cap;
    }
    hipStream_t st = (hipStream_t)stream;
    const unsigned nb = (unsigned)ceil_div64(n_rows, 4);
#define PUSH_LN(S, D_, V) ring_push_ln_kernel<S, D_, V><<<nb, 256, 0, st>>>(src, ring->data, n_rows, ring->dim, ring->head, ring->cap, gamma, beta, eps)
#define PUSH_LN_V(S, D_)                               \
    do {                                               \
        if (ring->dim <= 512) PUSH_LN(S, D_, 8);       \
        else PUSH_LN(S, D_, 32);                       \
    } while (0)
    if (src_dtype == WSAE_DT_F32 && ring->dtype == WSAE_DT_F32) PUSH_LN_V(WSAE_DT_F32, WSAE_DT_F32);
    else if (src_dtype == WSAE_DT_F32) PUSH_LN_V(WSAE_DT_F32, WSAE_DT_BF16);
    else if (ring->dtype == WSAE_DT_F32) PUSH_LN_V(WSAE_DT_BF16, WSAE_DT_F32);
    else PUSH_LN_V(WSAE_DT_BF16, WSAE_DT_BF16);
#undef PUSH_LN_V
#undef PUSH_LN
    WSAE_LAUNCH_CHECK();
    ring->head = (ring->head + n_rows) % ring->cap;
    ring->size = min(ring->cap, ring->size + n_rows);
    return WSAE_OK;
}

// ---- seeded shuffle: a bijection of [0, n) --------------------------------------------------------
// Feistel network on the enclosing power-of-two domain, cycle-walked back into [0, n).
__host__ __device__ __forceinline__ uint64_t mix64(uint64_t z) {
    z += 0x9E3779B97F4A7C15ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}

__host__ __device__ __forceinline__ uint64_t feistel_perm(uint64_t i, uint64_t n, uint64_t key, int half_bits) {
    const uint64_t mask = (1ull << half_bits) - 1ull;
    uint64_t v = i;
    do {
        uint64_t l = v >> half_bits, r = v & mask;
#pragma unroll
        for (int round = 0; round < 4; ++round) {
            const uint64_t f = mix64(r ^ key ^ ((uint64_t)round << 56)) & mask;
            const uint64_t nl = r;
            r = l ^ f;
            l = nl;
        }
        v = (l << half_bits) | r;
    } while (v >= n);
    return v;
}

__global__ void __launch_bounds__(256) ring_sample_kernel(uint64_t key, int64_t size, int half_bits, int64_t offset,
                                                          int n, int32_t* __restrict__ out) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    out[i] = (int32_t)feistel_perm((uint64_t)((offset + i) % size), (uint64_t)size, key, half_bits);
}

extern "C" int wsae_ring_sample(wsae_ring* ring, uint64_t seed, int64_t epoch, int64_t offset, int32_t n,
                                int32_t* rows_out, void* stream) {
    WSAE_REQUIRE(ring && rows_out && n >= 1 && offset >= 0, "wsae_ring_sample: bad argument");
    WSAE_REQUIRE(ring->size >= 1, "wsae_ring_sample: the ring is empty");
    WSAE_REQUIRE(ring->size < (1ll << 31), "wsae_ring_sample: ring too large for int32 row indices");
    int bits = 1;
    while ((1ll << bits) < ring->size) ++bits;
    const int half_bits = (bits + 1) / 2;  // domain 2^(2*half_bits) >= size, < 4*size
    const uint64_t key = mix64(mix64(seed) ^ (uint64_t)epoch * 0xD1342543DE82EF95ull);
    ring_sample_kernel<<<ceil_div(n, 256), 256, 0, (hipStream_t)stream>>>(key, ring->size, half_bits, offset, n, rows_out);
    WSAE_LAUNCH_CHECK();
    return WSAE_OK;
}

// ---- synthetic fill: bit-identical to oracle/synth.py normal(seed, stream 0) ----------------------
__global__ void __launch_bounds__(256) ring_fill_kernel(void* __restrict__ dst, int dtype, int64_t n, uint64_t key,
                                                        double sd) {
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        const uint64_t w = mix64((uint64_t)i * 0xD1342543DE82EF95ull + key);
        const uint64_t s = (w & 0xFFFF) + ((w >> 16) & 0xFFFF) + ((w >> 32) & 0xFFFF) + (w >> 48);
        const float v = (float)(((double)s - 131070.0) / sd);
        if (dtype == WSAE_DT_BF16)
            ((bf16_t*)dst)[i] = (bf16_t)v;
        else
            ((float*)dst)[i] = v;
    }
}

extern "C" int wsae_ring_fill_synthetic(wsae_ring* ring, uint64_t seed, int64_t n_rows, void* stream) {
    WSAE_REQUIRE(ring && n_rows >= 1 && n_rows <= ring->cap, "wsae_ring_fill_synthetic: n_rows outside [1, capacity]");
    uint64_t key = mix64(seed);
    key = mix64(key ^ 1ull);  // stream 0 of oracle/synth.py: key ^ (0 * 0x100000001B3 + 1)
    const double sd = sqrt(4.0 * (65536.0 * 65536.0 - 1.0) / 12.0);
    const int64_t n = n_rows * ring->dim;
    const int nb = (int)min((int64_t)4096, ceil_div64(n, 256));
    ring_fill_kernel<<<nb, 256, 0, (hipStream_t)stream>>>(ring->data, ring->dtype, n, key, sd);
    WSAE_LAUNCH_CHECK();
    ring->size = n_rows;
    ring->head = n_rows % ring->cap;
    return WSAE_OK;
}

// ---- dense-code decode (API path: TopKSAE.decode on an arbitrary [B,H] code, model.py:120-129) ----
__global__ void __launch_bounds__(256) decode_dense_kernel(const float* __restrict__ WdT, const float* __restrict__ bd,
                                                           const float* __restrict__ bpre, const float* __restrict__ hidden,
                                                           int B, int H, int D, float* __restrict__ recon) {
    const int lane = threadIdx.x & 63;
    const int b = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (b >= B) return;
    const float* hrow = hidden + (int64_t)b * H;
    for (int d0 = 0; d0 < D; d0 += 64 * 8) {
        float acc[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const int d = d0 + lane + 64 * e;
            acc[e] = d < D ? bd[d] + (bpre ? bpre[d] : 0.f) : 0.f;
        }
        for (int h0 = 0; h0 < H; h0 += 64) {
            const float hv = (h0 + lane < H) ? hrow[h0 + lane] : 0.f;
            unsigned long long m = __ballot(hv != 0.f);
            while (m) {
                const int j = __ffsll((long long)m) - 1;
                m &= m - 1;
                const float v = __uint_as_float(__builtin_amdgcn_readlane(__float_as_uint(hv), j));
                const float* w = WdT + (int64_t)(h0 + j) * D;
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const int d = d0 + lane + 64 * e;
                    if (d < D) acc[e] = fmaf(v, w[d], acc[e]);
                }
            }
        }
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const int d = d0 + lane + 64 * e;
            if (d < D) recon[(int64_t)b * D + d] = acc[e];
        }
    }
}

extern "C" int wsae_decode_dense(wsae_ctx* ctx, const float* params, const float* hidden, int32_t B, float* recon,
                                 void* stream) {
    WSAE_REQUIRE(ctx && params && hidden && recon && B >= 1, "wsae_decode_dense: bad argument");
    decode_dense_kernel<<<ceil_div(B, 4), 256, 0, (hipStream_t)stream>>>(
        params + ctx->off[1], params + ctx->off[3], params + ctx->off[4], hidden, B, ctx->H, ctx->D, recon);
    WSAE_LAUNCH_CHECK();
    return WSAE_OK;
}
