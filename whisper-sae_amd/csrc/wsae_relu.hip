// ReLU + L1 sparse autoencoder, dense path (reference ReLUSAE, model.py:260-322).
//   forward : hidden = relu(x W_e^T + b_e) ; recon = hidden W_d^T + b_d
//             loss = mean((recon - x)^2) + sparsity_weight * mean(|hidden|)      (means over B*D and B*H)
//   backward: g = 2 (recon - x) / (B D) ; dW_d = g^T hidden ; db_d = sum_b g
//             dpre = (g W_d + sparsity_weight / (B H)) * 1[hidden > 0]
//             dW_e = dpre^T x ; db_e = sum_b dpre                                 (no dL/dx: five GEMM passes)
// The module has no pre-bias: it runs on the TopK parameter pack with b_pre = 0 (include/wsae.h), so the
// staging kernel, the encoder GEMM, the optimizer tail and the decoder renorm are the TopK path's own.
// Everything dense goes through ONE generic NT MFMA GEMM (C = A Bt^T, K contiguous in both operands,
// optional split-K into slabs); the element-wise kernels in between also write the transposed copies
// the next GEMM needs as its K-contiguous operand, through 64 x 64 LDS tiles.  No float atomics: every
// reduction is a two-level partial sum in fixed order.
#include "wsae_common.h"
#include "wsae_mfma.h"

namespace {

// ---- workspace -----------------------------------------------------------------------------------
struct ReluWs {
    void* hid;      // [maxB][H]   hidden in the contraction dtype (A operand of the decoder GEMM)
    void* hidT;     // [H][ldT]    its transpose (A operand of dW_d)
    void* dpreT;    // [H][ldT]    dpre transposed (A operand of dW_e)
    void* wd_nt;    // [D][H]      decoder weight in the reference's [D, H] layout (Bt operand of the decoder GEMM)
    float* part;    // [3][nblk]   per-tile partial sums: |hidden|, count(hidden > 0), squared residual
    float* colpart; // [ceil(maxB/64)][H]  per-tile column sums of dpre (db_e)
    float* scal;    // [4]         sparsity, mse
    int nblk;
    // fp8 forward (wsae_ctx_set_relu_fp8): e4m3 copies of the two GEMMs' operands and their per-row dequantisation scales
    uint8_t* xq;    // [maxB][D]
    uint8_t* weq;   // [H][D]
    uint8_t* hidq;  // [maxB][H]
    uint8_t* wdq;   // [D][H]
    float* sx;      // [maxB]
    float* swe;     // [H]
    float* sh;      // [maxB]
    float* swd;     // [D]
};

int reserve_ws(wsae_ctx* c) {
    if (c->relu_ws) return WSAE_OK;
    int prev = -1;
    (void)hipGetDevice(&prev);
    struct Restore {
        int d;
        ~Restore() { if (d >= 0) (void)hipSetDevice(d); }
    } restore{prev};
    if (hipSetDevice(c->device) != hipSuccess) {
        wsae_set_error("wsae_ctx_reserve_relu: hipSetDevice(%d) failed", c->device);
        return WSAE_ERR_HIP;
    }
    const size_t es = c->prec == WSAE_PREC_BF16 ? 2 : 4;
    const size_t ldT = ((size_t)c->maxB + 127) / 128 * 128;
    const size_t H = c->H, D = c->D, nb = (ldT + 63) / 64;
    const size_t nblk = nb * ((max(H, D) + 63) / 64);
    auto up = [](size_t v) { return (v + 255) / 256 * 256; };
    const size_t o0 = up(sizeof(ReluWs));
    const size_t o1 = o0 + up(ldT * H * es);
    const size_t o2 = o1 + up(H * ldT * es);
    const size_t o3 = o2 + up(H * ldT * es);
    const size_t o4 = o3 + up(D * H * es);
    const size_t o5 = o4 + up(3 * nblk * 4);
    const size_t o6 = o5 + up(nb * H * 4);
    const size_t q0 = o6 + 256;
    const size_t maxB = c->maxB;
    const size_t q1 = q0 + up(maxB * D), q2 = q1 + up(H * D), q3 = q2 + up(maxB * H), q4 = q3 + up(D * H);
    const size_t q5 = q4 + up(maxB * 4), q6 = q5 + up(H * 4), q7 = q6 + up(maxB * 4);
    const size_t total = q7 + up(D * 4);
    char* base = nullptr;
    if (hipMalloc((void**)&base, total) != hipSuccess) {
        wsae_set_error("wsae_relu: cannot allocate the %zu-byte ReLU workspace", total);
        return WSAE_ERR_HIP;
    }
    ReluWs h;
    h.hid = base + o0; h.hidT = base + o1; h.dpreT = base + o2; h.wd_nt = base + o3;
    h.part = (float*)(base + o4); h.colpart = (float*)(base + o5); h.scal = (float*)(base + o6);
    h.nblk = (int)nblk;
    h.xq = (uint8_t*)base + q0; h.weq = (uint8_t*)base + q1; h.hidq = (uint8_t*)base + q2; h.wdq = (uint8_t*)base + q3;
    h.sx = (float*)(base + q4); h.swe = (float*)(base + q5); h.sh = (float*)(base + q6); h.swd = (float*)(base + q7);
    if (hipMemset(base, 0, total) != hipSuccess || hipMemcpy(base, &h, sizeof(h), hipMemcpyHostToDevice) != hipSuccess) {
        (void)hipFree(base);
        wsae_set_error("wsae_relu: workspace initialisation failed");
        return WSAE_ERR_HIP;
    }
    c->relu_ws = base;
    return WSAE_OK;
}

ReluWs host_ws(wsae_ctx* c) {  // the descriptor is also kept at the head of the allocation; rebuild it on the host
    const size_t es = c->prec == WSAE_PREC_BF16 ? 2 : 4;
    const size_t ldT = ((size_t)c->maxB + 127) / 128 * 128;
    const size_t H = c->H, D = c->D, nb = (ldT + 63) / 64;
    const size_t nblk = nb * ((max(H, D) + 63) / 64);
    auto up = [](size_t v) { return (v + 255) / 256 * 256; };
    char* base = (char*)c->relu_ws;
    size_t o = up(sizeof(ReluWs));
    ReluWs h;
    h.hid = base + o; o += up(ldT * H * es);
    h.hidT = base + o; o += up(H * ldT * es);
    h.dpreT = base + o; o += up(H * ldT * es);
    h.wd_nt = base + o; o += up(D * H * es);
    h.part = (float*)(base + o); o += up(3 * nblk * 4);
    h.colpart = (float*)(base + o); o += up(nb * H * 4);
    h.scal = (float*)(base + o); o += 256;
    h.nblk = (int)nblk;
    const size_t maxB = c->maxB;
    h.xq = (uint8_t*)base + o; o += up(maxB * D);
    h.weq = (uint8_t*)base + o; o += up(H * D);
    h.hidq = (uint8_t*)base + o; o += up(maxB * H);
    h.wdq = (uint8_t*)base + o; o += up(D * H);
    h.sx = (float*)(base + o); o += up(maxB * 4);
    h.swe = (float*)(base + o); o += up(H * 4);
    h.sh = (float*)(base + o); o += up(maxB * 4);
    h.swd = (float*)(base + o);
    return h;
}

// ---- generic NT GEMM: C[M][N] (+ z * cz) = A[M][K] . Bt[N][K]^T over K range z (+ bias[n] on z = 0) -------
template <typename T>
__global__ void __launch_bounds__(256)
gemm_nt_kernel(const T* __restrict__ A, int64_t lda, const T* __restrict__ Bt, int64_t ldb,
               const float* __restrict__ bias, float* __restrict__ C, int64_t ldc, int M, int N, int K, int kper,
               int64_t cz) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* As = smem;
    char* Bs = smem + TILE_LDS_BYTES;
    constexpr int KT = Mfma<T>::KT;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int m0 = blockIdx.y * TILE_M, n0 = blockIdx.x * TILE_N;
    const int k_begin = blockIdx.z * kper, k_end = min(K, k_begin + kper);
    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    SlabRegs<T> ra, rb;
    if (k_begin < k_end) {
        slab_load<T>(ra, A, lda, m0, M, k_begin, k_end, tid);
        slab_load<T>(rb, Bt, ldb, n0, N, k_begin, k_end, tid);
    }
    for (int k = k_begin; k < k_end; k += KT) {
        slab_store<T>(ra, As, tid);
        slab_store<T>(rb, Bs, tid);
        __syncthreads();
        if (k + KT < k_end) {
            slab_load<T>(ra, A, lda, m0, M, k + KT, k_end, tid);
            slab_load<T>(rb, Bt, ldb, n0, N, k + KT, k_end, tid);
        }
        Mfma<T>::slab(As, Bs, wm * 64, wn * 64, lane, acc);
        __syncthreads();
    }
    float* Cz = C + (int64_t)blockIdx.z * cz;
    const int col = lane & 31, rq = lane >> 5;
#pragma unroll
    for (int mi = 0; mi < 2; ++mi)
#pragma unroll
        for (int ni = 0; ni < 2; ++ni) {
            const int n = n0 + wn * 64 + ni * 32 + col;
            if (n >= N) continue;
            const float bv = (bias && blockIdx.z == 0) ? bias[n] : 0.f;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int m = m0 + wm * 64 + mi * 32 + (r & 3) + 8 * (r >> 2) + 4 * rq;
                if (m < M) Cz[(int64_t)m * ldc + n] = acc[mi][ni][r] + bv;
            }
        }
}

// nz_out: the number of K ranges actually used (slabs to sum).  Large shapes go to the persistent LDS-DMA GEMM of the
// encoder (wsae_encode.hip); everything else to the simple 128 x 128 kernel above.
template <typename T>
void gemm_nt(wsae_ctx* c, hipStream_t st, const void* A, int64_t lda, const void* Bt, int64_t ldb, const float* bias, float* C,
             int64_t ldc, int M, int N, int K, int nz, int64_t cz, int* nz_out = nullptr) {
    constexpr int KT = Mfma<T>::KT;
    for (int ns = nz; ns >= 1; ns >>= 1) {  // largest power-of-two split (<= nz) whose ranges are whole pairs of K slabs
        if (K % ns || (K / ns) % (2 * KT)) continue;
        if (wsae_internal_gemm256d(c, A, lda, Bt, ldb, bias, C, ldc, M, N, K, ns, cz, st)) {
            if (nz_out) *nz_out = ns;
            return;
        }
        break;
    }
    const int kper = ceil_div(ceil_div(K, nz), KT) * KT;
    dim3 grid(ceil_div(N, TILE_N), ceil_div(M, TILE_M), ceil_div(K, kper));
    if (nz_out) *nz_out = (int)grid.z;
    gemm_nt_kernel<T><<<grid, 256, 2 * TILE_LDS_BYTES, st>>>((const T*)A, lda, (const T*)Bt, ldb, bias, C, ldc, M, N, K, kper, cz);
}

template <typename T>
__device__ __forceinline__ void store4(T* p, float a, float b, float c, float d) {
    if constexpr (sizeof(T) == 2) {
        bf16x4 o;
        o[0] = (bf16_t)a; o[1] = (bf16_t)b; o[2] = (bf16_t)c; o[3] = (bf16_t)d;
        *(bf16x4*)p = o;
    } else {
        *(float4*)p = make_float4(a, b, c, d);
    }
}

// ---- W_dT [H][D] (shadow or master) -> wd_nt [D][H] ------------------------------------------------
template <typename T>
__global__ void __launch_bounds__(256) transpose_w_kernel(const T* __restrict__ src, T* __restrict__ dst, int H, int D) {
    __shared__ float tile[64][65];
    const int h0 = blockIdx.x * 64, d0 = blockIdx.y * 64;
    for (int i = threadIdx.x; i < 4096; i += 256) {
        const int hl = i >> 6, dl = i & 63;
        tile[hl][dl] = (h0 + hl < H && d0 + dl < D) ? (float)src[(int64_t)(h0 + hl) * D + d0 + dl] : 0.f;
    }
    __syncthreads();
    for (int i = threadIdx.x; i < 4096; i += 256) {
        const int dl = i >> 6, hl = i & 63;
        if (h0 + hl < H && d0 + dl < D) dst[(int64_t)(d0 + dl) * H + h0 + hl] = (T)tile[hl][dl];
    }
}

// ---- hidden = relu(pre): f32 API copy, contraction-dtype copy, transposed copy, L1 / l0 partials ------
template <typename T>
__global__ void __launch_bounds__(256)
relu_act_kernel(const float* __restrict__ pre, float* __restrict__ hidden, T* __restrict__ hid, T* __restrict__ hidT,
                int B, int H, int ldT, float* __restrict__ part, int nblk, const float* __restrict__ l1w) {
    __shared__ float tile[64][65];
    __shared__ float red[8];
    const int h0 = blockIdx.x * 64, b0 = blockIdx.y * 64;
    const int q = threadIdx.x & 15, r16 = threadIdx.x >> 4;
    float s = 0.f, c = 0.f;
    // per-feature weights of the L1 term (wsae_ctx_set_relu_l1_weights; the crosscoder's decoder norms), else 1
    float4 w4 = make_float4(1.f, 1.f, 1.f, 1.f);
    if (l1w && h0 + 4 * q < H) w4 = *(const float4*)(l1w + h0 + 4 * q);
#pragma unroll
    for (int p = 0; p < 4; ++p) {
        const int bl = r16 + 16 * p, b = b0 + bl, h = h0 + 4 * q;
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (b < B && h < H) {
            v = *(const float4*)(pre + (int64_t)b * H + h);
            v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f);
            *(float4*)(hidden + (int64_t)b * H + h) = v;
            store4<T>(hid + (int64_t)b * H + h, v.x, v.y, v.z, v.w);
            s += (v.x * w4.x + v.y * w4.y) + (v.z * w4.z + v.w * w4.w);
            c += (float)((v.x > 0.f) + (v.y > 0.f) + (v.z > 0.f) + (v.w > 0.f));
        }
        tile[bl][4 * q] = v.x; tile[bl][4 * q + 1] = v.y; tile[bl][4 * q + 2] = v.z; tile[bl][4 * q + 3] = v.w;
    }
    __syncthreads();
#pragma unroll
    for (int p = 0; p < 4; ++p) {
        const int hl = r16 + 16 * p, h = h0 + hl, b = b0 + 4 * q;
        if (h < H && b < ldT)
            store4<T>(hidT + (int64_t)h * ldT + b, tile[4 * q][hl], tile[4 * q + 1][hl], tile[4 * q + 2][hl], tile[4 * q + 3][hl]);
    }
    const int blk = blockIdx.y * gridDim.x + blockIdx.x;
    const float ts = block_sum(s, red);
    const float tc = block_sum(c, red);
    if (threadIdx.x == 0) {
        part[blk] = ts;
        part[nblk + blk] = tc;
    }
}

// ---- residual: loss partials; in backward also g (contraction dtype, row-major into xb), gT, db_d partials ----
// recon2 / recon_out (forward of the row-major-GEMM flow): the reconstruction arrives as two split-K slabs, their sum is
// written to recon_out on the way
// what the forward's last block needs to turn the partial sums into the loss (relu_fwd_finish_kernel's arguments)
struct FwdFinish {
    const float* part;
    int nblk_act, nblk_res, nblk, B, cols, H;
    float weight;
    wsae_stats* stats;
    float* sparsity_out;
    float* scal;
    unsigned long long* ticket;  // null: no finish in this launch
};

__device__ __forceinline__ void relu_fwd_finish(const FwdFinish& f, float* red) {
    float s = 0.f, c = 0.f, l = 0.f;
    for (int i = threadIdx.x; i < f.nblk_act; i += 256) {  // (left by the launch before this one)
        s += f.part[i];
        c += f.part[f.nblk + i];
    }
    for (int i = threadIdx.x; i < f.nblk_res; i += 256)
        l += __hip_atomic_load(f.part + 2 * f.nblk + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const float ts = block_sum(s, red), tc = block_sum(c, red), tl = block_sum(l, red);
    if (threadIdx.x == 0) {
        const float sparsity = ts / ((float)f.B * (float)f.H);
        const float mse = tl / ((float)f.B * (float)f.cols);
        f.scal[0] = sparsity;
        f.scal[1] = mse;
        if (f.sparsity_out) *f.sparsity_out = sparsity;
        if (f.stats) {
            f.stats->loss = mse + f.weight * sparsity;
            f.stats->l0 = tc / (float)f.B;
            f.stats->reserved = __float_as_int(sparsity);  // TrainingMetrics: reconstruction = loss - weight * sparsity
        }
    }
}

template <typename T, int XDT, bool BWD>
__global__ void __launch_bounds__(256)
resid_kernel(const float* __restrict__ recon, const void* __restrict__ x, const int32_t* __restrict__ rows, int B, int D,
             int ldT, float scale, T* __restrict__ g_rm, T* __restrict__ gT, float* __restrict__ part_dbd,
             float* __restrict__ part_loss, const float* __restrict__ recon2 = nullptr, float* __restrict__ recon_out = nullptr,
             FwdFinish fin = FwdFinish()) {
    __shared__ float tile[64][65];
    __shared__ float red[8];
    __shared__ int last_s;
    const int d0 = blockIdx.x * 64, b0 = blockIdx.y * 64;
    const int q = threadIdx.x & 15, r16 = threadIdx.x >> 4;
    float loss = 0.f;
#pragma unroll
    for (int p = 0; p < 4; ++p) {
        const int bl = r16 + 16 * p, b = b0 + bl, d = d0 + 4 * q;
        float r[4] = {0.f, 0.f, 0.f, 0.f};
        if (b < B && d < D) {
            const int64_t src = rows ? (int64_t)rows[b] : (int64_t)b;
            float4 rc = *(const float4*)(recon + (int64_t)b * D + d);
            if (recon2) {
                const float4 r2 = *(const float4*)(recon2 + (int64_t)b * D + d);
                rc.x += r2.x; rc.y += r2.y; rc.z += r2.z; rc.w += r2.w;
            }
            if (recon_out) *(float4*)(recon_out + (int64_t)b * D + d) = rc;
            r[0] = rc.x - load_act<XDT>(x, src * D + d);
            r[1] = rc.y - load_act<XDT>(x, src * D + d + 1);
            r[2] = rc.z - load_act<XDT>(x, src * D + d + 2);
            r[3] = rc.w - load_act<XDT>(x, src * D + d + 3);
            loss += (r[0] * r[0] + r[1] * r[1]) + (r[2] * r[2] + r[3] * r[3]);
            if (BWD) store4<T>(g_rm + (int64_t)b * D + d, r[0] * scale, r[1] * scale, r[2] * scale, r[3] * scale);
        }
        if (BWD) {
#pragma unroll
            for (int i = 0; i < 4; ++i) tile[bl][4 * q + i] = r[i] * scale;
        }
    }
    if (BWD) {
        __syncthreads();
#pragma unroll
        for (int p = 0; p < 4; ++p) {
            const int dl = r16 + 16 * p, d = d0 + dl, b = b0 + 4 * q;
            if (gT && d < D && b < ldT)
                store4<T>(gT + (int64_t)d * ldT + b, tile[4 * q][dl], tile[4 * q + 1][dl], tile[4 * q + 2][dl], tile[4 * q + 3][dl]);
        }
        if (threadIdx.x < 64 && d0 + (int)threadIdx.x < D) {  // column sums of this tile's 64 rows, fixed order
            float a = 0.f;
            for (int bl = 0; bl < 64; ++bl) a += tile[bl][threadIdx.x];
            part_dbd[(int64_t)blockIdx.y * D + d0 + threadIdx.x] = a;
        }
    }
    const float t = block_sum(loss, red);
    if (!fin.ticket) {
        if (threadIdx.x == 0) part_loss[blockIdx.y * gridDim.x + blockIdx.x] = t;
        return;
    }
    // the forward's loss in this launch: the last block to arrive sums the partials (same order as relu_fwd_finish_kernel)
    if (threadIdx.x == 0) {
        __hip_atomic_store(part_loss + blockIdx.y * gridDim.x + blockIdx.x, t, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        unsigned unused;
        last_s = grid_ticket(fin.ticket, 0u, &unused);
    }
    __syncthreads();
    if (last_s) relu_fwd_finish(fin, red);
}

// ---- dpre = (dh + l1) * 1[hidden > 0] -> dpreT, db_e partials ----------------------------------------
template <typename T>
__global__ void __launch_bounds__(256)
dpre_kernel(const float* __restrict__ dh, const float* __restrict__ hidden, int B, int H, int ldT, float l1,
            T* __restrict__ dpreT, float* __restrict__ colpart, const float* __restrict__ l1w) {
    __shared__ float tile[64][65];
    const int h0 = blockIdx.x * 64, b0 = blockIdx.y * 64;
    const int q = threadIdx.x & 15, r16 = threadIdx.x >> 4;
    float4 l4 = make_float4(l1, l1, l1, l1);
    if (l1w && h0 + 4 * q < H) {
        const float4 w4 = *(const float4*)(l1w + h0 + 4 * q);
        l4 = make_float4(l1 * w4.x, l1 * w4.y, l1 * w4.z, l1 * w4.w);
    }
#pragma unroll
    for (int p = 0; p < 4; ++p) {
        const int bl = r16 + 16 * p, b = b0 + bl, h = h0 + 4 * q;
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (b < B && h < H) {
            const float4 d4 = *(const float4*)(dh + (int64_t)b * H + h);
            const float4 h4 = *(const float4*)(hidden + (int64_t)b * H + h);
            v.x = h4.x > 0.f ? d4.x + l4.x : 0.f;
            v.y = h4.y > 0.f ? d4.y + l4.y : 0.f;
            v.z = h4.z > 0.f ? d4.z + l4.z : 0.f;
            v.w = h4.w > 0.f ? d4.w + l4.w : 0.f;
        }
        tile[bl][4 * q] = v.x; tile[bl][4 * q + 1] = v.y; tile[bl][4 * q + 2] = v.z; tile[bl][4 * q + 3] = v.w;
    }
    __syncthreads();
#pragma unroll
    for (int p = 0; p < 4; ++p) {
        const int hl = r16 + 16 * p, h = h0 + hl, b = b0 + 4 * q;
        if (h < H && b < ldT)
            store4<T>(dpreT + (int64_t)h * ldT + b, tile[4 * q][hl], tile[4 * q + 1][hl], tile[4 * q + 2][hl], tile[4 * q + 3][hl]);
    }
    if (threadIdx.x < 64 && h0 + (int)threadIdx.x < H) {
        float a = 0.f;
        for (int bl = 0; bl < 64; ++bl) a += tile[bl][threadIdx.x];
        colpart[(int64_t)blockIdx.y * H + h0 + threadIdx.x] = a;
    }
}

// ---- small finishing kernels (fixed summation order) ---------------------------------------------------
__global__ void __launch_bounds__(256)
relu_fwd_finish_kernel(const float* __restrict__ part, int nblk_act, int nblk_res, int nblk, int B, int D, int H,
                       float weight, wsae_stats* __restrict__ stats, float* __restrict__ sparsity_out,
                       float* __restrict__ scal) {
    __shared__ float red[8];
    float s = 0.f, c = 0.f, l = 0.f;
    for (int i = threadIdx.x; i < nblk_act; i += 256) {
        s += part[i];
        c += part[nblk + i];
    }
    for (int i = threadIdx.x; i < nblk_res; i += 256) l += part[2 * nblk + i];
    const float ts = block_sum(s, red), tc = block_sum(c, red), tl = block_sum(l, red);
    if (threadIdx.x == 0) {
        const float sparsity = ts / ((float)B * (float)H);
        const float mse = tl / ((float)B * (float)D);  // (D here = the ctx's loss_cols)
        scal[0] = sparsity;
        scal[1] = mse;
        if (sparsity_out) *sparsity_out = sparsity;
        if (stats) {
            stats->loss = mse + weight * sparsity;
            stats->l0 = tc / (float)B;
            stats->reserved = __float_as_int(sparsity);  // TrainingMetrics: reconstruction = loss - weight * sparsity
        }
    }
}

__global__ void __launch_bounds__(256) colsum_kernel(const float* __restrict__ part, int nrows, int N, float* __restrict__ out) {
    const int n = blockIdx.x * 256 + threadIdx.x;
    if (n >= N) return;
    float a = 0.f;
    for (int i0 = 0; i0 < nrows; i0 += 8) {
        float v[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) v[i] = (i0 + i < nrows) ? part[(int64_t)min(i0 + i, nrows - 1) * N + n] : 0.f;
#pragma unroll
        for (int i = 0; i < 8; ++i) a += v[i];
    }
    out[n] = a;
}

// Both bias gradients in ONE launch with the rows of a column spread over the block (the form above walks a column's rows in
// one thread: 17 us per call at B = 16384 for a few hundred KB).  Block = 64 columns x 4 row parts, every thread sums its
// rows 8 loads at a time, the four part sums meet in LDS in fixed order.  Blocks [0, nb1) serve (part1, N1), the rest (part2, N2).
// sq_out / ticket (nullable together): the last block to arrive adds the squares of both results as one more global-norm
// partial (the results are then handed over as agent-scope stores; fixed summation order).
__global__ void __launch_bounds__(256)
colsum2_kernel(const float* __restrict__ part1, int nrows1, int N1, float* __restrict__ out1, int nb1,
               const float* __restrict__ part2, int nrows2, int N2, float* __restrict__ out2, float* __restrict__ zero_n2,
               float* __restrict__ sq_out = nullptr, unsigned long long* __restrict__ ticket = nullptr) {
    __shared__ float s[4][64];
    __shared__ float red[8];
    __shared__ int last_s;
    const bool second = (int)blockIdx.x >= nb1;
    // (no pre-bias in this module: its gradient slot, N2 values, stays exactly 0 - written here instead of by a fill launch)
    if (second && zero_n2 && threadIdx.x < 64 && ((int)blockIdx.x - nb1) * 64 + (int)threadIdx.x < N2)
        zero_n2[((int)blockIdx.x - nb1) * 64 + threadIdx.x] = 0.f;
    const float* part = second ? part2 : part1;
    const int nrows = second ? nrows2 : nrows1, N = second ? N2 : N1;
    float* out = second ? out2 : out1;
    const int c = threadIdx.x & 63, rp = threadIdx.x >> 6;
    const int n = ((int)blockIdx.x - (second ? nb1 : 0)) * 64 + c;
    const int per = (nrows + 3) / 4, r0 = rp * per, r1 = min(nrows, r0 + per);
    float a = 0.f;
    if (n < N) {
        for (int i0 = r0; i0 < r1; i0 += 8) {
            float v[8];
#pragma unroll
            for (int i = 0; i < 8; ++i) v[i] = part[(int64_t)min(i0 + i, r1 - 1) * N + n];
#pragma unroll
            for (int i = 0; i < 8; ++i) a += (i0 + i < r1) ? v[i] : 0.f;
        }
    }
    s[rp][c] = a;
    __syncthreads();
    if (rp == 0 && n < N) {
        const float t = (s[0][c] + s[1][c]) + (s[2][c] + s[3][c]);
        if (ticket) __hip_atomic_store(out + n, t, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        else out[n] = t;
    }
    if (!ticket) return;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0) {
        unsigned unused;
        last_s = grid_ticket(ticket, 0u, &unused);
    }
    __syncthreads();
    if (!last_s) return;
    float sq = 0.f;
    for (int i = threadIdx.x; i < N1; i += 256) {
        const float v = __hip_atomic_load(out1 + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        sq = fmaf(v, v, sq);
    }
    for (int i = threadIdx.x; i < N2; i += 256) {
        const float v = __hip_atomic_load(out2 + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        sq = fmaf(v, v, sq);
    }
    const float t = block_sum(sq, red);
    if (threadIdx.x == 0) *sq_out = t;
}

__global__ void __launch_bounds__(256)
slab_sum_kernel(const float* __restrict__ slabs, int64_t slab_stride, int nz, int64_t n4, float* __restrict__ grads,
                float* __restrict__ g_bpre, int D) {
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (int64_t)gridDim.x * 256) {
        float4 a = ((const float4*)slabs)[i];
        for (int z = 1; z < nz; ++z) {
            const float4 b = ((const float4*)(slabs + z * slab_stride))[i];
            a.x += b.x; a.y += b.y; a.z += b.z; a.w += b.w;
        }
        ((float4*)grads)[i] = a;
    }
    if (blockIdx.x == 0)
        for (int d = threadIdx.x; d < D; d += 256) g_bpre[d] = 0.f;  // no pre-bias in this module: it stays exactly 0
}

// The same sum with every slab's load of an element in flight at once (the loop above issues them one dependent trip at a
// time), one float4 per thread and pass, and the block's sum of squares left as a global-norm partial: the optimizer then
// needs no norm pass over the matrices (wsae_adamw_step norm_from_wgrad = 1; the bias gradients' squares are added by
// bias_sq_kernel below).  nz <= 8.
__global__ void __launch_bounds__(256)
slab_sum8_kernel(const float* __restrict__ slabs, int64_t slab_stride, int nz, int64_t n4, float* __restrict__ grads,
                 float* __restrict__ g_bpre, int D, float* __restrict__ part_sq) {
    __shared__ float red[8];
    float sq = 0.f;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (int64_t)gridDim.x * 256) {
        float4 v[8];
#pragma unroll
        for (int z = 0; z < 8; ++z) v[z] = ((const float4*)(slabs + (int64_t)min(z, nz - 1) * slab_stride))[i];
        float4 a = v[0];
#pragma unroll
        for (int z = 1; z < 8; ++z) {
            const float w = z < nz ? 1.f : 0.f;
            a.x = fmaf(w, v[z].x, a.x); a.y = fmaf(w, v[z].y, a.y); a.z = fmaf(w, v[z].z, a.z); a.w = fmaf(w, v[z].w, a.w);
        }
        ((float4*)grads)[i] = a;
        sq += (a.x * a.x + a.y * a.y) + (a.z * a.z + a.w * a.w);
    }
    if (blockIdx.x == 0)
        for (int d = threadIdx.x; d < D; d += 256) g_bpre[d] = 0.f;  // no pre-bias in this module: it stays exactly 0
    const float t = block_sum(sq, red);
    if (threadIdx.x == 0) part_sq[blockIdx.x] = t;
}

// squares of the two bias gradients (H + D values) as one more norm partial
__global__ void __launch_bounds__(1024) bias_sq_kernel(const float* __restrict__ a, int na, const float* __restrict__ b, int nb,
                                                       float* __restrict__ out) {
    __shared__ float red[16];
    float sq = 0.f;
    for (int i = threadIdx.x; i < na; i += 1024) sq = fmaf(a[i], a[i], sq);
    for (int i = threadIdx.x; i < nb; i += 1024) sq = fmaf(b[i], b[i], sq);
    const float t = block_sum(sq, red);
    if (threadIdx.x == 0) *out = t;
}

int check_dims(wsae_ctx* ctx, int B, const char* who) {
    WSAE_REQUIRE(B >= 1 && B <= ctx->maxB, "%s: batch %d outside [1, %d]", who, B, ctx->maxB);
    WSAE_REQUIRE(ctx->D % 8 == 0 && ctx->H % 8 == 0, "%s: input_dim %d and hidden_dim %d must be multiples of 8", who, ctx->D,
                 ctx->H);
    WSAE_REQUIRE(ceil_div(B, 64) <= WSAE_MAX_PARTIALS, "%s: batch %d too large for the bias-gradient partials", who, B);
    return WSAE_OK;
}

// ---- fp8 e4m3 quantisation of a row-major matrix, one scale per row --------------------------------------
// q[r][c] = e4m3(v[r][c] * (448 / amax_r)) (v_cvt_pk_fp8_f32: round to nearest even), scale[r] = amax_r / 448 (1 for an
// all-zero row).  One wave per row, two passes (the second one from L2); 8 values -> one 8-byte store per lane.
template <int SRC>
__global__ void __launch_bounds__(256) quant_rows_kernel(const void* __restrict__ src, const int32_t* __restrict__ rows, int R, int C,
                                                         uint8_t* __restrict__ q, float* __restrict__ scale) {
    const int lane = threadIdx.x & 63;
    const int r = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (r >= R) return;
    const int64_t sr = rows ? (int64_t)rows[r] : (int64_t)r;
    auto load8 = [&](int c, float (&v)[8]) {
        if (SRC == WSAE_DT_F32) {
            const float4 a = *(const float4*)((const float*)src + sr * C + c), b = *(const float4*)((const float*)src + sr * C + c + 4);
            v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w; v[4] = b.x; v[5] = b.y; v[6] = b.z; v[7] = b.w;
        } else {
            const bf16x8 a = *(const bf16x8*)((const bf16_t*)src + sr * C + c);
#pragma unroll
            for (int i = 0; i < 8; ++i) v[i] = (float)a[i];
        }
    };
    float amax = 0.f;
    for (int c = lane * 8; c < C; c += 512) {
        float v[8];
        load8(c, v);
#pragma unroll
        for (int i = 0; i < 8; ++i) amax = fmaxf(amax, fabsf(v[i]));
    }
    amax = wave_max(amax);
    const float inv = amax > 0.f ? 448.f / amax : 1.f;
    if (lane == 0) scale[r] = amax > 0.f ? amax / 448.f : 1.f;
    for (int c = lane * 8; c < C; c += 512) {
        float v[8];
        load8(c, v);
        int lo = 0, hi = 0;
        lo = __builtin_amdgcn_cvt_pk_fp8_f32(v[0] * inv, v[1] * inv, lo, false);
        lo = __builtin_amdgcn_cvt_pk_fp8_f32(v[2] * inv, v[3] * inv, lo, true);
        hi = __builtin_amdgcn_cvt_pk_fp8_f32(v[4] * inv, v[5] * inv, hi, false);
        hi = __builtin_amdgcn_cvt_pk_fp8_f32(v[6] * inv, v[7] * inv, hi, true);
        *(int2*)(q + (int64_t)r * C + c) = make_int2(lo, hi);
    }
}

// ---- per-TENSOR e4m3 quantisation of the decoder weight (its columns are kept at unit norm: one dynamic range) -------------
// amax partials (one per block, no atomics; the consumers below reduce them again - at most 1024 values - in their prologue)
__global__ void __launch_bounds__(256) tensor_amax_kernel(const bf16_t* __restrict__ src, int64_t n8, float* __restrict__ pmax) {
    __shared__ float red[4];
    float m = 0.f;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n8; i += (int64_t)gridDim.x * 256) {
        const bf16x8 v = ((const bf16x8*)src)[i];
#pragma unroll
        for (int e = 0; e < 8; ++e) m = fmaxf(m, fabsf((float)v[e]));
    }
    m = wave_max(m);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x == 0) pmax[blockIdx.x] = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
}

__device__ __forceinline__ float amax_of_partials(const float* __restrict__ pmax, int nparts, float* red) {
    float m = 0.f;
    for (int i = threadIdx.x; i < nparts; i += 256) m = fmaxf(m, pmax[i]);
    m = wave_max(m);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = m;
    __syncthreads();
    return fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
}

// W_dT bf16 [H][D] -> e4m3 W_d [D][H] (K = h contiguous: the NT operand of the decoder GEMM) with the tensor's scale, one pass
__global__ void __launch_bounds__(256)
transpose_quant_kernel(const bf16_t* __restrict__ src, const float* __restrict__ pmax, int nparts, uint8_t* __restrict__ q,
                       float* __restrict__ scale, int H, int D) {
    __shared__ float tile[64][65];
    __shared__ float red[4];
    const float amax = amax_of_partials(pmax, nparts, red);
    const float inv = amax > 0.f ? 448.f / amax : 1.f;
    const int h0 = blockIdx.x * 64, d0 = blockIdx.y * 64;
    if (blockIdx.x == 0 && threadIdx.x < 64 && d0 + (int)threadIdx.x < D) scale[d0 + threadIdx.x] = amax > 0.f ? amax / 448.f : 1.f;
    for (int i = threadIdx.x; i < 4096; i += 256) {
        const int hl = i >> 6, dl = i & 63;
        tile[hl][dl] = (h0 + hl < H && d0 + dl < D) ? (float)src[(int64_t)(h0 + hl) * D + d0 + dl] : 0.f;
    }
    __syncthreads();
    for (int i = threadIdx.x; i < 1024; i += 256) {  // 64 output rows (d) x 16 groups of 4 consecutive h
        const int dl = i >> 4, g4 = (i & 15) * 4;
        if (d0 + dl >= D || h0 + g4 >= H) continue;  // (H is a multiple of 4)
        int w = 0;
        w = __builtin_amdgcn_cvt_pk_fp8_f32(tile[g4][dl] * inv, tile[g4 + 1][dl] * inv, w, false);
        w = __builtin_amdgcn_cvt_pk_fp8_f32(tile[g4 + 2][dl] * inv, tile[g4 + 3][dl] * inv, w, true);
        *(int*)(q + (int64_t)(d0 + dl) * H + h0 + g4) = w;
    }
}

// the general path's W_d [D][H] bf16 (already transposed) with the tensor's scale, one pass
__global__ void __launch_bounds__(256)
quant_fixed_kernel(const bf16_t* __restrict__ src, const float* __restrict__ pmax, int nparts, int R, int C, uint8_t* __restrict__ q,
                   float* __restrict__ scale) {
    __shared__ float red[4];
    const float amax = amax_of_partials(pmax, nparts, red);
    const float inv = amax > 0.f ? 448.f / amax : 1.f;
    const int64_t n8 = (int64_t)R * C / 8;
    if (blockIdx.x == 0)
        for (int r = threadIdx.x; r < R; r += 256) scale[r] = amax > 0.f ? amax / 448.f : 1.f;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n8; i += (int64_t)gridDim.x * 256) {
        const bf16x8 a = ((const bf16x8*)src)[i];
        int lo = 0, hi = 0;
        lo = __builtin_amdgcn_cvt_pk_fp8_f32((float)a[0] * inv, (float)a[1] * inv, lo, false);
        lo = __builtin_amdgcn_cvt_pk_fp8_f32((float)a[2] * inv, (float)a[3] * inv, lo, true);
        hi = __builtin_amdgcn_cvt_pk_fp8_f32((float)a[4] * inv, (float)a[5] * inv, hi, false);
        hi = __builtin_amdgcn_cvt_pk_fp8_f32((float)a[6] * inv, (float)a[7] * inv, hi, true);
        ((int2*)q)[i] = make_int2(lo, hi);
    }
}

static void quant_rows(hipStream_t st, const void* src, int dtype, const int32_t* rows, int R, int C, uint8_t* q, float* scale) {
    if (dtype == WSAE_DT_F32) quant_rows_kernel<WSAE_DT_F32><<<ceil_div(R, 4), 256, 0, st>>>(src, rows, R, C, q, scale);
    else quant_rows_kernel<WSAE_DT_BF16><<<ceil_div(R, 4), 256, 0, st>>>(src, rows, R, C, q, scale);
}

// ---- the row-major-GEMM flow (bf16, wsae_gemm256x.hip): no transposed copies, relu / dpre fused into the GEMM epilogues ----
// Eligible shapes: whole 128-row groups of batch rows, D a multiple of 128, H a multiple of 256.  Then per step the dense
// [B, H] traffic is: bf16 hidden out (1x), in (3x: decoder GEMM, dpre epilogue, dW_d contraction), bf16 dpre out + in - against
// fp32 pre, fp32 hidden, fp32 dh and two transposed bf16 copies out and back in on the general path below (1.2 GB at
// 384 -> 3072 / B = 16384).
bool x_flow_ok(const wsae_ctx* c, int B) {
    const bool base = c->prec == WSAE_PREC_BF16 && B >= 256 && B % 128 == 0 && c->D % 128 == 0 && c->H % 256 == 0 && 2 * c->D <= c->H;
    // precision = "fp8": the two forward GEMMs walk 128-element slabs in pairs (D % 256, (H / 2) % 256); other fp8 shapes keep the
    // general path below
    return base && (!c->relu_fp8 || (c->D % 256 == 0 && c->H % 512 == 0));
}

// bf16 hidden [R][C] -> e4m3 with one scale per row, the row's maximum taken from the [C / 64][ld] partial maxima the encoder
// GEMM's epilogue left (one pass over hidden instead of two).  One block per row.
__global__ void __launch_bounds__(256)
quant_hidden_kernel(const bf16_t* __restrict__ src, const float* __restrict__ pmax, int64_t ld, int R, int C, uint8_t* __restrict__ q,
                    float* __restrict__ scale) {
    __shared__ float red[4];
    const int r = blockIdx.x;
    float amax = 0.f;
    for (int w = threadIdx.x; w < C / 64; w += 256) amax = fmaxf(amax, pmax[(int64_t)w * ld + r]);
    amax = wave_max(amax);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = amax;
    __syncthreads();
    amax = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
    const float inv = amax > 0.f ? 448.f / amax : 1.f;
    if (threadIdx.x == 0) scale[r] = amax > 0.f ? amax / 448.f : 1.f;
    for (int c = threadIdx.x * 8; c < C; c += 2048) {
        const bf16x8 a = *(const bf16x8*)(src + (int64_t)r * C + c);
        int lo = 0, hi = 0;
        lo = __builtin_amdgcn_cvt_pk_fp8_f32((float)a[0] * inv, (float)a[1] * inv, lo, false);
        lo = __builtin_amdgcn_cvt_pk_fp8_f32((float)a[2] * inv, (float)a[3] * inv, lo, true);
        hi = __builtin_amdgcn_cvt_pk_fp8_f32((float)a[4] * inv, (float)a[5] * inv, hi, false);
        hi = __builtin_amdgcn_cvt_pk_fp8_f32((float)a[6] * inv, (float)a[7] * inv, hi, true);
        *(int2*)(q + (int64_t)r * C + c) = make_int2(lo, hi);
    }
}

int x_split(int B) {  // split-K over the batch: ranges of whole 128-row groups, at least 512 rows each
    int nz = WSAE_WGRAD_MAX_SPLIT;
    while (nz > 1 && (B % nz || (B / nz) % 128 || B / nz < 512)) nz >>= 1;
    return nz;
}

int forward_x(wsae_ctx* ctx, const float* params, const void* x, int x_dtype, const int32_t* rows, int B, float weight,
              float* hidden, float* recon, wsae_stats* stats, float* sparsity_out, hipStream_t st) {
    const int D = ctx->D, H = ctx->H;
    const int ldT = (B + 127) / 128 * 128;
    const ReluWs ws = host_ws(ctx);
    int rc = wsae_internal_stage_rows(ctx, params, x, x_dtype, rows, B, st);  // xb = bf16(x) [B][D] (the contractions read it row-major)
    if (rc) return rc;
    // precision = "fp8" (wsae_ctx_set_relu_fp8; BASELINE.json configs[4]): the two forward GEMMs on OCP e4m3 operands - x and
    // bf16(hidden) quantised per batch row, the bf16 shadows of W_e per feature row and of W_d per output row - through the
    // block-scaled MFMA with unit block scales (twice the bf16 rate); the row x column dequantisation scales are applied in the
    // epilogues.  hidden, loss and the whole backward stay on the bf16 values.  The oracle's "fp8" mode mirrors exactly this.
    const bool fp8 = ctx->relu_fp8 != 0;
    GxEpi e1 = {};
    e1.bias = params + ctx->off[2];
    e1.c = hidden; e1.ldc = H;
    e1.out16 = (bf16_t*)ws.hid; e1.ld16 = H;
    e1.colw = ctx->relu_l1w;
    e1.part = ws.part; e1.nslots = ws.nblk;
    e1.bits = (uint64_t*)ws.dpreT; e1.ldbits = ldT;  // (the transposed-dpre buffer of the general path is free in this flow)
    float* rowmax = (float*)((uint64_t*)ws.dpreT + (int64_t)ceil_div(H, 64) * ldT);  // [H / 64][ldT], behind the activity words
    if (fp8) {
        quant_rows(st, x, x_dtype, rows, B, D, ws.xq, ws.sx);
        quant_rows(st, ctx->We_bf16, WSAE_DT_BF16, nullptr, H, D, ws.weq, ws.swe);
        e1.rscale = ws.sx; e1.cscale = ws.swe; e1.rowmax = rowmax;
        WSAE_REQUIRE(wsae_internal_gemm256x(ctx, 0, 0, GX_EPI_RELU, ws.xq, D, ws.weq, D, B, H, D, 1, e1, st, 1),
                     "wsae_relu_forward: fp8 encoder GEMM rejected B %d, D %d, H %d", B, D, H);
    } else {
        WSAE_REQUIRE(wsae_internal_gemm256x(ctx, 0, 0, GX_EPI_RELU, ctx->xb, D, ctx->We_bf16, D, B, H, D, 1, e1, st),
                     "wsae_relu_forward: encoder GEMM rejected B %d, D %d, H %d", B, D, H);
    }
    GxEpi e2 = {};
    e2.bias = params + ctx->off[3];
    e2.c = ctx->pre; e2.ldc = D; e2.cz = (int64_t)B * D;  // two split-K slabs in the (otherwise unused) pre-activation scratch
    if (fp8) {
        quant_hidden_kernel<<<B, 256, 0, st>>>((const bf16_t*)ws.hid, rowmax, ldT, B, H, ws.hidq, ws.sh);
        // the decoder weight with ONE scale (unit-norm columns: one dynamic range): amax partials, then transposition and
        // quantisation in a single pass (round-3 first cut: transpose + two-pass per-row quantisation, 184 us at 1280 -> 40960)
        const int npm = 1024;
        tensor_amax_kernel<<<npm, 256, 0, st>>>(ctx->WdT_bf16, (int64_t)H * D / 8, ctx->part_sq);  // (the norm-partial slots are free here)
        transpose_quant_kernel<<<dim3(ceil_div(H, 64), ceil_div(D, 64)), 256, 0, st>>>(ctx->WdT_bf16, ctx->part_sq, npm, ws.wdq, ws.swd, H, D);
        e2.rscale = ws.sh; e2.cscale = ws.swd;
        WSAE_REQUIRE(wsae_internal_gemm256x(ctx, 0, 0, GX_EPI_PLAIN, ws.hidq, H, ws.wdq, H, B, D, H, 2, e2, st, 1),
                     "wsae_relu_forward: fp8 decoder GEMM rejected B %d, D %d, H %d", B, D, H);
    } else {
        WSAE_REQUIRE(wsae_internal_gemm256x(ctx, 0, 1, GX_EPI_PLAIN, ws.hid, H, ctx->WdT_bf16, D, B, D, H, 2, e2, st),
                     "wsae_relu_forward: decoder GEMM rejected B %d, D %d, H %d", B, D, H);
    }
    // residual pass: recon = slab 0 + slab 1 (written out only when the caller wants it), loss partials - and already
    // g = 2 (recon - x) / (B cols) as bf16 [B][D] + the db_d partials, which is everything the backward would re-read the
    // 4 B D bytes of recon for: wsae_relu_backward of this batch finds them (relu_g_B) and launches no residual pass
    dim3 gr(ceil_div(D, 64), ceil_div(ldT, 64));
    const float scale = 2.0f / ((float)B * (float)ctx->loss_cols);
    // ... and the loss: the launch's last block sums the partials (no finishing launch)
    const int nt1 = ceil_div(H, 256) * ceil_div(B, 256);
    const FwdFinish fin = {ws.part, nt1, (int)(gr.x * gr.y), ws.nblk, B, ctx->loss_cols, H, weight, stats, sparsity_out, ws.scal,
                           (unsigned long long*)(ctx->counters + 16 + 2 * TICKET_WORDS)};
    if (x_dtype == WSAE_DT_F32)
        resid_kernel<bf16_t, WSAE_DT_F32, true><<<gr, 256, 0, st>>>(ctx->pre, x, rows, B, D, ldT, scale, ctx->gb, nullptr, ctx->part_dbd,
                                                                   ws.part + 2 * ws.nblk, ctx->pre + (int64_t)B * D, recon, fin);
    else
        resid_kernel<bf16_t, WSAE_DT_BF16, true><<<gr, 256, 0, st>>>(ctx->pre, x, rows, B, D, ldT, scale, ctx->gb, nullptr, ctx->part_dbd,
                                                                    ws.part + 2 * ws.nblk, ctx->pre + (int64_t)B * D, recon, fin);
    ctx->relu_g_B = B;
    WSAE_LAUNCH_CHECK();
    ctx->relu_x_B = B;
    return WSAE_OK;
}

int backward_x(wsae_ctx* ctx, const float* params, const void* x, int x_dtype, const int32_t* rows, int B, float weight,
               const float* recon, float* grads, hipStream_t st) {
    const int D = ctx->D, H = ctx->H;
    const int ldT = (B + 127) / 128 * 128;
    const ReluWs ws = host_ws(ctx);
    const float scale = 2.0f / ((float)B * (float)ctx->loss_cols);
    dim3 gr(ceil_div(D, 64), ceil_div(ldT, 64));
    // g = 2 (recon - x) / (B cols) as bf16 [B][D] (row-major: what both consumers read), db_d partials: left by the
    // forward's residual pass; recomputed from recon only when something else has used the buffers since
    if (ctx->relu_g_B != B) {
        WSAE_REQUIRE(recon, "wsae_relu_backward: the forward's residual gradient is gone and no recon was passed to rebuild it");
        if (x_dtype == WSAE_DT_F32)
            resid_kernel<bf16_t, WSAE_DT_F32, true><<<gr, 256, 0, st>>>(recon, x, rows, B, D, ldT, scale, ctx->gb, nullptr, ctx->part_dbd,
                                                                       ws.part + 2 * ws.nblk);
        else
            resid_kernel<bf16_t, WSAE_DT_BF16, true><<<gr, 256, 0, st>>>(recon, x, rows, B, D, ldT, scale, ctx->gb, nullptr, ctx->part_dbd,
                                                                        ws.part + 2 * ws.nblk);
    }
    bf16_t* dpre = (bf16_t*)ws.hidT;  // [B][H] (the transposed-hidden buffer of the general path is free in this flow)
    GxEpi e3 = {};
    e3.out16 = dpre; e3.ld16 = H;
    e3.bits = (uint64_t*)ws.dpreT; e3.ldbits = ldT;  // the activity bits the forward's epilogue left: 1/16 of re-reading hidden
    e3.colw = ctx->relu_l1w;
    e3.l1 = weight / ((float)B * (float)H);
    e3.colpart = ws.colpart;
    WSAE_REQUIRE(wsae_internal_gemm256x(ctx, 0, 0, GX_EPI_DPRE, ctx->gb, D, ctx->WdT_bf16, D, B, H, D, 1, e3, st),
                 "wsae_relu_backward: dh GEMM rejected B %d, D %d, H %d", B, D, H);
    // the two contractions: wgrad2's 192 x 384 geometry with a dense left operand in ONE launch + the slab reduction of the
    // TopK path (matrix rows + norm partials); the 256 x 256 GEMM twice for shapes that launch does not serve
    int nsq = wsae_internal_relu_wgrad(ctx, ws.hid, dpre, ctx->xb, ctx->gb, B, grads, st);
    if (nsq == 0) {
        const int nz = x_split(B);
        const int64_t hd = (int64_t)H * D, slab_stride = 2 * hd;
        GxEpi e4 = {};
        e4.c = ctx->wg_slabs; e4.ldc = D; e4.cz = slab_stride;
        WSAE_REQUIRE(wsae_internal_gemm256x(ctx, 1, 1, GX_EPI_PLAIN, dpre, H, ctx->xb, D, H, D, B, nz, e4, st),
                     "wsae_relu_backward: dW_e contraction rejected B %d (split %d)", B, nz);
        GxEpi e5 = e4;
        e5.c = ctx->wg_slabs + hd;
        WSAE_REQUIRE(wsae_internal_gemm256x(ctx, 1, 1, GX_EPI_PLAIN, ws.hid, H, ctx->gb, D, H, D, B, nz, e5, st),
                     "wsae_relu_backward: dW_d contraction rejected B %d (split %d)", B, nz);
        nsq = 1022;  // (blocks = norm partial slots; + 1 for the biases, WSAE_MAX_PARTIALS = 1024)
        slab_sum8_kernel<<<nsq, 256, 0, st>>>(ctx->wg_slabs, slab_stride, nz, slab_stride / 4, grads, grads + ctx->off[4], D, ctx->part_sq);
    }
    const int nb1 = ceil_div(H, 64);
    // (the squares of the two bias gradients - one more norm partial - by the launch's last block: no bias_sq launch)
    colsum2_kernel<<<nb1 + ceil_div(D, 64), 256, 0, st>>>(ws.colpart, B / 128, H, grads + ctx->off[2], nb1, ctx->part_dbd,
                                                          ceil_div(B, 64), D, grads + ctx->off[3], grads + ctx->off[4],
                                                          ctx->part_sq + nsq, (unsigned long long*)(ctx->counters + 16 + 4 * TICKET_WORDS));
    WSAE_LAUNCH_CHECK();
    ctx->n_sq_parts = nsq + 1;  // wsae_adamw_step(norm_from_wgrad = 1) sums these: no separate norm pass
    ctx->g_is_bf16 = 1;
    return WSAE_OK;
}

template <typename T>
int forward_t(wsae_ctx* ctx, const float* params, const void* x, int x_dtype, const int32_t* rows, int B, float weight,
              float* hidden, float* recon, wsae_stats* stats, float* sparsity_out, hipStream_t st) {
    const int D = ctx->D, H = ctx->H;
    const int ldT = (B + 127) / 128 * 128;
    const ReluWs ws = host_ws(ctx);
    // fp8 forward (BF16 mode + wsae_ctx_set_relu_fp8): the two forward GEMMs take e4m3 copies of their operands, one
    // dequantisation scale per row; everything else (hidden, residual, loss, the whole backward) is the bf16 path's
    // Eligibility is decided from the shape BEFORE anything is queued (the conditions of wsae_internal_gemm256d_fp8 for
    // both GEMMs): a tail batch of an epoch or a small eval batch under precision = "fp8" runs on the bf16 path instead of
    // failing after its quantisation launches.
    const bool fp8 = sizeof(T) == 2 && ctx->relu_fp8 && B >= 512 && D >= 128 && D % 256 == 0 && H % 256 == 0;
    int rc = fp8 ? wsae_internal_stage(ctx, params, x, x_dtype, rows, B, st)  // (the backward reads the staged xT)
                 : wsae_internal_stage_and_gemm(ctx, params, x, x_dtype, rows, B, ctx->pre, nullptr, 0, st);  // pre = x W_e^T + b_e
    if (rc) return rc;
    if (fp8) {
        quant_rows(st, x, x_dtype, rows, B, D, ws.xq, ws.sx);
        quant_rows(st, ctx->We_bf16, WSAE_DT_BF16, nullptr, H, D, ws.weq, ws.swe);
        WSAE_REQUIRE(wsae_internal_gemm256d_fp8(ctx, ws.xq, D, ws.weq, D, ws.sx, ws.swe, params + ctx->off[2], ctx->pre, H, B, H, D, st),
                     "wsae_relu_forward: the fp8 forward needs batch >= 512, hidden_dim %% 4 == 0, input_dim %% 256 == 0 (got B %d, D %d)", B, D);
    }
    const T* wdt = sizeof(T) == 2 ? (const T*)ctx->WdT_bf16 : (const T*)(params + ctx->off[1]);
    transpose_w_kernel<T><<<dim3(ceil_div(H, 64), ceil_div(D, 64)), 256, 0, st>>>(wdt, (T*)ws.wd_nt, H, D);
    dim3 ga(ceil_div(H, 64), ceil_div(ldT, 64));
    relu_act_kernel<T><<<ga, 256, 0, st>>>(ctx->pre, hidden, (T*)ws.hid, (T*)ws.hidT, B, H, ldT, ws.part, ws.nblk, ctx->relu_l1w);
    if (fp8) {
        quant_rows(st, ws.hid, WSAE_DT_BF16, nullptr, B, H, ws.hidq, ws.sh);
        tensor_amax_kernel<<<1024, 256, 0, st>>>((const bf16_t*)ws.wd_nt, (int64_t)H * D / 8, ctx->part_sq);
        quant_fixed_kernel<<<1024, 256, 0, st>>>((const bf16_t*)ws.wd_nt, ctx->part_sq, 1024, D, H, ws.wdq, ws.swd);
        WSAE_REQUIRE(wsae_internal_gemm256d_fp8(ctx, ws.hidq, H, ws.wdq, H, ws.sh, ws.swd, params + ctx->off[3], recon, D, B, D, H, st),
                     "wsae_relu_forward: the fp8 forward needs input_dim >= 128 and hidden_dim %% 256 == 0 (got D %d, H %d)", D, H);
    } else {
        gemm_nt<T>(ctx, st, ws.hid, H, ws.wd_nt, H, params + ctx->off[3], recon, D, B, D, H, 1, 0);  // recon = hidden W_d^T + b_d
    }
    dim3 gr(ceil_div(D, 64), ceil_div(ldT, 64));
    if (x_dtype == WSAE_DT_F32)
        resid_kernel<T, WSAE_DT_F32, false><<<gr, 256, 0, st>>>(recon, x, rows, B, D, ldT, 0.f, nullptr, nullptr, nullptr,
                                                               ws.part + 2 * ws.nblk);
    else
        resid_kernel<T, WSAE_DT_BF16, false><<<gr, 256, 0, st>>>(recon, x, rows, B, D, ldT, 0.f, nullptr, nullptr, nullptr,
                                                                ws.part + 2 * ws.nblk);
    relu_fwd_finish_kernel<<<1, 256, 0, st>>>(ws.part, (int)(ga.x * ga.y), (int)(gr.x * gr.y), ws.nblk, B, ctx->loss_cols, H, weight,
                                              stats, sparsity_out, ws.scal);
    WSAE_LAUNCH_CHECK();
    return WSAE_OK;
}

template <typename T>
int backward_t(wsae_ctx* ctx, const float* params, const void* x, int x_dtype, const int32_t* rows, int B, float weight,
               const float* hidden, const float* recon, float* grads, hipStream_t st) {
    const int D = ctx->D, H = ctx->H;
    const int ldT = (B + 127) / 128 * 128;
    const ReluWs ws = host_ws(ctx);
    const float scale = 2.0f / ((float)B * (float)ctx->loss_cols);  // (the crosscoder sums per-layer means: loss_cols = d_model)
    dim3 gr(ceil_div(D, 64), ceil_div(ldT, 64));
    // g (row-major, into the staging buffer xb: the encoder GEMM is done with it), gT, db_d partials
    if (x_dtype == WSAE_DT_F32)
        resid_kernel<T, WSAE_DT_F32, true><<<gr, 256, 0, st>>>(recon, x, rows, B, D, ldT, scale, (T*)ctx->xb, (T*)ctx->gT,
                                                              ctx->part_dbd, ws.part + 2 * ws.nblk);
    else
        resid_kernel<T, WSAE_DT_BF16, true><<<gr, 256, 0, st>>>(recon, x, rows, B, D, ldT, scale, (T*)ctx->xb, (T*)ctx->gT,
                                                               ctx->part_dbd, ws.part + 2 * ws.nblk);
    const T* wdt = sizeof(T) == 2 ? (const T*)ctx->WdT_bf16 : (const T*)(params + ctx->off[1]);
    gemm_nt<T>(ctx, st, ctx->xb, D, wdt, D, nullptr, ctx->pre, H, B, H, D, 1, 0);  // dh = g W_d  [B][H]
    dim3 ga(ceil_div(H, 64), ceil_div(ldT, 64));
    dpre_kernel<T><<<ga, 256, 0, st>>>(ctx->pre, hidden, B, H, ldT, weight / ((float)B * (float)H), (T*)ws.dpreT, ws.colpart, ctx->relu_l1w);
    // split-K contractions over the batch into the slabs: [z][ dW_e (H*D) | dW_dT (H*D) ]
    const int nz = min(WSAE_WGRAD_MAX_SPLIT, max(1, ldT / 512));
    const int64_t hd = (int64_t)H * D, slab_stride = 2 * hd;
    int nz_e = 1, nz_d = 1;
    gemm_nt<T>(ctx, st, ws.dpreT, ldT, ctx->xT, ldT, nullptr, ctx->wg_slabs, D, H, D, ldT, nz, slab_stride, &nz_e);
    gemm_nt<T>(ctx, st, ws.hidT, ldT, ctx->gT, ldT, nullptr, ctx->wg_slabs + hd, D, H, D, ldT, nz, slab_stride, &nz_d);
    const int nz_eff = nz_e;  // both contractions have the same shape, hence the same split
    (void)nz_d;
    slab_sum_kernel<<<512, 256, 0, st>>>(ctx->wg_slabs, slab_stride, nz_eff, slab_stride / 4, grads, grads + ctx->off[4], D);
    colsum_kernel<<<ceil_div(H, 256), 256, 0, st>>>(ws.colpart, ceil_div(B, 64), H, grads + ctx->off[2]);
    colsum_kernel<<<ceil_div(D, 256), 256, 0, st>>>(ctx->part_dbd, ceil_div(B, 64), D, grads + ctx->off[3]);
    WSAE_LAUNCH_CHECK();
    ctx->n_sq_parts = 0;  // the optimizer computes the norm itself (norm_from_wgrad = 0)
    return WSAE_OK;
}

}  // namespace

extern "C" int wsae_ctx_set_relu_fp8(wsae_ctx* ctx, int32_t on) {
    WSAE_REQUIRE(ctx, "wsae_ctx_set_relu_fp8: null ctx");
    WSAE_REQUIRE(!on || ctx->prec == WSAE_PREC_BF16, "wsae_ctx_set_relu_fp8: the fp8 forward belongs to the BF16 mode");
    ctx->relu_fp8 = on ? 1 : 0;
    return WSAE_OK;
}

extern "C" int wsae_ctx_reserve_relu(wsae_ctx* ctx) {
    WSAE_REQUIRE(ctx, "wsae_ctx_reserve_relu: null ctx");
    return reserve_ws(ctx);
}

extern "C" int wsae_ctx_set_relu_l1_weights(wsae_ctx* ctx, const float* weights) {
    WSAE_REQUIRE(ctx, "wsae_ctx_set_relu_l1_weights: null ctx");
    ctx->relu_l1w = weights;
    return WSAE_OK;
}

extern "C" int wsae_relu_needs_hidden(const wsae_ctx* ctx, int32_t B) { return (ctx && x_flow_ok(ctx, B)) ? 0 : 1; }

extern "C" int wsae_relu_forward(wsae_ctx* ctx, const float* params, const void* x, int32_t x_dtype, const int32_t* rows,
                                 int32_t B, float sparsity_weight, float* hidden, float* recon, wsae_stats* stats,
                                 float* sparsity_loss_out, void* stream) {
    WSAE_REQUIRE(ctx && params && x, "wsae_relu_forward: null argument");
    WSAE_REQUIRE(x_dtype == WSAE_DT_F32 || x_dtype == WSAE_DT_BF16, "wsae_relu_forward: unknown activation dtype %d", x_dtype);
    int rc = check_dims(ctx, B, "wsae_relu_forward");
    if (rc) return rc;
    WSAE_REQUIRE(ctx->relu_ws, "wsae_relu_forward: call wsae_ctx_reserve_relu(ctx) once after wsae_ctx_create");
    hipStream_t st = (hipStream_t)stream;
    ctx->relu_x_B = ctx->relu_g_B = 0;
    if (x_flow_ok(ctx, B)) return forward_x(ctx, params, x, x_dtype, rows, B, sparsity_weight, hidden, recon, stats, sparsity_loss_out, st);
    WSAE_REQUIRE(hidden && recon, "wsae_relu_forward: this shape needs the fp32 hidden and recon buffers (only the row-major-GEMM flow runs without them)");
    return ctx->prec == WSAE_PREC_BF16
               ? forward_t<bf16_t>(ctx, params, x, x_dtype, rows, B, sparsity_weight, hidden, recon, stats, sparsity_loss_out, st)
               : forward_t<float>(ctx, params, x, x_dtype, rows, B, sparsity_weight, hidden, recon, stats, sparsity_loss_out, st);
}

extern "C" int wsae_relu_backward(wsae_ctx* ctx, const float* params, const void* x, int32_t x_dtype, const int32_t* rows,
                                  int32_t B, float sparsity_weight, const float* hidden, const float* recon, float* grads,
                                  void* stream) {
    WSAE_REQUIRE(ctx && params && x && grads, "wsae_relu_backward: null argument");
    WSAE_REQUIRE(x_dtype == WSAE_DT_F32 || x_dtype == WSAE_DT_BF16, "wsae_relu_backward: unknown activation dtype %d", x_dtype);
    int rc = check_dims(ctx, B, "wsae_relu_backward");
    if (rc) return rc;
    WSAE_REQUIRE(ctx->relu_ws, "wsae_relu_backward: no preceding wsae_relu_forward on this ctx");
    hipStream_t st = (hipStream_t)stream;
    // (the forward of this batch ran the row-major-GEMM flow: its bf16 hidden is still in the workspace)
    if (ctx->relu_x_B == B && x_flow_ok(ctx, B)) return backward_x(ctx, params, x, x_dtype, rows, B, sparsity_weight, recon, grads, st);
    WSAE_REQUIRE(hidden && recon, "wsae_relu_backward: the fp32 hidden and recon of the forward are needed on this path");
    return ctx->prec == WSAE_PREC_BF16
               ? backward_t<bf16_t>(ctx, params, x, x_dtype, rows, B, sparsity_weight, hidden, recon, grads, st)
               : backward_t<float>(ctx, params, x, x_dtype, rows, B, sparsity_weight, hidden, recon, grads, st);
}
