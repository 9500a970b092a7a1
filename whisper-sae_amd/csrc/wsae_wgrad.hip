// Weight gradients (second half of backward).
//   reference: autograd of nn.Linear encoder/decoder, model.py:111,129:
//     dW_d = g^T hidden          -> here dW_dT[h,:] = sum_b hidden[b,h] * g[b,:]
//     dW_e = dpre^T (x - b_pre)  -> here dW_e[h,:]  = sum_b dpre[b,h]   * x_c[b,:]
//     db_e = sum_b dpre ; db_d = sum_b g ; db_pre = db_d - W_e^T db_e
//
// Both contractions are [H x B] x [B x D] with a left operand that is 99% zeros and only exists
// as the compact TopK code (vals, idx)[B,k].  A scatter-add of B*k rows would be atomic-bound
// (2 x 805 MB of float atomics at cfg2/B=16384 against ~1.3 TB/s); instead each workgroup rebuilds
// its 128-feature x KT-row slice of the left operand in LDS from the compact code (zero fill +
// scatter of the entries whose feature falls in the tile) and runs it through the same NT MFMA
// slab code as the encode GEMM.  The right operands are the transposed batch operands xT / gT
// left in the ctx by the encode and decode launches.
#include "wsae_common.h"
#include "wsae_mfma.h"

template <typename T>
__global__ void __launch_bounds__(256)
wgrad_kernel(const float* __restrict__ vals, const int32_t* __restrict__ idx, const float* __restrict__ dpre,
             const T* __restrict__ xT, const T* __restrict__ gT, int B, int ldT, int H, int D, int K, int nsplit,
             float* __restrict__ dWe, float* __restrict__ dWdT, float* __restrict__ dbe) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* As = smem;
    char* Bs = smem + TILE_LDS_BYTES;
    float* dbe_s = (float*)(smem + 2 * TILE_LDS_BYTES);  // [128]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int f0 = blockIdx.x * TILE_M, d0 = blockIdx.y * TILE_N;
    const int which = blockIdx.z / nsplit;  // 0: dW_dT (hidden, g)   1: dW_e (dpre, x_c)
    const int split = blockIdx.z % nsplit;
    constexpr int KT = Mfma<T>::KT;
    const T* Bt = which == 0 ? gT : xT;
    const float* sv = which == 0 ? vals : dpre;
    const bool do_dbe = (which == 1) && (blockIdx.y == 0);

    const int nchunks = (B + KT - 1) / KT;
    const int per = (nchunks + nsplit - 1) / nsplit;
    const int c_begin = split * per, c_end = min(nchunks, c_begin + per);

    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    if (tid < 128) dbe_s[tid] = 0.f;

    SlabRegs<T> rb;
    if (c_begin < c_end) slab_load<T>(rb, Bt, ldT, d0, D, c_begin * KT, ldT, tid);
    for (int ck = c_begin; ck < c_end; ++ck) {
        const int b0 = ck * KT;
        // zero the sparse slice, stage the dense one
        for (int c = tid; c < TILE_LDS_BYTES / 16; c += 256) *(uint4*)(As + c * 16) = make_uint4(0, 0, 0, 0);
        slab_store<T>(rb, Bs, tid);
        __syncthreads();
        if (ck + 1 < c_end) slab_load<T>(rb, Bt, ldT, d0, D, (ck + 1) * KT, ldT, tid);
        // scatter the compact entries of rows [b0, b0+KT) whose feature lies in [f0, f0+128)
        const int nrow = min(KT, B - b0);
        const int nent = nrow * K;
        const int64_t base = (int64_t)b0 * K;
        for (int e = tid; e < nent; e += 256) {
            const int f = idx[base + e] - f0;
            if ((unsigned)f < 128u) {
                float v = sv[base + e];
                if (which == 0) v = v > 0.f ? v : 0.f;  // hidden = relu(topk value)
                if (v != 0.f) {
                    const int bl = e / K;
                    *(T*)(As + f * LDS_ROW_BYTES + bl * (int)sizeof(T)) = (T)v;
                    if (do_dbe) atomicAdd(&dbe_s[f], v);
                }
            }
        }
        __syncthreads();
        Mfma<T>::slab(As, Bs, wm * 64, wn * 64, lane, acc);
        __syncthreads();
    }

    float* out = which == 0 ? dWdT : dWe;
    const int col = lane & 31, rq = lane >> 5;
#pragma unroll
    for (int mi = 0; mi < 2; ++mi)
#pragma unroll
        for (int ni = 0; ni < 2; ++ni) {
            const int d = d0 + wn * 64 + ni * 32 + col;
            if (d >= D) continue;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int f = f0 + wm * 64 + mi * 32 + (r & 3) + 8 * (r >> 2) + 4 * rq;
                if (f < H) {
                    const float v = acc[mi][ni][r];
                    if (nsplit == 1)
                        out[(int64_t)f * D + d] = v;
                    else if (v != 0.f)
                        atomicAdd(out + (int64_t)f * D + d, v);
                }
            }
        }
    if (do_dbe && tid < 128 && f0 + tid < H) {
        const float v = dbe_s[tid];
        if (v != 0.f) atomicAdd(dbe + f0 + tid, v);
    }
}

// db_d, db_pre and (BF16 mode) the rank-1 correction of the folded pre-bias:
//   db_pre = db_d - W^T db_e            (W = the encoder weights the forward actually used)
//   dW_e[h,:] -= db_e[h] * b_pre        (BF16 mode: the MFMA contraction saw raw x, not x - b_pre)
// One block per 32 feature rows; partial GEMV sums join db_pre by float atomics.
template <typename TW, bool FOLD>
__global__ void __launch_bounds__(256)
bias_grads_kernel(const TW* __restrict__ W, const float* __restrict__ bpre, const float* __restrict__ dbd_acc,
                  float* __restrict__ dWe, const float* __restrict__ dbe, float* __restrict__ dbd,
                  float* __restrict__ dbpre, int H, int D) {
    __shared__ float e_s[32];
    const int h0 = blockIdx.x * 32;
    if (threadIdx.x < 32) e_s[threadIdx.x] = (h0 + (int)threadIdx.x < H) ? dbe[h0 + threadIdx.x] : 0.f;
    __syncthreads();
    for (int d = threadIdx.x; d < D; d += 256) {
        float a = 0.f;
        const float bp = bpre[d];
#pragma unroll 4
        for (int i = 0; i < 32; ++i) {
            const int h = h0 + i;
            if (h < H) {
                const float e = e_s[i];
                a = fmaf(e, (float)W[(int64_t)h * D + d], a);
                if (FOLD && e != 0.f) dWe[(int64_t)h * D + d] -= e * bp;
            }
        }
        float add = -a;
        if (blockIdx.x == 0) {
            const float s = dbd_acc[d];
            dbd[d] = s;
            add += s;
        }
        if (add != 0.f) atomicAdd(dbpre + d, add);
    }
}

// (sum_b g was accumulated by the decode launch into the [D] accumulator at the head of part_dbd.)
extern "C" int wsae_weight_grads(wsae_ctx* ctx, const float* params, const void* x, int32_t x_dtype,
                                 const int32_t* rows, const float* vals, const int32_t* idx, const float* dpre,
                                 int32_t B, float* grads, void* stream) {
    (void)x; (void)x_dtype; (void)rows;  // the staged transposes in ctx (xT, gT) carry the batch
    WSAE_REQUIRE(ctx && params && vals && idx && dpre && grads, "wsae_weight_grads: null argument");
    WSAE_REQUIRE(B >= 1 && B <= ctx->maxB, "wsae_weight_grads: batch %d outside [1, %d]", B, ctx->maxB);
    hipStream_t st = (hipStream_t)stream;
    const int D = ctx->D, H = ctx->H, K = ctx->K;
    const int ldT = (B + 127) / 128 * 128;
    WSAE_PROF_BEGIN(ctx, WSAE_K_MEMSET, st);
    WSAE_HIP_CHECK(hipMemsetAsync(grads, 0, (size_t)ctx->P * 4, st));
    WSAE_PROF_END(ctx, WSAE_K_MEMSET, st);
    float* dWe = grads + ctx->off[0];
    float* dWdT = grads + ctx->off[1];
    float* dbe = grads + ctx->off[2];
    float* dbd = grads + ctx->off[3];
    float* dbpre = grads + ctx->off[4];
    const int tiles = ceil_div(H, TILE_M) * ceil_div(D, TILE_N) * 2;
    const int kt = ctx->prec == WSAE_PREC_BF16 ? 64 : 32;
    const int nchunks = ceil_div(B, kt);
    int nsplit = max(1, min(nchunks, ceil_div(1024, tiles)));  // aim for ~4 workgroups per CU
    if (nsplit > 1 && nchunks / nsplit < 8) nsplit = max(1, nchunks / 8);
    dim3 grid(ceil_div(H, TILE_M), ceil_div(D, TILE_N), 2 * nsplit);
    const size_t sh = 2 * TILE_LDS_BYTES + 128 * sizeof(float);
    WSAE_PROF_BEGIN(ctx, WSAE_K_WGRAD, st);
    if (ctx->prec == WSAE_PREC_BF16)
        wgrad_kernel<bf16_t><<<grid, 256, sh, st>>>(vals, idx, dpre, (const bf16_t*)ctx->xT, (const bf16_t*)ctx->gT, B,
                                                    ldT, H, D, K, nsplit, dWe, dWdT, dbe);
    else
        wgrad_kernel<float><<<grid, 256, sh, st>>>(vals, idx, dpre, (const float*)ctx->xT, (const float*)ctx->gT, B, ldT,
                                                   H, D, K, nsplit, dWe, dWdT, dbe);
    WSAE_PROF_END(ctx, WSAE_K_WGRAD, st);
    WSAE_LAUNCH_CHECK();
    const float* bpre = params + ctx->off[4];
    WSAE_PROF_BEGIN(ctx, WSAE_K_BIAS_GRADS, st);
    if (ctx->prec == WSAE_PREC_BF16)
        bias_grads_kernel<bf16_t, true><<<ceil_div(H, 32), 256, 0, st>>>(ctx->We_bf16, bpre, ctx->part_dbd, dWe, dbe, dbd,
                                                                        dbpre, H, D);
    else
        bias_grads_kernel<float, false><<<ceil_div(H, 32), 256, 0, st>>>(params + ctx->off[0], bpre, ctx->part_dbd, dWe,
                                                                        dbe, dbd, dbpre, H, D);
    WSAE_PROF_END(ctx, WSAE_K_BIAS_GRADS, st);
    WSAE_LAUNCH_CHECK();
    return WSAE_OK;
}
