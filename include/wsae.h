/*
 * wsae.h -- C ABI of libwsae_hip.so: the MI355X (gfx950 / CDNA4) SAE train-step kernels.
 *
 * This is the drop-in boundary underneath the reference's Python module API.  The reference
 * (omarkhursheed/whisper-sae) has no FFI layer of its own: its hot path is an implicit ATen op
 * sequence issued from src/whisper_sae/sae/model.py and src/whisper_sae/sae/training.py.  Each
 * entry point below names the reference lines whose arithmetic it replaces; the ctypes binding a
 * maintainer adds on the reference side is shown in INTEGRATION.md and lives, for this build, in
 * whisper-sae_amd/whisper_sae/_native.py.
 *
 * Conventions
 *   - extern "C", plain pointers and sizes, no torch types.
 *   - every data pointer is a caller-owned DEVICE pointer (e.g. torch.Tensor.data_ptr()); nothing
 *     is retained past the call except what a wsae_ctx / wsae_ring handle allocates for itself.
 *   - every launch goes to the caller's stream (hipStream_t passed as void*); no hidden
 *     synchronisation, no allocation inside launch functions (graph-capturable).
 *   - return value: 0 = ok, negative = error; wsae_last_error() gives the thread-local message.
 *   - one ctx per device per process; a ctx is not thread-safe.
 *
 * Flat parameter layout ("pack"): all five parameter tensors of a TopKSAE live in ONE float32
 * buffer of wsae_param_count(D,H) elements (gradients, Adam exp_avg and exp_avg_sq use the same
 * layout in their own buffers), so the optimizer is one pass and the data-parallel exchange is one
 * RCCL all-reduce:
 *     [ W_e  : H*D ]  encoder.weight, row-major [H][D]            (model.py:63)
 *     [ W_dT : H*D ]  decoder.weight TRANSPOSED, row-major [H][D] (model.py:64; row h = decoder
 *                     column h, so decode gathers contiguous rows and the unit-norm constraint of
 *                     model.py:91-96 is a per-row operation)
 *     [ b_e  : H   ]  encoder.bias
 *     [ b_d  : D   ]  decoder.bias
 *     [ b_pre: D   ]  pre-encoder bias                            (model.py:67)
 */
#ifndef WSAE_H_
#define WSAE_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define WSAE_VERSION 1

/* error codes */
#define WSAE_OK 0
#define WSAE_ERR_INVALID (-1) /* bad argument / unsupported shape */
#define WSAE_ERR_HIP (-2)     /* a HIP runtime call failed */
#define WSAE_ERR_NOMEM (-3)

/* arithmetic mode of the two contractions (encode GEMM, weight-gradient GEMMs) */
#define WSAE_PREC_BF16 0 /* bf16 MFMA operands, fp32 accumulate ("use_amp": training.py:73-75,179) */
#define WSAE_PREC_FP32 1 /* fp32 MFMA (v_mfma_f32_32x32x2_f32), the reference's CPU fp32 semantics */

/* element type of an activation buffer handed in */
#define WSAE_DT_F32 0
#define WSAE_DT_BF16 1

typedef struct wsae_ctx wsae_ctx;
typedef struct wsae_ring wsae_ring;

typedef struct wsae_config {
    int32_t input_dim;  /* D: multiple of 32, <= 2048 */
    int32_t hidden_dim; /* H: multiple of 32 */
    int32_t k;          /* TopK k: 1..128, <= H */
    int32_t max_batch;  /* largest B any call will pass */
    int32_t precision;  /* WSAE_PREC_* */
    int32_t device;     /* HIP device ordinal */
} wsae_config;

/* device-side step record written by the kernels, fetched lazily by the host
 * (replaces the five .item() syncs of training.py:207-213) */
typedef struct wsae_stats {
    float loss;        /* mean((recon-x)^2)                 model.py:145 */
    float l0;          /* mean_b count(hidden>0)            model.py:148 */
    float grad_norm;   /* global L2 norm before clipping    training.py:188 */
    float clip_coef;   /* min(1, max_norm/(norm+1e-6)) */
    float dead_ratio;  /* get_dead_feature_ratio()          model.py:192-195 */
    int32_t dead_count;
    int32_t topk_fallback_rows; /* rows that took the exact bisection path of the TopK kernel */
    int32_t reserved;
} wsae_stats;

const char* wsae_last_error(void);
int wsae_version(void);

/* number of float32 elements of the flat pack, and element offsets of its five segments
 * (order: W_e, W_dT, b_e, b_d, b_pre) */
int64_t wsae_param_count(int32_t input_dim, int32_t hidden_dim);
int wsae_param_offsets(int32_t input_dim, int32_t hidden_dim, int64_t offsets[5]);

/* ctx: dims + mode + all workspace (bf16 weight shadows, TopK scratch, partial-sum slabs). */
int wsae_ctx_create(const wsae_config* cfg, wsae_ctx** out);
int wsae_ctx_destroy(wsae_ctx* ctx);
/* Data-parallel dead-feature clock without a second collective.  `fired` = float[hidden_dim], zero
 * before the first step, or NULL to switch the mechanism off (the default).  When set:
 *   - wsae_decode_loss stores 1.0f to fired[f] for every feature f it stamps in last_activated
 *     (model.py:178-181);
 *   - the caller sums `fired` over the ranks (it rides at the tail of the gradient all-reduce);
 *   - wsae_adamw_step then sets last_activated[f] = *step_count wherever fired[f] > 0 and clears
 *     fired for the next step.
 * With clocks that agreed before the step this equals all_reduce(MAX) of last_activated: a feature
 * either fired somewhere in this step (new value = the step) or nowhere (value unchanged, equal on
 * every rank).  Replaces the 8*H-byte MAX all-reduce the reference's semantics would otherwise need
 * under DDP (the reference itself is single-process). */
int wsae_ctx_set_fired(wsae_ctx* ctx, float* fired);
/* Selective strip stores of the encoder GEMM (no counterpart in the reference: a property of this implementation of
 * model.py:111-114).  The TopK of wsae_encode_topk / wsae_encode_decode reads a 16-column strip of the pre-activation
 * matrix only when the strip's maximum reaches the row's threshold T, so for batches served by the persistent GEMM
 * (bf16 mode, B >= 2048) that GEMM writes to HBM only the strips whose maximum reaches s x the smallest T of the
 * previous TWO batches on this ctx, with a margin s that adapts by itself (0.25 .. 1; wsae_topk.h).  The prediction is
 * verified row by row in the TopK launch, and a row that needs a strip that was not stored recomputes it with the GEMM's
 * own arithmetic: outputs are bit-identical with the feature on or off, whatever the history; only the time differs
 * (DESIGN.md section 4.1).
 *   on: 0 disables, non-zero enables (default: enabled, or disabled when the environment has WSAE_STRIP_PREDICT=0 at
 *   wsae_ctx_create time - for A/B timing runs; WSAE_STRIP_SAFETY=<s> there fixes the margin).
 *   assume_store_threshold: NaN forgets the history (the next launch stores every strip); any other value is the store
 *   threshold the next launch uses (tests pass a huge value to force every row through the recompute path). */
int wsae_ctx_set_strip_predict(wsae_ctx* ctx, int32_t on, float assume_store_threshold);
/* refilled_rows: rows that recomputed strips since the ctx was created (cumulative); last_min_threshold: the smallest row
 * threshold of the last predicated TopK launch (NaN when there is none); margin: the s of the last predicated GEMM launch
 * (0 when there is none).  Synchronises the device. */
int wsae_ctx_strip_stats(wsae_ctx* ctx, int64_t* refilled_rows, float* last_min_threshold, float* margin);
/* Number of columns the reconstruction MSE (and g = 2 r / (B cols)) averages over; default input_dim.  A transcoder
 * whose output is narrower than its input runs on a ctx padded to the wider of the two and sets this to its
 * output_dim (F.mse_loss means over B * output_dim, transcoder.py:152). */
int wsae_ctx_set_loss_cols(wsae_ctx* ctx, int32_t cols);
size_t wsae_ctx_workspace_bytes(const wsae_ctx* ctx);
/* Allocate the dense workspace of the ReLU SAE path (3 x max_batch x hidden_dim operand copies) on the ctx's
 * device.  Call once after wsae_ctx_create, before the first wsae_relu_forward: launch functions never allocate. */
int wsae_ctx_reserve_relu(wsae_ctx* ctx);

/* Refresh what the kernels derive from the master weights: bf16 shadow of W_e and the folded
 * encoder bias c[h] = b_e[h] - bf16(W_e)[h,:] . b_pre (BF16 mode).  Must be called after any
 * change to `params` that did not go through wsae_adamw_step (which refreshes them itself). */
int wsae_prepare(wsae_ctx* ctx, const float* params, void* stream);

/* ---- forward --------------------------------------------------------------------------------
 * x: [B, D] activations of dtype x_dtype; `rows` (nullable) gathers: batch row b is x[rows[b], :]
 * (this is how batches are drawn from the on-device ring buffer without a copy). */

/* TopKSAE.encode up to the TopK (model.py:108-114): pre = (x - b_pre) W_e^T + b_e, then the k
 * largest per row, sorted descending (ties: lowest index first).  Compact code out:
 * vals [B,k] f32 (pre-activation values, NOT yet relu'd), idx [B,k] i32.
 * step_count (nullable, device int64): incremented by one = the dead-feature clock of
 * model.py:175; pass it only for training-mode forwards. */
int wsae_encode_topk(wsae_ctx* ctx, const float* params, const void* x, int32_t x_dtype,
                     const int32_t* rows, int32_t B, float* vals, int32_t* idx,
                     int64_t* step_count, wsae_stats* stats, void* stream);

/* Dense pre-activations [B,H] f32 (model.py:111), for API users / tests. */
int wsae_encode_dense(wsae_ctx* ctx, const float* params, const void* x, int32_t x_dtype,
                      const int32_t* rows, int32_t B, float* pre, void* stream);

/* hidden = zeros; hidden[idx] = relu(vals) (model.py:115-116).  hidden: [B,H] f32. */
int wsae_densify(wsae_ctx* ctx, const float* vals, const int32_t* idx, int32_t B, float* hidden,
                 void* stream);

/* TopKSAE.decode for an arbitrary dense code (model.py:129): recon = hidden W_d^T + b_d + b_pre. */
int wsae_decode_dense(wsae_ctx* ctx, const float* params, const float* hidden, int32_t B,
                      float* recon, void* stream);

/* Sparse decode + MSE + (optionally) the first half of backward, one pass over the compact code
 * (model.py:129,145,148,168-181 and the autograd of them):
 *   recon = sum_j relu(v_j) W_dT[idx_j,:] + b_d + b_pre ; loss = mean((recon-x)^2) ; l0
 *   want_bwd bit 0: g = 2 (recon-x)/(B*D) kept in ctx workspace (in the contraction dtype),
 *            dpre[b,j] = (v_j>0) ? g . W_dT[idx_j,:] : 0;   bit 1 (value 2, with bit 0): also keep the fp32 g that
 *            wsae_input_grad reads (only the autograd API path needs dL/dx)
 *   last_activated (nullable, device int64[H]) with step_count (device int64): features with
 *   v_j > 0 get last_activated = *step_count.
 * recon (nullable): [B,D] f32.  dpre (required when want_bwd): [B,k] f32.
 * stats->loss / stats->l0 are written (device). */
int wsae_decode_loss(wsae_ctx* ctx, const float* params, const void* x, int32_t x_dtype,
                     const int32_t* rows, const float* vals, const int32_t* idx, int32_t B,
                     float* recon, int32_t want_bwd, float* dpre, int64_t* last_activated,
                     const int64_t* step_count, wsae_stats* stats, void* stream);

/* TopKSAE.forward in one call (model.py:131-166) = wsae_encode_topk followed by wsae_decode_loss on the same batch:
 * same arguments, same outputs (vals / idx are OUTPUTS here), same record. */
int wsae_encode_decode(wsae_ctx* ctx, const float* params, const void* x, int32_t x_dtype, const int32_t* rows,
                       int32_t B, float* vals, int32_t* idx, int64_t* step_count, float* recon, int32_t want_bwd,
                       float* dpre, int64_t* last_activated, wsae_stats* stats, void* stream);

/* Second half of backward (autograd of model.py:111,129 w.r.t. the parameters): the two
 * [H,B]x[B,D] contractions on MFMA with the sparse operand rebuilt in LDS from the compact code,
 * plus the three bias gradients.  Needs the g left in ctx by wsae_decode_loss(want_bwd=1) on the
 * same batch, and the SAME x / x_dtype / rows as that forward (a bf16 batch in BF16 mode is not staged by the
 * forward: this call transposes it for the dW_e contraction).  grads: flat pack, overwritten. */
int wsae_weight_grads(wsae_ctx* ctx, const float* params, const void* x, int32_t x_dtype,
                      const int32_t* rows, const float* vals, const int32_t* idx,
                      const float* dpre, int32_t B, float* grads, void* stream);

/* ---- data parallel: the gradients go straight onto the exchange buffer -------------------------------------------
 * (absent from the reference, which is single-process).  The WIRE is what the ranks
 * all-reduce(SUM): `hidden_dim * input_dim` elements of dW_dT, then dW_e, db_e, db_d, db_pre (the rest of the pack) and
 * the `hidden_dim` fired indicators of wsae_ctx_set_fired, then WSAE_WIRE_METRIC_SLOTS metric digits (below): P + hidden_dim +
 * WSAE_WIRE_METRIC_SLOTS elements of `wire_dtype` (WSAE_DT_F32: the exact data-parallel gradient; WSAE_DT_BF16: half the bytes
 * over xGMI, every rank's gradient rounded once to bf16, the indicators - sums of at most world_size ones - and the digits
 * exact).  Normally ONE call with part = WSAE_PART_ALL and one all-reduce.  The decoder matrix comes FIRST so that the
 * optional two halves of the backward fill two contiguous ranges:
 *   part = WSAE_PART_DECODER  contraction of dW_dT alone (split-K 16) + its reduction -> wire[0, H D)
 *   part = WSAE_PART_ENCODER  contraction of dW_e alone + reduction + the three bias gradients + the indicators
 *                             -> wire[H D, P + H)          (same batch, after the decoder part)
 *   part = WSAE_PART_ALL      both contractions in one launch (the single-GPU geometry) -> the whole wire
 * With halves the caller starts the all-reduce of wire[0, H D) after the decoder part, on its communication stream, and it runs
 * under the encoder part's contraction (measured on MI355X: the split costs +86 us of kernels and stream hand-overs per step
 * against +16 us for WSAE_PART_ALL with one in-stream collective - DESIGN.md section 6).  Halves need input_dim > 256 (wsae_wgrad_parts_supported); the gradient pack in
 * `grads` form is NOT written by these calls: wsae_grads_unpack_wire produces it from the summed wire. */
#define WSAE_PART_ALL (-1)
#define WSAE_PART_DECODER 0
#define WSAE_PART_ENCODER 1
int wsae_wgrad_parts_supported(const wsae_ctx* ctx);
/* Compute units the ENCODER part leaves without a workgroup (default 0).  The contraction's workgroups fill the register
 * file of the CU they sit on (two waves x ~246 VGPRs per SIMD), so the collective of the decoder half, issued on another
 * stream while the encoder part runs, only overlaps if some CUs are free for its kernels; the part then runs split-K 13
 * or 14 instead of 16 (a few percent longer).  Unmeasured on hardware so far: the builder's boxes have one GPU. */
int wsae_ctx_set_comm_reserve(wsae_ctx* ctx, int32_t n_cus);
int wsae_weight_grads_wire(wsae_ctx* ctx, const float* params, const void* x, int32_t x_dtype,
                           const int32_t* rows, const float* vals, const int32_t* idx, const float* dpre,
                           int32_t B, int32_t part, void* wire, int32_t wire_dtype, void* stream);

/* Copy of g = 2 (recon - x) / (B cols), fp32 [B, D], as left by the last wsae_decode_loss with want_bwd = 3: the
 * gradient of the loss w.r.t. the reconstruction (= minus its gradient w.r.t. the target; transcoder skip path). */
int wsae_last_residual_grad(wsae_ctx* ctx, int32_t B, float* g_out, void* stream);

/* dL/dx (only the autograd API path needs it): dx = dpre W_e - g (subtract_g = 1: the SAE, whose target is its
 * input; needs the fp32 g, want_bwd = 3) or dx = dpre W_e (subtract_g = 0: transcoders).  dx: [B,D] f32. */
int wsae_input_grad(wsae_ctx* ctx, const float* params, const int32_t* idx, const float* dpre,
                    int32_t B, float* dx, int32_t subtract_g, void* stream);

/* ---- optimizer tail (training.py:186-198, :212) -------------------------------------------------
 * global-L2 clip (clip_grad_norm_, max_norm <= 0 disables) -> AdamW (torch semantics, step is the
 * 1-based update count) -> decoder column renorm (model.py:91-96, if normalize_decoder) ->
 * refresh of the derived shadows -> optional dead-feature scan (model.py:183-195) into stats.
 * grads are scaled by grad_scale first (1/world_size after a SUM all-reduce).
 * norm_from_wgrad: 1 = `grads` is exactly what the preceding wsae_weight_grads on this ctx wrote
 *   (single GPU): the global norm comes from the partial sums that call left behind and one pass
 *   over the gradients is saved; 0 = the gradients were touched since (all-reduce): recompute; 2 = use the partial sums
 *   if the preceding backward on this ctx left any (wsae_relu_backward does on its row-major-GEMM flow), else recompute.
 * last_activated (nullable) / step_count / dead_threshold: when given, stats->dead_count and
 *   stats->dead_ratio are written (get_dead_feature_ratio() of training.py:212).
 * All four buffers use the flat pack layout. */
/* Data parallel: turn the SUMMED wire (layout above, P + hidden_dim elements of wire_dtype) into the fp32 buffer
 * `grads_ext` = [gradient pack in pack order | fired] that wsae_adamw_step reads, and leave the gradient part's
 * squared-norm partials in the ctx, so that the following wsae_adamw_step(norm_from_wgrad = 1, grad_scale = 1 / world)
 * needs no norm pass of its own.  metrics_sum: float[2] = the ranks' summed (loss, l0) of this step from an exchange of the
 * caller's own (may be the loss / l0 words of `stats` themselves), or NULL = take them from the wire's metric elements (when
 * wsae_ctx_set_wire_metrics named a source for them); stats->loss / stats->l0 are overwritten with their means over `world`
 * ranks. */
/* The step's two metric scalars ride on the wire as well (a separate 8-byte all-reduce costs a data-parallel step 16 us of
 * stream hand-over on MI355X): behind the fired indicators the wire carries WSAE_WIRE_METRIC_SLOTS more elements, written by
 * the reduction launch that writes the indicators from the two floats at `loss_l0` (device memory, e.g. the first two words
 * of the step's wsae_stats record; NULL = zeros) - the loss as 40-bit fixed point (2^-24 resolution, range 65536) in ten
 * base-16 digits, l0 as 32-bit fixed point (2^-16) in eight, one digit per element, a non-finite flag in the element after:
 * digit sums over up to 16 ranks stay below 256 and are therefore EXACT in a bf16 all-reduce as well.
 * wsae_grads_unpack_wire(metrics_sum = NULL) decodes them and writes the rank means into stats.  The wire is thus
 * P + hidden_dim + WSAE_WIRE_METRIC_SLOTS elements long. */
#define WSAE_WIRE_METRIC_SLOTS 24
int wsae_ctx_set_wire_metrics(wsae_ctx* ctx, const float* loss_l0);
int wsae_grads_unpack_wire(wsae_ctx* ctx, const void* wire, int32_t wire_dtype, float* grads_ext,
                           const float* metrics_sum, int32_t world, wsae_stats* stats, void* stream);
int wsae_adamw_step(wsae_ctx* ctx, float* params, const float* grads, float* exp_avg,
                    float* exp_avg_sq, float lr, float beta1, float beta2, float eps,
                    float weight_decay, int32_t step, float max_norm, float grad_scale,
                    int32_t normalize_decoder, int32_t norm_from_wgrad,
                    int64_t* last_activated, const int64_t* step_count,
                    int64_t dead_threshold, wsae_stats* stats, void* stream);

/* F.normalize(decoder.weight, dim=0) alone (model.py:91-96) + shadow refresh. */
int wsae_normalize_decoder(wsae_ctx* ctx, float* params, void* stream);

/* ---- dead features (model.py:183-257) -------------------------------------------------------- */
/* stats->dead_count / dead_ratio = #(step_count - last_activated > threshold); mask (nullable):
 * uint8[H]. */
int wsae_dead_scan(wsae_ctx* ctx, const int64_t* last_activated, const int64_t* step_count,
                   int64_t threshold, uint8_t* mask, wsae_stats* stats, void* stream);

/* resample_dead_features (model.py:197-257).  Call order mirrors the reference: (1) wsae_dead_scan
 * -> dead_mask (model.py:215, BEFORE the forward); (2) the forward on `inputs` (encode_topk +
 * decode_loss with a recon buffer; in train mode it bumps the dead-feature clock, model.py:229);
 * (3) wsae_row_errors -> row_err [Br] = sum_d (x-recon)^2; (4) wsae_resample_dead: dead features
 * ascending (at most num_cap, <0 = all), rows by error descending, the L2-normalised raw input row
 * goes to W_e[f,:] and W_dT[f,:], b_e[f] = 0, last_activated[f] = *step_count.  Adam moments
 * untouched.  n_dead_out (device int32): the capped dead count the reference returns
 * (model.py:257), even when fewer than that many rows exist. */
/* Transcoders (sae/transcoder.py:207-252): wsae_row_errors takes the TARGET as x and may also leave the residual
 * rows resid [Br, D] = x - recon (nullable); wsae_resample_dead then writes the L2-normalised row of dec_src
 * (nullable: the residuals) into the decoder column instead of the input direction. */
int wsae_row_errors(wsae_ctx* ctx, const void* x, int32_t x_dtype, const int32_t* rows,
                    const float* recon, int32_t B, float* row_err, float* resid, void* stream);
int wsae_resample_dead(wsae_ctx* ctx, float* params, const void* inputs, int32_t x_dtype,
                       const int32_t* rows, int32_t Br, const float* row_err,
                       const uint8_t* dead_mask, int64_t* last_activated,
                       const int64_t* step_count, int32_t num_cap, int32_t* n_dead_out,
                       const float* dec_src, void* stream);

/* ---- on-device activation ring buffer (replaces data/feature_cache.py:169-197) ---------------
 * capacity rows of D elements (bf16 or f32) resident in HBM; producers push blocks of rows,
 * the trainer draws batches as row-index lists (a seeded permutation per epoch), and the kernels
 * gather rows straight from the ring. */
int wsae_ring_create(int32_t device, int64_t capacity_rows, int32_t dim, int32_t dtype,
                     wsae_ring** out);
int wsae_ring_destroy(wsae_ring* ring);
/* Producer side (row N2, sae/hooks.py:86-92): rows of hidden states [n_rows, dim] go through LayerNorm(gamma, beta, eps;
 * biased variance, as torch.nn.LayerNorm - Whisper's final encoder / decoder norm) and into the ring in its dtype,
 * one pass, nothing on the host. */
int wsae_ring_push_layernorm(wsae_ring* ring, const void* src, int32_t src_dtype, int64_t n_rows,
                             const float* gamma, const float* beta, float eps, void* stream);
void* wsae_ring_data(wsae_ring* ring);         /* device pointer of row 0 */
int64_t wsae_ring_size(const wsae_ring* ring); /* rows currently valid */
/* append n_rows rows ([n_rows, D], device pointer, src_dtype f32/bf16 -> converted to the ring's
 * dtype), wrapping around and overwriting the oldest rows once full */
int wsae_ring_push(wsae_ring* ring, const void* src, int32_t src_dtype, int64_t n_rows, void* stream);
/* rows_out[i] = perm_{seed,epoch}(offset + i) mod size, i < n: a bijective shuffle of [0,size)
 * (the RandomSampler of feature_cache.py:191-197), computed on device */
int wsae_ring_sample(wsae_ring* ring, uint64_t seed, int64_t epoch, int64_t offset, int32_t n,
                     int32_t* rows_out, void* stream);
/* fill with deterministic synthetic activations ~N(0,1) (bench / tests) */
int wsae_ring_fill_synthetic(wsae_ring* ring, uint64_t seed, int64_t n_rows, void* stream);

/* ---- in-library kernel timing (bench.py's roofline leg) ----------------------------------------
 * When enabled for a kernel id, every launch of that kernel on this ctx is bracketed by a pair of
 * HIP events recorded on the launch stream (up to max_samples launches, then recording stops).
 * wsae_profile_read synchronises the recorded events and returns launch count and summed
 * duration.  kernel_id -1 = all kernels.  Ids: see wsae_kernel_name(). */
#define WSAE_K_STAGE_BATCH 0
#define WSAE_K_ENCODE_GEMM 1
#define WSAE_K_TOPK 2          /* standalone TopK launch (absent from the fused training forward) */
#define WSAE_K_DECODE 3        /* decode + loss + dpre (+ the fused TopK) */
#define WSAE_K_BUCKET 4        /* counting sort of the compact code + g transposition */
#define WSAE_K_WGRAD 5         /* the two weight-gradient contractions */
#define WSAE_K_WGRAD_REDUCE 6  /* split-K reduction + bias gradients */
#define WSAE_K_SQNORM 7
#define WSAE_K_ADAMW 8         /* fused optimizer tail */
#define WSAE_K_ROWNORM 9
#define WSAE_K_PREPARE 10
#define WSAE_K_DEAD_SCAN 11
#define WSAE_K_COUNT 12
const char* wsae_kernel_name(int32_t kernel_id);
int wsae_profile_enable(wsae_ctx* ctx, int32_t kernel_id, int32_t max_samples);
int wsae_profile_disable(wsae_ctx* ctx);
int wsae_profile_read(wsae_ctx* ctx, int32_t kernel_id, int32_t* n_launches, double* total_ms);

/* ---- per-feature top activations (row N4: analysis/feature_viz.py:94-158, TopKTracker.update) -----------------
 * The consumer right after the path: for every feature keep the `keep` (<= 64) strongest positive activations seen
 * so far.  One call = one batch of `rows` activation rows, given either as the compact code the TopK kernel emits
 * (vals/idx [rows][width], width = k) or as a dense matrix (idx == NULL, vals [rows][H], width == H).  Row r of the
 * call is activation number ord_base + r; the caller maps ordinals to (sample, position).  State (caller-owned,
 * zero-initialised, device memory on the current device): top_vals [H][keep] f32 and top_ord [H][keep] i64, both
 * sorted (value descending, then ordinal ascending: the reference keeps the earlier of two equal values,
 * feature_viz.py:153), top_cnt [H] i32, total_active [1] i64 (+= number of entries > 0, feature_viz.py:136).
 * workspace: wsae_feature_topk_workspace_bytes(rows * width, H) bytes of scratch. */
int64_t wsae_feature_topk_workspace_bytes(int64_t max_entries, int32_t H);
int wsae_feature_topk_update(const float* vals, const int32_t* idx, int64_t rows, int32_t width, int32_t H,
                             int32_t keep, int64_t ord_base, float* top_vals, int64_t* top_ord,
                             int32_t* top_cnt, int64_t* total_active, void* workspace,
                             int64_t workspace_bytes, void* stream);

/* ---- ReLU SAE (model.py:260-322), dense path ------------------------------------------------- */
/* ReLUSAE has no pre-bias: pass the TopK pack [W_e | W_dT | b_e | b_d | b_pre] with b_pre = 0 (wsae_prepare
 * first, as for the TopK path).  forward:  hidden [B,H] f32 = relu(x W_e^T + b_e) (model.py:307),
 * recon [B,D] f32 = hidden W_d^T + b_d (:308), stats->loss = mse + sparsity_weight * mean|hidden| (:309-311),
 * stats->l0 (:313), stats->reserved = the float bits of mean|hidden|; *sparsity_loss_out likewise (may be
 * NULL, as may stats).  backward (must follow the forward of the same batch on the same ctx: it reuses the
 * staged x^T and hidden^T): grads in pack layout, dW_e, dW_dT, db_e, db_d as autograd of model.py:304-311
 * gives them, the b_pre slot set to 0; no dL/dx (the reference has none either).  Needs wsae_ctx_reserve_relu. */
/* configs[4] of BASELINE.json ("fp8 MFMA encode/decode"): on != 0 makes the two forward GEMMs of wsae_relu_forward take
 * OCP e4m3 copies of their operands (x and hidden quantised per batch row, W_e per feature row, W_d per output row;
 * q = e4m3(v * 448 / amax_row), v_mfma_f32_32x32x16_fp8_fp8, fp32 accumulate, dequantised in the GEMM epilogue).  BF16
 * mode only; hidden, loss and the whole backward stay on the bf16 path.  Needs batch >= 512, input_dim % 256 == 0,
 * hidden_dim % 256 == 0. */
int wsae_ctx_set_relu_fp8(wsae_ctx* ctx, int32_t on);
/* 1 when wsae_relu_forward / wsae_relu_backward at batch size B read / write the dense fp32 `hidden` buffer, 0 when it is
 * optional (bf16 mode, whole 128-row groups, input_dim a multiple of 128, hidden_dim of 256: the hidden code then lives as
 * bf16 in the ctx workspace, written by the encoder GEMM's epilogue, and `hidden` may be NULL - a trainer that never looks at
 * it saves 4 B H bytes of stores per step; a non-NULL `hidden` still receives the fp32 copy).  Where this returns 0, `recon` is
 * optional as well: the forward's residual pass already leaves g = 2 (recon - x) / (B cols) and the db_d partials for the
 * backward of the same batch, which then needs no recon (a non-NULL `recon` still receives the fp32 reconstruction). */
int wsae_relu_needs_hidden(const wsae_ctx* ctx, int32_t B);
/* Per-feature weights w[hidden_dim] of the L1 term (device memory, caller-owned, must outlive the calls; NULL = all ones, the
 * default): the sparsity term of wsae_relu_forward becomes sum_b sum_s w[s] |hidden[b][s]| / (B H) and wsae_relu_backward adds
 * sparsity_weight * w[s] / (B H) to dL/dhidden[b][s] (w itself is a constant of the step).  The cross-layer crosscoder
 * (crosscoder.py:213-217: decoder-norm-weighted L1, mean over the batch of the row sums) passes the decoder norms and
 * sparsity_weight * H.  The MSE of the ReLU path divides by B * loss_cols (wsae_ctx_set_loss_cols), as the TopK path does. */
int wsae_ctx_set_relu_l1_weights(wsae_ctx* ctx, const float* weights);
int wsae_relu_forward(wsae_ctx* ctx, const float* params, const void* x, int32_t x_dtype,
                      const int32_t* rows, int32_t B, float sparsity_weight, float* hidden,
                      float* recon, wsae_stats* stats, float* sparsity_loss_out, void* stream);
int wsae_relu_backward(wsae_ctx* ctx, const float* params, const void* x, int32_t x_dtype,
                       const int32_t* rows, int32_t B, float sparsity_weight, const float* hidden,
                       const float* recon, float* grads, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* WSAE_H_ */
