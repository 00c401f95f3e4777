/*
 * dir_hip.h -- C ABI of libdir_hip.so: the MI355X (gfx950) drop-in boundary for the
 * embedding-lookup + feature-interaction hot path of yinyajun/Details-In-Recommendation.
 *
 * Every entry point is extern "C", takes plain pointers and sizes, and is asynchronous on the
 * hipStream_t it is given (passed as void*).  The caller owns every buffer; the library never
 * allocates device memory and keeps no global mutable state besides a thread-local error string.
 * Return value: DIR_OK (0) or a negative DIR_E_* code; dir_last_error() describes the failure.
 *
 * Citations are relative to the reference tree (/root/reference).  "[TF-upstream]" marks TensorFlow
 * 1.x library behaviour that the reference calls but does not contain.
 *
 * All device pointers must be 16-byte aligned when the row width in bytes is a multiple of 16
 * (the kernels use 16-byte vector accesses on that path); otherwise 4-byte alignment suffices.
 * Row width limits of the gather / bag kernels (one wave covers a row): K <= 256 floats on the 16-byte path (K a multiple
 * of 4, 16-byte aligned tables and output, out_ld a multiple of 4), K <= 64 otherwise; wider rows return DIR_E_UNSUPPORTED.
 */
#ifndef DIR_HIP_H_
#define DIR_HIP_H_

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef void* dir_stream_t; /* hipStream_t */

#define DIR_VERSION 202 /* 0.2.1: + packed training rows (dir_gather_fm_rows_f32, dir_sparse_adagrad_sorted_rows_f32) */

enum {
    DIR_OK = 0,
    DIR_E_BADARG = -1,      /* null pointer, non-positive size, unsupported shape */
    DIR_E_RANGE = -2,       /* an id outside [0, vocab) found by dir_check_ids */
    DIR_E_HIP = -3,         /* a HIP runtime call failed */
    DIR_E_UNSUPPORTED = -4  /* shape outside what the kernels are built for */
};

/* combiner of an embedding bag ([TF-upstream] embedding_lookup_sparse combiner=) */
enum { DIR_COMBINER_SUM = 0, DIR_COMBINER_MEAN = 1, DIR_COMBINER_SQRTN = 2 };

/* flags for dir_embedding_bag_f32 */
enum {
    DIR_BAG_PRUNE_NONPOSITIVE_WEIGHTS = 1, /* drop entries with weight <= 0 ([TF-upstream]
                                              safe_embedding_lookup_sparse, later 1.x) */
    DIR_GATHER_STREAM_ROWS = 2             /* one-hot path: read table rows with non-temporal loads.  Right
                                              when ids are spread over tables far larger than the 256 MiB
                                              Infinity Cache (measured 70 -> 62 us at BASELINE config 2,
                                              uniform ids); wrong for skewed ids whose hot rows should stay
                                              cached (Zipf 1.05: 50 -> 60 us).  Values are identical. */
};

int dir_version(void);
const char* dir_last_error(void);

/* HOST: CRC-32C (Castagnoli) of n bytes, continuing from `crc` (0 to start) -- the checksum of [TF-upstream] checkpoint bundles
 * (tensor_bundle: per-tensor crc32c, table block trailers), used by tf_bundle.py to write / verify model_dir checkpoints
 * (models/DeepCrossNetwork/train.py:170-175, the Estimators' model_dir). */
uint32_t dir_crc32c(uint32_t crc, const void* data, int64_t n);

/* --------------------------------------------------------------------------------------------
 * A1/A2  multi-slot embedding bag.
 * Replaces: myself_input_layer                        models/DeepFM/deepFM.py:363-400
 *           column._get_dense_tensor(...)              models/DeepFM/deepFM.py:387-390
 *           tf.feature_column.input_layer(...)         models/DeepCrossNetwork/DeepCrossNetwork.py:126
 *           [TF-upstream] safe_embedding_lookup_sparse -> embedding_lookup_sparse
 *
 * tables   device array [F] of device pointers, slot f -> fp32 [vocab_f, K] row-major
 * ids      one-hot  (offsets == NULL): id of (sample b, slot f) = ids[b*stride_b + f*stride_f]
 *          multi-hot(offsets != NULL): bag(b,f) = b*stride_b + f*stride_f; its entries are
 *                   ids[offsets[bag] .. offsets[bag+1]) in order; offsets has B*F+1 entries
 * weights  NULL or per-entry fp32 weights (multi-hot only)
 * out      out[b*out_ld + f*K + k]; out_ld >= F*K  (concat of the slots in slot order)
 *
 * Semantics (A2): entries with id < 0 are dropped; an empty bag yields zeros; the bag is reduced in
 * entry order in fp32 (sum of w*row), then 'mean' divides by sum(w) (count when weights == NULL),
 * 'sqrtn' by sqrt(sum(w*w)) (sqrt(count)).  0 <= id < vocab_f is a precondition of this entry point (see dir_check_ids and
 * the vocab argument of dir_embedding_bag_ex_f32).
 * ------------------------------------------------------------------------------------------ */
int dir_embedding_bag_f32(const float* const* tables, int F, int K,
                          const int64_t* ids, const int64_t* offsets, const float* weights,
                          int64_t stride_b, int64_t stride_f, int combiner, int flags, int64_t B,
                          float* out, int64_t out_ld, dir_stream_t stream);

/* The same with the per-column attributes a tf.feature_column.embedding_column carries (constructor call sites
 * models/DeepCrossNetwork/train.py:99, dataset/SequenceTensorFlowDataset/test4.py:53-55):
 *   vocab          DEVICE int64 [F] or NULL.  With it an id >= vocab_f is pruned like id < 0 (what [TF-upstream] GPU lookups
 *                  return for an out-of-range id: zeros; the CPU kernels raise InvalidArgument -- the host mirror's
 *                  IdentityCategoricalColumn does that check).  NULL: 0 <= id < vocab_f stays a precondition.
 *   slot_combiner  DEVICE int32 [F] or NULL: one DIR_COMBINER_* per slot (every embedding_column has its own combiner=);
 *                  NULL: `combiner` for all slots.
 *   max_norm       > 0: [TF-upstream] embedding_lookup(max_norm=): every looked-up row is clipped to that l2 norm before it
 *                  is weighted -- row * max_norm / max(||row||, max_norm), ||row|| = sqrt(sum_k row_k^2) summed k-ascending
 *                  (clip_ops.clip_by_norm, r1.10+ form).  0: no clipping.  One-hot ids (offsets == NULL) are accepted too. */
int dir_embedding_bag_ex_f32(const float* const* tables, const int64_t* vocab, int F, int K,
                             const int64_t* ids, const int64_t* offsets, const float* weights,
                             int64_t stride_b, int64_t stride_f, const int32_t* slot_combiner, int combiner,
                             float max_norm, int flags, int64_t B, float* out, int64_t out_ld, dir_stream_t stream);

/* ... and with one max_norm PER SLOT (every embedding_column carries its own max_norm=, like its own combiner=):
 *   slot_max_norm  DEVICE fp32 [F] or NULL; entry f > 0 clips slot f's rows to that norm, 0 leaves them; NULL: `max_norm` for all. */
int dir_embedding_bag_ex2_f32(const float* const* tables, const int64_t* vocab, int F, int K,
                              const int64_t* ids, const int64_t* offsets, const float* weights,
                              int64_t stride_b, int64_t stride_f, const int32_t* slot_combiner, int combiner,
                              const float* slot_max_norm, float max_norm, int flags, int64_t B, float* out, int64_t out_ld,
                              dir_stream_t stream);

/* Validation of the precondition above (debug aid, asynchronous like everything else).
 * vocab: DEVICE array [F].  bad_count: DEVICE int32; zeroed on the stream, then set to the number of
 * ids >= vocab_f or < -1 found (-1 is the legal "missing" marker: pruned).  The caller reads it back and treats a
 * non-zero count as DIR_E_RANGE ([TF-upstream] CPU kernels raise InvalidArgument there). */
int dir_check_ids(const int64_t* vocab, int F, const int64_t* ids, const int64_t* offsets,
                  int64_t stride_b, int64_t stride_f, int64_t B, int32_t* bad_count,
                  dir_stream_t stream);

/* --------------------------------------------------------------------------------------------
 * A4  FM second-order term.
 * Replaces: fm_logit_fn                                 models/DeepFM/deepFM.py:321-335
 * emb[b*emb_ld + f*K + k] -> out[b] = 0.5 * sum_k( (sum_f e)^2 - sum_f e^2 )
 * Summation order is f ascending, then k ascending (deterministic; matches oracle/ bit for bit).
 * ------------------------------------------------------------------------------------------ */
int dir_fm_second_order_f32(const float* emb, int64_t emb_ld, int64_t B, int F, int K, float* out,
                            dir_stream_t stream);

/* A1+A4 fused for one-hot slots: one pass over the rows.  out may be NULL (FM only); fm may be NULL
 * (gather only).  flags: DIR_GATHER_STREAM_ROWS.  Replaces deepFM.py:169-177 (inputs) + :321-335
 * (fm_logit_fn) in one launch.  vocab: DEVICE int64 [F] or NULL, as in dir_embedding_bag_ex_f32 (an id >= vocab_f
 * contributes a zero row; one scalar compare per slot). */
int dir_gather_fm_fused_f32(const float* const* tables, const int64_t* vocab, int F, int K, const int64_t* ids,
                            int64_t stride_b, int64_t stride_f, int flags, int64_t B, float* out,
                            int64_t out_ld, float* fm, dir_stream_t stream);

/* Packed serving layout (an MI355X-specific option; the reference layout above stays the default): slot f's table is
 * [vocab_f, ld] fp32 with the embedding at columns [0, K) and the first-order weight of the same feature value at
 * column lin_col (-1: none).  With ld*4 = 128 B and 128-byte aligned tables a row is exactly one memory line, so
 * DeepFM's three sparse terms -- inputs (deepFM.py:169-177), fm_logit_fn (:321-335) and the linear term (:255-275)
 * over the same columns -- cost one line fetch per (sample, slot).  out / fm / lin_out may each be NULL;
 * lin_out[b] = sum_f w_f[id] (+ *bias), summed in slot order: bit-identical to dir_linear_sparse_sum_f32.
 * vocab: DEVICE int64 [F] or NULL (ids >= vocab_f pruned), as in dir_embedding_bag_ex_f32. */
int dir_gather_fm_linear_packed_f32(const float* const* tables, const int64_t* vocab, int F, int K, int64_t ld, int lin_col,
                                    const int64_t* ids, int64_t stride_b, int64_t stride_f, int flags,
                                    int64_t B, float* out, int64_t out_ld, float* fm, const float* bias,
                                    float* lin_out, dir_stream_t stream);

/* Packed TRAINING layout (MI355X-specific option): slot f's table is [vocab_f, ld] fp32 with the embedding at columns
 * [0, K) and -- by the convention of dir_sparse_adagrad_sorted_rows_f32 below -- its Adagrad accumulator at [K, 2K): with
 * K = 16, ld = 32 a row and its optimiser state are ONE 128-byte line, so the update issues one read and one write per touched
 * row instead of two each, and the forward's 64-byte row read costs what it did (a 64-byte miss occupies a 128-byte
 * DRAM-side slot anyway).  The same one-hot gather + fm_logit_fn (deepFM.py:169-177, :321-335) as
 * dir_gather_fm_fused_f32, values bit-identical, plus fsum [B, K] (may be NULL): S[b] = sum_f e[b,f] (f ascending, fp32) --
 * what the FM backward needs of the forward, so that the training graph keeps [B, K] instead of [B, F*K]. */
int dir_gather_fm_rows_f32(const float* const* tables, const int64_t* vocab, int F, int K, int64_t ld, const int64_t* ids,
                           int64_t stride_b, int64_t stride_f, int flags, int64_t B, float* out, int64_t out_ld,
                           float* fm, float* fsum, dir_stream_t stream);
/* dir_gather_fm_rows_f32 that ALSO leaves what a row-scaled fp16 x 2 layer reading `out` needs (the outputs of dir_row_absmax_bits_f32,
 * without that entry's pass over out): row_bits[b] = bit pattern of max_k |out[b, k]|, *all_bits = of max |out|.  workspace: as
 * dir_row_absmax_bits_f32's (dir_row_absmax_workspace_words() unsigned ints of device memory, the first one zero before the first call). */
int dir_gather_fm_rows_bits_f32(const float* const* tables, const int64_t* vocab, int F, int K, int64_t ld, const int64_t* ids,
                                int64_t stride_b, int64_t stride_f, int flags, int64_t B, float* out, int64_t out_ld,
                                float* fm, float* fsum, unsigned int* row_bits, unsigned int* all_bits, unsigned int* workspace,
                                dir_stream_t stream);

/* --------------------------------------------------------------------------------------------
 * A6  first-order (linear) term, units = 1.
 * Replaces: _linear_logit_fn_builder                    models/DeepFM/deepFM.py:255-275
 *           [TF-upstream] feature_column.linear_model(..., sparse_combiner=)
 * weights  device array [F] of device pointers, slot f -> fp32 [vocab_f]
 * ids/offsets/entry_weights/strides as in dir_embedding_bag_f32; vocab: DEVICE int64 [F] or NULL (ids >= vocab_f pruned).
 * out[b] = (accumulate ? out[b] : 0) + (bias ? *bias : 0) + sum_f combine_f(bag(b,f))
 * ------------------------------------------------------------------------------------------ */
int dir_linear_sparse_sum_f32(const float* const* weights, const int64_t* vocab, int F, const int64_t* ids,
                              const int64_t* offsets, const float* entry_weights, int64_t stride_b,
                              int64_t stride_f, int combiner, const float* bias, int accumulate,
                              int64_t B, float* out, dir_stream_t stream);

/* --------------------------------------------------------------------------------------------
 * A8  DCN cross network, all L layers in one launch.
 * Replaces: _cross_op / _cross_architecture   models/DeepCrossNetwork/DeepCrossNetwork.py:336-367
 * x0 [B, d] (row stride x_ld), w,b [L, d] contiguous; out [B, d] (row stride out_ld).
 * x_{l+1} = ((x0 * (x_l . w_l)) + b_l) + x_l     (evaluation order of DeepCrossNetwork.py:346)
 * Limits: L*d*8 B <= 64 KiB (LDS weight image); d <= 4096 on the 16-byte path (d and the row strides multiples of 4,
 * 16-byte aligned pointers), d <= 1024 otherwise (DIR_E_UNSUPPORTED beyond; the same limits hold for the backward).
 * ------------------------------------------------------------------------------------------ */
int dir_dcn_cross_f32(const float* x0, int64_t x_ld, const float* w, const float* b, int L,
                      int64_t B, int d, float* out, int64_t out_ld, dir_stream_t stream);

/* One cross layer on a distinct x, the literal _cross_op(x0, x, w, b) of DeepCrossNetwork.py:336-347:
 * out = ((x0 * (x . w)) + b) + x;  x0 and x share the row stride x_ld; w, b are [d]. */
int dir_dcn_cross_op_f32(const float* x0, const float* x, int64_t x_ld, const float* w, const float* b,
                         int64_t B, int d, float* out, int64_t out_ld, dir_stream_t stream);

/* dir_dcn_cross_f32 followed by the cross branch's share of the final dense(1) over concat([cross, deep])
 * (DeepCrossNetwork.py:136-137): head_out[r] = x_L[r] . head_w  (head_w DEVICE [d], head_out DEVICE [B]); out may be NULL -- the
 * cross output then never reaches memory (inference: it has no other reader). */
int dir_dcn_cross_head_f32(const float* x0, int64_t x_ld, const float* w, const float* b, int L, int64_t B, int d,
                           const float* head_w, float* out, int64_t out_ld, float* head_out, dir_stream_t stream);

/* --------------------------------------------------------------------------------------------
 * A13 DIN local activation unit + weighted pooling (no reference code: README.md:27 links
 * arXiv:1706.06978; definition in DESIGN.md / oracle).
 * table [vocab, K]; hist [B, T] int64 (entries j >= hist_len[b] or id < 0 are masked);
 * cand [B] int64.  For every valid j:  h = table[hist[b,j]], a = table[cand[b]],
 *   z1 = sigmoid(W1^T [h, a, h-a, h*a] + b1)   W1 [4K, H1]
 *   z2 = sigmoid(W2^T z1 + b2)                 W2 [H1, H2]
 *   s_j = W3 . z2 + b3                         W3 [H2]
 * normalize = 0: w_j = s_j (paper); normalize = 1: w = softmax over valid j of s_j / sqrt(K).
 * out[b, :] = sum_j w_j h_j  (zeros when no valid j).  scores (optional) [B, T] receives w_j
 * (0 for masked j).
 * ------------------------------------------------------------------------------------------ */
int dir_din_attention_pool_f32(const float* table, int K, const int64_t* hist,
                               const int32_t* hist_len, const int64_t* cand, int T,
                               const float* W1, const float* b1, int H1, const float* W2,
                               const float* b2, int H2, const float* W3, const float* b3,
                               int normalize, int64_t B, float* out, float* scores,
                               dir_stream_t stream);

/* The same unit with the paper's own hidden activations (arXiv:1706.06978 section 5.3; no reference code: README.md:27) in place of the
 * two sigmoids:  activation = DIR_DIN_ACT_PRELU:  f(s) = s > 0 ? s : alpha s;
 *                activation = DIR_DIN_ACT_DICE:   f(s) = p s + (1 - p) alpha s,  p = sigmoid(scale s + shift)  -- the inference form of
 *                Dice: scale = 1 / sqrt(Var[s] + eps), shift = -E[s] scale with the layer's moving statistics;
 *                activation = DIR_DIN_ACT_SIGMOID forwards to dir_din_attention_pool_f32.
 * act_params: DEVICE float [3 H1 + 3 H2] = alpha1 [H1], scale1 [H1], shift1 [H1], alpha2 [H2], scale2 [H2], shift2 [H2] (PReLU reads
 * the alphas only).  Inference only; K = 64, H1 <= 80, H2 <= 48 (multiples of 4), T <= 64 (DIR_E_UNSUPPORTED otherwise). */
enum { DIR_DIN_ACT_SIGMOID = 0, DIR_DIN_ACT_PRELU = 1, DIR_DIN_ACT_DICE = 2 };
/* PReLU / Dice (the same inference forms) over a [B, N] activation IN PLACE, rows ld floats apart: the hidden layers of DIN's 200-80 MLP.
 * alpha, scale, shift: DEVICE [N] (PReLU: scale / shift may be NULL).  N and ld multiples of 4, 16-byte aligned. */
int dir_din_activation_rows_f32(float* x, int64_t ld, int64_t B, int N, int activation, const float* alpha, const float* scale,
                                const float* shift, dir_stream_t stream);
int dir_din_attention_pool_act_f32(const float* table, int K, const int64_t* hist, const int32_t* hist_len, const int64_t* cand, int T,
                                   const float* W1, const float* b1, int H1, const float* W2, const float* b2, int H2, const float* W3,
                                   const float* b3, int normalize, int activation, const float* act_params, int64_t B, float* out,
                                   float* scores, dir_stream_t stream);

/* --------------------------------------------------------------------------------------------
 * A14 xDeepFM compressed interaction network, one layer (no reference code: README.md:28 links
 * arXiv:1803.05170).
 * x0 [B, m, D], xk [B, Hp, D], W [H, Hp*m] (column index i*m + j) ->
 *   xout[b, h, d] = sum_{i<Hp} sum_{j<m} W[h, i*m+j] * xk[b,i,d] * x0[b,j,d]
 *   pooled[b*pooled_ld + h] = sum_d xout[b,h,d]      (pooled may be NULL; xout may be NULL when
 *                                                     pooled is given: the last layer of a stack)
 * The outer product Z is never materialised: it is formed in registers and fed to fp32 MFMA.
 * Shapes: D in {4, 8, 16, 32}; any m <= 40 (instantiated for padded field counts 8/16/26/40); any Hp, H.
 * ------------------------------------------------------------------------------------------ */
int dir_cin_layer_f32(const float* x0, const float* xk, const float* W, int m, int Hp, int H, int D,
                      int64_t B, float* xout, float* pooled, int64_t pooled_ld,
                      dir_stream_t stream);
/* The same layer on the bf16 matrix pipe with fp32-equivalent arithmetic (csrc/cin_bf3.hip).  Evaluated as
 *   xout = sum_j x0[.,j] * T_j,  T_j[r,h] = sum_i xk[r,i] * W[h, i*m+j]:
 * xk and W are each split into three bf16 pieces (round to nearest; the pieces sum to the operand exactly, fp32 exponent range), the
 * six piece products of weight >= 2^-16 are accumulated in fp32 by v_mfma_f32_16x16x32_bf16, and the field factor is one fp32 fma
 * per accumulator and field.  Same 1e-5 bar against the double-accumulating oracle as dir_cin_layer_f32 (measured error 2-3e-7),
 * half its time on 128-wide layers.  Results are not bitwise those of dir_cin_layer_f32 (different summation tree).
 * workspace: dir_cin_bf16x3_workspace_bytes(m, Hp, H) device bytes, 16-byte aligned (the packed bf16 image of W, rebuilt by
 * every call).  Shapes: D in {4, 8, 16, 32}; m <= 40; any Hp, H (columns are computed in blocks of 128 plus one last block of
 * 32 / 64 / 96 / 128, i in blocks of 32 (Hp <= 32) or 64: narrow layers are better served by dir_cin_layer_f32). */
int64_t dir_cin_bf16x3_workspace_bytes(int m, int Hp, int H);
int dir_cin_layer_bf16x3_f32(const float* x0, const float* xk, const float* W, int m, int Hp, int H, int D,
                             int64_t B, float* xout, float* pooled, int64_t pooled_ld,
                             void* workspace, int64_t workspace_bytes, dir_stream_t stream);
/* The LAST layer of a stack, whose map feeds nothing but its pooled sums (csrc/cin_pool.hip): the sum over d commutes with the contraction,
 *   pooled[b,h] = sum_{i,j} W[h,i,j] Z[b,i,j],   Z[b, i*m + j] = sum_d xk[b,i,d] x0[b,j,d]   -- 1/D of the layer's matrix work;
 * dir_cin_pool_z_f32 forms Z [B, Hp*m] (the pooled sums are then dir_dense_* on (Z, W [H, Hp*m])).  Backward given g = dL/dpooled [B, H]:
 * dW = g^T Z (dir_dense_dw_*), dZ = g W (dir_dense_*), and dir_cin_pool_dx_f32: dxk[b,i,d] = sum_j dZ[b,i,j] x0[b,j,d] (+ add_pooled[b,i],
 * NULL: none -- the pooled gradient of the layer below), dx0[b,j,d] (+)= sum_i dZ[b,i,j] xk[b,i,d] (accumulate_dx0).
 * m <= 64, D in {4, 8, 16, 32}; x0 / xk / dxk / dx0 16-byte aligned. */
int dir_cin_pool_z_f32(const float* x0, const float* xk, int m, int Hp, int D, int64_t B, float* Z, dir_stream_t stream);
/* ... and z_row_bits [B]: the bit pattern of max |Z[b, :]| (one workgroup owns a sample's row: plain stores) -- the row scales
 * dir_dense_f16x2_rows_f32 multiplies Z by, without a max pass over the [B, Hp*m] matrix. */
int dir_cin_pool_z_bits_f32(const float* x0, const float* xk, int m, int Hp, int D, int64_t B, float* Z, unsigned int* z_row_bits, dir_stream_t stream);
/* The pooled last layer FUSED (round 6; inference, where nothing else reads Z): pooled[b,h] = sum_{i,j} W[h,i,j] sum_d xk[b,i,d] x0[b,j,d] in one
 * launch, Z formed in registers and fed to the bf16 matrix pipe as three bf16 pieces (fp32's exponent range: no row maxima needed; the sums
 * over d in cin_pool_z_k's order) -- the 872 MB of Z at the BASELINE shape are neither written nor read (0.64 -> 0.2 ms for the layer).
 *   image: dir_cin_pooled_image_bytes(m, Hp, H, D) device bytes, 16-byte aligned, built by dir_cin_pooled_pack_f32 from W [H, Hp * m] once per
 *          weight version (the weights of one input channel as [H / 16 tiles][3 pieces] of MFMA operands, streamed through LDS).
 * Covers D = 16, m <= 32, H <= 128 (a multiple of 4), any Hp (dir_cin_pooled_image_bytes returns 0 / DIR_E_UNSUPPORTED otherwise); pooled rows
 * pooled_ld floats apart (a multiple of 4), 16-byte aligned.  No reference code (README.md:28 -> arXiv:1803.05170). */
int64_t dir_cin_pooled_image_bytes(int m, int Hp, int H, int D);
int dir_cin_pooled_pack_f32(const float* W, int m, int Hp, int H, int D, void* image, int64_t image_bytes, dir_stream_t stream);
int dir_cin_pooled_last_bf16x3_f32(const float* x0, const float* xk, const void* image, int m, int Hp, int H, int D, int64_t B, float* pooled,
                                   int64_t pooled_ld, dir_stream_t stream);
int dir_cin_pool_dx_f32(const float* x0, const float* xk, const float* dZ, int m, int Hp, int D, int64_t B, const float* add_pooled,
                        int64_t add_pooled_ld, float* dxk, float* dx0, int accumulate_dx0, dir_stream_t stream);

/* The FIRST layer of a stack (xk = x0, Hp = m) on the same recipe, over the m (m + 1) / 2 unordered field pairs: xout[b,h,d] =
 * sum_{i<=j} (W[h,i,j] + W[h,j,i] | W[h,i,i]) x0[b,i,d] x0[b,j,d] -- the pair products are formed and split inside the kernel, 351 reduction
 * slots instead of 26 x 32 at m = 26 (csrc/cin_bf3.hip, PAIRS).  8 <= m <= 40, D in {4, 8, 16, 32}; workspace:
 * dir_cin_layer1_bf16x3_workspace_bytes(m, H) bytes, 256-byte aligned.  Same outputs as dir_cin_layer_bf16x3_f32(x0, x0, W, ...) up to the
 * summation order (the same 1e-5 bar), bitwise reproducible. */
int64_t dir_cin_layer1_bf16x3_workspace_bytes(int m, int H);
int dir_cin_layer1_bf16x3_f32(const float* x0, const float* W, int m, int H, int D, int64_t B, float* xout, float* pooled, int64_t pooled_ld,
                              void* workspace, int64_t workspace_bytes, dir_stream_t stream);

/* The same two forward layers on "fp16 x 2" arithmetic (csrc/cin_bf3.hip, round 4): every fp32 operand as two fp16 pieces by
 * round-to-nearest, the three piece products of weight >= 2^-11 on v_mfma_f32_16x16x32_f16 with fp32 accumulation -- half the matrix
 * instructions of bf16 x 3.  Same arguments, shapes and workspaces (dir_cin_bf16x3_workspace_bytes /
 * dir_cin_layer1_bf16x3_workspace_bytes) as the bf16 x 3 entries.
 * dir_cin_layer_f16x2_f32 is the UNSCALED form: preconditions |x0|, |xk|, |W| < 65 504 (fp16's range; larger values become inf), and
 * operand elements below 2^-3 in magnitude carry an ABSOLUTE representation error of up to 2^-25 each (relative 2^-22 above); on
 * embedding-scale operands within 3-6e-7 (scaled) of the double-accumulating oracle.  Kept for A/B; nothing routes to it by default.
 * dir_cin_layer1_f16x2_f32 (round 5) scales BOTH operands by exact powers of two inside the launch: every row r = (b, d) of the x0 slice so
 * that its largest |element| lands in [2^6, 2^7) (pair products below 2^14), the symmetrised pair weights as a tensor into [2^14, 2^15);
 * 2^-(2k + kw) is taken out where the products meet the accumulators.  No precondition on magnitudes: scaling x0 by 2^a and W by 2^b
 * scales the result by exactly 2^(2a + b), bit for bit (tests/test_gpu_range.py).
 * dir_cin_layer1_bits_f16x2_f32: the same, and xout_row_bits [B * D] (row r = b * D + d) receives the bit pattern of max_h |xout[b, h, d]| --
 * the next layer's row scales (needs xout and H <= 128: one column block). */
int dir_cin_layer_f16x2_f32(const float* x0, const float* xk, const float* W, int m, int Hp, int H, int D, int64_t B, float* xout,
                            float* pooled, int64_t pooled_ld, void* workspace, int64_t workspace_bytes, dir_stream_t stream);
int dir_cin_layer1_f16x2_f32(const float* x0, const float* W, int m, int H, int D, int64_t B, float* xout, float* pooled, int64_t pooled_ld,
                             void* workspace, int64_t workspace_bytes, dir_stream_t stream);
int dir_cin_layer1_bits_f16x2_f32(const float* x0, const float* W, int m, int H, int D, int64_t B, float* xout, float* pooled, int64_t pooled_ld,
                                  void* workspace, int64_t workspace_bytes, unsigned int* xout_row_bits, dir_stream_t stream);
/* The same kernel with a second result per field, for the layer's data gradients (y's tile in registers; 64-column blocks):
 *   xout[b,h,d] as above, and  dot[b,j,d] = sum_h y[b,h,d] * T_j[(b,d),h],  T_j[r,h] = sum_i xk[r,i] * W[h, i*m+j]
 * written as dir_cin_bf16x3_dot_partials(m, Hp, H) partial sums [P][B, m, D] (one per 64-column block and half of i; the caller adds
 * them: a fixed order, no atomics).  Backward of xout = CIN(x0, xk_l, W) given G = dL/dxout: call with xk := G [B,H,D],
 * W := W1 [Hp_l, H*m] (W1[i, h*m+j] = W[h, i*m+j]), y := xk_l [B,Hp_l,D] -- then xout is dL/dxk_l and the partials add up to
 * dL/dx0.  Same shape limits and workspace as dir_cin_layer_bf16x3_f32. */
int dir_cin_bf16x3_dot_partials(int m, int Hp, int H);
int dir_cin_layer_dot_bf16x3_f32(const float* x0, const float* xk, const float* W, const float* y, int m, int Hp, int H, int D,
                                 int64_t B, float* xout, float* dot_partials, void* workspace, int64_t workspace_bytes,
                                 dir_stream_t stream);
/* dir_cin_layer_dot_bf16x3_f32 with add_pooled [B, H] (row stride add_pooled_ld; NULL: none) added to xout[b, h, :] in the epilogue: in the
 * backward of a stack xout is dL/dxk of layer k+1 and add_pooled the pooled gradient of layer k -- their sum is layer k's dL/dxout, formed
 * without a pass of its own.  dir_sum_partials_f32: out[e] (+)= sum_p parts[p][e] in p order (n % 4 == 0, 16-byte aligned): the dot
 * partials into dx0, accumulated over the layers of a stack. */
int dir_cin_layer_dot_add_bf16x3_f32(const float* x0, const float* xk, const float* W, const float* y, int m, int Hp, int H, int D, int64_t B,
                                     const float* add_pooled, int64_t add_pooled_ld, float* xout, float* dot_partials, void* workspace,
                                     int64_t workspace_bytes, dir_stream_t stream);
int dir_sum_partials_f32(const float* parts, int P, int64_t n, int accumulate, float* out, dir_stream_t stream);
/* fp16 x 2 with a left operand xk of UNKNOWN magnitude (the backward: xk := G).  Inside the kernel every row r = (b, d) of xk is multiplied
 * by a power of two chosen from that row's largest |element| over its Hp channels (it lands in [2^14, 2^15): the scaling is exact), split
 * into two fp16 pieces, and the inverse power of two is applied where the row's T tile meets the field factor (and y, in the dot form) --
 * again exact.  Elements within 2^-17 of their row's largest carry 22 bits; smaller ones an absolute error of 2^-39 of the row's largest,
 * which is what the sum over the row's channels needs.  Since round 5 W is scaled as well -- ONE power of two for the tensor (its largest
 * |element| into [2^14, 2^15), found by a 32-workgroup max kernel in front of the pack; the inverse leaves with the rows' scales) -- so
 * neither xk nor W has a magnitude precondition; x0 and y meet the products in fp32 and never had one.
 * dir_cin_layer_rows_f16x2_f32: the FORWARD layer in this form (what ops.cin_layer's "auto" runs for every layer but the first): arguments
 * of dir_cin_layer_f16x2_f32 plus xk_row_bits (optional, [B * D]: the bit pattern of max_i |xk[b, i, d]| left by the layer that produced xk
 * -- the prologue then reads one word per row instead of scanning it) and xout_row_bits (optional, [B * D]: the same of this layer's output;
 * needs xout and H <= 128).
 * dir_cin_layer_grad_f16x2_f32: arguments of dir_cin_layer_f16x2_f32 (the forward-form contractions of the backward) plus
 * xk_absmax_bits_out (as below; NULL: not wanted);
 * dir_cin_layer_dot_add_f16x2_f32: arguments, partial-sum layout (dir_cin_bf16x3_dot_partials) and workspace of
 * dir_cin_layer_dot_add_bf16x3_f32, plus xk_absmax_bits_out (DEVICE, one unsigned; NULL: not wanted): the bit pattern of max |xk| over the
 * whole tensor, a by-product of the kernel's row maxima -- what dir_cin_dw_f16x2_f32 takes as g_absmax_bits. */
int dir_cin_layer_grad_f16x2_f32(const float* x0, const float* xk, const float* W, int m, int Hp, int H, int D, int64_t B, float* xout,
                                 float* pooled, int64_t pooled_ld, void* workspace, int64_t workspace_bytes, unsigned int* xk_absmax_bits_out,
                                 dir_stream_t stream);
int dir_cin_layer_rows_f16x2_f32(const float* x0, const float* xk, const float* W, int m, int Hp, int H, int D, int64_t B, float* xout,
                                 float* pooled, int64_t pooled_ld, void* workspace, int64_t workspace_bytes, const unsigned int* xk_row_bits,
                                 unsigned int* xout_row_bits, dir_stream_t stream);
/* dir_cin_layer_auto_f16x2_f32: the forward layer with a DEVICE-SIDE verdict between the plain fp16 x 2 kernel (dir_cin_layer_f16x2_f32's) and
 * the row-scaled one: xk_row_bits (required: the producer's row maxima, dir_cin_layer1_bits_f16x2_f32 / xout_row_bits of these entries) and W are
 * reduced to their largest magnitudes by two small kernels; with max |xk| in [2^-4, 2^15) -- where splitting the rows unscaled loses nothing
 * that matters -- the plain kernel does the work (8-9 % faster: no scan of the rows), outside it the row-scaled kernel; W carries its tensor
 * scale in both; both kernels are launched and the one the verdict does not name leaves at once.  No host read: capturable in a HIP graph.
 * Results are those of whichever kernel ran (each bitwise reproducible). */
int dir_cin_layer_auto_f16x2_f32(const float* x0, const float* xk, const float* W, int m, int Hp, int H, int D, int64_t B, float* xout,
                                 float* pooled, int64_t pooled_ld, void* workspace, int64_t workspace_bytes, const unsigned int* xk_row_bits,
                                 unsigned int* xout_row_bits, dir_stream_t stream);

/* The CIN forward of an inference stack with x0 READ THROUGH INVERSE POSITIONS (round 6: the row-sharded lookup WITHOUT its finish pass --
 * ShardedTables.lookup_rows; precedent for the sharding: /root/reference/models/DeepFM/deepFM.py:163-167, the CIN itself README.md:28).
 *   x0_rows [n, D] fp32: the rows as the exchange left them (any order);  x0_inv [B, m] int64: the position of (sample, field)'s row in
 *   x0_rows, < 0 = a zero row (a pruned / out-of-range id).
 * Only the staging of x0 differs from the entries above (the [B, m*D] concatenation is never written or read); results are bit for bit
 * those of dir_cin_layer1_bits_f16x2_f32 / dir_cin_layer_auto_f16x2_f32 (xk_row_bits given) / dir_cin_layer_rows_f16x2_f32 (xk_row_bits
 * null) / dir_cin_pooled_last_bf16x3_f32 on the materialised x0.  Same workspaces, coverage and errors as those entries; a null x0_inv
 * with B > 0 is DIR_E_BADARG. */
int dir_cin_layer1_f16x2_gather_f32(const float* x0_rows, const int64_t* x0_inv, const float* W, int m, int H, int D, int64_t B, float* xout,
                                    float* pooled, int64_t pooled_ld, void* workspace, int64_t workspace_bytes,
                                    unsigned int* xout_row_bits /* nullable */, dir_stream_t stream);
int dir_cin_layer_f16x2_gather_f32(const float* x0_rows, const int64_t* x0_inv, const float* xk, const float* W, int m, int Hp, int H, int D,
                                   int64_t B, float* xout, float* pooled, int64_t pooled_ld, void* workspace, int64_t workspace_bytes,
                                   const unsigned int* xk_row_bits /* nullable */, unsigned int* xout_row_bits /* nullable */,
                                   dir_stream_t stream);
int dir_cin_pooled_last_bf16x3_gather_f32(const float* x0_rows, const int64_t* x0_inv, const float* xk, const void* image, int m, int Hp, int H,
                                          int D, int64_t B, float* pooled, int64_t pooled_ld, dir_stream_t stream);
int dir_cin_layer_dot_add_f16x2_f32(const float* x0, const float* xk, const float* W, const float* y, int m, int Hp, int H, int D, int64_t B,
                                    const float* add_pooled, int64_t add_pooled_ld, float* xout, float* dot_partials, void* workspace,
                                    int64_t workspace_bytes, unsigned int* xk_absmax_bits_out, dir_stream_t stream);

/* --------------------------------------------------------------------------------------------
 * A3  categorical id paths.
 * dir_hash_bucket_fast: HOST function. [TF-upstream] string_to_hash_bucket_fast =
 *   FarmHash Fingerprint64(bytes) mod num_buckets; call site categorical_column_with_hash_bucket
 *   models/DeepCrossNetwork/train.py:85-86.  strs: n pointers, lens: n byte lengths.
 * dir_fingerprint64: HOST, the raw 64-bit fingerprint.
 * dir_hash_bucket_i64_device: device kernel; ids are formatted as decimal strings (what
 *   [TF-upstream] as_string does for integer keys) and hashed the same way.
 * dir_bucketize_f32: device kernel. [TF-upstream] bucketized_column: out = number of boundaries
 *   <= x (boundaries ascending, nb of them); docstring models/DeepFM/deepFM.py:95.
 * ------------------------------------------------------------------------------------------ */
uint64_t dir_fingerprint64(const char* s, int64_t len);
int dir_hash_bucket_fast(const char* const* strs, const int64_t* lens, int64_t n,
                         int64_t num_buckets, int64_t* out);
int dir_hash_bucket_i64_device(const int64_t* keys, int64_t n, int64_t num_buckets, int64_t* out,
                               dir_stream_t stream);
/* the same over a flattened [.., F] key array with one bucket count per field (buckets_f: DEVICE [F]):
 * raw Criteo-style categorical keys -> the [B, F] id matrix the gather takes, in one launch */
int dir_hash_bucket_i64_fields_device(const int64_t* keys, int64_t n, const int64_t* buckets_f, int F,
                                      int64_t* out, dir_stream_t stream);
/* byte strings on the device: string i = bytes[offsets[i] .. offsets[i+1]); every FarmHash length branch */
int dir_hash_bucket_bytes_device(const char* bytes, const int64_t* offsets, int64_t n, int64_t num_buckets,
                                 int64_t* out, dir_stream_t stream);
int dir_bucketize_f32(const float* x, int64_t n, const float* boundaries, int nb, int64_t* out,
                      dir_stream_t stream);

/* --------------------------------------------------------------------------------------------
 * A12 row sharding, TF 'div' rule.
 * Replaces: min_max_variable_partitioner + partition_strategy='div'
 *           models/DeepFM/deepFM.py:163-167
 * dir_shard_div_owner: HOST helper; owner and local row of one id for vocab rows over P shards.
 * dir_shard_route: device kernel.  ids is a flattened [.., F] array (field of entry i is i % F);
 *   vocab: DEVICE array [F].  owner[i], local[i] of every id (ids < 0: owner = i % P, local = -1).
 * dir_gather_rows_f32: the owner-side lookup of the sharded path: out[i, :] = tables[slot[i]][row[i], :]
 *   (row < 0 -> zeros; slot == NULL -> table 0).  tables: device array of device pointers.
 * ------------------------------------------------------------------------------------------ */
void dir_shard_div_owner(int64_t id, int64_t vocab, int P, int* owner, int64_t* local);
int dir_shard_route(const int64_t* ids, int64_t n, const int64_t* vocab, int F, int P, int32_t* owner,
                    int64_t* local, dir_stream_t stream);
int dir_gather_rows_f32(const float* const* tables, int K, const int32_t* slot, const int64_t* row,
                        int64_t n, float* out, dir_stream_t stream);

/* Requester side of the sharded lookup in one call: route every id of the flattened [.., F] array and
 * counting-sort the entries by owner (P <= 64).
 *   payload[dst] = local_row * F + slot   (-1 for a pruned id), grouped by owner, owner o at
 *                  [starts[o], starts[o] + counts[o]);  inv[i] = dst of entry i;
 *   counts, starts: DEVICE int64 [P];  workspace: dir_shard_bucket_workspace_bytes(n, P) device bytes.
 * dir_gather_packed_f32 is the owner side for that payload: out[i,:] = tables[p % F][p / F, :]. */
int64_t dir_shard_bucket_workspace_bytes(int64_t n, int P);
int dir_shard_bucket(const int64_t* ids, int64_t n, const int64_t* vocab, const int32_t* parts, const int32_t* first, int F, int P,
                     int64_t* payload, int64_t* inv, int64_t* counts, int64_t* starts, void* workspace, dir_stream_t stream);

/* Partitioning of one table, as the reference's partitioner decides it (models/DeepFM/deepFM.py:163-167:
 * min_max_variable_partitioner(max_partitions=num_ps_replicas, min_slice_size=64 << 20)): table f is cut into parts[f] <= P
 * contiguous 'div' row slices and slice j lives on rank (first[f] + j) % P.  parts / first: DEVICE int32 [F] or NULL
 * (= every table cut P ways starting at rank 0, what the routines above did before).  Ids outside [0, vocab_f) have no owner:
 * they are treated as pruned.
 *
 * Fixed-capacity form of the requester side -- no split sizes ever reach the host, so a lookup is one stream-ordered chain
 * that can be captured in a graph and pipelined in micro-batches:
 *   payload  [P * (cap + 1)]: owner o's SLAB = one header word (number of valid slots, <= cap) + cap slots local_row*F + slot
 *   inv      [n]: row of entry i in the [P*cap, K] buffer the row exchange returns (o*cap + position), -1 for a pruned id or
 *            an entry that did not fit its slab
 *   counts   [P] int64: the true per-owner demand (can exceed cap);  overflow [1] int32: 1 iff some count > cap -- the caller
 *            then repeats the lookup on the variable-size path (dir_shard_bucket);  stat (optional, may be NULL): int64 [2] =
 *            {overflow, max_o counts[o]} -- the two numbers a caller MAX-reduces over micro-batches and ranks
 *   workspace: dir_shard_bucket_cap_workspace_bytes(P) bytes, ZERO before the first call (left zero on return).
 * Both exchanges are all-to-alls with EQUAL splits (cap + 1 words / cap rows per peer).
 * dir_gather_slabs_f32 is the owner side: recv = the P slabs as received; out[(s*cap + j), :] = the row of slot j of slab s for
 * j < header_s; the other rows are left untouched (never read by the requester).  flags: DIR_GATHER_STREAM_ROWS,
 * DIR_SLAB_SANITIZE (overwrite the slots behind each header with -1: the slab can then be walked as a flat payload by
 * dir_sparse_adagrad_sorted_payload_f32 -- the owner side of a sharded backward).
 *
 * Header word: low 32 bits = valid slots; high 32 bits = the SENDER's largest per-owner demand of that micro-batch, so the id
 * exchange itself tells every rank every other rank's demand.  dir_shard_slab_stat reads the headers of n_slabs RECEIVED slabs
 * (micro-batches x P senders, cap + 1 words apart) -> stat int64 [2] = {1 iff some demand > cap, the largest demand}: the same
 * verdict on every rank without a collective of its own.
 *
 * dir_shard_bucket_cap_dedup: the same requester side with duplicates removed before they travel ([TF-upstream]
 * embedding_lookup_sparse gathers unique ids; reference call sites models/DeepFM/deepFM.py:387, partitioner :163-167).  ids is a
 * [B, F] array with element strides (stride_b, stride_f); one workgroup de-duplicates a tile of 2048 / 4096 samples of one slot
 * through an LDS hash table (duplicates in different tiles travel once per tile: same result, a few more rows than an exact
 * unique).  inv is written FIELD-MAJOR: inv[f * B + b]; counts / overflow / stat / headers describe the de-duplicated demand.
 * Needs P * cap < 2^31. */
enum { DIR_SLAB_SANITIZE = 4 };
int64_t dir_shard_bucket_cap_workspace_bytes(int P);
int dir_shard_bucket_cap(const int64_t* ids, int64_t n, const int64_t* vocab, const int32_t* parts, const int32_t* first, int F, int P,
                         int64_t cap, int64_t* payload, int64_t* inv, int64_t* counts, int32_t* overflow, int64_t* stat,
                         void* workspace, dir_stream_t stream);
int dir_shard_bucket_cap_dedup(const int64_t* ids, int64_t stride_b, int64_t stride_f, int64_t B, const int64_t* vocab,
                               const int32_t* parts, const int32_t* first, int F, int P, int64_t cap, int64_t* payload, int64_t* inv,
                               int64_t* counts, int32_t* overflow, int64_t* stat, void* workspace, dir_stream_t stream);
int dir_shard_slab_stat(const int64_t* recv, int n_slabs, int64_t cap, int64_t* stat, dir_stream_t stream);
int dir_gather_slabs_f32(const float* const* tables, int F, int K, int64_t* recv, int P, int64_t cap, int flags, float* out,
                         dir_stream_t stream);
int dir_gather_packed_f32(const float* const* tables, int F, int K, const int64_t* payload, int64_t n,
                          int flags /* DIR_GATHER_STREAM_ROWS */, float* out, dir_stream_t stream);

/* --------------------------------------------------------------------------------------------
 * A5 / A9  hidden layers of the DNN towers: Y[M, N] = act(X[M, Kd] . Wt[N, Kd]^T + bias[N])   (row strides x_ld, w_ld, y_ld).
 *   reference: dnn_logit_fn, models/DeepFM/deepFM.py:295-300; _deep_architecture,
 *              models/DeepCrossNetwork/DeepCrossNetwork.py:394-399; _base_model, models/ESMM/ESMM.py:139-142
 *              ([TF-upstream] tf.layers.dense = matmul + bias + activation; Wt is the transpose of the TF kernel)
 * fp32 MFMA, bias and activation applied to the accumulators (one pass over Y).  bias may be NULL.
 * Limits: Kd, x_ld and w_ld multiples of 4, X / Wt 16-byte aligned (DIR_E_UNSUPPORTED otherwise: use a library GEMM).
 * ------------------------------------------------------------------------------------------ */
#define DIR_ACT_NONE 0
#define DIR_ACT_RELU 1
int dir_dense_f32(const float* X, int64_t x_ld, const float* Wt, int64_t w_ld, const float* bias, int act, int64_t M, int Kd, int N,
                  float* Y, int64_t y_ld, dir_stream_t stream);
/* dir_dense_affine_f32: Y = act(X . Wt^T + bias) * post_scale[N] + post_shift[N] -- the layer followed by its INFERENCE batch
 * normalisation (models/DeepFM/deepFM.py:303-308, DeepCrossNetwork.py:400-403: the reference normalises AFTER the activation),
 * folded to one per-column affine: post_scale = gamma * rsqrt(moving_variance + eps), post_shift = beta - moving_mean * post_scale
 * (gamma = 1 for contrib batch_norm).  Multiply then add, unfused.  One pass over Y instead of three. */
int dir_dense_affine_f32(const float* X, int64_t x_ld, const float* Wt, int64_t w_ld, const float* bias, int act, const float* post_scale,
                         const float* post_shift, int64_t M, int Kd, int N, float* Y, int64_t y_ld, dir_stream_t stream);
/* The same layer for SMALL batches (round 5): the reference trains and evaluates at batch 100 / 256 (models/DeepCrossNetwork/train.py:16-17),
 * where a layer is bound by the longest dependent chain, not by any throughput.  One wave per 16 x 16 output tile, operands straight from
 * L2 as 16-byte loads, v_mfma_f32_16x16x4_f32 on four independent accumulators, no LDS, no barrier: (M / 16) x (N / 16) waves.  Arguments
 * of dir_dense_affine_f32 (post_scale / post_shift NULL: none); Kd, x_ld, w_ld multiples of 4, X / Wt 16-byte aligned; any N.  What
 * ops.dense routes batches of at most ops.DENSE_SMALL_ROWS rows to. */
int dir_dense_small_f32(const float* X, int64_t x_ld, const float* Wt, int64_t w_ld, const float* bias, int act, const float* post_scale,
                        const float* post_shift, int64_t M, int Kd, int N, float* Y, int64_t y_ld, dir_stream_t stream);
/* ... and for MID-SIZE batches (round 6: a few hundred to a few thousand rows -- where dir_dense_small_f32's one-tile workgroups re-read both
 * operands for every 16 x 16 outputs and dir_dense_f32's 128-row workgroups leave most of the chip idle): a workgroup per 32 / 64 x 64 tile of
 * Y, the reduction staged through LDS in double-buffered chunks of 32, fp32-input MFMA (exact products).  Arguments and limits of
 * dir_dense_small_f32.  What ops.dense routes batches between ops.DENSE_SMALL_ROWS and ops.DENSE_MID_ROWS to (reference batch sizes are set
 * by flag: models/DeepCrossNetwork/train.py:16-17). */
int dir_dense_mid_f32(const float* X, int64_t x_ld, const float* Wt, int64_t w_ld, const float* bias, int act, const float* post_scale,
                      const float* post_shift, int64_t M, int Kd, int N, float* Y, int64_t y_ld, dir_stream_t stream);
/* The same layer on the bf16 matrix pipe with fp32-equivalent arithmetic (csrc/dense_bf3.hip; the recipe of
 * dir_cin_layer_bf16x3_f32): X and W are each split into three bf16 pieces (round to nearest; the pieces sum to the operand exactly,
 * fp32 exponent range), the six piece products of weight >= 2^-16 are accumulated in fp32 by v_mfma_f32_16x16x32_bf16.  Same 1e-5
 * bar against float64 as dir_dense_f32; not bitwise equal to it.
 *   dir_dense_bf16x3_image_bytes(Kd, N): device bytes of the packed bf16 image of a [N, Kd] weight.
 *   dir_dense_bf16x3_pack_f32: W [N, Kd] fp32 (row stride w_ld) -> image (once per weight version; ~5 us).
 *   dir_dense_bf16x3_f32: Y = epilogue(X . W^T) from the image: + bias, ReLU (act), * post_scale + post_shift (the folded inference
 *     batch-norm of dir_dense_affine_f32), and/or the ReLU gate of dir_dense_gated_f32 (gate [M, N], NULL: none).
 * Limits: Kd, N, x_ld, y_ld, gate_ld multiples of 4, every operand 16-byte aligned (DIR_E_UNSUPPORTED otherwise).  Columns are
 * computed in blocks of 128 / 208 / 256 (whichever pads N least), k in steps of 32. */
int64_t dir_dense_bf16x3_image_bytes(int Kd, int N);
int dir_dense_bf16x3_pack_f32(const float* W, int64_t w_ld, int Kd, int N, void* image, int64_t image_bytes, dir_stream_t stream);
/* The same image from a weight with any element strides (w_rs between rows n, w_cs between columns k): the TRANSPOSE of a layer's kernel
 * for its data gradient g . W is packed straight from W's storage (w_rs = 1, w_cs = W's row stride) -- no transposed copy per step. */
int dir_dense_bf16x3_pack_strided_f32(const float* W, int64_t w_rs, int64_t w_cs, int Kd, int N, void* image, int64_t image_bytes,
                                      dir_stream_t stream);
int dir_dense_bf16x3_f32(const float* X, int64_t x_ld, const void* image, const float* bias, int act, const float* post_scale,
                         const float* post_shift, const float* gate, int64_t gate_ld, int64_t M, int Kd, int N, float* Y, int64_t y_ld,
                         dir_stream_t stream);

/* The layer on "fp16 x 2" arithmetic (csrc/dense_bf3.hip, round 4; dir_cin_layer_f16x2_f32 states the arithmetic and its preconditions):
 * for layers whose input is bounded by construction -- embedding concatenations, ReLU / batch-normalised activations, the CIN's pooled
 * products; |x|, |W| < 65 504.  No gate argument: an X of unknown magnitude (a gradient) goes through dir_dense_f16x2_rows_f32.  The image comes from
 * dir_dense_f16x2_pack_strided_f32 (dir_dense_bf16x3_image_bytes(Kd, N) bytes hold it; not interchangeable with the bf16 x 3 image). */
int dir_dense_f16x2_pack_strided_f32(const float* W, int64_t w_rs, int64_t w_cs, int Kd, int N, void* image, int64_t image_bytes,
                                     dir_stream_t stream);
int dir_dense_f16x2_f32(const float* X, int64_t x_ld, const void* image, const float* bias, int act, const float* post_scale,
                        const float* post_shift, int64_t M, int Kd, int N, float* Y, int64_t y_ld, dir_stream_t stream);
/* fp16 x 2 for an X of UNKNOWN magnitude -- the backward's data gradient dL/dx = g W (X := g, image := W^T from
 * dir_dense_f16x2_pack_strided_f32), with dir_dense_bf16x3_f32's gate argument: inside the kernel row r of X is multiplied by a power of two
 * chosen from row_bits[r] (DEVICE, [M]: the bit pattern of max_k |X[r, k]|; the row's largest element lands in [2^14, 2^15)) before the
 * split and the row's sums by the inverse before the epilogue -- both exact.  Elements within 2^-17 of their row's largest carry 22 bits,
 * smaller ones an absolute error of 2^-39 of the largest.  dir_row_absmax_bits_f32 writes row_bits and, if all_bits != NULL, the maximum
 * over all rows (what dir_dense_dw_f16x2_f32 takes); N, x_ld multiples of 4, X 16-byte aligned.  workspace (needed with all_bits):
 * dir_row_absmax_workspace_words() unsigned ints of DEVICE memory whose FIRST word is zero before the first call -- the kernel's ticket
 * counter, which every call leaves at zero again (one launch, no zeroing pass; calls sharing a workspace must be ordered on one stream). */
int dir_row_absmax_workspace_words(void);
int dir_row_absmax_bits_f32(const float* X, int64_t x_ld, int64_t M, int N, unsigned int* row_bits, unsigned int* all_bits,
                            unsigned int* workspace, dir_stream_t stream);
/* y_row_bits_out [M] / y_all_bits_out [1] (DEVICE, both or neither; zeroed here where needed): the kernel's epilogue leaves the bit patterns of
 * max_n |Y[r, n]| and of max |Y| there (atomicMax per column block) -- in a backward chain Y is the next layer's gradient, whose row-scaled
 * kernels then need no max pass of their own.  dir_units1_relu_backward_bits_f32 does the same for the chain's first gradient (an upper
 * bound |g[r]| max_n |w[n]| per row). */
int dir_dense_f16x2_rows_f32(const float* X, int64_t x_ld, const void* image, const float* bias, int act, const float* post_scale,
                             const float* post_shift, const float* gate, int64_t gate_ld, int64_t M, int Kd, int N, float* Y, int64_t y_ld,
                             const unsigned int* row_bits, unsigned int* y_row_bits_out, unsigned int* y_all_bits_out, dir_stream_t stream);
/* dir_dense_bf16x3_head_f32 on the fp16 x 2 arithmetic (the image from dir_dense_f16x2_pack_strided_f32): the last deep layer of DCN reads a
 * batch-normalised ReLU activation (DeepCrossNetwork.py:400-403) -- bounded by construction. */
int dir_dense_f16x2_head_f32(const float* X, int64_t x_ld, const void* image, const float* bias, int act, const float* post_scale,
                             const float* post_shift, int64_t M, int Kd, int N, const float* head_w, float* Y, int64_t y_ld, float* head_part,
                             dir_stream_t stream);

/* dir_dense_bf16x3_f32 (no gate) with the head of a tower folded into the epilogue -- DCN's last deep layer and the deep branch's share of
 * the final dense(1) over concat([cross, deep]) (DeepCrossNetwork.py:136-137): head_part [ncb, M] (DEVICE), ncb =
 * dir_dense_bf16x3_head_blocks(N): head_part[cb][r] = the dot product of row r's activations in column block cb with head_w [N]
 * (DEVICE, 16-byte aligned); the caller adds the ncb rows in block order.  Y may be NULL: the layer's output then never reaches memory. */
int dir_dense_bf16x3_head_blocks(int N);
int dir_dense_bf16x3_head_f32(const float* X, int64_t x_ld, const void* image, const float* bias, int act, const float* post_scale,
                              const float* post_shift, int64_t M, int Kd, int N, const float* head_w, float* Y, int64_t y_ld,
                              float* head_part, dir_stream_t stream);
/* A whole DNN tower in ONE launch (csrc/tower_bf3.hip): L <= 4 hidden layers of width <= 416 over X [M, Kd <= 416] and, optionally, the
 * units = 1 logit layer behind them -- dnn_logit_fn, models/DeepFM/deepFM.py:284-319 (concat -> [dense(units, act) ->
 * batch_normalization]* -> dense(units=1)); _base_model, models/ESMM/ESMM.py:139-146.  The arithmetic of dir_dense_bf16x3_f32 (bf16 x 3
 * split of both operands, fp32 accumulate, 1e-5 against float64); a 128-row tile's activations stay in registers from layer to layer
 * (an accumulator tile of v_mfma_f32_32x32x16_bf16 is the next layer's B operand when the weight image enumerates k the same way), so
 * only X is read and only the result is written.
 *   dir_tower_bf16x3_image_bytes(K, N) / dir_tower_bf16x3_pack_f32: the packed image of ONE layer's weight W [N, K] (nn.Linear layout,
 *     row stride w_ld) in that k order (once per weight version).
 *   dir_tower_bf16x3_f32: N, images, bias, post_scale, post_shift, act are HOST arrays of L entries (device pointers / ints; bias,
 *     post_scale, post_shift may be NULL or hold NULL entries); layer l: y = act(x W_l^T + bias_l) (* post_scale_l + post_shift_l: the
 *     folded inference batch-norm of dir_dense_affine_f32).  head_w [N_last], head_b [1] (both or neither): out[r * out_ld] =
 *     y_last[r] . head_w + head_b (+ add0[r] + add1[r]: the other logits of add_n, deepFM.py:217-223; NULL: none); without a head
 *     out [M, N_last] (row stride out_ld) receives the last activation.
 * Limits: Kd, every N_l, x_ld (and out_ld without a head) multiples of 4, <= 416; 16-byte aligned operands (DIR_E_UNSUPPORTED). */
int64_t dir_tower_bf16x3_image_bytes(int K, int N);
int dir_tower_bf16x3_pack_f32(const float* W, int64_t w_ld, int K, int N, void* image, int64_t image_bytes, dir_stream_t stream);
int dir_tower_bf16x3_f32(const float* X, int64_t x_ld, int64_t M, int Kd, int L, const int* N, const void* const* images,
                         const float* const* bias, const float* const* post_scale, const float* const* post_shift, const int* act,
                         const float* head_w, const float* head_b, const float* add0, const float* add1, float* out, int64_t out_ld,
                         dir_stream_t stream);

/* DeepFM inference in ONE launch (deepFM.py:107-117,217-223,284-338 on the serving layout of dir_gather_fm_linear_packed_f32): the tower
 * above with layer 1's input rows looked up inside the kernel -- row b = the concatenation of slot f's K = 16 embedding floats of
 * tables[f][ids[b, f]] (an id outside [0, vocab[f]) reads as a zero row) -- and the FM second-order term and the first-order term
 * (sum_f tables[f][id][lin_col] + lin_bias[0], skipped when lin_col < 0) added to the head's logit:
 *   out[b] = head(tower(concat_b)) + fm_b + lin_b (+ add0[b] + add1[b]).
 * Every sum runs in the order of dir_gather_fm_linear_packed_f32 + dir_tower_bf16x3_f32(add0 = fm, add1 = lin): the result is that
 * two-launch path's bit for bit, without the [M, F*K] concat ever reaching memory.  want_fm = 0 leaves the FM term out (with lin_col < 0:
 * a plain tower over looked-up rows -- ESMM's towers on [vocab, 16] tables, ld = 16; the head is then optional).  F <= 26, K = 16;
 * tables / vocab / ids / strides as dir_gather_fm_linear_packed_f32, the other arguments as dir_tower_bf16x3_f32. */
int dir_deepfm_tower_bf16x3_f32(const float* const* tables, const int64_t* vocab, int F, int K, int64_t ld, int lin_col,
                                const int64_t* ids, int64_t stride_b, int64_t stride_f, int want_fm, int64_t M, const float* lin_bias, int L,
                                const int* N, const void* const* images, const float* const* bias, const float* const* post_scale,
                                const float* const* post_shift, const int* act, const float* head_w, const float* head_b,
                                const float* add0, const float* add1, float* out, int64_t out_ld, dir_stream_t stream);

/* The tower on "fp16 x 2" arithmetic (csrc/tower_bf3.hip, round 4; dir_cin_layer_f16x2_f32 states the arithmetic and its preconditions:
 * |activations|, |weights| < 65 504, an absolute representation error of up to 2^-25 per operand element below 2^-3): a stage of W is two
 * 13 KB pieces instead of three -- two thirds of the LDS reads of a kernel that was LDS-read bound -- and a tile costs three matrix
 * instructions instead of six.  Same arguments as the bf16 x 3 entries; the images come from dir_tower_f16x2_pack_f32
 * (dir_tower_bf16x3_image_bytes(K, N) bytes hold them) and are NOT interchangeable with the bf16 x 3 images. */
int dir_tower_f16x2_pack_f32(const float* W, int64_t w_ld, int K, int N, void* image, int64_t image_bytes, dir_stream_t stream);
int dir_tower_f16x2_f32(const float* X, int64_t x_ld, int64_t M, int Kd, int L, const int* N, const void* const* images,
                        const float* const* bias, const float* const* post_scale, const float* const* post_shift, const int* act,
                        const float* head_w, const float* head_b, const float* add0, const float* add1, float* out, int64_t out_ld,
                        dir_stream_t stream);
int dir_deepfm_tower_f16x2_f32(const float* const* tables, const int64_t* vocab, int F, int K, int64_t ld, int lin_col,
                               const int64_t* ids, int64_t stride_b, int64_t stride_f, int want_fm, int64_t M, const float* lin_bias, int L,
                               const int* N, const void* const* images, const float* const* bias, const float* const* post_scale,
                               const float* const* post_shift, const int* act, const float* head_w, const float* head_b,
                               const float* add0, const float* add1, float* out, int64_t out_ld, dir_stream_t stream);
/* The same tower, COLUMN-SPLIT (csrc/tower_cs.hip, round 6; reference dnn_logit_fn, /root/reference/models/DeepFM/deepFM.py:284-319, and with
 * the lookups inside the _model_fn graph, deepFM.py:217-223,321-335): a workgroup is 64 batch rows whose layer input sits in LDS as fp16 hi / lo
 * pieces; every wave owns 3-4 output column tiles for all rows and reads ITS weight fragments straight from the L2-resident image (no LDS copy
 * of the weights, two barriers per layer instead of one per stage).  fp16 x 2 arithmetic and every argument as dir_tower_f16x2_f32 /
 * dir_deepfm_tower_f16x2_f32 (same preconditions, same error behaviour); the images come from dir_tower_cs_f16x2_pack_f32
 * (dir_tower_cs_image_bytes(K, N) bytes: [k-step][column tile][piece][lane] in the matrix instruction's natural k order) and are NOT
 * interchangeable with the other tower images.  Results agree with dir_tower_f16x2_f32 to rounding (another summation order inside a k-step and in
 * the head's dot product), not bit for bit; the GATHER form and the plain form of THIS kernel agree bit for bit, and its FM / first-order terms are
 * bit for bit dir_gather_fm_linear_packed_f32's.
 * Row scaling (round 6; environment DIR_TOWER_RS = 1 (default) | 0, read per call): every row the kernel stores as a layer's input -- the
 * looked-up / given input rows and the activations between layers -- is multiplied by its own power of two first (exact; the reading layer's
 * accumulators are multiplied back), so that neither the inputs' nor the interior activations' magnitudes are a precondition: only the WEIGHTS
 * must lie inside fp16's range (|w| < 65 504; the host routes on max |W|).  The image ends in a 256-byte trailer, 64 floats whose maximum is
 * max_n sum_k |W[n][k]| (formed by the pack launch): the scales of layers 2.. come from a bound through it. */
int64_t dir_tower_cs_image_bytes(int K, int N);
int dir_tower_cs_f16x2_pack_f32(const float* W, int64_t w_ld, int K, int N, void* image, int64_t image_bytes, dir_stream_t stream);
int dir_tower_cs_f16x2_f32(const float* X, int64_t x_ld, int64_t M, int Kd, int L, const int* N, const void* const* images,
                           const float* const* bias, const float* const* post_scale, const float* const* post_shift, const int* act,
                           const float* head_w, const float* head_b, const float* add0, const float* add1, float* out, int64_t out_ld,
                           dir_stream_t stream);
int dir_deepfm_tower_cs_f16x2_f32(const float* const* tables, const int64_t* vocab, int F, int K, int64_t ld, int lin_col,
                                  const int64_t* ids, int64_t stride_b, int64_t stride_f, int want_fm, int64_t M, const float* lin_bias, int L,
                                  const int* N, const void* const* images, const float* const* bias, const float* const* post_scale,
                                  const float* const* post_shift, const int* act, const float* head_w, const float* head_b,
                                  const float* add0, const float* add1, float* out, int64_t out_ld, dir_stream_t stream);
/* dir_esmm_head_f32: ESMM's prediction head in inference (/root/reference/models/ESMM/ESMM.py:67-77): p = sigmoid(ctr_logit) * sigmoid(cvr_logit),
 * clipped to [eps, 1 - eps] (the reference's 1e-7), ctcvr_logit = log(p / (1 - p)); [B] contiguous each.  One launch for seven elementwise ops. */
int dir_esmm_head_f32(const float* ctr_logit, const float* cvr_logit, int64_t B, float eps, float* ctcvr_logit, dir_stream_t stream);
/* dir_dense_gated_f32: Y = (gate > 0) ? X . Wt^T : 0 -- the data gradient of a dense layer taken straight through the previous
 * layer's ReLU: X = dL/d(pre-activation of layer l) [M, Kd = units of l], Wt = the TRANSPOSE of layer l's nn.Linear weight
 * ([in_l, units_l] rows), gate = layer l-1's output [M, N = in_l]; the result is dL/d(pre-activation of layer l-1). */
int dir_dense_gated_f32(const float* X, int64_t x_ld, const float* Wt, int64_t w_ld, const float* gate, int64_t gate_ld, int64_t M,
                        int Kd, int N, float* Y, int64_t y_ld, dir_stream_t stream);
/* Weight gradient of a dense layer (the kernel gradient of tf.layers.dense in deepFM.py:295-300, DeepCrossNetwork.py:394-399,
 * ESMM.py:139-142 under TensorFlow autodiff):  dW[n, k] = sum_r g[r, n] * x[r, k],  g = dL/d(pre-activation) [M, N] (row stride g_ld),
 * x = the layer's input [M, K] (row stride x_ld), dW [N, K] (row stride dw_ld; nn.Linear's weight layout) -- on the bf16 matrix pipe
 * with fp32-equivalent arithmetic (csrc/dense_dw_bf3.hip; the recipe of dir_dense_bf16x3_f32), both operands transposed and split
 * on the fly.  db [N] (or NULL): the layer's bias gradient sum_r g[r, n], formed from the g rows the kernel stages anyway.
 * Row spans leave partial sums in `workspace` (dir_dense_dw_bf16x3_workspace_bytes(M, N, K) bytes, 16-byte aligned)
 * that are added in span order: bitwise reproducible, no atomics.  Any M (row tails are zero-filled); N, K, g_ld, x_ld multiples of 4
 * and g / x 16-byte aligned (DIR_E_UNSUPPORTED otherwise: rows are staged with 16-byte loads). */
int64_t dir_dense_dw_bf16x3_workspace_bytes(int64_t M, int N, int K);
int dir_dense_dw_bf16x3_f32(const float* g, int64_t g_ld, const float* x, int64_t x_ld, int64_t M, int N, int K, float* dW, int64_t dw_ld,
                            float* db, void* workspace, int64_t workspace_bytes, dir_stream_t stream);
/* The same weight gradient on fp16 x 2: g is multiplied by ONE power of two for the whole tensor (from g_absmax_bits, DEVICE: the bit pattern
 * of an upper bound of max |g| -- dir_row_absmax_bits_f32's all_bits) before its split and dW by the inverse in the reduce pass (exact; db
 * sums the raw g); x is split as in the fp16 x 2 forward: |x| < 65 504 and O(1) (embedding concatenations, activations).  Arguments,
 * workspace and determinism of dir_dense_dw_bf16x3_f32. */
int dir_dense_dw_f16x2_f32(const float* g, int64_t g_ld, const float* x, int64_t x_ld, int64_t M, int N, int K, float* dW, int64_t dw_ld,
                           float* db, void* workspace, int64_t workspace_bytes, const unsigned int* g_absmax_bits, dir_stream_t stream);
/* ... with x scaled by ONE power of two as well (x_absmax_bits, DEVICE: the bit pattern of an upper bound of max |x| -- the all_bits that
 * dir_row_absmax_bits_f32 or the row-scaled forward kernel's epilogue left for this layer's input): neither operand's magnitude is assumed
 * (round 5: what the training towers run; the entry above remains for callers that hold no bound for x). */
int dir_dense_dw_f16x2_scaled_f32(const float* g, int64_t g_ld, const float* x, int64_t x_ld, int64_t M, int N, int K, float* dW, int64_t dw_ld,
                                  float* db, void* workspace, int64_t workspace_bytes, const unsigned int* g_absmax_bits,
                                  const unsigned int* x_absmax_bits, dir_stream_t stream);
/* The same product for small gradients (N <= 128, K <= 256, at most 256 tiles of 8 x 8; multiples of 4): fp32 FMAs on register tiles over row spans (exact fp32 products),
 * for the tall-and-skinny TN products the library runs at 200 us -- the per-sample term of the DIN unit's first layer (S^T a, 80 x 64)
 * and the narrow last layers of the towers.  Same arguments, workspace query and determinism as dir_dense_dw_bf16x3_f32. */
int64_t dir_dense_dw_small_workspace_bytes(int64_t M, int N, int K);
int dir_dense_dw_small_f32(const float* g, int64_t g_ld, const float* x, int64_t x_ld, int64_t M, int N, int K, float* dW, int64_t dw_ld,
                           float* db, void* workspace, int64_t workspace_bytes, dir_stream_t stream);
/* Backward of the units = 1 logit layer (models/DeepFM/deepFM.py:311-317, models/ESMM/ESMM.py:146) taken straight through the ReLU
 * of the hidden layer below it, in one pass (csrc/head_bwd.hip).  g = dL/dlogit [B], w = the logit layer's weight [N], y = the hidden
 * layer's output [B, N] (row stride y_ld):
 *   gx[b,n] = (y[b,n] > 0) ? g[b] * w[n] : 0          dL/d(pre-activation of the hidden layer), row stride gx_ld
 *   partials[p][0][n] = this block's share of sum_b gx[b,n]        (the hidden layer's bias gradient)
 *   partials[p][1][n] = this block's share of sum_b g[b] * y[b,n]  (the logit layer's weight gradient)
 * for p < dir_units1_relu_backward_partials(B, N); the caller adds the row pairs (fixed order: bitwise reproducible, no atomics).
 * Limits: N, y_ld, gx_ld multiples of 4, N <= 4096, w / y / gx / partials 16-byte aligned (DIR_E_UNSUPPORTED / DIR_E_BADARG). */
int64_t dir_units1_relu_backward_partials(int64_t B, int N);
int dir_units1_relu_backward_f32(const float* g, const float* w, const float* y, int64_t y_ld, int64_t B, int N, float* gx, int64_t gx_ld,
                                 float* partials, int64_t n_partials, dir_stream_t stream);
int dir_units1_relu_backward_bits_f32(const float* g, const float* w, const float* y, int64_t y_ld, int64_t B, int N, float* gx, int64_t gx_ld,
                                      float* partials, int64_t n_partials, unsigned int* gx_row_bits, unsigned int* gx_all_bits,
                                      dir_stream_t stream);

/* The forward of a units = 1 layer, y[b * y_ld] = x[b, :] . w + bias[0] (bias NULL: none): the logit heads in training -- where the hidden
 * activation is kept for the backward, so the inference kernels' fused head does not apply -- (models/DeepFM/deepFM.py:311-317,
 * models/ESMM/ESMM.py:146), DCN's final dense(1) over concat([cross, deep]) (models/DeepCrossNetwork/DeepCrossNetwork.py:136-137) and
 * xDeepFM's CIN output layer: one pass over x, any width (16-byte loads when N and x_ld are multiples of 4 and x, w 16-byte aligned),
 * bitwise reproducible.  x [B, N] row stride x_ld, w [N], bias device [1]. */
int dir_units1_f32(const float* x, int64_t x_ld, int64_t B, int N, const float* w, const float* bias, float* y, int64_t y_ld,
                   dir_stream_t stream);

/* The units = 1 logit layer on top of an activation of any width that is not a ReLU output -- DCN's cross output under the final dense(1)
 * over concat([cross, deep]) (models/DeepCrossNetwork/DeepCrossNetwork.py:136-137), d = 429: gx[b,n] = g[b] * w[n] (gx NULL: skipped),
 * dw[n] = sum_b g[b] * x[b,n], one pass over x.  partials: N * dir_units1_backward_partials(B, N) floats of scratch; the column sums are
 * added in workgroup order (bitwise reproducible, no atomics).  No alignment or width limits (4-byte accesses). */
int64_t dir_units1_backward_partials(int64_t B, int N);
int dir_units1_backward_f32(const float* g, const float* w, const float* x, int64_t x_ld, int64_t B, int N, float* gx, int64_t gx_ld, float* dw,
                            float* partials, int64_t n_partials, dir_stream_t stream);
/* Training-mode batch normalisation of a hidden layer's activation y [B, N] (row stride y_ld): tf.layers.batch_normalization(training=True)
 * after each hidden layer of dnn_logit_fn (models/DeepFM/deepFM.py:303-308) and tf.contrib.layers.batch_norm(scale=False) in
 * _deep_architecture (models/DeepCrossNetwork/DeepCrossNetwork.py:400-403, 413-419); csrc/bn_train.hip.  [TF-upstream] rank-2 input:
 * the batch's mean and POPULATION variance normalise; moving = moving * momentum + batch * (1 - momentum).
 * dir_bn_train_stats_f32: one read of y -> mean[n], inv[n] = rsqrt(var[n] + eps), scale[n] = inv * gamma (gamma NULL: inv),
 *   shift[n] = beta - mean * scale (beta NULL: 0); moving_mean / moving_var (NULL: skipped) updated in place.  The normalised activation
 *   is y * scale + shift (one elementwise pass, the caller's).
 * dir_bn_train_backward_f32: given g = dL/d(normalised activation) [B, N] -> gy = scale * (g - mean_b(g) - xhat * mean_b(g * xhat)),
 *   xhat = (y - mean) * inv; relu_gate != 0: gy is zeroed where y <= 0 (y is the ReLU output of the layer below: gy is then dL/d of that
 *   layer's pre-activation); gbeta[n] = sum_b g, ggamma[n] = sum_b g * xhat (either may be NULL).  coef: 3 * N floats of scratch.
 * partials: 2 * N * dir_bn_train_partials(B, N) floats of scratch; column sums are fp32 within a workgroup and fp64 across workgroups in
 * workgroup order (bitwise reproducible, no atomics).  Limits: N and the row strides multiples of 4, N <= 4096, B > 0, 16-byte aligned
 * y / g / gy / coef / partials (DIR_E_UNSUPPORTED / DIR_E_BADARG). */
int64_t dir_bn_train_partials(int64_t B, int N);
int dir_bn_train_stats_f32(const float* y, int64_t y_ld, int64_t B, int N, float eps, float momentum, const float* gamma, const float* beta,
                           float* moving_mean, float* moving_var, float* mean, float* inv, float* scale, float* shift, float* partials,
                           int64_t n_partials, dir_stream_t stream);
int dir_bn_train_backward_f32(const float* g, int64_t g_ld, const float* y, int64_t y_ld, int64_t B, int N, const float* mean,
                              const float* inv, const float* gamma, int relu_gate, float* gy, int64_t gy_ld, float* gbeta, float* ggamma,
                              float* coef, float* partials, int64_t n_partials, dir_stream_t stream);
/* dir_dice_train_backward_f32 (round 6): the whole backward of Dice in TRAINING mode over rows s [B, N] -- y = s (alpha + (1 - alpha) p),
 *   p = sigmoid(scale s + shift) with THIS batch's statistics (mean, inv, scale = inv, shift = -mean inv from dir_bn_train_stats_f32) -- in two
 *   passes over (g, s): ds = g (alpha + (1 - alpha) p) + the batch-norm backward of gx = g s (1 - alpha) p (1 - p); galpha[n] = sum_b g s (1 - p).
 *   Replaces dir_act_rows_backward_f32 + dir_bn_train_backward_f32 + an add (4 writes, 7 reads of [B, N] -> 1 write, 4 reads); the same
 *   expressions, column sums fp32 within a workgroup and fp64 across workgroups in workgroup order (bitwise reproducible).
 *   coef: 3 * N floats of scratch; partials: 3 * N * dir_bn_train_partials(B, N) floats.  Limits as dir_bn_train_backward_f32; alpha / scale /
 *   shift 16-byte aligned.  No reference code: arXiv:1706.06978 section 5.3 (README.md:25 lists DIN). */
int dir_dice_train_backward_f32(const float* g, int64_t g_ld, const float* s, int64_t s_ld, int64_t B, int N, const float* alpha,
                                const float* scale, const float* shift, const float* mean, const float* inv, float* ds, int64_t ds_ld,
                                float* galpha, float* coef, float* partials, int64_t n_partials, dir_stream_t stream);

/* --------------------------------------------------------------------------------------------
 * Backward of the HBM-bound interaction ops: derivatives of the same reference
 * expressions (the reference trains through TensorFlow autodiff of deepFM.py:321-335 and
 * DeepCrossNetwork.py:336-367).
 *
 * dir_fm_second_order_backward_f32: demb[b,f,:] = g[b] * (sum_f' e[b,f',:] - e[b,f,:]) (+ add_in[b,f,:]).
 *   add_in (optional) is the gradient arriving from the DNN branch: the total d/d(embedding) of DeepFM in one pass.
 * dir_dcn_cross_backward_f32: given gout = dL/dx_L, returns gx0 = dL/dx0 [B,d] and gw, gb = dL/dw, dL/db [L,d]
 *   (summed over the batch through per-workgroup partials added in a fixed order: bitwise reproducible).
 *   workspace: dir_dcn_cross_backward_workspace_bytes(L, d) device bytes.
 * ------------------------------------------------------------------------------------------ */
int dir_fm_second_order_backward_f32(const float* emb, int64_t emb_ld, const float* g, const float* add_in,
                                     int64_t add_ld, int64_t B, int F, int K, float* demb, int64_t demb_ld,
                                     dir_stream_t stream);
int64_t dir_dcn_cross_backward_workspace_bytes(int L, int d);
int dir_dcn_cross_backward_f32(const float* x0, int64_t x_ld, const float* w, const float* b, int L,
                               const float* gout, int64_t g_ld, int64_t B, int d, float* gx0, int64_t gx_ld,
                               float* gw, float* gb, void* workspace, dir_stream_t stream);

/* --------------------------------------------------------------------------------------------
 * Backward of the CIN layer (A14; no reference code).  With G = dL/dxout [B, H, D] (the caller adds the pooled
 * gradient, broadcast over d, into G):
 *   dW[h, i*m+j]  = sum_{b,d} G[b,h,d] * xk[b,i,d] * x0[b,j,d]          -> dir_cin_dw_f32 (fp32 MFMA)
 *   dxk[b,i,d]    = sum_{h,j} W[h, i*m+j] * G[b,h,d] * x0[b,j,d]        = dir_cin_layer_f32(x0, G, W1) with
 *                   W1[i, h*m+j] = W[h, i*m+j]
 *   dx0[b,j,d]    = sum_{h,i} W[h, i*m+j] * G[b,h,d] * xk[b,i,d]        = dir_cin_layer_f32(xk, G, W2) with
 *                   W2[j, h*Hp+i] = W[h, i*m+j]   (xk in channel groups of <= 40 when Hp > 40)
 * i.e. the two data gradients are the forward contraction with permuted weights (host mirror: autograd.CinLayer).
 * dir_cin_dw_f32: accumulate != 0 adds into dW; workspace: dir_cin_dw_workspace_bytes(...) device bytes,
 * 16-byte aligned.  Row ranges are reduced in a fixed order: bitwise reproducible.
 * ------------------------------------------------------------------------------------------ */
int64_t dir_cin_dw_workspace_bytes(int m, int Hp, int H, int D, int64_t B);
/* dir_cin_dx_f32: both data gradients in one pass (T_j = G x W_j on fp32 MFMA with G held in registers, then an FMA
 * epilogue for dxk and a half-wave reduction for dx0).  Wp is W permuted to [NB][m][H][32][CT]: CT = 1, 2 or 4 column tiles
 * (Hp <= 32, <= 64, else 4), NB = ceil(Hp / (32*CT)) column blocks (2 when 128 < Hp <= 256),
 * Wp[nb][j][h][n][cc] = W[h, i*m + j] with i = nb*32*CT + 32*cc + n, zero where i >= Hp.
 * 128 < H <= 256 runs as two 128-row slices of H per field; with two column blocks each adds its dx0 share (two addends on a
 * zeroed buffer: order-free, still reproducible).
 * Limits: H <= 256, Hp <= 256, m <= 64 (DIR_E_UNSUPPORTED otherwise: use the dir_cin_layer_f32 formulation above). */
/* dir_cin_dw_bf16x3_f32: the weight gradient on the bf16 matrix pipe with fp32-equivalent arithmetic (csrc/cin_dw_bf3.hip): per field j,
 * dW_j = (G * x0_j)^T . xk with both operands split into three bf16 pieces (six piece products, fp32 accumulate; the field factor is
 * inside an operand, so that operand is formed and split per field).  Same semantics as dir_cin_dw_f32 (accumulate, fixed-order
 * reduction of the row spans: bitwise reproducible); D in {8, 16, 32}; best for Hp >= 96 (narrower layers leave its 8 column tiles
 * idle: dir_cin_dw_f32).  workspace: dir_cin_dw_bf16x3_workspace_bytes(...) device bytes, 16-byte aligned. */
int64_t dir_cin_dw_bf16x3_workspace_bytes(int m, int Hp, int H, int D, int64_t B);
int dir_cin_dw_bf16x3_f32(const float* x0, const float* xk, const float* G, int m, int Hp, int H, int D, int64_t B, int accumulate,
                          float* dW, void* workspace, int64_t workspace_bytes, dir_stream_t stream);
/* The same weight gradient on fp16 x 2 (csrc/cin_dw_bf3.hip, round 4): G is multiplied by ONE power of two for the whole tensor (its largest
 * |element| lands in [2^12, 2^13); exact) before A_j = G x0_j is formed and split, and the reduce pass takes the scale out again; xk is
 * split as in the fp16 x 2 forward.  Preconditions: |x0| < 8, |xk| < 65 504, both O(1) (embeddings / activations).  g_absmax_bits: device
 * pointer to the bit pattern of max |G| as an unsigned (any value >= the true maximum works, a power-of-two bound included) or NULL -- then
 * a max pass over G runs first, into the workspace's tail.  workspace: dir_cin_dw_f16x2_workspace_bytes(...) bytes, 16-byte aligned. */
int64_t dir_cin_dw_f16x2_workspace_bytes(int m, int Hp, int H, int D, int64_t B);
int dir_cin_dw_f16x2_f32(const float* x0, const float* xk, const float* G, int m, int Hp, int H, int D, int64_t B, int accumulate, float* dW,
                         void* workspace, int64_t workspace_bytes, const unsigned int* g_absmax_bits, dir_stream_t stream);
/* The same gradient for the FIRST layer of a stack (xk = x0, Hp = m), where the result is symmetric in (i, j): the m (m + 1) / 2
 * unordered pairs are the GEMM's columns (operand x0_i * x0_j formed and split per k-step, G split once per step), csrc/cin_dw_sym_bf3.hip;
 * dW [H, m*m] gets both halves.  D in {8, 16, 32}, m <= 64; workspace: dir_cin_dw_sym_bf16x3_workspace_bytes(m, H, D, B) bytes, 16-byte
 * aligned; partial sums are added in span order (bitwise reproducible). */
int64_t dir_cin_dw_sym_bf16x3_workspace_bytes(int m, int H, int D, int64_t B);
int dir_cin_dw_sym_bf16x3_f32(const float* x0, const float* G, int m, int H, int D, int64_t B, int accumulate, float* dW, void* workspace,
                              int64_t workspace_bytes, dir_stream_t stream);
/* The same first-layer weight gradient on fp16 x 2: G times one power of two from g_absmax_bits (DEVICE; required: the bit pattern of an upper
 * bound of max |G| -- dir_cin_layer_grad_f16x2_f32 / dir_cin_layer_dot_add_f16x2_f32 leave it), the pair products split as in the fp16 x 2
 * forward; workspace and determinism of the bf16 x 3 entry. */
int dir_cin_dw_sym_f16x2_f32(const float* x0, const float* G, int m, int H, int D, int64_t B, int accumulate, float* dW, void* workspace,
                             int64_t workspace_bytes, const unsigned int* g_absmax_bits, dir_stream_t stream);
int dir_cin_dx_f32(const float* x0, const float* xk, const float* Wp, const float* G, int m, int Hp, int H, int D,
                   int64_t B, float* dxk, float* dx0, dir_stream_t stream);
int dir_cin_dw_f32(const float* x0, const float* xk, const float* G, int m, int Hp, int H, int D, int64_t B,
                   int accumulate, float* dW, void* workspace, dir_stream_t stream);

/* --------------------------------------------------------------------------------------------
 * Backward of the DIN unit + pooling (A13; no reference code: the derivative of dir_din_attention_pool_f32's definition).
 * One fused pass over the VALID (sample, position) rows: the unit is recomputed, then with g = dL/dout [B, K]
 *   dw_j = g . h_j;   ds_j = dw_j  (normalize = 0)   or   w_j (dw_j - sum_i w_i dw_i) / sqrt(K)  (masked softmax)
 *   dpre2 = ds W3 z2 (1 - z2);   dpre1 = (dpre2 W2^T) z1 (1 - z1)
 * and, in the regrouped form of layer 1 ([h, a, h-a, h*a].W1 = h.(Wh+Wd) + (h*a).Wp + a.(Wa-Wd), W1 = [Wh; Wa; Wd; Wp]):
 *   gAP [2K, H1]  = [h | h*a]^T dpre1      (rows [0,K): d(Wh+Wd); rows [K,2K): dWp)
 *   gW2 [H1, H2]  = z1^T dpre2,   gb2 [H2] = sum dpre2,   gW3 [H2] = sum ds z2,   gb3 [1] = sum ds
 *   gh  [N, K]    = dX_h + dX_p * a + w_j g      with dX = dpre1 [Wh+Wd | Wp]^T: the gradient of history row j, written to
 *                   compact row  row_off[b] + (number of valid positions before j in sample b)
 *   ga  [B, K]    = sum_j dX_p * h_j            the candidate row's gradient WITHOUT the per-sample term
 *   S   [B, H1]   = sum_j dpre1                 the per-sample term's seed: the caller finishes
 *                   ga += S (Wa-Wd)^T,  d(Wa-Wd) = a^T S,  db1 = sum_b S,
 *                   dWh = gAP[:K],  dWa = d(Wa-Wd),  dWd = dWh - dWa,  dWp = gAP[K:]     (host mirror: ops.din_attention_pool_backward)
 * A position is valid when j < min(hist_len[b], T) and hist[b, j] >= 0; row_off [B] int64 is the exclusive prefix sum of the
 * per-sample valid counts (so gh has N = row_off[B-1] + count[B-1] rows, in (b, j) order; gh may be NULL when N = 0).
 * Weight gradients are summed through one partial record per workgroup, added in a fixed order: bitwise reproducible.
 * workspace: dir_din_backward_workspace_bytes(K, H1, H2) device bytes (0 = shape not covered).
 * Limits: K = 64, H1 <= 80, H2 <= 48 (multiples of 4), T <= 64; DIR_E_UNSUPPORTED otherwise.
 * ------------------------------------------------------------------------------------------ */
int64_t dir_din_backward_workspace_bytes(int K, int H1, int H2);
int dir_din_attention_pool_backward_f32(const float* table, int K, const int64_t* hist, const int32_t* hist_len,
                                        const int64_t* cand, int T, const float* W1, const float* b1, int H1,
                                        const float* W2, const float* b2, int H2, const float* W3, const float* b3,
                                        int normalize, int64_t B, const float* gout, const int64_t* row_off, float* gh,
                                        float* ga, float* S, float* gAP, float* gW2, float* gb2, float* gW3, float* gb3,
                                        void* workspace, dir_stream_t stream);

/* The same backward as two kernels (csrc/din_bwd_rows.hip): a wave-per-sample pass for everything that is per history row or per
 * sample (gh, ga, S) and a streaming pass over its scratch records for the batch-wide weight gradients.  Two more inputs:
 *   scores   [B, T]: the attention weights of the forward (dir_din_attention_pool_f32's `scores` output: softmax weights when
 *            normalize, raw scores otherwise) -- the softmax is not recomputed;
 *   tile_off [B] DEVICE int64: exclusive prefix sum of ceil(min(max(hist_len[b], 0), T) / 16), n_tiles = its total (a host value:
 *            it sizes the scratch, dir_din_backward_rows_workspace_bytes(K, H1, H2, n_tiles) bytes, 16-byte aligned).
 * Outputs, row_off and the meaning of every other argument as dir_din_attention_pool_backward_f32; b3 is not read. */
int64_t dir_din_backward_rows_workspace_bytes(int K, int H1, int H2, int64_t n_tiles);
int dir_din_attention_pool_backward_rows_f32(const float* table, int K, const int64_t* hist, const int32_t* hist_len,
                                             const int64_t* cand, int T, const float* W1, const float* b1, int H1,
                                             const float* W2, const float* b2, int H2, const float* W3, const float* b3,
                                             int normalize, int64_t B, const float* gout, const float* scores,
                                             const int64_t* row_off, const int64_t* tile_off, int64_t n_tiles, float* gh,
                                             float* ga, float* S, float* gAP, float* gW2, float* gb2, float* gW3, float* gb3,
                                             void* workspace, int64_t workspace_bytes, dir_stream_t stream);

/* Training pair that does not recompute the forward (K = 64, H1 <= 80, H2 <= 48, T <= 64 only).
 * dir_din_attention_pool_save_f32: dir_din_attention_pool_f32 (same outputs; `scores` is required) that also leaves, for every
 *   history position inside its sample's length, the two hidden layers' activations z1 [H1 -> 80] and z2 [H2 -> 48] in the row's record
 *   of `workspace` (212 floats per row, 16 rows per tile, sample b's tiles from tile_off[b]; tile_off / n_tiles as above;
 *   workspace_bytes >= dir_din_backward_rows_workspace_bytes(K, H1, H2, n_tiles), 16-byte aligned).  3.4 KB per 16-row tile of HBM
 *   instead of 220 of the backward's 440 MFMAs per tile.
 * dir_din_attention_pool_backward_saved_f32: dir_din_attention_pool_backward_rows_f32 on the SAME workspace, hist, hist_len and
 *   tile_off, reading z1 / z2 from the records instead of recomputing them (W1's candidate block, b1 and b2 are not read).
 * The workspace belongs to the caller between the two calls (one per forward in flight). */
int dir_din_attention_pool_save_f32(const float* table, int K, const int64_t* hist, const int32_t* hist_len, const int64_t* cand,
                                    int T, const float* W1, const float* b1, int H1, const float* W2, const float* b2, int H2,
                                    const float* W3, const float* b3, int normalize, int64_t B, float* out, float* scores,
                                    const int64_t* tile_off, int64_t n_tiles, void* workspace, int64_t workspace_bytes,
                                    dir_stream_t stream);
/* The two entries above with the arithmetic of the unit's two MFMA layers given by argument instead of by the DIR_DIN_ARITH environment
 * switch (round 5): DIR_DIN_ARITH_F32 = fp32-input MFMA (any magnitudes), DIR_DIN_ARITH_BF16X3 = three bf16 pieces per operand (fp32's
 * exponent range), DIR_DIN_ARITH_F16X2 = two fp16 pieces, UNSCALED (|table|, |W| < 65 504 and not far below 1: an element under 2^-3 keeps an
 * absolute 2^-25), DIR_DIN_ARITH_DEFAULT = what the other entries run.  The caller that knows its operands' magnitudes chooses; the Python
 * surface (ops.din_attention_pool) measures max |table| / max |W| per tensor version and takes bf16 x 3 outside [2^-6, 2^15). */
#define DIR_DIN_ARITH_DEFAULT (-1)
#define DIR_DIN_ARITH_F32 0
#define DIR_DIN_ARITH_BF16X3 1
#define DIR_DIN_ARITH_F16X2 2
int dir_din_attention_pool_arith_f32(const float* table, int K, const int64_t* hist, const int32_t* hist_len, const int64_t* cand, int T,
                                     const float* W1, const float* b1, int H1, const float* W2, const float* b2, int H2, const float* W3,
                                     const float* b3, int normalize, int activation, const float* act_params, int arith, int64_t B, float* out,
                                     float* scores, dir_stream_t stream);
/* The PACKED form of the (K = 64, H1 <= 80, H2 <= 48; any T <= 65 535) unit on fp16 x 2 (round 6; csrc/din_pack.hip): a wave lays the valid
 * history rows of a run of consecutive samples end to end in 16-row MFMA tiles (no per-sample tile padding: lengths U{1..50} cost 26 rows, not
 * 33), forms the per-sample term of 16 samples as one MFMA operand, and takes its samples from a STATIC equal-weight partition (no queue
 * atomics; bit for bit the same result on every run).  Same definition, arguments and outputs as dir_din_attention_pool_arith_f32 with
 * arith = DIR_DIN_ARITH_F16X2 (operands UNSCALED: the caller vouches for |table|, |W| inside fp16 x 2's window, as there), plus
 *   image:     the unit's weights in the kernel's operand order, dir_din_pack_image_bytes() device bytes, 16-byte aligned, built by
 *              dir_din_pack_weights_f32 -- once per weight version: every workgroup copies it into LDS instead of re-deriving it from
 *              W1 / W2 (20 us of a 230 us launch).  NULL: the entry builds one in the workspace on every call (W1 .. W3 and act_params
 *              are read then, and only then).
 *   workspace: dir_din_pack_workspace_bytes(B, scores != NULL) device bytes, 16-byte aligned (the partition's per-chunk weight sums; with
 *              scores also the samples' softmax maxima / sums, from which a second small kernel turns the raw scores into weights; room
 *              for an image).
 * Two launches on `stream` (+ 1 with scores, + 1 without an image); graph-capturable.  No reference code (README.md:27 -> arXiv:1706.06978). */
int64_t dir_din_pack_workspace_bytes(int64_t B, int want_scores);
int64_t dir_din_pack_image_bytes(void);
int dir_din_pack_weights_f32(const float* W1, const float* b1, int H1, const float* W2, const float* b2, int H2, const float* W3,
                             int activation, const float* act_params, void* image, dir_stream_t stream);
int dir_din_attention_pool_packed_f32(const float* table, int K, const int64_t* hist, const int32_t* hist_len, const int64_t* cand, int T,
                                      const float* W1, const float* b1, int H1, const float* W2, const float* b2, int H2, const float* W3,
                                      const float* b3, int normalize, int activation, const float* act_params, const void* image, int64_t B,
                                      float* out, float* scores, void* workspace, int64_t workspace_bytes, dir_stream_t stream);
int dir_din_attention_pool_save_arith_f32(const float* table, int K, const int64_t* hist, const int32_t* hist_len, const int64_t* cand, int T,
                                          const float* W1, const float* b1, int H1, const float* W2, const float* b2, int H2, const float* W3,
                                          const float* b3, int normalize, int arith, int64_t B, float* out, float* scores,
                                          const int64_t* tile_off, int64_t n_tiles, void* workspace, int64_t workspace_bytes, dir_stream_t stream);
int dir_din_attention_pool_backward_saved_f32(const float* table, int K, const int64_t* hist, const int32_t* hist_len,
                                              const int64_t* cand, int T, const float* W1, const float* b1, int H1,
                                              const float* W2, const float* b2, int H2, const float* W3, const float* b3,
                                              int normalize, int64_t B, const float* gout, const float* scores,
                                              const int64_t* row_off, const int64_t* tile_off, int64_t n_tiles, float* gh,
                                              float* ga, float* S, float* gAP, float* gW2, float* gb2, float* gW3, float* gb3,
                                              void* workspace, int64_t workspace_bytes, dir_stream_t stream);

/* Fused sparse Adagrad on the embedding tables (the reference's dnn_optimizer='Adagrad', deepFM.py:61):
 * for every distinct id of slot f in the batch: g = SUM of the gradient rows of its occurrences ([TF-upstream]
 * duplicate indices are summed before the update), accum[f][id] += g*g, tables[f][id] -= lr * g / sqrt(accum).
 * tables / accums: device arrays [F] of device pointers ([vocab_f, K] fp32, updated in place);
 * ids / strides as in dir_embedding_bag_f32 (one-hot; ids < 0 are skipped); grad [B, F*K] (row stride grad_ld);
 * head_base: DEVICE int64 [F], slot f's first entry in head; total_rows = sum vocab_f (slot f has
 * head_base[f+1] - head_base[f] rows, the last one total_rows - head_base[F-1]: an id outside its slot is skipped, never
 * written); head: persistent DEVICE int32 [total_rows], all -1 before the first call (left all -1 again on return);
 * next: DEVICE int32 [B*F] scratch. */
int dir_sparse_adagrad_f32(float* const* tables, float* const* accums, int F, int K, const int64_t* ids,
                           int64_t stride_b, int64_t stride_f, const float* grad, int64_t grad_ld, float lr,
                           int64_t B, const int64_t* head_base, int64_t total_rows, int32_t* head, int32_t* next,
                           dir_stream_t stream);

/* The same update with the (row, entry) pairs radix-sorted first (stable: duplicates are summed in batch order; no
 * atomics; bitwise reproducible) and the runs of equal rows reduced per tile of 256 sorted entries -- the longest serial
 * walk is 256 entries whatever the skew of the ids (the chain walk above serialises on hot rows).
 * row_base: DEVICE int64 [F], slot f's first row in the concatenation of all tables; total_rows = sum of the vocab sizes
 * (< 2^32 - 1).  An id outside [0, vocab_f) (vocab_f from row_base / total_rows) is skipped like a pruned id: the update
 * never writes outside slot f's table.  workspace: dir_sparse_adagrad_sorted_workspace_bytes(B, F, K, total_rows) device bytes, 256-byte aligned. */
int64_t dir_sparse_adagrad_sorted_workspace_bytes(int64_t B, int F, int K, int64_t total_rows);
int dir_sparse_adagrad_sorted_f32(float* const* tables, float* const* accums, int F, int K, const int64_t* ids,
                                  int64_t stride_b, int64_t stride_f, const float* grad, int64_t grad_ld, float lr,
                                  int64_t B, const int64_t* row_base, int64_t total_rows, void* workspace,
                                  int64_t workspace_bytes, dir_stream_t stream);

/* dir_sparse_adagrad_sorted_f32 over rows with a stride: table f's row id is tables[f] + id*row_ld, its accumulator row
 * accums[f] + id*row_ld (the packed training layout passes accums[f] = tables[f] + K, row_ld = 2K; row_ld = K with separate
 * arrays is dir_sparse_adagrad_sorted_f32).  Optionally with the FM backward folded in: fm_g [B] = d loss / d fm_logit and
 * fm_sum [B, K] = S of dir_gather_fm_rows_f32 (both or neither); the gradient of entry (b, f) is then
 *     (fm_sum[b] - tables[f][id]) * fm_g[b] + grad[b, f]          (grad may be NULL: zero)
 * -- dir_fm_second_order_backward_f32's arithmetic on the row the update reads anyway, so the [B, F*K] gradient tensor of
 * the FM term is never written or read.  Same sort, same summation order, same update: the tables end up bit-identical to
 * dir_fm_second_order_backward_f32(add_in = grad) followed by dir_sparse_adagrad_sorted_f32. */
int dir_sparse_adagrad_sorted_rows_f32(float* const* tables, float* const* accums, int64_t row_ld, int F, int K,
                                       const int64_t* ids, int64_t stride_b, int64_t stride_f, const float* grad,
                                       int64_t grad_ld, const float* fm_g, const float* fm_sum, float lr, int64_t B,
                                       const int64_t* row_base, int64_t total_rows, void* workspace,
                                       int64_t workspace_bytes, dir_stream_t stream);

/* tf.train.AdamOptimizer on the embedding tables, as the reference's train_op applies it (models/DeepCrossNetwork/DeepCrossNetwork.py:264-290:
 * every gradient clipped on its own with tf.clip_by_norm(g, 100.0); models/DeepCrossNetwork/train.py:119-124: Adam, epsilon 1e-4).  The
 * tables' gradients are IndexedSlices; [TF-upstream] AdamOptimizer._apply_sparse decays m and v of EVERY row, adds the summed row
 * gradients of the looked-up rows, and steps every row:
 *     g_r   = (sum of the entries' gradient rows of row r) * clip / max(||g of the whole table||_2, clip)     (clip_norm = 0: unclipped)
 *     m     = m * beta1 + g * (1 - beta1);   v = v * beta2 + g*g * (1 - beta2);   var -= lr_t * m / (sqrt(v) + eps)
 * with lr_t = lr * sqrt(1 - beta2^t) / (1 - beta1^t) computed by the caller (TF's "epsilon hat" form: eps is added AFTER the bias
 * corrections were folded into the step size).  ids [B, F] with element strides, grad [B, F*K] (row stride grad_ld): d loss / d
 * gathered rows; tables / ms / vs: device arrays of F device pointers to contiguous [vocab_f, K] arrays; row_base: device int64 [F].
 * One radix sort of the (row, entry) pairs serves both sorted passes (per-table gradient norm in fixed point -- order-independent --
 * then the touched rows' update); a streaming pass steps all other rows.  Bitwise reproducible.
 * workspace: dir_sparse_adam_workspace_bytes(B, F, K, total_rows) bytes, 256-byte aligned, the SAME buffer every step (it carries
 * the row marks); first_call != 0 on its first use.  Limits: F <= 64, K % 4 == 0 with K / 4 a power of two <= 64. */
int64_t dir_sparse_adam_workspace_bytes(int64_t B, int F, int K, int64_t total_rows);
int dir_sparse_adam_f32(float* const* tables, float* const* ms, float* const* vs, int F, int K, const int64_t* ids, int64_t stride_b,
                        int64_t stride_f, const float* grad, int64_t grad_ld, float lr_t, float beta1, float beta2, float eps,
                        float clip_norm, int64_t B, const int64_t* row_base, int64_t total_rows, void* workspace, int64_t workspace_bytes,
                        int first_call, dir_stream_t stream);

/* Owner side of a SHARDED backward: the n entries are the payload of dir_shard_bucket / dir_gather_packed_f32
 * (p = local_row * F + slot, p < 0 pruned) as received from all ranks, grad is [n, K] in the same order, tables / accums /
 * row_base / total_rows describe this rank's shards.  Workspace: dir_sparse_adagrad_sorted_workspace_bytes(n, 1, K, total_rows). */
int dir_sparse_adagrad_sorted_payload_f32(float* const* tables, float* const* accums, int F, int K, const int64_t* payload,
                                          int64_t n, const float* grad, float lr, const int64_t* row_base,
                                          int64_t total_rows, void* workspace, int64_t workspace_bytes, dir_stream_t stream);

/* FTRL-Proximal on sparse rows through the same sorted machinery (the reference's linear_optimizer='Ftrl', deepFM.py:58;
 * [TF-upstream] tf.train.FtrlOptimizer with learning_rate_power = -0.5): per touched row, g = sum of its gradient rows,
 *   n' = n + g^2; sigma = (sqrt(n') - sqrt(n)) / lr; z' = z + g - sigma*w;
 *   w' = |z'| > l1 ? (sign(z')*l1 - z') / (sqrt(n')/lr + 2*l2) : 0.
 * tables / accums (n) / linears (z): device arrays [F] of device pointers, [vocab_f, K] each.  The gradient row of entry
 * (b, f) is grad + b*grad_ld + f*grad_slot_stride (K floats): grad_slot_stride = 0 hands every slot the same [B, K] rows
 * -- the linear term's d logit (K = 1, deepFM.py:258-263).  Workspace as for dir_sparse_adagrad_sorted_f32. */
int dir_sparse_ftrl_sorted_f32(float* const* tables, float* const* accums, float* const* linears, int F, int K,
                               const int64_t* ids, int64_t stride_b, int64_t stride_f, const float* grad, int64_t grad_ld,
                               int64_t grad_slot_stride, float lr, float l1, float l2, int64_t B, const int64_t* row_base,
                               int64_t total_rows, void* workspace, int64_t workspace_bytes, dir_stream_t stream);

/* The same update rule on a dense variable of `count` elements with a dense gradient (the linear model's bias, deepFM.py:268-275, under
 * linear_optimizer='Ftrl', :58): one elementwise pass instead of some twenty library kernels. */
int dir_ftrl_dense_f32(float* w, float* accum, float* linear, const float* grad, int64_t count, float lr, float l1, float l2,
                       dir_stream_t stream);

/* Adagrad on a dense variable of `count` elements with a dense gradient (dnn_optimizer='Adagrad', deepFM.py:61, on the hidden layers'
 * kernels and biases; [TF-upstream] tf.train.AdagradOptimizer): accum += g^2; w -= lr * g / (sqrt(accum) + eps) (eps = 0 is TensorFlow's
 * rule) -- one elementwise pass per variable. */
int dir_adagrad_dense_f32(float* w, float* accum, const float* grad, int64_t count, float lr, float eps, dir_stream_t stream);
/* dir_adagrad_dense_f32 on n_vars variables in one launch per 16 of them: w / accum / grad / count are HOST arrays of n_vars DEVICE pointers /
 * element counts (they travel in the kernel arguments; nothing is copied to the device). */
int dir_adagrad_dense_multi_f32(float* const* w, float* const* accum, const float* const* grad, const int64_t* count, int n_vars, float lr,
                                float eps, dir_stream_t stream);

/* The same two updates taking their sorted (row, entry) pairs from the workspace of an EARLIER sorted update of the same entries on the
 * same stream (same ids / strides / B / F and the same vocabularies, i.e. equal row_base and total_rows; sorted_from = that call's
 * 256-byte aligned workspace, still intact): the key building and the radix sort are skipped.  One training step of a DeepFM updates
 * the embedding tables (Adagrad, deepFM.py:61) and the linear columns (FTRL, :58) from the same ids; results are bit-identical to the
 * calls that sort themselves. */
int dir_sparse_adagrad_sorted_rows_from_f32(float* const* tables, float* const* accums, int64_t row_ld, int F, int K,
                                            const int64_t* ids, int64_t stride_b, int64_t stride_f, const float* grad,
                                            int64_t grad_ld, const float* fm_g, const float* fm_sum, float lr, int64_t B,
                                            const int64_t* row_base, int64_t total_rows, void* workspace,
                                            int64_t workspace_bytes, const void* sorted_from, dir_stream_t stream);
int dir_sparse_ftrl_sorted_from_f32(float* const* tables, float* const* accums, float* const* linears, int F, int K,
                                    const int64_t* ids, int64_t stride_b, int64_t stride_f, const float* grad, int64_t grad_ld,
                                    int64_t grad_slot_stride, float lr, float l1, float l2, int64_t B, const int64_t* row_base,
                                    int64_t total_rows, void* workspace, int64_t workspace_bytes, const void* sorted_from,
                                    dir_stream_t stream);

/* The first-order weights as packed linear TRAINING rows: rows[f] is [vocab_f, 4] floats = [w | n | z | unused], 16-byte aligned -- the
 * weight (linear_model's column, deepFM.py:258-263) beside its FTRL state (linear_optimizer='Ftrl', deepFM.py:58), so that the update of a
 * touched id reads and writes ONE 16-byte row instead of three 4-byte elements of three arrays (each of them a memory slot of its own).
 *   dir_linear_onehot_rows_f32       the forward: out[b] (+)= bias + sum_f rows[f][ids[b, f] * row_ld] (ids outside [0, vocab_f) pruned),
 *                                    the arithmetic and order of dir_linear_sparse_sum_f32's one-hot case, bit for bit;
 *   dir_sparse_ftrl_rows_sorted_f32  dir_sparse_ftrl_sorted_f32 / _from_f32 (sorted_from optional) on those rows, units = 1: the same bits
 *                                    in w, n and z as the three-array form. */
int dir_linear_onehot_rows_f32(const float* const* rows, int64_t row_ld, const int64_t* vocab, int F, const int64_t* ids,
                               int64_t stride_b, int64_t stride_f, const float* bias, int accumulate, int64_t B, float* out,
                               dir_stream_t stream);
int dir_sparse_ftrl_rows_sorted_f32(float* const* rows, int F, const int64_t* ids, int64_t stride_b, int64_t stride_f,
                                    const float* grad, int64_t grad_ld, int64_t grad_slot_stride, float lr, float l1, float l2,
                                    int64_t B, const int64_t* row_base, int64_t total_rows, void* workspace,
                                    int64_t workspace_bytes, const void* sorted_from, dir_stream_t stream);

/* Diagnostic only (never on the product path): cycle stamps of the DIR_CIN_STAMP=1 build of the CIN kernel, summed
 * over waves since the last call: [0] chunk start -> end of its MFMA stream, [1] -> past the chunk barrier,
 * [2] chunks, [3] prologue, [4] epilogue, [5] waves.  Synchronises the device. */
int dir_debug_cin_stamps(unsigned long long* out8);

/* Diagnostic only: the streaming ceilings of this box, for bench.py's roofline (measured in the run they are compared in).
 * dir_debug_stream_read_f32: a pure linear read of n floats (16 B per lane per load, non-temporal; sink is written only if the
 * sum takes a value it cannot take).  dir_debug_stream_copy_f32: q[i] = p[i] the same way (n floats read + n written).
 * n a multiple of 4, pointers 16-byte aligned. */
int dir_debug_stream_read_f32(const float* p, int64_t n, float* sink, dir_stream_t stream);
int dir_debug_stream_copy_f32(const float* p, float* q, int64_t n, dir_stream_t stream);

/* The in-tree stable LSD radix sort of (uint32 key, uint32 value) pairs behind every sorted sparse update (csrc/radix_sort.hip; the
 * reference's optimisers on the embedding tables: models/DeepFM/deepFM.py:58,61), exposed for the parity tests: sorts n < 2^30 pairs
 * by the low `bits` (1..32) bits of the key, stable (equal keys keep their input order), into keys_out / vals_out; keys_in / vals_in
 * are clobbered.  Kernel launches only -- no memset / memcpy nodes -- so the call can sit inside a HIP-graph capture.
 * workspace: dir_debug_radix_sort_workspace_bytes(n, bits) device bytes, any content. */
int64_t dir_debug_radix_sort_workspace_bytes(int64_t n, int bits);
int dir_debug_radix_sort_pairs_u32(uint32_t* keys_in, uint32_t* vals_in, int64_t n, int bits, uint32_t* keys_out, uint32_t* vals_out,
                                   void* workspace, int64_t workspace_bytes, dir_stream_t stream);

/* The slot-major form of that sort, which the sorted sparse updates take for one-hot entries ids [B, F] with B >= 4096: slot f's entries
 * are written as one segment (the slot is known from the entry's position: no sort pass for it) and every segment is sorted by its LOCAL
 * row, 10 bits per launch -- two launches for a 10^6-row vocabulary where the 25-bit global keys needed three, look-backs inside a segment.
 * -> keys_out = global rows row_base[f] + id (total_rows for a pruned id: id < 0 or >= slot f's vocabulary), vals_out = entries b F + f,
 * ordered by (slot, local row with pruned ids last, b).  row_base: [F] device, total_rows < 2^32 - 1.  workspace:
 * dir_debug_slot_sort_workspace_bytes device bytes (0: the shape is not covered, the updates sort global keys), any content. */
int64_t dir_debug_slot_sort_workspace_bytes(int64_t B, int F, int64_t total_rows);
int dir_debug_slot_sort_entries(const int64_t* ids, int64_t stride_b, int64_t stride_f, int F, int64_t B, const int64_t* row_base,
                                int64_t total_rows, uint32_t* keys_out, uint32_t* vals_out, void* workspace, int64_t workspace_bytes,
                                dir_stream_t stream);

/* ---- round 5: TRAINING of the DIN unit with PReLU / Dice activations, and of PReLU / Dice layers in general (csrc/din_rows_train.hip) --------
 * NO REFERENCE CODE (README.md:27 -> arXiv:1706.06978; Dice: section 5.3).  Dice in TRAIN mode normalises with the mini-batch's statistics over
 * all valid (sample, position) rows, so the unit trains as a chain of whole-batch passes over the compact row list n = (b, j) in (b, j)
 * order (N rows; row_off [B] = first row of sample b; b_idx [N] = its sample; ids_h [N] = its history id): these entries + the dense /
 * weight-gradient / batch-norm entries above.  Everything bitwise reproducible (fixed summation orders).
 *
 * dir_din_feat_rows_f32: X [N, 3K] = [h_n | h_n * a_b | a_b] (h_n = table[ids_h[n]], a_b = table[cand[b]], zeros for cand < 0) and Hc [N, K] = h_n:
 *   [h, a, h - a, h * a] W1 + b1 = X [Wh + Wd; Wp; Wa - Wd] + b1.  K a multiple of 4, <= 256.
 * dir_din_feat_rows_backward_f32: from dX [N, 3K] and dH [N, K]: grows [N + B, K], rows 0..N-1 = dL/dh_n = dX[n, 0:K] + dX[n, K:2K] * a_b + dH[n],
 *   rows N.. = dL/da_b = sum over the sample's rows (in order) of dX[n, K:2K] * h_n + dX[n, 2K:3K]  (0 for a pruned candidate).
 * dir_act_rows_train_f32: y = f(s) out of place, s [M, N] (N % 4 == 0, <= 1024): activation DIR_DIN_ACT_PRELU  y = s > 0 ? s : alpha s;
 *   DIR_DIN_ACT_DICE  y = s (alpha + (1 - alpha) sigmoid(scale s + shift)) with (scale, shift) = (rsqrt(var + eps), -mean scale) of the batch
 *   (dir_bn_train_stats_f32 without gamma / beta gives them and advances the moving statistics).
 * dir_act_rows_backward_f32: g = dL/dy -> d1 = g df/ds with the normalised pre-activation held fixed (PReLU: all of dL/ds); Dice: gx = dL/d(normalised
 *   pre-activation) = g s (1 - alpha) p (1 - p) -- dir_bn_train_backward_f32(gx, s, mean, inv) turns it into the statistics' share, to be
 *   added to d1; galpha [N] = sum over rows of g * (PReLU: min(s, 0); Dice: s (1 - p)).  partials: dir_act_rows_backward_partials(M, N) * N floats.
 * dir_din_pool_rows_f32: per sample w = scores of its rows (normalize: softmax of score / sqrt(K)), out [B, K] = sum_n w_n Hc[n]; w [N] is kept.
 * dir_din_pool_rows_backward_f32: g [B, K] -> ds [N] (through the softmax when normalize) and dH [N, K] = w_n g_b. */
int dir_din_feat_rows_f32(const float* table, int K, const int64_t* ids_h, const int64_t* b_idx, const int64_t* cand, int64_t N, float* X, float* Hc,
                          dir_stream_t stream);
int dir_din_feat_rows_backward_f32(const float* table, int K, const int64_t* ids_h, const int64_t* row_off, const int64_t* cand, int64_t B, int64_t N,
                                   const float* dX, const float* dH, float* grows, dir_stream_t stream);
int dir_act_rows_train_f32(const float* s, int64_t s_ld, int64_t M, int N, int activation, const float* alpha, const float* scale, const float* shift,
                           float* y, int64_t y_ld, dir_stream_t stream);
int dir_act_rows_backward_partials(int64_t M, int N);
int dir_act_rows_backward_f32(const float* g, int64_t g_ld, const float* s, int64_t s_ld, int64_t M, int N, int activation, const float* alpha,
                              const float* scale, const float* shift, float* d1, int64_t d1_ld, float* gx, int64_t gx_ld, float* galpha,
                              float* partials, int n_partials, dir_stream_t stream);
int dir_din_pool_rows_f32(const float* scores, const float* Hc, int K, const int64_t* row_off, int64_t B, int64_t N, int normalize, float* w, float* out,
                          dir_stream_t stream);
int dir_din_pool_rows_backward_f32(const float* g, const float* Hc, int K, const float* w, const int64_t* row_off, int64_t B, int64_t N, int normalize,
                                   float* ds, float* dH, dir_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* DIR_HIP_H_ */
