/*
 * dir_oracle.c -- CPU restatement of the reference's embedding-lookup + feature-interaction path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under oracle/ is part of the product: only tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this file's library, and only
 * as the checker / the reported CPU baseline.  The product path (details-in-recommendation_amd/)
 * never imports it and has no CPU fallback.
 *
 * PARITY UNPINNED.  The reference (yinyajun/Details-In-Recommendation) holds no tests, golden
 * vectors or fixtures for this path (models/DeepFM/test01.py, test02.py only print), and its
 * arithmetic lives in TensorFlow 1.x, which is neither vendored nor installable here, so the
 * reference cannot be executed to produce vectors.  Each function below follows the cited reference
 * lines; "[TF-upstream]" marks semantics of TensorFlow 1.x library code the reference calls
 * (restated from knowledge of the r1.10-r1.13 sources, not checkable in this container).  What pins
 * this file: the hand known-answer tests derivable from the cited lines (tests/test_oracle_kat.py)
 * and an independent NumPy restatement (oracle/np_ref.py) that must agree with it.
 * DIN and CIN have NO reference code (README.md:27-28 are table rows); they restate the papers.
 *
 * Build: gcc -O2 -ffp-contract=off -fopenmp -shared -fPIC  (strict IEEE fp32: no contraction, no
 * fast-math), see oracle/Makefile.  All pointers are host pointers.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

enum { ORC_SUM = 0, ORC_MEAN = 1, ORC_SQRTN = 2 };
enum { ORC_PRUNE_NONPOSITIVE_WEIGHTS = 1 };

int orc_version(void) { return 100; }

/* ------------------------------------------------------------------------------------------------
 * Embedding bag.  Follows:
 *   myself_input_layer: per column, column._get_dense_tensor(...) reshaped to [B, num_elements]
 *       models/DeepFM/deepFM.py:383-393
 *   tf.feature_column.input_layer(...)            models/DeepCrossNetwork/DeepCrossNetwork.py:126
 *   [TF-upstream] _EmbeddingColumn._get_dense_tensor -> safe_embedding_lookup_sparse:
 *       ids < 0 pruned; empty bag -> zero vector; embedding_lookup_sparse: rows scaled by weight,
 *       segment-summed IN ENTRY ORDER, then mean: / sum(w), sqrtn: / sqrt(sum(w^2));
 *       without weights: sparse_segment_{sum,mean,sqrt_n}: sum / count, sum / sqrt(count).
 * Addressing is that of include/dir_hip.h (dir_embedding_bag_f32).
 * ---------------------------------------------------------------------------------------------- */
/* Extended form (include/dir_hip.h: dir_embedding_bag_ex_f32): vocab [F] or NULL (id >= vocab_f pruned like id < 0),
 * slot_combiner [F] or NULL (one combiner per column: every tf.feature_column.embedding_column carries its own,
 * models/DeepCrossNetwork/train.py:99), max_norm > 0: [TF-upstream] embedding_lookup(max_norm=) -> clip_ops.clip_by_norm
 * (r1.10+): row * max_norm / max(l2norm, max_norm) with l2norm = sqrt(sum_k row_k^2), k ascending, 0 when the sum is 0;
 * applied to each looked-up row before it is weighted. */
static void orc_clip_row(const float* row, int K, float max_norm, float* dst) {
    float l2sum = 0.0f;
    for (int k = 0; k < K; ++k) l2sum = l2sum + row[k] * row[k];
    float l2norm = l2sum > 0.0f ? sqrtf(l2sum) : l2sum;
    float den = l2norm > max_norm ? l2norm : max_norm;
    for (int k = 0; k < K; ++k) dst[k] = (row[k] * max_norm) / den;
}

/* ex2: slot_max_norm [F] or NULL -- one max_norm per column (an embedding_column carries its own, like its combiner); entry 0 = none. */
int orc_embedding_bag_ex2_f32(const float* const* tables, const int64_t* vocab, int F, int K, const int64_t* ids,
                              const int64_t* offsets, const float* weights, int64_t stride_b, int64_t stride_f,
                              const int32_t* slot_combiner, int combiner, const float* slot_max_norm, float max_norm_all, int flags,
                              int64_t B, float* out, int64_t out_ld) {
    if (!tables || !ids || !out || F <= 0 || K <= 0 || B < 0 || K > 4096) return -1;
#pragma omp parallel for schedule(static)
    for (int64_t b = 0; b < B; ++b) {
        float clipped[4096];
        for (int f = 0; f < F; ++f) {
            float* o = out + b * out_ld + (int64_t)f * K;
            const float* tab = tables[f];
            const int comb = slot_combiner ? slot_combiner[f] : combiner;
            const float max_norm = slot_max_norm ? slot_max_norm[f] : max_norm_all;
            const int64_t vf = vocab ? vocab[f] : INT64_MAX;
            for (int k = 0; k < K; ++k) o[k] = 0.0f;
            int64_t bag = b * stride_b + f * stride_f;
            int64_t beg = offsets ? offsets[bag] : bag, end = offsets ? offsets[bag + 1] : bag + 1; /* one-hot: a bag of one entry */
            if (!offsets && !(max_norm > 0.0f)) { /* weight 1, no clipping: every combiner is the identity */
                int64_t id = ids[bag];
                if (id >= 0 && id < vf) memcpy(o, tab + id * K, sizeof(float) * (size_t)K);
                continue;
            }
            float wsum = 0.0f, w2sum = 0.0f;
            int64_t cnt = 0;
            for (int64_t e = beg; e < end; ++e) {
                int64_t id = ids[e];
                if (id < 0 || id >= vf) continue;
                float w = weights ? weights[e] : 1.0f;
                if (weights && (flags & ORC_PRUNE_NONPOSITIVE_WEIGHTS) && !(w > 0.0f)) continue;
                const float* row = tab + id * K;
                if (max_norm > 0.0f) {
                    orc_clip_row(row, K, max_norm, clipped);
                    row = clipped;
                }
                if (weights) {
                    for (int k = 0; k < K; ++k) o[k] = o[k] + w * row[k];
                } else {
                    for (int k = 0; k < K; ++k) o[k] = o[k] + row[k];
                }
                wsum = wsum + w;
                w2sum = w2sum + w * w;
                ++cnt;
            }
            if (cnt == 0) continue; /* empty bag -> zeros */
            if (comb == ORC_MEAN) {
                float den = weights ? wsum : (float)cnt;
                for (int k = 0; k < K; ++k) o[k] = o[k] / den;
            } else if (comb == ORC_SQRTN) {
                float den = weights ? sqrtf(w2sum) : sqrtf((float)cnt);
                for (int k = 0; k < K; ++k) o[k] = o[k] / den;
            }
        }
    }
    return 0;
}

int orc_embedding_bag_ex_f32(const float* const* tables, const int64_t* vocab, int F, int K, const int64_t* ids,
                             const int64_t* offsets, const float* weights, int64_t stride_b, int64_t stride_f,
                             const int32_t* slot_combiner, int combiner, float max_norm, int flags, int64_t B,
                             float* out, int64_t out_ld) {
    return orc_embedding_bag_ex2_f32(tables, vocab, F, K, ids, offsets, weights, stride_b, stride_f, slot_combiner, combiner, NULL, max_norm,
                                     flags, B, out, out_ld);
}

int orc_embedding_bag_f32(const float* const* tables, int F, int K, const int64_t* ids,
                          const int64_t* offsets, const float* weights, int64_t stride_b,
                          int64_t stride_f, int combiner, int flags, int64_t B, float* out,
                          int64_t out_ld) {
    return orc_embedding_bag_ex_f32(tables, NULL, F, K, ids, offsets, weights, stride_b, stride_f, NULL, combiner, 0.0f, flags, B,
                                    out, out_ld);
}

/* ------------------------------------------------------------------------------------------------
 * FM second-order term.  Follows fm_logit_fn, models/DeepFM/deepFM.py:329-334:
 *   embeddings = reshape(net, (-1, F, K))                                   :329
 *   summed_squared = square(reduce_sum(embeddings, -2))                     :331
 *   squared_summed = reduce_sum(square(embeddings), -2)                     :332
 *   logits = 0.5 * reduce_sum(subtract(summed_squared, squared_summed), -1) :333
 * acc64 = 0: fp32, f ascending then k ascending.  acc64 = 1: double accumulators.
 * ---------------------------------------------------------------------------------------------- */
int orc_fm_second_order_f32(const float* emb, int64_t emb_ld, int64_t B, int F, int K, float* out,
                            int acc64) {
    if (!emb || !out || F <= 0 || K <= 0 || B < 0) return -1;
#pragma omp parallel for schedule(static)
    for (int64_t b = 0; b < B; ++b) {
        const float* e = emb + b * emb_ld;
        if (acc64) {
            double tot = 0.0;
            for (int k = 0; k < K; ++k) {
                double s = 0.0, q = 0.0;
                for (int f = 0; f < F; ++f) {
                    double v = e[f * K + k];
                    s += v;
                    q += v * v;
                }
                tot += s * s - q;
            }
            out[b] = (float)(0.5 * tot);
        } else {
            float tot = 0.0f;
            for (int k = 0; k < K; ++k) {
                float s = 0.0f, q = 0.0f;
                for (int f = 0; f < F; ++f) {
                    float v = e[f * K + k];
                    s = s + v;
                    q = q + v * v;
                }
                tot = tot + (s * s - q);
            }
            out[b] = 0.5f * tot;
        }
    }
    return 0;
}

/* ------------------------------------------------------------------------------------------------
 * First-order (linear) term.  Follows _linear_logit_fn_builder, models/DeepFM/deepFM.py:255-275:
 *   logits = linear_model(features, columns, units, sparse_combiner)        :258-263
 *   [TF-upstream] per categorical column a [vocab, units] weight looked up with the bag combiner
 *   (default 'sum'), the per-column results added in column order, then + bias.
 * units = 1.  out[b] = (accumulate ? out[b] : 0) + bias + sum_f combine(bag(b,f)).
 * ---------------------------------------------------------------------------------------------- */
int orc_linear_sparse_sum_f32(const float* const* wts, int F, const int64_t* ids,
                              const int64_t* offsets, const float* entry_weights, int64_t stride_b,
                              int64_t stride_f, int combiner, const float* bias, int accumulate,
                              int64_t B, float* out) {
    if (!wts || !ids || !out || F <= 0 || B < 0) return -1;
#pragma omp parallel for schedule(static)
    for (int64_t b = 0; b < B; ++b) {
        float acc = 0.0f;
        for (int f = 0; f < F; ++f) {
            float v = 0.0f;
            if (!offsets) {
                int64_t id = ids[b * stride_b + f * stride_f];
                if (id >= 0) v = wts[f][id];
            } else {
                int64_t bag = b * stride_b + f * stride_f;
                float wsum = 0.0f, w2sum = 0.0f;
                int64_t cnt = 0;
                for (int64_t e = offsets[bag]; e < offsets[bag + 1]; ++e) {
                    int64_t id = ids[e];
                    if (id < 0) continue;
                    float w = entry_weights ? entry_weights[e] : 1.0f;
                    v = entry_weights ? v + w * wts[f][id] : v + wts[f][id];
                    wsum = wsum + w;
                    w2sum = w2sum + w * w;
                    ++cnt;
                }
                if (cnt > 0 && combiner == ORC_MEAN) v = v / (entry_weights ? wsum : (float)cnt);
                if (cnt > 0 && combiner == ORC_SQRTN)
                    v = v / (entry_weights ? sqrtf(w2sum) : sqrtf((float)cnt));
            }
            acc = acc + v;
        }
        float r = acc + (bias ? bias[0] : 0.0f);
        out[b] = accumulate ? out[b] + r : r;
    }
    return 0;
}

/* ------------------------------------------------------------------------------------------------
 * DCN cross network.  Follows models/DeepCrossNetwork/DeepCrossNetwork.py:
 *   _cross_op:  x_w = tensordot(x, w, axes=1)                                :345
 *               y = x0 * expand_dims(x_w, -1) + b + x                        :346
 *   _cross_architecture: xl = x0; for l in range(L): xl = _cross_op(x0, xl, w[l], b[l])  :361-365
 * acc64 selects the accumulator of the dot product (fp32 j-ascending, or double).
 * ---------------------------------------------------------------------------------------------- */
int orc_dcn_cross_f32(const float* x0, int64_t x_ld, const float* w, const float* b, int L,
                      int64_t B, int d, float* out, int64_t out_ld, int acc64) {
    if (!x0 || !w || !b || !out || L < 0 || d <= 0 || B < 0) return -1;
#pragma omp parallel for schedule(static)
    for (int64_t r = 0; r < B; ++r) {
        const float* x = x0 + r * x_ld;
        float* xl = out + r * out_ld;
        for (int j = 0; j < d; ++j) xl[j] = x[j];
        for (int l = 0; l < L; ++l) {
            const float* wl = w + (int64_t)l * d;
            const float* bl = b + (int64_t)l * d;
            float xw;
            if (acc64) {
                double s = 0.0;
                for (int j = 0; j < d; ++j) s += (double)xl[j] * (double)wl[j];
                xw = (float)s;
            } else {
                float s = 0.0f;
                for (int j = 0; j < d; ++j) s = s + xl[j] * wl[j];
                xw = s;
            }
            for (int j = 0; j < d; ++j) xl[j] = ((x[j] * xw) + bl[j]) + xl[j];
        }
    }
    return 0;
}

/* ------------------------------------------------------------------------------------------------
 * DIN local activation unit + pooling.  NO REFERENCE CODE (README.md:27 -> arXiv:1706.06978).
 * Definition (include/dir_hip.h):
 *   u = [h, a, h-a, h*a] (4K);  z1 = sigmoid(W1^T u + b1);  z2 = sigmoid(W2^T z1 + b2);
 *   s = W3.z2 + b3;  normalize=0: w_j = s_j;  normalize=1: w = softmax_valid(s / sqrt(K));
 *   out = sum_j w_j h_j.
 * ---------------------------------------------------------------------------------------------- */
static float orc_sigmoid(float x) { return 1.0f / (1.0f + expf(-x)); }

/* the unit's hidden activation (paper-derived, arXiv:1706.06978 section 5.3; no reference code): 0 sigmoid, 1 PReLU f(s) = s > 0 ? s : alpha s,
 * 2 Dice (inference form) f(s) = p s + (1 - p) alpha s, p = sigmoid(scale s + shift).  ap = {alpha, scale, shift} rows of n entries. */
static float orc_din_act(float s, int act, const float* ap, int n, int h) {
    if (act == 1) return s > 0.0f ? s : ap[h] * s;
    if (act == 2) {
        const float pg = orc_sigmoid(ap[n + h] * s + ap[2 * n + h]);
        return pg * s + (1.0f - pg) * ap[h] * s;
    }
    return orc_sigmoid(s);
}

int orc_din_attention_pool_act_f32(const float* table, int K, const int64_t* hist, const int32_t* hist_len, const int64_t* cand, int T,
                                   const float* W1, const float* b1, int H1, const float* W2, const float* b2, int H2, const float* W3,
                                   const float* b3, int normalize, int activation, const float* act_params, int64_t B, float* out,
                                   float* scores, int acc64);

int orc_din_attention_pool_f32(const float* table, int K, const int64_t* hist,
                               const int32_t* hist_len, const int64_t* cand, int T,
                               const float* W1, const float* b1, int H1, const float* W2,
                               const float* b2, int H2, const float* W3, const float* b3,
                               int normalize, int64_t B, float* out, float* scores, int acc64) {
    return orc_din_attention_pool_act_f32(table, K, hist, hist_len, cand, T, W1, b1, H1, W2, b2, H2, W3, b3, normalize, 0, NULL, B, out, scores, acc64);
}

int orc_din_attention_pool_act_f32(const float* table, int K, const int64_t* hist, const int32_t* hist_len, const int64_t* cand, int T,
                                   const float* W1, const float* b1, int H1, const float* W2, const float* b2, int H2, const float* W3,
                                   const float* b3, int normalize, int activation, const float* act_params, int64_t B, float* out,
                                   float* scores, int acc64) {
    if (!table || !hist || !cand || !W1 || !b1 || !W2 || !b2 || !W3 || !b3 || !out) return -1;
    if (K <= 0 || T <= 0 || H1 <= 0 || H2 <= 0) return -1;
    if (activation < 0 || activation > 2 || (activation != 0 && !act_params)) return -1;
    const float* ap1 = act_params;
    const float* ap2 = act_params ? act_params + 3 * (size_t)H1 : NULL;
#pragma omp parallel
    {
        float* u = (float*)malloc(sizeof(float) * (size_t)(4 * K));
        float* z1 = (float*)malloc(sizeof(float) * (size_t)H1);
        float* z2 = (float*)malloc(sizeof(float) * (size_t)H2);
        float* s = (float*)malloc(sizeof(float) * (size_t)T);
        char* valid = (char*)malloc((size_t)T);
#pragma omp for schedule(static)
        for (int64_t b = 0; b < B; ++b) {
            int len = hist_len ? hist_len[b] : T;
            if (len > T) len = T;
            const float* a = cand[b] >= 0 ? table + cand[b] * K : NULL;
            for (int j = 0; j < T; ++j) {
                int64_t id = hist[b * T + j];
                valid[j] = (j < len && id >= 0);
                s[j] = 0.0f;
                if (!valid[j]) continue;
                const float* h = table + id * K;
                for (int k = 0; k < K; ++k) {
                    float av = a ? a[k] : 0.0f;
                    u[k] = h[k];
                    u[K + k] = av;
                    u[2 * K + k] = h[k] - av;
                    u[3 * K + k] = h[k] * av;
                }
                for (int n = 0; n < H1; ++n) {
                    if (acc64) {
                        double acc = 0.0;
                        for (int i = 0; i < 4 * K; ++i) acc += (double)u[i] * (double)W1[i * H1 + n];
                        z1[n] = orc_din_act((float)(acc + (double)b1[n]), activation, ap1, H1, n);
                    } else {
                        float acc = 0.0f;
                        for (int i = 0; i < 4 * K; ++i) acc = acc + u[i] * W1[i * H1 + n];
                        z1[n] = orc_din_act(acc + b1[n], activation, ap1, H1, n);
                    }
                }
                for (int n = 0; n < H2; ++n) {
                    if (acc64) {
                        double acc = 0.0;
                        for (int i = 0; i < H1; ++i) acc += (double)z1[i] * (double)W2[i * H2 + n];
                        z2[n] = orc_din_act((float)(acc + (double)b2[n]), activation, ap2, H2, n);
                    } else {
                        float acc = 0.0f;
                        for (int i = 0; i < H1; ++i) acc = acc + z1[i] * W2[i * H2 + n];
                        z2[n] = orc_din_act(acc + b2[n], activation, ap2, H2, n);
                    }
                }
                if (acc64) {
                    double acc = 0.0;
                    for (int i = 0; i < H2; ++i) acc += (double)z2[i] * (double)W3[i];
                    s[j] = (float)(acc + (double)b3[0]);
                } else {
                    float acc = 0.0f;
                    for (int i = 0; i < H2; ++i) acc = acc + z2[i] * W3[i];
                    s[j] = acc + b3[0];
                }
            }
            if (normalize) {
                float scale = 1.0f / sqrtf((float)K);
                float mx = -INFINITY;
                for (int j = 0; j < T; ++j)
                    if (valid[j] && s[j] * scale > mx) mx = s[j] * scale;
                double den = 0.0;
                for (int j = 0; j < T; ++j)
                    if (valid[j]) {
                        s[j] = expf(s[j] * scale - mx);
                        den += s[j];
                    }
                for (int j = 0; j < T; ++j)
                    if (valid[j]) s[j] = (float)(s[j] / den);
            }
            float* o = out + b * K;
            for (int k = 0; k < K; ++k) {
                if (acc64) {
                    double acc = 0.0;
                    for (int j = 0; j < T; ++j)
                        if (valid[j]) acc += (double)s[j] * (double)table[hist[b * T + j] * K + k];
                    o[k] = (float)acc;
                } else {
                    float acc = 0.0f;
                    for (int j = 0; j < T; ++j)
                        if (valid[j]) acc = acc + s[j] * table[hist[b * T + j] * K + k];
                    o[k] = acc;
                }
            }
            if (scores)
                for (int j = 0; j < T; ++j) scores[b * T + j] = valid[j] ? s[j] : 0.0f;
        }
        free(u); free(z1); free(z2); free(s); free(valid);
    }
    return 0;
}

/* ------------------------------------------------------------------------------------------------
 * xDeepFM CIN layer.  NO REFERENCE CODE (README.md:28 -> arXiv:1803.05170, eq. 6).
 *   xout[b,h,d] = sum_{i<Hp} sum_{j<m} W[h, i*m+j] * xk[b,i,d] * x0[b,j,d]
 *   pooled[b,h] = sum_d xout[b,h,d]
 * acc64 = 0: fp32, (i,j) ascending, product (xk*x0) rounded first then * W accumulated.
 * ---------------------------------------------------------------------------------------------- */
int orc_cin_layer_f32(const float* x0, const float* xk, const float* W, int m, int Hp, int H, int D,
                      int64_t B, float* xout, float* pooled, int64_t pooled_ld, int acc64) {
    if (!x0 || !xk || !W || !xout || m <= 0 || Hp <= 0 || H <= 0 || D <= 0 || B < 0) return -1;
#pragma omp parallel for schedule(static)
    for (int64_t b = 0; b < B; ++b) {
        const float* X0 = x0 + b * (int64_t)m * D;
        const float* XK = xk + b * (int64_t)Hp * D;
        for (int h = 0; h < H; ++h) {
            const float* Wh = W + (int64_t)h * Hp * m;
            double p64 = 0.0;
            float p32 = 0.0f;
            for (int d = 0; d < D; ++d) {
                float r;
                if (acc64) {
                    double acc = 0.0;
                    for (int i = 0; i < Hp; ++i)
                        for (int j = 0; j < m; ++j)
                            acc += (double)Wh[i * m + j] * (double)XK[i * D + d] * (double)X0[j * D + d];
                    r = (float)acc;
                } else {
                    float acc = 0.0f;
                    for (int i = 0; i < Hp; ++i)
                        for (int j = 0; j < m; ++j)
                            acc = acc + (XK[i * D + d] * X0[j * D + d]) * Wh[i * m + j];
                    r = acc;
                }
                xout[(b * H + h) * (int64_t)D + d] = r;
                p64 += r;
                p32 = p32 + r;
            }
            if (pooled) pooled[b * pooled_ld + h] = acc64 ? (float)p64 : p32;
        }
    }
    return 0;
}

/* ------------------------------------------------------------------------------------------------
 * Backward of the CIN layer (no reference code; derivatives of the definition above), double accumulation:
 *   dW[h,i*m+j] = sum_{b,d} G[b,h,d] xk[b,i,d] x0[b,j,d]
 *   dxk[b,i,d]  = sum_{h,j} W[h,i*m+j] G[b,h,d] x0[b,j,d]
 *   dx0[b,j,d]  = sum_{h,i} W[h,i*m+j] G[b,h,d] xk[b,i,d]
 * dW is returned in double (dW64, caller rounds) so that tests can bound the fp32 reduction error.
 * ---------------------------------------------------------------------------------------------- */
int orc_cin_backward(const float* x0, const float* xk, const float* W, const float* G, int m, int Hp, int H, int D,
                     int64_t B, double* dW64, float* dxk, float* dx0) {
    if (!x0 || !xk || !W || !G || !dW64 || !dxk || !dx0 || m <= 0 || Hp <= 0 || H <= 0 || D <= 0 || B < 0) return -1;
    const int64_t Kd = (int64_t)Hp * m;
    for (int64_t e = 0; e < (int64_t)H * Kd; ++e) dW64[e] = 0.0;
#pragma omp parallel for schedule(static)
    for (int h = 0; h < H; ++h) {              /* each thread owns rows of dW: no races, fixed order over b,d */
        double* dWh = dW64 + (int64_t)h * Kd;
        for (int64_t b = 0; b < B; ++b) {
            const float* X0 = x0 + b * (int64_t)m * D;
            const float* XK = xk + b * (int64_t)Hp * D;
            const float* Gb = G + (b * H + h) * (int64_t)D;
            for (int d = 0; d < D; ++d) {
                const double g = Gb[d];
                for (int i = 0; i < Hp; ++i) {
                    const double gx = g * (double)XK[i * D + d];
                    for (int j = 0; j < m; ++j) dWh[i * m + j] += gx * (double)X0[j * D + d];
                }
            }
        }
    }
#pragma omp parallel for schedule(static)
    for (int64_t b = 0; b < B; ++b) {
        const float* X0 = x0 + b * (int64_t)m * D;
        const float* XK = xk + b * (int64_t)Hp * D;
        for (int d = 0; d < D; ++d) {
            for (int i = 0; i < Hp; ++i) {
                double a = 0.0;
                for (int h = 0; h < H; ++h) {
                    const double g = G[(b * H + h) * (int64_t)D + d];
                    const float* Wh = W + (int64_t)h * Kd + (int64_t)i * m;
                    double t = 0.0;
                    for (int j = 0; j < m; ++j) t += (double)Wh[j] * (double)X0[j * D + d];
                    a += g * t;
                }
                dxk[(b * Hp + i) * (int64_t)D + d] = (float)a;
            }
            for (int j = 0; j < m; ++j) {
                double a = 0.0;
                for (int h = 0; h < H; ++h) {
                    const double g = G[(b * H + h) * (int64_t)D + d];
                    const float* Wh = W + (int64_t)h * Kd + j;
                    double t = 0.0;
                    for (int i = 0; i < Hp; ++i) t += (double)Wh[(int64_t)i * m] * (double)XK[i * D + d];
                    a += g * t;
                }
                dx0[(b * m + j) * (int64_t)D + d] = (float)a;
            }
        }
    }
    return 0;
}

/* ------------------------------------------------------------------------------------------------
 * Bucketized column. [TF-upstream] bucketized_column -> math_ops._bucketize: bucket = number of
 * boundaries <= x (boundaries ascending).  Docstring use: models/DeepFM/deepFM.py:95.
 * ---------------------------------------------------------------------------------------------- */
int orc_bucketize_f32(const float* x, int64_t n, const float* boundaries, int nb, int64_t* out) {
    if (!x || !out || nb < 0 || (nb > 0 && !boundaries)) return -1;
    for (int64_t i = 0; i < n; ++i) {
        int c = 0;
        for (int j = 0; j < nb; ++j)
            if (boundaries[j] <= x[i]) ++c;
        out[i] = c;
    }
    return 0;
}

/* ------------------------------------------------------------------------------------------------
 * Row sharding, 'div' rule.  Follows the partitioner at models/DeepFM/deepFM.py:163-167;
 * [TF-upstream] embedding_lookup partition_strategy='div': with q = V / P, r = V % P the first r
 * shards hold q+1 consecutive rows, the rest q.
 * ---------------------------------------------------------------------------------------------- */
void orc_shard_div_owner(int64_t id, int64_t vocab, int P, int* owner, int64_t* local) {
    int64_t q = vocab / P, r = vocab % P;
    int64_t thr = r * (q + 1);
    if (id < thr) {
        *owner = (int)(id / (q + 1));
        *local = id - (int64_t)(*owner) * (q + 1);
    } else {
        int64_t o = r + (q > 0 ? (id - thr) / q : 0);
        *owner = (int)o;
        *local = id - (thr + (o - r) * q);
    }
}
