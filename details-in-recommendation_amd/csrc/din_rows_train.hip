// din_rows_train.hip -- TRAINING of the DIN local activation unit with PReLU / Dice hidden activations, and of the model's PReLU / Dice MLP
// layers (round 5).  NO REFERENCE CODE (README.md:27 links arXiv:1706.06978, section 5.3 for Dice); definitions: include/dir_hip.h (A13),
// details-in-recommendation_amd/din.py (Dice, _PReLU), oracle/np_ref.py.
//
// The sigmoid unit trains through one fused forward / backward kernel pair (din_wave.hip, din_bwd_rows.hip).  Dice in TRAIN mode
// normalises with the statistics of the mini-batch over every valid (sample, position) row, so a layer's activation cannot be applied
// before ALL rows' pre-activations exist, and its backward carries the two batch-norm reductions: the unit becomes a chain of whole-batch
// passes over the compact row list n = (b, j) (N = sum of the history lengths), most of them the dense / weight-gradient kernels the towers
// already use.  This file holds what those do not cover:
//   din_feat_rows_k / din_feat_rows_bwd_k   X'[n] = [h_n | h_n * a_b | a_b] (so that [h, a, h-a, h*a] W1 + b1 = X' [Wh+Wd; Wp; Wa-Wd] + b1) and a
//                                           copy of h_n for the pooling; backward: the history rows' and the candidates' gradient rows
//   act_rows_train_k / act_rows_bwd_k       y = f(s) out of place for PReLU and Dice given (scale, shift) of THIS batch; backward: the direct
//                                           term, the gradient with respect to the normalised pre-activation (Dice: fed to the batch-norm
//                                           backward, csrc/bn_train.hip) and the per-unit gradient of alpha (block partials added in block order)
//   din_pool_rows_k / din_pool_rows_bwd_k   per sample: attention weights from the rows' scores (raw, or softmax of s / sqrt(K)) and the
//                                           weighted sum of its history rows; backward: ds and w_n * g_b
// All HBM-bound elementwise / segment passes; every reduction in a fixed order (bitwise reproducible).
#include "common.hpp"

namespace dir {

// ---- X' rows ------------------------------------------------------------------------------------------------------------------------------------
// one wave per row, lane = 16-byte piece of the K-wide row (K <= 256, K % 4 == 0); pruned candidate (cand < 0): a = 0
__global__ __launch_bounds__(256) void din_feat_rows_k(const float* __restrict__ table, int K, const int64_t* __restrict__ ids_h,
                                                        const int64_t* __restrict__ b_idx, const int64_t* __restrict__ cand, int64_t N,
                                                        float* __restrict__ X /* [N, 3K] */, float* __restrict__ Hc /* [N, K] */, int lpr) {
    // lpr lanes per row (16 for K <= 64: four rows per wave -- one row per wave left 48 of 64 lanes idle: 749 -> 389 us at 1.7 M rows)
    const int sub = threadIdx.x % lpr;
    const int64_t n = ((int64_t)blockIdx.x * 256 + threadIdx.x) / lpr;
    if (n >= N || 4 * sub >= K) return;
    const int64_t b = b_idx[n], c = cand[b];
    const float4 h = *reinterpret_cast<const float4*>(table + ids_h[n] * K + 4 * sub);
    const float4 a = c >= 0 ? *reinterpret_cast<const float4*>(table + c * K + 4 * sub) : make_float4(0.f, 0.f, 0.f, 0.f);
    float* x = X + n * 3 * K + 4 * sub;
    *reinterpret_cast<float4*>(x) = h;
    *reinterpret_cast<float4*>(x + K) = make_float4(h.x * a.x, h.y * a.y, h.z * a.z, h.w * a.w);
    *reinterpret_cast<float4*>(x + 2 * K) = a;
    *reinterpret_cast<float4*>(Hc + n * K + 4 * sub) = h;
}

// gh_n = dX'[n, 0:K] + dX'[n, K:2K] * a_b + dH_n;   ga_b = sum over the sample's rows of (dX'[n, K:2K] * h_n + dX'[n, 2K:3K])   (rows in order)
// one wave per SAMPLE (its rows are contiguous: row_off[b] .. row_off[b] + cnt), lane = 16-byte piece; grows = [gh (N rows) | ga (B rows)]
__global__ __launch_bounds__(256) void din_feat_rows_bwd_k(const float* __restrict__ table, int K, const int64_t* __restrict__ ids_h,
                                                            const int64_t* __restrict__ row_off, const int64_t* __restrict__ cand, int64_t B, int64_t N,
                                                            const float* __restrict__ dX, const float* __restrict__ dH, float* __restrict__ grows, int lpr) {
    const int sub = threadIdx.x % lpr;                                  // lpr lanes per SAMPLE (16 for K <= 64: four samples per wave)
    const int64_t b = ((int64_t)blockIdx.x * 256 + threadIdx.x) / lpr;
    if (b >= B || 4 * sub >= K) return;
    const int64_t n0 = row_off[b], n1 = b + 1 < B ? row_off[b + 1] : N, c = cand[b];
    const float4 a = c >= 0 ? *reinterpret_cast<const float4*>(table + c * K + 4 * sub) : make_float4(0.f, 0.f, 0.f, 0.f);
    float4 ga = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int64_t n = n0; n < n1; ++n) {
        const float* dx = dX + n * 3 * K + 4 * sub;
        const float4 d0 = *reinterpret_cast<const float4*>(dx), d1 = *reinterpret_cast<const float4*>(dx + K), d2 = *reinterpret_cast<const float4*>(dx + 2 * K);
        const float4 dh = *reinterpret_cast<const float4*>(dH + n * K + 4 * sub);
        const float4 h = *reinterpret_cast<const float4*>(table + ids_h[n] * K + 4 * sub);
        *reinterpret_cast<float4*>(grows + n * K + 4 * sub) =
            make_float4(fmaf(d1.x, a.x, d0.x) + dh.x, fmaf(d1.y, a.y, d0.y) + dh.y, fmaf(d1.z, a.z, d0.z) + dh.z, fmaf(d1.w, a.w, d0.w) + dh.w);
        ga.x += fmaf(d1.x, h.x, d2.x); ga.y += fmaf(d1.y, h.y, d2.y); ga.z += fmaf(d1.z, h.z, d2.z); ga.w += fmaf(d1.w, h.w, d2.w);
    }
    if (c < 0) ga = make_float4(0.f, 0.f, 0.f, 0.f);                 // a pruned candidate contributed the zero vector: no gradient
    *reinterpret_cast<float4*>(grows + (N + b) * K + 4 * sub) = ga;
}

// ---- PReLU / Dice over rows, training form ------------------------------------------------------------------------------------------------------
// ACT 1: y = s > 0 ? s : alpha s.   ACT 2: p = sigmoid(scale s + shift) (scale = rsqrt(var + eps), shift = -mean scale of THIS batch),
// y = s (alpha + (1 - alpha) p).
__device__ __forceinline__ float art_sigmoid(float x) { return 1.0f / (1.0f + __expf(-x)); }

template <int ACT>
__global__ __launch_bounds__(256) void act_rows_train_k(const float* __restrict__ s, int64_t s_ld, int64_t M, int N, const float* __restrict__ alpha,
                                                         const float* __restrict__ scale, const float* __restrict__ shift, float* __restrict__ y,
                                                         int64_t y_ld) {
    const int nv = N >> 2;
    const int64_t total = M * nv;
    for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < total; e += (int64_t)gridDim.x * 256) {
        const int64_t r = e / nv;
        const int c = (int)(e - r * nv) * 4;
        const float4 sv = *reinterpret_cast<const float4*>(s + r * s_ld + c);
        const float4 al = *reinterpret_cast<const float4*>(alpha + c);
        float4 o;
        if constexpr (ACT == 1) {
            o = make_float4(sv.x > 0.f ? sv.x : al.x * sv.x, sv.y > 0.f ? sv.y : al.y * sv.y, sv.z > 0.f ? sv.z : al.z * sv.z,
                            sv.w > 0.f ? sv.w : al.w * sv.w);
        } else {
            const float4 sc = *reinterpret_cast<const float4*>(scale + c), sh = *reinterpret_cast<const float4*>(shift + c);
            o.x = sv.x * fmaf(art_sigmoid(fmaf(sv.x, sc.x, sh.x)), 1.f - al.x, al.x);
            o.y = sv.y * fmaf(art_sigmoid(fmaf(sv.y, sc.y, sh.y)), 1.f - al.y, al.y);
            o.z = sv.z * fmaf(art_sigmoid(fmaf(sv.z, sc.z, sh.z)), 1.f - al.z, al.z);
            o.w = sv.w * fmaf(art_sigmoid(fmaf(sv.w, sc.w, sh.w)), 1.f - al.w, al.w);
        }
        *reinterpret_cast<float4*>(y + r * y_ld + c) = o;
    }
}

// backward, one pass over (g, s):  d1 = g * df/ds with the normalised pre-activation held fixed (PReLU: the whole derivative),
//   Dice: gx = g * s (1 - alpha) p (1 - p)  = dL/d(normalised pre-activation); the caller's batch-norm backward turns it into the statistics' share;
//   galpha partials: column sums of g * (PReLU: min(s, 0); Dice: s (1 - p)) over this block's rows (thread (c, rr): rows rr, rr + RPI, ..; LDS
//   tree over rr in a fixed order).  N <= 1024 (N % 4 == 0).
template <int ACT>
__global__ __launch_bounds__(256) void act_rows_bwd_k(const float* __restrict__ g, int64_t g_ld, const float* __restrict__ s, int64_t s_ld, int64_t M,
                                                       int N, const float* __restrict__ alpha, const float* __restrict__ scale,
                                                       const float* __restrict__ shift, int64_t rows_per_block, float* __restrict__ d1, int64_t d1_ld,
                                                       float* __restrict__ gx, int64_t gx_ld, float* __restrict__ part /* [blocks][N] */) {
    __shared__ float4 red[256];
    const int nv = N >> 2;                                   // <= 256
    const int tid = threadIdx.x;
    const int TPR = nv;                                      // threads per row
    const int RPI = 256 / TPR > 0 ? 256 / TPR : 1;           // rows per iteration
    const int c0 = tid % TPR, rr = tid / TPR;
    const bool active = rr < RPI;
    const int64_t r0 = (int64_t)blockIdx.x * rows_per_block, r1 = min(M, r0 + rows_per_block);
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    if (active) {
        const int c = 4 * c0;
        const float4 al = *reinterpret_cast<const float4*>(alpha + c);
        float4 sc = make_float4(0.f, 0.f, 0.f, 0.f), sh = sc;
        if constexpr (ACT == 2) { sc = *reinterpret_cast<const float4*>(scale + c); sh = *reinterpret_cast<const float4*>(shift + c); }
        for (int64_t r = r0 + rr; r < r1; r += RPI) {
            const float4 gv = *reinterpret_cast<const float4*>(g + r * g_ld + c), sv = *reinterpret_cast<const float4*>(s + r * s_ld + c);
            float4 o1, o2;
            const float gs[4] = {gv.x, gv.y, gv.z, gv.w}, ss[4] = {sv.x, sv.y, sv.z, sv.w}, aa[4] = {al.x, al.y, al.z, al.w};
            const float cs[4] = {sc.x, sc.y, sc.z, sc.w}, ch[4] = {sh.x, sh.y, sh.z, sh.w};
            float r1v[4], r2v[4], ra[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                if constexpr (ACT == 1) {
                    r1v[q] = gs[q] * (ss[q] > 0.f ? 1.f : aa[q]);
                    r2v[q] = 0.f;
                    ra[q] = gs[q] * fminf(ss[q], 0.f);
                } else {
                    const float p = art_sigmoid(fmaf(ss[q], cs[q], ch[q]));
                    r1v[q] = gs[q] * fmaf(p, 1.f - aa[q], aa[q]);
                    r2v[q] = gs[q] * ss[q] * (1.f - aa[q]) * p * (1.f - p);
                    ra[q] = gs[q] * ss[q] * (1.f - p);
                }
            }
            o1 = make_float4(r1v[0], r1v[1], r1v[2], r1v[3]);
            *reinterpret_cast<float4*>(d1 + r * d1_ld + c) = o1;
            if constexpr (ACT == 2) {
                o2 = make_float4(r2v[0], r2v[1], r2v[2], r2v[3]);
                *reinterpret_cast<float4*>(gx + r * gx_ld + c) = o2;
            }
            acc.x += ra[0]; acc.y += ra[1]; acc.z += ra[2]; acc.w += ra[3];
        }
    }
    red[tid] = active ? acc : make_float4(0.f, 0.f, 0.f, 0.f);
    __syncthreads();
    if (tid < TPR) {                                         // rows lanes added in rr order
        float4 t = red[tid];
        for (int q = 1; q < RPI; ++q) {
            const float4 v = red[q * TPR + tid];
            t.x += v.x; t.y += v.y; t.z += v.z; t.w += v.w;
        }
        *reinterpret_cast<float4*>(part + (int64_t)blockIdx.x * N + 4 * tid) = t;
    }
}

// galpha[c] = sum of part[block][c] over the blocks: 16 columns per workgroup, thread (q, c) adds blocks q, q + 16, .. in order (fp64), the 16
// partial sums are added in q order -- a fixed order.  (One thread walking all 2048 blocks of a column: 650 us per call.)
__global__ __launch_bounds__(256) void act_rows_alpha_fin_k(const float* __restrict__ part, int nblocks, int N, float* __restrict__ galpha) {
    __shared__ double red[16][16];
    const int cl = threadIdx.x & 15, q = threadIdx.x >> 4;
    const int c = blockIdx.x * 16 + cl;
    double t = 0.0;
    if (c < N)
        for (int b = q; b < nblocks; b += 16) t += (double)part[(int64_t)b * N + c];
    red[q][cl] = t;
    __syncthreads();
    if (q == 0 && c < N) {
        double u = red[0][cl];
#pragma unroll
        for (int k = 1; k < 16; ++k) u += red[k][cl];
        galpha[c] = (float)u;
    }
}

// ---- pooling over a sample's rows ------------------------------------------------------------------------------------------------------------------
// one wave per sample; lane = 16-byte piece of the K-wide row (K <= 256).  normalize: w = softmax over the sample's rows of s / sqrt(K); else w = s.
// An empty sample: zeros.  w is written for the backward.
__global__ __launch_bounds__(256) void din_pool_rows_k(const float* __restrict__ sc, const float* __restrict__ Hc, int K, const int64_t* __restrict__ row_off,
                                                        int64_t B, int64_t N, int normalize, float inv_sqrt_k, float* __restrict__ w,
                                                        float* __restrict__ out, int lpr) {
    const int sub = threadIdx.x % lpr;                                  // lpr lanes per sample (16 for K <= 64: four samples per wave)
    const int64_t b = ((int64_t)blockIdx.x * 256 + threadIdx.x) / lpr;
    const bool live = b < B;
    const int64_t n0 = live ? row_off[b] : 0, n1 = live ? (b + 1 < B ? row_off[b + 1] : N) : 0;
    float mx = -INFINITY, inv_l = 1.f;
    if (normalize) {                                                    // (the group's shuffles: every lane of the wave takes part, dead groups with empty ranges)
        for (int64_t n = n0 + sub; n < n1; n += lpr) mx = fmaxf(mx, sc[n] * inv_sqrt_k);
        for (int o = lpr >> 1; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o, 64));
        float l = 0.f;
        for (int64_t n = n0 + sub; n < n1; n += lpr) l += __expf(sc[n] * inv_sqrt_k - mx);
        for (int o = lpr >> 1; o > 0; o >>= 1) l += __shfl_xor(l, o, 64);
        inv_l = l > 0.f ? 1.f / l : 0.f;
    }
    if (!live) return;
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    const bool col = 4 * sub < K;
    for (int64_t n = n0; n < n1; ++n) {
        const float wn = normalize ? __expf(sc[n] * inv_sqrt_k - mx) * inv_l : sc[n];
        if (sub == 0) w[n] = wn;
        if (col) {
            const float4 h = *reinterpret_cast<const float4*>(Hc + n * K + 4 * sub);
            acc.x = fmaf(wn, h.x, acc.x); acc.y = fmaf(wn, h.y, acc.y); acc.z = fmaf(wn, h.z, acc.z); acc.w = fmaf(wn, h.w, acc.w);
        }
    }
    if (col) *reinterpret_cast<float4*>(out + b * K + 4 * sub) = acc;
}

// backward: dw_n = g_b . h_n;  normalize: ds_n = w_n (dw_n - sum_m w_m dw_m) / sqrt(K), else ds_n = dw_n;  dH_n = w_n g_b
__global__ __launch_bounds__(256) void din_pool_rows_bwd_k(const float* __restrict__ g, const float* __restrict__ Hc, int K, const float* __restrict__ w,
                                                            const int64_t* __restrict__ row_off, int64_t B, int64_t N, int normalize, float inv_sqrt_k,
                                                            float* __restrict__ ds, float* __restrict__ dH, int lpr) {
    const int sub = threadIdx.x % lpr;
    const int64_t b = ((int64_t)blockIdx.x * 256 + threadIdx.x) / lpr;
    const bool live = b < B;
    const int64_t n0 = live ? row_off[b] : 0, n1 = live ? (b + 1 < B ? row_off[b + 1] : N) : 0;
    // the four groups of a wave walk their own samples: the loop runs to the LONGEST of them so that every lane reaches the shuffles
    int64_t cnt = n1 - n0;
    for (int o = 32; o >= lpr; o >>= 1) cnt = max(cnt, (int64_t)__shfl_xor((long long)cnt, o, 64));
    const bool col = live && 4 * sub < K;
    const float4 gb = col ? *reinterpret_cast<const float4*>(g + b * K + 4 * sub) : make_float4(0.f, 0.f, 0.f, 0.f);
    // two walks over the sample's rows (the second one's reads hit L2): first t = sum_m w_m dw_m and dH, then ds -- no value is handed from
    // one lane to the others through memory
    float t = 0.f;
    for (int64_t i = 0; i < cnt; ++i) {
        const int64_t n = n0 + i;
        const bool in = n < n1;
        float d = 0.f;
        const float wn = in ? w[n] : 0.f;
        if (col && in) {
            const float4 h = *reinterpret_cast<const float4*>(Hc + n * K + 4 * sub);
            d = (gb.x * h.x + gb.y * h.y) + (gb.z * h.z + gb.w * h.w);
            *reinterpret_cast<float4*>(dH + n * K + 4 * sub) = make_float4(wn * gb.x, wn * gb.y, wn * gb.z, wn * gb.w);
        }
        for (int o = lpr >> 1; o > 0; o >>= 1) d += __shfl_xor(d, o, 64);
        if (!normalize && in && sub == 0) ds[n] = d;
        t = fmaf(wn, d, t);
    }
    if (normalize) {
        for (int64_t i = 0; i < cnt; ++i) {
            const int64_t n = n0 + i;
            const bool in = n < n1;
            float d = 0.f;
            if (col && in) {
                const float4 h = *reinterpret_cast<const float4*>(Hc + n * K + 4 * sub);
                d = (gb.x * h.x + gb.y * h.y) + (gb.z * h.z + gb.w * h.w);
            }
            for (int o = lpr >> 1; o > 0; o >>= 1) d += __shfl_xor(d, o, 64);
            if (in && sub == 0) ds[n] = w[n] * (d - t) * inv_sqrt_k;
        }
    }
}

}  // namespace dir

using namespace dir;

extern "C" int dir_din_feat_rows_f32(const float* table, int K, const int64_t* ids_h, const int64_t* b_idx, const int64_t* cand, int64_t N, float* X,
                                     float* Hc, dir_stream_t stream) {
    const char* name = "dir_din_feat_rows_f32";
    DIR_CHECK_ARG(K > 0 && K <= 256 && (K & 3) == 0 && N >= 0, "%s: K=%d (a multiple of 4, <= 256) N=%lld", name, K, (long long)N);
    if (N == 0) return DIR_OK;
    DIR_CHECK_ARG(table && ids_h && b_idx && cand && X && Hc, "%s: null pointer", name);
    if (!aligned16(table) || !aligned16(X) || !aligned16(Hc)) return fail(DIR_E_BADARG, "%s: table / X / Hc must be 16-byte aligned", name);
    const int lpr = K <= 64 ? 16 : 64;
    hipLaunchKernelGGL(din_feat_rows_k, dim3((unsigned)((N * lpr + 255) / 256)), dim3(256), 0, as_stream(stream), table, K, ids_h, b_idx, cand, N, X, Hc, lpr);
    DIR_CHECK_LAUNCH(name);
    return DIR_OK;
}

extern "C" int dir_din_feat_rows_backward_f32(const float* table, int K, const int64_t* ids_h, const int64_t* row_off, const int64_t* cand, int64_t B,
                                              int64_t N, const float* dX, const float* dH, float* grows, dir_stream_t stream) {
    const char* name = "dir_din_feat_rows_backward_f32";
    DIR_CHECK_ARG(K > 0 && K <= 256 && (K & 3) == 0 && N >= 0 && B >= 0, "%s: K=%d N=%lld B=%lld", name, K, (long long)N, (long long)B);
    if (B == 0) return DIR_OK;
    DIR_CHECK_ARG(table && row_off && cand && grows && (N == 0 || (ids_h && dX && dH)), "%s: null pointer", name);
    if (!aligned16(table) || !aligned16(dX) || !aligned16(dH) || !aligned16(grows)) return fail(DIR_E_BADARG, "%s: operands must be 16-byte aligned", name);
    const int lpr = K <= 64 ? 16 : 64;
    hipLaunchKernelGGL(din_feat_rows_bwd_k, dim3((unsigned)((B * lpr + 255) / 256)), dim3(256), 0, as_stream(stream), table, K, ids_h, row_off, cand, B, N, dX,
                       dH, grows, lpr);
    DIR_CHECK_LAUNCH(name);
    return DIR_OK;
}

static int art_check(const char* name, int64_t M, int N, int activation, const void* alpha, const void* scale, const void* shift) {
    DIR_CHECK_ARG(M >= 0 && N > 0 && N <= 1024 && (N & 3) == 0, "%s: M=%lld N=%d (a multiple of 4, <= 1024)", name, (long long)M, N);
    DIR_CHECK_ARG(activation == DIR_DIN_ACT_PRELU || activation == DIR_DIN_ACT_DICE, "%s: activation %d (1 PReLU, 2 Dice)", name, activation);
    DIR_CHECK_ARG(M == 0 || (alpha && (activation == DIR_DIN_ACT_PRELU || (scale && shift))), "%s: null parameter vector", name);
    return DIR_OK;
}

extern "C" int dir_act_rows_train_f32(const float* s, int64_t s_ld, int64_t M, int N, int activation, const float* alpha, const float* scale,
                                      const float* shift, float* y, int64_t y_ld, dir_stream_t stream) {
    const char* name = "dir_act_rows_train_f32";
    if (int rc = art_check(name, M, N, activation, alpha, scale, shift)) return rc;
    if (M == 0) return DIR_OK;
    DIR_CHECK_ARG(s && y && s_ld >= N && y_ld >= N, "%s: null pointer or row stride smaller than N", name);
    if ((s_ld & 3) || (y_ld & 3) || !aligned16(s) || !aligned16(y) || !aligned16(alpha) || (scale && !aligned16(scale)) || (shift && !aligned16(shift)))
        return fail(DIR_E_UNSUPPORTED, "%s: row strides must be multiples of 4 and every operand 16-byte aligned", name);
    const dim3 grid((unsigned)grid_for((M * (N >> 2) + 255) / 256));
    if (activation == DIR_DIN_ACT_PRELU) hipLaunchKernelGGL(act_rows_train_k<1>, grid, dim3(256), 0, as_stream(stream), s, s_ld, M, N, alpha, scale, shift, y, y_ld);
    else hipLaunchKernelGGL(act_rows_train_k<2>, grid, dim3(256), 0, as_stream(stream), s, s_ld, M, N, alpha, scale, shift, y, y_ld);
    DIR_CHECK_LAUNCH(name);
    return DIR_OK;
}

extern "C" int dir_act_rows_backward_partials(int64_t M, int N) {
    if (M <= 0 || N <= 0) return 0;
    const int64_t want = (M + 255) / 256;                     // >= 256 rows per block
    return (int)(want < 2048 ? want : 2048);
}

extern "C" int dir_act_rows_backward_f32(const float* g, int64_t g_ld, const float* s, int64_t s_ld, int64_t M, int N, int activation, const float* alpha,
                                         const float* scale, const float* shift, float* d1, int64_t d1_ld, float* gx, int64_t gx_ld, float* galpha,
                                         float* partials, int n_partials, dir_stream_t stream) {
    const char* name = "dir_act_rows_backward_f32";
    if (int rc = art_check(name, M, N, activation, alpha, scale, shift)) return rc;
    DIR_CHECK_ARG(galpha, "%s: galpha is null", name);
    hipStream_t st = as_stream(stream);
    if (M == 0) {
        if (zero_async(galpha, (size_t)N * sizeof(float), st) != hipSuccess) return fail(DIR_E_HIP, "%s: zeroing failed", name);
        return DIR_OK;
    }
    DIR_CHECK_ARG(g && s && d1 && partials && (activation == DIR_DIN_ACT_PRELU || gx) && n_partials == dir_act_rows_backward_partials(M, N),
                  "%s: null pointer, or n_partials != dir_act_rows_backward_partials(M, N)", name);
    DIR_CHECK_ARG(g_ld >= N && s_ld >= N && d1_ld >= N && (!gx || gx_ld >= N), "%s: a row stride is smaller than N", name);
    if ((g_ld & 3) || (s_ld & 3) || (d1_ld & 3) || (gx && (gx_ld & 3)) || !aligned16(g) || !aligned16(s) || !aligned16(d1) || (gx && !aligned16(gx)) ||
        !aligned16(alpha) || (scale && !aligned16(scale)) || (shift && !aligned16(shift)) || !aligned16(partials))
        return fail(DIR_E_UNSUPPORTED, "%s: row strides must be multiples of 4 and every operand 16-byte aligned", name);
    const int64_t rpb = (M + n_partials - 1) / n_partials;
    if (activation == DIR_DIN_ACT_PRELU)
        hipLaunchKernelGGL(act_rows_bwd_k<1>, dim3((unsigned)n_partials), dim3(256), 0, st, g, g_ld, s, s_ld, M, N, alpha, scale, shift, rpb, d1, d1_ld, gx,
                           gx_ld, partials);
    else
        hipLaunchKernelGGL(act_rows_bwd_k<2>, dim3((unsigned)n_partials), dim3(256), 0, st, g, g_ld, s, s_ld, M, N, alpha, scale, shift, rpb, d1, d1_ld, gx,
                           gx_ld, partials);
    hipLaunchKernelGGL(act_rows_alpha_fin_k, dim3((unsigned)((N + 15) / 16)), dim3(256), 0, st, partials, n_partials, N, galpha);
    DIR_CHECK_LAUNCH(name);
    return DIR_OK;
}

extern "C" int dir_din_pool_rows_f32(const float* scores, const float* Hc, int K, const int64_t* row_off, int64_t B, int64_t N, int normalize, float* w,
                                     float* out, dir_stream_t stream) {
    const char* name = "dir_din_pool_rows_f32";
    DIR_CHECK_ARG(K > 0 && K <= 256 && (K & 3) == 0 && N >= 0 && B >= 0, "%s: K=%d N=%lld B=%lld", name, K, (long long)N, (long long)B);
    if (B == 0) return DIR_OK;
    DIR_CHECK_ARG(row_off && out && (N == 0 || (scores && Hc && w)), "%s: null pointer", name);
    if (!aligned16(Hc) || !aligned16(out)) return fail(DIR_E_BADARG, "%s: Hc / out must be 16-byte aligned", name);
    const int lpr = K <= 64 ? 16 : 64;
    hipLaunchKernelGGL(din_pool_rows_k, dim3((unsigned)((B * lpr + 255) / 256)), dim3(256), 0, as_stream(stream), scores, Hc, K, row_off, B, N,
                       normalize ? 1 : 0, 1.0f / sqrtf((float)K), w, out, lpr);
    DIR_CHECK_LAUNCH(name);
    return DIR_OK;
}

extern "C" int dir_din_pool_rows_backward_f32(const float* g, const float* Hc, int K, const float* w, const int64_t* row_off, int64_t B, int64_t N,
                                              int normalize, float* ds, float* dH, dir_stream_t stream) {
    const char* name = "dir_din_pool_rows_backward_f32";
    DIR_CHECK_ARG(K > 0 && K <= 256 && (K & 3) == 0 && N >= 0 && B >= 0, "%s: K=%d N=%lld B=%lld", name, K, (long long)N, (long long)B);
    if (B == 0 || N == 0) return DIR_OK;
    DIR_CHECK_ARG(g && Hc && w && row_off && ds && dH, "%s: null pointer", name);
    if (!aligned16(g) || !aligned16(Hc) || !aligned16(dH)) return fail(DIR_E_BADARG, "%s: g / Hc / dH must be 16-byte aligned", name);
    const int lpr = K <= 64 ? 16 : 64;
    hipLaunchKernelGGL(din_pool_rows_bwd_k, dim3((unsigned)((B * lpr + 255) / 256)), dim3(256), 0, as_stream(stream), g, Hc, K, w, row_off, B, N,
                       normalize ? 1 : 0, 1.0f / sqrtf((float)K), ds, dH, lpr);
    DIR_CHECK_LAUNCH(name);
    return DIR_OK;
}
