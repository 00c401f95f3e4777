// bn_train.hip -- training-mode batch normalisation of a hidden layer's [B, N] activation, forward statistics and backward.
//
// Reference: tf.layers.batch_normalization(net, training=True) after each hidden layer of dnn_logit_fn (models/DeepFM/deepFM.py:303-308)
// and tf.contrib.layers.batch_norm(scale=False) in _deep_architecture (models/DeepCrossNetwork/DeepCrossNetwork.py:400-403, 413-419); the
// reference trains through TensorFlow autodiff of those lines.  [TF-upstream] for a rank-2 input: the batch's mean and POPULATION variance
// normalise, moving = moving * momentum + batch * (1 - momentum), out = (y - mean) * rsqrt(var + eps) * gamma + beta.
//
// As torch ops the forward is a mean pass, a Welford pass and three elementwise passes over [B, N], the backward two column sums and
// five elementwise passes plus the ReLU gate of the layer below (2.7 ms of glue in the DCN training step at 65 536 x 1024).  Here:
//   forward   bn_cols_k<false>: per-workgroup column sums of y and y^2 (one read of y)  -> bn_fin_fwd_k: mean, inv, scale, shift, moving stats
//             (the apply, out = y * scale + shift, is one elementwise pass the caller runs)
//   backward  bn_cols_k<true>:  column sums of g and g * y (one read of g and y)         -> bn_fin_bwd_k: dbeta, dgamma, three coefficients
//             bn_bwd_apply_k:   dy = c1 * g + c2 * y + c3, optionally gated by y > 0 (y is the ReLU output of the layer below: the
//             result is then dL/d(pre-activation) of that layer) -- one read of g and y, one write.
// Sums: fp32 within a thread's <= 64 rows and across a workgroup's row lanes, fp64 across workgroups in workgroup order (no atomics:
// bitwise reproducible); var = E[y^2] - mean^2 is formed in fp64.
// HBM-bound: forward 4 B, backward 20 B per element.
#include "common.hpp"

namespace dir {

constexpr int BN_MAXCH = 4;          // float4 column chunks per thread: N <= 4 * 4 * 256

// partials[block][0][n] = sum over the block's rows of (HAS_G ? g : y),  [1][n] = sum of (HAS_G ? g * y : y * y)
template <int TPR, bool HAS_G>
__global__ __launch_bounds__(256) void bn_cols_k(const float* __restrict__ y, int64_t y_ld, const float* __restrict__ g, int64_t g_ld, int64_t B,
                                                  int N, int64_t rows_per_block, float* __restrict__ part) {
    constexpr int RPI = 256 / TPR;
    __shared__ float4 red[2][256];
    const int tid = threadIdx.x, c0 = tid % TPR, rr = tid / TPR;
    const int nv = N >> 2;
    const int64_t r0 = (int64_t)blockIdx.x * rows_per_block;
    const int64_t r1 = min(B, r0 + rows_per_block);
    float4 s0[BN_MAXCH], s1[BN_MAXCH];
#pragma unroll
    for (int ch = 0; ch < BN_MAXCH; ++ch) s0[ch] = s1[ch] = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int64_t r = r0 + rr; r < r1; r += RPI) {
#pragma unroll
        for (int ch = 0; ch < BN_MAXCH; ++ch) {
            const int c = c0 + ch * TPR;
            if (c < nv) {
                const float4 yv = *reinterpret_cast<const float4*>(y + r * y_ld + 4 * c);
                if constexpr (HAS_G) {
                    const float4 gv = *reinterpret_cast<const float4*>(g + r * g_ld + 4 * c);
                    s0[ch].x += gv.x; s0[ch].y += gv.y; s0[ch].z += gv.z; s0[ch].w += gv.w;
                    s1[ch].x += gv.x * yv.x; s1[ch].y += gv.y * yv.y; s1[ch].z += gv.z * yv.z; s1[ch].w += gv.w * yv.w;
                } else {
                    s0[ch].x += yv.x; s0[ch].y += yv.y; s0[ch].z += yv.z; s0[ch].w += yv.w;
                    s1[ch].x += yv.x * yv.x; s1[ch].y += yv.y * yv.y; s1[ch].z += yv.z * yv.z; s1[ch].w += yv.w * yv.w;
                }
            }
        }
    }
    float* p0 = part + (int64_t)blockIdx.x * 2 * N;
#pragma unroll
    for (int ch = 0; ch < BN_MAXCH; ++ch) {
        const int c = c0 + ch * TPR;
        if (ch * TPR < nv) {                        // uniform: the chunk exists for some thread
            red[0][tid] = s0[ch];
            red[1][tid] = s1[ch];
            __syncthreads();
            if (rr == 0 && c < nv) {
                float4 a = red[0][c0], b = red[1][c0];
#pragma unroll
                for (int q = 1; q < RPI; ++q) {      // row lanes in lane order
                    const float4 u = red[0][q * TPR + c0], v = red[1][q * TPR + c0];
                    a.x += u.x; a.y += u.y; a.z += u.z; a.w += u.w;
                    b.x += v.x; b.y += v.y; b.z += v.z; b.w += v.w;
                }
                *reinterpret_cast<float4*>(p0 + 4 * c) = a;
                *reinterpret_cast<float4*>(p0 + N + 4 * c) = b;
            }
            __syncthreads();
        }
    }
}

// Column sums of the workgroups' partials in fp64.  A 256-thread workgroup takes 32 columns: thread (lane l = tid >> 5, column c = tid & 31)
// adds partials l, l + 8, ... in order, the eight lanes are added in lane order through LDS (a fixed order: bitwise reproducible);
// the threads of lane 0 return true and hold the sums.  (One thread per column walking all 1024 partials took 0.29 ms.)
constexpr int BN_FIN_COLS = 32;
// ROWS = 3 (the Dice backward's partials): a third sum per column, s2.
template <int ROWS = 2>
__device__ __forceinline__ bool bn_col_sums(const float* __restrict__ part, int64_t P, int N, int& n, double& s0, double& s1, double* s2 = nullptr) {
    __shared__ double red[ROWS][8][BN_FIN_COLS];
    const int c = threadIdx.x & 31, l = threadIdx.x >> 5;
    n = blockIdx.x * BN_FIN_COLS + c;
    s0 = s1 = 0.0;
    double t2 = 0.0;
    if (n < N) {
#pragma unroll 4
        for (int64_t p = l; p < P; p += 8) {
            s0 += (double)part[(p * ROWS) * N + n];
            s1 += (double)part[(p * ROWS + 1) * N + n];
            if constexpr (ROWS == 3) t2 += (double)part[(p * ROWS + 2) * N + n];
        }
    }
    red[0][l][c] = s0;
    red[1][l][c] = s1;
    if constexpr (ROWS == 3) red[2][l][c] = t2;
    __syncthreads();
    if (l != 0 || n >= N) return false;
#pragma unroll
    for (int q = 1; q < 8; ++q) {
        s0 += red[0][q][c];
        s1 += red[1][q][c];
        if constexpr (ROWS == 3) t2 += red[2][q][c];
    }
    if constexpr (ROWS == 3) *s2 = t2;
    return true;
}

__global__ __launch_bounds__(256) void bn_fin_fwd_k(const float* __restrict__ part, int64_t P, int64_t B, int N, float eps, float momentum,
                                                     const float* __restrict__ gamma, const float* __restrict__ beta,
                                                     float* __restrict__ moving_mean, float* __restrict__ moving_var, float* __restrict__ mean_out,
                                                     float* __restrict__ inv_out, float* __restrict__ scale_out, float* __restrict__ shift_out) {
    int n;
    double s, q;
    if (!bn_col_sums(part, P, N, n, s, q)) return;
    const double m = s / (double)B;
    double v = q / (double)B - m * m;
    if (v < 0.0) v = 0.0;
    const float mean = (float)m, var = (float)v;
    const float inv = 1.0f / sqrtf(var + eps);
    const float scale = gamma ? inv * gamma[n] : inv;
    mean_out[n] = mean;
    inv_out[n] = inv;
    scale_out[n] = scale;
    shift_out[n] = (beta ? beta[n] : 0.f) - mean * scale;
    if (moving_mean) moving_mean[n] = moving_mean[n] * momentum + mean * (1.0f - momentum);
    if (moving_var) moving_var[n] = moving_var[n] * momentum + var * (1.0f - momentum);
}

// coef[0..2][n]: dy = coef0 * g + coef1 * y + coef2   (= scale * (g - mean_b(g) - xhat * mean_b(g * xhat)), xhat = (y - mean) * inv)
// ROWS = 3: the partials' third row is summed into extra[n] (the Dice backward's dL/dalpha)
template <int ROWS = 2>
__global__ __launch_bounds__(256) void bn_fin_bwd_k(const float* __restrict__ part, int64_t P, int64_t B, int N, const float* __restrict__ mean,
                                                     const float* __restrict__ inv, const float* __restrict__ gamma, float* __restrict__ coef,
                                                     float* __restrict__ gbeta, float* __restrict__ ggamma, float* __restrict__ extra = nullptr) {
    int n;
    double sg, sgy, sx = 0.0;
    if (!bn_col_sums<ROWS>(part, P, N, n, sg, sgy, &sx)) return;
    if constexpr (ROWS == 3) extra[n] = (float)sx;
    const double m = (double)mean[n], iv = (double)inv[n];
    const double sgx = iv * (sgy - m * sg);                      // sum_b g * xhat
    const double scale = gamma ? iv * (double)gamma[n] : iv;
    const double a = sg / (double)B, b = sgx / (double)B;
    coef[n] = (float)scale;
    coef[N + n] = (float)(-scale * b * iv);
    coef[2 * N + n] = (float)(-scale * a + scale * b * iv * m);
    if (gbeta) gbeta[n] = (float)sg;
    if (ggamma) ggamma[n] = (float)sgx;
}

template <int TPR, bool GATE>
__global__ __launch_bounds__(256) void bn_bwd_apply_k(const float* __restrict__ g, int64_t g_ld, const float* __restrict__ y, int64_t y_ld,
                                                       const float* __restrict__ coef, int64_t B, int N, int64_t rows_per_block,
                                                       float* __restrict__ gy, int64_t gy_ld) {
    constexpr int RPI = 256 / TPR;
    const int tid = threadIdx.x, c0 = tid % TPR, rr = tid / TPR;
    const int nv = N >> 2;
    const int64_t r0 = (int64_t)blockIdx.x * rows_per_block;
    const int64_t r1 = min(B, r0 + rows_per_block);
    float4 k0[BN_MAXCH], k1[BN_MAXCH], k2[BN_MAXCH];
#pragma unroll
    for (int ch = 0; ch < BN_MAXCH; ++ch) {
        const int c = c0 + ch * TPR;
        const float4 z = make_float4(0.f, 0.f, 0.f, 0.f);
        k0[ch] = c < nv ? *reinterpret_cast<const float4*>(coef + 4 * c) : z;
        k1[ch] = c < nv ? *reinterpret_cast<const float4*>(coef + N + 4 * c) : z;
        k2[ch] = c < nv ? *reinterpret_cast<const float4*>(coef + 2 * N + 4 * c) : z;
    }
    for (int64_t r = r0 + rr; r < r1; r += RPI) {
#pragma unroll
        for (int ch = 0; ch < BN_MAXCH; ++ch) {
            const int c = c0 + ch * TPR;
            if (c < nv) {
                const float4 gv = *reinterpret_cast<const float4*>(g + r * g_ld + 4 * c);
                const float4 yv = *reinterpret_cast<const float4*>(y + r * y_ld + 4 * c);
                float4 v;
                v.x = fmaf(k0[ch].x, gv.x, fmaf(k1[ch].x, yv.x, k2[ch].x));
                v.y = fmaf(k0[ch].y, gv.y, fmaf(k1[ch].y, yv.y, k2[ch].y));
                v.z = fmaf(k0[ch].z, gv.z, fmaf(k1[ch].z, yv.z, k2[ch].z));
                v.w = fmaf(k0[ch].w, gv.w, fmaf(k1[ch].w, yv.w, k2[ch].w));
                if constexpr (GATE) {
                    v.x = yv.x > 0.f ? v.x : 0.f; v.y = yv.y > 0.f ? v.y : 0.f; v.z = yv.z > 0.f ? v.z : 0.f; v.w = yv.w > 0.f ? v.w : 0.f;
                }
                *reinterpret_cast<float4*>(gy + r * gy_ld + 4 * c) = v;
            }
        }
    }
}

// ---- Dice in training mode, backward in two passes over (g, s) (round 6; arXiv:1706.06978 section 5.3, no reference code) -------------------
// y = s (alpha + (1 - alpha) p), p = sigmoid(xhat), xhat = scale s + shift with THIS batch's statistics.  dL/ds has the direct term
// d1 = g (alpha + (1 - alpha) p) and the statistics' share, the batch-norm backward of gx = g s (1 - alpha) p (1 - p) = dL/dxhat.  As separate
// passes (dir_act_rows_backward_f32 -> d1, gx; dir_bn_train_backward_f32 (gx, s) -> dbn; d1 += dbn) that is 4 writes and 7 reads of [M, N];
// here: pass 1 reads (g, s) and only forms the column sums (gx, gx s, and dL/dalpha's g s (1 - p)); pass 2 reads (g, s) again, recomputes d1
// and gx and writes ds = d1 + (c0 gx + c1 s + c2) -- the same expressions, so the same values as the separate passes given the same sums.
__device__ __forceinline__ float bn_sigmoid(float x) { return 1.0f / (1.0f + __expf(-x)); }

template <int TPR>
__global__ __launch_bounds__(256) void dice_cols_k(const float* __restrict__ g, int64_t g_ld, const float* __restrict__ s, int64_t s_ld, int64_t B, int N,
                                                    const float* __restrict__ alpha, const float* __restrict__ scale, const float* __restrict__ shift,
                                                    int64_t rows_per_block, float* __restrict__ part /* [block][3][N] */) {
    constexpr int RPI = 256 / TPR;
    __shared__ float4 red[3][256];
    const int tid = threadIdx.x, c0 = tid % TPR, rr = tid / TPR;
    const int nv = N >> 2;
    const int64_t r0 = (int64_t)blockIdx.x * rows_per_block;
    const int64_t r1 = min(B, r0 + rows_per_block);
    float* p0 = part + (int64_t)blockIdx.x * 3 * N;
#pragma unroll 1
    for (int ch = 0; ch * TPR < nv; ++ch) {              // (uniform) one column chunk at a time: the per-column vectors stay in registers
        const int c = c0 + ch * TPR;
        float q0[4] = {0.f, 0.f, 0.f, 0.f}, q1[4] = {0.f, 0.f, 0.f, 0.f}, q2[4] = {0.f, 0.f, 0.f, 0.f};
        if (c < nv) {
            const float4 al4 = *reinterpret_cast<const float4*>(alpha + 4 * c), sc4 = *reinterpret_cast<const float4*>(scale + 4 * c),
                         sh4 = *reinterpret_cast<const float4*>(shift + 4 * c);
            const float al[4] = {al4.x, al4.y, al4.z, al4.w}, sc[4] = {sc4.x, sc4.y, sc4.z, sc4.w}, sh[4] = {sh4.x, sh4.y, sh4.z, sh4.w};
            for (int64_t r = r0 + rr; r < r1; r += RPI) {
                const float4 g4 = *reinterpret_cast<const float4*>(g + r * g_ld + 4 * c), s4 = *reinterpret_cast<const float4*>(s + r * s_ld + 4 * c);
                const float gv[4] = {g4.x, g4.y, g4.z, g4.w}, sv[4] = {s4.x, s4.y, s4.z, s4.w};
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const float pq = bn_sigmoid(fmaf(sv[q], sc[q], sh[q]));
                    const float gx = gv[q] * sv[q] * (1.f - al[q]) * pq * (1.f - pq);       // (act_rows_bwd_k's expression)
                    q0[q] += gx;
                    q1[q] += gx * sv[q];
                    q2[q] += gv[q] * sv[q] * (1.f - pq);
                }
            }
        }
        red[0][tid] = make_float4(q0[0], q0[1], q0[2], q0[3]);
        red[1][tid] = make_float4(q1[0], q1[1], q1[2], q1[3]);
        red[2][tid] = make_float4(q2[0], q2[1], q2[2], q2[3]);
        __syncthreads();
        if (rr == 0 && c < nv) {
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                float4 a = red[k][c0];
#pragma unroll
                for (int q = 1; q < RPI; ++q) {          // row lanes in lane order
                    const float4 u = red[k][q * TPR + c0];
                    a.x += u.x; a.y += u.y; a.z += u.z; a.w += u.w;
                }
                *reinterpret_cast<float4*>(p0 + k * N + 4 * c) = a;
            }
        }
        __syncthreads();
    }
}

template <int TPR>
__global__ __launch_bounds__(256) void dice_bwd_apply_k(const float* __restrict__ g, int64_t g_ld, const float* __restrict__ s, int64_t s_ld,
                                                         const float* __restrict__ coef, const float* __restrict__ alpha, const float* __restrict__ scale,
                                                         const float* __restrict__ shift, int64_t B, int N, int64_t rows_per_block,
                                                         float* __restrict__ ds, int64_t ds_ld) {
    constexpr int RPI = 256 / TPR;
    const int tid = threadIdx.x, c0 = tid % TPR, rr = tid / TPR;
    const int nv = N >> 2;
    const int64_t r0 = (int64_t)blockIdx.x * rows_per_block;
    const int64_t r1 = min(B, r0 + rows_per_block);
#pragma unroll 1
    for (int ch = 0; ch * TPR < nv; ++ch) {
        const int c = c0 + ch * TPR;
        if (c >= nv) continue;
        const float4 al4 = *reinterpret_cast<const float4*>(alpha + 4 * c), sc4 = *reinterpret_cast<const float4*>(scale + 4 * c),
                     sh4 = *reinterpret_cast<const float4*>(shift + 4 * c), k04 = *reinterpret_cast<const float4*>(coef + 4 * c),
                     k14 = *reinterpret_cast<const float4*>(coef + N + 4 * c), k24 = *reinterpret_cast<const float4*>(coef + 2 * N + 4 * c);
        const float al[4] = {al4.x, al4.y, al4.z, al4.w}, sc[4] = {sc4.x, sc4.y, sc4.z, sc4.w}, sh[4] = {sh4.x, sh4.y, sh4.z, sh4.w};
        const float k0[4] = {k04.x, k04.y, k04.z, k04.w}, k1[4] = {k14.x, k14.y, k14.z, k14.w}, k2[4] = {k24.x, k24.y, k24.z, k24.w};
        for (int64_t r = r0 + rr; r < r1; r += RPI) {
            const float4 g4 = *reinterpret_cast<const float4*>(g + r * g_ld + 4 * c), s4 = *reinterpret_cast<const float4*>(s + r * s_ld + 4 * c);
            const float gv[4] = {g4.x, g4.y, g4.z, g4.w}, sv[4] = {s4.x, s4.y, s4.z, s4.w};
            float o[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const float pq = bn_sigmoid(fmaf(sv[q], sc[q], sh[q]));
                const float d1 = gv[q] * fmaf(pq, 1.f - al[q], al[q]);
                const float gx = gv[q] * sv[q] * (1.f - al[q]) * pq * (1.f - pq);
                o[q] = d1 + fmaf(k0[q], gx, fmaf(k1[q], sv[q], k2[q]));
            }
            *reinterpret_cast<float4*>(ds + r * ds_ld + 4 * c) = make_float4(o[0], o[1], o[2], o[3]);
        }
    }
}

struct BnPlan { int tpr; int64_t nblk, rows_per_block; };
static BnPlan bn_plan(int64_t B, int N) {
    BnPlan p;
    const int nv = N / 4;
    p.tpr = nv <= 64 ? 64 : (nv <= 128 ? 128 : 256);
    const int rpi = 256 / p.tpr;
    int64_t nblk = (B + rpi * 16 - 1) / (rpi * 16);          // >= 16 rows per row lane
    if (nblk > 4 * kCUs) nblk = 4 * kCUs;
    if (nblk < 1) nblk = 1;
    int64_t rpb = (B + nblk - 1) / nblk;
    rpb = (rpb + rpi - 1) / rpi * rpi;
    p.rows_per_block = rpb > 0 ? rpb : rpi;
    p.nblk = B > 0 ? (B + p.rows_per_block - 1) / p.rows_per_block : 0;
    return p;
}

static int bn_check(const char* name, int64_t B, int N, int64_t ld0, int64_t ld1, int64_t ld2) {
    DIR_CHECK_ARG(B > 0 && N > 0, "%s: B=%lld N=%d", name, (long long)B, N);
    if (N % 4 || ld0 % 4 || ld1 % 4 || ld2 % 4 || N > 4 * BN_MAXCH * 256)
        return fail(DIR_E_UNSUPPORTED, "%s: N=%d and the row strides must be multiples of 4, N <= %d", name, N, 4 * BN_MAXCH * 256);
    DIR_CHECK_ARG(ld0 >= N && ld1 >= N && ld2 >= N, "%s: row strides smaller than N", name);
    return DIR_OK;
}

}  // namespace dir

using namespace dir;

extern "C" int64_t dir_bn_train_partials(int64_t B, int N) {
    if (B <= 0 || N <= 0) return 0;
    return bn_plan(B, N).nblk;
}

extern "C" int dir_bn_train_stats_f32(const float* y, int64_t y_ld, int64_t B, int N, float eps, float momentum, const float* gamma,
                                      const float* beta, float* moving_mean, float* moving_var, float* mean, float* inv, float* scale,
                                      float* shift, float* partials, int64_t n_partials, dir_stream_t stream) {
    const char* name = "dir_bn_train_stats_f32";
    if (int rc = bn_check(name, B, N, y_ld, y_ld, y_ld)) return rc;
    DIR_CHECK_ARG(y && mean && inv && scale && shift && partials, "%s: null pointer", name);
    if (!(aligned16(y) && aligned16(partials))) return fail(DIR_E_BADARG, "%s: y / partials must be 16-byte aligned", name);
    const BnPlan p = bn_plan(B, N);
    DIR_CHECK_ARG(n_partials >= p.nblk, "%s: partials holds %lld row pairs, dir_bn_train_partials(B, N) = %lld", name, (long long)n_partials,
                  (long long)p.nblk);
    hipStream_t st = as_stream(stream);
    if (p.tpr == 64)
        hipLaunchKernelGGL((bn_cols_k<64, false>), dim3((unsigned)p.nblk), dim3(256), 0, st, y, y_ld, nullptr, 0, B, N, p.rows_per_block, partials);
    else if (p.tpr == 128)
        hipLaunchKernelGGL((bn_cols_k<128, false>), dim3((unsigned)p.nblk), dim3(256), 0, st, y, y_ld, nullptr, 0, B, N, p.rows_per_block, partials);
    else
        hipLaunchKernelGGL((bn_cols_k<256, false>), dim3((unsigned)p.nblk), dim3(256), 0, st, y, y_ld, nullptr, 0, B, N, p.rows_per_block, partials);
    DIR_CHECK_LAUNCH(name);
    hipLaunchKernelGGL(bn_fin_fwd_k, dim3((unsigned)((N + BN_FIN_COLS - 1) / BN_FIN_COLS)), dim3(256), 0, st, partials, p.nblk, B, N, eps, momentum, gamma, beta,
                       moving_mean, moving_var, mean, inv, scale, shift);
    DIR_CHECK_LAUNCH(name);
    return DIR_OK;
}

extern "C" int dir_bn_train_backward_f32(const float* g, int64_t g_ld, const float* y, int64_t y_ld, int64_t B, int N, const float* mean,
                                         const float* inv, const float* gamma, int relu_gate, float* gy, int64_t gy_ld, float* gbeta,
                                         float* ggamma, float* coef, float* partials, int64_t n_partials, dir_stream_t stream) {
    const char* name = "dir_bn_train_backward_f32";
    if (int rc = bn_check(name, B, N, y_ld, g_ld, gy_ld)) return rc;
    DIR_CHECK_ARG(g && y && mean && inv && gy && coef && partials, "%s: null pointer", name);
    if (!(aligned16(g) && aligned16(y) && aligned16(gy) && aligned16(coef) && aligned16(partials)))
        return fail(DIR_E_BADARG, "%s: g / y / gy / coef / partials must be 16-byte aligned", name);
    const BnPlan p = bn_plan(B, N);
    DIR_CHECK_ARG(n_partials >= p.nblk, "%s: partials holds %lld row pairs, dir_bn_train_partials(B, N) = %lld", name, (long long)n_partials,
                  (long long)p.nblk);
    hipStream_t st = as_stream(stream);
    if (p.tpr == 64)
        hipLaunchKernelGGL((bn_cols_k<64, true>), dim3((unsigned)p.nblk), dim3(256), 0, st, y, y_ld, g, g_ld, B, N, p.rows_per_block, partials);
    else if (p.tpr == 128)
        hipLaunchKernelGGL((bn_cols_k<128, true>), dim3((unsigned)p.nblk), dim3(256), 0, st, y, y_ld, g, g_ld, B, N, p.rows_per_block, partials);
    else
        hipLaunchKernelGGL((bn_cols_k<256, true>), dim3((unsigned)p.nblk), dim3(256), 0, st, y, y_ld, g, g_ld, B, N, p.rows_per_block, partials);
    DIR_CHECK_LAUNCH(name);
    hipLaunchKernelGGL(bn_fin_bwd_k<2>, dim3((unsigned)((N + BN_FIN_COLS - 1) / BN_FIN_COLS)), dim3(256), 0, st, partials, p.nblk, B, N, mean, inv, gamma, coef, gbeta, ggamma);
    DIR_CHECK_LAUNCH(name);
#define DIR_BN_APPLY(T)                                                                                                                        \
    do {                                                                                                                                       \
        if (relu_gate)                                                                                                                         \
            hipLaunchKernelGGL((bn_bwd_apply_k<T, true>), dim3((unsigned)p.nblk), dim3(256), 0, st, g, g_ld, y, y_ld, coef, B, N, p.rows_per_block, \
                               gy, gy_ld);                                                                                                     \
        else                                                                                                                                   \
            hipLaunchKernelGGL((bn_bwd_apply_k<T, false>), dim3((unsigned)p.nblk), dim3(256), 0, st, g, g_ld, y, y_ld, coef, B, N,               \
                               p.rows_per_block, gy, gy_ld);                                                                                   \
    } while (0)
    if (p.tpr == 64) DIR_BN_APPLY(64);
    else if (p.tpr == 128) DIR_BN_APPLY(128);
    else DIR_BN_APPLY(256);
#undef DIR_BN_APPLY
    DIR_CHECK_LAUNCH(name);
    return DIR_OK;
}

extern "C" int dir_dice_train_backward_f32(const float* g, int64_t g_ld, const float* s, int64_t s_ld, int64_t B, int N, const float* alpha,
                                           const float* scale, const float* shift, const float* mean, const float* inv, float* ds, int64_t ds_ld,
                                           float* galpha, float* coef, float* partials, int64_t n_partials, dir_stream_t stream) {
    const char* name = "dir_dice_train_backward_f32";
    if (int rc = bn_check(name, B, N, s_ld, g_ld, ds_ld)) return rc;
    DIR_CHECK_ARG(g && s && alpha && scale && shift && mean && inv && ds && galpha && coef && partials, "%s: null pointer", name);
    if (!(aligned16(g) && aligned16(s) && aligned16(ds) && aligned16(coef) && aligned16(partials) && aligned16(alpha) && aligned16(scale) && aligned16(shift)))
        return fail(DIR_E_BADARG, "%s: g / s / ds / coef / partials / alpha / scale / shift must be 16-byte aligned", name);
    const BnPlan p = bn_plan(B, N);
    DIR_CHECK_ARG(n_partials >= p.nblk, "%s: partials holds %lld row triples, dir_bn_train_partials(B, N) = %lld", name, (long long)n_partials,
                  (long long)p.nblk);
    hipStream_t st = as_stream(stream);
#define DIR_DICE_PASS(T)                                                                                                                          \
    do {                                                                                                                                          \
        hipLaunchKernelGGL((dice_cols_k<T>), dim3((unsigned)p.nblk), dim3(256), 0, st, g, g_ld, s, s_ld, B, N, alpha, scale, shift, p.rows_per_block, \
                           partials);                                                                                                             \
        hipLaunchKernelGGL(bn_fin_bwd_k<3>, dim3((unsigned)((N + BN_FIN_COLS - 1) / BN_FIN_COLS)), dim3(256), 0, st, partials, p.nblk, B, N, mean, inv, \
                           (const float*)nullptr, coef, (float*)nullptr, (float*)nullptr, galpha);                                                \
        hipLaunchKernelGGL((dice_bwd_apply_k<T>), dim3((unsigned)p.nblk), dim3(256), 0, st, g, g_ld, s, s_ld, coef, alpha, scale, shift, B, N,     \
                           p.rows_per_block, ds, ds_ld);                                                                                          \
    } while (0)
    if (p.tpr == 64) DIR_DICE_PASS(64);
    else if (p.tpr == 128) DIR_DICE_PASS(128);
    else DIR_DICE_PASS(256);
#undef DIR_DICE_PASS
    DIR_CHECK_LAUNCH(name);
    return DIR_OK;
}
