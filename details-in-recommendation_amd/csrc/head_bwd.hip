// head_bwd.hip -- backward of the units = 1 logit layer taken straight through the ReLU of the hidden layer below it.
//
// Reference: the logits layer of dnn_logit_fn (models/DeepFM/deepFM.py:311-317) and of ESMM's _base_model (models/ESMM/ESMM.py:146) on
// top of a ReLU hidden layer (deepFM.py:293-300, ESMM.py:137-142); the reference trains through TensorFlow autodiff of those lines.
// With logit[b] = sum_n y[b,n] * w[n] + bias and y = relu(pre):
//   dpre[b,n] = (y[b,n] > 0) ? g[b] * w[n] : 0        g = dL/dlogit [B]
//   dw[n]     = sum_b g[b] * y[b,n]
//   dbias_y[n]= sum_b dpre[b,n]                       (the bias gradient of the hidden layer)
// As torch ops this is five passes over [B, N] (outer product, g*y, compare, mask multiply, two column sums: 189 us at 65 536 x 400);
// here y is read once and dpre written once.  HBM-bound: 8 bytes per element.
//
// A workgroup owns a span of rows; thread (rr, c) walks rows rr, rr + RPI, ... of the span for the float4 column chunks c, c + TPR, ...
// and keeps both column sums in registers (rows in ascending order); the RPI row lanes are added in lane order through LDS and the
// workgroup writes one partial row pair part[block][2][N].  The caller adds the partials (a fixed order: bitwise reproducible).
#include "common.hpp"

namespace dir {

constexpr int HB_MAXCH = 4;          // float4 column chunks per thread: N <= 4 * 4 * 256

// NCH: float4 column chunks per thread (1 while N <= 4 TPR; the chunk arrays are registers: four of them for a one-chunk width cost occupancy)
template <int TPR, int NCH = HB_MAXCH>
__global__ __launch_bounds__(256) void head_bwd_k(const float* __restrict__ g, const float* __restrict__ w, const float* __restrict__ y,
                                                   int64_t y_ld, int64_t B, int N, int64_t rows_per_block, float* __restrict__ gx,
                                                   int64_t gx_ld, float* __restrict__ part,
                                                   unsigned int* __restrict__ row_bits = nullptr /* [B]: bit pattern of an upper bound of max_n |gx[r, n]| */,
                                                   unsigned int* __restrict__ all_bits = nullptr /* the same over all rows: atomicMax, one per workgroup (zeroed by the launcher) */) {
    constexpr int RPI = 256 / TPR;
    __shared__ float4 red[2][256];
    const int tid = threadIdx.x, c0 = tid % TPR, rr = tid / TPR;
    const int nv = N >> 2;
    const int64_t r0 = (int64_t)blockIdx.x * rows_per_block;
    const int64_t r1 = min(B, r0 + rows_per_block);
    float4 wv[NCH], sx[NCH], sw[NCH];
#pragma unroll
    for (int ch = 0; ch < NCH; ++ch) {
        const int c = c0 + ch * TPR;
        wv[ch] = c < nv ? *reinterpret_cast<const float4*>(w + 4 * c) : make_float4(0.f, 0.f, 0.f, 0.f);
        sx[ch] = sw[ch] = make_float4(0.f, 0.f, 0.f, 0.f);
    }
    // row_bits: |g[r]| * max_n |w[n]| bounds row r of gx (it IS the row's maximum when the unit of the largest |w| is active; the fp32
    // product is monotonic, so the bound is never below a |g[r] * w[n]|) -- what the row-scaled fp16 x 2 kernel needs, without a pass over gx
    float wmax = 0.f;
    if (row_bits) {                                  // (uniform)
        float m = 0.f;
#pragma unroll
        for (int ch = 0; ch < NCH; ++ch)
            m = fmaxf(fmaxf(m, fmaxf(fabsf(wv[ch].x), fabsf(wv[ch].y))), fmaxf(fabsf(wv[ch].z), fabsf(wv[ch].w)));
        float* rf = reinterpret_cast<float*>(&red[0][0]);
        rf[tid] = m;
        __syncthreads();
        for (int o = 128; o > 0; o >>= 1) {
            if (tid < o) rf[tid] = fmaxf(rf[tid], rf[tid + o]);
            __syncthreads();
        }
        wmax = rf[0];
        __syncthreads();
    }
    float gmax = 0.f;                                // all_bits: the largest bound of this workgroup's rows, ONE atomic per workgroup at the end
    // HB_UNR rows of a thread are loaded before any of them is used (one row per trip left one 16-byte load per thread in flight: 2 TB/s of
    // y at four workgroups per CU); the sums still take the rows in ascending order: the same bits as the one-row loop.
    constexpr int HB_UNR = 4;
    for (int64_t rb = r0 + rr; rb < r1; rb += HB_UNR * RPI) {
        float gr[HB_UNR];
        float4 yv[HB_UNR][NCH];
#pragma unroll
        for (int u = 0; u < HB_UNR; ++u) {
            const int64_t r = rb + u * RPI;
            gr[u] = r < r1 ? g[r] : 0.f;
#pragma unroll
            for (int ch = 0; ch < NCH; ++ch) {
                const int c = c0 + ch * TPR;
                if (c < nv && r < r1) yv[u][ch] = *reinterpret_cast<const float4*>(y + r * y_ld + 4 * c);
            }
        }
#pragma unroll
        for (int u = 0; u < HB_UNR; ++u) {
            const int64_t r = rb + u * RPI;
            if (r >= r1) break;
            if (row_bits && c0 == 0) {
                const float bnd = fabsf(gr[u] * wmax);
                row_bits[r] = __builtin_bit_cast(unsigned int, bnd);
                gmax = fmaxf(gmax, bnd);
            }
#pragma unroll
            for (int ch = 0; ch < NCH; ++ch) {
                const int c = c0 + ch * TPR;
                if (c < nv) {
                    const float4 t = yv[u][ch];
                    float4 v;
                    v.x = t.x > 0.f ? gr[u] * wv[ch].x : 0.f;
                    v.y = t.y > 0.f ? gr[u] * wv[ch].y : 0.f;
                    v.z = t.z > 0.f ? gr[u] * wv[ch].z : 0.f;
                    v.w = t.w > 0.f ? gr[u] * wv[ch].w : 0.f;
                    *reinterpret_cast<float4*>(gx + r * gx_ld + 4 * c) = v;
                    sx[ch].x += v.x; sx[ch].y += v.y; sx[ch].z += v.z; sx[ch].w += v.w;
                    sw[ch].x += gr[u] * t.x; sw[ch].y += gr[u] * t.y; sw[ch].z += gr[u] * t.z; sw[ch].w += gr[u] * t.w;
                }
            }
        }
    }
    if (all_bits) {                                  // (uniform; workgroup 0 walking all of g instead took 60 us at B = 65 536: a serial chain of loads)
        float* rf = reinterpret_cast<float*>(&red[0][0]);
        rf[tid] = gmax;
        __syncthreads();
        for (int o = 128; o > 0; o >>= 1) {
            if (tid < o) rf[tid] = fmaxf(rf[tid], rf[tid + o]);
            __syncthreads();
        }
        if (tid == 0 && rf[0] > 0.f) atomicMax(all_bits, __builtin_bit_cast(unsigned int, rf[0]));
        __syncthreads();
    }
    float* p0 = part + (int64_t)blockIdx.x * 2 * N;
#pragma unroll
    for (int ch = 0; ch < NCH; ++ch) {
        const int c = c0 + ch * TPR;
        if (ch * TPR < nv) {                        // uniform: the chunk exists for some thread
            red[0][tid] = sx[ch];
            red[1][tid] = sw[ch];
            __syncthreads();
            if (rr == 0 && c < nv) {
                float4 a = red[0][c0], b = red[1][c0];
#pragma unroll
                for (int q = 1; q < RPI; ++q) {
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

// The same layer on top of an activation that is NOT a ReLU output and whose width need not be a multiple of 4 -- DCN's cross output
// x_L [B, d] (d = 26 x 16 + 13 = 429) under the final dense(1) over concat([cross, deep]) (DeepCrossNetwork.py:136-137):
//   gx[b,n] = g[b] * w[n],   dw[n] = sum_b g[b] * x[b,n]
// Thread = one column (4-byte loads, consecutive threads on consecutive columns), a workgroup = a span of rows walked in ascending order,
// four rows in flight; per-workgroup partial sums part[block][N], added in workgroup order by head_lin_fin_k (fp64: bitwise reproducible).
// (As torch ops the column sum of g * x at 65 536 x 429 alone took 0.68 ms.)
__global__ __launch_bounds__(256) void head_lin_bwd_k(const float* __restrict__ g, const float* __restrict__ w, const float* __restrict__ x,
                                                       int64_t x_ld, int64_t B, int N, int64_t rows_per_block, float* __restrict__ gx,
                                                       int64_t gx_ld, float* __restrict__ part) {
    const int64_t r0 = (int64_t)blockIdx.x * rows_per_block;
    const int64_t r1 = min(B, r0 + rows_per_block);
    for (int n = threadIdx.x; n < N; n += 256) {
        const float wn = w[n];
        float acc = 0.f;
        int64_t r = r0;
        for (; r + 4 <= r1; r += 4) {
            const float g0 = g[r], g1 = g[r + 1], g2 = g[r + 2], g3 = g[r + 3];
            const float x0 = x[r * x_ld + n], x1 = x[(r + 1) * x_ld + n], x2 = x[(r + 2) * x_ld + n], x3 = x[(r + 3) * x_ld + n];
            if (gx) {
                gx[r * gx_ld + n] = g0 * wn; gx[(r + 1) * gx_ld + n] = g1 * wn; gx[(r + 2) * gx_ld + n] = g2 * wn; gx[(r + 3) * gx_ld + n] = g3 * wn;
            }
            acc += g0 * x0; acc += g1 * x1; acc += g2 * x2; acc += g3 * x3;
        }
        for (; r < r1; ++r) {
            const float gr = g[r];
            if (gx) gx[r * gx_ld + n] = gr * wn;
            acc += gr * x[r * x_ld + n];
        }
        part[(int64_t)blockIdx.x * N + n] = acc;
    }
}

// 32 columns per workgroup, eight lanes of partials per column added in lane order (as bn_train.hip's bn_col_sums)
__global__ __launch_bounds__(256) void head_lin_fin_k(const float* __restrict__ part, int64_t P, int N, float* __restrict__ dw) {
    __shared__ double red[8][32];
    const int c = threadIdx.x & 31, l = threadIdx.x >> 5;
    const int n = blockIdx.x * 32 + c;
    double s = 0.0;
    if (n < N) {
#pragma unroll 4
        for (int64_t p = l; p < P; p += 8) s += (double)part[p * N + n];
    }
    red[l][c] = s;
    __syncthreads();
    if (l != 0 || n >= N) return;
#pragma unroll
    for (int q = 1; q < 8; ++q) s += red[q][c];
    dw[n] = (float)s;
}

static int64_t hl_blocks(int64_t B) {
    int64_t nblk = (B + 63) / 64;
    if (nblk > 4 * kCUs) nblk = 4 * kCUs;
    return nblk < 1 ? 1 : nblk;
}

struct HbPlan { int tpr; int64_t nblk, rows_per_block; };
static HbPlan hb_plan(int64_t B, int N) {
    HbPlan p;
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

// The FORWARD of a units = 1 layer: y[b] = x[b, :] . w + bias -- the logit heads of the towers in training (deepFM.py:311-317, ESMM.py:146:
// the hidden activation has to be kept for the backward, so the fused head of the inference kernels does not apply), DCN's final dense(1)
// over the cross output (DeepCrossNetwork.py:136-137) and xDeepFM's CIN output layer.  One pass over x (the library ran these
// [B, N] x [N, 1] products as a GEMM with one output column: 26 us at 65 536 x 400 = 4 TB/s of x; its GEMV: 62 us).  LPR lanes share a row
// (16-byte loads where rows are 16-byte aligned, 4-byte loads otherwise), partial sums meet in a fixed butterfly: bitwise reproducible.
template <int VEC, int LPR>
__global__ __launch_bounds__(256) void units1_fwd_k(const float* __restrict__ x, int64_t x_ld, int64_t B, int N, const float* __restrict__ w,
                                                    const float* __restrict__ bias, float* __restrict__ y, int64_t y_ld) {
    constexpr int RPB = 256 / LPR;                       // rows per workgroup and step
    const int lane = threadIdx.x % LPR, rl = threadIdx.x / LPR;
    const float b0 = bias ? bias[0] : 0.f;
    for (int64_t r0 = (int64_t)blockIdx.x * RPB; r0 < B; r0 += (int64_t)gridDim.x * RPB) {
        const int64_t r = r0 + rl;
        float acc = 0.f;
        if (r < B) {
            const float* __restrict__ xr = x + r * x_ld;
            if (VEC == 4) {
                float4 a4 = make_float4(0.f, 0.f, 0.f, 0.f);
                for (int k = lane * 4; k < N; k += LPR * 4) {
                    const float4 xv = *reinterpret_cast<const float4*>(xr + k);
                    const float4 wv = *reinterpret_cast<const float4*>(w + k);
                    a4.x = fmaf(xv.x, wv.x, a4.x);
                    a4.y = fmaf(xv.y, wv.y, a4.y);
                    a4.z = fmaf(xv.z, wv.z, a4.z);
                    a4.w = fmaf(xv.w, wv.w, a4.w);
                }
                acc = (a4.x + a4.y) + (a4.z + a4.w);
            } else {
                for (int k = lane; k < N; k += LPR) acc = fmaf(xr[k], w[k], acc);
            }
        }
#pragma unroll
        for (int o = LPR / 2; o > 0; o >>= 1) acc += __shfl_xor(acc, o, LPR);
        if (lane == 0 && r < B) y[r * y_ld] = acc + b0;
    }
}

}  // namespace dir

using namespace dir;

extern "C" int dir_units1_f32(const float* x, int64_t x_ld, int64_t B, int N, const float* w, const float* bias, float* y, int64_t y_ld,
                              dir_stream_t stream) {
    const char* name = "dir_units1_f32";
    DIR_CHECK_ARG(B >= 0 && N > 0, "%s: B=%lld N=%d", name, (long long)B, N);
    if (B == 0) return DIR_OK;
    DIR_CHECK_ARG(x && w && y && x_ld >= N && y_ld >= 1, "%s: null pointer or a row stride smaller than the width", name);
    hipStream_t st = as_stream(stream);
    const bool vec = N % 4 == 0 && x_ld % 4 == 0 && aligned16(x) && aligned16(w);
    if (vec && N <= 256)
        hipLaunchKernelGGL((units1_fwd_k<4, 16>), dim3(grid_for((B + 15) / 16)), dim3(256), 0, st, x, x_ld, B, N, w, bias, y, y_ld);
    else if (vec)
        hipLaunchKernelGGL((units1_fwd_k<4, 32>), dim3(grid_for((B + 7) / 8)), dim3(256), 0, st, x, x_ld, B, N, w, bias, y, y_ld);
    else
        hipLaunchKernelGGL((units1_fwd_k<1, 64>), dim3(grid_for((B + 3) / 4)), dim3(256), 0, st, x, x_ld, B, N, w, bias, y, y_ld);
    DIR_CHECK_LAUNCH(name);
    return DIR_OK;
}

extern "C" int64_t dir_units1_relu_backward_partials(int64_t B, int N) {
    if (B <= 0 || N <= 0) return 0;
    return hb_plan(B, N).nblk;
}

static int units1_relu_backward(const char* name, const float* g, const float* w, const float* y, int64_t y_ld, int64_t B, int N, float* gx,
                                int64_t gx_ld, float* partials, int64_t n_partials, unsigned int* row_bits, unsigned int* all_bits,
                                dir_stream_t stream) {
    DIR_CHECK_ARG(B >= 0 && N > 0, "%s: B=%lld N=%d", name, (long long)B, N);
    if (B == 0) return DIR_OK;
    DIR_CHECK_ARG(g && w && y && gx && partials, "%s: null pointer", name);
    if (N % 4 || y_ld % 4 || gx_ld % 4 || N > 4 * HB_MAXCH * 256)
        return fail(DIR_E_UNSUPPORTED, "%s: N=%d y_ld=%lld gx_ld=%lld (multiples of 4, N <= %d)", name, N, (long long)y_ld, (long long)gx_ld,
                    4 * HB_MAXCH * 256);
    DIR_CHECK_ARG(y_ld >= N && gx_ld >= N, "%s: row strides smaller than N", name);
    if (!(aligned16(w) && aligned16(y) && aligned16(gx) && aligned16(partials)))
        return fail(DIR_E_BADARG, "%s: w / y / gx / partials must be 16-byte aligned", name);
    const HbPlan p = hb_plan(B, N);
    DIR_CHECK_ARG(n_partials >= p.nblk, "%s: partials holds %lld row pairs, dir_units1_relu_backward_partials(B, N) = %lld", name,
                  (long long)n_partials, (long long)p.nblk);
    hipStream_t st = as_stream(stream);
    if (all_bits && zero_async(all_bits, sizeof(unsigned int), st) != hipSuccess) return fail(DIR_E_HIP, "%s: zeroing failed", name);
#define HB_LAUNCH(TPR_, NCH_)                                                                                                                  \
    hipLaunchKernelGGL((head_bwd_k<TPR_, NCH_>), dim3((unsigned)p.nblk), dim3(256), 0, st, g, w, y, y_ld, B, N, p.rows_per_block, gx, gx_ld, partials, \
                       row_bits, all_bits)
    const int nch = (N / 4 + p.tpr - 1) / p.tpr;          // 1 unless N > 1024 (tpr = 256 then)
    if (p.tpr == 64) HB_LAUNCH(64, 1);
    else if (p.tpr == 128) HB_LAUNCH(128, 1);
    else if (nch <= 1) HB_LAUNCH(256, 1);
    else if (nch == 2) HB_LAUNCH(256, 2);
    else HB_LAUNCH(256, 4);
#undef HB_LAUNCH
    DIR_CHECK_LAUNCH(name);
    return DIR_OK;
}

extern "C" int dir_units1_relu_backward_f32(const float* g, const float* w, const float* y, int64_t y_ld, int64_t B, int N, float* gx,
                                            int64_t gx_ld, float* partials, int64_t n_partials, dir_stream_t stream) {
    return units1_relu_backward("dir_units1_relu_backward_f32", g, w, y, y_ld, B, N, gx, gx_ld, partials, n_partials, nullptr, nullptr, stream);
}

// ... and the bit patterns of an upper bound of every gx row's largest |element| (|g[r]| max_n |w[n]|) and of their maximum (all_bits: one
// unsigned, zeroed here): the scales of the fp16 x 2 backward kernels that consume gx (dir_dense_f16x2_rows_f32,
// dir_dense_dw_f16x2_f32), without a pass over gx
extern "C" int dir_units1_relu_backward_bits_f32(const float* g, const float* w, const float* y, int64_t y_ld, int64_t B, int N, float* gx,
                                                 int64_t gx_ld, float* partials, int64_t n_partials, unsigned int* gx_row_bits,
                                                 unsigned int* gx_all_bits, dir_stream_t stream) {
    DIR_CHECK_ARG(B == 0 || (gx_row_bits && gx_all_bits), "dir_units1_relu_backward_bits_f32: null pointer");
    return units1_relu_backward("dir_units1_relu_backward_bits_f32", g, w, y, y_ld, B, N, gx, gx_ld, partials, n_partials, gx_row_bits, gx_all_bits,
                                stream);
}

extern "C" int64_t dir_units1_backward_partials(int64_t B, int N) {
    if (B <= 0 || N <= 0) return 0;
    return hl_blocks(B);
}

extern "C" int dir_units1_backward_f32(const float* g, const float* w, const float* x, int64_t x_ld, int64_t B, int N, float* gx, int64_t gx_ld,
                                       float* dw, float* partials, int64_t n_partials, dir_stream_t stream) {
    const char* name = "dir_units1_backward_f32";
    DIR_CHECK_ARG(B >= 0 && N > 0, "%s: B=%lld N=%d", name, (long long)B, N);
    DIR_CHECK_ARG(dw, "%s: null pointer", name);
    hipStream_t st = as_stream(stream);
    if (B == 0) {
        if (zero_async(dw, sizeof(float) * N, st) != hipSuccess) return fail(DIR_E_HIP, "%s: memset failed", name);
        return DIR_OK;
    }
    DIR_CHECK_ARG(g && w && x && partials, "%s: null pointer", name);
    DIR_CHECK_ARG(x_ld >= N && (!gx || gx_ld >= N), "%s: row strides smaller than N", name);
    const int64_t nblk = hl_blocks(B);
    DIR_CHECK_ARG(n_partials >= nblk, "%s: partials holds %lld rows, dir_units1_backward_partials(B, N) = %lld", name, (long long)n_partials,
                  (long long)nblk);
    const int64_t rpb = (B + nblk - 1) / nblk;
    hipLaunchKernelGGL(head_lin_bwd_k, dim3((unsigned)nblk), dim3(256), 0, st, g, w, x, x_ld, B, N, rpb, gx, gx_ld, partials);
    DIR_CHECK_LAUNCH(name);
    hipLaunchKernelGGL(head_lin_fin_k, dim3((unsigned)((N + 31) / 32)), dim3(256), 0, st, partials, (B + rpb - 1) / rpb, N, dw);
    DIR_CHECK_LAUNCH(name);
    return DIR_OK;
}

// ESMM's prediction head in inference (models/ESMM/ESMM.py:67-77, stated here as in esmm.ESMM.forward): p = sigmoid(ctr) * sigmoid(cvr),
// clipped to [1e-7, 1 - 1e-7], ctcvr_logit = log(p / (1 - p)) -- seven elementwise library launches over [B, 1] as one.  The operations and
// their order are the reference's; exp / log are the device library's single-precision functions.
namespace dir {
__global__ __launch_bounds__(256) void esmm_head_k(const float* __restrict__ ctr, const float* __restrict__ cvr, int64_t B, float eps,
                                                   float* __restrict__ ctcvr_logit) {
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < B; i += (int64_t)gridDim.x * 256) {
        const float a = 1.0f / (1.0f + expf(-ctr[i])), b = 1.0f / (1.0f + expf(-cvr[i]));
        float p = a * b;
        p = fminf(fmaxf(p, eps), 1.0f - eps);
        ctcvr_logit[i] = logf(p / (1.0f - p));
    }
}
}  // namespace dir

extern "C" int dir_esmm_head_f32(const float* ctr_logit, const float* cvr_logit, int64_t B, float eps, float* ctcvr_logit, dir_stream_t stream) {
    const char* name = "dir_esmm_head_f32";
    DIR_CHECK_ARG(B >= 0 && eps > 0.f && eps < 0.5f, "%s: B=%lld eps=%g", name, (long long)B, eps);
    if (B == 0) return DIR_OK;
    DIR_CHECK_ARG(ctr_logit && cvr_logit && ctcvr_logit, "%s: null pointer", name);
    hipLaunchKernelGGL(dir::esmm_head_k, dim3(grid_for((B + 255) / 256)), dim3(256), 0, as_stream(stream), ctr_logit, cvr_logit, B, eps, ctcvr_logit);
    DIR_CHECK_LAUNCH(name);
    return DIR_OK;
}
