// backward.hip -- gradients of the HBM-bound interaction ops, gfx950.
//
// The reference trains through TensorFlow's autodiff of the closures this library replaces:
//   fm_logit_fn                     models/DeepFM/deepFM.py:321-335
//   _cross_op / _cross_architecture models/DeepCrossNetwork/DeepCrossNetwork.py:336-367
// so the backward formulas below are the derivatives of exactly those expressions.
//
//   FM:    y_b = 0.5 * sum_k[(sum_f e)^2 - sum_f e^2]   =>   dL/de[b,f,k] = g_b * (S[b,k] - e[b,f,k]),  S = sum_f e
//   cross: x_{l+1} = x0 * (x_l . w_l) + b_l + x_l ; with g = dL/dx_{l+1}, s_l = x_l . w_l, t = g . x0:
//              db_l += g ;  dw_l += t * x_l ;  dx0 += s_l * g ;  dL/dx_l = g + t * w_l ;  finally dx0 += dL/dx_0
//
// Both are streaming kernels.  FM backward keeps the forward's lane mapping (K/4 lanes per sample) and can add
// the gradient arriving from the DNN branch in the same pass (the total d/d(embedding) of DeepFM in one launch).
// Cross backward gives one row to a wave: x_0..x_L of the row are recomputed into an LDS slab, the weight /
// bias gradients accumulate in per-wave LDS slabs (lane-private addresses, no atomics), each workgroup writes
// one partial [2, L, d] and a second launch adds the partials in a fixed order (bitwise reproducible).
#include <cstdlib>
#include <cstring>
#if defined(DIR_WITH_ROCPRIM_SORT)          // development builds only (A/B against the in-tree sort): tools/ab_variant.sh backward.hip rp -DDIR_WITH_ROCPRIM_SORT
#include <rocprim/device/device_radix_sort.hpp>
#endif
#include "common.hpp"
#include <type_traits>

namespace dir {

template <int VEC> struct BV;
template <> struct BV<4> {
    using T = float4;
    static __device__ __forceinline__ T ld(const float* p) { return *reinterpret_cast<const float4*>(p); }
    static __device__ __forceinline__ void st(float* p, T v) { *reinterpret_cast<float4*>(p) = v; }
    static __device__ __forceinline__ T zero() { return make_float4(0.f, 0.f, 0.f, 0.f); }
    static __device__ __forceinline__ T add(T a, T b) { return make_float4(a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w); }
    static __device__ __forceinline__ T sub(T a, T b) { return make_float4(a.x - b.x, a.y - b.y, a.z - b.z, a.w - b.w); }
    static __device__ __forceinline__ T scale(T a, float s) { return make_float4(a.x * s, a.y * s, a.z * s, a.w * s); }
    static __device__ __forceinline__ T fma(T a, float s, T c) { return make_float4(fmaf(a.x, s, c.x), fmaf(a.y, s, c.y), fmaf(a.z, s, c.z), fmaf(a.w, s, c.w)); }
    static __device__ __forceinline__ float dot(T a, T b, float acc) {
        acc = fmaf(a.x, b.x, acc); acc = fmaf(a.y, b.y, acc); acc = fmaf(a.z, b.z, acc); acc = fmaf(a.w, b.w, acc);
        return acc;
    }
    static __device__ __forceinline__ T upd(T x0, float xw, T b, T xl) {
        return make_float4(((x0.x * xw) + b.x) + xl.x, ((x0.y * xw) + b.y) + xl.y, ((x0.z * xw) + b.z) + xl.z, ((x0.w * xw) + b.w) + xl.w);
    }
};
template <> struct BV<1> {
    using T = float;
    static __device__ __forceinline__ T ld(const float* p) { return *p; }
    static __device__ __forceinline__ void st(float* p, T v) { *p = v; }
    static __device__ __forceinline__ T zero() { return 0.f; }
    static __device__ __forceinline__ T add(T a, T b) { return a + b; }
    static __device__ __forceinline__ T sub(T a, T b) { return a - b; }
    static __device__ __forceinline__ T scale(T a, float s) { return a * s; }
    static __device__ __forceinline__ T fma(T a, float s, T c) { return fmaf(a, s, c); }
    static __device__ __forceinline__ float dot(T a, T b, float acc) { return fmaf(a, b, acc); }
    static __device__ __forceinline__ T upd(T x0, float xw, T b, T xl) { return ((x0 * xw) + b) + xl; }
};

// ------------------------------------------------------------------------------------------------
// FM backward: demb[b, f, :] = g[b] * (S[b, :] - e[b, f, :]) (+ add_in[b, f, :])
// ------------------------------------------------------------------------------------------------
template <int LPS, int VEC>
__global__ __launch_bounds__(256) void fm_bwd_k(const float* __restrict__ emb, int64_t ld, const float* __restrict__ g,
                                                const float* __restrict__ add_in, int64_t ld_a, int64_t B, int F, int K,
                                                float* __restrict__ demb, int64_t ld_o) {
    using V = BV<VEC>;
    using T = typename V::T;
    constexpr int SPW = 64 / LPS;
    const int lane = threadIdx.x & 63;
    const int c = lane & (LPS - 1);
    const int s = lane / LPS;
    const int kv = (K + VEC - 1) / VEC;
    const bool cact = c < kv;
    const int64_t nwave = (int64_t)gridDim.x * (blockDim.x >> 6);
    for (int64_t w = (int64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6); w * SPW < B; w += nwave) {
        const int64_t b = w * SPW + s;
        const bool act = cact && b < B;
        if (!act) continue;
        const float* ep = emb + b * ld + c * VEC;
        T S = V::zero();
#pragma unroll 8
        for (int f = 0; f < F; ++f) S = V::add(S, V::ld(ep + (int64_t)f * K));   // f ascending, as the forward
        const float gb = g[b];
        float* op = demb + b * ld_o + c * VEC;
        const float* ap = add_in ? add_in + b * ld_a + c * VEC : nullptr;
#pragma unroll 8
        for (int f = 0; f < F; ++f) {
            T d = V::scale(V::sub(S, V::ld(ep + (int64_t)f * K)), gb);
            if (ap) d = V::add(d, V::ld(ap + (int64_t)f * K));
            V::st(op + (int64_t)f * K, d);
        }
    }
}

// ------------------------------------------------------------------------------------------------
// cross backward.  One row per wave; NV chunks of VEC floats per lane (chunk ch = i*64 + lane).
// LDS: wb [2][L][d] (shared), then per wave: xs [(L+1)][d] (x_0..x_L of the current row), gacc [2][L][d].
// ------------------------------------------------------------------------------------------------
template <int NV, int VEC>
__global__ __launch_bounds__(256) void cross_bwd_k(const float* __restrict__ x0, int64_t x_ld,
                                                   const float* __restrict__ w, const float* __restrict__ bvec, int L,
                                                   const float* __restrict__ gout, int64_t g_ld, int64_t B, int d,
                                                   float* __restrict__ gx0, int64_t gx_ld,
                                                   float* __restrict__ partial /* [gridDim.x][2][L][d] */) {
    using V = BV<VEC>;
    using T = typename V::T;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int Ld = L * d;
    const int nw = blockDim.x >> 6;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    float* wsh = smem;                       // [L][d]
    float* bsh = smem + Ld;                  // [L][d]
    const int Lp = (L + 3) & ~3;
    const int wstride = (L + 1) * d + 2 * Ld + Lp;               // floats per wave slab
    float* xs = smem + 2 * Ld + wave * wstride;                  // [(L+1)][d]
    float* gw = xs + (L + 1) * d;            // [L][d]
    float* gb = gw + Ld;                     // [L][d]
    float* sls = gb + Ld;                    // [L]: s_l = x_l . w_l of the current row
    for (int i = threadIdx.x; i < Ld; i += blockDim.x) {
        wsh[i] = w[i];
        bsh[i] = bvec[i];
    }
    for (int i = lane; i < 2 * Ld; i += 64) gw[i] = 0.f;   // gw and gb are contiguous
    __syncthreads();
    const int nchunk = d / VEC;
    const int64_t nwave = (int64_t)gridDim.x * nw;
    for (int64_t r = (int64_t)blockIdx.x * nw + wave; r < B; r += nwave) {
        const float* xp = x0 + r * x_ld;
        const float* gp = gout + r * g_ld;
        T xv[NV], g[NV], gx[NV];
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const int ch = i * 64 + lane;
            const bool ok = ch < nchunk;
            xv[i] = ok ? V::ld(xp + ch * VEC) : V::zero();
            g[i] = ok ? V::ld(gp + ch * VEC) : V::zero();
            gx[i] = V::zero();
            if (ok) V::st(xs + ch * VEC, xv[i]);          // x_0
        }
        // forward recompute: x_1..x_L and s_0..s_{L-1} into the wave's slab
#pragma unroll 1
        for (int l = 0; l < L; ++l) {
            float part = 0.f;
#pragma unroll
            for (int i = 0; i < NV; ++i) {
                const int ch = i * 64 + lane;
                if (ch < nchunk) part = V::dot(V::ld(xs + l * d + ch * VEC), V::ld(wsh + l * d + ch * VEC), part);
            }
            const float s = wave_sum(part);
            if (lane == 0) sls[l] = s;
#pragma unroll
            for (int i = 0; i < NV; ++i) {
                const int ch = i * 64 + lane;
                if (ch < nchunk)
                    V::st(xs + (l + 1) * d + ch * VEC,
                          V::upd(xv[i], s, V::ld(bsh + l * d + ch * VEC), V::ld(xs + l * d + ch * VEC)));
            }
        }
        // backward through the layers
#pragma unroll 1
        for (int l = L - 1; l >= 0; --l) {
            float tp = 0.f;
#pragma unroll
            for (int i = 0; i < NV; ++i) tp = V::dot(g[i], xv[i], tp);
            const float t = wave_sum(tp);
            const float s = sls[l];
#pragma unroll
            for (int i = 0; i < NV; ++i) {
                const int ch = i * 64 + lane;
                if (ch < nchunk) {
                    const T xl = V::ld(xs + l * d + ch * VEC);
                    V::st(gb + l * d + ch * VEC, V::add(V::ld(gb + l * d + ch * VEC), g[i]));          // db_l += g
                    V::st(gw + l * d + ch * VEC, V::fma(xl, t, V::ld(gw + l * d + ch * VEC)));         // dw_l += t * x_l
                    gx[i] = V::fma(g[i], s, gx[i]);                                                    // dx0 += s_l * g
                    g[i] = V::fma(V::ld(wsh + l * d + ch * VEC), t, g[i]);                             // dL/dx_l
                }
            }
        }
        float* op = gx0 + r * gx_ld;
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const int ch = i * 64 + lane;
            if (ch < nchunk) V::st(op + ch * VEC, V::add(gx[i], g[i]));                                // + dL/dx_0
        }
    }
    __syncthreads();
    // workgroup partial = sum of the per-wave slabs in wave order
    float* pout = partial + (int64_t)blockIdx.x * 2 * Ld;
    for (int i = threadIdx.x; i < 2 * Ld; i += blockDim.x) {
        float acc = 0.f;
        for (int wv = 0; wv < nw; ++wv) acc += smem[2 * Ld + wv * wstride + (L + 1) * d + i];
        pout[i] = acc;
    }
}

// Register-resident variant for the common shapes (1 <= L <= 4, a row fits NV chunks per lane): one row per wave, the
// lane keeps its chunks of x_0..x_{L-1}, of the running gradient and of dL/dx0 in registers, and its share of dw / db as
// register accumulators over all the rows the wave visits -- no LDS in the row loop (the slab kernel above makes ~10 LDS
// accesses per chunk and layer: 279 us at d = 416 against a 65 us traffic floor).  The next row is loaded before the
// current one is processed; two DPP wave sums per layer.  Same per-workgroup partial layout as the slab kernel.
template <int NV, int VEC, int LT>
__global__ __launch_bounds__(256) void cross_bwd_reg_k(const float* __restrict__ x0, int64_t x_ld, const float* __restrict__ w,
                                                       const float* __restrict__ bvec, const float* __restrict__ gout, int64_t g_ld,
                                                       int64_t B, int d, float* __restrict__ gx0, int64_t gx_ld,
                                                       float* __restrict__ partial /* [gridDim.x][2][LT][d] */) {
    using V = BV<VEC>;
    using T = typename V::T;
    extern __shared__ __attribute__((aligned(16))) float smem[];   // [4 waves][2][LT][d]: only for the final cross-wave sum
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int nchunk = d / VEC;
    bool ok[NV];
    T wv[LT][NV], bv[LT][NV], aw[LT][NV], ab[LT][NV];
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const int ch = i * 64 + lane;
        ok[i] = ch < nchunk;
#pragma unroll
        for (int l = 0; l < LT; ++l) {
            wv[l][i] = ok[i] ? V::ld(w + l * d + ch * VEC) : V::zero();
            bv[l][i] = ok[i] ? V::ld(bvec + l * d + ch * VEC) : V::zero();
            aw[l][i] = V::zero();
            ab[l][i] = V::zero();
        }
    }
    const int64_t nwave = (int64_t)gridDim.x * 4;
    int64_t r = (int64_t)blockIdx.x * 4 + wave;
    T xn[NV], gn[NV];
    auto load_row = [&](int64_t row) {
        const int64_t rc = row < B ? row : B - 1;
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const int ch = i * 64 + lane;
            xn[i] = ok[i] ? V::ld(x0 + rc * x_ld + ch * VEC) : V::zero();
            gn[i] = ok[i] ? V::ld(gout + rc * g_ld + ch * VEC) : V::zero();
        }
    };
    if (r < B) load_row(r);
    for (; r < B; r += nwave) {
        T xl[LT][NV], g[NV], gx[NV];
        float sl[LT];
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            xl[0][i] = xn[i];
            g[i] = gn[i];
            gx[i] = V::zero();
        }
        load_row(r + nwave);                                   // lands while this row is processed
        // forward recompute: s_l = x_l . w_l, x_{l+1} = ((x0 * s_l) + b_l) + x_l   (DeepCrossNetwork.py:345-346)
#pragma unroll
        for (int l = 0; l < LT; ++l) {
            float part = 0.f;
#pragma unroll
            for (int i = 0; i < NV; ++i) part = V::dot(xl[l][i], wv[l][i], part);
            sl[l] = wave_sum_dpp(part);
            if (l + 1 < LT) {
#pragma unroll
                for (int i = 0; i < NV; ++i) xl[l + 1][i] = V::upd(xl[0][i], sl[l], bv[l][i], xl[l][i]);
            }
        }
        // backward through the layers
#pragma unroll
        for (int l = LT - 1; l >= 0; --l) {
            float tp = 0.f;
#pragma unroll
            for (int i = 0; i < NV; ++i) tp = V::dot(g[i], xl[0][i], tp);
            const float t = wave_sum_dpp(tp);                  // dL/ds_l
#pragma unroll
            for (int i = 0; i < NV; ++i) {
                ab[l][i] = V::add(ab[l][i], g[i]);             // db_l += g
                aw[l][i] = V::fma(xl[l][i], t, aw[l][i]);      // dw_l += t * x_l
                gx[i] = V::fma(g[i], sl[l], gx[i]);            // dx0 += s_l * g
                g[i] = V::fma(wv[l][i], t, g[i]);              // dL/dx_l
            }
        }
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const int ch = i * 64 + lane;
            if (ok[i]) V::st(gx0 + r * gx_ld + ch * VEC, V::add(gx[i], g[i]));      // + dL/dx_0
        }
    }
    // workgroup partial = the four waves' accumulators added in wave order
    const int Ld = LT * d;
    float* slab = smem + wave * 2 * Ld;
#pragma unroll
    for (int l = 0; l < LT; ++l)
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const int ch = i * 64 + lane;
            if (ok[i]) {
                V::st(slab + l * d + ch * VEC, aw[l][i]);
                V::st(slab + Ld + l * d + ch * VEC, ab[l][i]);
            }
        }
    __syncthreads();
    float* pout = partial + (int64_t)blockIdx.x * 2 * Ld;
    for (int i = threadIdx.x; i < 2 * Ld; i += blockDim.x)
        pout[i] = ((smem[i] + smem[2 * Ld + i]) + smem[4 * Ld + i]) + smem[6 * Ld + i];
}

// gw / gb = sum of the per-workgroup partials.  One WAVE per output element: lane j adds partials j, j+64, ... in order,
// then a fixed xor tree joins the 64 sums (reproducible).  A single thread per element spent 120 us on 512 dependent loads,
// eight lanes per element still 21 us; this is 8 dependent loads per lane.
__global__ __launch_bounds__(256) void reduce_partials_k(const float* __restrict__ partial, int nblk, int n, float* __restrict__ gw,
                                                         float* __restrict__ gb, int Ld) {
    const int lane = threadIdx.x & 63;
    const int i = blockIdx.x * 4 + (threadIdx.x >> 6);          // wave-uniform
    if (i >= n) return;
    float acc = 0.f;
    for (int b = lane; b < nblk; b += 64) acc += partial[(int64_t)b * n + i];
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) acc += __shfl_xor(acc, o, 64);
    if (lane == 0) {
        if (i < Ld) gw[i] = acc; else gb[i - Ld] = acc;
    }
}

// ------------------------------------------------------------------------------------------------
// Fused sparse Adagrad on the embedding tables (the reference's dnn_optimizer='Adagrad', deepFM.py:61, applied to
// the IndexedSlices gradients of its lookups).  [TF-upstream] semantics: gradients of duplicate ids are SUMMED
// first (Optimizer._apply_sparse_duplicate_indices), then  accum += g*g ;  var -= lr * g / sqrt(accum).
//
//   link : every entry e = b*F+f with id >= 0 pushes itself on the chain of its row:
//          next[e] = atomicExch(&head[base_f + id], e)            (one 4-byte integer atomic per entry)
//   apply: the entry that is still the chain head is the row's leader: it walks the chain, adds the gradient rows
//          (LPS lanes per row, 16 B per lane), updates accumulator and weights in place and resets head to -1.
// head is a persistent int32 array over all rows of all tables (-1 = no chain); no float atomics are used, and a
// row with <= 2 contributions is bitwise reproducible (a + b commutes); longer chains add in chain order.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void adagrad_link_k(const int64_t* __restrict__ ids, int64_t sb, int64_t sf, int F,
                                                      int64_t n, const int64_t* __restrict__ head_base, int64_t total_rows,
                                                      int32_t* __restrict__ head, int32_t* __restrict__ next) {
    for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < n; e += (int64_t)gridDim.x * blockDim.x) {
        const int64_t b = e / F;
        const int f = (int)(e - b * F);
        const int64_t id = ids[b * sb + f * sf];
        const int64_t vf = (f + 1 < F ? head_base[f + 1] : total_rows) - head_base[f];       // an id >= vocab_f is pruned, never written
        next[e] = (uint64_t)id < (uint64_t)vf ? atomicExch(&head[head_base[f] + id], (int32_t)e) : -2;
    }
}

template <int LPS, int VEC>
__global__ __launch_bounds__(256) void adagrad_apply_k(float* const* __restrict__ tables, float* const* __restrict__ accums,
                                                       const int64_t* __restrict__ ids, int64_t sb, int64_t sf, int F, int K,
                                                       int64_t n, const float* __restrict__ grad, int64_t g_ld, float lr,
                                                       const int64_t* __restrict__ head_base, int64_t total_rows,
                                                       int32_t* __restrict__ head, const int32_t* __restrict__ next) {
    using V = BV<VEC>;
    using T = typename V::T;
    const int kv = K / VEC;
    const int64_t tid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t nthr = (int64_t)gridDim.x * blockDim.x;
    for (int64_t q = tid; q < n * LPS; q += nthr) {
        const int64_t e = q / LPS;
        const int c = (int)(q - e * LPS);
        const int64_t b = e / F;
        const int f = (int)(e - b * F);
        const int64_t id = ids[b * sb + f * sf];
        const int64_t vf = (f + 1 < F ? head_base[f + 1] : total_rows) - head_base[f];
        if (!((uint64_t)id < (uint64_t)vf) || c >= kv) continue;
        int32_t* hp = head + head_base[f] + id;
        if (*hp != (int32_t)e) continue;                  // not the chain head: some other entry leads this row
        T g = V::zero();
        for (int32_t cur = (int32_t)e; cur >= 0; cur = next[cur]) {
            const int64_t cb = cur / F;
            const int cf = cur - (int)(cb * F);           // == f
            g = V::add(g, V::ld(grad + cb * g_ld + (int64_t)cf * K + c * VEC));
        }
        float* ap = accums[f] + id * K + c * VEC;
        float* wp = tables[f] + id * K + c * VEC;
        T acc = V::ld(ap), wv = V::ld(wp);
        if constexpr (VEC == 4) {
            acc = make_float4(acc.x + g.x * g.x, acc.y + g.y * g.y, acc.z + g.z * g.z, acc.w + g.w * g.w);
            wv = make_float4(wv.x - lr * g.x / sqrtf(acc.x), wv.y - lr * g.y / sqrtf(acc.y), wv.z - lr * g.z / sqrtf(acc.z),
                             wv.w - lr * g.w / sqrtf(acc.w));
        } else {
            acc = acc + g * g;
            wv = wv - lr * g / sqrtf(acc);
        }
        V::st(ap, acc);
        V::st(wp, wv);
        if (c == 0) *hp = -1;                             // unlink for the next step
    }
}

// ------------------------------------------------------------------------------------------------
// Sorted sparse Adagrad: the chain walk above serialises on hot rows (Zipf ids: one thread group sums thousands of
// gradient rows).  Here the (global row, entry) pairs are radix-sorted (rocPRIM, stable: duplicates keep their batch
// order, so every sum has a fixed order), a workgroup reduces the runs of equal rows inside its tile of 256 sorted
// entries and applies the update for runs that lie inside the tile; a run that crosses tile borders leaves one
// partial per tile (`carry`) and adagrad_fix_k adds them in tile order.  No atomics, bitwise reproducible, and the
// longest serial walk is 256 entries whatever the skew.
// ------------------------------------------------------------------------------------------------
constexpr int ADA_TILE = 256;

__global__ __launch_bounds__(256) void adagrad_keys_k(const int64_t* __restrict__ ids, int64_t sb, int64_t sf, int F, int64_t n,
                                                      const int64_t* __restrict__ row_base, uint32_t total_rows,
                                                      uint32_t* __restrict__ keys, uint32_t* __restrict__ vals) {
    for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < n; e += (int64_t)gridDim.x * 256) {
        const int64_t b = e / F;
        const int f = (int)(e - b * F);
        const int64_t id = ids[b * sb + f * sf];
        const int64_t vf = (f + 1 < F ? row_base[f + 1] : (int64_t)total_rows) - row_base[f];
        // pruned ids (id < 0, and id >= vocab_f: never written) sort behind every row
        keys[e] = (uint64_t)id < (uint64_t)vf ? (uint32_t)(row_base[f] + id) : total_rows;
        vals[e] = (uint32_t)e;
    }
}

// payload entries of the sharded lookup (p = local_row * F + slot, p < 0 pruned): the owner side of a sharded backward
__global__ __launch_bounds__(256) void adagrad_keys_payload_k(const int64_t* __restrict__ payload, int F, int64_t n,
                                                              const int64_t* __restrict__ row_base, uint32_t total_rows,
                                                              uint32_t* __restrict__ keys, uint32_t* __restrict__ vals) {
    for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < n; e += (int64_t)gridDim.x * 256) {
        const int64_t p = payload[e];
        const int f = p >= 0 ? (int)(p % F) : 0;
        const int64_t row = p >= 0 ? p / F : -1;
        const int64_t vf = (f + 1 < F ? row_base[f + 1] : (int64_t)total_rows) - row_base[f];
        keys[e] = (uint64_t)row < (uint64_t)vf ? (uint32_t)(row_base[f] + row) : total_rows;
        vals[e] = (uint32_t)e;
    }
}

// per-row update rules of the sorted path: g = the summed gradient chunk of one row
struct AdagradUpd {   // accum += g^2; w -= lr * g / sqrt(accum)   ([TF-upstream] tf.train.AdagradOptimizer, deepFM.py:61)
    float* const* tables;
    float* const* accums;
    float lr;
    int64_t ld;            // floats between consecutive rows of a table / accumulator (K, or 2K for the packed [w | accum] rows)
    template <int VEC>
    __device__ __forceinline__ void apply(int f, int64_t id, int col, typename BV<VEC>::T g) const {
        using V = BV<VEC>;
        const int64_t off = id * ld + col;
        float* ap = accums[f] + off;
        float* wp = tables[f] + off;
        typename V::T acc = V::ld(ap), wv = V::ld(wp);
        if constexpr (VEC == 4) {
            acc = make_float4(acc.x + g.x * g.x, acc.y + g.y * g.y, acc.z + g.z * g.z, acc.w + g.w * g.w);
            wv = make_float4(wv.x - lr * g.x / sqrtf(acc.x), wv.y - lr * g.y / sqrtf(acc.y), wv.z - lr * g.z / sqrtf(acc.z),
                             wv.w - lr * g.w / sqrtf(acc.w));
        } else {
            acc = acc + g * g;
            wv = wv - lr * g / sqrtf(acc);
        }
        V::st(ap, acc);
        V::st(wp, wv);
    }
};

struct FtrlUpd {      // FTRL-Proximal, learning_rate_power = -0.5 ([TF-upstream] tf.train.FtrlOptimizer, deepFM.py:58)
    float* const* tables;
    float* const* accums;    // n
    float* const* linears;   // z
    float lr, l1, l2;
    int64_t ld;
    bool rows = false;       // tables[f] are packed [vocab, 4] rows [w | n | z | -] (K = 1): one 16-byte read and write per touched id
    __device__ __forceinline__ void one(float g, float& n, float& z, float& w) const {
        const float n_new = n + g * g;
        const float sigma = (sqrtf(n_new) - sqrtf(n)) / lr;
        const float z_new = z + g - sigma * w;
        const float quad = sqrtf(n_new) / lr + 2.f * l2;
        const float sgn = z_new > 0.f ? 1.f : (z_new < 0.f ? -1.f : 0.f);
        w = fabsf(z_new) > l1 ? (sgn * l1 - z_new) / quad : 0.f;
        n = n_new;
        z = z_new;
    }
    template <int VEC>
    __device__ __forceinline__ void apply(int f, int64_t id, int col, typename BV<VEC>::T g) const {
        using V = BV<VEC>;
        if constexpr (VEC == 1) {
            if (rows) {                                    // (uniform)
                float4* rp = reinterpret_cast<float4*>(tables[f] + id * 4);
                float4 r = *rp;
                one(g, r.y, r.z, r.x);
                *rp = r;
                return;
            }
        }
        const int64_t off = id * ld + col;
        float* np_ = accums[f] + off;
        float* zp = linears[f] + off;
        float* wp = tables[f] + off;
        typename V::T n = V::ld(np_), z = V::ld(zp), w = V::ld(wp);
        if constexpr (VEC == 4) {
            one(g.x, n.x, z.x, w.x); one(g.y, n.y, z.y, w.y); one(g.z, n.z, z.z, w.z); one(g.w, n.w, z.w, w.w);
        } else {
            one(g, n, z, w);
        }
        V::st(np_, n);
        V::st(zp, z);
        V::st(wp, w);
    }
};

// tf.train.AdamOptimizer on an IndexedSlices gradient ([TF-upstream] _apply_sparse_shared, as the reference's train_op applies it to
// the embedding tables: DeepCrossNetwork.py:264-290, DeepCrossNetwork/train.py:119-124): m and v of EVERY row decay, the summed
// (clipped) row gradients are added to the touched rows, and every row of var steps by lr_t * m / (sqrt(v) + eps), lr_t = lr *
// sqrt(1 - b2^t) / (1 - b1^t).  Two sorted passes share one sort: AdamNormUpd adds up ||S_r||^2 per table (S_r = a row's summed
// gradient) for tf.clip_by_norm(g, clip) -- in 2^-32 fixed point with integer atomics, so the sum does not depend on arrival order --
// and AdamUpd applies the touched rows and marks them; adam_decay_k then streams over all rows and steps the unmarked ones.
constexpr double ADAM_FX = 4294967296.0;       // 2^32
struct AdamNormUpd {
    static constexpr bool kNorm = true;
    unsigned long long* norm2;                 // [F] fixed point
    template <int VEC>
    __device__ __forceinline__ void apply(int, int64_t, int, typename BV<VEC>::T) const {}
};
struct AdamUpd {
    float* const* tables;
    float* const* ms;
    float* const* vs;
    const unsigned long long* norm2;           // [F] fixed-point ||gradient of table f||^2 (NULL: no clipping)
    unsigned char* mark;                       // [total_rows]: 1 = this row was stepped here
    const int64_t* row_base;
    float lr_t, b1, b2, eps, clip;
    int64_t ld;
    __device__ __forceinline__ void one(float g, float den, float& m, float& v, float& w) const {
        if (clip > 0.f) g = (g * clip) / den;                     // tf.clip_by_norm: t * clip_norm / max(l2norm, clip_norm)
        m = m * b1 + g * (1.f - b1);
        v = v * b2 + (g * g) * (1.f - b2);
        w = w - (lr_t * m) / (sqrtf(v) + eps);
    }
    template <int VEC>
    __device__ __forceinline__ void apply(int f, int64_t id, int col, typename BV<VEC>::T g) const {
        using V = BV<VEC>;
        float den = clip;
        if (clip > 0.f && norm2) den = fmaxf(sqrtf((float)((double)norm2[f] / ADAM_FX)), clip);
        const int64_t off = id * ld + col;
        float* mp = ms[f] + off;
        float* vp = vs[f] + off;
        float* wp = tables[f] + off;
        typename V::T m = V::ld(mp), v = V::ld(vp), w = V::ld(wp);
        if constexpr (VEC == 4) {
            one(g.x, den, m.x, v.x, w.x); one(g.y, den, m.y, v.y, w.y); one(g.z, den, m.z, v.z, w.z); one(g.w, den, m.w, v.w, w.w);
        } else {
            one(g, den, m, v, w);
        }
        V::st(mp, m);
        V::st(vp, v);
        V::st(wp, w);
        if (col == 0) mark[row_base[f] + id] = 1;
    }
};
template <class U, class = void> struct IsNorm { static constexpr bool value = false; };
template <class U> struct IsNorm<U, std::void_t<decltype(U::kNorm)>> { static constexpr bool value = U::kNorm; };

// every row of table f NOT stepped by AdamUpd this step (mark == 0): zero gradient -> m *= b1, v *= b2, var -= lr_t * m / (sqrt(v) + eps);
// marked rows are skipped and their mark cleared.  LPS = K / 4 adjacent lanes own a row (all of them read the mark before lane 0 of
// the group clears it: one wave instruction apart).
template <int LPS>
__global__ __launch_bounds__(256) void adam_decay_k(float* const* __restrict__ tables, float* const* __restrict__ ms, float* const* __restrict__ vs,
                                                    const int64_t* __restrict__ row_base, int64_t total_rows, int F,
                                                    unsigned char* __restrict__ mark, float lr_t, float b1, float b2, float eps) {
    const int f = blockIdx.y;
    const int64_t r0 = row_base[f];
    const int64_t V = (f + 1 < F ? row_base[f + 1] : total_rows) - r0;
    float* __restrict__ w = tables[f];
    float* __restrict__ m = ms[f];
    float* __restrict__ v = vs[f];
    const int c = threadIdx.x & (LPS - 1);
    // A workgroup walks ONE contiguous chunk of the table's rows, non-temporal loads and stores: six streams over 1.66 GB each.  (The
    // grid-stride form of round 3 -- consecutive accesses of a lane 2048 workgroups apart -- ran at 4.3-5.0 TB/s; the same change took the
    // plain copy kernel from 4.4-5.5 to 5.65-6.2 TB/s, NOTES R4.4.)
    constexpr int RPI = 256 / LPS;                                     // rows per workgroup iteration
    const int64_t chunk = ((V + gridDim.x - 1) / gridDim.x + RPI - 1) / RPI * RPI;
    const int64_t rbeg = (int64_t)blockIdx.x * chunk;
    const int64_t rend = rbeg + chunk < V ? rbeg + chunk : V;
    for (int64_t row = rbeg + threadIdx.x / LPS; row < rend; row += RPI) {
        const unsigned char mk = mark[r0 + row];
        if (mk) {
            if (c == 0) mark[r0 + row] = 0;
            continue;
        }
        const int64_t off = row * (LPS * 4) + c * 4;
        typedef float f32x4n __attribute__((ext_vector_type(4)));
        f32x4n mm = __builtin_nontemporal_load(reinterpret_cast<const f32x4n*>(m + off));
        f32x4n vv = __builtin_nontemporal_load(reinterpret_cast<const f32x4n*>(v + off));
        f32x4n ww = __builtin_nontemporal_load(reinterpret_cast<const f32x4n*>(w + off));
        mm = (f32x4n){mm.x * b1, mm.y * b1, mm.z * b1, mm.w * b1};
        vv = (f32x4n){vv.x * b2, vv.y * b2, vv.z * b2, vv.w * b2};
        ww = (f32x4n){ww.x - (lr_t * mm.x) / (sqrtf(vv.x) + eps), ww.y - (lr_t * mm.y) / (sqrtf(vv.y) + eps),
                      ww.z - (lr_t * mm.z) / (sqrtf(vv.z) + eps), ww.w - (lr_t * mm.w) / (sqrtf(vv.w) + eps)};
        __builtin_nontemporal_store(mm, reinterpret_cast<f32x4n*>(m + off));
        __builtin_nontemporal_store(vv, reinterpret_cast<f32x4n*>(v + off));
        __builtin_nontemporal_store(ww, reinterpret_cast<f32x4n*>(w + off));
    }
}

template <int LPS, int VEC, class U, bool FM = false>
__global__ __launch_bounds__(256) void adagrad_tile_k(U upd, int F, int K,
                                                      int64_t n, const uint32_t* __restrict__ keys, const uint32_t* __restrict__ vals,
                                                      const float* __restrict__ grad, int64_t g_ld, int64_t g_fs /* grad stride per slot */,
                                                      const int64_t* __restrict__ row_base, uint32_t total_rows,
                                                      int nt /* > 0: payload mode, the table is found from the key among nt tables */,
                                                      float* __restrict__ carry /* [tiles][2][K] */,
                                                      const float* __restrict__ fm_g = nullptr /* FM: d loss / d fm_logit [B] */,
                                                      const float* __restrict__ fm_sum = nullptr /* FM: S[b] = sum_f e[b,f], [B, K] */,
                                                      int stage_min_dups = 8 /* stage the tile's gradients in LDS when it holds at least this many duplicates */) {
    using V = BV<VEC>;
    using T = typename V::T;
    constexpr int NG = 256 / LPS;
    __shared__ uint32_t skey[ADA_TILE], sval[ADA_TILE];
    __shared__ int rstart[ADA_TILE + 1];
    __shared__ int wcnt[4];
    __shared__ unsigned long long nsq[IsNorm<U>::value ? 64 : 1];     // norm pass: this tile's ||S_r||^2 per table, fixed point
    if constexpr (IsNorm<U>::value) {
        if (threadIdx.x < 64) nsq[threadIdx.x] = 0ull;
    }
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int64_t t = blockIdx.x, e0 = t * ADA_TILE;
    const int ne = (int)((n - e0) < ADA_TILE ? (n - e0) : ADA_TILE);
    const bool in = tid < ne;
    const uint32_t key = in ? keys[e0 + tid] : 0xffffffffu;
    skey[tid] = key;
    sval[tid] = in ? vals[e0 + tid] : 0u;
    const bool start = in && (tid == 0 || key != keys[e0 + tid - 1]);
    const unsigned long long m = __ballot(start);
    if (lane == 0) wcnt[wave] = __popcll(m);
    __syncthreads();
    int woff = 0;
#pragma unroll
    for (int w = 0; w < 4; ++w) woff += (w < wave) ? wcnt[w] : 0;
    const int nruns = wcnt[0] + wcnt[1] + wcnt[2] + wcnt[3];
    if (start) rstart[woff + __popcll(m & ((1ull << lane) - 1ull))] = tid;
    if (tid == 0) rstart[nruns] = ne;
    __syncthreads();
    const bool cont_l = t > 0 && keys[e0 - 1] == skey[0];
    const bool cont_r = e0 + ne < n && keys[e0 + ne] == skey[ne - 1];
    const int g = tid / LPS, c = tid - g * LPS, kv = K / VEC;
    // Phase 1 (rows up to 32 floats): every entry's gradient chunk is formed by its own thread group -- all global loads of the tile
    // in flight at once -- and parked in LDS; phase 2 then walks each run over LDS, in entry order (the same sums bit for bit).
    // (A hot row's run was a serial chain of HBM round trips before.)
    constexpr bool STAGE = LPS * VEC <= 32;
    __shared__ __attribute__((aligned(16))) float sd[STAGE ? ADA_TILE * LPS * VEC : 4];
    auto entry_grad = [&](int i, T wv, bool have_w) {
        const uint32_t ent = sval[i];
        const uint32_t b = ent / (uint32_t)F;
        const int f = (int)(ent - b * (uint32_t)F);
        if constexpr (FM) {
            // the FM backward folded in (fm_bwd_k's arithmetic, so the row sums are the unfused path's bit for bit): entry (b, f)'s
            // gradient is (S[b] - e) * g[b] + grad[b, f], and e IS the entry's table row
            if (!have_w) wv = V::ld(upd.tables[f] + ((int64_t)skey[i] - row_base[f]) * upd.ld + c * VEC);
            T d = V::scale(V::sub(V::ld(fm_sum + (int64_t)b * K + c * VEC), wv), fm_g[b]);
            if (grad) d = V::add(d, V::ld(grad + (int64_t)b * g_ld + (int64_t)f * g_fs + c * VEC));
            return d;
        } else {
            return V::ld(grad + (int64_t)b * g_ld + (int64_t)f * g_fs + c * VEC);
        }
    };
    // a tile of (nearly) distinct rows has nothing serial to hide: it skips the LDS round trip (workgroup-uniform choice)
    const bool staged = STAGE && ne - nruns >= stage_min_dups;
    // Adagrad on a tile of (nearly) distinct rows -- the common tile: uniform ids over 10^6-row tables -- is a random 128-byte
    // read-modify-write stream plus a random gradient row per entry (tools/rmw_probe.hip: 107 us for the BASELINE step's 1.7 M rows; this
    // kernel took 167).  What it lost was a chain of dependent round trips per run -- the table pointer, the row (for the folded FM
    // backward), the gradients, then in apply() the accumulator pointer and the accumulator and weight rows -- times the four runs a
    // thread group walks one after the other.  Here the table pointers and row bases come from LDS, a run's weight and accumulator chunks
    // are loaded together and the NEXT run's before the current one is worked on; the arithmetic is apply()'s, bit for bit.
    constexpr bool ADA = std::is_same<U, AdagradUpd>::value;
    __shared__ float* stab[ADA ? 64 : 1];
    __shared__ float* sacc[ADA ? 64 : 1];
    __shared__ int64_t sbase[ADA ? 64 : 1];
    if constexpr (ADA) {
        if (!staged && nt == 0 && F <= 64) {                // (workgroup-uniform)
            if (tid < F) {
                stab[tid] = upd.tables[tid];
                sacc[tid] = upd.accums[tid];
                sbase[tid] = row_base[tid];
            }
            __syncthreads();
            struct Run { int s, e, f; uint32_t rk; float* wp; float* ap; T wv, av; bool live; };
            auto fetch = [&](int r, Run& q) {
                q.live = false;
                if (r >= nruns) return;
                q.s = rstart[r];
                q.e = rstart[r + 1];
                q.rk = skey[q.s];
                if (q.rk >= total_rows || c >= kv) return;          // pruned ids / idle lanes of a padded group
                q.f = (int)(sval[q.s] % (uint32_t)F);
                const int64_t off = ((int64_t)q.rk - sbase[q.f]) * upd.ld + c * VEC;
                q.wp = stab[q.f] + off;
                q.ap = sacc[q.f] + off;
                q.wv = V::ld(q.wp);
                q.av = V::ld(q.ap);
                q.live = true;
            };
            Run cur, nxt;
            fetch(g, nxt);
            for (int r = g; r < nruns; r += NG) {
                cur = nxt;
                fetch(r + NG, nxt);
                if (!cur.live) continue;
                T sum = V::zero();
                int i = cur.s;
                for (; i + 4 <= cur.e; i += 4) {                    // four entries' loads in flight before their (ordered) adds
                    const T d0 = entry_grad(i, cur.wv, true), d1 = entry_grad(i + 1, cur.wv, true), d2 = entry_grad(i + 2, cur.wv, true),
                            d3 = entry_grad(i + 3, cur.wv, true);
                    sum = V::add(V::add(V::add(V::add(sum, d0), d1), d2), d3);
                }
                for (; i < cur.e; ++i) sum = V::add(sum, entry_grad(i, cur.wv, true));
                const bool open_l = r == 0 && cont_l, open_r = r == nruns - 1 && cont_r;
                if (!open_l && !open_r) {
                    T acc = cur.av, wv = cur.wv;
                    if constexpr (VEC == 4) {
                        acc = make_float4(acc.x + sum.x * sum.x, acc.y + sum.y * sum.y, acc.z + sum.z * sum.z, acc.w + sum.w * sum.w);
                        wv = make_float4(wv.x - upd.lr * sum.x / sqrtf(acc.x), wv.y - upd.lr * sum.y / sqrtf(acc.y),
                                         wv.z - upd.lr * sum.z / sqrtf(acc.z), wv.w - upd.lr * sum.w / sqrtf(acc.w));
                    } else {
                        acc = acc + sum * sum;
                        wv = wv - upd.lr * sum / sqrtf(acc);
                    }
                    V::st(cur.ap, acc);
                    V::st(cur.wp, wv);
                } else {
                    V::st(carry + (t * 2 + (open_l ? 0 : 1)) * K + c * VEC, sum);   // a run open on both sides goes to slot 0
                }
            }
            return;
        }
    }
    if (staged) {
#pragma unroll
        for (int q = 0; q < LPS; ++q) {                     // ADA_TILE / NG = LPS entries per group
            const int i = g + q * NG;
            if (i < ne && c < kv && skey[i] < total_rows) V::st(sd + i * (LPS * VEC) + c * VEC, entry_grad(i, V::zero(), false));
        }
        __syncthreads();
    }
    for (int r = g; r < nruns; r += NG) {
        const int s = rstart[r], e = rstart[r + 1];
        const uint32_t rk = skey[s];
        if (rk >= total_rows || c >= kv) continue;          // pruned ids / idle lanes of a padded group
        T sum = V::zero();
        int f = (int)(sval[s] % (uint32_t)F);
        if (staged) {
            for (int i = s; i < e; ++i) sum = V::add(sum, V::ld(sd + i * (LPS * VEC) + c * VEC));
        } else {
            T wv = V::zero();
            if constexpr (FM) wv = V::ld(upd.tables[f] + ((int64_t)rk - row_base[f]) * upd.ld + c * VEC);
            int i = s;
            for (; i + 4 <= e; i += 4) {                    // four entries' loads in flight before their (ordered) adds
                const T d0 = entry_grad(i, wv, true), d1 = entry_grad(i + 1, wv, true), d2 = entry_grad(i + 2, wv, true),
                        d3 = entry_grad(i + 3, wv, true);
                sum = V::add(V::add(V::add(V::add(sum, d0), d1), d2), d3);
            }
            for (; i < e; ++i) sum = V::add(sum, entry_grad(i, wv, true));
        }
        if (nt > 0) {                                      // payload mode: entries carry no slot; row_base is ascending
            f = 0;
            for (int q = 1; q < nt; ++q) f += (int64_t)rk >= row_base[q] ? 1 : 0;
        }
        const bool open_l = r == 0 && cont_l, open_r = r == nruns - 1 && cont_r;
        if (!open_l && !open_r) {
            const int64_t id = (int64_t)rk - row_base[f];
            if constexpr (IsNorm<U>::value) {
                const double d = (double)V::dot(sum, sum, 0.f) * ADAM_FX;
                atomicAdd(&nsq[f & 63], (unsigned long long)(d < 4.0e18 ? d : 4.0e18));
            } else {
                upd.template apply<VEC>(f, id, c * VEC, sum);
            }
        } else {
            V::st(carry + (t * 2 + (open_l ? 0 : 1)) * K + c * VEC, sum);   // a run open on both sides goes to slot 0
        }
    }
    if constexpr (IsNorm<U>::value) {
        __syncthreads();
        if (threadIdx.x < 64 && nsq[threadIdx.x]) atomicAdd(upd.norm2 + threadIdx.x, nsq[threadIdx.x]);
    }
}

// one thread group per tile: if a run STARTS in this tile and continues to the right, add the partials of the tiles it
// runs through (in tile order) and apply
template <int LPS, int VEC, class U>
__global__ __launch_bounds__(256) void adagrad_fix_k(U upd, int F, int K,
                                                     int64_t n, int64_t ntiles, const uint32_t* __restrict__ keys,
                                                     const uint32_t* __restrict__ vals, const int64_t* __restrict__ row_base,
                                                     uint32_t total_rows, int nt, const float* __restrict__ carry) {
    using V = BV<VEC>;
    using T = typename V::T;
    const int64_t q = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int64_t t = q / LPS;
    const int c = (int)(q - t * LPS), kv = K / VEC;
    if (t >= ntiles || c >= kv) return;
    const int64_t e0 = t * ADA_TILE;
    const int64_t e1 = (e0 + ADA_TILE < n) ? e0 + ADA_TILE : n;      // first entry of the next tile
    if (e1 >= n) return;
    const uint32_t kl = keys[e1 - 1];
    if (kl >= total_rows || keys[e1] != kl) return;                   // not open to the right
    if (t > 0 && keys[e0] == kl && keys[e0 - 1] == kl) return;        // the run started in an earlier tile
    T sum = V::ld(carry + (t * 2 + 1) * K + c * VEC);
    for (int64_t u = t + 1; u < ntiles; ++u) {
        sum = V::add(sum, V::ld(carry + (u * 2) * K + c * VEC));
        const int64_t u1 = (u + 1) * ADA_TILE;
        if (!(u1 < n && keys[u1] == kl)) break;                       // the run ends inside tile u
    }
    const uint32_t ent = vals[e1 - 1];
    int f = (int)(ent % (uint32_t)F);
    if (nt > 0) {
        f = 0;
        for (int q = 1; q < nt; ++q) f += (int64_t)kl >= row_base[q] ? 1 : 0;
    }
    const int64_t id = (int64_t)kl - row_base[f];
    if constexpr (IsNorm<U>::value) {
        const double d = (double)V::dot(sum, sum, 0.f) * ADAM_FX;
        atomicAdd(upd.norm2 + (f & 63), (unsigned long long)(d < 4.0e18 ? d : 4.0e18));
    } else {
        upd.template apply<VEC>(f, id, c * VEC, sum);
    }
}

// The (row, entry) sort: csrc/radix_sort.hip (in-tree, kernels only).  Rounds 1-3 called rocPRIM's onesweep sort here (9 bits per pass
// for the 25-bit keys of the BASELINE shape: 80 us of kernels + 7 hipMemsetAsync = 34 us per sort, and a HIP graph holding those memset
// nodes faults on replay after other eager sorts have run: NOTES R4.3).  DIR_SORT=rocprim keeps that call for A/B runs.
#if defined(DIR_WITH_ROCPRIM_SORT)
using AdaSort9 = rocprim::radix_sort_config<rocprim::default_config, rocprim::default_config,
                                            rocprim::radix_sort_onesweep_config<rocprim::kernel_config<512, 12>, rocprim::kernel_config<1024, 8>, 9,
                                                                                rocprim::block_radix_rank_algorithm::match>, 0>;
static hipError_t rocprim_sort_pairs(void* tmp, size_t& tmp_bytes, const uint32_t* k0, uint32_t* k1, const uint32_t* v0, uint32_t* v1, size_t n,
                                     unsigned bits, hipStream_t st) {
    if (n >= ((size_t)1 << 18) && (bits + 8) / 9 < (bits + 7) / 8)     // (small inputs keep the library's own choice of algorithm)
        return rocprim::radix_sort_pairs<AdaSort9>(tmp, tmp_bytes, k0, k1, v0, v1, n, 0u, bits, st);
    return rocprim::radix_sort_pairs(tmp, tmp_bytes, k0, k1, v0, v1, n, 0u, bits, st);
}
static bool use_rocprim_sort() {
    static const bool v = dev_env("DIR_SORT") && !strcmp(dev_env("DIR_SORT"), "rocprim");
    return v;
}
#else
static bool use_rocprim_sort() { return false; }
#endif

struct AdaSortedPlan { size_t n, ntiles, off_keys[2], off_vals[2], off_carry, off_tmp, tmp_bytes, total; unsigned bits; };
static bool adagrad_sorted_plan(int64_t n, int K, int64_t total_rows, AdaSortedPlan& p, int64_t B = 0, int F = 0 /* ids [B, F]: the slot-major sort */) {
    p.n = (size_t)n;
    p.ntiles = (p.n + ADA_TILE - 1) / ADA_TILE;
    unsigned bits = 1;
    while (bits < 32 && (((uint64_t)1 << bits) <= (uint64_t)total_rows)) ++bits;    // keys go up to total_rows (pruned)
    p.bits = bits;
    size_t off = 0;
    auto take = [&](size_t bytes) { const size_t o = off; off += (bytes + 255) & ~(size_t)255; return o; };
    p.off_keys[0] = take(p.n * 4); p.off_keys[1] = take(p.n * 4);
    p.off_vals[0] = take(p.n * 4); p.off_vals[1] = take(p.n * 4);
    p.off_carry = take(p.ntiles * 2 * (size_t)K * 4);
    size_t tmp = radix_sort_temp_bytes(p.n, bits);
    if (F > 0 && radix_slot_sort_ok(B, F, bits)) {
        const size_t t2 = radix_slot_sort_temp_bytes(B, F, bits);
        if (t2 > tmp) tmp = t2;
    }
#if defined(DIR_WITH_ROCPRIM_SORT)
    size_t tmp_r = 0;
    if (rocprim_sort_pairs(nullptr, tmp_r, nullptr, nullptr, nullptr, nullptr, p.n, bits, (hipStream_t)0) != hipSuccess) return false;
    if (tmp_r > tmp) tmp = tmp_r;
#endif
    p.tmp_bytes = tmp;
    p.off_tmp = take(tmp ? tmp : 256);
    p.total = off;
    return true;
}

// FTRL on a dense variable (the linear model's bias under linear_optimizer='Ftrl', deepFM.py:58,268-275): FtrlUpd::one per element
__global__ __launch_bounds__(256) void ftrl_dense_k(float* __restrict__ w, float* __restrict__ n, float* __restrict__ z, const float* __restrict__ g,
                                                     int64_t count, float lr, float l1, float l2) {
    FtrlUpd u{nullptr, nullptr, nullptr, lr, l1, l2, 0};
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < count; i += (int64_t)gridDim.x * 256) {
        float nn = n[i], zz = z[i], ww = w[i];
        u.one(g[i], nn, zz, ww);
        n[i] = nn;
        z[i] = zz;
        w[i] = ww;
    }
}

// Adagrad on a dense variable (dnn_optimizer='Adagrad', deepFM.py:61, on the hidden layers' kernels and biases): AdagradUpd's rule per element
__global__ __launch_bounds__(256) void adagrad_dense_k(float* __restrict__ w, float* __restrict__ acc, const float* __restrict__ g, int64_t count,
                                                        float lr, float eps) {
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < count; i += (int64_t)gridDim.x * 256) {
        const float gi = g[i];
        const float a = acc[i] + gi * gi;
        acc[i] = a;
        w[i] = w[i] - lr * (gi / (sqrtf(a) + eps));
    }
}

// The same step for up to ADM_MAX variables in ONE launch (a DeepFM tower has 8 dense variables: eight launches of a few microseconds each were
// 0.1 ms of host time per step): pointers and sizes travel in the kernel arguments, blockIdx.y = variable.
constexpr int ADM_MAX = 16;
struct AdagradMultiArgs { float* w[ADM_MAX]; float* acc[ADM_MAX]; const float* g[ADM_MAX]; int64_t n[ADM_MAX]; };
__global__ __launch_bounds__(256) void adagrad_dense_multi_k(AdagradMultiArgs a, float lr, float eps) {
    const int v = blockIdx.y;
    float* __restrict__ w = a.w[v];
    float* __restrict__ acc = a.acc[v];
    const float* __restrict__ g = a.g[v];
    const int64_t count = a.n[v];
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < count; i += (int64_t)gridDim.x * 256) {
        const float gi = g[i];
        const float s = acc[i] + gi * gi;
        acc[i] = s;
        w[i] = w[i] - lr * (gi / (sqrtf(s) + eps));
    }
}

}  // namespace dir

using namespace dir;

extern "C" int dir_adagrad_dense_multi_f32(float* const* w, float* const* accum, const float* const* grad, const int64_t* count, int n_vars,
                                           float lr, float eps, dir_stream_t stream) {
    const char* name = "dir_adagrad_dense_multi_f32";
    DIR_CHECK_ARG(n_vars >= 0 && (n_vars == 0 || (w && accum && grad && count)), "%s: null pointer", name);      // HOST arrays of n_vars entries
    for (int v0 = 0; v0 < n_vars; v0 += ADM_MAX) {
        AdagradMultiArgs a;
        const int nv = n_vars - v0 < ADM_MAX ? n_vars - v0 : ADM_MAX;
        int64_t big = 0;
        for (int v = 0; v < ADM_MAX; ++v) {
            const bool in = v < nv;
            a.w[v] = in ? w[v0 + v] : nullptr;
            a.acc[v] = in ? accum[v0 + v] : nullptr;
            a.g[v] = in ? grad[v0 + v] : nullptr;
            a.n[v] = in ? count[v0 + v] : 0;
            if (in) {
                DIR_CHECK_ARG(a.n[v] >= 0 && (a.n[v] == 0 || (a.w[v] && a.acc[v] && a.g[v])), "%s: variable %d: null pointer or negative count", name, v0 + v);
                if (a.n[v] > big) big = a.n[v];
            }
        }
        if (big == 0) continue;
        hipLaunchKernelGGL(adagrad_dense_multi_k, dim3(grid_for((big + 255) / 256, 4), (unsigned)nv), dim3(256), 0, as_stream(stream), a, lr, eps);
        DIR_CHECK_LAUNCH(name);
    }
    return DIR_OK;
}

extern "C" int64_t dir_sparse_adagrad_sorted_workspace_bytes(int64_t B, int F, int K, int64_t total_rows) {
    if (B <= 0 || F <= 0 || K <= 0 || total_rows <= 0 || total_rows >= 0xffffffffll || B * F >= 0x7fffffffll) return 0;
    AdaSortedPlan p;
    return adagrad_sorted_plan(B * F, K, total_rows, p, B, F) ? (int64_t)p.total : 0;
}

template <class U>
static int sparse_sorted_update(const char* name, U upd, int F, int K, const int64_t* ids, int64_t stride_b, int64_t stride_f,
                                const float* grad, int64_t grad_ld, int64_t grad_fs, int64_t B, const int64_t* row_base,
                                int64_t total_rows, void* workspace, int64_t workspace_bytes, dir_stream_t stream,
                                const int64_t* payload = nullptr /* payload mode: B entries, ids unused, grad is [B, K] */,
                                const float* fm_g = nullptr, const float* fm_sum = nullptr /* fold the FM backward in (AdagradUpd only) */,
                                const void* sorted_from = nullptr /* the workspace of an earlier sorted update of the SAME entries (ids,
                                                                     strides, B, F, row_base, total_rows) on this stream: its sorted
                                                                     (row, entry) pairs are used and the sort is skipped */) {
    const int nt = payload ? F : 0;        // tables to search by key
    if (payload) {                         // entries are a flat list: one "slot" per entry
        ids = payload;
        F = 1;
    }
    DIR_CHECK_ARG(F > 0 && K > 0 && B >= 0, "%s: bad shape", name);
    if (B == 0) return DIR_OK;
    DIR_CHECK_ARG(ids && (grad || fm_g) && row_base && workspace, "%s: null pointer", name);
    if (B * F >= 0x7fffffffll) return fail(DIR_E_UNSUPPORTED, "%s: B*F must fit int32", name);
    if (total_rows <= 0 || total_rows >= 0xffffffffll) return fail(DIR_E_UNSUPPORTED, "%s: total_rows must be in [1, 2^32-1)", name);
    const int64_t n = B * F;
    AdaSortedPlan p;
    if (!adagrad_sorted_plan(n, K, total_rows, p, payload ? 0 : B, payload ? 0 : F)) return fail(DIR_E_HIP, "%s: sort size query failed", name);
    // ids [B, F]: the slot-major sort (csrc/radix_sort.hip: the slot is known from the entry's position, the sort runs on local rows)
    const bool slot_sort = !payload && !sorted_from && !use_rocprim_sort() && radix_slot_sort_ok(B, F, p.bits);
    if ((int64_t)p.total > workspace_bytes || (reinterpret_cast<uintptr_t>(workspace) & 255u))
        return fail(DIR_E_BADARG, "%s: workspace needs %lld bytes, 256-byte aligned", name, (long long)p.total);
    hipStream_t st = as_stream(stream);
    char* ws = static_cast<char*>(workspace);
    uint32_t* k0 = reinterpret_cast<uint32_t*>(ws + p.off_keys[0]);
    uint32_t* k1 = reinterpret_cast<uint32_t*>(ws + p.off_keys[1]);
    uint32_t* v0 = reinterpret_cast<uint32_t*>(ws + p.off_vals[0]);
    uint32_t* v1 = reinterpret_cast<uint32_t*>(ws + p.off_vals[1]);
    float* carry = reinterpret_cast<float*>(ws + p.off_carry);
    if (sorted_from) {                     // the pair arrays sit at offsets that depend on the entry count only
        if (reinterpret_cast<uintptr_t>(sorted_from) & 255u) return fail(DIR_E_BADARG, "%s: sorted_from must be a 256-byte aligned workspace", name);
        char* src = const_cast<char*>(static_cast<const char*>(sorted_from));
        k1 = reinterpret_cast<uint32_t*>(src + p.off_keys[1]);
        v1 = reinterpret_cast<uint32_t*>(src + p.off_vals[1]);
    } else if (slot_sort) {
        if (radix_slot_sort_entries(ws + p.off_tmp, ids, stride_b, stride_f, F, B, row_base, (uint32_t)total_rows, p.bits, k0, k1, v0, v1, st) != hipSuccess)
            return fail(DIR_E_HIP, "%s: radix sort failed", name);
    } else {
        // the key pass writes where the sort wants its input (an even number of digit passes starts from the second pair of buffers)
        const bool second = !use_rocprim_sort() && radix_sort_input_buffer((size_t)n, p.bits) == 1;
        uint32_t* kin = second ? k1 : k0;
        uint32_t* vin = second ? v1 : v0;
        if (payload)
            hipLaunchKernelGGL(adagrad_keys_payload_k, dim3(grid_for((n + 255) / 256)), dim3(256), 0, st, payload, nt, n, row_base,
                               (uint32_t)total_rows, kin, vin);
        else
            hipLaunchKernelGGL(adagrad_keys_k, dim3(grid_for((n + 255) / 256)), dim3(256), 0, st, ids, stride_b, stride_f, F, n, row_base,
                               (uint32_t)total_rows, kin, vin);
    }
    DIR_CHECK_LAUNCH(name);
    if (!sorted_from && !slot_sort) {
#if defined(DIR_WITH_ROCPRIM_SORT)
        if (use_rocprim_sort()) {
            size_t tmp = p.tmp_bytes;
            if (rocprim_sort_pairs(ws + p.off_tmp, tmp, k0, k1, v0, v1, (size_t)n, p.bits, st) != hipSuccess)
                return fail(DIR_E_HIP, "%s: radix sort failed", name);
        } else
#endif
        if (radix_sort_pairs_u32(ws + p.off_tmp, k0, k1, v0, v1, (size_t)n, p.bits, st) != hipSuccess)
            return fail(DIR_E_HIP, "%s: radix sort failed", name);
    }
    const bool vec = (K % 4 == 0) && (grad_ld % 4 == 0) && (grad_fs % 4 == 0) && aligned16(grad) && (!fm_sum || aligned16(fm_sum));
    int lps = 1;
    while (lps < (vec ? K / 4 : K)) lps <<= 1;
    if (lps > 64) return fail(DIR_E_UNSUPPORTED, "%s: K=%d too wide", name, K);
    const int64_t ntiles = (int64_t)p.ntiles;
    static const int stage_min = dev_env_int("DIR_ADA_STAGE_MIN", 8);       // development A/B switch
    dim3 gfix((unsigned)((ntiles * lps + 255) / 256));
#define DIR_CASE(L, V)                                                                                                         \
    do {                                                                                                                       \
        if constexpr (std::is_same<U, AdagradUpd>::value) {                                                                    \
            if (fm_g)                                                                                                          \
                hipLaunchKernelGGL((adagrad_tile_k<L, V, U, true>), dim3((unsigned)ntiles), dim3(256), 0, st, upd, F, K, n, k1, v1, grad, \
                                   grad_ld, grad_fs, row_base, (uint32_t)total_rows, nt, carry, fm_g, fm_sum, stage_min);       \
        }                                                                                                                      \
        if (!fm_g)                                                                                                             \
            hipLaunchKernelGGL((adagrad_tile_k<L, V, U>), dim3((unsigned)ntiles), dim3(256), 0, st, upd, F, K, n, k1, v1, grad, grad_ld, \
                               grad_fs, row_base, (uint32_t)total_rows, nt, carry, nullptr, nullptr, stage_min);                \
        hipLaunchKernelGGL((adagrad_fix_k<L, V, U>), gfix, dim3(256), 0, st, upd, F, K, n, ntiles, k1, v1, row_base,             \
                           (uint32_t)total_rows, nt, carry);                                                                    \
    } while (0)
    if (vec) {
        switch (lps) {
            case 1: DIR_CASE(1, 4); break;
            case 2: DIR_CASE(2, 4); break;
            case 4: DIR_CASE(4, 4); break;
            case 8: DIR_CASE(8, 4); break;
            case 16: DIR_CASE(16, 4); break;
            case 32: DIR_CASE(32, 4); break;
            default: DIR_CASE(64, 4); break;
        }
    } else {
        switch (lps) {
            case 1: DIR_CASE(1, 1); break;
            case 2: DIR_CASE(2, 1); break;
            case 4: DIR_CASE(4, 1); break;
            case 8: DIR_CASE(8, 1); break;
            case 16: DIR_CASE(16, 1); break;
            case 32: DIR_CASE(32, 1); break;
            default: DIR_CASE(64, 1); break;
        }
    }
#undef DIR_CASE
    DIR_CHECK_LAUNCH(name);
    return DIR_OK;
}

extern "C" int dir_sparse_adagrad_sorted_f32(float* const* tables, float* const* accums, int F, int K, const int64_t* ids,
                                             int64_t stride_b, int64_t stride_f, const float* grad, int64_t grad_ld, float lr,
                                             int64_t B, const int64_t* row_base, int64_t total_rows, void* workspace,
                                             int64_t workspace_bytes, dir_stream_t stream) {
    DIR_CHECK_ARG(tables && accums, "dir_sparse_adagrad_sorted_f32: null pointer");
    DIR_CHECK_ARG(grad_ld >= (int64_t)F * K, "dir_sparse_adagrad_sorted_f32: grad_ld");
    return sparse_sorted_update("dir_sparse_adagrad_sorted_f32", AdagradUpd{tables, accums, lr, (int64_t)K}, F, K, ids, stride_b, stride_f, grad,
                                grad_ld, (int64_t)K, B, row_base, total_rows, workspace, workspace_bytes, stream);
}

extern "C" int dir_sparse_adagrad_sorted_rows_f32(float* const* tables, float* const* accums, int64_t row_ld, int F, int K,
                                                  const int64_t* ids, int64_t stride_b, int64_t stride_f, const float* grad,
                                                  int64_t grad_ld, const float* fm_g, const float* fm_sum, float lr, int64_t B,
                                                  const int64_t* row_base, int64_t total_rows, void* workspace,
                                                  int64_t workspace_bytes, dir_stream_t stream) {
    DIR_CHECK_ARG(tables && accums, "dir_sparse_adagrad_sorted_rows_f32: null pointer");
    DIR_CHECK_ARG(row_ld >= K && (!grad || grad_ld >= (int64_t)F * K), "dir_sparse_adagrad_sorted_rows_f32: row_ld / grad_ld");
    DIR_CHECK_ARG((fm_g == nullptr) == (fm_sum == nullptr), "dir_sparse_adagrad_sorted_rows_f32: fm_g and fm_sum go together");
    if (K % 4 == 0 && (row_ld & 3)) return fail(DIR_E_UNSUPPORTED, "dir_sparse_adagrad_sorted_rows_f32: row_ld must be a multiple of 4");
    return sparse_sorted_update("dir_sparse_adagrad_sorted_rows_f32", AdagradUpd{tables, accums, lr, row_ld}, F, K, ids, stride_b, stride_f,
                                grad, grad ? grad_ld : (int64_t)F * K, (int64_t)K, B, row_base, total_rows, workspace, workspace_bytes, stream,
                                nullptr, fm_g, fm_sum);
}

extern "C" int dir_sparse_ftrl_sorted_f32(float* const* tables, float* const* accums, float* const* linears, int F, int K,
                                          const int64_t* ids, int64_t stride_b, int64_t stride_f, const float* grad, int64_t grad_ld,
                                          int64_t grad_slot_stride, float lr, float l1, float l2, int64_t B, const int64_t* row_base,
                                          int64_t total_rows, void* workspace, int64_t workspace_bytes, dir_stream_t stream) {
    DIR_CHECK_ARG(tables && accums && linears, "dir_sparse_ftrl_sorted_f32: null pointer");
    DIR_CHECK_ARG(lr > 0.f && l1 >= 0.f && l2 >= 0.f, "dir_sparse_ftrl_sorted_f32: lr=%g l1=%g l2=%g", lr, l1, l2);
    return sparse_sorted_update("dir_sparse_ftrl_sorted_f32", FtrlUpd{tables, accums, linears, lr, l1, l2, (int64_t)K}, F, K, ids, stride_b,
                                stride_f, grad, grad_ld, grad_slot_stride, B, row_base, total_rows, workspace, workspace_bytes, stream);
}

// The two updates of one training step of a DeepFM -- Adagrad on the embedding tables (deepFM.py:61), FTRL on the linear columns (:58) --
// see the same (row, entry) pairs when both table sets have the same vocabularies: the second one takes the first one's sorted pairs.
extern "C" int dir_sparse_adagrad_sorted_rows_from_f32(float* const* tables, float* const* accums, int64_t row_ld, int F, int K,
                                                       const int64_t* ids, int64_t stride_b, int64_t stride_f, const float* grad,
                                                       int64_t grad_ld, const float* fm_g, const float* fm_sum, float lr, int64_t B,
                                                       const int64_t* row_base, int64_t total_rows, void* workspace,
                                                       int64_t workspace_bytes, const void* sorted_from, dir_stream_t stream) {
    const char* name = "dir_sparse_adagrad_sorted_rows_from_f32";
    DIR_CHECK_ARG(tables && accums && sorted_from, "%s: null pointer", name);
    DIR_CHECK_ARG(row_ld >= K && (!grad || grad_ld >= (int64_t)F * K), "%s: row_ld / grad_ld", name);
    DIR_CHECK_ARG((fm_g == nullptr) == (fm_sum == nullptr), "%s: fm_g and fm_sum go together", name);
    if (K % 4 == 0 && (row_ld & 3)) return fail(DIR_E_UNSUPPORTED, "%s: row_ld must be a multiple of 4", name);
    return sparse_sorted_update(name, AdagradUpd{tables, accums, lr, row_ld}, F, K, ids, stride_b, stride_f, grad,
                                grad ? grad_ld : (int64_t)F * K, (int64_t)K, B, row_base, total_rows, workspace, workspace_bytes, stream, nullptr,
                                fm_g, fm_sum, sorted_from);
}

extern "C" int dir_sparse_ftrl_sorted_from_f32(float* const* tables, float* const* accums, float* const* linears, int F, int K,
                                               const int64_t* ids, int64_t stride_b, int64_t stride_f, const float* grad, int64_t grad_ld,
                                               int64_t grad_slot_stride, float lr, float l1, float l2, int64_t B, const int64_t* row_base,
                                               int64_t total_rows, void* workspace, int64_t workspace_bytes, const void* sorted_from,
                                               dir_stream_t stream) {
    const char* name = "dir_sparse_ftrl_sorted_from_f32";
    DIR_CHECK_ARG(tables && accums && linears && sorted_from, "%s: null pointer", name);
    DIR_CHECK_ARG(lr > 0.f && l1 >= 0.f && l2 >= 0.f, "%s: lr=%g l1=%g l2=%g", name, lr, l1, l2);
    return sparse_sorted_update(name, FtrlUpd{tables, accums, linears, lr, l1, l2, (int64_t)K}, F, K, ids, stride_b, stride_f, grad, grad_ld,
                                grad_slot_stride, B, row_base, total_rows, workspace, workspace_bytes, stream, nullptr, nullptr, nullptr,
                                sorted_from);
}

// FTRL on packed linear training rows: rows[f] is [vocab_f, 4] = [w | n | z | unused], 16-byte aligned (units = 1).  sorted_from (optional):
// as dir_sparse_ftrl_sorted_from_f32.  grad is [B, 1] (grad_slot_stride 0: one d logit for every column of a sample) or [B, F].
extern "C" int dir_sparse_ftrl_rows_sorted_f32(float* const* rows, int F, const int64_t* ids, int64_t stride_b, int64_t stride_f,
                                               const float* grad, int64_t grad_ld, int64_t grad_slot_stride, float lr, float l1, float l2,
                                               int64_t B, const int64_t* row_base, int64_t total_rows, void* workspace,
                                               int64_t workspace_bytes, const void* sorted_from, dir_stream_t stream) {
    const char* name = "dir_sparse_ftrl_rows_sorted_f32";
    DIR_CHECK_ARG(rows, "%s: null pointer", name);
    DIR_CHECK_ARG(lr > 0.f && l1 >= 0.f && l2 >= 0.f, "%s: lr=%g l1=%g l2=%g", name, lr, l1, l2);
    FtrlUpd upd{rows, rows, rows, lr, l1, l2, (int64_t)4};
    upd.rows = true;
    return sparse_sorted_update(name, upd, F, 1, ids, stride_b, stride_f, grad, grad_ld, grad_slot_stride, B, row_base, total_rows, workspace,
                                workspace_bytes, stream, nullptr, nullptr, nullptr, sorted_from);
}

extern "C" int dir_fm_second_order_backward_f32(const float* emb, int64_t emb_ld, const float* g, const float* add_in,
                                                int64_t add_ld, int64_t B, int F, int K, float* demb, int64_t demb_ld,
                                                dir_stream_t stream) {
    DIR_CHECK_ARG(F > 0 && K > 0 && B >= 0 && emb_ld >= (int64_t)F * K && demb_ld >= (int64_t)F * K, "dir_fm_second_order_backward_f32: bad shape");
    if (B == 0) return DIR_OK;
    DIR_CHECK_ARG(emb && g && demb, "dir_fm_second_order_backward_f32: null pointer");
    DIR_CHECK_ARG(!add_in || add_ld >= (int64_t)F * K, "dir_fm_second_order_backward_f32: add_ld");
    const bool vec = (K % 4 == 0) && (emb_ld % 4 == 0) && (demb_ld % 4 == 0) && aligned16(emb) && aligned16(demb) &&
                     (!add_in || ((add_ld % 4 == 0) && aligned16(add_in)));
    int lps = 1;
    while (lps < (vec ? K / 4 : K)) lps <<= 1;
    if (lps > 64) return fail(DIR_E_UNSUPPORTED, "dir_fm_second_order_backward_f32: K=%d too wide", K);
    const int spw = 64 / lps;
    const int64_t waves = (B + spw - 1) / spw;
    dim3 grid(grid_for((waves + 3) / 4));
    hipStream_t st = as_stream(stream);
#define DIR_CASE(L, V) hipLaunchKernelGGL((fm_bwd_k<L, V>), grid, dim3(256), 0, st, emb, emb_ld, g, add_in, add_ld, B, F, K, demb, demb_ld)
    if (vec) {
        switch (lps) {
            case 1: DIR_CASE(1, 4); break;
            case 2: DIR_CASE(2, 4); break;
            case 4: DIR_CASE(4, 4); break;
            case 8: DIR_CASE(8, 4); break;
            case 16: DIR_CASE(16, 4); break;
            case 32: DIR_CASE(32, 4); break;
            default: DIR_CASE(64, 4); break;
        }
    } else {
        switch (lps) {
            case 1: DIR_CASE(1, 1); break;
            case 2: DIR_CASE(2, 1); break;
            case 4: DIR_CASE(4, 1); break;
            case 8: DIR_CASE(8, 1); break;
            case 16: DIR_CASE(16, 1); break;
            case 32: DIR_CASE(32, 1); break;
            default: DIR_CASE(64, 1); break;
        }
    }
#undef DIR_CASE
    DIR_CHECK_LAUNCH("fm_second_order_backward");
    return DIR_OK;
}

// workspace: dir_dcn_cross_backward_workspace_bytes(L, d) device bytes (the per-workgroup partials)
static int cross_bwd_blocks() { return kCUs * 2; }   // measured: 256 and 1024 workgroups are both slower at d = 416

extern "C" int64_t dir_dcn_cross_backward_workspace_bytes(int L, int d) {
    return (int64_t)cross_bwd_blocks() * 2 * L * d * (int64_t)sizeof(float);
}

extern "C" int dir_dcn_cross_backward_f32(const float* x0, int64_t x_ld, const float* w, const float* b, int L,
                                          const float* gout, int64_t g_ld, int64_t B, int d, float* gx0, int64_t gx_ld,
                                          float* gw, float* gb, void* workspace, dir_stream_t stream) {
    DIR_CHECK_ARG(L >= 0 && d > 0 && B >= 0 && x_ld >= d && g_ld >= d && gx_ld >= d, "dir_dcn_cross_backward_f32: bad shape");
    hipStream_t st = as_stream(stream);
    if (L > 0) DIR_CHECK_ARG(gw && gb, "dir_dcn_cross_backward_f32: null pointer");
    if (B == 0) {   // no rows: the gradients are zero (otherwise the reduce kernel writes every element of gw / gb)
        if (L > 0 && (zero_async(gw, sizeof(float) * L * d, st) != hipSuccess || zero_async(gb, sizeof(float) * L * d, st) != hipSuccess))
            return fail(DIR_E_HIP, "dir_dcn_cross_backward_f32: memset failed");
        return DIR_OK;
    }
    DIR_CHECK_ARG(x0 && gout && gx0 && ((w && b && workspace) || L == 0), "dir_dcn_cross_backward_f32: null pointer");
    const bool vec = (d % 4 == 0) && (x_ld % 4 == 0) && (g_ld % 4 == 0) && (gx_ld % 4 == 0) && aligned16(x0) && aligned16(gout) &&
                     aligned16(gx0) && (L == 0 || (aligned16(w) && aligned16(b)));
    const int nchunk = vec ? d / 4 : d;
    const int nv = (nchunk + 63) / 64;
    if (nv > 16) return fail(DIR_E_UNSUPPORTED, "dir_dcn_cross_backward_f32: d=%d too wide", d);
    // waves per workgroup so that the LDS image fits: 2*L*d shared + per wave (L+1)*d + 2*L*d
    int nw = 4;
    auto lds = [&](int waves) { return sizeof(float) * ((size_t)2 * L * d + (size_t)waves * ((size_t)(L + 1) * d + (size_t)2 * L * d + (size_t)((L + 3) & ~3))); };
    while (nw > 1 && lds(nw) > 150 * 1024) nw >>= 1;
    if (lds(nw) > 150 * 1024) return fail(DIR_E_UNSUPPORTED, "dir_dcn_cross_backward_f32: L=%d d=%d needs %zu B of LDS", L, d, lds(nw));
    float* partial = static_cast<float*>(workspace);
    static const int reg_env = getenv("DIR_CROSS_BWD_REG") ? atoi(getenv("DIR_CROSS_BWD_REG")) : 1;   // 0: slab kernel (A/B runs)
    if (reg_env && L >= 1 && L <= 4 && nv <= (vec ? 4 : 8)) {
        const int nblk = (int)((B + 3) / 4 < cross_bwd_blocks() ? (B + 3) / 4 : cross_bwd_blocks());
        const size_t sh = sizeof(float) * (size_t)4 * 2 * L * d;
#define DIR_REG(NV, V, LT)                                                                                                      \
    do {                                                                                                                        \
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&cross_bwd_reg_k<NV, V, LT>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); \
        hipLaunchKernelGGL((cross_bwd_reg_k<NV, V, LT>), dim3(nblk), dim3(256), sh, st, x0, x_ld, w, b, gout, g_ld, B, d, gx0, gx_ld, partial); \
    } while (0)
#define DIR_REG_L(NV, V)                                                                                                        \
    do {                                                                                                                        \
        if (L == 1) DIR_REG(NV, V, 1); else if (L == 2) DIR_REG(NV, V, 2); else if (L == 3) DIR_REG(NV, V, 3); else DIR_REG(NV, V, 4); \
    } while (0)
        if (vec) {
            if (nv <= 1) DIR_REG_L(1, 4); else if (nv <= 2) DIR_REG_L(2, 4); else DIR_REG_L(4, 4);
        } else {
            if (nv <= 1) DIR_REG_L(1, 1); else if (nv <= 2) DIR_REG_L(2, 1); else if (nv <= 4) DIR_REG_L(4, 1); else DIR_REG_L(8, 1);
        }
#undef DIR_REG_L
#undef DIR_REG
        DIR_CHECK_LAUNCH("dcn_cross_backward(reg)");
        const int n = 2 * L * d;
        hipLaunchKernelGGL(reduce_partials_k, dim3((unsigned)((n + 3) / 4)), dim3(256), 0, st, partial, nblk, n, gw, gb, L * d);
        DIR_CHECK_LAUNCH("dcn_cross_backward(reduce)");
        return DIR_OK;
    }
    const size_t shmem = lds(nw);
    const int nblk = (int)((B + nw - 1) / nw < cross_bwd_blocks() ? (B + nw - 1) / nw : cross_bwd_blocks());
#define DIR_GO(NV, V)                                                                                                  \
    do {                                                                                                               \
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&cross_bwd_k<NV, V>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); \
        hipLaunchKernelGGL((cross_bwd_k<NV, V>), dim3(nblk), dim3(64 * nw), shmem, st, x0, x_ld, w, b, L, gout, g_ld, B, d, gx0, gx_ld, partial); \
    } while (0)
    if (vec) {
        if (nv <= 1) DIR_GO(1, 4); else if (nv <= 2) DIR_GO(2, 4); else if (nv <= 4) DIR_GO(4, 4); else if (nv <= 8) DIR_GO(8, 4); else DIR_GO(16, 4);
    } else {
        if (nv <= 1) DIR_GO(1, 1); else if (nv <= 2) DIR_GO(2, 1); else if (nv <= 4) DIR_GO(4, 1); else if (nv <= 8) DIR_GO(8, 1); else DIR_GO(16, 1);
    }
#undef DIR_GO
    DIR_CHECK_LAUNCH("dcn_cross_backward");
    if (L > 0) {
        const int n = 2 * L * d;
        hipLaunchKernelGGL(reduce_partials_k, dim3((unsigned)((n + 3) / 4)), dim3(256), 0, st, partial, nblk, n, gw, gb, L * d);
        DIR_CHECK_LAUNCH("dcn_cross_backward(reduce)");
    }
    return DIR_OK;
}

extern "C" int dir_sparse_adagrad_f32(float* const* tables, float* const* accums, int F, int K, const int64_t* ids,
                                      int64_t stride_b, int64_t stride_f, const float* grad, int64_t grad_ld, float lr,
                                      int64_t B, const int64_t* head_base, int64_t total_rows, int32_t* head, int32_t* next,
                                      dir_stream_t stream) {
    DIR_CHECK_ARG(F > 0 && K > 0 && B >= 0 && grad_ld >= (int64_t)F * K, "dir_sparse_adagrad_f32: bad shape");
    if (B == 0) return DIR_OK;
    DIR_CHECK_ARG(tables && accums && ids && grad && head_base && head && next, "dir_sparse_adagrad_f32: null pointer");
    DIR_CHECK_ARG(total_rows > 0, "dir_sparse_adagrad_f32: total_rows=%lld", (long long)total_rows);
    if (B * F >= (int64_t)0x7fffffff) return fail(DIR_E_UNSUPPORTED, "dir_sparse_adagrad_f32: B*F must fit int32");
    const int64_t n = B * F;
    hipStream_t st = as_stream(stream);
    hipLaunchKernelGGL(adagrad_link_k, dim3(grid_for((n + 255) / 256)), dim3(256), 0, st, ids, stride_b, stride_f, F, n, head_base,
                       total_rows, head, next);
    const bool vec = (K % 4 == 0) && (grad_ld % 4 == 0) && aligned16(grad);
    int lps = 1;
    while (lps < (vec ? K / 4 : K)) lps <<= 1;
    if (lps > 64) return fail(DIR_E_UNSUPPORTED, "dir_sparse_adagrad_f32: K=%d too wide", K);
    dim3 grid(grid_for((n * lps + 255) / 256));
#define DIR_CASE(L, V) hipLaunchKernelGGL((adagrad_apply_k<L, V>), grid, dim3(256), 0, st, tables, accums, ids, stride_b, stride_f, F, K, n, grad, grad_ld, lr, head_base, total_rows, head, next)
    if (vec) {
        switch (lps) {
            case 1: DIR_CASE(1, 4); break;
            case 2: DIR_CASE(2, 4); break;
            case 4: DIR_CASE(4, 4); break;
            case 8: DIR_CASE(8, 4); break;
            case 16: DIR_CASE(16, 4); break;
            case 32: DIR_CASE(32, 4); break;
            default: DIR_CASE(64, 4); break;
        }
    } else {
        switch (lps) {
            case 1: DIR_CASE(1, 1); break;
            case 2: DIR_CASE(2, 1); break;
            case 4: DIR_CASE(4, 1); break;
            case 8: DIR_CASE(8, 1); break;
            case 16: DIR_CASE(16, 1); break;
            case 32: DIR_CASE(32, 1); break;
            default: DIR_CASE(64, 1); break;
        }
    }
#undef DIR_CASE
    DIR_CHECK_LAUNCH("sparse_adagrad");
    return DIR_OK;
}

extern "C" int dir_sparse_adagrad_sorted_payload_f32(float* const* tables, float* const* accums, int F, int K, const int64_t* payload,
                                                     int64_t n, const float* grad, float lr, const int64_t* row_base,
                                                     int64_t total_rows, void* workspace, int64_t workspace_bytes,
                                                     dir_stream_t stream) {
    DIR_CHECK_ARG(tables && accums && (payload || n == 0), "dir_sparse_adagrad_sorted_payload_f32: null pointer");
    return sparse_sorted_update("dir_sparse_adagrad_sorted_payload_f32", AdagradUpd{tables, accums, lr, (int64_t)K}, F, K, nullptr, 0, 0, grad,
                                (int64_t)K, 0, n, row_base, total_rows, workspace, workspace_bytes, stream, payload);
}

// ---- tf.train.AdamOptimizer on the embedding tables ------------------------------------------------------------------------------
extern "C" int64_t dir_sparse_adam_workspace_bytes(int64_t B, int F, int K, int64_t total_rows) {
    if (B < 0 || F <= 0 || F > 64 || K <= 0 || (K & 3) || 64 % (K / 4) || total_rows <= 0 || total_rows >= 0xffffffffll || B * F >= 0x7fffffffll) return 0;
    AdaSortedPlan p;
    const int64_t sorted = B > 0 ? (adagrad_sorted_plan(B * F, K, total_rows, p, B, F) ? (int64_t)p.total : -1) : 0;
    if (sorted < 0) return 0;
    return sorted + 512 + ((total_rows + 255) & ~(int64_t)255);      // + norm2 [64] u64 + the row marks (one byte per row)
}

extern "C" int dir_sparse_adam_f32(float* const* tables, float* const* ms, float* const* vs, int F, int K, const int64_t* ids,
                                   int64_t stride_b, int64_t stride_f, const float* grad, int64_t grad_ld, float lr_t, float beta1,
                                   float beta2, float eps, float clip_norm, int64_t B, const int64_t* row_base, int64_t total_rows,
                                   void* workspace, int64_t workspace_bytes, int first_call, dir_stream_t stream) {
    const char* name = "dir_sparse_adam_f32";
    DIR_CHECK_ARG(tables && ms && vs && row_base && workspace && F > 0 && K > 0 && B >= 0, "%s: bad argument", name);
    if (F > 64 || (K & 3) || 64 % (K / 4) || K / 4 > 64) return fail(DIR_E_UNSUPPORTED, "%s: F <= 64, K a multiple of 4 with K / 4 dividing 64 (F=%d K=%d)", name, F, K);
    DIR_CHECK_ARG(B == 0 || (ids && grad && grad_ld >= (int64_t)F * K), "%s: ids / grad", name);
    DIR_CHECK_ARG(lr_t >= 0.f && beta1 >= 0.f && beta1 < 1.f && beta2 >= 0.f && beta2 < 1.f && eps > 0.f && clip_norm >= 0.f, "%s: hyper-parameters", name);
    const int64_t need = dir_sparse_adam_workspace_bytes(B, F, K, total_rows);
    if (need <= 0 || need > workspace_bytes || (reinterpret_cast<uintptr_t>(workspace) & 255u))
        return fail(DIR_E_BADARG, "%s: workspace needs %lld bytes, 256-byte aligned", name, (long long)need);
    hipStream_t st = as_stream(stream);
    char* ws = static_cast<char*>(workspace);
    // layout: [norm2: 64 x u64 = 512 B][marks: total_rows bytes, zero between calls][sorted-update workspace]
    unsigned long long* norm2 = reinterpret_cast<unsigned long long*>(ws);
    unsigned char* mark = reinterpret_cast<unsigned char*>(ws + 512);
    const int64_t mark_bytes = (total_rows + 255) & ~(int64_t)255;
    char* sorted_ws = ws + 512 + mark_bytes;
    const int64_t sorted_bytes = workspace_bytes - 512 - mark_bytes;
    if (first_call && zero_async(mark, (size_t)mark_bytes, st) != hipSuccess) return fail(DIR_E_HIP, "%s: memset failed", name);
    if (B > 0) {
        const bool clip = clip_norm > 0.f;
        if (clip) {
            if (zero_async(norm2, 512, st) != hipSuccess) return fail(DIR_E_HIP, "%s: memset failed", name);
            const int rc = sparse_sorted_update(name, AdamNormUpd{norm2}, F, K, ids, stride_b, stride_f, grad, grad_ld, (int64_t)K, B, row_base,
                                                total_rows, sorted_ws, sorted_bytes, stream);
            if (rc != DIR_OK) return rc;
        }
        const int rc = sparse_sorted_update(name, AdamUpd{tables, ms, vs, clip ? norm2 : nullptr, mark, row_base, lr_t, beta1, beta2, eps, clip_norm, (int64_t)K},
                                            F, K, ids, stride_b, stride_f, grad, grad_ld, (int64_t)K, B, row_base, total_rows, sorted_ws, sorted_bytes,
                                            stream, nullptr, nullptr, nullptr, clip ? sorted_ws : nullptr);
        if (rc != DIR_OK) return rc;
    }
    const int lps = K / 4;
    dim3 grid((unsigned)(kCUs * 8), (unsigned)F);
    switch (lps) {
#define DIR_DECAY(L) case L: hipLaunchKernelGGL((adam_decay_k<L>), grid, dim3(256), 0, st, tables, ms, vs, row_base, total_rows, F, mark, lr_t, beta1, beta2, eps); break
        DIR_DECAY(1); DIR_DECAY(2); DIR_DECAY(4); DIR_DECAY(8); DIR_DECAY(16); DIR_DECAY(32); DIR_DECAY(64);
#undef DIR_DECAY
        default: return fail(DIR_E_UNSUPPORTED, "%s: K=%d", name, K);
    }
    DIR_CHECK_LAUNCH(name);
    return DIR_OK;
}

extern "C" int dir_ftrl_dense_f32(float* w, float* accum, float* linear, const float* grad, int64_t count, float lr, float l1, float l2,
                                  dir_stream_t stream) {
    const char* name = "dir_ftrl_dense_f32";
    DIR_CHECK_ARG(count >= 0 && lr > 0.f, "%s: count=%lld lr=%g", name, (long long)count, (double)lr);
    if (count == 0) return DIR_OK;
    DIR_CHECK_ARG(w && accum && linear && grad, "%s: null pointer", name);
    hipLaunchKernelGGL(ftrl_dense_k, dim3(grid_for((count + 255) / 256)), dim3(256), 0, as_stream(stream), w, accum, linear, grad, count, lr, l1, l2);
    DIR_CHECK_LAUNCH(name);
    return DIR_OK;
}

extern "C" int dir_adagrad_dense_f32(float* w, float* accum, const float* grad, int64_t count, float lr, float eps, dir_stream_t stream) {
    const char* name = "dir_adagrad_dense_f32";
    DIR_CHECK_ARG(count >= 0, "%s: count=%lld", name, (long long)count);
    if (count == 0) return DIR_OK;
    DIR_CHECK_ARG(w && accum && grad, "%s: null pointer", name);
    hipLaunchKernelGGL(adagrad_dense_k, dim3(grid_for((count + 255) / 256)), dim3(256), 0, as_stream(stream), w, accum, grad, count, lr, eps);
    DIR_CHECK_LAUNCH(name);
    return DIR_OK;
}
