// linear_cross.hip -- first-order sparse term and DCN cross network for gfx950.
//
// Replaces (reference, /root/reference):
//   _linear_logit_fn_builder / linear_model(sparse_combiner=)   models/DeepFM/deepFM.py:255-275
//   _cross_op / _cross_architecture               models/DeepCrossNetwork/DeepCrossNetwork.py:336-367
//
// Both are HBM/L2-bound with no reuse across samples:
//  * linear: one lane per sample walks the F slots; each slot is one 4-byte read of a [vocab] weight
//    column.  UF reads are issued before the first add; the adds run in slot order (bit-exact vs
//    oracle).
//  * cross: G lanes own one sample; x0 and x_l stay in registers for all L layers (one read of x0, one
//    write of x_L per sample); w_l / b_l are staged once per block in LDS; the dot x_l . w_l is a
//    G-lane butterfly.
#include "common.hpp"

namespace dir {

// ------------------------------------------------------------------------------------------------
// linear term
// ------------------------------------------------------------------------------------------------
template <int UF>
__global__ __launch_bounds__(256) void linear_onehot_k(const float* const* __restrict__ wts,
                                                       const int64_t* __restrict__ vocab,
                                                       const int64_t* __restrict__ ids, int64_t sb, int64_t sf,
                                                       int F, const float* __restrict__ bias, int accumulate,
                                                       int64_t B, float* __restrict__ out, int64_t ld /* floats between two weights: 1, or
                                                       the stride of packed linear training rows [w | n | z | -] */) {
    for (int64_t b = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; b < B; b += (int64_t)gridDim.x * blockDim.x) {
        const int64_t* idp = ids + b * sb;
        float acc = 0.f;
        for (int f0 = 0; f0 < F; f0 += UF) {
            int64_t id[UF];
            float v[UF];
#pragma unroll
            for (int u = 0; u < UF; ++u) id[u] = (f0 + u < F) ? idp[(int64_t)(f0 + u) * sf] : (int64_t)-1;
#pragma unroll
            for (int u = 0; u < UF; ++u) {
                v[u] = 0.f;
                if (f0 + u < F) {
                    const float* w = wts[f0 + u];
                    const uint64_t bound = vocab ? (uint64_t)vocab[f0 + u] : (uint64_t)1 << 63;   // id < 0 / id >= vocab_f: pruned
                    if ((uint64_t)id[u] < bound) v[u] = w[id[u] * ld];
                }
            }
#pragma unroll
            for (int u = 0; u < UF; ++u)
                if (f0 + u < F) acc = acc + v[u];
        }
        float r = acc + (bias ? bias[0] : 0.f);
        out[b] = accumulate ? out[b] + r : r;
    }
}

// The same sum with FOUR lanes per sample (round 5): one thread per sample left 4 waves per CU to cover 26 dependent-free but
// latency-long reads each (33.7 us for 1.7 M reads, 0.077 of HBM by bytes); lane c of a sample's quad reads the weights of fields
// 4 j + c, and the quad adds them up in FIELD order through quad broadcasts (DPP) -- the oracle's sequential fp32 sum, bit for bit.
template <int UFL>
__global__ __launch_bounds__(256) void linear_onehot4_k(const float* const* __restrict__ wts, const int64_t* __restrict__ vocab,
                                                        const int64_t* __restrict__ ids, int64_t sb, int64_t sf, int F,
                                                        const float* __restrict__ bias, int accumulate, int64_t B, float* __restrict__ out,
                                                        int64_t ld) {
    const int c = threadIdx.x & 3;
    const int64_t per_grid = ((int64_t)gridDim.x * blockDim.x) >> 2;
    for (int64_t b0 = ((int64_t)blockIdx.x * blockDim.x) >> 2; b0 < B; b0 += per_grid) {      // (block-uniform trip count: the DPP reads see live lanes)
        const int64_t b = b0 + (threadIdx.x >> 2);
        const bool live = b < B;
        const int64_t* idp = ids + (live ? b : 0) * sb;
        float acc = 0.f;
        for (int f0 = 0; f0 < F; f0 += 4 * UFL) {
            int64_t id[UFL];
            float v[UFL];
#pragma unroll
            for (int j = 0; j < UFL; ++j) {
                const int f = f0 + 4 * j + c;
                id[j] = (live && f < F) ? idp[(int64_t)f * sf] : (int64_t)-1;
            }
#pragma unroll
            for (int j = 0; j < UFL; ++j) {
                const int f = f0 + 4 * j + c;
                v[j] = 0.f;
                if (f < F) {
                    const uint64_t bound = vocab ? (uint64_t)vocab[f] : (uint64_t)1 << 63;   // id < 0 / id >= vocab_f: pruned
                    if ((uint64_t)id[j] < bound) v[j] = wts[f][id[j] * ld];
                }
            }
#pragma unroll
            for (int j = 0; j < UFL; ++j) {
                const int vi = __builtin_bit_cast(int, v[j]);
                const float q0 = __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(vi, 0x00, 0xf, 0xf, true));   // quad_perm [0,0,0,0]
                const float q1 = __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(vi, 0x55, 0xf, 0xf, true));   // [1,1,1,1]
                const float q2 = __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(vi, 0xaa, 0xf, 0xf, true));   // [2,2,2,2]
                const float q3 = __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(vi, 0xff, 0xf, 0xf, true));   // [3,3,3,3]
                const int f = f0 + 4 * j;
                if (f < F) acc = acc + q0;
                if (f + 1 < F) acc = acc + q1;
                if (f + 2 < F) acc = acc + q2;
                if (f + 3 < F) acc = acc + q3;
            }
        }
        if (live && c == 0) {
            const float r = acc + (bias ? bias[0] : 0.f);
            out[b] = accumulate ? out[b] + r : r;
        }
    }
}

__global__ __launch_bounds__(256) void linear_csr_k(const float* const* __restrict__ wts,
                                                    const int64_t* __restrict__ vocab,
                                                    const int64_t* __restrict__ ids,
                                                    const int64_t* __restrict__ offsets,
                                                    const float* __restrict__ ew, int64_t sb, int64_t sf, int F,
                                                    int combiner, const float* __restrict__ bias, int accumulate,
                                                    int64_t B, float* __restrict__ out) {
    for (int64_t b = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; b < B; b += (int64_t)gridDim.x * blockDim.x) {
        float acc = 0.f;
        for (int f = 0; f < F; ++f) {
            const float* w = wts[f];
            const uint64_t bound = vocab ? (uint64_t)vocab[f] : (uint64_t)1 << 63;
            const int64_t bag = b * sb + (int64_t)f * sf;
            const int64_t beg = offsets[bag], end = offsets[bag + 1];
            float v = 0.f, wsum = 0.f, w2sum = 0.f;
            int cnt = 0;
            for (int64_t e = beg; e < end; ++e) {
                const int64_t id = ids[e];
                if (!((uint64_t)id < bound)) continue;
                const float wt = ew ? ew[e] : 1.0f;
                v = ew ? v + wt * w[id] : v + w[id];
                wsum = wsum + wt;
                w2sum = w2sum + wt * wt;
                ++cnt;
            }
            if (cnt > 0 && combiner == DIR_COMBINER_MEAN) v = v / (ew ? wsum : (float)cnt);
            if (cnt > 0 && combiner == DIR_COMBINER_SQRTN) v = v / (ew ? sqrtf(w2sum) : sqrtf((float)cnt));
            acc = acc + v;
        }
        float r = acc + (bias ? bias[0] : 0.f);
        out[b] = accumulate ? out[b] + r : r;
    }
}

// ------------------------------------------------------------------------------------------------
// cross network.  G lanes per sample, NV chunks of VEC floats per lane, all L layers in registers.
// LDS image: w then b, [L][d] each, as given.
// ------------------------------------------------------------------------------------------------
template <int VEC> struct CV;
template <> struct CV<4> {
    using T = float4;
    static __device__ __forceinline__ T ld(const float* p) { return *reinterpret_cast<const float4*>(p); }
    static __device__ __forceinline__ void st(float* p, T v) { *reinterpret_cast<float4*>(p) = v; }
    static __device__ __forceinline__ T zero() { return make_float4(0.f, 0.f, 0.f, 0.f); }
    static __device__ __forceinline__ float dot(T a, T b, float acc) {
        acc = fmaf(a.x, b.x, acc); acc = fmaf(a.y, b.y, acc); acc = fmaf(a.z, b.z, acc); acc = fmaf(a.w, b.w, acc);
        return acc;
    }
    // ((x0 * xw) + b) + xl, elementwise, unfused (DeepCrossNetwork.py:346 evaluation order)
    static __device__ __forceinline__ T upd(T x0, float xw, T b, T xl) {
        return make_float4(((x0.x * xw) + b.x) + xl.x, ((x0.y * xw) + b.y) + xl.y,
                           ((x0.z * xw) + b.z) + xl.z, ((x0.w * xw) + b.w) + xl.w);
    }
};
template <> struct CV<1> {
    using T = float;
    static __device__ __forceinline__ T ld(const float* p) { return *p; }
    static __device__ __forceinline__ void st(float* p, T v) { *p = v; }
    static __device__ __forceinline__ T zero() { return 0.f; }
    static __device__ __forceinline__ float dot(T a, T b, float acc) { return fmaf(a, b, acc); }
    static __device__ __forceinline__ T upd(T x0, float xw, T b, T xl) { return ((x0 * xw) + b) + xl; }
};

template <int G, int NV, int VEC>
__global__ __launch_bounds__(256) void cross_k(const float* __restrict__ x0, int64_t x_ld,
                                               const float* __restrict__ xinit /* nullptr: x_0 = x0 */,
                                               const float* __restrict__ w, const float* __restrict__ bvec,
                                               int L, int64_t B, int d, float* __restrict__ out, int64_t out_ld,
                                               const float* __restrict__ head_w /* [d] or nullptr */, float* __restrict__ head_out /* [B] */) {
    using C = CV<VEC>;
    using V = typename C::T;
    extern __shared__ __attribute__((aligned(16))) float smem[];  // [2][L][d] (+ [d]: the head's weights)
    constexpr int SPW = 64 / G;
    const int nchunk = d / VEC;
    const int Ld = L * d;
    if (VEC == 4) {  // d % 4 == 0: w, b and the LDS image are 16-byte aligned
        const int n4 = Ld >> 2;
        const float4* w4 = reinterpret_cast<const float4*>(w);
        const float4* b4 = reinterpret_cast<const float4*>(bvec);
        float4* s4 = reinterpret_cast<float4*>(smem);
        for (int i = threadIdx.x; i < n4; i += blockDim.x) {
            s4[i] = w4[i];
            s4[n4 + i] = b4[i];
        }
    } else {
        for (int i = threadIdx.x; i < Ld; i += blockDim.x) {
            smem[i] = w[i];
            smem[Ld + i] = bvec[i];
        }
    }
    if (head_w)
        for (int i = threadIdx.x; i < d; i += blockDim.x) smem[2 * Ld + i] = head_w[i];
    __syncthreads();
    const int lane = threadIdx.x & 63;
    const int j = lane & (G - 1);
    const int s = lane / G;
    const int64_t nwave = (int64_t)gridDim.x * (blockDim.x >> 6);
    for (int64_t g = (int64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6); g * SPW < B; g += nwave) {
        const int64_t r = g * SPW + s;
        const bool act = r < B;
        const float* xp = x0 + (act ? r * x_ld : 0);
        const float* xi = xinit ? xinit + (act ? r * x_ld : 0) : nullptr;
        V xv[NV], xl[NV];
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const int ch = i * G + j;
            xv[i] = (act && ch < nchunk) ? C::ld(xp + ch * VEC) : C::zero();
            xl[i] = xi ? ((act && ch < nchunk) ? C::ld(xi + ch * VEC) : C::zero()) : xv[i];
        }
        for (int l = 0; l < L; ++l) {
            const float* wl = smem + l * d;
            const float* bl = smem + Ld + l * d;
            float part = 0.f;
#pragma unroll
            for (int i = 0; i < NV; ++i) {
                const int ch = i * G + j;
                if (ch < nchunk) part = C::dot(xl[i], C::ld(wl + ch * VEC), part);
            }
            const float xw = group_sum<G>(part);
#pragma unroll
            for (int i = 0; i < NV; ++i) {
                const int ch = i * G + j;
                if (ch < nchunk) xl[i] = C::upd(xv[i], xw, C::ld(bl + ch * VEC), xl[i]);
            }
        }
        if (head_w) {     // the cross branch's share of the final dense(1) over concat([cross, deep]) (DeepCrossNetwork.py:136-137): x_L . w_c
            const float* hw = smem + 2 * Ld;
            float part = 0.f;
#pragma unroll
            for (int i = 0; i < NV; ++i) {
                const int ch = i * G + j;
                if (ch < nchunk) part = C::dot(xl[i], C::ld(hw + ch * VEC), part);
            }
            const float hs = group_sum<G>(part);
            if (act && j == 0) head_out[r] = hs;
        }
        if (act && out) {
            float* op = out + r * out_ld;
#pragma unroll
            for (int i = 0; i < NV; ++i) {
                const int ch = i * G + j;
                if (ch < nchunk) C::st(op + ch * VEC, xl[i]);
            }
        }
    }
}

static bool linear_quads() {          // development A/B switch (DIR_DEVELOPMENT builds only): DIR_LINEAR_QUADS=0 = one thread per sample
    static const int v = dev_env_int("DIR_LINEAR_QUADS", 1);
    return v != 0;
}

}  // namespace dir

using namespace dir;

extern "C" int dir_linear_sparse_sum_f32(const float* const* weights, const int64_t* vocab, int F, const int64_t* ids,
                                         const int64_t* offsets, const float* entry_weights, int64_t stride_b,
                                         int64_t stride_f, int combiner, const float* bias, int accumulate,
                                         int64_t B, float* out, dir_stream_t stream) {
    DIR_CHECK_ARG(weights && ids && out, "dir_linear_sparse_sum_f32: null pointer");
    DIR_CHECK_ARG(F > 0 && B >= 0, "dir_linear_sparse_sum_f32: F=%d B=%lld", F, (long long)B);
    DIR_CHECK_ARG(combiner >= DIR_COMBINER_SUM && combiner <= DIR_COMBINER_SQRTN, "dir_linear_sparse_sum_f32: combiner=%d", combiner);
    DIR_CHECK_ARG(offsets || !entry_weights, "dir_linear_sparse_sum_f32: entry weights need offsets");
    if (B == 0) return DIR_OK;
    hipStream_t st = as_stream(stream);
    dim3 grid(grid_for((B + 255) / 256));
    if (!offsets) {
        if (linear_quads())
            hipLaunchKernelGGL((linear_onehot4_k<7>), dim3(grid_for((B + 63) / 64)), dim3(256), 0, st, weights, vocab, ids, stride_b, stride_f, F, bias,
                               accumulate, B, out, (int64_t)1);
        else
        hipLaunchKernelGGL((linear_onehot_k<13>), grid, dim3(256), 0, st, weights, vocab, ids, stride_b, stride_f, F, bias, accumulate, B, out, (int64_t)1);
    } else {
        hipLaunchKernelGGL(linear_csr_k, grid, dim3(256), 0, st, weights, vocab, ids, offsets, entry_weights, stride_b, stride_f, F, combiner, bias, accumulate, B, out);
    }
    DIR_CHECK_LAUNCH("linear_sparse_sum");
    return DIR_OK;
}

// The first-order term read from packed linear TRAINING rows (dir_sparse_ftrl_rows_sorted_f32): rows[f] is [vocab_f, row_ld] with the weight
// in column 0 -- the row's FTRL state (n, z) shares its 16 bytes, so the update moves one memory slot per touched id instead of three.
extern "C" int dir_linear_onehot_rows_f32(const float* const* rows, int64_t row_ld, const int64_t* vocab, int F, const int64_t* ids,
                                          int64_t stride_b, int64_t stride_f, const float* bias, int accumulate, int64_t B, float* out,
                                          dir_stream_t stream) {
    DIR_CHECK_ARG(rows && ids && out, "dir_linear_onehot_rows_f32: null pointer");
    DIR_CHECK_ARG(F > 0 && B >= 0 && row_ld >= 1, "dir_linear_onehot_rows_f32: F=%d B=%lld row_ld=%lld", F, (long long)B, (long long)row_ld);
    if (B == 0) return DIR_OK;
    hipStream_t st = as_stream(stream);
    dim3 grid(grid_for((B + 255) / 256));
    if (linear_quads())
        hipLaunchKernelGGL((linear_onehot4_k<7>), dim3(grid_for((B + 63) / 64)), dim3(256), 0, st, rows, vocab, ids, stride_b, stride_f, F, bias, accumulate,
                           B, out, row_ld);
    else
    hipLaunchKernelGGL((linear_onehot_k<13>), grid, dim3(256), 0, st, rows, vocab, ids, stride_b, stride_f, F, bias, accumulate, B, out, row_ld);
    DIR_CHECK_LAUNCH("linear_onehot_rows");
    return DIR_OK;
}

template <int G, int VEC>
static int launch_cross_nv(int nv, int64_t work_blocks, size_t shmem, hipStream_t st, const float* x0, int64_t x_ld,
                           const float* xinit, const float* w, const float* b, int L, int64_t B, int d, float* out, int64_t out_ld,
                           const float* head_w, float* head_out) {
#define DIR_GO(NV)                                                                                          \
    do {                                                                                                    \
        dim3 grid(grid_resident(work_blocks, resident_blocks(cross_k<G, NV, VEC>, shmem)));                 \
        hipLaunchKernelGGL((cross_k<G, NV, VEC>), grid, dim3(256), shmem, st, x0, x_ld, xinit, w, b, L, B, d, out, out_ld, head_w, head_out); \
    } while (0)
    if (nv <= 2) DIR_GO(2);
    else if (nv <= 4) DIR_GO(4);
    else if (nv <= 8) DIR_GO(8);
    else if (nv <= 13) DIR_GO(13);
    else DIR_GO(16);
#undef DIR_GO
    return 0;
}

static int cross_dispatch(const float* x0, int64_t x_ld, const float* xinit, const float* w, const float* b, int L,
                          int64_t B, int d, float* out, int64_t out_ld, dir_stream_t stream, const float* head_w = nullptr,
                          float* head_out = nullptr) {
    DIR_CHECK_ARG(x0 && (out || head_out) && ((w && b) || L == 0) && ((head_w == nullptr) == (head_out == nullptr)), "dir_dcn_cross_f32: null pointer");
    DIR_CHECK_ARG(L >= 0 && d > 0 && B >= 0 && x_ld >= d && (!out || out_ld >= d), "dir_dcn_cross_f32: L=%d d=%d B=%lld x_ld=%lld out_ld=%lld", L, d, (long long)B, (long long)x_ld, (long long)out_ld);
    if (B == 0) return DIR_OK;
    const size_t shmem = ((size_t)2 * L + (head_w ? 1 : 0)) * d * sizeof(float);
    if (shmem > 64 * 1024) return fail(DIR_E_UNSUPPORTED, "dir_dcn_cross_f32: L*d=%d exceeds the 64 KiB LDS weight image", L * d);
    const bool vec = (d % 4 == 0) && (x_ld % 4 == 0) && (!out || ((out_ld % 4 == 0) && aligned16(out))) && aligned16(x0) &&
                     (!xinit || aligned16(xinit)) && (L == 0 || (aligned16(w) && aligned16(b)));
    const int nchunk = vec ? d / 4 : d;
    // smallest group width that keeps <= 16 chunks per lane
    static const int g_env = getenv("DIR_CROSS_G") ? atoi(getenv("DIR_CROSS_G")) : 0;
    // lanes per sample: the smallest power of two (>= 8) that leaves <= 4 chunks per lane.  Few registers
    // per lane = many waves in flight; measured at d = 416: G = 8 / 16 / 32 -> 45.1 / 42.0 / 40.6 us.
    int G = 8;
    while (G < 64 && (nchunk + G - 1) / G > 4) G <<= 1;
    if (g_env == 8 || g_env == 16 || g_env == 32 || g_env == 64) G = g_env;
    while (G < 64 && (nchunk + G - 1) / G > 16) G <<= 1;
    const int nv = (nchunk + G - 1) / G;
    if (nv > 16) return fail(DIR_E_UNSUPPORTED, "dir_dcn_cross_f32: d=%d too wide for the register-resident kernel", d);
    const int spw = 64 / G;
    const int64_t waves = (B + spw - 1) / spw;
    const int64_t grid = (waves + 3) / 4;   // work blocks; the launch picks the resident count
    hipStream_t st = as_stream(stream);
#define DIR_G(GG)                                                                                        \
    if (vec) launch_cross_nv<GG, 4>(nv, grid, shmem, st, x0, x_ld, xinit, w, b, L, B, d, out, out_ld, head_w, head_out);   \
    else launch_cross_nv<GG, 1>(nv, grid, shmem, st, x0, x_ld, xinit, w, b, L, B, d, out, out_ld, head_w, head_out)
    switch (G) {
        case 8: DIR_G(8); break;
        case 16: DIR_G(16); break;
        case 32: DIR_G(32); break;
        default: DIR_G(64); break;
    }
#undef DIR_G
    DIR_CHECK_LAUNCH("dcn_cross");
    return DIR_OK;
}

extern "C" int dir_dcn_cross_f32(const float* x0, int64_t x_ld, const float* w, const float* b, int L, int64_t B,
                                 int d, float* out, int64_t out_ld, dir_stream_t stream) {
    return cross_dispatch(x0, x_ld, nullptr, w, b, L, B, d, out, out_ld, stream);
}

extern "C" int dir_dcn_cross_op_f32(const float* x0, const float* x, int64_t x_ld, const float* w, const float* b,
                                    int64_t B, int d, float* out, int64_t out_ld, dir_stream_t stream) {
    DIR_CHECK_ARG(x, "dir_dcn_cross_op_f32: null pointer");
    return cross_dispatch(x0, x_ld, x, w, b, 1, B, d, out, out_ld, stream);
}

// _cross_architecture followed by the cross branch's share of the final dense(1) (DeepCrossNetwork.py:136-137):
// head_out[r] = x_L[r] . head_w.  out may be NULL: x_L then never reaches memory.
extern "C" int dir_dcn_cross_head_f32(const float* x0, int64_t x_ld, const float* w, const float* b, int L, int64_t B, int d,
                                      const float* head_w, float* out, int64_t out_ld, float* head_out, dir_stream_t stream) {
    DIR_CHECK_ARG(head_w && head_out, "dir_dcn_cross_head_f32: null pointer");
    return cross_dispatch(x0, x_ld, nullptr, w, b, L, B, d, out, out_ld, stream, head_w, head_out);
}
