// din.hip -- DIN local activation unit + weighted history pooling for gfx950.
//
// NO REFERENCE CODE: /root/reference/README.md:27 only links arXiv:1706.06978.  The definition this
// kernel implements is the one in include/dir_hip.h and oracle/dir_oracle.c.
//
// One 256-thread workgroup owns one sample at a time (grid-stride over samples):
//   1. gather: the T history rows and the candidate row are read from HBM as whole rows (K*4 bytes,
//      16 B per lane) and the four-part activation input u_j = [h, a, h-a, h*a] is written to LDS;
//   2. layer 1 (4K -> H1) and layer 2 (H1 -> H2): each thread owns a JT x 4 register tile of outputs,
//      reads u / z1 rows from LDS as 16-byte broadcasts and the weights as 16-byte coalesced loads
//      (L1/L2 resident: W1 + W2 = 93 KB at the BASELINE shape);
//   3. layer 3 + optional masked softmax across the T scores (wave64 max / sum butterflies);
//   4. pooling: thread k sums w_j * h_j[k] over the LDS-staged history.
// LDS rows are padded by 16 B so that the row-broadcast reads of different history positions land in
// different banks.
#include "common.hpp"

namespace dir {

__device__ __forceinline__ float sigmoidf_(float x) { return 1.0f / (1.0f + expf(-x)); }

struct DinDims {
    int K, T, H1, H2;
    int us;   // u row stride (floats)  = 4K + 4
    int z1s;  // z1 row stride          = H1 + 4
    int z2s;  // z2 row stride          = H2 + 4
};

// dense layer over LDS rows: outp[j][n] = act( sum_i in[j][i] * W[i][n] + bias[n] ), n in groups of 4
template <int JT, bool SIGMOID>
__device__ __forceinline__ void dense_rows(const float* __restrict__ in, int in_stride, int in_dim,
                                           const float* __restrict__ W, const float* __restrict__ bias, int N,
                                           float* __restrict__ outp, int out_stride, int T) {
    const int NG = N >> 2;
    const int njt = (T + JT - 1) / JT;
    const int nitems = NG * njt;
    for (int it = threadIdx.x; it < nitems; it += blockDim.x) {
        const int ng = it % NG;
        const int jt = it / NG;
        float4 acc[JT];
#pragma unroll
        for (int jj = 0; jj < JT; ++jj) acc[jj] = make_float4(0.f, 0.f, 0.f, 0.f);
        const float* wp = W + 4 * ng;
        for (int i4 = 0; i4 < (in_dim >> 2); ++i4) {
            const float4 w0 = *reinterpret_cast<const float4*>(wp + (size_t)(4 * i4 + 0) * N);
            const float4 w1 = *reinterpret_cast<const float4*>(wp + (size_t)(4 * i4 + 1) * N);
            const float4 w2 = *reinterpret_cast<const float4*>(wp + (size_t)(4 * i4 + 2) * N);
            const float4 w3 = *reinterpret_cast<const float4*>(wp + (size_t)(4 * i4 + 3) * N);
#pragma unroll
            for (int jj = 0; jj < JT; ++jj) {
                int j = jt * JT + jj;
                j = j < T ? j : T - 1;
                const float4 u = *reinterpret_cast<const float4*>(in + j * in_stride + 4 * i4);
                float4 a = acc[jj];
                a.x = fmaf(u.x, w0.x, a.x); a.y = fmaf(u.x, w0.y, a.y); a.z = fmaf(u.x, w0.z, a.z); a.w = fmaf(u.x, w0.w, a.w);
                a.x = fmaf(u.y, w1.x, a.x); a.y = fmaf(u.y, w1.y, a.y); a.z = fmaf(u.y, w1.z, a.z); a.w = fmaf(u.y, w1.w, a.w);
                a.x = fmaf(u.z, w2.x, a.x); a.y = fmaf(u.z, w2.y, a.y); a.z = fmaf(u.z, w2.z, a.z); a.w = fmaf(u.z, w2.w, a.w);
                a.x = fmaf(u.w, w3.x, a.x); a.y = fmaf(u.w, w3.y, a.y); a.z = fmaf(u.w, w3.z, a.z); a.w = fmaf(u.w, w3.w, a.w);
                acc[jj] = a;
            }
        }
        const float4 bv = *reinterpret_cast<const float4*>(bias + 4 * ng);
#pragma unroll
        for (int jj = 0; jj < JT; ++jj) {
            const int j = jt * JT + jj;
            if (j < T) {
                float4 r = make_float4(acc[jj].x + bv.x, acc[jj].y + bv.y, acc[jj].z + bv.z, acc[jj].w + bv.w);
                if (SIGMOID) r = make_float4(sigmoidf_(r.x), sigmoidf_(r.y), sigmoidf_(r.z), sigmoidf_(r.w));
                *reinterpret_cast<float4*>(outp + j * out_stride + 4 * ng) = r;
            }
        }
    }
}

template <int JT>
__global__ __launch_bounds__(256) void din_k(const float* __restrict__ table, DinDims dm,
                                             const int64_t* __restrict__ hist,
                                             const int32_t* __restrict__ hist_len,
                                             const int64_t* __restrict__ cand, const float* __restrict__ W1,
                                             const float* __restrict__ b1, const float* __restrict__ W2,
                                             const float* __restrict__ b2, const float* __restrict__ W3,
                                             const float* __restrict__ b3, int normalize, int64_t B,
                                             float* __restrict__ out, float* __restrict__ scores) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int K = dm.K, T = dm.T, H1 = dm.H1, H2 = dm.H2;
    float* u = smem;                       // [T][us]
    float* z1 = u + T * dm.us;             // [T][z1s]
    float* z2 = z1 + T * dm.z1s;           // [T][z2s]
    float* sc = z2 + T * dm.z2s;           // [T rounded up to 64]
    int* valid = reinterpret_cast<int*>(sc + ((T + 63) & ~63));  // [T]
    const int kc = K >> 2;  // 16-byte chunks per row
    const float inv_sqrt_k = 1.0f / sqrtf((float)K);

    for (int64_t b = blockIdx.x; b < B; b += gridDim.x) {
        const int len = hist_len ? min((int)hist_len[b], T) : T;   // rows j >= len are masked: never computed
        const int64_t cid = cand[b];
        // 1. gather rows -> u
        for (int j = len + threadIdx.x; j < T; j += blockDim.x) valid[j] = 0;
        for (int q = threadIdx.x; q < len * kc; q += blockDim.x) {
            const int j = q / kc, c = q - j * kc;
            const int64_t id = hist[b * T + j];
            const bool ok = (j < len) && (id >= 0);
            float4 h = make_float4(0.f, 0.f, 0.f, 0.f), a = h;
            if (ok) {
                h = *reinterpret_cast<const float4*>(table + id * K + 4 * c);
                if (cid >= 0) a = *reinterpret_cast<const float4*>(table + cid * K + 4 * c);
            }
            float* ur = u + j * dm.us + 4 * c;
            *reinterpret_cast<float4*>(ur) = h;
            *reinterpret_cast<float4*>(ur + K) = a;
            *reinterpret_cast<float4*>(ur + 2 * K) = make_float4(h.x - a.x, h.y - a.y, h.z - a.z, h.w - a.w);
            *reinterpret_cast<float4*>(ur + 3 * K) = make_float4(h.x * a.x, h.y * a.y, h.z * a.z, h.w * a.w);
            if (c == 0) valid[j] = ok ? 1 : 0;
        }
        __syncthreads();
        // 2. the two hidden layers
        dense_rows<JT, true>(u, dm.us, 4 * K, W1, b1, H1, z1, dm.z1s, len);
        __syncthreads();
        dense_rows<JT, true>(z1, dm.z1s, H1, W2, b2, H2, z2, dm.z2s, len);
        __syncthreads();
        // 3. scores
        for (int j = threadIdx.x; j < T; j += blockDim.x) {
            float acc = 0.f;
            if (j < len) {
                const float* zr = z2 + j * dm.z2s;
                for (int i = 0; i < H2; ++i) acc = fmaf(zr[i], W3[i], acc);
            }
            sc[j] = (j < len && valid[j]) ? acc + b3[0] : 0.f;
        }
        __syncthreads();
        if (normalize) {
            if (threadIdx.x < 64) {  // one wave: masked softmax over the T scores
                float mx = -INFINITY;
                for (int j = threadIdx.x; j < T; j += 64)
                    if (valid[j]) mx = fmaxf(mx, sc[j] * inv_sqrt_k);
                mx = wave_max(mx);
                float sum = 0.f;
                for (int j = threadIdx.x; j < T; j += 64) {
                    float e = 0.f;
                    if (valid[j]) e = expf(sc[j] * inv_sqrt_k - mx);
                    sc[j] = e;
                    sum += e;
                }
                sum = wave_sum(sum);
                for (int j = threadIdx.x; j < T; j += 64)
                    if (valid[j]) sc[j] = sc[j] / sum;
            }
            __syncthreads();
        }
        // 4. pooling + optional score output
        for (int k = threadIdx.x; k < K; k += blockDim.x) {
            float acc = 0.f;
            for (int j = 0; j < len; ++j)
                if (valid[j]) acc = fmaf(sc[j], u[j * dm.us + k], acc);
            out[b * K + k] = acc;
        }
        if (scores)
            for (int j = threadIdx.x; j < T; j += blockDim.x) scores[b * T + j] = sc[j];
        __syncthreads();
    }
}

}  // namespace dir

using namespace dir;

extern "C" int dir_din_attention_pool_f32(const float* table, int K, const int64_t* hist, const int32_t* hist_len,
                                          const int64_t* cand, int T, const float* W1, const float* b1, int H1,
                                          const float* W2, const float* b2, int H2, const float* W3,
                                          const float* b3, int normalize, int64_t B, float* out, float* scores,
                                          dir_stream_t stream) {
    DIR_CHECK_ARG(table && hist && cand && W1 && b1 && W2 && b2 && W3 && b3 && out, "dir_din_attention_pool_f32: null pointer");
    DIR_CHECK_ARG(K > 0 && T > 0 && H1 > 0 && H2 > 0 && B >= 0, "dir_din_attention_pool_f32: K=%d T=%d H1=%d H2=%d", K, T, H1, H2);
    if ((K & 3) || (H1 & 3) || (H2 & 3))
        return fail(DIR_E_UNSUPPORTED, "dir_din_attention_pool_f32: K, H1, H2 must be multiples of 4 (K=%d H1=%d H2=%d)", K, H1, H2);
    if (!aligned16(table) || !aligned16(W1) || !aligned16(W2) || !aligned16(b1) || !aligned16(b2))
        return fail(DIR_E_BADARG, "dir_din_attention_pool_f32: table / W1 / W2 / b1 / b2 must be 16-byte aligned");
    if (B == 0) return DIR_OK;
    DinDims dm{K, T, H1, H2, 4 * K + 4, H1 + 4, H2 + 4};
    const size_t shmem = sizeof(float) * ((size_t)T * (dm.us + dm.z1s + dm.z2s) + ((T + 63) & ~63)) + sizeof(int) * (size_t)T;
    if (shmem > 160 * 1024) return fail(DIR_E_UNSUPPORTED, "dir_din_attention_pool_f32: T=%d K=%d needs %zu B of LDS (> 160 KiB)", T, K, shmem);
    static bool attr_set = false;
    if (!attr_set) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&din_k<5>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        attr_set = true;
    }
    const int per_cu = (int)((160 * 1024) / shmem) < 1 ? 1 : (int)((160 * 1024) / shmem);
    dim3 grid((unsigned)(B < (int64_t)kCUs * per_cu * 4 ? B : (int64_t)kCUs * per_cu * 4));
    hipLaunchKernelGGL((din_k<5>), grid, dim3(256), shmem, as_stream(stream), table, dm, hist, hist_len, cand, W1, b1, W2, b2, W3, b3, normalize, B, out, scores);
    DIR_CHECK_LAUNCH("din_attention_pool");
    return DIR_OK;
}
