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

// ------------------------------------------------------------------------------------------------
// MFMA path (v_mfma_f32_16x16x4_f32, exact fp32) for T <= 64 and the instantiated (K, H1, H2) shapes.
//
// Layer 1 is regrouped (same unit, fewer flops):
//     [h, a, h-a, h*a] . W1  =  h . (Wh + Wd)  +  (h*a) . Wp  +  a . (Wa - Wd)
// The last term does not depend on the history position: it is one K-long dot per output column per
// SAMPLE (VALU, folded into the accumulator's initial value together with b1).  Per history row the
// reduction is 2K instead of 4K.  fp32 rounding differs from the sequential 4K sum by ~1e-7 relative
// (pre-added weights, different association), inside the 1e-5 bar; tests compare against the oracle.
//
// GEMM view: rows = history positions (RT tiles of 16), cols = H1 (tiles of 16).  Wave w owns column tile w
// and keeps its B fragments (Wh+Wd, Wp, Wa-Wd: 3 x K/4 registers) for the whole launch.  In the 16x16x4
// instruction lane l supplies A[row l&15][k = l>>4]; lane group kk covers features [kk*K/4, (kk+1)*K/4), so
// a lane reads its K/4 features of h[row] from LDS as 16-byte chunks, uses them directly as the A operand
// of the (Wh+Wd) chain and multiplied by its K/4 candidate values (registers) for the Wp chain.
// Layer 2 (waves 0..NC2-1) reads z1 from LDS the same way; layer 3, the masked softmax (wave butterflies)
// and the pooling are VALU work on the LDS-staged rows.  The row-tile count is a template parameter, so
// the MFMA loops are branch-free.
// LDS per workgroup at (64, 80, 40): 64 x (68 + 84 + 52) x 4 B = 52 KB.
// ------------------------------------------------------------------------------------------------
typedef float f32x4m __attribute__((ext_vector_type(4)));

// v_exp + v_rcp (1 ulp each): a correctly rounded division is a 10-instruction sequence, and VALU work is not
// hidden behind fp32 MFMAs
__device__ __forceinline__ float fast_sigmoid(float x) { return __builtin_amdgcn_rcpf(1.0f + __expf(-x)); }

template <int K, int NC1, int NC2>
struct DinSh {
    static constexpr int NW = 4;                      // waves per workgroup: one per SIMD
    static constexpr bool SHARED = NC1 == 5;          // a 5th column tile, its reduction split over the 4 waves
    static constexpr int H1P = 16 * NC1, H2P = 16 * NC2;
    static constexpr int HS = K + 4, Z1S = H1P + 4;
    static constexpr int NPART = (64 * NW) / K;       // pooling partitions (threads per output column)
    float uh[64 * HS];          // history rows
    float av[K];                // candidate row
    float z1[64 * Z1S];         // layer-1 activations
    float part[SHARED ? NW * 64 * 16 : 4];   // shared-tile partial pre-activations, one slab per wave
    float scp[NC2 * 64];        // layer-3 partial scores, one slab per layer-2 column tile
    float sc[64];               // final weights
    float pool[NPART * K];      // pooling partials
    int valid[64];
};

// B fragments of one wave (registers, loaded once per launch).  Step (it, e) of the own tile covers feature
// kk*KQ + 4*((it + rot) % QN) + e, where rot = wave id when a shared tile exists (so that every wave meets
// ITS quarter of the shared tile's reduction in iteration 0) and 0 otherwise.
template <int K, int NC1>
struct DinFrag {
    float whd[K / 4], wp[K / 4], wc[K / 4];   // own column tile: Wh+Wd, Wp, Wa-Wd
    float whd_s[4], wp_s[4], wc_s[4];         // this wave's quarter of the shared tile (NC1 == 5)
    float wr2[4 * NC1];                       // layer 2 (waves 0..NC2-1)
};

// Layers 1-3 for a compile-time number of row tiles.  Leaves the per-column-tile partial scores in sh.scp.
// KEEP (the backward kernel's recompute): also leaves the layer-2 activations in z2out[row][H2P + 4].
template <int K, int NC1, int NC2, int RT, bool KEEP = false>
__device__ __forceinline__ void din_mlp_tiles(DinSh<K, NC1, NC2>& sh, const DinFrag<K, NC1>& fr, float bias1, float bias1_s,
                                              float bias2, float w3v, int w, int r16, int kk, float* z2out = nullptr) {
    using S = DinSh<K, NC1, NC2>;
    constexpr int KQ = K / 4;          // features per lane group = k-steps per chain
    constexpr int QN = KQ / 4;         // 16-byte groups per lane
    constexpr int KS2 = S::H1P / 4;
    constexpr int FT = NC1 < 4 ? NC1 : 4;   // column tiles owned by whole waves
    static_assert(!S::SHARED || QN == 4, "the shared-tile split needs K = 64");
    const int rot = S::SHARED ? w : 0;
    const int tid = threadIdx.x;
    if (w < FT || S::SHARED) {
        // per-sample term a.(Wa-Wd), folded into the accumulators' initial values
        float cpart = 0.f, cpart_s = 0.f;
#pragma unroll
        for (int it = 0; it < QN; ++it) {
            const int g = (it + rot) % QN;
            const float4 a4 = *reinterpret_cast<const float4*>(sh.av + kk * KQ + 4 * g);
            cpart = fmaf(a4.x, fr.wc[4 * it], cpart);
            cpart = fmaf(a4.y, fr.wc[4 * it + 1], cpart);
            cpart = fmaf(a4.z, fr.wc[4 * it + 2], cpart);
            cpart = fmaf(a4.w, fr.wc[4 * it + 3], cpart);
            if (S::SHARED && it == 0) {
                cpart_s = fmaf(a4.x, fr.wc_s[0], cpart_s);
                cpart_s = fmaf(a4.y, fr.wc_s[1], cpart_s);
                cpart_s = fmaf(a4.z, fr.wc_s[2], cpart_s);
                cpart_s = fmaf(a4.w, fr.wc_s[3], cpart_s);
            }
        }
        cpart += __shfl_xor(cpart, 16, 64);
        cpart += __shfl_xor(cpart, 32, 64);
        const float init = cpart + bias1;
        float init_s = 0.f;
        if (S::SHARED) {
            cpart_s += __shfl_xor(cpart_s, 16, 64);
            cpart_s += __shfl_xor(cpart_s, 32, 64);
            init_s = cpart_s + (w == 0 ? bias1_s : 0.f);
        }
        f32x4m acch[RT], accp[RT], accs[RT];
#pragma unroll
        for (int rt = 0; rt < RT; ++rt) {
            acch[rt] = (f32x4m){init, init, init, init};
            accp[rt] = (f32x4m){0.f, 0.f, 0.f, 0.f};
            accs[rt] = (f32x4m){init_s, init_s, init_s, init_s};
        }
#pragma unroll
        for (int it = 0; it < QN; ++it) {
            const int g = (it + rot) % QN;
            const float4 a4 = *reinterpret_cast<const float4*>(sh.av + kk * KQ + 4 * g);   // group-uniform: broadcast
            const float aq[4] = {a4.x, a4.y, a4.z, a4.w};
            float hv[RT][4];
#pragma unroll
            for (int rt = 0; rt < RT; ++rt) {
                const float4 h4 = *reinterpret_cast<const float4*>(sh.uh + (rt * 16 + r16) * S::HS + kk * KQ + 4 * g);
                hv[rt][0] = h4.x; hv[rt][1] = h4.y; hv[rt][2] = h4.z; hv[rt][3] = h4.w;
            }
#pragma unroll
            for (int e = 0; e < 4; ++e) {
#pragma unroll
                for (int rt = 0; rt < RT; ++rt) {
                    const float hp = hv[rt][e] * aq[e];
                    acch[rt] = __builtin_amdgcn_mfma_f32_16x16x4f32(hv[rt][e], fr.whd[4 * it + e], acch[rt], 0, 0, 0);
                    accp[rt] = __builtin_amdgcn_mfma_f32_16x16x4f32(hp, fr.wp[4 * it + e], accp[rt], 0, 0, 0);
                    if (S::SHARED && it == 0) {
                        accs[rt] = __builtin_amdgcn_mfma_f32_16x16x4f32(hv[rt][e], fr.whd_s[e], accs[rt], 0, 0, 0);
                        accs[rt] = __builtin_amdgcn_mfma_f32_16x16x4f32(hp, fr.wp_s[e], accs[rt], 0, 0, 0);
                    }
                }
            }
        }
        if (w < FT) {
#pragma unroll
            for (int rt = 0; rt < RT; ++rt)
#pragma unroll
                for (int g = 0; g < 4; ++g)
                    sh.z1[(rt * 16 + 4 * kk + g) * S::Z1S + 16 * w + r16] = fast_sigmoid(acch[rt][g] + accp[rt][g]);
        }
        if (S::SHARED) {
#pragma unroll
            for (int rt = 0; rt < RT; ++rt)
#pragma unroll
                for (int g = 0; g < 4; ++g) sh.part[(w * 64 + rt * 16 + 4 * kk + g) * 16 + r16] = accs[rt][g];
        }
    }
    __syncthreads();
    if (S::SHARED) {   // finish the shared tile: add the four partial pre-activations
        for (int idx = tid; idx < RT * 256; idx += 64 * S::NW) {
            const int row = idx >> 4, col = idx & 15;
            const float v = (sh.part[(0 * 64 + row) * 16 + col] + sh.part[(1 * 64 + row) * 16 + col]) +
                            (sh.part[(2 * 64 + row) * 16 + col] + sh.part[(3 * 64 + row) * 16 + col]);
            sh.z1[row * S::Z1S + 64 + col] = fast_sigmoid(v);
        }
        __syncthreads();
    }
    if (w < NC2) {
        f32x4m acc2[RT];
#pragma unroll
        for (int rt = 0; rt < RT; ++rt) acc2[rt] = (f32x4m){bias2, bias2, bias2, bias2};
#pragma unroll
        for (int q = 0; q < KS2 / 4; ++q) {
            float zv[RT][4];
#pragma unroll
            for (int rt = 0; rt < RT; ++rt) {
                const float4 z4 = *reinterpret_cast<const float4*>(sh.z1 + (rt * 16 + r16) * S::Z1S + kk * KS2 + 4 * q);
                zv[rt][0] = z4.x; zv[rt][1] = z4.y; zv[rt][2] = z4.z; zv[rt][3] = z4.w;
            }
#pragma unroll
            for (int e = 0; e < 4; ++e)
#pragma unroll
                for (int rt = 0; rt < RT; ++rt)
                    acc2[rt] = __builtin_amdgcn_mfma_f32_16x16x4f32(zv[rt][e], fr.wr2[4 * q + e], acc2[rt], 0, 0, 0);
        }
        // layer 3 fused: s[row] += sum over this tile's 16 columns of sigmoid(z2) * W3[col]
#pragma unroll
        for (int rt = 0; rt < RT; ++rt)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const float z2v = fast_sigmoid(acc2[rt][g]);
                if (KEEP) z2out[(rt * 16 + 4 * kk + g) * (S::H2P + 4) + 16 * w + r16] = z2v;
                const float v = row16_sum(z2v * w3v);
                if (r16 == 0) sh.scp[w * 64 + rt * 16 + 4 * kk + g] = v;
            }
    }
}

#ifdef DIN_STAMP   // tools/din_probe.hip only: cycles per phase, summed over workgroups (thread 0) -- [0] stage [1] mlp [2] softmax
__device__ unsigned long long din_stamp[8];   // [3] pool+out [4] samples [5] workgroups [6] total
#define DIN_T(v) const unsigned long long v = __builtin_readcyclecounter()
#define DIN_ACC(i, a, b) if (threadIdx.x == 0) atomicAdd(&din_stamp[i], (b) - (a))
#else
#define DIN_T(v)
#define DIN_ACC(i, a, b)
#endif

template <int K, int NC1, int NC2>
__global__ __launch_bounds__(256, 2) void din_mfma_k(const float* __restrict__ table,
                                                       const int64_t* __restrict__ hist,
                                                       const int32_t* __restrict__ hist_len,
                                                       const int64_t* __restrict__ cand, int T,
                                                       const float* __restrict__ W1, const float* __restrict__ b1,
                                                       int H1, const float* __restrict__ W2,
                                                       const float* __restrict__ b2, int H2,
                                                       const float* __restrict__ W3, const float* __restrict__ b3,
                                                       int normalize, int64_t B, float* __restrict__ out,
                                                       float* __restrict__ scores) {
    using S = DinSh<K, NC1, NC2>;
    constexpr int NT = 64 * S::NW;
    constexpr int KQ = K / 4;
    constexpr int QN = KQ / 4;
    constexpr int KS2 = S::H1P / 4;
    constexpr int KC = K / 4;
    constexpr int NPF = (64 * KC + NT - 1) / NT;   // row chunks a thread stages per sample (T <= 64)
    static_assert(NC1 <= 5 && NC2 <= 4, "column tiles: at most 4 whole waves + 1 shared tile");
    __shared__ __attribute__((aligned(16))) S sh;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int w = tid >> 6;
    const int r16 = lane & 15;
    const int kk = lane >> 4;
    const float inv_sqrt_k = 1.0f / sqrtf((float)K);

    // ---- B fragments, resident for the whole launch -----------------------------------------------------
    DinFrag<K, NC1> fr;
    auto w1parts = [&](int f, int col, float& hd, float& c, float& p) {
        float vh = 0.f, va = 0.f, vd = 0.f, vp = 0.f;
        if (col < H1) {
            vh = W1[(size_t)f * H1 + col];
            va = W1[(size_t)(K + f) * H1 + col];
            vd = W1[(size_t)(2 * K + f) * H1 + col];
            vp = W1[(size_t)(3 * K + f) * H1 + col];
        }
        hd = vh + vd;
        c = va - vd;
        p = vp;
    };
    {
        const int rot = S::SHARED ? w : 0;
        const int col = 16 * w + r16;      // own tile (waves >= NC1 get zeros through the col < H1 test)
#pragma unroll
        for (int it = 0; it < QN; ++it)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int f = kk * KQ + 4 * ((it + rot) % QN) + e;
                w1parts(f, w < NC1 ? col : H1, fr.whd[4 * it + e], fr.wc[4 * it + e], fr.wp[4 * it + e]);
            }
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int f = kk * KQ + 4 * (rot % QN) + e;
            w1parts(f, S::SHARED ? 64 + r16 : H1, fr.whd_s[e], fr.wc_s[e], fr.wp_s[e]);
        }
    }
    const float bias1 = (w < NC1 && (16 * w + r16) < H1) ? b1[16 * w + r16] : 0.f;
    const float bias1_s = (S::SHARED && (64 + r16) < H1) ? b1[64 + r16] : 0.f;
    float bias2 = 0.f, w3v = 0.f;
#pragma unroll
    for (int s2 = 0; s2 < KS2; ++s2) {
        const int i = kk * KS2 + s2;
        const int col = 16 * w + r16;
        fr.wr2[s2] = (w < NC2 && i < H1 && col < H2) ? W2[(size_t)i * H2 + col] : 0.f;
    }
    if (w < NC2 && (16 * w + r16) < H2) {
        bias2 = b2[16 * w + r16];
        w3v = W3[16 * w + r16];
    }
    const float bias3 = b3[0];

    // ---- software pipeline over this workgroup's samples: per-sample scalars (length, candidate id) three samples ahead,
    // history ids two ahead, rows one ahead.  The loads are conditional, so the compiler can only wait for one with vmcnt(0):
    // every consumer therefore runs BEFORE the iteration issues anything new (a younger load in flight would be waited for too).
    const int64_t G = gridDim.x;
    auto load_scalars = [&](int64_t bb, int& len, int64_t& cid) {
        len = 0;
        cid = -1;
        if (bb < B) {
            len = hist_len ? min((int)hist_len[bb], T) : T;
            cid = cand[bb];
        }
    };
    auto load_ids = [&](int64_t bb, int len, int64_t (&ids)[NPF]) {
#pragma unroll
        for (int k = 0; k < NPF; ++k) {
            const int q = tid + k * NT;
            ids[k] = -1;
            if (q < len * KC) ids[k] = hist[bb * T + q / KC];      // len == 0 beyond the batch
        }
    };
    auto load_rows = [&](int len, int64_t cid, const int64_t (&ids)[NPF], float4 (&h)[NPF], float4& a, int& vmask) {
        vmask = 0;
#pragma unroll
        for (int k = 0; k < NPF; ++k) {
            const int q = tid + k * NT;
            h[k] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (q < len * KC && ids[k] >= 0) {
                h[k] = *reinterpret_cast<const float4*>(table + ids[k] * K + 4 * (q % KC));
                vmask |= 1 << k;
            }
        }
        a = make_float4(0.f, 0.f, 0.f, 0.f);
        if (tid < KC && cid >= 0) a = *reinterpret_cast<const float4*>(table + cid * K + 4 * tid);
    };
    int len0, len1, len2, vmask;                 // 0: the sample whose rows are in hreg; 1: ids in idn; 2: the one after
    int64_t cid0, cid1, cid2, idn[NPF];
    float4 hreg[NPF], areg4;
    load_scalars(blockIdx.x, len0, cid0);
    load_scalars(blockIdx.x + G, len1, cid1);
    load_scalars(blockIdx.x + 2 * G, len2, cid2);
    load_ids(blockIdx.x, len0, idn);
    load_rows(len0, cid0, idn, hreg, areg4, vmask);
    load_ids(blockIdx.x + G, len1, idn);

    DIN_T(st_begin);
    for (int64_t b = blockIdx.x; b < B; b += G) {
        DIN_T(st0);
        const int len = len0;
        const int RT = (len + 15) >> 4;
        // ---- stage this sample's rows (already in registers) into LDS --------------------------------------
#pragma unroll
        for (int k = 0; k < NPF; ++k) {
            const int q = tid + k * NT;
            if (q < len * KC) {
                const int j = q / KC, c = q - j * KC;
                *reinterpret_cast<float4*>(sh.uh + j * S::HS + 4 * c) = hreg[k];
                if (c == 0) sh.valid[j] = (vmask >> k) & 1;
            }
        }
        for (int j = len + tid; j < 64; j += NT) sh.valid[j] = 0;
        if (tid < KC) *reinterpret_cast<float4*>(sh.av + 4 * tid) = areg4;
        __syncthreads();
        DIN_T(st1);
        // ---- issue the next sample's row reads and the ids of the one after; both land during the MFMAs -------
        len0 = len1; cid0 = cid1;
        len1 = len2; cid1 = cid2;
        load_rows(len0, cid0, idn, hreg, areg4, vmask);
        load_ids(b + 2 * G, len1, idn);
        load_scalars(b + 3 * G, len2, cid2);
        // ---- hidden layers + fused layer 3, branch-free per row-tile count -----------------------------------
        switch (RT) {   // block-uniform
            case 1: din_mlp_tiles<K, NC1, NC2, 1>(sh, fr, bias1, bias1_s, bias2, w3v, w, r16, kk); break;
            case 2: din_mlp_tiles<K, NC1, NC2, 2>(sh, fr, bias1, bias1_s, bias2, w3v, w, r16, kk); break;
            case 3: din_mlp_tiles<K, NC1, NC2, 3>(sh, fr, bias1, bias1_s, bias2, w3v, w, r16, kk); break;
            case 4: din_mlp_tiles<K, NC1, NC2, 4>(sh, fr, bias1, bias1_s, bias2, w3v, w, r16, kk); break;
            default:   // len == 0: keep the barrier count uniform
                __syncthreads();
                if (S::SHARED) __syncthreads();
                break;
        }
        __syncthreads();
        DIN_T(st2);
        // ---- scores: sum the column-tile partials, masked softmax (wave 0) -----------------------------------
        if (tid < 64) {
            const bool ok = tid < len && sh.valid[tid];
            float sv = 0.f;
            if (ok) {
                sv = bias3;
#pragma unroll
                for (int c2 = 0; c2 < NC2; ++c2) sv += sh.scp[c2 * 64 + tid];
            }
            if (normalize) {
                const float x = sv * inv_sqrt_k;
                const float mx = wave_max_dpp(ok ? x : -INFINITY);
                const float ex = ok ? __expf(x - mx) : 0.f;
                const float sum = wave_sum_dpp(ex);
                sv = ok ? ex / sum : 0.f;
            }
            sh.sc[tid] = sv;
        }
        __syncthreads();
        DIN_T(st3);
        // ---- pooling: NPART threads per output column, then one add tree -------------------------------------
        if (tid < S::NPART * K) {
            const int k = tid % K, part = tid / K;
            float acc_o = 0.f;
            for (int j = part; j < len; j += S::NPART)
                if (sh.valid[j]) acc_o = fmaf(sh.sc[j], sh.uh[j * S::HS + k], acc_o);
            sh.pool[part * K + k] = acc_o;
        }
        if (scores)
            for (int j = tid; j < T; j += NT) scores[b * T + j] = j < 64 ? sh.sc[j] : 0.f;
        __syncthreads();
        if (tid < K) {
            float acc_o = 0.f;
#pragma unroll
            for (int p = 0; p < S::NPART; ++p) acc_o += sh.pool[p * K + tid];
            out[b * K + tid] = acc_o;
        }
        DIN_T(st4);
        DIN_ACC(0, st0, st1); DIN_ACC(1, st1, st2); DIN_ACC(2, st2, st3); DIN_ACC(3, st3, st4);
        DIN_ACC(4, 0ull, 1ull);
        // next iteration's first LDS writes (uh, av, valid) do not touch pool/sc/scp; its later phases are
        // ordered behind its own barriers
    }
    DIN_T(st_end);
    DIN_ACC(6, st_begin, st_end);
    DIN_ACC(5, 0ull, 1ull);
}

template <int K, int NC1, int NC2>
static int launch_din_mfma(hipStream_t st, const float* table, const int64_t* hist, const int32_t* hist_len,
                           const int64_t* cand, int T, const float* W1, const float* b1, int H1, const float* W2,
                           const float* b2, int H2, const float* W3, const float* b3, int normalize, int64_t B,
                           float* out, float* scores) {
    const int res = resident_blocks(din_mfma_k<K, NC1, NC2>, 0, 256);
    dim3 grid((unsigned)(B < res ? B : res));
    hipLaunchKernelGGL((din_mfma_k<K, NC1, NC2>), grid, dim3(256), 0, st, table, hist, hist_len, cand, T, W1, b1, H1, W2,
                       b2, H2, W3, b3, normalize, B, out, scores);
    return 0;
}


// ------------------------------------------------------------------------------------------------
// Fused backward of the unit + pooling (training).  Shape class (K, H1, H2) = (64, <= 80, <= 48), T <= 64.
// Same workgroup / sample structure as din_mfma_k; per sample:
//   1. recompute layers 1-3 with the forward's code (din_mlp_tiles<KEEP>: z1 and z2 stay in LDS);
//   2. wave 0: weights w (masked softmax or raw scores), d score from dw_j = g . h_j, compact output rows;
//   3. dpre2 = ds * W3 * z2 (1 - z2) in place of z2 (+ the running sums for dW3, db2, db3);
//   4. MFMA, reduction over rows: dW2 += z1^T dpre2;   reduction over H2: dz1 = dpre2 W2^T -> dpre1 = dz1 z1 (1 - z1);
//   5. MFMA, reduction over rows: d[Wh+Wd | Wp] += [h | h*a]^T dpre1;   reduction over H1: dX = dpre1 [Wh+Wd | Wp]^T
//      -> d h_j = dXh + dXp * a + w_j g  (written to the compact row list),  d a += sum_j dXp * h_j.
// The weight gradients accumulate in registers over the whole launch (56 accumulator registers per lane) and leave as one
// partial record per workgroup; din_bwd_sum_k / din_bwd_finish_k add the records.  The per-sample term's gradients (through a.(Wa-Wd) + b1)
// only need S_b = sum_j dpre1, which is written out: the host finishes them with two small GEMMs.
// Row-reduction operands come from the row-major LDS tiles with 4-byte reads (k-step s, lane group kk <-> row 4s + kk);
// column-reduction operands are 16-byte reads as in the forward.
// ------------------------------------------------------------------------------------------------
template <int K, int NC1, int NC2>
struct DinBwdSh {
    using F = DinSh<K, NC1, NC2>;
    static constexpr int Z2S = F::H2P + 4;
    F f;                       // the forward's staging: uh, av, z1, part, scp, sc, valid
    float z2[64 * Z2S];        // layer-2 activations, then dpre2 in place
    float dp1[64 * F::Z1S];    // dpre1
    float gv[K];               // d out of this sample
    float ds[64];              // d score
    float dw[64];              // g . h_j
    float S[64];               // column sums of dpre1, H1 tiles 0-3
    float S4[4 * 16];          // H1 tile 4: one partial per wave (row tile)
    float gaout[K];            // d a of this sample (leaves with the flush)
    int rank[64];              // compact output row of position j (-1: not a valid row)
    float ap[2 * K * F::Z1S];  // [Wh+Wd | Wp] as [2K][H1P + 4]: B fragments of dX
    float w2[F::H1P * Z2S];    // W2 as [H1P][H2P + 4]: B fragments of dz1
};

#ifdef DIN_STAMP   // tools/din_bwd_probe.hip only: [0] stage [1] dw + recompute [2] weights/ds [3] dpre2 [4] dW2/dz1 [5] dAP/dX [6] samples
__device__ unsigned long long din_bwd_stamp[12];
#define DINB_ACC(i, a, b) if (threadIdx.x == 0) atomicAdd(&din_bwd_stamp[i], (b) - (a))
#else
#define DINB_ACC(i, a, b)
#endif

constexpr int kDinBwdGAP = 2 * 64 * 80, kDinBwdGW2 = 80 * 48;
// per-workgroup partial record: gAP [128][80] | gW2 [80][48] | gb2 [4][48] | gW3 [4][48] | gb3 [64]
constexpr int kDinBwdRec = kDinBwdGAP + kDinBwdGW2 + 4 * 48 + 4 * 48 + 64;

template <int K, int NC1, int NC2>
__global__ __launch_bounds__(256, 1) void din_bwd_k(const float* __restrict__ table, const int64_t* __restrict__ hist,
                                                      const int32_t* __restrict__ hist_len, const int64_t* __restrict__ cand,
                                                      int T, const float* __restrict__ W1, const float* __restrict__ b1, int H1,
                                                      const float* __restrict__ W2, const float* __restrict__ b2, int H2,
                                                      const float* __restrict__ W3, const float* __restrict__ b3, int normalize,
                                                      int64_t B, const float* __restrict__ gout,
                                                      const int64_t* __restrict__ row_off, float* __restrict__ gh,
                                                      float* __restrict__ ga, float* __restrict__ Sout,
                                                      float* __restrict__ partials) {
    static_assert(K == 64 && NC1 == 5 && NC2 == 3, "din_bwd_k is written for the (64, 80, 48) shape class");
    using S = DinSh<K, NC1, NC2>;
    using SB = DinBwdSh<K, NC1, NC2>;
    constexpr int NT = 64 * S::NW;
    constexpr int KQ = K / 4;
    constexpr int QN = KQ / 4;
    constexpr int KS2 = S::H1P / 4;
    constexpr int KC = K / 4;
    constexpr int NPF = (64 * KC + NT - 1) / NT;
    constexpr int HS = S::HS, Z1S = S::Z1S, Z2S = SB::Z2S;
    extern __shared__ __attribute__((aligned(16))) unsigned char din_bwd_smem[];
    SB& sb = *reinterpret_cast<SB*>(din_bwd_smem);
    S& sh = sb.f;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int w = tid >> 6;
    const int r16 = lane & 15;
    const int kk = lane >> 4;
    const float inv_sqrt_k = 1.0f / sqrtf((float)K);

    // ---- the forward's B fragments (as in din_mfma_k) ---------------------------------------------------------------
    DinFrag<K, NC1> fr;
    auto w1parts = [&](int f, int col, float& hd, float& c, float& p) {
        float vh = 0.f, va = 0.f, vd = 0.f, vp = 0.f;
        if (col < H1) {
            vh = W1[(size_t)f * H1 + col];
            va = W1[(size_t)(K + f) * H1 + col];
            vd = W1[(size_t)(2 * K + f) * H1 + col];
            vp = W1[(size_t)(3 * K + f) * H1 + col];
        }
        hd = vh + vd;
        c = va - vd;
        p = vp;
    };
    {
        const int rot = w;
        const int col = 16 * w + r16;
#pragma unroll
        for (int it = 0; it < QN; ++it)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int f = kk * KQ + 4 * ((it + rot) % QN) + e;
                w1parts(f, col, fr.whd[4 * it + e], fr.wc[4 * it + e], fr.wp[4 * it + e]);
            }
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int f = kk * KQ + 4 * (rot % QN) + e;
            w1parts(f, 64 + r16, fr.whd_s[e], fr.wc_s[e], fr.wp_s[e]);
        }
    }
    const float bias1 = (16 * w + r16) < H1 ? b1[16 * w + r16] : 0.f;
    const float bias1_s = (64 + r16) < H1 ? b1[64 + r16] : 0.f;
    float bias2 = 0.f, w3v = 0.f;
#pragma unroll
    for (int s2 = 0; s2 < KS2; ++s2) {
        const int i = kk * KS2 + s2;
        const int col = 16 * w + r16;
        fr.wr2[s2] = (w < NC2 && i < H1 && col < H2) ? W2[(size_t)i * H2 + col] : 0.f;
    }
    if (w < NC2 && (16 * w + r16) < H2) {
        bias2 = b2[16 * w + r16];
        w3v = W3[16 * w + r16];
    }
    const float bias3 = b3[0];
    // ---- backward B fragments: W2^T for dz1 (own H1 tile; wave 3 also the 5th), [Wh+Wd | Wp]^T for dX (own feature tile) ----
    // (LDS-resident, read as 16-byte B fragments per row tile: keeping them in registers as well put the kernel past 512)
    for (int i = tid; i < 2 * K * S::H1P; i += NT) {
        const int x = i / S::H1P, c1 = i - x * S::H1P;
        float v = 0.f;
        if (c1 < H1) v = x < K ? W1[(size_t)x * H1 + c1] + W1[(size_t)(2 * K + x) * H1 + c1] : W1[(size_t)(2 * K + x) * H1 + c1];
        sb.ap[x * Z1S + c1] = v;                         // rows [K, 2K): Wp = W1[3K + f]
    }
    for (int i = tid; i < S::H1P * S::H2P; i += NT) {
        const int c1 = i / S::H2P, c2 = i - c1 * S::H2P;
        sb.w2[c1 * Z2S + c2] = (c1 < H1 && c2 < H2) ? W2[(size_t)c1 * H2 + c2] : 0.f;
    }
    // (the first sample's staging barrier orders these writes before their first readers)
    const int c2e = tid % 48, rge = tid / 48;             // step 3: thread <-> (H2 column, row group), tid < 192
    const float w3c = (tid < 192 && c2e < H2) ? W3[c2e] : 0.f;

    // ---- accumulators that live for the whole launch ------------------------------------------------------------------
    f32x4m gAPh[5], gAPp[5], gW2a[4];
#pragma unroll
    for (int i = 0; i < 5; ++i) gAPh[i] = gAPp[i] = (f32x4m){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int i = 0; i < 4; ++i) gW2a[i] = (f32x4m){0.f, 0.f, 0.f, 0.f};
    float gW3acc = 0.f, gb2acc = 0.f, gb3acc = 0.f;

    // ---- software pipeline: ids two samples ahead, rows (and d out) one ahead ---------------------------------------
    const int64_t G = gridDim.x;
    // Three load stages -- per-sample scalars (length, candidate id, output row) three samples ahead, history ids two ahead, rows
    // one ahead -- ordered so that every consumer runs BEFORE the iteration issues anything new: the loads are conditional, so
    // the compiler can only wait with vmcnt(0), and a younger load in flight would be waited for as well.
    auto load_scalars = [&](int64_t bb, int& len, int64_t& cid, int64_t& base) {
        len = 0;
        cid = -1;
        base = 0;
        if (bb < B) {
            len = hist_len ? min((int)hist_len[bb], T) : T;
            len = max(len, 0);
            cid = cand[bb];
            base = row_off[bb];
        }
    };
    auto load_ids = [&](int64_t bb, int len, int64_t (&ids)[NPF]) {
#pragma unroll
        for (int k = 0; k < NPF; ++k) {
            const int q = tid + k * NT;
            ids[k] = -1;
            if (q < len * KC) ids[k] = hist[bb * T + q / KC];      // len == 0 beyond the batch
        }
    };
    auto load_rows = [&](int64_t bb, int len, int64_t cid, const int64_t (&ids)[NPF], float4 (&h)[NPF], float4& a, int& vmask) {
        vmask = 0;
#pragma unroll
        for (int k = 0; k < NPF; ++k) {
            const int q = tid + k * NT;
            h[k] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (q < len * KC && ids[k] >= 0) {
                h[k] = *reinterpret_cast<const float4*>(table + ids[k] * K + 4 * (q % KC));
                vmask |= 1 << k;
            }
        }
        a = make_float4(0.f, 0.f, 0.f, 0.f);
        if (tid < KC && cid >= 0) a = *reinterpret_cast<const float4*>(table + cid * K + 4 * tid);
        if (tid >= KC && tid < 2 * KC && bb < B) a = *reinterpret_cast<const float4*>(gout + bb * K + 4 * (tid - KC));
    };
    int lenC, lenN, lenNN, vmask;                         // C: the sample whose rows are in hreg; N: ids in idn; NN: the one after
    int64_t cidC, cidN, cidNN, baseC, baseN, baseNN, idn[NPF];
    float4 hreg[NPF], areg4;
    load_scalars(blockIdx.x, lenC, cidC, baseC);
    load_scalars(blockIdx.x + G, lenN, cidN, baseN);
    load_scalars(blockIdx.x + 2 * G, lenNN, cidNN, baseNN);
    load_ids(blockIdx.x, lenC, idn);
    load_rows(blockIdx.x, lenC, cidC, idn, hreg, areg4, vmask);
    load_ids(blockIdx.x + G, lenN, idn);

    // d h rows, d a and S of a sample leave one iteration later (flush): they wait in LDS (the rows in z1's place) and are
    // written after the NEXT sample's loads have been issued, so that no store sits between a load and its wait
    int64_t b_prev = -1, base_prev = 0;
    int len_prev = 0;
    float* ghbuf = sh.z1;                               // [64][Z1S], dead between step 4 and the next sample's layer 1
    auto flush_read = [&](float4 (&fl)[NPF], int (&frk)[NPF], float& fga, float& fs) {
#pragma unroll
        for (int k = 0; k < NPF; ++k) {
            const int q = tid + k * NT;
            const int j = q / KC, c = q - j * KC;
            frk[k] = (b_prev >= 0 && j < len_prev) ? sb.rank[j] : -1;
            fl[k] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (frk[k] >= 0) fl[k] = *reinterpret_cast<const float4*>(ghbuf + j * Z1S + 4 * c);
        }
        fga = (b_prev >= 0 && tid < K) ? sb.gaout[tid] : 0.f;
        fs = 0.f;
        if (b_prev >= 0 && tid < S::H1P)
            fs = tid < 64 ? sb.S[tid] : (sb.S4[tid - 64] + sb.S4[tid - 48]) + (sb.S4[tid - 32] + sb.S4[tid - 16]);
    };
    auto flush_write = [&](const float4 (&fl)[NPF], const int (&frk)[NPF], float fga, float fs) {
        if (b_prev < 0) return;
#pragma unroll
        for (int k = 0; k < NPF; ++k) {
            const int q = tid + k * NT;
            const int c = q % KC;
            if (frk[k] >= 0) *reinterpret_cast<float4*>(gh + (base_prev + frk[k]) * K + 4 * c) = fl[k];
        }
        if (tid < K) ga[b_prev * K + tid] = fga;
        if (tid < H1) Sout[b_prev * H1 + tid] = fs;
    };

    for (int64_t b = blockIdx.x; b < B; b += G) {
        DIN_T(t0);
        const int len = lenC;
        const int RT = (len + 15) >> 4;
        const int64_t base = baseC;
        float4 fl[NPF];
        int frk[NPF];
        float fga, fs;
        flush_read(fl, frk, fga, fs);
        // ---- stage rows (zero-padded to whole row tiles: every later product with a padded row is an exact zero) -------
#pragma unroll
        for (int k = 0; k < NPF; ++k) {
            const int q = tid + k * NT;
            if (q < len * KC) {
                const int j = q / KC, c = q - j * KC;
                *reinterpret_cast<float4*>(sh.uh + j * HS + 4 * c) = hreg[k];
                if (c == 0) sh.valid[j] = (vmask >> k) & 1;
            }
        }
        for (int q = len * KC + tid; q < RT * 16 * KC; q += NT) {
            const int j = q / KC, c = q - j * KC;
            *reinterpret_cast<float4*>(sh.uh + j * HS + 4 * c) = make_float4(0.f, 0.f, 0.f, 0.f);
        }
        for (int j = len + tid; j < 64; j += NT) sh.valid[j] = 0;
        if (tid < KC) *reinterpret_cast<float4*>(sh.av + 4 * tid) = areg4;
        if (tid >= KC && tid < 2 * KC) *reinterpret_cast<float4*>(sb.gv + 4 * (tid - KC)) = areg4;
        __syncthreads();
        DIN_T(t0a);
        lenC = lenN; cidC = cidN; baseC = baseN;          // arrived an iteration ago: nothing younger is in flight yet
        lenN = lenNN; cidN = cidNN; baseN = baseNN;
        load_rows(b + G, lenC, cidC, idn, hreg, areg4, vmask);
        load_ids(b + 2 * G, lenN, idn);
        load_scalars(b + 3 * G, lenNN, cidNN, baseNN);
        flush_write(fl, frk, fga, fs);
        b_prev = b;
        base_prev = base;
        len_prev = len;
        if (RT == 0) {   // block-uniform: nothing reaches the unit; the per-sample outputs are zero
            if (tid < K) sb.gaout[tid] = 0.f;
            if (tid < 64) sb.S[tid] = sb.S4[tid] = 0.f;
            __syncthreads();
            continue;
        }
        DIN_T(t1);
        // ---- dw_j = g . h_j (four threads per row) --------------------------------------------------------------------
        {
            const int row = tid >> 2, qd = tid & 3;
            float v = 0.f;
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                const float4 h4 = *reinterpret_cast<const float4*>(sh.uh + row * HS + qd * 16 + 4 * c);
                const float4 g4 = *reinterpret_cast<const float4*>(sb.gv + qd * 16 + 4 * c);
                v = fmaf(h4.x, g4.x, v); v = fmaf(h4.y, g4.y, v); v = fmaf(h4.z, g4.z, v); v = fmaf(h4.w, g4.w, v);
            }
            v += __shfl_xor(v, 1, 64);
            v += __shfl_xor(v, 2, 64);
            if (qd == 0) sb.dw[row] = v;
        }
        // ---- 1. recompute the unit ----------------------------------------------------------------------------------------
        switch (RT) {
            case 1: din_mlp_tiles<K, NC1, NC2, 1, true>(sh, fr, bias1, bias1_s, bias2, w3v, w, r16, kk, sb.z2); break;
            case 2: din_mlp_tiles<K, NC1, NC2, 2, true>(sh, fr, bias1, bias1_s, bias2, w3v, w, r16, kk, sb.z2); break;
            case 3: din_mlp_tiles<K, NC1, NC2, 3, true>(sh, fr, bias1, bias1_s, bias2, w3v, w, r16, kk, sb.z2); break;
            default: din_mlp_tiles<K, NC1, NC2, 4, true>(sh, fr, bias1, bias1_s, bias2, w3v, w, r16, kk, sb.z2); break;
        }
        __syncthreads();
        DIN_T(t2);
        // ---- 2. weights, d score, compact rows (wave 0) -------------------------------------------------------------------
        if (tid < 64) {
            const bool ok = tid < len && sh.valid[tid];
            float sv = 0.f;
            if (ok) {
                sv = bias3;
#pragma unroll
                for (int c2 = 0; c2 < NC2; ++c2) sv += sh.scp[c2 * 64 + tid];
            }
            const float dwv = ok ? sb.dw[tid] : 0.f;
            float dsv = dwv;
            if (normalize) {
                const float x = sv * inv_sqrt_k;
                const float mx = wave_max_dpp(ok ? x : -INFINITY);
                const float ex = ok ? __expf(x - mx) : 0.f;
                const float sum = wave_sum_dpp(ex);
                sv = ok ? ex / sum : 0.f;
                const float tsum = wave_sum_dpp(sv * dwv);
                dsv = sv * (dwv - tsum) * inv_sqrt_k;
            }
            dsv = ok ? dsv : 0.f;
            sh.sc[tid] = sv;
            sb.ds[tid] = dsv;
            const unsigned long long bal = __ballot(ok);
            sb.rank[tid] = ok ? __popcll(bal & ((1ull << tid) - 1ull)) : -1;
            gb3acc += dsv;
        }
        __syncthreads();
        DIN_T(t3);
        // ---- 3. dpre2 in place of z2; running sums for dW3 and db2 -------------------------------------------------------
        if (tid < 192) {
            for (int row = rge; row < 16 * RT; row += 4) {
                const float z = sb.z2[row * Z2S + c2e];
                const float d = sb.ds[row];
                gW3acc = fmaf(d, z, gW3acc);
                const float dp = d * w3c * z * (1.0f - z);
                gb2acc += dp;
                sb.z2[row * Z2S + c2e] = dp;
            }
        }
        __syncthreads();
        DIN_T(t4);
        // ---- 4. dW2 += z1^T dpre2;  dz1 = dpre2 W2^T -> dpre1 ---------------------------------------------------------------
        // (no branch inside the MFMA loops: a branch makes the compiler shuttle the live accumulators between AGPRs and VGPRs.)
        // Row reductions: k-step s of lane group kk is row 4 kk + s (rows 4 apart sit 16 banks apart at all three strides).
        const int c2w = 16 * (w < 3 ? w : 2) + r16;     // H1 tile 4 x H2 tile w; wave 3 repeats wave 2's (its copy is dropped)
        {
            // dz1 of one (row tile, H1 tile): 12 k-steps over H2, then dpre1 = dz1 z1 (1 - z1) -> dp1, returns the column sum
            auto dz1_tile = [&](int rt, int col) -> float {
                f32x4m acc = (f32x4m){0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int q = 0; q < 3; ++q) {
                    const float4 d4 = *reinterpret_cast<const float4*>(sb.z2 + (rt * 16 + r16) * Z2S + kk * 12 + 4 * q);
                    const float4 w4 = *reinterpret_cast<const float4*>(sb.w2 + col * Z2S + kk * 12 + 4 * q);
                    const float dv[4] = {d4.x, d4.y, d4.z, d4.w}, wt[4] = {w4.x, w4.y, w4.z, w4.w};
#pragma unroll
                    for (int e = 0; e < 4; ++e) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(dv[e], wt[e], acc, 0, 0, 0);
                }
                float cs = 0.f;
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const int row = rt * 16 + 4 * kk + g;
                    const float z = sh.z1[row * Z1S + col];
                    const float d1 = acc[g] * z * (1.0f - z);
                    sb.dp1[row * Z1S + col] = d1;
                    cs += d1;
                }
                return cs;
            };
            float colsum = 0.f;
            for (int rt = 0; rt < RT; ++rt) {   // one block per row tile: the scheduler overlaps one product's LDS reads with the other's MFMAs
#pragma unroll
                for (int s = 0; s < 4; ++s) {
                    const int row = rt * 16 + 4 * kk + s;
                    const float a = sh.z1[row * Z1S + 16 * w + r16];
                    const float a4 = sh.z1[row * Z1S + 64 + r16];
                    const float b0 = sb.z2[row * Z2S + r16], b1v = sb.z2[row * Z2S + 16 + r16], b2v = sb.z2[row * Z2S + 32 + r16];
                    const float bw = sb.z2[row * Z2S + c2w];
                    gW2a[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b0, gW2a[0], 0, 0, 0);
                    gW2a[1] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b1v, gW2a[1], 0, 0, 0);
                    gW2a[2] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b2v, gW2a[2], 0, 0, 0);
                    gW2a[3] = __builtin_amdgcn_mfma_f32_16x16x4f32(a4, bw, gW2a[3], 0, 0, 0);
                }
                colsum += dz1_tile(rt, 16 * w + r16);
            }
            colsum += __shfl_xor(colsum, 16, 64);
            colsum += __shfl_xor(colsum, 32, 64);
            if (kk == 0) sb.S[16 * w + r16] = colsum;
            // the 5th H1 tile: wave w takes row tile w (outside the loop; S4[w] = its column sums, zero when it has none)
            float colsum4 = 0.f;
            if (w < RT) colsum4 = dz1_tile(w, 64 + r16);
            colsum4 += __shfl_xor(colsum4, 16, 64);
            colsum4 += __shfl_xor(colsum4, 32, 64);
            if (kk == 0) sb.S4[w * 16 + r16] = colsum4;
        }
        __syncthreads();
        DIN_T(t5);
        // ---- 5. d[Wh+Wd | Wp] += [h | h*a]^T dpre1;  dX = dpre1 [Wh+Wd | Wp]^T -> d h rows, d a ----------------------------
        const float a_f = sh.av[16 * w + r16], g_f = sb.gv[16 * w + r16];
        {
            float ga_acc = 0.f;
            for (int rt = 0; rt < RT; ++rt) {
                float hv4[4];                       // h[rows 4 kk .. 4 kk + 3][own feature]: A operand here, and the epilogue's h values
#pragma unroll
                for (int s = 0; s < 4; ++s) {
                    const int row = rt * 16 + 4 * kk + s;
                    hv4[s] = sh.uh[row * HS + 16 * w + r16];
                    const float hp = hv4[s] * a_f;
#pragma unroll
                    for (int ni = 0; ni < 5; ++ni) {
                        const float bv = sb.dp1[row * Z1S + 16 * ni + r16];
                        gAPh[ni] = __builtin_amdgcn_mfma_f32_16x16x4f32(hv4[s], bv, gAPh[ni], 0, 0, 0);
                        gAPp[ni] = __builtin_amdgcn_mfma_f32_16x16x4f32(hp, bv, gAPp[ni], 0, 0, 0);
                    }
                }
                f32x4m acch = (f32x4m){0.f, 0.f, 0.f, 0.f}, accp = (f32x4m){0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int q = 0; q < 5; ++q) {
                    const float4 d4 = *reinterpret_cast<const float4*>(sb.dp1 + (rt * 16 + r16) * Z1S + kk * 20 + 4 * q);
                    const float4 h4 = *reinterpret_cast<const float4*>(sb.ap + (16 * w + r16) * Z1S + kk * 20 + 4 * q);
                    const float4 p4 = *reinterpret_cast<const float4*>(sb.ap + (K + 16 * w + r16) * Z1S + kk * 20 + 4 * q);
                    const float dv[4] = {d4.x, d4.y, d4.z, d4.w}, aph[4] = {h4.x, h4.y, h4.z, h4.w}, app[4] = {p4.x, p4.y, p4.z, p4.w};
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        acch = __builtin_amdgcn_mfma_f32_16x16x4f32(dv[e], aph[e], acch, 0, 0, 0);
                        accp = __builtin_amdgcn_mfma_f32_16x16x4f32(dv[e], app[e], accp, 0, 0, 0);
                    }
                }
                const float4 sc4 = *reinterpret_cast<const float4*>(sh.sc + rt * 16 + 4 * kk);
                const float scv[4] = {sc4.x, sc4.y, sc4.z, sc4.w};
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const int row = rt * 16 + 4 * kk + g;
                    ga_acc = fmaf(accp[g], hv4[g], ga_acc);
                    ghbuf[row * Z1S + 16 * w + r16] = fmaf(accp[g], a_f, acch[g]) + scv[g] * g_f;
                }
            }
            ga_acc += __shfl_xor(ga_acc, 16, 64);
            ga_acc += __shfl_xor(ga_acc, 32, 64);
            if (kk == 0) sb.gaout[16 * w + r16] = ga_acc;
        }
        __syncthreads();   // the next sample's staging overwrites uh / av / gv; its flush reads ghbuf / gaout / S
        DIN_T(t6);
        DINB_ACC(0, t0, t0a); DINB_ACC(7, t0a, t1); DINB_ACC(1, t1, t2); DINB_ACC(2, t2, t3); DINB_ACC(3, t3, t4); DINB_ACC(4, t4, t5); DINB_ACC(5, t5, t6);
        DINB_ACC(6, 0ull, 1ull);
    }

    {
        float4 fl[NPF];
        int frk[NPF];
        float fga, fs;
        flush_read(fl, frk, fga, fs);
        flush_write(fl, frk, fga, fs);
    }
    // ---- this workgroup's partial record ------------------------------------------------------------------------------------
    float* rec = partials + (size_t)blockIdx.x * kDinBwdRec;
#pragma unroll
    for (int ni = 0; ni < 5; ++ni)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            rec[(16 * w + 4 * kk + g) * 80 + 16 * ni + r16] = gAPh[ni][g];
            rec[(64 + 16 * w + 4 * kk + g) * 80 + 16 * ni + r16] = gAPp[ni][g];
        }
#pragma unroll
    for (int ni = 0; ni < 3; ++ni)
#pragma unroll
        for (int g = 0; g < 4; ++g) rec[kDinBwdGAP + (16 * w + 4 * kk + g) * 48 + 16 * ni + r16] = gW2a[ni][g];
    if (w < 3) {
#pragma unroll
        for (int g = 0; g < 4; ++g) rec[kDinBwdGAP + (64 + 4 * kk + g) * 48 + 16 * w + r16] = gW2a[3][g];
    }
    if (tid < 192) {
        rec[kDinBwdGAP + kDinBwdGW2 + rge * 48 + c2e] = gb2acc;
        rec[kDinBwdGAP + kDinBwdGW2 + 192 + rge * 48 + c2e] = gW3acc;
    }
    if (tid < 64) rec[kDinBwdGAP + kDinBwdGW2 + 384 + tid] = gb3acc;
}

// sums the per-workgroup records into one record: 16 waves per 64 values, each wave adds a contiguous range of records, the 16
// partial sums meet in LDS in a fixed order
__global__ __launch_bounds__(1024) void din_bwd_sum_k(const float* __restrict__ partials, int nwg, float* __restrict__ red) {
    __shared__ float part[16][64];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int e = blockIdx.x * 64 + lane;
    const int per = (nwg + 15) / 16;
    const int g0 = wv * per, g1 = min(nwg, g0 + per);
    float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
    if (e < kDinBwdRec) {
        const float* p = partials + e;
        int g = g0;
        for (; g + 4 <= g1; g += 4) {
            a0 += p[(size_t)(g + 0) * kDinBwdRec];
            a1 += p[(size_t)(g + 1) * kDinBwdRec];
            a2 += p[(size_t)(g + 2) * kDinBwdRec];
            a3 += p[(size_t)(g + 3) * kDinBwdRec];
        }
        for (; g < g1; ++g) a0 += p[(size_t)g * kDinBwdRec];
    }
    part[wv][lane] = (a0 + a1) + (a2 + a3);
    __syncthreads();
    if (wv == 0 && e < kDinBwdRec) {
        float acc = 0.f;
#pragma unroll
        for (int i = 0; i < 16; ++i) acc += part[i][lane];
        red[e] = acc;
    }
}

// the summed record -> the caller's arrays (drops the padding columns, adds the row-group / lane slots)
__global__ __launch_bounds__(256) void din_bwd_finish_k(const float* __restrict__ red, int K, int H1, int H2,
                                                         float* __restrict__ gAP, float* __restrict__ gW2,
                                                         float* __restrict__ gb2, float* __restrict__ gW3,
                                                         float* __restrict__ gb3) {
    const int n_ap = 2 * K * H1, n_w2 = H1 * H2;
    const int idx = blockIdx.x * 256 + threadIdx.x;
    if (idx >= n_ap + n_w2 + 2 * H2 + 1) return;
    if (idx < n_ap) {
        gAP[idx] = red[(idx / H1) * 80 + idx % H1];
    } else if (idx < n_ap + n_w2) {
        const int i = idx - n_ap;
        gW2[i] = red[kDinBwdGAP + (i / H2) * 48 + i % H2];
    } else if (idx < n_ap + n_w2 + 2 * H2) {
        const int i = idx - n_ap - n_w2;
        const bool w3 = i >= H2;
        const float* p = red + kDinBwdGAP + kDinBwdGW2 + (w3 ? 192 + i - H2 : i);
        (w3 ? gW3 : gb2)[w3 ? i - H2 : i] = (p[0] + p[48]) + (p[96] + p[144]);
    } else {
        const float* p = red + kDinBwdGAP + kDinBwdGW2 + 384;
        float acc = 0.f;
        for (int l = 0; l < 64; ++l) acc += p[l];
        gb3[0] = acc;
    }
}

constexpr int kDinBwdMaxWg = kCUs;   // one workgroup per CU (153 KB of LDS)

}  // namespace dir

using namespace dir;

// PReLU / Dice (inference form) over the rows of a [B, N] activation, in place: the hidden layers of DIN's 200-80 MLP (arXiv:1706.06978
// section 5.3) -- one read and one write instead of the six elementwise library passes of p = sigmoid(scale s + shift),
// f = s (alpha + p (1 - alpha)).  N % 4 == 0, ld % 4 == 0, 16-byte aligned rows.
template <int ACT>
__global__ __launch_bounds__(256) void din_act_rows_k(float* __restrict__ x, int64_t ld, int64_t B, int N, const float* __restrict__ alpha,
                                                      const float* __restrict__ scale, const float* __restrict__ shift) {
    const int nv = N >> 2;
    const int64_t total = B * nv;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int64_t r = i / nv;
        const int c = (int)(i - r * nv) * 4;
        float4 v = *reinterpret_cast<float4*>(x + r * ld + c);
        const float4 a = *reinterpret_cast<const float4*>(alpha + c);
        float s[4] = {v.x, v.y, v.z, v.w};
        const float al[4] = {a.x, a.y, a.z, a.w};
        if (ACT == 1) {
#pragma unroll
            for (int e = 0; e < 4; ++e) s[e] = s[e] > 0.f ? s[e] : al[e] * s[e];
        } else {
            const float4 sc = *reinterpret_cast<const float4*>(scale + c), sh = *reinterpret_cast<const float4*>(shift + c);
            const float scv[4] = {sc.x, sc.y, sc.z, sc.w}, shv[4] = {sh.x, sh.y, sh.z, sh.w};
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float pg = 1.0f / (1.0f + __expf(-(s[e] * scv[e] + shv[e])));
                s[e] = pg * s[e] + (1.0f - pg) * al[e] * s[e];
            }
        }
        *reinterpret_cast<float4*>(x + r * ld + c) = make_float4(s[0], s[1], s[2], s[3]);
    }
}

extern "C" int dir_din_activation_rows_f32(float* x, int64_t ld, int64_t B, int N, int activation, const float* alpha, const float* scale,
                                           const float* shift, dir_stream_t stream) {
    const char* name = "dir_din_activation_rows_f32";
    DIR_CHECK_ARG(B >= 0 && N > 0 && ld >= N, "%s: B=%lld N=%d ld=%lld", name, (long long)B, N, (long long)ld);
    DIR_CHECK_ARG(activation == DIR_DIN_ACT_PRELU || activation == DIR_DIN_ACT_DICE, "%s: activation %d (1 PReLU, 2 Dice)", name, activation);
    if (B == 0) return DIR_OK;
    DIR_CHECK_ARG(x && alpha && (activation == DIR_DIN_ACT_PRELU || (scale && shift)), "%s: null pointer", name);
    if ((N & 3) || (ld & 3) || !aligned16(x) || !aligned16(alpha) || (scale && !aligned16(scale)) || (shift && !aligned16(shift)))
        return fail(DIR_E_UNSUPPORTED, "%s: N and ld must be multiples of 4, x / alpha / scale / shift 16-byte aligned", name);
    const int64_t blocks = (B * (N >> 2) + 255) / 256;
    dim3 grid((unsigned)grid_for(blocks));
    if (activation == DIR_DIN_ACT_PRELU) hipLaunchKernelGGL(din_act_rows_k<1>, grid, dim3(256), 0, as_stream(stream), x, ld, B, N, alpha, scale, shift);
    else hipLaunchKernelGGL(din_act_rows_k<2>, grid, dim3(256), 0, as_stream(stream), x, ld, B, N, alpha, scale, shift);
    DIR_CHECK_LAUNCH(name);
    return DIR_OK;
}

extern "C" int dir_din_attention_pool_act_f32(const float* table, int K, const int64_t* hist, const int32_t* hist_len, const int64_t* cand,
                                              int T, const float* W1, const float* b1, int H1, const float* W2, const float* b2, int H2,
                                              const float* W3, const float* b3, int normalize, int activation, const float* act_params,
                                              int64_t B, float* out, float* scores, dir_stream_t stream) {
    const char* name = "dir_din_attention_pool_act_f32";
    if (activation == DIR_DIN_ACT_SIGMOID)
        return dir_din_attention_pool_f32(table, K, hist, hist_len, cand, T, W1, b1, H1, W2, b2, H2, W3, b3, normalize, B, out, scores, stream);
    DIR_CHECK_ARG(K > 0 && T > 0 && H1 > 0 && H2 > 0 && B >= 0, "%s: K=%d T=%d H1=%d H2=%d", name, K, T, H1, H2);
    DIR_CHECK_ARG(activation == DIR_DIN_ACT_PRELU || activation == DIR_DIN_ACT_DICE, "%s: activation %d (0 sigmoid, 1 PReLU, 2 Dice)", name, activation);
    if (B == 0) return DIR_OK;
    DIR_CHECK_ARG(table && hist && cand && W1 && b1 && W2 && b2 && W3 && b3 && out && act_params, "%s: null pointer", name);
    if ((H1 & 3) || (H2 & 3) || !din_wave_covers(K, T, H1, H2))
        return fail(DIR_E_UNSUPPORTED, "%s: the PReLU / Dice unit covers K = 64, H1 <= 80, H2 <= 48 (multiples of 4), T <= 64 (K=%d T=%d H1=%d H2=%d)", name, K,
                    T, H1, H2);
    if (!aligned16(table) || !aligned16(W1) || !aligned16(W2) || !aligned16(b1) || !aligned16(b2))
        return fail(DIR_E_BADARG, "%s: table / W1 / W2 / b1 / b2 must be 16-byte aligned", name);
    const int rc = launch_din_wave(as_stream(stream), table, hist, hist_len, cand, T, W1, b1, H1, W2, b2, H2, W3, b3, normalize, B, out, scores, nullptr,
                                   nullptr, activation, act_params);
    if (rc != DIR_OK) return rc;
    DIR_CHECK_LAUNCH(name);
    return DIR_OK;
}

extern "C" int dir_din_attention_pool_f32(const float* table, int K, const int64_t* hist, const int32_t* hist_len,
                                          const int64_t* cand, int T, const float* W1, const float* b1, int H1,
                                          const float* W2, const float* b2, int H2, const float* W3,
                                          const float* b3, int normalize, int64_t B, float* out, float* scores,
                                          dir_stream_t stream) {
    DIR_CHECK_ARG(K > 0 && T > 0 && H1 > 0 && H2 > 0 && B >= 0, "dir_din_attention_pool_f32: K=%d T=%d H1=%d H2=%d", K, T, H1, H2);
    if (B == 0) return DIR_OK;
    DIR_CHECK_ARG(table && hist && cand && W1 && b1 && W2 && b2 && W3 && b3 && out, "dir_din_attention_pool_f32: null pointer");
    if ((K & 3) || (H1 & 3) || (H2 & 3))
        return fail(DIR_E_UNSUPPORTED, "dir_din_attention_pool_f32: K, H1, H2 must be multiples of 4 (K=%d H1=%d H2=%d)", K, H1, H2);
    if (!aligned16(table) || !aligned16(W1) || !aligned16(W2) || !aligned16(b1) || !aligned16(b2))
        return fail(DIR_E_BADARG, "dir_din_attention_pool_f32: table / W1 / W2 / b1 / b2 must be 16-byte aligned");
    static const int mfma_env = getenv("DIR_DIN_MFMA") ? atoi(getenv("DIR_DIN_MFMA")) : 1;
    static const int wave_env = getenv("DIR_DIN_WAVE") ? atoi(getenv("DIR_DIN_WAVE")) : 1;     // 0: the round-1 workgroup-per-sample kernel (A/B)
    if (mfma_env && wave_env && din_wave_covers(K, T, H1, H2)) {
        const int rc = launch_din_wave(as_stream(stream), table, hist, hist_len, cand, T, W1, b1, H1, W2, b2, H2, W3, b3, normalize, B, out, scores);
        if (rc != DIR_OK) return rc;
        DIR_CHECK_LAUNCH("din_attention_pool(wave)");
        return DIR_OK;
    }
    if (mfma_env && T <= 64) {
        const int nc1 = (H1 + 15) / 16, nc2 = (H2 + 15) / 16;
        hipStream_t st = as_stream(stream);
        bool done = true;
        // narrower hidden layers run on the same instantiations: columns >= H1 / H2 are zero weights (guards in the loads)
        if (K == 64 && nc1 <= 5 && nc2 <= 3)
            launch_din_mfma<64, 5, 3>(st, table, hist, hist_len, cand, T, W1, b1, H1, W2, b2, H2, W3, b3, normalize, B, out, scores);
        else if (K == 16 && nc1 <= 2 && nc2 <= 1)
            launch_din_mfma<16, 2, 1>(st, table, hist, hist_len, cand, T, W1, b1, H1, W2, b2, H2, W3, b3, normalize, B, out, scores);
        else if (K == 32 && nc1 <= 3 && nc2 <= 1)
            launch_din_mfma<32, 3, 1>(st, table, hist, hist_len, cand, T, W1, b1, H1, W2, b2, H2, W3, b3, normalize, B, out, scores);
        else
            done = false;
        if (done) {
            DIR_CHECK_LAUNCH("din_attention_pool(mfma)");
            return DIR_OK;
        }
    }
    DinDims dm{K, T, H1, H2, 4 * K + 4, H1 + 4, H2 + 4};
    const size_t shmem = sizeof(float) * ((size_t)T * (dm.us + dm.z1s + dm.z2s) + ((T + 63) & ~63)) + sizeof(int) * (size_t)T;
    if (shmem > 160 * 1024) return fail(DIR_E_UNSUPPORTED, "dir_din_attention_pool_f32: T=%d K=%d needs %zu B of LDS (> 160 KiB)", T, K, shmem);
    static LdsOnce once;
    if (!lds_limit(once, 160 * 1024, &din_k<5>)) return fail(DIR_E_HIP, "dir_din_attention_pool_f32: cannot reserve 160 KiB of LDS");
    const int per_cu = (int)((160 * 1024) / shmem) < 1 ? 1 : (int)((160 * 1024) / shmem);
    dim3 grid((unsigned)(B < (int64_t)kCUs * per_cu * 4 ? B : (int64_t)kCUs * per_cu * 4));
    hipLaunchKernelGGL((din_k<5>), grid, dim3(256), shmem, as_stream(stream), table, dm, hist, hist_len, cand, W1, b1, W2, b2, W3, b3, normalize, B, out, scores);
    DIR_CHECK_LAUNCH("din_attention_pool");
    return DIR_OK;
}

extern "C" int64_t dir_din_backward_rows_workspace_bytes(int K, int H1, int H2, int64_t n_tiles);

extern "C" int dir_din_attention_pool_save_f32(const float* table, int K, const int64_t* hist, const int32_t* hist_len, const int64_t* cand,
                                               int T, const float* W1, const float* b1, int H1, const float* W2, const float* b2, int H2,
                                               const float* W3, const float* b3, int normalize, int64_t B, float* out, float* scores,
                                               const int64_t* tile_off, int64_t n_tiles, void* workspace, int64_t workspace_bytes,
                                               dir_stream_t stream) {
    const char* name = "dir_din_attention_pool_save_f32";
    DIR_CHECK_ARG(K > 0 && T > 0 && H1 > 0 && H2 > 0 && B >= 0 && n_tiles >= 0, "%s: K=%d T=%d H1=%d H2=%d", name, K, T, H1, H2);
    if ((H1 & 3) || (H2 & 3) || !din_wave_covers(K, T, H1, H2))
        return fail(DIR_E_UNSUPPORTED, "%s: covers K = 64, H1 <= 80, H2 <= 48 (multiples of 4), T <= 64 (K=%d H1=%d H2=%d T=%d)", name, K, H1, H2, T);
    if (B == 0) return DIR_OK;
    DIR_CHECK_ARG(table && hist && cand && W1 && b1 && W2 && b2 && W3 && b3 && out && scores && tile_off && workspace, "%s: null pointer", name);
    if (!aligned16(table) || !aligned16(W1) || !aligned16(W2) || !aligned16(b1) || !aligned16(b2) || !aligned16(workspace))
        return fail(DIR_E_BADARG, "%s: table / W1 / W2 / b1 / b2 / workspace must be 16-byte aligned", name);
    const int64_t need = dir_din_backward_rows_workspace_bytes(K, H1, H2, n_tiles);
    if (workspace_bytes < need) return fail(DIR_E_BADARG, "%s: workspace needs %lld bytes", name, (long long)need);
    const int rc = launch_din_wave(as_stream(stream), table, hist, hist_len, cand, T, W1, b1, H1, W2, b2, H2, W3, b3, normalize, B, out, scores,
                                   tile_off, static_cast<float*>(workspace));
    if (rc != DIR_OK) return rc;
    DIR_CHECK_LAUNCH(name);
    return DIR_OK;
}

// ---- the same entries with the arithmetic of the two MFMA layers given by ARGUMENT (round 5): DIR_DIN_ARITH_F32 (fp32 MFMA: any
// magnitudes), DIR_DIN_ARITH_BF16X3 (three bf16 pieces: fp32's exponent range), DIR_DIN_ARITH_F16X2 (two fp16 pieces, UNSCALED: table rows,
// their products with the candidate row and the weights inside fp16's range and not far below 1), DIR_DIN_ARITH_DEFAULT (-1: the
// DIR_DIN_ARITH environment switch, fp16 x 2 by default).  The Python surface picks bf16 x 3 when the table or a weight leaves the fp16 x 2
// window (ops.din_attention_pool: measured per version, not assumed).
namespace dir { extern thread_local int dw_arith_override; }
namespace {
struct DwArithScope {
    int prev;
    explicit DwArithScope(int a) : prev(dir::dw_arith_override) { dir::dw_arith_override = a; }
    ~DwArithScope() { dir::dw_arith_override = prev; }
};
}  // namespace

extern "C" int dir_din_attention_pool_arith_f32(const float* table, int K, const int64_t* hist, const int32_t* hist_len, const int64_t* cand,
                                                int T, const float* W1, const float* b1, int H1, const float* W2, const float* b2, int H2,
                                                const float* W3, const float* b3, int normalize, int activation, const float* act_params,
                                                int arith, int64_t B, float* out, float* scores, dir_stream_t stream) {
    DIR_CHECK_ARG(arith >= -1 && arith <= 2, "dir_din_attention_pool_arith_f32: arith %d (-1 default, 0 f32, 1 bf16x3, 2 f16x2)", arith);
    DwArithScope scope(arith);
    return dir_din_attention_pool_act_f32(table, K, hist, hist_len, cand, T, W1, b1, H1, W2, b2, H2, W3, b3, normalize, activation, act_params, B,
                                          out, scores, stream);
}

extern "C" int dir_din_attention_pool_save_arith_f32(const float* table, int K, const int64_t* hist, const int32_t* hist_len, const int64_t* cand,
                                                     int T, const float* W1, const float* b1, int H1, const float* W2, const float* b2, int H2,
                                                     const float* W3, const float* b3, int normalize, int arith, int64_t B, float* out,
                                                     float* scores, const int64_t* tile_off, int64_t n_tiles, void* workspace,
                                                     int64_t workspace_bytes, dir_stream_t stream) {
    DIR_CHECK_ARG(arith >= -1 && arith <= 2, "dir_din_attention_pool_save_arith_f32: arith %d (-1 default, 0 f32, 1 bf16x3, 2 f16x2)", arith);
    DwArithScope scope(arith);
    return dir_din_attention_pool_save_f32(table, K, hist, hist_len, cand, T, W1, b1, H1, W2, b2, H2, W3, b3, normalize, B, out, scores, tile_off,
                                           n_tiles, workspace, workspace_bytes, stream);
}

extern "C" int64_t dir_din_backward_workspace_bytes(int K, int H1, int H2) {
    if (K != 64 || H1 <= 0 || H2 <= 0 || H1 > 80 || H2 > 48) return 0;
    return (int64_t)(kDinBwdMaxWg + 1) * kDinBwdRec * (int64_t)sizeof(float);   // one record per workgroup + their sum
}

extern "C" int dir_din_attention_pool_backward_f32(const float* table, int K, const int64_t* hist, const int32_t* hist_len,
                                                   const int64_t* cand, int T, const float* W1, const float* b1, int H1,
                                                   const float* W2, const float* b2, int H2, const float* W3, const float* b3,
                                                   int normalize, int64_t B, const float* gout, const int64_t* row_off,
                                                   float* gh, float* ga, float* S, float* gAP, float* gW2, float* gb2,
                                                   float* gW3, float* gb3, void* workspace, dir_stream_t stream) {
    const char* name = "dir_din_attention_pool_backward_f32";
    DIR_CHECK_ARG(K > 0 && T > 0 && H1 > 0 && H2 > 0 && B >= 0, "%s: K=%d T=%d H1=%d H2=%d", name, K, T, H1, H2);
    if (K != 64 || H1 > 80 || H2 > 48 || (H1 & 3) || (H2 & 3) || T > 64)
        return fail(DIR_E_UNSUPPORTED, "%s: the fused backward covers K = 64, H1 <= 80, H2 <= 48 (multiples of 4), T <= 64 (K=%d H1=%d H2=%d T=%d)",
                    name, K, H1, H2, T);
    DIR_CHECK_ARG(table && hist && cand && W1 && b1 && W2 && b2 && W3 && b3 && gAP && gW2 && gb2 && gW3 && gb3 && workspace,
                  "%s: null pointer", name);
    DIR_CHECK_ARG(B == 0 || (gout && row_off && ga && S), "%s: null pointer", name);   // gh may be NULL when there is no valid row
    if (!aligned16(table) || !aligned16(gout))
        return fail(DIR_E_BADARG, "%s: table / gout must be 16-byte aligned", name);
    hipStream_t st = as_stream(stream);
    static LdsOnce once;
    const size_t shmem = sizeof(DinBwdSh<64, 5, 3>);
    if (!lds_limit(once, (int)shmem, &din_bwd_k<64, 5, 3>)) return fail(DIR_E_HIP, "%s: cannot reserve %zu B of LDS", name, shmem);
    int nwg = (int)(B < kDinBwdMaxWg ? B : kDinBwdMaxWg);
    float* partials = static_cast<float*>(workspace);
    if (nwg > 0) {
        hipLaunchKernelGGL((din_bwd_k<64, 5, 3>), dim3((unsigned)nwg), dim3(256), shmem, st, table, hist, hist_len, cand, T, W1, b1, H1,
                           W2, b2, H2, W3, b3, normalize, B, gout, row_off, gh, ga, S, partials);
        DIR_CHECK_LAUNCH(name);
    }
    float* red = partials + (size_t)kDinBwdMaxWg * kDinBwdRec;
    hipLaunchKernelGGL(din_bwd_sum_k, dim3((unsigned)((kDinBwdRec + 63) / 64)), dim3(1024), 0, st, partials, nwg, red);
    const int nout = 2 * K * H1 + H1 * H2 + 2 * H2 + 1;
    hipLaunchKernelGGL(din_bwd_finish_k, dim3((unsigned)((nout + 255) / 256)), dim3(256), 0, st, red, K, H1, H2, gAP, gW2, gb2, gW3, gb3);
    DIR_CHECK_LAUNCH(name);
    return DIR_OK;
}
