// din_bwd_rows.hip -- DIN local activation unit + pooling, BACKWARD, for the (K = 64, H1 <= 80, H2 <= 48, T <= 64) shape class, as two
// kernels.  NO REFERENCE CODE (README.md:27 links arXiv:1706.06978); the definition is include/dir_hip.h (A13 backward) and
// oracle/dir_oracle.c.
//
// The round-1 backward (din.hip: din_bwd_k) does everything per sample in one workgroup: seven barriers per sample, single-wave
// softmax phases, and 56 accumulator registers per lane for the weight gradients that pin it at one wave per SIMD.  Here the work is
// split by what it reduces over:
//
//   din_rows_k   ONE WAVE OWNS ONE SAMPLE (the structure of the forward, din_wave.hip): everything that is per history row or per
//                sample -- the recompute of z1 / z2, d score from the saved attention weights, dpre2, dz1 -> dpre1, dX -> the rows'
//                gradients, d candidate, S -- with all four GEMMs chained in registers in the transposed layout
//                  pre1^T = W^T h^T      ->  z1^T (C layout) = B operand of   pre2^T = W2^T z1^T
//                  dpre2^T (C layout)    =  B operand of  dz1^T = W2 dpre2^T  ->  dpre1^T = B operand of  dX^T = [Wh+Wd | Wp] dpre1^T
//                (lane (kk, r) holds hidden units / features {16 t + 4 kk + g} of history row r; weights are A operands read from LDS
//                with one ds_read_b128 per four MFMA steps).  No barrier in the sample loop, no cross-sample reduction -> samples
//                are drawn from a device-side queue.  Per 16-row tile it leaves z1 | dpre1 | z2 | d score in a scratch record.
//   din_wgrad_k  everything that reduces over ALL rows of the batch -- d[Wh+Wd | Wp] += [h | h*a]^T dpre1, dW2 += z1^T dpre2, db2,
//                dW3, db3 -- as a streaming pass over the scratch tiles: four waves share a tile through LDS, each owning a quarter
//                of the outputs in registers (din_bwd_k's step 4/5 row reductions), static sample assignment and one partial
//                record per workgroup (summed in a fixed order by din.hip's din_bwd_sum_k / din_bwd_finish_k): bitwise reproducible.
//
// The attention weights of the forward (dir_din_attention_pool_f32's `scores` output) are an INPUT: the softmax is not recomputed.
#include <atomic>
#include <cstring>

#include "common.hpp"

namespace dir {

typedef float f32x4r __attribute__((ext_vector_type(4)));

// defined in din.hip
__global__ void din_bwd_sum_k(const float* __restrict__ partials, int nwg, float* __restrict__ red);
__global__ void din_bwd_finish_k(const float* __restrict__ red, int K, int H1, int H2, float* __restrict__ gAP, float* __restrict__ gW2,
                                 float* __restrict__ gb2, float* __restrict__ gW3, float* __restrict__ gb3);

constexpr int RB_K = 64, RB_H1P = 80, RB_H2P = 48;
constexpr int RB_WS = RB_K + 4;         // row stride of the [hidden][feature] forward images
constexpr int RB_W2S = RB_H1P + 4;      // [h2][hidden] forward image
constexpr int RB_W2BS = RB_H2P + 4;     // [hidden][h2] backward image (A of dz1^T)
constexpr int RB_WCS = RB_H1P + 4;      // [feature 2K][hidden] backward image (A of dX^T)
constexpr int RB_WAVES = 8;
constexpr int RB_GROUPS = 8, RB_SLOTS = 64, RB_CH = 2;
constexpr int RB_ROW = kDinRecRow;       // floats per history row's record (common.hpp): z1 [80] | dpre1 [80] | z2 [48] | d score | 3 pad
constexpr int RB_Z1 = kDinRecZ1, RB_DP1 = kDinRecDp1, RB_Z2 = kDinRecZ2, RB_DS = kDinRecDs;
constexpr int kRowsBwdGAP = 2 * 64 * 80, kRowsBwdGW2 = 80 * 48;
constexpr int kRowsBwdRec = kRowsBwdGAP + kRowsBwdGW2 + 4 * 48 + 4 * 48 + 64;      // == din.hip's kDinBwdRec (checked by the host code)

struct DinRowsSh {
    static constexpr bool kBf3 = false;
    float whd[RB_H1P * RB_WS];          // (Wh + Wd)^T  x -log2 e
    float wp[RB_H1P * RB_WS];           // Wp^T         x -log2 e
    float wc[RB_H1P * RB_WS];           // (Wa - Wd)^T  x -log2 e
    float w2[RB_H2P * RB_W2S];          // W2^T         x -log2 e
    float w2b[RB_H1P * RB_W2BS];        // W2            (backward)
    float wcat[2 * RB_K * RB_WCS];      // [Wh + Wd ; Wp] as [feature][hidden]  (backward)
    float b1[RB_H1P], b2[RB_H2P], w3[RB_H2P];
    float cvec[RB_WAVES][RB_H1P];
    float av[RB_WAVES][RB_K];           // candidate row of the wave's sample
    float gv[RB_WAVES][RB_K];           // d out of the wave's sample
};

// The row pass on saved activations with its two GEMMs (dz1^T = W2 dpre2^T, dX^T = [Wh+Wd ; Wp] dpre1^T) on the bf16 x 3 recipe of
// din_wave.hip: only the two backward images, split into three bf16 pieces in MFMA A-operand order [k-step][m tile][piece][lane][8]
// (element j of lane (kk, r) in k-step ks = A[16 mtile + r][16 (2 ks + (j >> 2)) + 4 kk + (j & 3)]) -- 102 KB; the B operands are the
// lane's own accumulator tiles 2 ks, 2 ks + 1 of the GEMM before (dpre2, dpre1), as in the forward.
// This file is compiled without packed fp32 VALU instructions (build.py; the gfx950 hazard of isa_check.py).
struct DinRowsSh3 {
    static constexpr bool kBf3 = true;
    unsigned int w2b3[2 * 5 * 3 * 256];     // W2 [hidden 80 -> 5 tiles][h2 48 -> 2 k-steps of 32]
    unsigned int wcat3[3 * 8 * 3 * 256];    // [Wh + Wd ; Wp] [feature 128 -> 8 tiles][hidden 80 -> 3 k-steps of 32]
    float w3[RB_H2P];
    float av[RB_WAVES][RB_K];
    float gv[RB_WAVES][RB_K];
};

typedef __bf16 rb_bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned int rb_u32x4 __attribute__((ext_vector_type(4)));
#define RB_MFMA3(a, b, c) __builtin_amdgcn_mfma_f32_16x16x32_bf16((a), (b), (c), 0, 0, 0)
__device__ __forceinline__ unsigned int rb_pk(float a, float b) {      // v_cvt_pk_bf16_f32 (round to nearest even), a in the low half
    typedef __bf16 pk2_t __attribute__((ext_vector_type(2)));
    const pk2_t v = {(__bf16)a, (__bf16)b};
    unsigned int w = __builtin_bit_cast(unsigned int, v);
    asm("" : "+v"(w));
    return w;
}
__device__ __forceinline__ void rb_split_pair(float a, float b, unsigned int& w0, unsigned int& w1, unsigned int& w2) {
    w0 = rb_pk(a, b);
    const float ra = a - __builtin_bit_cast(float, w0 << 16), rb = b - __builtin_bit_cast(float, w0 & 0xffff0000u);
    w1 = rb_pk(ra, rb);
    const float sa = ra - __builtin_bit_cast(float, w1 << 16), sb = rb - __builtin_bit_cast(float, w1 & 0xffff0000u);
    w2 = rb_pk(sa, sb);
}
typedef float f32x4r_ __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void rb_split8(const f32x4r_ s0, const f32x4r_ s1, rb_bf16x8 (&x)[3]) {
    unsigned int w[3][4];
    rb_split_pair(s0[0], s0[1], w[0][0], w[1][0], w[2][0]);
    rb_split_pair(s0[2], s0[3], w[0][1], w[1][1], w[2][1]);
    rb_split_pair(s1[0], s1[1], w[0][2], w[1][2], w[2][2]);
    rb_split_pair(s1[2], s1[3], w[0][3], w[1][3], w[2][3]);
#pragma unroll
    for (int pc = 0; pc < 3; ++pc) x[pc] = __builtin_bit_cast(rb_bf16x8, (rb_u32x4){w[pc][0], w[pc][1], w[pc][2], w[pc][3]});
}
__device__ __forceinline__ f32x4r_ rb_mfma6(const rb_bf16x8 (&a)[3], const rb_bf16x8 (&x)[3], f32x4r_ c) {
    c = RB_MFMA3(a[0], x[2], c);
    c = RB_MFMA3(a[2], x[0], c);
    c = RB_MFMA3(a[1], x[1], c);
    c = RB_MFMA3(a[0], x[1], c);
    c = RB_MFMA3(a[1], x[0], c);
    c = RB_MFMA3(a[0], x[0], c);
    return c;
}
__device__ __forceinline__ void rb_lda(const unsigned int* img, int tile, int lane4, rb_bf16x8 (&a)[3]) {
#pragma unroll
    for (int pc = 0; pc < 3; ++pc) a[pc] = __builtin_bit_cast(rb_bf16x8, *reinterpret_cast<const rb_u32x4*>(img + (tile * 3 + pc) * 256 + lane4));
}

__device__ unsigned int rb_queue[RB_SLOTS][16];

constexpr float RB_NLOG2E = -1.4426950408889634f;
__device__ __forceinline__ float rb_sigmoid_pre(float y) { return __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(y)); }
__device__ __forceinline__ float4 rb_ld4(const float* p) { return *reinterpret_cast<const float4*>(p); }
__device__ __forceinline__ float rb_dot4(float4 a, float4 b, float acc) {
    acc = fmaf(a.x, b.x, acc); acc = fmaf(a.y, b.y, acc); acc = fmaf(a.z, b.z, acc); acc = fmaf(a.w, b.w, acc);
    return acc;
}
#define RB_MFMA(a, b, c) __builtin_amdgcn_mfma_f32_16x16x4f32((a), (b), (c), 0, 0, 0)
// keeps the scheduler from hoisting a later block's LDS operand reads over this one (register pressure: 2 waves per SIMD = 256 VGPRs)
#define RB_SCHED_FENCE() __builtin_amdgcn_sched_barrier(0)
__device__ __forceinline__ float rb_c(const float4& v, int e) { return e == 0 ? v.x : e == 1 ? v.y : e == 2 ? v.z : v.w; }

// the forward's sample queue (din_wave.hip: DwQueue), on its own records
struct RbQueue {
    unsigned int* q;
    long long lo, hi, next;
    int left;
    bool dead;
    unsigned int ticket;
    long long stride_next, stride;
    __device__ __forceinline__ void issue() {
        ticket = 0;
        if ((threadIdx.x & 63) == 0) ticket = atomicAdd(q, (unsigned int)RB_CH);
    }
    __device__ __forceinline__ void init(unsigned int* qrec, long long B) {
        const int G = (int)gridDim.x < RB_GROUPS ? (int)gridDim.x : RB_GROUPS;
        const int g = (int)(blockIdx.x % (unsigned)G);
        q = qrec ? qrec + g : nullptr;
        lo = (long long)g * B / G;
        hi = (long long)(g + 1) * B / G;
        left = 0;
        dead = false;
        next = 0;
        stride = (long long)gridDim.x * RB_WAVES;
        stride_next = (long long)blockIdx.x * RB_WAVES + (threadIdx.x >> 6);
        hi = q ? hi : B;
        if (q) issue();
    }
    __device__ __forceinline__ long long take() {
        if (!q) {
            const long long bb = stride_next;
            stride_next += stride;
            return bb < hi ? bb : -1;
        }
        if (left == 0) {
            if (dead) return -1;
            const long long base = lo + (long long)(unsigned int)__builtin_amdgcn_readfirstlane((int)ticket);
            if (base >= hi) {
                dead = true;
                return -1;
            }
            next = base;
            left = (int)((hi - base) < RB_CH ? (hi - base) : RB_CH);
            issue();
        }
        --left;
        return next++;
    }
    __device__ __forceinline__ void drain() {
        if (q && !dead) (void)__builtin_amdgcn_readfirstlane((int)ticket);
    }
};

__device__ __forceinline__ void rb_load_row(const float* __restrict__ table, const int kk, const long long id, float4 (&hv)[4]) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        hv[i] = make_float4(0.f, 0.f, 0.f, 0.f);
        if (id >= 0) hv[i] = rb_ld4(table + id * RB_K + 16 * i + 4 * kk);
    }
}

// SAVED: the records already hold z1 and z2 (written by the training forward, dir_din_attention_pool_save_f32): nothing of the forward is
// recomputed -- half of the MFMAs, no per-sample term, no forward weight images.
template <bool SAVED, typename Sh>
__global__ __launch_bounds__(64 * RB_WAVES, 2) void din_rows_k(const float* __restrict__ table, const int64_t* __restrict__ hist,
                                                               const int32_t* __restrict__ hist_len, const int64_t* __restrict__ cand,
                                                               int T, const float* __restrict__ W1, const float* __restrict__ b1, int H1,
                                                               const float* __restrict__ W2, const float* __restrict__ b2, int H2,
                                                               const float* __restrict__ W3, int normalize, long long B,
                                                               const float* __restrict__ gout, const float* __restrict__ scores,
                                                               const int64_t* __restrict__ row_off, const int64_t* __restrict__ tile_off,
                                                               float* __restrict__ gh, float* __restrict__ ga, float* __restrict__ Sout,
                                                               float* __restrict__ scratch, int slot) {
    extern __shared__ __attribute__((aligned(16))) unsigned char rb_smem[];
    static_assert(SAVED || !Sh::kBf3, "the bf16x3 row pass has no forward images: saved activations only");
    Sh& sh = *reinterpret_cast<Sh*>(rb_smem);
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, r16 = lane & 15, kk = lane >> 4;
    // ---- weight images, once per workgroup ---------------------------------------------------------------------------------------------------
    if constexpr (Sh::kBf3) {
        for (int idx = tid; idx < 2 * 5 * 64 * 4; idx += 64 * RB_WAVES) {          // W2 as A[m = hidden][k = h2]: dword jp of lane l of tile t = (ks, mt)
            const int jp = idx & 3, l = (idx >> 2) & 63, t = idx >> 8;
            const int ks = t / 5, mt = t - 5 * ks;
            const int hid = 16 * mt + (l & 15);
            const int h2 = 16 * (2 * ks + (jp >> 1)) + 4 * (l >> 4) + 2 * (jp & 1);
            float v[2];
#pragma unroll
            for (int e = 0; e < 2; ++e) v[e] = (hid < H1 && h2 + e < H2) ? W2[(size_t)hid * H2 + h2 + e] : 0.f;
            unsigned int q0, q1, q2;
            rb_split_pair(v[0], v[1], q0, q1, q2);
            sh.w2b3[(t * 3 + 0) * 256 + l * 4 + jp] = q0;
            sh.w2b3[(t * 3 + 1) * 256 + l * 4 + jp] = q1;
            sh.w2b3[(t * 3 + 2) * 256 + l * 4 + jp] = q2;
        }
        for (int idx = tid; idx < 3 * 8 * 64 * 4; idx += 64 * RB_WAVES) {          // [Wh + Wd ; Wp] as A[m = feature][k = hidden], t = (ks, ft)
            const int jp = idx & 3, l = (idx >> 2) & 63, t = idx >> 8;
            const int ks = t / 8, ft = t - 8 * ks;
            const int feat = 16 * ft + (l & 15);
            const int hid = 16 * (2 * ks + (jp >> 1)) + 4 * (l >> 4) + 2 * (jp & 1);
            float v[2];
#pragma unroll
            for (int e = 0; e < 2; ++e) {
                v[e] = 0.f;
                if (hid + e < H1)
                    v[e] = feat < RB_K ? W1[(size_t)feat * H1 + hid + e] + W1[(size_t)(2 * RB_K + feat) * H1 + hid + e]
                                       : W1[(size_t)(3 * RB_K + feat - RB_K) * H1 + hid + e];
            }
            unsigned int q0, q1, q2;
            rb_split_pair(v[0], v[1], q0, q1, q2);
            sh.wcat3[(t * 3 + 0) * 256 + l * 4 + jp] = q0;
            sh.wcat3[(t * 3 + 1) * 256 + l * 4 + jp] = q1;
            sh.wcat3[(t * 3 + 2) * 256 + l * 4 + jp] = q2;
        }
        for (int idx = tid; idx < RB_H2P; idx += 64 * RB_WAVES) sh.w3[idx] = idx < H2 ? W3[idx] : 0.f;
    } else {
#pragma unroll 2
        for (int idx = tid; idx < (RB_H1P / 4) * RB_K; idx += 64 * RB_WAVES) {
            const int f = idx / (RB_H1P / 4), m = 4 * (idx - f * (RB_H1P / 4));
            float4 vh = make_float4(0.f, 0.f, 0.f, 0.f), va = vh, vd = vh, vp = vh;
            if (m < H1) {
                vh = rb_ld4(W1 + (size_t)f * H1 + m);
                vd = rb_ld4(W1 + (size_t)(2 * RB_K + f) * H1 + m);
                vp = rb_ld4(W1 + (size_t)(3 * RB_K + f) * H1 + m);
                if (!SAVED) va = rb_ld4(W1 + (size_t)(RB_K + f) * H1 + m);
            }
            const float h4[4] = {vh.x + vd.x, vh.y + vd.y, vh.z + vd.z, vh.w + vd.w};
            const float p4[4] = {vp.x, vp.y, vp.z, vp.w};
            if (!SAVED) {
                const float c4[4] = {va.x - vd.x, va.y - vd.y, va.z - vd.z, va.w - vd.w};
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    sh.whd[(m + e) * RB_WS + f] = h4[e] * RB_NLOG2E;
                    sh.wp[(m + e) * RB_WS + f] = p4[e] * RB_NLOG2E;
                    sh.wc[(m + e) * RB_WS + f] = c4[e] * RB_NLOG2E;
                }
            }
            *reinterpret_cast<float4*>(&sh.wcat[f * RB_WCS + m]) = make_float4(h4[0], h4[1], h4[2], h4[3]);
            *reinterpret_cast<float4*>(&sh.wcat[(RB_K + f) * RB_WCS + m]) = make_float4(p4[0], p4[1], p4[2], p4[3]);
        }
        for (int idx = tid; idx < RB_H2P * RB_H1P; idx += 64 * RB_WAVES) {
            const int hid = idx / RB_H2P, h2 = idx - hid * RB_H2P;
            const float v = (hid < H1 && h2 < H2) ? W2[(size_t)hid * H2 + h2] : 0.f;
            if (!SAVED) sh.w2[h2 * RB_W2S + hid] = v * RB_NLOG2E;
            sh.w2b[hid * RB_W2BS + h2] = v;
        }
        if (!SAVED) {
            for (int idx = tid; idx < RB_H1P; idx += 64 * RB_WAVES) sh.b1[idx] = idx < H1 ? b1[idx] * RB_NLOG2E : 0.f;
        }
        for (int idx = tid; idx < RB_H2P; idx += 64 * RB_WAVES) {
            if (!SAVED) sh.b2[idx] = idx < H2 ? b2[idx] * RB_NLOG2E : 0.f;
            sh.w3[idx] = idx < H2 ? W3[idx] : 0.f;
        }
    }
    __syncthreads();
    const float inv_sqrt_k = 1.0f / sqrtf((float)RB_K);
    unsigned int* q = slot >= 0 ? rb_queue[slot] : nullptr;
    RbQueue dq;
    dq.init(q, B);
    auto uniform64 = [](long long v) {
        const unsigned int lo = (unsigned int)__builtin_amdgcn_readfirstlane((int)(unsigned int)(unsigned long long)v);
        const unsigned int hi = (unsigned int)__builtin_amdgcn_readfirstlane((int)(unsigned int)((unsigned long long)v >> 32));
        return (long long)(((unsigned long long)hi << 32) | lo);
    };
    auto load_scalars = [&](const long long bb, int& len, long long& cid, long long& base, long long& toff) {
        len = 0;
        cid = -1;
        base = 0;
        toff = 0;
        if (bb >= 0) {
            const long long bs = uniform64(bb);
            len = hist_len ? min((int)hist_len[bs], T) : T;
            len = max(len, 0);
            cid = cand[bs];
            base = row_off[bs];
            toff = tile_off[bs];
        }
    };
    auto load_ids = [&](const long long bb, const int len, long long (&id)[4]) {
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const int j = 16 * t + r16;
            id[t] = j < len ? hist[bb * T + j] : -1;
        }
    };
    long long b = dq.take();
    long long bn = b >= 0 ? dq.take() : -1;
    long long bt = bn >= 0 ? dq.take() : -1;
    long long bq = bt >= 0 ? dq.take() : -1;
    int len, len_n, len_t;
    long long cid, cid_n, cid_t, base, base_n, base_t, toff, toff_n, toff_t, id[4], id_n[4];
    load_scalars(b, len, cid, base, toff);
    load_scalars(bn, len_n, cid_n, base_n, toff_n);
    load_scalars(bt, len_t, cid_t, base_t, toff_t);
    load_ids(b, len, id);
    load_ids(bn, len_n, id_n);

    while (b >= 0) {
        const int RT = (len + 15) >> 4;
        // ---- candidate row, d out, the forward's weights of this sample's rows ---------------------------------------------------------------------
        float4 an[4], gn[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            an[i] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (cid >= 0) an[i] = rb_ld4(table + cid * RB_K + 16 * i + 4 * kk);
            gn[i] = rb_ld4(gout + b * RB_K + 16 * i + 4 * kk);
        }
        float wrow[4];
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const int j = 16 * t + r16;
            wrow[t] = (t < RT && j < T) ? scores[b * T + j] : 0.f;
        }
        // ---- pre-pass: dw_j = g . h_j over all tiles (the rows come back from L2 in the main pass) ------------------------------------------------
        float dwv[4] = {0.f, 0.f, 0.f, 0.f};
        {
            float4 hall[4][4];
#pragma unroll
            for (int t = 0; t < 4; ++t)
                if (t < RT) rb_load_row(table, kk, id[t], hall[t]);
#pragma unroll
            for (int t = 0; t < 4; ++t)
                if (t < RT) {
                    float part = 0.f;
#pragma unroll
                    for (int i = 0; i < 4; ++i) part = rb_dot4(hall[t][i], gn[i], part);
                    part += __shfl_xor(part, 16, 64);
                    part += __shfl_xor(part, 32, 64);
                    dwv[t] = part;
                }
        }
        float ds[4];
        {
            float tsum = 0.f;
            if (normalize) {
#pragma unroll
                for (int t = 0; t < 4; ++t) tsum += row16_sum(wrow[t] * dwv[t]);
            }
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                const bool ok = t < RT && id[t] >= 0;
                const float v = normalize ? wrow[t] * (dwv[t] - tsum) * inv_sqrt_k : dwv[t];
                ds[t] = ok ? v : 0.f;
            }
        }
        // compact output rows: rank of (t, r) among the sample's valid rows
        int rank[4];
        {
            int before = 0;
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                const unsigned long long bal = __ballot(id[t] >= 0) & 0xffffull;      // lane group kk = 0 lists the tile's 16 rows
                rank[t] = before + __popcll(bal & ((1ull << r16) - 1ull));
                before += __popcll(bal);
            }
        }
        // ---- per-sample LDS slots: candidate row, d out, c[m] = sum_f a[f] (Wa - Wd)[f][m] + b1[m] -------------------------------------------
        if (r16 == 0) {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                *reinterpret_cast<float4*>(&sh.av[w][16 * i + 4 * kk]) = an[i];
                *reinterpret_cast<float4*>(&sh.gv[w][16 * i + 4 * kk]) = gn[i];
            }
        }
        if constexpr (!SAVED) {
#pragma unroll
            for (int mt = 0; mt < 5; ++mt) {
                float part = 0.f;
#pragma unroll
                for (int i = 0; i < 4; ++i) part = rb_dot4(an[i], rb_ld4(&sh.wc[(16 * mt + r16) * RB_WS + 16 * i + 4 * kk]), part);
                part += __shfl_xor(part, 16, 64);
                part += __shfl_xor(part, 32, 64);
                if (kk == 0) sh.cvec[w][16 * mt + r16] = part + sh.b1[16 * mt + r16];
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");

        f32x4r Sacc[5];
        float4 gacc[4];
#pragma unroll
        for (int mt = 0; mt < 5; ++mt) Sacc[mt] = (f32x4r){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int i = 0; i < 4; ++i) gacc[i] = make_float4(0.f, 0.f, 0.f, 0.f);

        float4 hv[4], hvn[4];
        rb_load_row(table, kk, id[0], hv);
#pragma unroll 1
        for (int t = 0; t < RT; ++t) {
            const long long idt = t == 0 ? id[0] : t == 1 ? id[1] : t == 2 ? id[2] : id[3];
            const float dst = t == 0 ? ds[0] : t == 1 ? ds[1] : t == 2 ? ds[2] : ds[3];
            const float wt = t == 0 ? wrow[0] : t == 1 ? wrow[1] : t == 2 ? wrow[2] : wrow[3];
            const int rk = t == 0 ? rank[0] : t == 1 ? rank[1] : t == 2 ? rank[2] : rank[3];
            if (t + 1 < RT) {
                const long long idn = t == 0 ? id[1] : t == 1 ? id[2] : id[3];
                rb_load_row(table, kk, idn, hvn);
            }
            float* srow = scratch + ((toff + t) * 16 + r16) * RB_ROW;
            const bool inlen = 16 * t + r16 < len;              // rows past the history's end leave no record (din_wgrad_k reads zeros)
            f32x4r z1[5], dp2[3];
            if constexpr (SAVED) {
                // z1, z2 of this row from its record; dpre2^T = d score * W3 * z2 (1 - z2)
#pragma unroll
                for (int mt = 0; mt < 5; ++mt) {
                    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
                    if (inlen) v = din_rec_load(srow + RB_Z1 + 16 * mt + 4 * kk);
                    z1[mt] = (f32x4r){v.x, v.y, v.z, v.w};
                }
#pragma unroll
                for (int m2 = 0; m2 < 3; ++m2) {
                    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
                    if (inlen) v = din_rec_load(srow + RB_Z2 + 16 * m2 + 4 * kk);
                    const float4 w4 = rb_ld4(&sh.w3[16 * m2 + 4 * kk]);
                    const float zz[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
                    for (int g = 0; g < 4; ++g) dp2[m2][g] = dst * rb_c(w4, g) * zz[g] * (1.0f - zz[g]);
                }
                if (kk == 0 && inlen) srow[RB_DS] = dst;
            } else {
            // ---- layer 1 (recompute): pre1^T -> z1^T -------------------------------------------------------------------------------------------------
            f32x4r acc1[5][2];
            {
                float4 hp[4];
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const float4 a4 = rb_ld4(&sh.av[w][16 * i + 4 * kk]);
                    hp[i] = make_float4(hv[i].x * a4.x, hv[i].y * a4.y, hv[i].z * a4.z, hv[i].w * a4.w);
                }
#pragma unroll
                for (int mt = 0; mt < 5; ++mt) {
                    const float4 c4 = rb_ld4(&sh.cvec[w][16 * mt + 4 * kk]);
                    acc1[mt][0] = (f32x4r){c4.x, c4.y, c4.z, c4.w};
                    acc1[mt][1] = (f32x4r){0.f, 0.f, 0.f, 0.f};
                }
#pragma unroll
                for (int mt = 0; mt < 5; ++mt) {
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        const float4 ah = rb_ld4(&sh.whd[(16 * mt + r16) * RB_WS + 16 * i + 4 * kk]);
                        const float4 ap = rb_ld4(&sh.wp[(16 * mt + r16) * RB_WS + 16 * i + 4 * kk]);
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            acc1[mt][0] = RB_MFMA(rb_c(ah, e), rb_c(hv[i], e), acc1[mt][0]);
                            acc1[mt][1] = RB_MFMA(rb_c(ap, e), rb_c(hp[i], e), acc1[mt][1]);
                        }
                    }
                    RB_SCHED_FENCE();
                }
            }
#pragma unroll
            for (int mt = 0; mt < 5; ++mt) {
#pragma unroll
                for (int g = 0; g < 4; ++g) z1[mt][g] = rb_sigmoid_pre(acc1[mt][0][g] + acc1[mt][1][g]);
                if (inlen) *reinterpret_cast<float4*>(srow + RB_Z1 + 16 * mt + 4 * kk) = make_float4(z1[mt][0], z1[mt][1], z1[mt][2], z1[mt][3]);
            }
            // ---- layer 2 (recompute): pre2^T -> z2^T; dpre2^T = d score * W3 * z2 (1 - z2) ---------------------------------------------------------------
            {
                f32x4r acc2[3];
#pragma unroll
                for (int m2 = 0; m2 < 3; ++m2) {
                    const float4 c4 = rb_ld4(&sh.b2[16 * m2 + 4 * kk]);
                    acc2[m2] = (f32x4r){c4.x, c4.y, c4.z, c4.w};
                }
#pragma unroll
                for (int mt = 0; mt < 5; ++mt) {
                    float4 aw[3];
#pragma unroll
                    for (int m2 = 0; m2 < 3; ++m2) aw[m2] = rb_ld4(&sh.w2[(16 * m2 + r16) * RB_W2S + 16 * mt + 4 * kk]);
#pragma unroll
                    for (int g = 0; g < 4; ++g)
#pragma unroll
                        for (int m2 = 0; m2 < 3; ++m2) acc2[m2] = RB_MFMA(rb_c(aw[m2], g), z1[mt][g], acc2[m2]);
                    RB_SCHED_FENCE();
                }
#pragma unroll
                for (int m2 = 0; m2 < 3; ++m2) {
                    const float4 w4 = rb_ld4(&sh.w3[16 * m2 + 4 * kk]);
                    float zz[4];
#pragma unroll
                    for (int g = 0; g < 4; ++g) {
                        zz[g] = rb_sigmoid_pre(acc2[m2][g]);
                        dp2[m2][g] = dst * rb_c(w4, g) * zz[g] * (1.0f - zz[g]);
                    }
                    if (inlen) *reinterpret_cast<float4*>(srow + RB_Z2 + 16 * m2 + 4 * kk) = make_float4(zz[0], zz[1], zz[2], zz[3]);
                }
                if (kk == 0 && inlen) srow[RB_DS] = dst;
            }
            }
            // ---- dz1^T = W2 dpre2^T -> dpre1^T = dz1 z1 (1 - z1) ------------------------------------------------------------------------------------------
            f32x4r dp1[5];
#pragma unroll
            for (int mt = 0; mt < 5; ++mt) dp1[mt] = (f32x4r){0.f, 0.f, 0.f, 0.f};
            if constexpr (Sh::kBf3) {
                const int lane4 = 4 * lane;
#pragma unroll
                for (int ks = 0; ks < 2; ++ks) {           // h2 0..31 | 32..47 (+ 16 zeros)
                    rb_bf16x8 xb[3];
                    rb_split8(dp2[2 * ks], ks == 0 ? dp2[1] : (f32x4r){0.f, 0.f, 0.f, 0.f}, xb);
#pragma unroll
                    for (int mt = 0; mt < 5; ++mt) {
                        rb_bf16x8 aw[3];
                        rb_lda(sh.w2b3, ks * 5 + mt, lane4, aw);
                        dp1[mt] = rb_mfma6(aw, xb, dp1[mt]);
                    }
                }
            } else {
#pragma unroll
                for (int m2 = 0; m2 < 3; ++m2) {
                    float4 aw[5];
#pragma unroll
                    for (int mt = 0; mt < 5; ++mt) aw[mt] = rb_ld4(&sh.w2b[(16 * mt + r16) * RB_W2BS + 16 * m2 + 4 * kk]);
#pragma unroll
                    for (int g = 0; g < 4; ++g)
#pragma unroll
                        for (int mt = 0; mt < 5; ++mt) dp1[mt] = RB_MFMA(rb_c(aw[mt], g), dp2[m2][g], dp1[mt]);
                    RB_SCHED_FENCE();
                }
            }
#pragma unroll
            for (int mt = 0; mt < 5; ++mt) {
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    dp1[mt][g] = dp1[mt][g] * z1[mt][g] * (1.0f - z1[mt][g]);
                    Sacc[mt][g] += dp1[mt][g];
                }
                if (inlen) din_rec_store(srow + RB_DP1 + 16 * mt + 4 * kk, dp1[mt][0], dp1[mt][1], dp1[mt][2], dp1[mt][3]);
            }
            // ---- dX^T = [Wh+Wd ; Wp] dpre1^T: features 0..63 -> d h through h, 64..127 -> through h * a -------------------------------------------------
            f32x4r dx[8];
#pragma unroll
            for (int ft = 0; ft < 8; ++ft) dx[ft] = (f32x4r){0.f, 0.f, 0.f, 0.f};
            if constexpr (Sh::kBf3) {
                const int lane4 = 4 * lane;
#pragma unroll
                for (int ks = 0; ks < 3; ++ks) {           // hidden 0..31 | 32..63 | 64..79 (+ 16 zeros)
                    rb_bf16x8 xb[3];
                    rb_split8(dp1[2 * ks], 2 * ks + 1 < 5 ? dp1[2 * ks + 1 < 5 ? 2 * ks + 1 : 0] : (f32x4r){0.f, 0.f, 0.f, 0.f}, xb);
#pragma unroll
                    for (int ft = 0; ft < 8; ++ft) {
                        rb_bf16x8 aw[3];
                        rb_lda(sh.wcat3, ks * 8 + ft, lane4, aw);
                        dx[ft] = rb_mfma6(aw, xb, dx[ft]);
                    }
                }
            } else {
#pragma unroll
                for (int mt = 0; mt < 5; ++mt) {
#pragma unroll
                    for (int half = 0; half < 2; ++half) {
                        float4 aw[4];
#pragma unroll
                        for (int qd = 0; qd < 4; ++qd) aw[qd] = rb_ld4(&sh.wcat[(16 * (4 * half + qd) + r16) * RB_WCS + 16 * mt + 4 * kk]);
#pragma unroll
                        for (int g = 0; g < 4; ++g)
#pragma unroll
                            for (int qd = 0; qd < 4; ++qd) dx[4 * half + qd] = RB_MFMA(rb_c(aw[qd], g), dp1[mt][g], dx[4 * half + qd]);
                        RB_SCHED_FENCE();
                    }
                }
            }
            // d h_j = dXh + dXp * a + w_j g (compact row list); d a += dXp * h_j
            {
                float* ghrow = gh + (base + rk) * RB_K;
                const bool okrow = idt >= 0;
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const float4 a4 = rb_ld4(&sh.av[w][16 * i + 4 * kk]);
                    const float4 g4 = rb_ld4(&sh.gv[w][16 * i + 4 * kk]);
                    float4 o;
                    o.x = fmaf(dx[4 + i][0], a4.x, dx[i][0]) + wt * g4.x;
                    o.y = fmaf(dx[4 + i][1], a4.y, dx[i][1]) + wt * g4.y;
                    o.z = fmaf(dx[4 + i][2], a4.z, dx[i][2]) + wt * g4.z;
                    o.w = fmaf(dx[4 + i][3], a4.w, dx[i][3]) + wt * g4.w;
                    if (okrow) din_rec_store(ghrow + 16 * i + 4 * kk, o.x, o.y, o.z, o.w);       // the table gradient's rows: read later by torch
                    gacc[i].x = fmaf(dx[4 + i][0], hv[i].x, gacc[i].x);
                    gacc[i].y = fmaf(dx[4 + i][1], hv[i].y, gacc[i].y);
                    gacc[i].z = fmaf(dx[4 + i][2], hv[i].z, gacc[i].z);
                    gacc[i].w = fmaf(dx[4 + i][3], hv[i].w, gacc[i].w);
                }
            }
#pragma unroll
            for (int i = 0; i < 4; ++i) hv[i] = hvn[i];
        }
        // ---- per-sample outputs: d a (without the c term, added by the host as before) and S = column sums of dpre1 ---------------------------------------
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            float4 t4;
            t4.x = row16_sum(gacc[i].x); t4.y = row16_sum(gacc[i].y); t4.z = row16_sum(gacc[i].z); t4.w = row16_sum(gacc[i].w);
            if (r16 == 0) *reinterpret_cast<float4*>(ga + b * RB_K + 16 * i + 4 * kk) = t4;
        }
#pragma unroll
        for (int mt = 0; mt < 5; ++mt) {
            float4 t4;
            t4.x = row16_sum(Sacc[mt][0]); t4.y = row16_sum(Sacc[mt][1]); t4.z = row16_sum(Sacc[mt][2]); t4.w = row16_sum(Sacc[mt][3]);
            if (r16 == 0 && 16 * mt + 4 * kk < H1) *reinterpret_cast<float4*>(Sout + b * H1 + 16 * mt + 4 * kk) = t4;
        }
        // ---- advance the descriptor pipeline --------------------------------------------------------------------------------------------------------
        b = bn; len = len_n; cid = cid_n; base = base_n; toff = toff_n;
#pragma unroll
        for (int t = 0; t < 4; ++t) id[t] = id_n[t];
        bn = bt; len_n = len_t; cid_n = cid_t; base_n = base_t; toff_n = toff_t;
        load_ids(bn, len_n, id_n);
        bt = bq;
        load_scalars(bt, len_t, cid_t, base_t, toff_t);
        bq = bt >= 0 ? dq.take() : -1;
    }
    dq.drain();
    if (lane == 0 && q) {
        const unsigned int done = atomicAdd(&q[RB_GROUPS], 1u);
        if (done == gridDim.x * RB_WAVES - 1) {
#pragma unroll
            for (int i = 0; i <= RB_GROUPS; ++i) atomicExch(&q[i], 0u);
        }
    }
}

// ------------------------------------------------------------------------------------------------------------------------------------
// din_wgrad_k: the batch-wide reductions, streaming over the scratch tiles.  Workgroup g takes samples g, g + G, ...; per 16-row tile
// the four waves stage [h rows | scratch record] into LDS (double buffered, one barrier per tile) and run din_bwd_k's row-reduction
// MFMAs on it: wave w owns feature tile w of d[Wh+Wd] and dWp (10 accumulators) and hidden tile w (+ a share of the fifth) of dW2 (4).
// ------------------------------------------------------------------------------------------------------------------------------------
constexpr int WG_HS = RB_K + 4;           // LDS row stride of the staged history rows
constexpr int WG_RS = RB_ROW;             // LDS row stride of the staged scratch rows: 212 -> rows 4 apart sit 16 banks apart (the
                                          // row-reduction reads of lane groups kk = 0..3 hit 64 different banks); 68 does the same for h
struct DinWgradSh {
    float h[2][16 * WG_HS];
    float r[2][16 * WG_RS];               // z1 | dpre1 | z2 -> dpre2 in place | ds
    float av[2][RB_K];
};

__global__ __launch_bounds__(256, 4) void din_wgrad_k(const float* __restrict__ table, const int64_t* __restrict__ hist,
                                                      const int32_t* __restrict__ hist_len, const int64_t* __restrict__ cand, int T,
                                                      const float* __restrict__ W3, int H2, long long B,
                                                      const int64_t* __restrict__ tile_off, const float* __restrict__ scratch,
                                                      float* __restrict__ partials) {
    __shared__ __attribute__((aligned(16))) DinWgradSh sh;
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, r16 = lane & 15, kk = lane >> 4;
    f32x4r gAPh[5], gAPp[5], gW2a[4];
#pragma unroll
    for (int i = 0; i < 5; ++i) gAPh[i] = gAPp[i] = (f32x4r){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int i = 0; i < 4; ++i) gW2a[i] = (f32x4r){0.f, 0.f, 0.f, 0.f};
    float gW3acc = 0.f, gb2acc = 0.f, gb3acc = 0.f;
    const int c2e = tid % 48, rge = tid / 48;             // dpre2 transform: thread <-> (H2 column, row group), tid < 192
    const float w3c = (tid < 192 && c2e < H2) ? W3[c2e] : 0.f;
    const int c2w = 16 * (w < 3 ? w : 2) + r16;           // dW2: hidden tile 4 x H2 tile w; wave 3 repeats wave 2's (its copy is dropped)

    // staging registers of the NEXT tile: 16 rows x 64 floats of history (one float4 per thread) + 16 x 212 scratch floats (848 float4)
    constexpr int NSC = (16 * RB_ROW / 4 + 255) / 256;    // 4 float4 per thread (the last one partly idle)
    float4 hreg, sreg[NSC], areg;
    auto tile_count = [&](long long bb) {
        int len = hist_len ? min((int)hist_len[bb], T) : T;
        len = max(len, 0);
        return (len + 15) >> 4;
    };
    auto issue = [&](long long bb, int t, long long toff) {
        const int row = tid >> 4, c = tid & 15;
        const int j = 16 * t + row;
        int len = hist_len ? min((int)hist_len[bb], T) : T;
        const long long idv = j < len ? hist[bb * T + j] : -1;
        hreg = make_float4(0.f, 0.f, 0.f, 0.f);
        if (idv >= 0) hreg = rb_ld4(table + idv * RB_K + 4 * c);
        const float4* src = reinterpret_cast<const float4*>(scratch + (toff + t) * 16 * RB_ROW);
        const int nrow4 = min(16, len - 16 * t) * (RB_ROW / 4);        // rows past the history's end were never written: zeros
#pragma unroll
        for (int k = 0; k < NSC; ++k) {
            const int e = tid + 256 * k;
            sreg[k] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (e < nrow4) sreg[k] = din_rec_load(reinterpret_cast<const float*>(src + e));
        }
        areg = make_float4(0.f, 0.f, 0.f, 0.f);
        if (t == 0 && tid < 16) {
            const long long cid = cand[bb];
            if (cid >= 0) areg = rb_ld4(table + cid * RB_K + 4 * tid);
        }
    };
    auto stage = [&](int buf, int t, int abuf) {
        const int row = tid >> 4, c = tid & 15;
        *reinterpret_cast<float4*>(&sh.h[buf][row * WG_HS + 4 * c]) = hreg;
#pragma unroll
        for (int k = 0; k < NSC; ++k) {
            const int e = tid + 256 * k;
            if (e < 16 * RB_ROW / 4) {
                const int rr = e / (RB_ROW / 4), cc = e - rr * (RB_ROW / 4);
                *reinterpret_cast<float4*>(&sh.r[buf][rr * WG_RS + 4 * cc]) = sreg[k];
            }
        }
        if (t == 0 && tid < 16) *reinterpret_cast<float4*>(&sh.av[abuf][4 * tid]) = areg;
    };

    // tile stream of this workgroup: (sample, tile) pairs in order
    const long long G = gridDim.x;
    long long bcur = blockIdx.x;
    int tcur = 0, ntc = 0;
    long long toffc = 0;
    auto advance_to_valid = [&]() {           // skip samples without rows
        while (bcur < B) {
            ntc = tile_count(bcur);
            if (ntc > 0) {
                toffc = tile_off[bcur];
                return true;
            }
            bcur += G;
        }
        return false;
    };
    bool have = advance_to_valid();
    int buf = 0, abuf = 0;
    if (have) {
        issue(bcur, 0, toffc);
        stage(0, 0, 0);
    }
    __syncthreads();
    while (have) {
        // the tile after this one
        long long bnx = bcur;
        int tnx = tcur + 1, ntn = ntc;
        long long toffn = toffc;
        bool more = true;
        if (tnx >= ntc) {
            const long long bsave = bcur;
            const int nts = ntc;
            const long long tos = toffc;
            bcur += G;
            more = advance_to_valid();
            bnx = bcur; tnx = 0; ntn = ntc; toffn = toffc;
            bcur = bsave; ntc = nts; toffc = tos;
        }
        if (more) issue(bnx, tnx, toffn);
        float* rr = sh.r[buf];
        const float* hh = sh.h[buf];
        // dpre2 in place of z2 (+ the running sums of dW3, db2, db3)
        if (tid < 192) {
#pragma unroll
            for (int row = rge; row < 16; row += 4) {
                const float z = rr[row * WG_RS + RB_Z2 + c2e];
                const float d = rr[row * WG_RS + RB_DS];
                gW3acc = fmaf(d, z, gW3acc);
                const float dp = d * w3c * z * (1.0f - z);
                gb2acc += dp;
                rr[row * WG_RS + RB_Z2 + c2e] = dp;
            }
        }
        if (tid < 16) gb3acc += rr[tid * WG_RS + RB_DS];
        __syncthreads();
        const float a_f = sh.av[abuf][16 * w + r16];
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            const int row = 4 * kk + s;
            const float a = rr[row * WG_RS + RB_Z1 + 16 * w + r16];
            const float a4 = rr[row * WG_RS + RB_Z1 + 64 + r16];
            const float b0 = rr[row * WG_RS + RB_Z2 + r16], b1v = rr[row * WG_RS + RB_Z2 + 16 + r16], b2v = rr[row * WG_RS + RB_Z2 + 32 + r16];
            const float bw = rr[row * WG_RS + RB_Z2 + c2w];
            gW2a[0] = RB_MFMA(a, b0, gW2a[0]);
            gW2a[1] = RB_MFMA(a, b1v, gW2a[1]);
            gW2a[2] = RB_MFMA(a, b2v, gW2a[2]);
            gW2a[3] = RB_MFMA(a4, bw, gW2a[3]);
            const float hv = hh[row * WG_HS + 16 * w + r16];
            const float hp = hv * a_f;
#pragma unroll
            for (int ni = 0; ni < 5; ++ni) {
                const float bv = rr[row * WG_RS + RB_DP1 + 16 * ni + r16];
                gAPh[ni] = RB_MFMA(hv, bv, gAPh[ni]);
                gAPp[ni] = RB_MFMA(hp, bv, gAPp[ni]);
            }
        }
        if (more) stage(buf ^ 1, tnx, tnx == 0 ? (abuf ^ 1) : abuf);
        __syncthreads();
        buf ^= 1;
        if (more && tnx == 0) abuf ^= 1;
        have = more;
        bcur = bnx; tcur = tnx; ntc = ntn; toffc = toffn;
    }
    // ---- this workgroup's partial record (din.hip's record layout) ---------------------------------------------------------------------------------
    float* rec = partials + (size_t)blockIdx.x * kRowsBwdRec;
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
        for (int g = 0; g < 4; ++g) rec[kRowsBwdGAP + (16 * w + 4 * kk + g) * 48 + 16 * ni + r16] = gW2a[ni][g];
    if (w < 3) {
#pragma unroll
        for (int g = 0; g < 4; ++g) rec[kRowsBwdGAP + (64 + 4 * kk + g) * 48 + 16 * w + r16] = gW2a[3][g];
    }
    if (tid < 192) {
        rec[kRowsBwdGAP + kRowsBwdGW2 + rge * 48 + c2e] = gb2acc;
        rec[kRowsBwdGAP + kRowsBwdGW2 + 192 + rge * 48 + c2e] = gW3acc;
    }
    if (tid < 64) rec[kRowsBwdGAP + kRowsBwdGW2 + 384 + tid] = tid < 16 ? gb3acc : 0.f;
}

static std::atomic<unsigned int> rb_next_slot{0};
constexpr int kRowsWgradWg = 4 * kCUs;

}  // namespace dir

using namespace dir;

extern "C" int64_t dir_din_backward_rows_workspace_bytes(int K, int H1, int H2, int64_t n_tiles) {
    if (K != 64 || H1 <= 0 || H2 <= 0 || H1 > 80 || H2 > 48 || n_tiles < 0) return 0;
    const int64_t scratch = ((n_tiles > 0 ? n_tiles : 1) * 16 * RB_ROW * (int64_t)sizeof(float) + 255) & ~(int64_t)255;
    return scratch + (int64_t)(kRowsWgradWg + 1) * kRowsBwdRec * (int64_t)sizeof(float);
}

static int din_backward_rows(const char* name, bool saved, const float* table, int K, const int64_t* hist, const int32_t* hist_len,
                             const int64_t* cand, int T, const float* W1, const float* b1, int H1, const float* W2, const float* b2, int H2,
                             const float* W3, const float* b3, int normalize, int64_t B, const float* gout, const float* scores,
                             const int64_t* row_off, const int64_t* tile_off, int64_t n_tiles, float* gh, float* ga, float* S, float* gAP,
                             float* gW2, float* gb2, float* gW3, float* gb3, void* workspace, int64_t workspace_bytes, dir_stream_t stream) {
    (void)b3;
    DIR_CHECK_ARG(K > 0 && T > 0 && H1 > 0 && H2 > 0 && B >= 0 && n_tiles >= 0, "%s: K=%d T=%d H1=%d H2=%d", name, K, T, H1, H2);
    if (K != 64 || H1 > 80 || H2 > 48 || (H1 & 3) || (H2 & 3) || T > 64)
        return fail(DIR_E_UNSUPPORTED, "%s: covers K = 64, H1 <= 80, H2 <= 48 (multiples of 4), T <= 64 (K=%d H1=%d H2=%d T=%d)", name, K, H1, H2, T);
    DIR_CHECK_ARG(table && hist && cand && W1 && b1 && W2 && b2 && W3 && gAP && gW2 && gb2 && gW3 && gb3 && workspace, "%s: null pointer", name);
    DIR_CHECK_ARG(B == 0 || (gout && scores && row_off && tile_off && ga && S), "%s: null pointer", name);
    if (!aligned16(table) || !aligned16(gout) || !aligned16(W1) || !aligned16(gh) || !aligned16(ga) || !aligned16(S) || !aligned16(workspace))
        return fail(DIR_E_BADARG, "%s: table / gout / W1 / gh / ga / S / workspace must be 16-byte aligned", name);
    const int64_t need = dir_din_backward_rows_workspace_bytes(K, H1, H2, n_tiles);
    if (workspace_bytes < need) return fail(DIR_E_BADARG, "%s: workspace needs %lld bytes", name, (long long)need);
    static_assert(kRowsBwdRec == 2 * 64 * 80 + 80 * 48 + 4 * 48 + 4 * 48 + 64, "record layout shared with din.hip");
    hipStream_t st = as_stream(stream);
    float* scratch = static_cast<float*>(workspace);
    const int64_t scratch_bytes = ((n_tiles > 0 ? n_tiles : 1) * 16 * RB_ROW * (int64_t)sizeof(float) + 255) & ~(int64_t)255;
    float* partials = reinterpret_cast<float*>(static_cast<char*>(workspace) + scratch_bytes);
    int nwg2 = 0;
    if (B > 0) {
        // DIR_DIN_BWD_ARITH = bf16x3 (default) | f32: the arithmetic of the saved-activation row pass's two GEMMs (read per call)
        const char* arith = getenv("DIR_DIN_BWD_ARITH");
        const bool bf3 = saved && !(arith && strcmp(arith, "f32") == 0);
        typedef void (*kern_t)(const float*, const int64_t*, const int32_t*, const int64_t*, int, const float*, const float*, int, const float*,
                               const float*, int, const float*, int, long long, const float*, const float*, const int64_t*, const int64_t*, float*,
                               float*, float*, float*, int);
        const int which = bf3 ? 2 : saved ? 1 : 0;
        static const kern_t kerns[3] = {&din_rows_k<false, DinRowsSh>, &din_rows_k<true, DinRowsSh>, &din_rows_k<true, DinRowsSh3>};
        static LdsOnce once[3];
        const size_t shmem = bf3 ? sizeof(DinRowsSh3) : sizeof(DinRowsSh);
        if (!lds_limit(once[which], (int)shmem, kerns[which])) return fail(DIR_E_HIP, "%s: cannot reserve %zu B of LDS", name, shmem);
        const char* stat = getenv("DIR_DIN_STATIC");
        const bool static_split = stat && atoi(stat) != 0;
        const int slot = static_split ? -1 : (int)(rb_next_slot.fetch_add(1) % RB_SLOTS);
        const int64_t waves_wanted = (B + 1) / 2;
        int64_t nwg = (waves_wanted + RB_WAVES - 1) / RB_WAVES;
        if (nwg > kCUs) nwg = kCUs;
        if (nwg < 1) nwg = 1;
        hipLaunchKernelGGL(kerns[which], dim3((unsigned)nwg), dim3(64 * RB_WAVES), shmem, st, table, hist, hist_len, cand, T, W1, b1, H1, W2, b2, H2,
                           W3, normalize, (long long)B, gout, scores, row_off, tile_off, gh, ga, S, scratch, slot);
        DIR_CHECK_LAUNCH(name);
        nwg2 = (int)(B < kRowsWgradWg ? B : kRowsWgradWg);
        hipLaunchKernelGGL(din_wgrad_k, dim3((unsigned)nwg2), dim3(256), 0, st, table, hist, hist_len, cand, T, W3, H2, (long long)B, tile_off,
                           scratch, partials);
        DIR_CHECK_LAUNCH(name);
    }
    float* red = partials + (size_t)kRowsWgradWg * kRowsBwdRec;
    hipLaunchKernelGGL(din_bwd_sum_k, dim3((unsigned)((kRowsBwdRec + 63) / 64)), dim3(1024), 0, st, partials, nwg2, red);
    const int nout = 2 * K * H1 + H1 * H2 + 2 * H2 + 1;
    hipLaunchKernelGGL(din_bwd_finish_k, dim3((unsigned)((nout + 255) / 256)), dim3(256), 0, st, red, K, H1, H2, gAP, gW2, gb2, gW3, gb3);
    DIR_CHECK_LAUNCH(name);
    return DIR_OK;
}

extern "C" int dir_din_attention_pool_backward_rows_f32(const float* table, int K, const int64_t* hist, const int32_t* hist_len,
                                                        const int64_t* cand, int T, const float* W1, const float* b1, int H1,
                                                        const float* W2, const float* b2, int H2, const float* W3, const float* b3,
                                                        int normalize, int64_t B, const float* gout, const float* scores,
                                                        const int64_t* row_off, const int64_t* tile_off, int64_t n_tiles, float* gh,
                                                        float* ga, float* S, float* gAP, float* gW2, float* gb2, float* gW3, float* gb3,
                                                        void* workspace, int64_t workspace_bytes, dir_stream_t stream) {
    return din_backward_rows("dir_din_attention_pool_backward_rows_f32", false, table, K, hist, hist_len, cand, T, W1, b1, H1, W2, b2, H2, W3, b3,
                             normalize, B, gout, scores, row_off, tile_off, n_tiles, gh, ga, S, gAP, gW2, gb2, gW3, gb3, workspace,
                             workspace_bytes, stream);
}

// As above, with the records' z1 and z2 already in `workspace` (its first n_tiles * 16 * 212 floats): the workspace that
// dir_din_attention_pool_save_f32 filled on the same (hist, hist_len, tile_off) -- the forward is not recomputed.
extern "C" int dir_din_attention_pool_backward_saved_f32(const float* table, int K, const int64_t* hist, const int32_t* hist_len,
                                                         const int64_t* cand, int T, const float* W1, const float* b1, int H1,
                                                         const float* W2, const float* b2, int H2, const float* W3, const float* b3,
                                                         int normalize, int64_t B, const float* gout, const float* scores,
                                                         const int64_t* row_off, const int64_t* tile_off, int64_t n_tiles, float* gh,
                                                         float* ga, float* S, float* gAP, float* gW2, float* gb2, float* gW3, float* gb3,
                                                         void* workspace, int64_t workspace_bytes, dir_stream_t stream) {
    return din_backward_rows("dir_din_attention_pool_backward_saved_f32", true, table, K, hist, hist_len, cand, T, W1, b1, H1, W2, b2, H2, W3, b3,
                             normalize, B, gout, scores, row_off, tile_off, n_tiles, gh, ga, S, gAP, gW2, gb2, gW3, gb3, workspace,
                             workspace_bytes, stream);
}
