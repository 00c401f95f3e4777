// dense_bf3.hip -- the hidden layers of the DNN towers (dense.hip: y = act(x W^T + b)) on the bf16 matrix pipe with fp32-equivalent
// arithmetic ("bf16 x 3", the recipe of cin_bf3.hip).
//
// Reference: tf.layers.dense(units, activation) as called by dnn_logit_fn (models/DeepFM/deepFM.py:295-300), _deep_architecture
// (models/DeepCrossNetwork/DeepCrossNetwork.py:394-399) and _base_model (models/ESMM/ESMM.py:139-142): matmul + bias + activation;
// the batch normalisation that follows the activation (deepFM.py:303-308) is the optional per-column affine of the epilogue.
//
// Arithmetic.  Both fp32 operands are split into three bf16 pieces by round-to-nearest (v = v0 + v1 + v2 exactly, fp32 exponent
// range); the six piece products of weight >= 2^-16 are accumulated in fp32 by v_mfma_f32_16x16x32_bf16 (a bf16 x bf16 product is
// exact in fp32), the three of weight <= 2^-24 are dropped.  Same 1e-5 bar against float64 as dense.hip's fp32 MFMA kernel.
//
// Why it is faster than round 2's first attempt (dense_bf3_k, removed: 1.07-1.15 x the fp32 kernel).  That kernel split X once per
// 128 x 80 output tile -- 13 VALU instructions per pair of elements in front of only 5 column tiles of MFMAs -- and staged both
// operands through LDS.  Here a wave splits its X rows once per k-step for a block of 13 or 16 column tiles (0.5 VALU instructions
// per MFMA), X goes from global memory straight into the registers it is split in (a lane's 8 k-values are 32 contiguous bytes; a row
// tile's four lane groups read whole 128-byte lines), and W arrives pre-split as a packed bf16 image through global_load_lds: no
// staging registers, no ds_write, one ds_read_b128 per operand.  The product is evaluated transposed, D = W-piece (A operand:
// rows = output columns) x X-piece (B operand: columns = batch rows), so that a lane's four accumulator registers are four
// CONSECUTIVE output columns of one row: bias / ReLU / affine / gate on float4s and 16-byte stores.
//
// Work split.  A workgroup of DB3_NW = 4 waves (one per SIMD; 8 until round 5) owns 32 DB3_NW = 128 rows x one column block of CT tiles
// (CT = 13: 208 columns, 16: 256, 8: 128 -- the host picks the one that pads N least); wave w rows [32w, 32w+32) = 2 row tiles x CT
// column tiles.  One k-step of 32 per barrier: 12*CT MFMAs per wave (6*CT on fp16 x 2).  TWO persistent workgroups per CU, each with
// barriers of its own (they drift apart: one's loads, waits and epilogue fall beside the other's matrix instructions), walk the tiles
// in rounds (the column blocks of a row block on one XCD at the same time: the second finds X in the L2), software-pipelined across
// tiles so that a tile's stores drain under the next tile's first k-step.
//
// LDS: Wb [2][3 pieces][CT][64 lanes][8 bf16] -- the k-step's W image, in the order dense_bf3_pack_k writes the global image.
#include "common.hpp"

namespace dir {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));
typedef unsigned int u32x4_t __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) void* lds_ptr_t;
typedef const __attribute__((address_space(1))) void* glb_ptr_t;

#ifndef DB3_NW
#define DB3_NW 4           // waves per workgroup: 4 = two 256-thread workgroups per CU, each with barriers of its own (round 5: the two drift apart, one's loads and
                           // epilogue fall beside the other's matrix instructions: dense_bf3_k<13,2> 85.9 -> 81.7 us, deepfm_train 1.472 -> 1.442 ms); 8 = one 512-thread workgroup
#endif
constexpr int DB3_ROWS = 32 * DB3_NW;
// DB3_CHAIN: a scheduling fence behind every accumulator's six MFMAs keeps them one dependent chain.  A SIMD gives its other wave's VALU
// instructions (the next k-step's operand split) issue slots only while this wave waits on a dependent MFMA -- none while it has
// independent MFMAs to issue (tools/coexec_probe.hip); left alone the compiler interleaves the chains of the two row tiles.  Same
// results bit for bit; 65 536 x 1024 -> 1024: 623 -> 597 us, x 432 -> 1024: 345 -> 336, x 416 -> 400: 114.6 -> 111.8.
#ifndef DB3_CHAIN
#define DB3_CHAIN 1
#endif

__device__ __forceinline__ unsigned int db3_pk(float a, float b) {      // v_cvt_pk_bf16_f32 (round to nearest even), a in the low half
    typedef __bf16 pk2_t __attribute__((ext_vector_type(2)));
    const pk2_t v = {(__bf16)a, (__bf16)b};
    unsigned int w = __builtin_bit_cast(unsigned int, v);
    // An EMPTY asm: it only hides where w came from (otherwise the compiler converts `a` a second time, alone, to form float(bf16(a))
    // instead of shifting w).  The convert itself stays a compiler-generated instruction: its results become MFMA operands, and the
    // compiler's hazard recognizer does not see inside inline asm.
    asm("" : "+v"(w));
    return w;
}
__device__ __forceinline__ void db3_split_pair(float a, float b, unsigned int& w0, unsigned int& w1, unsigned int& w2) {
    w0 = db3_pk(a, b);
    const float ra = a - __builtin_bit_cast(float, w0 << 16), rb = b - __builtin_bit_cast(float, w0 & 0xffff0000u);
    w1 = db3_pk(ra, rb);
    const float sa = ra - __builtin_bit_cast(float, w1 << 16), sb = rb - __builtin_bit_cast(float, w1 & 0xffff0000u);
    w2 = db3_pk(sa, sb);
}

// Round 4: "fp16 x 2" (NP = 2; csrc/cin_bf3.hip explains the arithmetic and its preconditions): two fp16 pieces per operand, three products on
// v_mfma_f32_16x16x32_f16.  For layers whose input is bounded by construction (embedding concatenations, ReLU / batch-normalised
// activations, the CIN's pooled products): NOT for a general input layer that may carry raw numeric columns (fp16 ends at 65 504) and
// not for gradient operands (small magnitudes would sit in fp16's subnormal range) -- those callers keep bf16 x 3.
typedef _Float16 f16x8_t __attribute__((ext_vector_type(8)));
__device__ __forceinline__ unsigned int db3_pk_h(float a, float b) {     // v_cvt_pk_f16_f32 (round to nearest even), a in the low half
    typedef _Float16 h2_t __attribute__((ext_vector_type(2)));
    const h2_t v = {(_Float16)a, (_Float16)b};
    unsigned int w = __builtin_bit_cast(unsigned int, v);
    asm("" : "+v"(w));
    return w;
}
template <int NP> struct Db3Pc;
template <> struct Db3Pc<3> {
    using op_t = bf16x8_t;
    __device__ static __forceinline__ void split(float a, float b, unsigned int (&w)[3]) { db3_split_pair(a, b, w[0], w[1], w[2]); }
    __device__ static __forceinline__ f32x4 mma(const op_t (&wc)[3], const op_t (&xa)[3], f32x4 tt) {      // six products, smallest first
        tt = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wc[0], xa[2], tt, 0, 0, 0);
        tt = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wc[2], xa[0], tt, 0, 0, 0);
        tt = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wc[1], xa[1], tt, 0, 0, 0);
        tt = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wc[0], xa[1], tt, 0, 0, 0);
        tt = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wc[1], xa[0], tt, 0, 0, 0);
        tt = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wc[0], xa[0], tt, 0, 0, 0);
        return tt;
    }
};
template <> struct Db3Pc<2> {
    using op_t = f16x8_t;
    __device__ static __forceinline__ void split(float a, float b, unsigned int (&w)[2]) {
        typedef _Float16 h2_t __attribute__((ext_vector_type(2)));
        w[0] = db3_pk_h(a, b);
        const h2_t h = __builtin_bit_cast(h2_t, w[0]);
        w[1] = db3_pk_h(a - (float)h[0], b - (float)h[1]);
    }
    __device__ static __forceinline__ f32x4 mma(const op_t (&wc)[2], const op_t (&xa)[2], f32x4 tt) {      // three products, smallest first
        tt = __builtin_amdgcn_mfma_f32_16x16x32_f16(wc[1], xa[0], tt, 0, 0, 0);
        tt = __builtin_amdgcn_mfma_f32_16x16x32_f16(wc[0], xa[1], tt, 0, 0, 0);
        tt = __builtin_amdgcn_mfma_f32_16x16x32_f16(wc[0], xa[0], tt, 0, 0, 0);
        return tt;
    }
};

__host__ __device__ inline int db3_ct_for(int N) {     // column tiles per block: the choice that pads N least (ties: the wider block)
    const int tiles = (N + 15) / 16;
    int best = 16, pad = (tiles + 15) / 16 * 16;
    const int p13 = (tiles + 12) / 13 * 13, p8 = (tiles + 7) / 8 * 8;
    if (p13 < pad) { best = 13; pad = p13; }
    if (p8 < pad) { best = 8; pad = p8; }
    return best;
}

// W [N, Kd] fp32 (row stride w_ld) -> image [column block][k-step][piece][ct][lane][8 e] bf16: element e of lane l of column tile ct =
// piece of W[n = 16*(cb*CT + ct) + (l & 15)][k = 32*ks + 8*(l >> 4) + e]; zero where n >= N or k >= Kd.
template <int NP>
__global__ __launch_bounds__(256) void dense_bf3_pack_k(const float* __restrict__ W, int64_t w_ld, int64_t w_cs /* column stride: 1, or the row
                                                        stride of the tensor whose transpose W is */, int Kd, int N, int CT, int nks, int ncb,
                                                        unsigned int* __restrict__ img) {
    const int64_t total = (int64_t)ncb * nks * CT * 64 * 4;   // one thread per pair of e
    for (int64_t e_ = (int64_t)blockIdx.x * 256 + threadIdx.x; e_ < total; e_ += (int64_t)gridDim.x * 256) {
        int64_t q = e_;
        const int ep = (int)(q & 3); q >>= 2;
        const int l = (int)(q & 63); q >>= 6;
        const int ct = (int)(q % CT); q /= CT;
        const int ks = (int)(q % nks);
        const int cb = (int)(q / nks);
        const int n = 16 * (cb * CT + ct) + (l & 15);
        const int k = 32 * ks + 8 * (l >> 4) + 2 * ep;
        const float v0 = (n < N && k < Kd) ? W[(int64_t)n * w_ld + k * w_cs] : 0.f;
        const float v1 = (n < N && k + 1 < Kd) ? W[(int64_t)n * w_ld + (k + 1) * w_cs] : 0.f;
        unsigned int pw[NP];
        Db3Pc<NP>::split(v0, v1, pw);
        const int64_t step = (int64_t)cb * nks + ks;
        const int64_t base = step * (NP * CT * 64 * 4) + (ct * 64 + l) * 4 + ep;    // piece stride: CT*64*4 dwords
#pragma unroll
        for (int q = 0; q < NP; ++q) img[base + q * CT * 64 * 4] = pw[q];
    }
}

// scale 2^k (or its inverse) for a row / tensor whose largest |element| has the bit pattern `bits`: the largest lands in [2^14, 2^15)
// (k clamped to +-100; all-zero rows stay zero)
__device__ __forceinline__ float db3_scale(unsigned int bits, bool inverse) {
    int k = 141 - (int)((bits >> 23) & 0xffu);
    k = k > 100 ? 100 : (k < -100 ? -100 : k);
    return __builtin_bit_cast(float, (unsigned int)(inverse ? 127 - k : 127 + k) << 23);
}

// The fp16 x 2 image: every row n of W (one output column of the layer) is multiplied by its own power of two before the split -- its largest
// |element| lands in [2^14, 2^15), so that elements down to 2^-17 of it keep both pieces in fp16's NORMAL range (22 bits) instead of an
// absolute 2^-25 -- and the inverse goes into the image's tail, invw[n], which the kernel's epilogue multiplies the column by.  Both exact.
// (Unscaled, a weight of 2e-4 carries an absolute error of 3e-8; times a raw numeric input of 99 999 that is 3e-3 on an output of 20.)
// One wave per row: the row's maximum, then its 16 dwords per piece and k-step.  Rows N .. 16 CT ncb - 1 are zero, their invw 1.
__global__ __launch_bounds__(256) void dense_f16x2_pack_rows_k(const float* __restrict__ W, int64_t w_ld, int64_t w_cs, int Kd, int N, int CT, int nks,
                                                               int ncb, unsigned int* __restrict__ img, float* __restrict__ invw) {
    const int lane = threadIdx.x & 63;
    const int n = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (n >= ncb * CT * 16) return;
    const bool real = n < N;
    float mx = 0.f;
    if (real)
        for (int k = lane; k < Kd; k += 64) mx = fmaxf(mx, fabsf(W[(int64_t)n * w_ld + k * w_cs]));
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o, 64));
    const unsigned int bits = __builtin_bit_cast(unsigned int, mx);
    const float sc = real ? db3_scale(bits, false) : 1.f;
    if (lane == 0) invw[n] = real ? db3_scale(bits, true) : 1.f;
    const int tile = n >> 4, cb = tile / CT, ct = tile - cb * CT;
    const int ep = lane & 3, lg = (lane >> 2) & 3, ksl = lane >> 4;               // a lane: pair ep of lane group lg of k-step ks0 + ksl
    const int l = lg * 16 + (n & 15);
    for (int ks0 = 0; ks0 < nks; ks0 += 4) {
        const int ks = ks0 + ksl;
        if (ks >= nks) continue;
        const int k = 32 * ks + 8 * lg + 2 * ep;
        const float v0 = (real && k < Kd) ? W[(int64_t)n * w_ld + k * w_cs] * sc : 0.f;
        const float v1 = (real && k + 1 < Kd) ? W[(int64_t)n * w_ld + (k + 1) * w_cs] * sc : 0.f;
        unsigned int pw[2];
        Db3Pc<2>::split(v0, v1, pw);
        const int64_t base = ((int64_t)cb * nks + ks) * (2 * CT * 64 * 4) + (ct * 64 + l) * 4 + ep;
        img[base] = pw[0];
        img[base + CT * 64 * 4] = pw[1];
    }
}

// RS (fp16 x 2 with an X of unknown magnitude -- a gradient): row r of X is multiplied by the power of two db3_scale(row_bits[r]) before it
// is split and the row's accumulators by the inverse before the epilogue (both exact); row_bits[r] = bit pattern of max_k |X[r, k]|
// (dir_row_absmax_bits_f32).
template <int CT, int NP = 3, bool RS = false>
__global__ __launch_bounds__(64 * DB3_NW, 8 / DB3_NW) void dense_bf3_k(const float* __restrict__ X, int64_t x_ld, const unsigned char* __restrict__ img,
                                                     const float* __restrict__ bias, int relu, const float* __restrict__ post_scale,
                                                     const float* __restrict__ post_shift, const float* __restrict__ gate, int64_t gate_ld,
                                                     int64_t M, int Kd, int N, int nks, int ncb, float* __restrict__ Y, int64_t y_ld,
                                                     const float* __restrict__ head_w /* [N] or nullptr */,
                                                     float* __restrict__ head_part /* [ncb][M]: this column block's share of y . head_w */,
                                                     const unsigned int* __restrict__ row_bits = nullptr /* RS: [M] */,
                                                     unsigned int* __restrict__ y_row_bits = nullptr /* [M], zeroed: atomicMax of the bit pattern of max_n |Y[r, n]| */,
                                                     unsigned int* __restrict__ y_all_bits = nullptr /* one word, zeroed: the same over all rows */) {
    static_assert(!RS || NP == 2, "row scaling belongs to the fp16 x 2 split");
    constexpr int STEPB = NP * CT * 1024;                      // bytes of W image per k-step
    using Pc = Db3Pc<NP>;
    using op_t = typename Pc::op_t;
    extern __shared__ __attribute__((aligned(16))) unsigned char db3_smem[];
    unsigned char* Wb = db3_smem;                              // [2][STEPB]

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int n = lane & 15;
    const int lg = lane >> 4;
    // Persistent workgroups, software-pipelined across their tiles: the last k-step of a tile prefetches the next tile's first
    // operands, and the epilogue's stores are issued without waiting for them, so that they drain under the next tile's first k-step
    // (with one tile per workgroup the whole chip stores in lockstep and nothing computes meanwhile).  Tiles are numbered
    // (row block, column block) and dealt in rounds of gridDim.x; within a round hardware workgroup g (XCD g % 8) takes logical slot
    // xcd * per + g / 8, so that the column blocks of one row block run at the same time on one XCD: X is read from HBM once and
    // found in that XCD's L2 by the others.
    const int64_t nrb = (M + DB3_ROWS - 1) / DB3_ROWS;
    const int64_t ntiles = nrb * ncb;
    const int G = gridDim.x, per = G / 8, rem = G % 8, xcd = blockIdx.x % 8;
    const int lslot = xcd * per + (xcd < rem ? xcd : rem) + blockIdx.x / 8;
    const int t0 = lslot, t1 = (int)ntiles;             // this workgroup's tiles: t0, t0 + G, ... < t1
    if (t0 >= t1) return;

    struct Tile { const float* xs[2]; const unsigned char* gi; int cb; int64_t row0; float sc[RS ? 2 : 1], inv[RS ? 2 : 1]; };
    auto setup = [&](int t, Tile& tl) {
        tl.cb = t % ncb;
        tl.row0 = (int64_t)(t / ncb) * DB3_ROWS;
        tl.gi = img + (int64_t)tl.cb * nks * STEPB;
        // this lane's X rows: row tile rt -> row 32*wave + 16*rt + n of the tile (clamped: a row >= M only feeds outputs that are never
        // stored); k slots 8*lg .. 8*lg+7 of each k-step
#pragma unroll
        for (int rt = 0; rt < 2; ++rt) {
            const int64_t r = tl.row0 + wave * 32 + rt * 16 + n;
            tl.xs[rt] = X + (r < M ? r : M - 1) * x_ld + 8 * lg;
            if constexpr (RS) {
                const unsigned int rb = row_bits[r < M ? r : M - 1];
                tl.sc[rt] = db3_scale(rb, false);
                tl.inv[rt] = db3_scale(rb, true);
            }
        }
    };
    auto stage_w = [&](const Tile& tl, int ks, int buf) {     // 3*CT pieces of 1 KB over 8 waves, lane-linear
        for (int piece = wave; piece < NP * CT; piece += DB3_NW) {
            const unsigned char* src = tl.gi + (int64_t)ks * STEPB + piece * 1024 + lane * 16;
            unsigned char* dst = Wb + buf * STEPB + piece * 1024;
            __builtin_amdgcn_global_load_lds((glb_ptr_t)src, (lds_ptr_t)dst, 16, 0, 0);
        }
    };
    auto load_x = [&](const Tile& tl, int ks, f32x4 (&v)[2][2]) {      // Kd % 4 == 0: a 16-byte piece is wholly inside or outside the row
#pragma unroll
        for (int rt = 0; rt < 2; ++rt)
#pragma unroll
            for (int hq = 0; hq < 2; ++hq) {
                const int k = 32 * ks + 8 * lg + 4 * hq;
                v[rt][hq] = k < Kd ? *reinterpret_cast<const f32x4*>(tl.xs[rt] + 32 * ks + 4 * hq) : (f32x4){0.f, 0.f, 0.f, 0.f};
            }
    };

    f32x4 acc[2][CT];
#pragma unroll
    for (int rt = 0; rt < 2; ++rt)
#pragma unroll
        for (int ct = 0; ct < CT; ++ct) acc[rt][ct] = (f32x4){0.f, 0.f, 0.f, 0.f};

    Tile cur, nxt;
    setup(t0, cur);
    stage_w(cur, 0, 0);
    f32x4 xv[2][2], xn[2][2];
    load_x(cur, 0, xv);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();

    float run_max = 0.f;                                   // y_all_bits: this lane's largest |output| over all tiles of the workgroup
    const unsigned char* wlane = Wb + lane * 16;
    // fp16 x 2: the image's tail holds the inverse of every output column's weight scale (dense_f16x2_pack_rows_k)
    const float* invw = reinterpret_cast<const float*>(img + (int64_t)ncb * nks * STEPB);
    int buf = 0;
    for (int t = t0; t < t1; t += G) {
        const bool more = t + G < t1;
        setup(more ? t + G : t, nxt);
        for (int ks = 0; ks < nks; ++ks, buf ^= 1) {
            // the next step's operands: of this tile, or the first of the next tile (the very last step re-reads its own X: no
            // branch around the loads)
            if (ks + 1 < nks) {
                stage_w(cur, ks + 1, buf ^ 1);
                load_x(cur, ks + 1, xn);
            } else {
                if (more) stage_w(nxt, 0, buf ^ 1);
                load_x(nxt, more ? 0 : ks, xn);
            }
            // split this step's X: three bf16x8 operands per row tile
            op_t xa[2][NP];
#pragma unroll
            for (int rt = 0; rt < 2; ++rt) {
                unsigned int w[NP][4];
#pragma unroll
                for (int hq = 0; hq < 2; ++hq) {
                    unsigned int pa[NP], pb[NP];
                    f32x4 xs4 = xv[rt][hq];
                    if constexpr (RS) xs4 *= cur.sc[rt];
                    Pc::split(xs4[0], xs4[1], pa);
                    Pc::split(xs4[2], xs4[3], pb);
#pragma unroll
                    for (int p = 0; p < NP; ++p) {
                        w[p][2 * hq] = pa[p];
                        w[p][2 * hq + 1] = pb[p];
                    }
                }
#pragma unroll
                for (int p = 0; p < NP; ++p) xa[rt][p] = __builtin_bit_cast(op_t, (u32x4_t){w[p][0], w[p][1], w[p][2], w[p][3]});
            }
            const unsigned char* wl = wlane + buf * STEPB;
            // W operands one column tile ahead, in two register sets used alternately (the loop is unrolled: the set index is a
            // compile-time constant; copying "next" into "current" costs 12 v_mov per tile, one VALU instruction per MFMA)
            op_t wq[2][NP];
#pragma unroll
            for (int p = 0; p < NP; ++p) wq[0][p] = *reinterpret_cast<const op_t*>(wl + p * CT * 1024);
#pragma unroll
            for (int ct = 0; ct < CT; ++ct) {
                if (ct + 1 < CT) {
#pragma unroll
                    for (int p = 0; p < NP; ++p) wq[(ct + 1) & 1][p] = *reinterpret_cast<const op_t*>(wl + p * CT * 1024 + (ct + 1) * 1024);
                }
                const op_t (&wc)[NP] = wq[ct & 1];
#pragma unroll
                for (int rt = 0; rt < 2; ++rt) {
                    acc[rt][ct] = Pc::mma(wc, xa[rt], acc[rt][ct]);
#if DB3_CHAIN
                    __builtin_amdgcn_sched_barrier(0);      // one dependent chain per accumulator (tools/coexec_probe.hip)
#endif
                }
            }
#pragma unroll
            for (int rt = 0; rt < 2; ++rt)
#pragma unroll
                for (int hq = 0; hq < 2; ++hq) xv[rt][hq] = xn[rt][hq];
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // this wave's W pieces of the next step have landed in LDS (and, in the
            __syncthreads();                                    // first step of a tile, the previous tile's stores have drained)
        }

        // ---- epilogue: D = W-piece x X-piece, C/D map of 16x16x32: col = lane & 15 = batch row, row = 4*lg + reg = output column.
        // The stores are not waited for here.  The epilogue's operands (the columns' weight scales / bias / affine / head weights: one
        // float4 per column tile; the gate: one per row tile too) are loaded UNCONDITIONALLY from clamped addresses, one column tile AHEAD
        // of their use: written as `if (r < M && col < N) v += bias[col]` every load sits inside the guard's branch and is waited for right
        // there -- up to six dependent memory round trips per tile, 32 tiles per lane, with the matrix pipe idle (round 5, tools/serial_loads.py).
        struct EpiOps { f32x4 iw, bs, sc, sh, hw, gt[2]; };
        const int64_t er[2] = {cur.row0 + wave * 32 + n, cur.row0 + wave * 32 + 16 + n};
        const int64_t erc[2] = {er[0] < M ? er[0] : M - 1, er[1] < M ? er[1] : M - 1};
        auto epi_load = [&](int ct, EpiOps& o) {
            const int col = 16 * (cur.cb * CT + ct) + 4 * lg;
            const int cc = col < N ? col : N - 4;                // N % 4 == 0
            if constexpr (NP == 2) o.iw = *reinterpret_cast<const f32x4*>(invw + cc);
            if (bias) o.bs = *reinterpret_cast<const f32x4*>(bias + cc);
            if (post_scale) {
                o.sc = *reinterpret_cast<const f32x4*>(post_scale + cc);
                o.sh = *reinterpret_cast<const f32x4*>(post_shift + cc);
            }
            if (head_w) o.hw = *reinterpret_cast<const f32x4*>(head_w + cc);
            if (gate) {
#pragma unroll
                for (int rt = 0; rt < 2; ++rt) o.gt[rt] = *reinterpret_cast<const f32x4*>(gate + erc[rt] * gate_ld + cc);
            }
        };
        float hpart[2] = {0.f, 0.f};            // head: this lane's share of the row's dot product with head_w (columns of this block)
        float ymax[2] = {0.f, 0.f};             // y_row_bits: the largest |output| of this lane's columns of the row
        EpiOps eo[2];
        epi_load(0, eo[0]);
#pragma unroll
        for (int ct = 0; ct < CT; ++ct) {
            if (ct + 1 < CT) epi_load(ct + 1, eo[(ct + 1) & 1]);
            const EpiOps& o = eo[ct & 1];
            const int col = 16 * (cur.cb * CT + ct) + 4 * lg;      // N % 4 == 0: the lane's four columns are inside or outside together
#pragma unroll
            for (int rt = 0; rt < 2; ++rt) {
                const int64_t r = er[rt];
                f32x4 v = acc[rt][ct];
                if constexpr (NP == 2) v *= o.iw;                  // the columns' weight scales out again
                if constexpr (RS) v *= cur.inv[rt];                // (a lane's four accumulators are four columns of ITS row n)
                if (bias) v += o.bs;
                if (relu) {
#pragma unroll
                    for (int q = 0; q < 4; ++q) v[q] = v[q] > 0.f ? v[q] : 0.f;
                }
                if (post_scale) {               // multiply then add, unfused (dense.hip's affine epilogue)
#pragma unroll
                    for (int q = 0; q < 4; ++q) v[q] = v[q] * o.sc[q] + o.sh[q];
                }
                if (gate) {                     // data gradient through the previous layer's ReLU: y = (x W^T) where gate > 0, else 0
#pragma unroll
                    for (int q = 0; q < 4; ++q) v[q] = o.gt[rt][q] > 0.f ? v[q] : 0.f;
                }
                if (r < M && col < N) {
                    if (Y) *reinterpret_cast<f32x4*>(Y + r * y_ld + col) = v;
                    if (y_row_bits) ymax[rt] = fmaxf(fmaxf(ymax[rt], fmaxf(fabsf(v[0]), fabsf(v[1]))), fmaxf(fabsf(v[2]), fabsf(v[3])));
                    if (head_w) {
                        hpart[rt] += v[0] * o.hw[0];
                        hpart[rt] += v[1] * o.hw[1];
                        hpart[rt] += v[2] * o.hw[2];
                        hpart[rt] += v[3] * o.hw[3];
                    }
                }
                acc[rt][ct] = (f32x4){0.f, 0.f, 0.f, 0.f};
            }
        }
#pragma unroll
        for (int rt = 0; rt < 2; ++rt) {
            const int64_t r = er[rt];
            if (y_row_bits) {                   // (uniform) the output's own row maxima for the NEXT row-scaled kernel: no pass over Y
                float ym = ymax[rt];
                ym = fmaxf(ym, __shfl_xor(ym, 16, 64));
                ym = fmaxf(ym, __shfl_xor(ym, 32, 64));
                if (lg == 0 && r < M) {
                    if (ncb == 1) y_row_bits[r] = __builtin_bit_cast(unsigned int, ym);            // the row is complete in this workgroup
                    else if (ym > 0.f) atomicMax(y_row_bits + r, __builtin_bit_cast(unsigned int, ym));     // one per column block
                }
                if (r < M) run_max = fmaxf(run_max, ym);        // the tensor's maximum: ONE atomic per workgroup, after its last tile
            }
            if (head_w) {                       // the row's other columns of this block live in the other three lane groups
                float hp = hpart[rt];
                hp += __shfl_xor(hp, 16, 64);
                hp += __shfl_xor(hp, 32, 64);
                if (lg == 0 && r < M) head_part[(int64_t)cur.cb * M + r] = hp;
            }
        }
        cur = nxt;
    }
    if (y_all_bits) {                                       // (uniform) a same-address atomic per wave and tile cost 90 us per launch
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) run_max = fmaxf(run_max, __shfl_xor(run_max, o, 64));
        float* wmx = reinterpret_cast<float*>(Wb);          // the W stages are dead behind the loop's last barrier
        if (lane == 0) wmx[wave] = run_max;
        __syncthreads();
        if (tid == 0) {
            float m = wmx[0];
#pragma unroll
            for (int q = 1; q < DB3_NW; ++q) m = fmaxf(m, wmx[q]);
            if (m > 0.f) atomicMax(y_all_bits, __builtin_bit_cast(unsigned int, m));
        }
    }
}

template <int CT, int NP = 3, bool RS = false>
static void launch_dense_bf3(hipStream_t st, const float* X, int64_t x_ld, const unsigned char* img, const float* bias, int relu,
                             const float* ps, const float* psh, const float* gate, int64_t gate_ld, int64_t M, int Kd, int N, int nks, int ncb,
                             float* Y, int64_t y_ld, const float* head_w = nullptr, float* head_part = nullptr,
                             const unsigned int* row_bits = nullptr, unsigned int* y_row_bits = nullptr, unsigned int* y_all_bits = nullptr) {
    const size_t shmem = 2 * (size_t)NP * CT * 1024;
    static LdsOnce once;
    (void)lds_limit(once, 160 * 1024, &dense_bf3_k<CT, NP, RS>);
    const int64_t ntiles = (M + DB3_ROWS - 1) / DB3_ROWS * ncb;
    const int64_t cap = (int64_t)kCUs * (8 / DB3_NW);
    const int64_t nwg = ntiles < cap ? ntiles : cap;           // one persistent workgroup per CU (512 threads, 78-96 KB of LDS)
    hipLaunchKernelGGL((dense_bf3_k<CT, NP, RS>), dim3((unsigned)nwg), dim3(64 * DB3_NW), shmem, st, X, x_ld, img, bias, relu, ps, psh, gate, gate_ld, M, Kd, N,
                       nks, ncb, Y, y_ld, head_w, head_part, row_bits, y_row_bits, y_all_bits);
}

// row_bits[r] = bit pattern of max_k |X[r, k]| (non-negative floats order like their bits); *all_bits = the maximum over all rows.  A lane
// group of 16 per row, float4 loads, eight of a row's loads in flight at once.  all_bits without a zeroing launch and without a same-address
// atomic per row: every workgroup leaves its maximum in block_bits[blockIdx.x] and takes a ticket; the LAST one to arrive reduces the
// block maxima, writes *all_bits and puts the ticket counter back to zero (the caller's word of persistent, initially zero memory).
__global__ __launch_bounds__(256) void row_absmax_k(const float* __restrict__ X, int64_t x_ld, int64_t M, int N, unsigned int* __restrict__ row_bits,
                                                     unsigned int* __restrict__ all_bits, unsigned int* __restrict__ block_bits,
                                                     unsigned int* __restrict__ ticket) {
    const int sub = threadIdx.x & 15;
    float wmx = 0.f;
    for (int64_t r = (int64_t)blockIdx.x * 16 + (threadIdx.x >> 4); r < M; r += (int64_t)gridDim.x * 16) {
        const float* xr = X + r * x_ld;
        float mx = 0.f;
        for (int k0 = 4 * sub; k0 < N; k0 += 512) {
            f32x4 v[8];
#pragma unroll
            for (int q = 0; q < 8; ++q)
                v[q] = k0 + 64 * q < N ? __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(xr + k0 + 64 * q)) : (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int q = 0; q < 8; ++q) mx = fmaxf(fmaxf(mx, fmaxf(fabsf(v[q][0]), fabsf(v[q][1]))), fmaxf(fabsf(v[q][2]), fabsf(v[q][3])));
        }
#pragma unroll
        for (int o = 8; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o, 64));
        if (sub == 0) row_bits[r] = __builtin_bit_cast(unsigned int, mx);
        wmx = fmaxf(wmx, mx);
    }
    if (all_bits) grid_max_bits(wmx, all_bits, block_bits, ticket);      // (uniform)
}

}  // namespace dir

using namespace dir;

extern "C" int64_t dir_dense_bf16x3_image_bytes(int Kd, int N) {
    if (Kd <= 0 || N <= 0) return 0;
    const int CT = db3_ct_for(N);
    const int ncb = ((N + 15) / 16 + CT - 1) / CT, nks = (Kd + 31) / 32;
    return (int64_t)ncb * nks * 3 * CT * 1024;
}

static int dense_pack_strided(const char* name, int pieces, const float* W, int64_t w_rs, int64_t w_cs, int Kd, int N, void* image, int64_t image_bytes,
                              dir_stream_t stream) {
    DIR_CHECK_ARG(W && image && Kd > 0 && N > 0 && w_rs >= 1 && w_cs >= 1, "%s: bad argument (Kd=%d N=%d strides %lld, %lld)", name, Kd, N,
                  (long long)w_rs, (long long)w_cs);
    DIR_CHECK_ARG(aligned16(image) && image_bytes >= dir_dense_bf16x3_image_bytes(Kd, N), "%s: image must be 16-byte aligned and hold "
                  "dir_dense_bf16x3_image_bytes(Kd, N) bytes", name);
    const int CT = db3_ct_for(N);
    const int ncb = ((N + 15) / 16 + CT - 1) / CT, nks = (Kd + 31) / 32;
    const int64_t threads = (int64_t)ncb * nks * CT * 64 * 4;
    if (pieces == 2) {
        unsigned char* base = static_cast<unsigned char*>(image);
        float* invw = reinterpret_cast<float*>(base + (int64_t)ncb * nks * 2 * CT * 1024);      // behind the two piece planes: inside the 3-plane size
        hipLaunchKernelGGL(dense_f16x2_pack_rows_k, dim3((unsigned)((ncb * CT * 16 + 3) / 4)), dim3(256), 0, as_stream(stream), W, w_rs, w_cs, Kd, N, CT,
                           nks, ncb, static_cast<unsigned int*>(image), invw);
    } else
        hipLaunchKernelGGL(dense_bf3_pack_k<3>, dim3((unsigned)((threads + 255) / 256)), dim3(256), 0, as_stream(stream), W, w_rs, w_cs, Kd, N, CT, nks,
                           ncb, static_cast<unsigned int*>(image));
    DIR_CHECK_LAUNCH(name);
    return DIR_OK;
}

extern "C" int dir_dense_bf16x3_pack_strided_f32(const float* W, int64_t w_rs, int64_t w_cs, int Kd, int N, void* image, int64_t image_bytes,
                                                 dir_stream_t stream) {
    return dense_pack_strided("dir_dense_bf16x3_pack_strided_f32", 3, W, w_rs, w_cs, Kd, N, image, image_bytes, stream);
}

// the fp16 x 2 image (dir_dense_bf16x3_image_bytes(Kd, N) bytes hold it) for dir_dense_f16x2_f32 ONLY
extern "C" int dir_dense_f16x2_pack_strided_f32(const float* W, int64_t w_rs, int64_t w_cs, int Kd, int N, void* image, int64_t image_bytes,
                                                dir_stream_t stream) {
    return dense_pack_strided("dir_dense_f16x2_pack_strided_f32", 2, W, w_rs, w_cs, Kd, N, image, image_bytes, stream);
}

extern "C" int dir_dense_bf16x3_pack_f32(const float* W, int64_t w_ld, int Kd, int N, void* image, int64_t image_bytes, dir_stream_t stream) {
    const char* name = "dir_dense_bf16x3_pack_f32";
    DIR_CHECK_ARG(W && image && Kd > 0 && N > 0 && w_ld >= Kd, "%s: bad argument (Kd=%d N=%d w_ld=%lld)", name, Kd, N, (long long)w_ld);
    DIR_CHECK_ARG(aligned16(image) && image_bytes >= dir_dense_bf16x3_image_bytes(Kd, N), "%s: image must be 16-byte aligned and hold "
                  "dir_dense_bf16x3_image_bytes(Kd, N) bytes", name);
    const int CT = db3_ct_for(N);
    const int ncb = ((N + 15) / 16 + CT - 1) / CT, nks = (Kd + 31) / 32;
    const int64_t threads = (int64_t)ncb * nks * CT * 64 * 4;
    hipLaunchKernelGGL(dense_bf3_pack_k<3>, dim3((unsigned)((threads + 255) / 256)), dim3(256), 0, as_stream(stream), W, w_ld, (int64_t)1, Kd, N, CT, nks,
                       ncb, static_cast<unsigned int*>(image));
    DIR_CHECK_LAUNCH(name);
    return DIR_OK;
}

static int dense_run(const char* name, int pieces, const float* X, int64_t x_ld, const void* image, const float* bias, int act, const float* post_scale,
                     const float* post_shift, const float* gate, int64_t gate_ld, int64_t M, int Kd, int N, float* Y, int64_t y_ld,
                     dir_stream_t stream, const unsigned int* row_bits = nullptr, unsigned int* y_row_bits = nullptr,
                     unsigned int* y_all_bits = nullptr) {
    DIR_CHECK_ARG(M >= 0 && Kd > 0 && N > 0 && x_ld >= Kd && y_ld >= N, "%s: bad shape", name);
    DIR_CHECK_ARG(act == DIR_ACT_NONE || act == DIR_ACT_RELU, "%s: act=%d", name, act);
    DIR_CHECK_ARG((post_scale == nullptr) == (post_shift == nullptr), "%s: post_scale and post_shift come together", name);
    DIR_CHECK_ARG(!gate || gate_ld >= N, "%s: gate [M, N] with gate_ld >= N", name);
    if (M == 0) return DIR_OK;
    DIR_CHECK_ARG(X && image && Y, "%s: null pointer", name);
    if ((Kd & 3) || (N & 3) || (x_ld & 3) || (y_ld & 3) || (gate && (gate_ld & 3)) || !aligned16(X) || !aligned16(image) || !aligned16(Y) ||
        (bias && !aligned16(bias)) || (post_scale && (!aligned16(post_scale) || !aligned16(post_shift))) || (gate && !aligned16(gate)))
        return fail(DIR_E_UNSUPPORTED, "%s: Kd, N and the row strides must be multiples of 4 and every operand 16-byte aligned (Kd=%d N=%d)",
                    name, Kd, N);
    const int CT = db3_ct_for(N);
    const int ncb = ((N + 15) / 16 + CT - 1) / CT, nks = (Kd + 31) / 32;
    hipStream_t st = as_stream(stream);
    const unsigned char* img = static_cast<const unsigned char*>(image);
    const int relu = act == DIR_ACT_RELU;
    if (pieces == 2 && row_bits) {
        if (CT == 8) launch_dense_bf3<8, 2, true>(st, X, x_ld, img, bias, relu, post_scale, post_shift, gate, gate_ld, M, Kd, N, nks, ncb, Y, y_ld, nullptr, nullptr, row_bits, y_row_bits, y_all_bits);
        else if (CT == 13) launch_dense_bf3<13, 2, true>(st, X, x_ld, img, bias, relu, post_scale, post_shift, gate, gate_ld, M, Kd, N, nks, ncb, Y, y_ld, nullptr, nullptr, row_bits, y_row_bits, y_all_bits);
        else launch_dense_bf3<16, 2, true>(st, X, x_ld, img, bias, relu, post_scale, post_shift, gate, gate_ld, M, Kd, N, nks, ncb, Y, y_ld, nullptr, nullptr, row_bits, y_row_bits, y_all_bits);
    } else if (pieces == 2) {
        if (CT == 8) launch_dense_bf3<8, 2>(st, X, x_ld, img, bias, relu, post_scale, post_shift, gate, gate_ld, M, Kd, N, nks, ncb, Y, y_ld);
        else if (CT == 13) launch_dense_bf3<13, 2>(st, X, x_ld, img, bias, relu, post_scale, post_shift, gate, gate_ld, M, Kd, N, nks, ncb, Y, y_ld);
        else launch_dense_bf3<16, 2>(st, X, x_ld, img, bias, relu, post_scale, post_shift, gate, gate_ld, M, Kd, N, nks, ncb, Y, y_ld);
    } else if (CT == 8) launch_dense_bf3<8>(st, X, x_ld, img, bias, relu, post_scale, post_shift, gate, gate_ld, M, Kd, N, nks, ncb, Y, y_ld);
    else if (CT == 13) launch_dense_bf3<13>(st, X, x_ld, img, bias, relu, post_scale, post_shift, gate, gate_ld, M, Kd, N, nks, ncb, Y, y_ld);
    else launch_dense_bf3<16>(st, X, x_ld, img, bias, relu, post_scale, post_shift, gate, gate_ld, M, Kd, N, nks, ncb, Y, y_ld);
    DIR_CHECK_LAUNCH(name);
    return DIR_OK;
}

extern "C" int dir_dense_bf16x3_f32(const float* X, int64_t x_ld, const void* image, const float* bias, int act, const float* post_scale,
                                    const float* post_shift, const float* gate, int64_t gate_ld, int64_t M, int Kd, int N, float* Y,
                                    int64_t y_ld, dir_stream_t stream) {
    return dense_run("dir_dense_bf16x3_f32", 3, X, x_ld, image, bias, act, post_scale, post_shift, gate, gate_ld, M, Kd, N, Y, y_ld, stream);
}

extern "C" int dir_dense_f16x2_f32(const float* X, int64_t x_ld, const void* image, const float* bias, int act, const float* post_scale,
                                   const float* post_shift, int64_t M, int Kd, int N, float* Y, int64_t y_ld, dir_stream_t stream) {
    return dense_run("dir_dense_f16x2_f32", 2, X, x_ld, image, bias, act, post_scale, post_shift, nullptr, 0, M, Kd, N, Y, y_ld, stream);
}

// fp16 x 2 for an X of unknown magnitude (the backward's data gradient: X = dL/dy, image = W^T): every row of X scaled by a power of two from
// row_bits (dir_row_absmax_bits_f32), exact; gate as in dir_dense_bf16x3_f32
extern "C" int dir_dense_f16x2_rows_f32(const float* X, int64_t x_ld, const void* image, const float* bias, int act, const float* post_scale,
                                        const float* post_shift, const float* gate, int64_t gate_ld, int64_t M, int Kd, int N, float* Y,
                                        int64_t y_ld, const unsigned int* row_bits, unsigned int* y_row_bits_out, unsigned int* y_all_bits_out,
                                        dir_stream_t stream) {
    DIR_CHECK_ARG(row_bits || M == 0, "dir_dense_f16x2_rows_f32: row_bits is null");
    DIR_CHECK_ARG((y_row_bits_out == nullptr) == (y_all_bits_out == nullptr), "dir_dense_f16x2_rows_f32: y_row_bits_out and y_all_bits_out come together");
    if (y_row_bits_out && M > 0 && N > 0) {               // what the epilogue's atomicMax needs zeroed (one column block: plain stores of the rows' bits)
        const int CT = db3_ct_for(N);
        const int ncb = ((N + 15) / 16 + CT - 1) / CT;
        hipStream_t st = as_stream(stream);
        hipError_t e;
        if (ncb > 1 && y_all_bits_out == y_row_bits_out + M)            // one allocation (ops._out_bits): one launch
            e = zero_async(y_row_bits_out, (size_t)(M + 1) * sizeof(unsigned int), st);
        else {
            e = zero_async(y_all_bits_out, sizeof(unsigned int), st);
            if (e == hipSuccess && ncb > 1) e = zero_async(y_row_bits_out, (size_t)M * sizeof(unsigned int), st);
        }
        if (e != hipSuccess) return fail(DIR_E_HIP, "dir_dense_f16x2_rows_f32: zeroing failed");
    }
    return dense_run("dir_dense_f16x2_rows_f32", 2, X, x_ld, image, bias, act, post_scale, post_shift, gate, gate_ld, M, Kd, N, Y, y_ld, stream, row_bits,
                     y_row_bits_out, y_all_bits_out);
}

extern "C" int dir_row_absmax_workspace_words(void) { return 1 + kCUs * 8; }      // [ticket | block maxima]

extern "C" int dir_row_absmax_bits_f32(const float* X, int64_t x_ld, int64_t M, int N, unsigned int* row_bits, unsigned int* all_bits,
                                       unsigned int* workspace, dir_stream_t stream) {
    const char* name = "dir_row_absmax_bits_f32";
    DIR_CHECK_ARG(M >= 0 && N > 0 && x_ld >= N, "%s: bad shape", name);
    DIR_CHECK_ARG(!all_bits || workspace, "%s: all_bits needs the workspace", name);
    hipStream_t st = as_stream(stream);
    if (M == 0) {
        if (all_bits && zero_async(all_bits, sizeof(unsigned int), st) != hipSuccess) return fail(DIR_E_HIP, "%s: zeroing failed", name);
        return DIR_OK;
    }
    DIR_CHECK_ARG(X && row_bits, "%s: null pointer", name);
    if ((N & 3) || (x_ld & 3) || !aligned16(X)) return fail(DIR_E_UNSUPPORTED, "%s: N and x_ld must be multiples of 4, X 16-byte aligned", name);
    hipLaunchKernelGGL(row_absmax_k, dim3(grid_for((M + 15) / 16, 8)), dim3(256), 0, st, X, x_ld, M, N, row_bits, all_bits,
                       workspace ? workspace + 1 : nullptr, workspace);
    DIR_CHECK_LAUNCH(name);
    return DIR_OK;
}

// The layer with the head of the tower's output folded into its epilogue (DCN's last deep layer, DeepCrossNetwork.py:136-137: the deep
// branch's share of the final dense(1)): head_part[cb][r] = sum over column block cb of y[r, c] * head_w[c]  (ncb =
// dir_dense_bf16x3_head_blocks(N) blocks, added by the caller in block order: a fixed order).  Y may be NULL: the activation then
// never reaches memory.
extern "C" int dir_dense_bf16x3_head_blocks(int N) {
    if (N <= 0) return 0;
    const int CT = db3_ct_for(N);
    return ((N + 15) / 16 + CT - 1) / CT;
}

static int dense_head_run(const char* name, int pieces, const float* X, int64_t x_ld, const void* image, const float* bias, int act,
                          const float* post_scale, const float* post_shift, int64_t M, int Kd, int N, const float* head_w, float* Y, int64_t y_ld,
                          float* head_part, dir_stream_t stream) {
    DIR_CHECK_ARG(M >= 0 && Kd > 0 && N > 0 && x_ld >= Kd && (!Y || y_ld >= N), "%s: bad shape", name);
    DIR_CHECK_ARG(act == DIR_ACT_NONE || act == DIR_ACT_RELU, "%s: act=%d", name, act);
    DIR_CHECK_ARG((post_scale == nullptr) == (post_shift == nullptr), "%s: post_scale and post_shift come together", name);
    if (M == 0) return DIR_OK;
    DIR_CHECK_ARG(X && image && head_w && head_part, "%s: null pointer", name);
    if ((Kd & 3) || (N & 3) || (x_ld & 3) || (Y && ((y_ld & 3) || !aligned16(Y))) || !aligned16(X) || !aligned16(image) || !aligned16(head_w) ||
        (bias && !aligned16(bias)) || (post_scale && (!aligned16(post_scale) || !aligned16(post_shift))))
        return fail(DIR_E_UNSUPPORTED, "%s: Kd, N and the row strides must be multiples of 4 and every operand 16-byte aligned (Kd=%d N=%d)",
                    name, Kd, N);
    const int CT = db3_ct_for(N);
    const int ncb = ((N + 15) / 16 + CT - 1) / CT, nks = (Kd + 31) / 32;
    hipStream_t st = as_stream(stream);
    const unsigned char* img = static_cast<const unsigned char*>(image);
    const int relu = act == DIR_ACT_RELU;
    if (pieces == 2) {
        if (CT == 8) launch_dense_bf3<8, 2>(st, X, x_ld, img, bias, relu, post_scale, post_shift, nullptr, 0, M, Kd, N, nks, ncb, Y, y_ld, head_w, head_part);
        else if (CT == 13) launch_dense_bf3<13, 2>(st, X, x_ld, img, bias, relu, post_scale, post_shift, nullptr, 0, M, Kd, N, nks, ncb, Y, y_ld, head_w, head_part);
        else launch_dense_bf3<16, 2>(st, X, x_ld, img, bias, relu, post_scale, post_shift, nullptr, 0, M, Kd, N, nks, ncb, Y, y_ld, head_w, head_part);
    } else if (CT == 8) launch_dense_bf3<8>(st, X, x_ld, img, bias, relu, post_scale, post_shift, nullptr, 0, M, Kd, N, nks, ncb, Y, y_ld, head_w, head_part);
    else if (CT == 13) launch_dense_bf3<13>(st, X, x_ld, img, bias, relu, post_scale, post_shift, nullptr, 0, M, Kd, N, nks, ncb, Y, y_ld, head_w, head_part);
    else launch_dense_bf3<16>(st, X, x_ld, img, bias, relu, post_scale, post_shift, nullptr, 0, M, Kd, N, nks, ncb, Y, y_ld, head_w, head_part);
    DIR_CHECK_LAUNCH(name);
    return DIR_OK;
}

extern "C" int dir_dense_bf16x3_head_f32(const float* X, int64_t x_ld, const void* image, const float* bias, int act, const float* post_scale,
                                         const float* post_shift, int64_t M, int Kd, int N, const float* head_w, float* Y, int64_t y_ld,
                                         float* head_part, dir_stream_t stream) {
    return dense_head_run("dir_dense_bf16x3_head_f32", 3, X, x_ld, image, bias, act, post_scale, post_shift, M, Kd, N, head_w, Y, y_ld, head_part, stream);
}

extern "C" int dir_dense_f16x2_head_f32(const float* X, int64_t x_ld, const void* image, const float* bias, int act, const float* post_scale,
                                        const float* post_shift, int64_t M, int Kd, int N, const float* head_w, float* Y, int64_t y_ld,
                                        float* head_part, dir_stream_t stream) {
    return dense_head_run("dir_dense_f16x2_head_f32", 2, X, x_ld, image, bias, act, post_scale, post_shift, M, Kd, N, head_w, Y, y_ld, head_part, stream);
}
