// dense_dw_bf3.hip -- weight gradient of a dense layer, dW[n, k] = sum_r g[r, n] * x[r, k], on the bf16 matrix pipe with fp32-equivalent
// arithmetic ("bf16 x 3": dense_bf3.hip, cin_dw_bf3.hip).
//
// Reference: the hidden layers of dnn_logit_fn (models/DeepFM/deepFM.py:293-300), _deep_architecture
// (models/DeepCrossNetwork/DeepCrossNetwork.py:394-399) and _base_model (models/ESMM/ESMM.py:137-142); the reference trains through
// TensorFlow autodiff of tf.layers.dense, whose kernel gradient is this product (g = dL/d(pre-activation) [M, N], x = the layer's
// input [M, K], M = batch rows).
//
// The REDUCTION runs over the batch rows, the slow index of both operands -- the opposite of what the MFMA operand layout wants (a
// lane's 8 k-slots would be 8 consecutive rows of one column).  So both operands go through a transposing stage: per step of 32 rows
// a thread loads 8 consecutive rows of a QUAD of columns (eight 16-byte loads; a wave reads a contiguous kilobyte per row), i.e. four
// "octets" -- 8 consecutive rows of ONE column --, splits each octet into its three bf16 pieces in registers (52 VALU instructions) and
// writes three 16-byte operand fragments into LDS in the operand layout [piece][tile][slot][8] (XOR-swizzled slots: conflict-free
// ds_write_b128).  After a barrier
// every wave reads its A fragments (two 16-column tiles of g: the wave's 32 output rows) once and the B fragments of the block's x
// tiles one tile ahead of their 12 MFMAs.  One LDS buffer, two barriers per step: the split phase and the MFMA phase of the two waves
// of a SIMD do not overlap anyway (cin_dw_bf3.hip); the raw loads of step s+1 are issued before step s's MFMAs and land
// under them.
//
// Work split.  A work item is (block of 256 columns of g = 16 tiles, block of KB tiles of x, span of rows); workgroup of 8 waves (two
// per SIMD): wave w owns g tiles w and w+8 (32 output rows) x the KB x tiles = 2*KB accumulators.  KB = 13 / 16 / 8, whichever pads K
// least.  Every element is split once per workgroup that uses it: about 1.2 VALU instructions per MFMA at 400 x 416.  Spans leave
// partial sums part[span][N][K]; dense_dw_bf3_reduce_k adds them in span order (bitwise reproducible, no atomics).
#include "common.hpp"

#ifndef DDW_CHAIN
#define DDW_CHAIN 1      // see dense_bf3.hip (DB3_CHAIN): 65 536 x 400 x 416: 175 -> 169 us, 1024 x 1024: 786 -> 755 us
#endif

namespace dir {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));
typedef unsigned int u32x4_t __attribute__((ext_vector_type(4)));

constexpr int DDW_NT = 16;           // g tiles per block (two per wave)

__device__ __forceinline__ unsigned int ddw_pk(float a, float b) {      // v_cvt_pk_bf16_f32 (round to nearest even), a in the low half
    typedef __bf16 pk2_t __attribute__((ext_vector_type(2)));
    const pk2_t v = {(__bf16)a, (__bf16)b};
    unsigned int w = __builtin_bit_cast(unsigned int, v);
    asm("" : "+v"(w));      // (empty: hides w's origin so that float(bf16(a)) is formed by a shift, not a second convert; see cin_bf3.hip)
    return w;
}
__device__ __forceinline__ void ddw_split8(const float (&x)[8], bf16x8_t (&p)[3]) {     // 8 values -> three bf16x8 fragments that sum to them
    unsigned int w[3][4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const f32x2 v = {x[2 * i], x[2 * i + 1]};
        w[0][i] = ddw_pk(v[0], v[1]);
        const f32x2 r = v - (f32x2){__builtin_bit_cast(float, w[0][i] << 16), __builtin_bit_cast(float, w[0][i] & 0xffff0000u)};
        w[1][i] = ddw_pk(r[0], r[1]);
        const f32x2 t = r - (f32x2){__builtin_bit_cast(float, w[1][i] << 16), __builtin_bit_cast(float, w[1][i] & 0xffff0000u)};
        w[2][i] = ddw_pk(t[0], t[1]);
    }
#pragma unroll
    for (int q = 0; q < 3; ++q) p[q] = __builtin_bit_cast(bf16x8_t, (u32x4_t){w[q][0], w[q][1], w[q][2], w[q][3]});
}

// ---- round 4: "fp16 x 2" (cin_bf3.hip states the arithmetic): g, a gradient of unknown magnitude, is multiplied by ONE power of two for the
// whole tensor (largest |element| into [2^14, 2^15): exact) before its split, and the reduce pass takes the scale out of dW (the bias
// gradient sums the raw g); x is split as in the fp16 x 2 forward (the caller vouches for O(1) activations).
typedef _Float16 f16x8_t __attribute__((ext_vector_type(8)));
__device__ __forceinline__ unsigned int ddw_pk_h(float a, float b) {
    typedef _Float16 h2_t __attribute__((ext_vector_type(2)));
    const h2_t v = {(_Float16)a, (_Float16)b};
    unsigned int w = __builtin_bit_cast(unsigned int, v);
    asm("" : "+v"(w));
    return w;
}
__device__ __forceinline__ void ddw_split8h(const float (&x)[8], f16x8_t (&p)[2]) {
    typedef _Float16 h2_t __attribute__((ext_vector_type(2)));
    unsigned int w[2][4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const f32x2 v = {x[2 * i], x[2 * i + 1]};
        w[0][i] = ddw_pk_h(v[0], v[1]);
        const h2_t h = __builtin_bit_cast(h2_t, w[0][i]);
        const f32x2 r = v - (f32x2){(float)h[0], (float)h[1]};
        w[1][i] = ddw_pk_h(r[0], r[1]);
    }
#pragma unroll
    for (int q = 0; q < 2; ++q) p[q] = __builtin_bit_cast(f16x8_t, (u32x4_t){w[q][0], w[q][1], w[q][2], w[q][3]});
}
template <int NP> struct DdwPc;
template <> struct DdwPc<3> {
    using op_t = bf16x8_t;
    __device__ static __forceinline__ void split8(const float (&x)[8], op_t (&p)[3]) { ddw_split8(x, p); }
    __device__ static __forceinline__ f32x4 mma(const op_t (&a)[3], const op_t (&b)[3], f32x4 c) {
        c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[0], b[2], c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[2], b[0], c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[1], b[1], c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[0], b[1], c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[1], b[0], c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[0], b[0], c, 0, 0, 0);
        return c;
    }
};
template <> struct DdwPc<2> {
    using op_t = f16x8_t;
    __device__ static __forceinline__ void split8(const float (&x)[8], op_t (&p)[2]) { ddw_split8h(x, p); }
    __device__ static __forceinline__ f32x4 mma(const op_t (&a)[2], const op_t (&b)[2], f32x4 c) {
        c = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[1], b[0], c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[0], b[1], c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[0], b[0], c, 0, 0, 0);
        return c;
    }
};
__device__ __forceinline__ float ddw_scale(unsigned int bits, bool inverse) {     // largest |element| into [2^14, 2^15); k clamped to +-100
    int k = 141 - (int)((bits >> 23) & 0xffu);
    k = k > 100 ? 100 : (k < -100 ? -100 : k);
    return __builtin_bit_cast(float, (unsigned int)(inverse ? 127 - k : 127 + k) << 23);
}

__host__ __device__ inline int ddw_kb_for(int K) {     // x tiles per block: the choice that pads K least (ties: the wider block)
    const int tiles = (K + 15) / 16;
    int best = 16, pad = (tiles + 15) / 16 * 16;
    const int p13 = (tiles + 12) / 13 * 13, p8 = (tiles + 7) / 8 * 8;
    if (p13 < pad) { best = 13; pad = p13; }
    if (p8 < pad) { best = 8; pad = p8; }
    return best;
}

struct DdwPlan { int KB, nnb, nkb, nspan; int64_t steps, steps_per_span; };
static DdwPlan ddw_plan(int64_t M, int N, int K) {
    DdwPlan p;
    p.KB = ddw_kb_for(K);
    p.nnb = (N + 16 * DDW_NT - 1) / (16 * DDW_NT);
    p.nkb = ((K + 15) / 16 + p.KB - 1) / p.KB;
    p.steps = (M + 31) / 32;
    int ns = kCUs / (p.nnb * p.nkb);
    if (ns < 1) ns = 1;
    if ((int64_t)ns > p.steps) ns = (int)(p.steps > 0 ? p.steps : 1);
    p.nspan = ns;
    p.steps_per_span = (p.steps + ns - 1) / ns;
    return p;
}

template <int KB, int NP = 3>
__global__ __launch_bounds__(512, 1) void dense_dw_bf3_k(const float* __restrict__ g, int64_t g_ld, const float* __restrict__ x, int64_t x_ld,
                                                          int64_t M, int N, int K, int nkb, int nspan, int64_t steps_per_span, int64_t steps,
                                                          float* __restrict__ part, float* __restrict__ bpart /* [nspan][N] or NULL */,
                                                          const unsigned int* __restrict__ gbits = nullptr /* NP == 2: bit pattern of max |g| */,
                                                          const unsigned int* __restrict__ xbits = nullptr /* NP == 2, optional: bit pattern of max |x| */) {
    using Pc = DdwPc<NP>;
    using op_t = typename Pc::op_t;
    constexpr int TT = DDW_NT + KB;                   // tiles staged per step: 16 of g, KB of x
    extern __shared__ __attribute__((aligned(16))) unsigned char ddw_smem[];      // [NP pieces][TT tiles][64 lanes][8 halves]
    float gsc = 1.f;
    if constexpr (NP == 2) gsc = ddw_scale(*gbits, false);
    float xsc = 1.f;                                  // x scaled the same way when its bound is known (any magnitudes; without it |x| < 65 504)
    if constexpr (NP == 2) xsc = xbits ? ddw_scale(*xbits, false) : 1.f;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, lg = lane >> 4, ln = lane & 15;
    int q = blockIdx.x;
    const int span = q % nspan; q /= nspan;
    const int kb = q % nkb;
    const int nb = q / nkb;
    const int n0 = 16 * DDW_NT * nb, k0 = 16 * KB * kb;
    const int ntg = min(DDW_NT, (N - n0 + 15) / 16), ktx = min(KB, (K - k0 + 15) / 16);
    const int64_t s_begin = (int64_t)span * steps_per_span;
    int64_t s_end = s_begin + steps_per_span;
    if (s_end > steps) s_end = steps;

    // this thread's four octets of a step: one QUAD of columns (4 q .. 4 q + 3) x one row octet ro.  Waves 0..3 stage the g strip (256
    // columns = 64 quads, ro = wave), waves 4..7 the x strip (4 KB quads on the first lanes, ro = wave - 4): eight 16-byte loads per
    // thread whose row base is a scalar (a wave reads a contiguous kilobyte per row).  The fragment of column c = 4 q + j goes to tile
    // c / 16, slot ((c % 16) + 16 ro) ^ (tile & 3): the XOR spreads a store instruction's 64 lanes (16 tiles x 4 quads, 64 bytes apart)
    // over all LDS banks; readers apply the same XOR to their lane.  Columns past N / K are read as they come (N, K and the row strides
    // are multiples of 4 here, so a quad is inside the row): they only feed outputs that are never stored.  (The first version staged
    // single columns with scalar loads: 32 loads per thread and step, 39 of its 200 us.)
    constexpr int XQ = 4 * KB;
    const bool isg = wave < 4;
    const int ro = __builtin_amdgcn_readfirstlane(isg ? wave : wave - 4);
    const bool active = isg || lane < XQ;
    const int qd = active ? lane : 0;
    const float* sbase = isg ? g + n0 : x + k0;
    const int64_t sld = isg ? g_ld : x_ld;
    const int width = isg ? N - n0 : K - k0;                                          // valid columns of the strip
    const int soff = (4 * qd < width) ? 4 * qd : 0;
    const int stile = (isg ? 0 : DDW_NT) + (qd >> 2);
    const int sdst = stile * 1024 + (4 * (qd & 3) + 16 * ro) * 16;                    // + ((j ^ (tile & 3)) * 16): j is the slot's bits 0..1
    f32x4 raw[8];
    // the bias gradient db[n] = sum_r g[r, n] rides along: the workgroups of the first x block add up the g rows they stage anyway
    const bool want_b = bpart != nullptr && kb == 0 && isg;
    f32x4 bsum = {0.f, 0.f, 0.f, 0.f};
    auto load_raw = [&](int64_t s) {
        const int64_t r0 = 32 * s + 8 * ro;                                           // scalar
#pragma unroll
        for (int e = 0; e < 8; ++e)
            raw[e] = (active && r0 + e < M) ? *reinterpret_cast<const f32x4*>(sbase + (r0 + e) * sld + soff) : (f32x4){0.f, 0.f, 0.f, 0.f};
    };
    auto put_all = [&]() {
        if (want_b) {
#pragma unroll
            for (int e = 0; e < 8; ++e) bsum += raw[e];          // rows ascending
        }
        if (active) {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                float v[8];
#pragma unroll
                for (int e = 0; e < 8; ++e) v[e] = NP == 2 ? raw[e][j] * (isg ? gsc : xsc) : raw[e][j];
                op_t p[NP];
                Pc::split8(v, p);
                const int off = sdst + ((j ^ (stile & 3)) * 16);
#pragma unroll
                for (int pc = 0; pc < NP; ++pc) *reinterpret_cast<op_t*>(ddw_smem + pc * (TT * 1024) + off) = p[pc];
            }
        }
    };

    f32x4 acc[2][KB];
#pragma unroll
    for (int rt = 0; rt < 2; ++rt)
#pragma unroll
        for (int kt = 0; kt < KB; ++kt) acc[rt][kt] = (f32x4){0.f, 0.f, 0.f, 0.f};
    const bool has0 = wave < ntg, has1 = wave + 8 < ntg;

    // one step: split the raw octets into LDS, barrier, refill the same registers with step s + 1 (the loads land under the MFMAs;
    // two steps of loads in flight, in a second register set, measured slower: 227 against 200 us at 65 536 x 400 x 416), MFMAs, barrier
    if (s_begin < s_end) load_raw(s_begin);
    for (int64_t s = s_begin; s < s_end; ++s) {
        put_all();
        __syncthreads();
        if (s + 1 < s_end) load_raw(s + 1);
        if (has0) {
            // a lane's fragment of tile t sits in slot lane ^ (t & 3)
            const unsigned char* fa = ddw_smem + ((lane ^ (wave & 3)) * 16);
            const unsigned char* fb[4] = {ddw_smem + lane * 16, ddw_smem + ((lane ^ 1) * 16), ddw_smem + ((lane ^ 2) * 16), ddw_smem + ((lane ^ 3) * 16)};
            op_t a[2][NP];
#pragma unroll
            for (int pc = 0; pc < NP; ++pc) {
                a[0][pc] = *reinterpret_cast<const op_t*>(fa + pc * (TT * 1024) + wave * 1024);
                a[1][pc] = *reinterpret_cast<const op_t*>(fa + pc * (TT * 1024) + (has1 ? wave + 8 : wave) * 1024);
            }
            op_t bq[2][NP];
#pragma unroll
            for (int pc = 0; pc < NP; ++pc) bq[0][pc] = *reinterpret_cast<const op_t*>(fb[0] + pc * (TT * 1024) + DDW_NT * 1024);
#pragma unroll
            for (int kt = 0; kt < KB; ++kt) {
                if (kt < ktx) {
                    if (kt + 1 < KB) {
#pragma unroll
                        for (int pc = 0; pc < NP; ++pc)
                            bq[(kt + 1) & 1][pc] = *reinterpret_cast<const op_t*>(fb[(kt + 1) & 3] + pc * (TT * 1024) + (DDW_NT + kt + 1) * 1024);
                    }
                    const op_t (&b)[NP] = bq[kt & 1];
#pragma unroll
                    for (int rt = 0; rt < 2; ++rt) {
                        acc[rt][kt] = Pc::mma(a[rt], b, acc[rt][kt]);
#if DDW_CHAIN
                        __builtin_amdgcn_sched_barrier(0);      // one dependent chain per accumulator (dense_bf3.hip: DB3_CHAIN)
#endif
                    }
                }
            }
        }
        __syncthreads();
    }
    if (bpart != nullptr && kb == 0) {                           // (uniform) the four row octets' sums, added in octet order
        f32x4* bs = reinterpret_cast<f32x4*>(ddw_smem);          // the fragments are dead behind the loop's last barrier
        if (isg) bs[ro * 64 + lane] = bsum;
        __syncthreads();
        if (wave == 0 && n0 + 4 * lane < N) {
            const f32x4 t = ((bs[lane] + bs[64 + lane]) + bs[128 + lane]) + bs[192 + lane];
            *reinterpret_cast<f32x4*>(bpart + (int64_t)span * N + n0 + 4 * lane) = t;
        }
    }
    // partial sums: part[span][n][k]   (C/D map: col = lane & 15, row = 4 (lane >> 4) + reg)
    float* pp = part + (int64_t)span * N * K;
#pragma unroll
    for (int rt = 0; rt < 2; ++rt) {
        if (rt == 0 ? has0 : has1) {
#pragma unroll
            for (int kt = 0; kt < KB; ++kt) {
                const int k = k0 + 16 * kt + ln;
                if (kt < ktx && k < K) {
#pragma unroll
                    for (int qq = 0; qq < 4; ++qq) {
                        const int n = n0 + 16 * (wave + 8 * rt) + 4 * lg + qq;
                        if (n < N) pp[(int64_t)n * K + k] = acc[rt][kt][qq];
                    }
                }
            }
        }
    }
}

// dW[n, k] (row stride dw_ld) = sum over spans in a fixed order: a workgroup takes 256 / SG consecutive elements, span group q adds spans
// q, q + SG, ... (ascending) and the SG group sums are added in group order through LDS.  (One thread per element walking all spans is a
// chain of nspan dependent loads: 240 us for the 512 spans of an 80 x 64 gradient.)
template <int SG>
__global__ __launch_bounds__(256) void dense_dw_bf3_reduce_k(const float* __restrict__ part, const float* __restrict__ bpart, int N, int K,
                                                            int nspan, float* __restrict__ dW, int64_t dw_ld, float* __restrict__ db,
                                                            const unsigned int* __restrict__ gbits = nullptr /* fp16 x 2: the scale to take out of dW */,
                                                            const unsigned int* __restrict__ xbits = nullptr /* ... and x's, when x was scaled too */) {
    constexpr int EPB = 256 / SG;
    const float inv = gbits ? ddw_scale(*gbits, true) : 1.f;
    const float invx = xbits ? ddw_scale(*xbits, true) : 1.f;      // applied one after the other: each is a normal number, their product need not be
    __shared__ float red[SG][EPB];
    const int q = threadIdx.x / EPB, el = threadIdx.x % EPB;
    const int64_t total = (int64_t)N * K, all = total + (db != nullptr ? N : 0);
    for (int64_t e0 = (int64_t)blockIdx.x * EPB; e0 < all; e0 += (int64_t)gridDim.x * EPB) {
        const int64_t e = e0 + el;
        float s = 0.f;
        if (e < total) {
            for (int sp = q; sp < nspan; sp += SG) s += part[(int64_t)sp * total + e];
        } else if (e < all) {
            for (int sp = q; sp < nspan; sp += SG) s += bpart[(int64_t)sp * N + (e - total)];
        }
        red[q][el] = s;
        __syncthreads();
        if (q == 0 && e < all) {
            float t = red[0][el];
#pragma unroll
            for (int i = 1; i < SG; ++i) t += red[i][el];
            if (e < total) dW[(e / K) * dw_ld + (e % K)] = (t * inv) * invx;
            else db[e - total] = t;
        }
        __syncthreads();
    }
}

static void ddw_reduce(const float* part, const float* bpart, int N, int K, int nspan, float* dW, int64_t dw_ld, float* db, hipStream_t st,
                       const unsigned int* gbits = nullptr, const unsigned int* xbits = nullptr) {
    const int64_t all = (int64_t)N * K + (db ? N : 0);
    if (all <= 65536)
        hipLaunchKernelGGL(dense_dw_bf3_reduce_k<16>, dim3(grid_for((all + 15) / 16)), dim3(256), 0, st, part, bpart, N, K, nspan, dW, dw_ld, db, gbits, xbits);
    else
        hipLaunchKernelGGL(dense_dw_bf3_reduce_k<4>, dim3(grid_for((all + 63) / 64)), dim3(256), 0, st, part, bpart, N, K, nspan, dW, dw_ld, db, gbits, xbits);
}

// ---- small gradients (N <= 128, K <= 256): fp32 FMAs on a register tile ---------------------------------------------------------------------
// The same product for outputs of at most 128 x 256 -- the per-sample term of the DIN unit's first layer (csrc/din_bwd_rows.hip: S^T a,
// 80 x 64) and the narrow last layers of the towers.  The library runs these tall-and-skinny TN GEMMs at 200 us (65 536 rows, 80 x 64:
// 37 MB of operands); the MFMA kernel above at 137 (its stage-split-multiply structure is mostly overhead here).  A workgroup owns a
// span of rows; per step it stages 32 rows of both operands in LDS (16-byte loads) and every thread walks them with a TN x TK register
// tile of fp32 FMAs (rows ascending: exact fp32 products, one rounding per accumulate), two 16-byte LDS reads per operand and row at
// most.  Span partials and the final sums as above.
template <int TN, int TK>
__global__ __launch_bounds__(256) void dense_dw_small_k(const float* __restrict__ g, int64_t g_ld, const float* __restrict__ x, int64_t x_ld,
                                                        int64_t M, int N, int K, int64_t rows_per_span, float* __restrict__ part,
                                                        float* __restrict__ bpart) {
    constexpr int R = 32;
    __shared__ __attribute__((aligned(16))) float sg[R][128 + 4], sx[R][256 + 4];
    const int tid = threadIdx.x;
    const int ntk = (K + TK - 1) / TK, ntn = (N + TN - 1) / TN;
    const int in = tid / ntk, ik = tid % ntk;
    const bool act = in < ntn;
    const int n0 = in * TN, k0 = ik * TK;
    const int64_t r0 = (int64_t)blockIdx.x * rows_per_span, r1 = min(M, r0 + rows_per_span);
    float acc[TN][TK], bs[TN];
#pragma unroll
    for (int i = 0; i < TN; ++i) {
        bs[i] = 0.f;
#pragma unroll
        for (int j = 0; j < TK; ++j) acc[i][j] = 0.f;
    }
    const int gq = N >> 2, xq = K >> 2;                 // float4 per row of each operand
    for (int64_t rb = r0; rb < r1; rb += R) {
        const int rows = (int)min((int64_t)R, r1 - rb);
        for (int e = tid; e < R * gq; e += 256) {
            const int r = e / gq, q = e % gq;
            *reinterpret_cast<f32x4*>(&sg[r][4 * q]) = r < rows ? *reinterpret_cast<const f32x4*>(g + (rb + r) * g_ld + 4 * q) : (f32x4){0.f, 0.f, 0.f, 0.f};
        }
        for (int e = tid; e < R * xq; e += 256) {
            const int r = e / xq, q = e % xq;
            *reinterpret_cast<f32x4*>(&sx[r][4 * q]) = r < rows ? *reinterpret_cast<const f32x4*>(x + (rb + r) * x_ld + 4 * q) : (f32x4){0.f, 0.f, 0.f, 0.f};
        }
        __syncthreads();
        if (act) {
#pragma unroll 4
            for (int r = 0; r < R; ++r) {
                float gv[TN], xv[TK];
#pragma unroll
                for (int i = 0; i < TN; i += 4) {
                    const f32x4 t = *reinterpret_cast<const f32x4*>(&sg[r][n0 + i]);
                    gv[i] = t[0]; gv[i + 1] = t[1]; gv[i + 2] = t[2]; gv[i + 3] = t[3];
                }
#pragma unroll
                for (int j = 0; j < TK; j += 4) {
                    const f32x4 t = *reinterpret_cast<const f32x4*>(&sx[r][k0 + j]);
                    xv[j] = t[0]; xv[j + 1] = t[1]; xv[j + 2] = t[2]; xv[j + 3] = t[3];
                }
#pragma unroll
                for (int i = 0; i < TN; ++i) {
                    bs[i] += gv[i];
#pragma unroll
                    for (int j = 0; j < TK; ++j) acc[i][j] = __builtin_fmaf(gv[i], xv[j], acc[i][j]);
                }
            }
        }
        __syncthreads();
    }
    if (act) {
        float* pp = part + (int64_t)blockIdx.x * N * K;
#pragma unroll
        for (int i = 0; i < TN; ++i) {
            if (n0 + i < N) {
#pragma unroll
                for (int j = 0; j < TK; ++j)
                    if (k0 + j < K) pp[(int64_t)(n0 + i) * K + k0 + j] = acc[i][j];
                if (bpart != nullptr && ik == 0) bpart[(int64_t)blockIdx.x * N + n0 + i] = bs[i];
            }
        }
    }
}

struct DdwSmallPlan { int tn, tk, nspan; int64_t rows_per_span; };
static DdwSmallPlan ddw_small_plan(int64_t M, int N, int K) {
    DdwSmallPlan p;
    p.tn = 4;
    p.tk = ((N + 3) / 4) * ((K + 3) / 4) <= 256 ? 4 : 8;
    if (((N + p.tn - 1) / p.tn) * ((K + p.tk - 1) / p.tk) > 256) p.tn = 8;
    int64_t ns = 2 * kCUs;
    const int64_t steps = (M + 31) / 32;
    if (ns > steps) ns = steps > 0 ? steps : 1;
    p.rows_per_span = ((steps + ns - 1) / ns) * 32;
    p.nspan = (int)((M + p.rows_per_span - 1) / p.rows_per_span);
    if (p.nspan < 1) p.nspan = 1;
    return p;
}

}  // namespace dir

using namespace dir;

extern "C" int64_t dir_dense_dw_small_workspace_bytes(int64_t M, int N, int K) {
    if (M <= 0 || N <= 0 || K <= 0) return 0;
    const DdwSmallPlan p = ddw_small_plan(M, N, K);
    return (int64_t)p.nspan * ((int64_t)N * K + N) * (int64_t)sizeof(float);
}

extern "C" int dir_dense_dw_small_f32(const float* g, int64_t g_ld, const float* x, int64_t x_ld, int64_t M, int N, int K, float* dW,
                                      int64_t dw_ld, float* db, void* workspace, int64_t workspace_bytes, dir_stream_t stream) {
    const char* name = "dir_dense_dw_small_f32";
    DIR_CHECK_ARG(dW && M >= 0 && N > 0 && K > 0 && dw_ld >= K, "%s: bad argument (M=%lld N=%d K=%d dw_ld=%lld)", name, (long long)M, N, K,
                  (long long)dw_ld);
    hipStream_t st = as_stream(stream);
    if (M == 0) {
        if (zero_2d_async(dW, dw_ld * sizeof(float), K * sizeof(float), N, st) != hipSuccess) return fail(DIR_E_HIP, "%s: memset failed", name);
        if (db && zero_async(db, N * sizeof(float), st) != hipSuccess) return fail(DIR_E_HIP, "%s: memset failed", name);
        return DIR_OK;
    }
    DIR_CHECK_ARG(g && x && workspace && g_ld >= N && x_ld >= K, "%s: null pointer or row stride smaller than the width", name);
    if (N > 128 || K > 256 || ((N + 7) / 8) * ((K + 7) / 8) > 256 || N % 4 || K % 4 || g_ld % 4 || x_ld % 4 || !aligned16(g) || !aligned16(x))
        return fail(DIR_E_UNSUPPORTED, "%s: N=%d (<= 128) K=%d (<= 256; at most 256 register tiles of 8 x 8), and N, K, g_ld=%lld, x_ld=%lld "
                    "multiples of 4 with g / x 16-byte aligned", name, N, K, (long long)g_ld, (long long)x_ld);
    DIR_CHECK_ARG(aligned16(workspace) && workspace_bytes >= dir_dense_dw_small_workspace_bytes(M, N, K),
                  "%s: workspace must be 16-byte aligned and hold dir_dense_dw_small_workspace_bytes(M, N, K) bytes", name);
    const DdwSmallPlan p = ddw_small_plan(M, N, K);
    float* part = static_cast<float*>(workspace);
    float* bpart = db ? part + (int64_t)p.nspan * N * K : nullptr;
#define DDW_SMALL(TN_, TK_) hipLaunchKernelGGL((dense_dw_small_k<TN_, TK_>), dim3((unsigned)p.nspan), dim3(256), 0, st, g, g_ld, x, x_ld, M, N, K, \
                                               p.rows_per_span, part, bpart)
    if (p.tn == 4 && p.tk == 4) DDW_SMALL(4, 4);
    else if (p.tn == 4) DDW_SMALL(4, 8);
    else DDW_SMALL(8, 8);
#undef DDW_SMALL
    DIR_CHECK_LAUNCH(name);
    ddw_reduce(part, bpart, N, K, p.nspan, dW, dw_ld, db, st);
    DIR_CHECK_LAUNCH("dense_dw_small reduce");
    return DIR_OK;
}

extern "C" int64_t dir_dense_dw_bf16x3_workspace_bytes(int64_t M, int N, int K) {
    if (M <= 0 || N <= 0 || K <= 0) return 0;
    const DdwPlan p = ddw_plan(M, N, K);
    return (int64_t)p.nspan * ((int64_t)N * K + N) * (int64_t)sizeof(float);      // span partials of dW, then of db
}

static int ddw_run(const char* name, int np, const float* g, int64_t g_ld, const float* x, int64_t x_ld, int64_t M, int N, int K, float* dW,
                   int64_t dw_ld, float* db, void* workspace, int64_t workspace_bytes, const unsigned int* gbits, dir_stream_t stream,
                   const unsigned int* xbits = nullptr) {
    DIR_CHECK_ARG(dW && M >= 0 && N > 0 && K > 0 && dw_ld >= K, "%s: bad argument (M=%lld N=%d K=%d dw_ld=%lld)", name, (long long)M, N, K,
                  (long long)dw_ld);
    hipStream_t st = as_stream(stream);
    if (M == 0) {                                    // an empty batch has a zero gradient (empty operands have no storage: null allowed)
        if (zero_2d_async(dW, dw_ld * sizeof(float), K * sizeof(float), N, st) != hipSuccess) return fail(DIR_E_HIP, "%s: memset failed", name);
        if (db && zero_async(db, N * sizeof(float), st) != hipSuccess) return fail(DIR_E_HIP, "%s: memset failed", name);
        return DIR_OK;
    }
    DIR_CHECK_ARG(g && x && workspace && g_ld >= N && x_ld >= K, "%s: null pointer or row stride smaller than the width", name);
    if (N % 4 || K % 4 || g_ld % 4 || x_ld % 4 || !aligned16(g) || !aligned16(x))
        return fail(DIR_E_UNSUPPORTED, "%s: N=%d K=%d g_ld=%lld x_ld=%lld must be multiples of 4 and g / x 16-byte aligned (rows are staged "
                    "with 16-byte loads)", name, N, K, (long long)g_ld, (long long)x_ld);
    DIR_CHECK_ARG(aligned16(workspace) && workspace_bytes >= dir_dense_dw_bf16x3_workspace_bytes(M, N, K),
                  "%s: workspace must be 16-byte aligned and hold dir_dense_dw_bf16x3_workspace_bytes(M, N, K) bytes", name);
    const DdwPlan p = ddw_plan(M, N, K);
    const unsigned grid = (unsigned)(p.nnb * p.nkb * p.nspan);
    float* part = static_cast<float*>(workspace);
    float* bpart = db ? part + (int64_t)p.nspan * N * K : nullptr;
    DIR_CHECK_ARG(np == 3 || gbits, "%s: g_absmax_bits is null", name);
#define DDW_LAUNCH(KB_, NP_)                                                                                                       \
    do {                                                                                                                           \
        static LdsOnce once;                                                                                                      \
        const size_t lds = NP_ * (size_t)(DDW_NT + KB_) * 1024;                                                                    \
        (void)lds_limit(once, (int)lds, &dense_dw_bf3_k<KB_, NP_>);                                                               \
        hipLaunchKernelGGL((dense_dw_bf3_k<KB_, NP_>), dim3(grid), dim3(512), lds, st, g, g_ld, x, x_ld, M, N, K, p.nkb, p.nspan,  \
                           p.steps_per_span, p.steps, part, bpart, gbits, np == 2 ? xbits : nullptr);                              \
    } while (0)
    if (np == 2) {
        if (p.KB == 13) DDW_LAUNCH(13, 2);
        else if (p.KB == 8) DDW_LAUNCH(8, 2);
        else DDW_LAUNCH(16, 2);
    } else if (p.KB == 13) DDW_LAUNCH(13, 3);
    else if (p.KB == 8) DDW_LAUNCH(8, 3);
    else DDW_LAUNCH(16, 3);
#undef DDW_LAUNCH
    DIR_CHECK_LAUNCH(name);
    ddw_reduce(part, bpart, N, K, p.nspan, dW, dw_ld, db, st, np == 2 ? gbits : nullptr, np == 2 ? xbits : nullptr);
    DIR_CHECK_LAUNCH("dense_dw reduce");
    return DIR_OK;
}

extern "C" int dir_dense_dw_bf16x3_f32(const float* g, int64_t g_ld, const float* x, int64_t x_ld, int64_t M, int N, int K, float* dW,
                                       int64_t dw_ld, float* db, void* workspace, int64_t workspace_bytes, dir_stream_t stream) {
    return ddw_run("dir_dense_dw_bf16x3_f32", 3, g, g_ld, x, x_ld, M, N, K, dW, dw_ld, db, workspace, workspace_bytes, nullptr, stream);
}

// fp16 x 2: g scaled by one power of two from g_absmax_bits (device pointer: the bit pattern of an upper bound of max |g|, e.g. the all_bits
// of dir_row_absmax_bits_f32), x split as in the fp16 x 2 forward (|x| < 65 504, O(1) activations).  Workspace as the bf16 x 3 entry.
extern "C" int dir_dense_dw_f16x2_f32(const float* g, int64_t g_ld, const float* x, int64_t x_ld, int64_t M, int N, int K, float* dW,
                                      int64_t dw_ld, float* db, void* workspace, int64_t workspace_bytes, const unsigned int* g_absmax_bits,
                                      dir_stream_t stream) {
    return ddw_run("dir_dense_dw_f16x2_f32", 2, g, g_ld, x, x_ld, M, N, K, dW, dw_ld, db, workspace, workspace_bytes, g_absmax_bits, stream);
}

// ... with x scaled by one power of two as well (x_absmax_bits: the bit pattern of an upper bound of max |x|, e.g. the all_bits the
// row-scaled forward kernel or dir_row_absmax_bits_f32 left for this layer's input): no bound on either operand's magnitude is assumed
extern "C" int dir_dense_dw_f16x2_scaled_f32(const float* g, int64_t g_ld, const float* x, int64_t x_ld, int64_t M, int N, int K, float* dW,
                                             int64_t dw_ld, float* db, void* workspace, int64_t workspace_bytes, const unsigned int* g_absmax_bits,
                                             const unsigned int* x_absmax_bits, dir_stream_t stream) {
    DIR_CHECK_ARG(x_absmax_bits || M == 0, "dir_dense_dw_f16x2_scaled_f32: x_absmax_bits is null");
    return ddw_run("dir_dense_dw_f16x2_scaled_f32", 2, g, g_ld, x, x_ld, M, N, K, dW, dw_ld, db, workspace, workspace_bytes, g_absmax_bits, stream,
                   x_absmax_bits);
}
