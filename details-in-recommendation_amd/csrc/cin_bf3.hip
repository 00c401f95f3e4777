// cin_bf3.hip -- the CIN layer of cin.hip on the bf16 matrix pipe with fp32-equivalent arithmetic ("bf16 x 3").
//
// NO REFERENCE CODE (README.md:28 links arXiv:1803.05170); definition as in cin.hip / include/dir_hip.h:
//   xout[b,h,d] = sum_{i<Hp} sum_{j<m} W[h, i*m+j] * xk[b,i,d] * x0[b,j,d]
//               = sum_j x0[r,j] * T_j[r,h],      T_j[r,h] = sum_i xk[r,i] * W[h,i,j]        (r = (b,d))
//
// Arithmetic.  T_j is a plain GEMM.  Both of its fp32 operands are split into three bf16 pieces by round-to-nearest,
// v = v0 + v1 + v2 exactly (3 x 8 significant bits plus the pieces' signs cover fp32's 24; the exponent range is fp32's).  Of the nine
// piece products the six of weight >= 2^-16 are accumulated in fp32 by the matrix instruction (a bf16 x bf16 product is exact in
// fp32); the three dropped ones are <= 2^-24 relative each and of random sign.  The field factor is applied AFTER the product:
// out += x0[r,j] * T_j, one fp32 fma per accumulator register and field.  Measured against the double-accumulating oracle the
// result is closer than the fp32-MFMA kernel's (2-3e-7 vs 1-2.5e-6 scaled error; tools/cin_bf3_probe.py).
//
// Why this formulation.  The left operand xk of T_j does not depend on j, so its split is done ONCE per 64 values of i and kept in
// registers for all m fields: the MFMA loop has no split arithmetic at all.  (The first version of this kernel split the rounded
// product xk*x0 for every k-step: 2.3 VALU instructions per MFMA, 3.94 ms per 128-wide layer; this one 3.13 ms; fp32 MFMA 6.3 ms.)
// Matrix instruction: v_mfma_f32_16x16x32_bf16, which on random operands holds a higher clock than the 32x32x16 form
// (tools/bf16_shape_probe.hip: 2.20 vs 1.79 PFLOP/s chip-wide in bare six-deep chains -- the chip is power-bound there, so the
// bare-MFMA time of a 128-wide layer is 2.43 ms with this shape).
//
// Work split.  A workgroup of 8 waves (two per SIMD, <= 256 VGPRs, MFMA results kept in VGPRs) owns 256 rows x 128 columns; wave w
// rows [32w, 32w+32) = 2 row tiles x 8 column tiles of 16 x 16: 64 registers for T_j and 64 for out.  i runs in halves of KS*32
// values (KS = 2; 1 when Hp <= 32); a chunk is one (half, field j): KS k-steps x 16 tiles x 6 = 192 MFMAs per wave, one barrier
// per chunk.  Since round 5 a chunk is two blocks: its matrix instructions, column tile outermost so that a tile's KS * NMF instructions
// form one dependent chain (B operands read one column tile ahead), then its accumulate fmas -- see the comment in the kernel and
// profiles/NOTES.md R5.9 / R5.10 for why (rounds 2-4 placed the fmas of the PREVIOUS chunk between the matrix instructions of the first
// k-step).  H's columns are cut into 128-wide blocks plus ONE narrower last block of
// 2 / 4 / 6 column tiles (template parameter CT; its own launch with a column offset).
// Operand registers and two waves per SIMD: the A operands are rewritten once per half behind a barrier, the B operands arrive by
// LDS reads into alternating registers -- no VALU instruction writes an MFMA source register close behind the MFMAs that read it
// (the pattern that made a bf16x3 DIN unit irreproducible at two waves per SIMD); run-to-run bitwise equality is tested.
//
// The data-gradient form (template parameters DOT, FJ = 2, CT <= 4; dir_cin_layer_dot_bf16x3_f32) is described at the kernel.
//
// LDS: Wb [2][KS][3 planes][8 ct][64 lanes][8 bf16]  the chunk's W operand image, written by global_load_lds in the order
//                                                    cin_bf3_pack_w_k lays the global image out (one ds_read_b128 per operand);
//      x0s [m][256] f32                              the workgroup's x0 slice.
// What is left above the bare-MFMA time (timing ablations): the W image's LDS-DMA writes 0.26 ms (they share the LDS with the operand
// reads), the accumulate fmas 0.14 ms, barriers 0.04 ms.
#include "common.hpp"
#include <type_traits>

#ifdef CIN_ABL
#ifndef CIN_ABL_FWD
#define CIN_ABL_FWD 0      // timing ablations apply to the dot form (0) or to the forward / pair forms (1)
#endif
#define CIN_ABL_ON (CIN_ABL_FWD ? !DOT : DOT)
#endif


namespace dir {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));
typedef unsigned int u32x4_t __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) void* lds_ptr_t;
typedef const __attribute__((address_space(1))) void* glb_ptr_t;

__device__ __forceinline__ unsigned int bt_pk(float a, float b) {      // v_cvt_pk_bf16_f32 (round to nearest even), a in the low half
    typedef __bf16 pk2_t __attribute__((ext_vector_type(2)));
    const pk2_t v = {(__bf16)a, (__bf16)b};
    unsigned int w = __builtin_bit_cast(unsigned int, v);
    // An EMPTY asm: it only hides where w came from (otherwise the compiler converts `a` a second time, alone, to form float(bf16(a))
    // instead of shifting w).  The convert itself stays a compiler-generated instruction: its results become MFMA operands, and the
    // compiler's hazard recognizer does not see inside inline asm.
    asm("" : "+v"(w));
    return w;
}
__device__ __forceinline__ void bt_split_pair(float a, float b, unsigned int& w0, unsigned int& w1, unsigned int& w2) {
    w0 = bt_pk(a, b);
    const float ra = a - __builtin_bit_cast(float, w0 << 16), rb = b - __builtin_bit_cast(float, w0 & 0xffff0000u);
    w1 = bt_pk(ra, rb);
    const float sa = ra - __builtin_bit_cast(float, w1 << 16), sb = rb - __builtin_bit_cast(float, w1 & 0xffff0000u);
    w2 = bt_pk(sa, sb);
}

// ---- round 4: a second arithmetic, "fp16 x 2" (NP = 2) -----------------------------------------------------------------------------------
// An fp32 operand as the sum of TWO fp16 pieces by round-to-nearest, v ~ v0 + v1 (2 x 11 significant bits: relative error <= 2^-22 while
// the second piece is a normal fp16 number, i.e. |v| >= 2^-3; below that the residual is a subnormal and the ABSOLUTE error is <= 2^-25
// per element; |v| must stay below fp16's 65 504), and of the four piece products the THREE of weight >= 2^-11 on
// v_mfma_f32_16x16x32_f16 (same rate as the bf16 form; fp16 x fp16 products are exact in fp32): half the matrix instructions of bf16 x 3
// and two thirds of its split arithmetic, for a scaled error of 3-6e-7 against the double-accumulating oracle on embedding-scale operands
// (tools/f16x2_probe.py: about what fp32's own accumulation order costs; bf16 x 3: 2-3e-9) -- inside the 1e-5 bar with a factor 20 to
// spare.  Used by the FORWARD layers (operands are embeddings / activations / weights: O(1) magnitudes); the data-gradient form keeps
// bf16 x 3 (its left operand is a gradient: small magnitudes would sit in fp16's subnormal range).  ops.CIN_ARITH / DIR_CIN_ARITH select.
typedef _Float16 f16x8_t __attribute__((ext_vector_type(8)));
__device__ __forceinline__ unsigned int bt_pk_h(float a, float b) {     // v_cvt_pk_f16_f32 (round to nearest even), a in the low half
    typedef _Float16 h2_t __attribute__((ext_vector_type(2)));
    const h2_t v = {(_Float16)a, (_Float16)b};
    unsigned int w = __builtin_bit_cast(unsigned int, v);
    asm("" : "+v"(w));      // (as bt_pk: float(f16(a)) is then formed from w, not by a second rounding of a)
    return w;
}
__device__ __forceinline__ void bt_split_pair_h(float a, float b, unsigned int& w0, unsigned int& w1) {
    typedef _Float16 h2_t __attribute__((ext_vector_type(2)));
    w0 = bt_pk_h(a, b);
    const h2_t h = __builtin_bit_cast(h2_t, w0);
    w1 = bt_pk_h(a - (float)h[0], b - (float)h[1]);
}
template <int NP> struct BtPc;
template <> struct BtPc<3> {
    using op_t = bf16x8_t;
    static constexpr int NMF = 6;                                   // matrix instructions per (k-step, tile)
    __device__ static __forceinline__ void split(float a, float b, unsigned int (&w)[3]) { bt_split_pair(a, b, w[0], w[1], w[2]); }
    __device__ static __forceinline__ f32x4 mma(const op_t (&a)[3], const op_t (&b)[3], f32x4 t) {      // the six products, smallest first
        t = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[0], b[2], t, 0, 0, 0);
        t = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[2], b[0], t, 0, 0, 0);
        t = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[1], b[1], t, 0, 0, 0);
        t = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[0], b[1], t, 0, 0, 0);
        t = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[1], b[0], t, 0, 0, 0);
        t = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[0], b[0], t, 0, 0, 0);
        return t;
    }
};
template <> struct BtPc<2> {
    using op_t = f16x8_t;
    static constexpr int NMF = 3;
    __device__ static __forceinline__ void split(float a, float b, unsigned int (&w)[2]) { bt_split_pair_h(a, b, w[0], w[1]); }
    __device__ static __forceinline__ f32x4 mma(const op_t (&a)[2], const op_t (&b)[2], f32x4 t) {      // the three products, smallest first
        t = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[1], b[0], t, 0, 0, 0);
        t = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[0], b[1], t, 0, 0, 0);
        t = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[0], b[0], t, 0, 0, 0);
        return t;
    }
};

// W [H, Hp*m] fp32 -> image [column block of CT tiles][half kh][field j][ks][plane][ct][lane][8 e] bf16, element e of lane l of column
// tile ct in k-step ks = piece of W[h = hoff + 16*(CT*cb + ct) + (l & 15)][i = KS*32*kh + 32*ks + 8*(l >> 4) + e][j]; zero where h >= H or
// i >= Hp.  One launch per block width (the 128-wide blocks, then the narrower last block).
// The tensor scale of a weight that meets a row-scaled left operand (RS): W's largest |element| as WPARTS partial maxima (plain stores: no
// atomics, nothing to zero between calls); the pack kernel multiplies W by the power of two 2^k that puts it into [2^14, 2^15), the
// layer kernel takes 2^-k out again together with the rows' scales.  With both operands scaled the fp16 x 2 layer needs no promise about
// magnitudes at all (any W, any xk that fp32 holds), and W's small elements keep 22 bits relative to its largest instead of an absolute 2^-25.
constexpr int WPARTS = 32;
__global__ __launch_bounds__(256) void cin_w_absmax_k(const float* __restrict__ W, int64_t n, float* __restrict__ part) {
    __shared__ float red[4];
    float mx = 0.f;
    const int64_t n4 = (reinterpret_cast<uintptr_t>(W) & 15) == 0 ? n >> 2 : 0;      // (16-byte loads: 10 -> ~4 us for a 128 x 3 328 weight)
    const float4* W4 = reinterpret_cast<const float4*>(W);
    for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < n4; e += (int64_t)gridDim.x * 256) {
        const float4 v = W4[e];
        mx = fmaxf(fmaxf(mx, fmaxf(fabsf(v.x), fabsf(v.y))), fmaxf(fabsf(v.z), fabsf(v.w)));
    }
    for (int64_t e = 4 * n4 + (int64_t)blockIdx.x * 256 + threadIdx.x; e < n; e += (int64_t)gridDim.x * 256) mx = fmaxf(mx, fabsf(W[e]));
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o, 64));
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = mx;
    __syncthreads();
    if (threadIdx.x == 0) part[blockIdx.x] = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
}
__device__ __forceinline__ int w_scale_exp(const float* __restrict__ part) {          // k of the scale 2^k (every lane reads the WPARTS values)
    float mx = 0.f;
#pragma unroll
    for (int q = 0; q < WPARTS; ++q) mx = fmaxf(mx, part[q]);
    int k = 141 - (int)((__builtin_bit_cast(unsigned int, mx) >> 23) & 0xffu);
    return k > 60 ? 60 : (k < -60 ? -60 : k);
}
__device__ __forceinline__ float pow2f(int k) { return __builtin_bit_cast(float, (unsigned int)(127 + k) << 23); }     // |k| <= 126

// The VERDICT of a forward layer whose producer left xk's row maxima (round 5): when max |xk| sits in [2^-4, 2^15) -- where splitting the rows
// unscaled loses nothing that matters (an element below 2^-3 keeps an absolute 2^-25, at most 2^-21 of the tensor's largest) -- AND the smallest
// non-zero row maximum is at least 2^-8 (round 6, ADVICE r5: a tensor inside the window may still hold tiny rows; an element's absolute 2^-25
// is 2^-17 of such a row's largest at worst, inside the 1e-5 bar PER ROW, not just per tensor) the plain kernel runs (8-9 % faster than the
// row-scaled form: no scan of the rows, profiles/r05_cin_rs_probe.txt), otherwise the row-scaled one.  W is
// scaled by its tensor power of two in BOTH (typical CIN weights sit around 2^-5: unscaled they cost the plain form a factor ten in accuracy).
// Decided ON THE DEVICE from the partial maxima (wpart [WPARTS] then xpart [XPARTS] = XBLOCKS maxima + XBLOCKS minima of the non-zero row
// maxima, contiguous): both kernels are launched, each leaves at once unless the verdict names it.  No host read, graph-capturable.
constexpr int XPARTS = 256;
constexpr int XBLOCKS = XPARTS / 2;
__global__ __launch_bounds__(256) void cin_bits_absmax_k(const unsigned int* __restrict__ bits, int64_t n, float* __restrict__ xpart) {
    __shared__ unsigned int red[8];
    unsigned int mx = 0u, mn = 0x7f800000u;                   // bit patterns of non-negative floats order like the floats; mn over NON-ZERO rows
    // (16-byte loads: with one word per load a thread walked 32 dependent-latency iterations over the 4 MB of a config-5 layer: 13.6 us)
    const int64_t n4 = (reinterpret_cast<uintptr_t>(bits) & 15) == 0 ? n >> 2 : 0;
    const uint4* b4 = reinterpret_cast<const uint4*>(bits);
    for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < n4; e += (int64_t)gridDim.x * 256) {
        const uint4 v = b4[e];
        const unsigned int w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            mx = max(mx, w[q]);
            mn = w[q] ? min(mn, w[q]) : mn;
        }
    }
    for (int64_t e = 4 * n4 + (int64_t)blockIdx.x * 256 + threadIdx.x; e < n; e += (int64_t)gridDim.x * 256) {
        const unsigned int b = bits[e];
        mx = max(mx, b);
        mn = b ? min(mn, b) : mn;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        mx = max(mx, (unsigned int)__shfl_xor((int)mx, o, 64));
        mn = min(mn, (unsigned int)__shfl_xor((int)mn, o, 64));
    }
    if ((threadIdx.x & 63) == 0) { red[threadIdx.x >> 6] = mx; red[4 + (threadIdx.x >> 6)] = mn; }
    __syncthreads();
    if (threadIdx.x == 0) {
        xpart[blockIdx.x] = __builtin_bit_cast(float, max(max(red[0], red[1]), max(red[2], red[3])));
        xpart[XBLOCKS + blockIdx.x] = __builtin_bit_cast(float, min(min(red[4], red[5]), min(red[6], red[7])));
    }
}
__device__ __forceinline__ bool cin_plain_verdict(const float* __restrict__ vparts) {      // wave-uniform; every lane of the wave calls it
    const int lane = threadIdx.x & 63;
    float wm = lane < WPARTS ? vparts[lane] : 0.f, xm = 0.f, xn = __builtin_inff();
#pragma unroll
    for (int q = 0; q < XBLOCKS / 64; ++q) {
        xm = fmaxf(xm, vparts[WPARTS + 64 * q + lane]);
        xn = fminf(xn, vparts[WPARTS + XBLOCKS + 64 * q + lane]);
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        wm = fmaxf(wm, __shfl_xor(wm, o, 64));
        xm = fmaxf(xm, __shfl_xor(xm, o, 64));
        xn = fminf(xn, __shfl_xor(xn, o, 64));
    }
    (void)wm;            // W is scaled in BOTH forms (the pack kernel's tensor scale costs nothing; the plain kernel takes 2^-kw out of its
                         // accumulators in the epilogue): only the row operand's magnitudes decide
    return xm >= 0x1p-4f && xm < 0x1p15f && xn >= 0x1p-8f;
}

template <int NP>
__global__ __launch_bounds__(256) void cin_bf3_pack_w_k(const float* __restrict__ W, int m, int Hp, int H, int KS, int nkh, int ncb, int CT,
                                                        int hoff, unsigned int* __restrict__ img, const float* __restrict__ wpart = nullptr,
                                                        int verdict = 0 /* 1: wpart is followed by xpart, and a "plain" verdict leaves W unscaled */) {
    const int64_t total = (int64_t)ncb * nkh * m * KS * CT * 64 * 4;   // one thread per pair of e
    const int stepdw = NP * CT * 64 * 4;                               // dwords per k-step
    (void)verdict;                                                     // (W is scaled under either verdict)
    const float wsc = wpart ? pow2f(w_scale_exp(wpart)) : 1.f;
    for (int64_t e_ = (int64_t)blockIdx.x * 256 + threadIdx.x; e_ < total; e_ += (int64_t)gridDim.x * 256) {
        int64_t q = e_;
        const int ep = (int)(q & 3); q >>= 2;
        const int l = (int)(q & 63); q >>= 6;
        const int ct = (int)(q % CT); q /= CT;
        const int ks = (int)(q % KS); q /= KS;
        const int j = (int)(q % m); q /= m;
        const int kh = (int)(q % nkh);
        const int cb = (int)(q / nkh);
        const int h = hoff + 16 * (CT * cb + ct) + (l & 15);
        const int i = KS * 32 * kh + 32 * ks + 8 * (l >> 4) + 2 * ep;
        const float v0 = (h < H && i < Hp) ? W[(int64_t)h * Hp * m + (int64_t)i * m + j] * wsc : 0.f;
        const float v1 = (h < H && i + 1 < Hp) ? W[(int64_t)h * Hp * m + (int64_t)(i + 1) * m + j] * wsc : 0.f;
        unsigned int pw[NP];
        BtPc<NP>::split(v0, v1, pw);
        const int64_t chunk = ((int64_t)cb * nkh + kh) * m + j;
        const int64_t base = (chunk * KS + ks) * stepdw + (ct * 64 + l) * 4 + ep;     // plane stride: CT*64*4 dwords
#pragma unroll
        for (int q = 0; q < NP; ++q) img[base + q * CT * 64 * 4] = pw[q];
    }
}

// DOT (the backward's data gradients, dir_cin_dx_bf16x3_f32): besides out = sum_j x0_j * T_j the kernel also forms, for every field,
//   dot[r, j] = sum_h y[r, h] * T_j[r, h]      (y in the layout of xout; the sum runs over this workgroup's columns and this half of i)
// from the same T_j tiles, just before the next chunk overwrites them.  Called with xk := G, W := W1 (W1[i, h*m+j] = W[h, i*m+j]) and
// y := the layer's xk, `out` is dxk and the dot partials add up to dx0.  It keeps y's tile in registers next to `out` and T, so its
// column blocks are 64 wide (CT <= 4) and one staged chunk holds FJ = 2 fields: per barrier a wave still issues the forward's 192
// MFMAs on 256 rows (one row tile per wave on 128-column blocks -- the first version of this form -- halves the MFMAs per B-operand
// read and per barrier: 4.7 ms against the forward's 3.3 ms at 128 x 128).
// PAIRS (the FIRST layer of a stack, xk = x0; dir_cin_layer1_bf16x3_f32): xout[r,h] = sum_{i,j} W[h,i,j] x0_i x0_j is a quadratic form, so
// only the m (m + 1) / 2 unordered pairs need multiplying: the reduction runs over PAIRS p = (i <= j) with the weight
// W[h,i,j] + W[h,j,i] (W[h,i,i] on the diagonal) -- the kernel is called with ONE "field" whose factor is 1, "Hp" = the pair count, and the
// A operand of pair p is the product x0[r, i(p)] * x0[r, j(p)], formed from the LDS-resident x0 slice when the half's operands are split
// (ptab[p] = i | j << 8; mx = the real field count).  351 reduction slots instead of 26 x 32 = 832 at m = 26.
template <int KS, int CT /* column tiles of 16 per workgroup: 8 (128 columns), or 6 / 4 / 2 for the last block of a layer */,
          int RT = 2 /* row tiles of 16 per wave */, bool DOT = false, int FJ = 1 /* fields per staged chunk */, bool PAIRS = false,
          int NP = 3 /* pieces per operand: 3 = bf16 x 3, 2 = fp16 x 2 */,
          bool RS = false /* fp16 x 2 with a left operand of unknown magnitude (a gradient): every row of xk is scaled by a power of two so
                             that its largest element lands in [2^14, 2^15), the scale is taken out again where T meets the field factor */>
__global__ __launch_bounds__(512, 1) void cin_bf3_k(const float* __restrict__ x0, const float* __restrict__ xk,
                                                     const unsigned char* __restrict__ img, int m, int Hp, int H, int D, int dshift,
                                                     int nkh, int hoff /* first output column of this launch */, int64_t R,
                                                     float* __restrict__ xout, float* __restrict__ pooled, int64_t pooled_ld,
                                                     const float* __restrict__ y /* DOT: [B, H, D].  Forward kernels: null, or x0's INVERSE POSITIONS
                                                        (int64 [B, mx], round 6): x0 is then a ROW LIST [n, D] and entry (b, j) the position of (sample,
                                                        field)'s row in it, < 0 = a zero row -- the sharded lookup's received rows, no finish pass.  (The
                                                        slot is shared because one more kernel argument made the 256-register dot form spill.) */,
                                                     float* __restrict__ dotp /* DOT: partials [nkh][B, m, D] of this launch's column block */,
                                                     const float* __restrict__ addp /* optional [B, H] (row stride addp_ld): added to xout[b, h, :] */,
                                                     int64_t addp_ld, const unsigned short* __restrict__ ptab /* PAIRS: i | j << 8 per pair */,
                                                     int mx /* fields of the x0 slice (= m unless PAIRS) */,
                                                     unsigned int* __restrict__ amax_out = nullptr /* RS: atomicMax of the bit pattern of max |xk| */,
                                                     const float* __restrict__ wpart = nullptr /* RS: cin_w_absmax_k's partial maxima of the W the image was scaled by */,
                                                     const unsigned int* __restrict__ xk_bits = nullptr /* RS, optional [R]: bit pattern of max_i |xk[r, i]|, left by
                                                                                                          the kernel that produced xk: no scan of the rows in the prologue */,
                                                     unsigned int* __restrict__ xout_bits = nullptr /* optional [R]: the same of THIS launch's output rows (the launch
                                                                                                       must cover all H columns: one column block) */,
                                                     int vwant = -1 /* 1 / 0: wpart is followed by xpart; run only under the plain / the row-scaled verdict */,
                                                     float* __restrict__ sink = nullptr /* DOT: one writable word nobody reads (behind the workspace's partial maxima) */) {
    if (vwant >= 0) {                                              // (before anything else: the kernel the verdict does not name costs an empty launch)
        if (cin_plain_verdict(wpart) != (vwant == 1)) return;
    }
    using Pc = BtPc<NP>;
    using op_t = typename Pc::op_t;
    static_assert(NP == 3 || !DOT || RS, "the data-gradient form on fp16 x 2 needs the row-scaled left operand");
    static_assert(!RS || NP == 2, "row scaling belongs to the fp16 x 2 split");
    constexpr int STEPB = NP * CT * 1024;                        // bytes of W image per k-step of 32
    constexpr int CHB = KS * STEPB;                              // bytes of W image per (half, field); a staged chunk holds FJ of them
    constexpr int BT_ROWS = 8 * 16 * RT;                         // rows per workgroup (shadows the 256 of the forward)
    constexpr int WR = 16 * RT;                                  // rows per wave
    extern __shared__ __attribute__((aligned(16))) unsigned char bt_smem[];
    unsigned char* Wb = bt_smem;                                 // [2][FJ * CHB]
    float* x0s = reinterpret_cast<float*>(bt_smem + 2 * FJ * CHB);   // [m][BT_ROWS]

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int n = lane & 15;
    const int lg = lane >> 4;
    // (row block, column block) of this workgroup.  With TWO column blocks (the dot form at H = 128) the pair that shares a row block --
    // and so reads the same 128 KB of xk rows -- is made two CONSECUTIVE workgroups of one XCD (workgroup L runs on XCD L % 8): the second
    // one finds the rows in that XCD's L2 instead of reading them from HBM a second time (x-fastest dispatch put them 4096 workgroups apart).
    unsigned int rbi = blockIdx.x, cbi = blockIdx.y;
    if (gridDim.y == 2 && (gridDim.x & 7) == 0) {
        const unsigned int L = blockIdx.y * gridDim.x + blockIdx.x, xcd = L & 7, k = L >> 3;
        cbi = k & 1;
        rbi = (k >> 1) * 8 + xcd;
    }
    const int64_t row0 = (int64_t)rbi * BT_ROWS;
    const int hbase = hoff + cbi * (16 * CT);
    const int nchunk = nkh * m;
    const unsigned char* gimg = img + (int64_t)cbi * nchunk * CHB;

    auto stage_w = [&](int c, int nf, int buf) {     // fields c .. c + nf - 1: nf * KS * 3 * CT pieces of 1 KB over 8 waves, lane-linear
        for (int piece = wave; piece < nf * KS * NP * CT; piece += 8) {
            const unsigned char* src = gimg + (int64_t)c * CHB + piece * 1024 + lane * 16;
            unsigned char* dst = Wb + buf * (FJ * CHB) + piece * 1024;
            __builtin_amdgcn_global_load_lds((glb_ptr_t)src, (lds_ptr_t)dst, 16, 0, 0);
        }
    };

    // ---- prologue: W chunk 0 and the x0 slice (thread t: row t & 255, fields of parity t >> 8)
    stage_w(0, min(FJ, m), 0);
    {
        constexpr int TPR = 512 / BT_ROWS;                           // threads sharing a row: they take the fields j = t / BT_ROWS, + TPR, ...
        const int r = tid % BT_ROWS;
        const int64_t srow = (row0 + r < R) ? row0 + r : R - 1;     // a row >= R only feeds output rows that are never stored
        const int64_t* x0_inv = DOT ? nullptr : reinterpret_cast<const int64_t*>(y);
        if (x0_inv) {
            // the positions first, then every row read, then the LDS stores: written as one loop the compiler waits for each position and
            // each row in turn -- 2 x 13 dependent round trips per workgroup with nothing else resident on the CU (+ 0.2 ms per stack)
            constexpr int MAXJ = (40 + TPR - 1) / TPR;
            const int64_t* iv = x0_inv + (srow >> dshift) * mx;
            const float* x0d = x0 + (srow & (D - 1));
            const int j0 = tid / BT_ROWS;
            int64_t pos[MAXJ];
            float v[MAXJ];
#pragma unroll
            for (int q = 0; q < MAXJ; ++q) pos[q] = j0 + q * TPR < mx ? iv[j0 + q * TPR] : -1;
#pragma unroll
            for (int q = 0; q < MAXJ; ++q) v[q] = x0d[(pos[q] >= 0 ? pos[q] : 0) * D];          // (a clamped address: the load is unconditional)
#pragma unroll
            for (int q = 0; q < MAXJ; ++q)
                if (j0 + q * TPR < mx) x0s[(j0 + q * TPR) * BT_ROWS + r] = pos[q] >= 0 ? v[q] : 0.f;
        } else {
            const float* x0src = x0 + ((srow >> dshift) * mx) * D + (srow & (D - 1));
            for (int j = tid / BT_ROWS; j < mx; j += TPR) x0s[j * BT_ROWS + r] = x0src[(int64_t)j * D];
        }
    }

    // this lane's A rows: row tile rt -> row WR*wave + 16*rt + n of the workgroup; k slot 8*lg + e of each k-step
    const float* xsrc[RT];
#pragma unroll
    for (int rt = 0; rt < RT; ++rt) {
        const int64_t gr = row0 + wave * WR + rt * 16 + n;
        const int64_t grc = gr < R ? gr : R - 1;
        xsrc[rt] = xk + ((grc >> dshift) * Hp) * D + (grc & (D - 1));
    }

    // RS: the power-of-two scale of this lane's A rows and, for the rows of its accumulator registers, the inverse.  A row's largest
    // |element| over ALL its Hp channels decides (exponent e: scale 2^(141 - e), clamped to 2^+-100 -- an all-zero row stays zero): the
    // scaled row fits fp16 with its largest elements at 11 + 11 bits; elements more than 2^-17 below the row's largest keep an ABSOLUTE
    // error of 2^-25 (2^-39 of the largest), which is what a sum over the row's channels needs.  The scaling itself is exact.
    float rscale[RS ? RT : 1];
    float rowinv[RS ? RT : 1];          // the inverse scale of the lane's OWN row (row n of tile rt): what its x0 slice entries are multiplied by
    f32x4 rinv[RS ? RT : 1];            // ... of the rows of the lane's accumulator registers (DOT: y; PAIRS: the pseudo-field's factor)
    float wave_max = 0.f;
    float* bt_wmax = x0s + (size_t)mx * BT_ROWS;                      // [8]: behind the x0 slice (all LDS is dynamic: the 160 KiB limit is set for the kernel)
    const int kw = (RS && wpart) ? w_scale_exp(wpart) : 0;     // the W image's tensor scale 2^kw (|kw| <= 60), taken out with the rows' scales
    if constexpr (RS && !PAIRS) {
#pragma unroll
        for (int rt = 0; rt < RT; ++rt) {
            float mx = 0.f;
            if (xk_bits) {                                   // (uniform) the producer of xk left the rows' maxima
                const int64_t gr = row0 + wave * WR + rt * 16 + n;
                mx = __builtin_bit_cast(float, xk_bits[gr < R ? gr : R - 1]);
            } else {
            for (int i0 = 8 * lg; i0 < Hp; i0 += 32) {
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const int i = i0 + e;
                    const float x = xsrc[rt][(int64_t)(i < Hp ? i : Hp - 1) * D];
                    mx = fmaxf(mx, fabsf(x));
                }
            }
            mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
            mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
            }
            wave_max = fmaxf(wave_max, mx);             // the tensor's maximum rides along (the weight-gradient kernels scale G by it)
            int k = 141 - (int)((__builtin_bit_cast(unsigned int, mx) >> 23) & 0xffu);
            k = k > 100 ? 100 : (k < -100 ? -100 : k);
            k = k + kw > 126 ? 126 - kw : (k + kw < -126 ? -126 - kw : k);      // the combined inverse 2^-(k + kw) stays a normal number
            rscale[rt] = __builtin_bit_cast(float, (unsigned int)(127 + k) << 23);
            const float inv = pow2f(-(k + kw));
            rowinv[rt] = inv;
#pragma unroll
            for (int q = 0; q < 4; ++q) rinv[rt][q] = __shfl(inv, 4 * lg + q, 64);      // lane 4 lg + q holds row 4 lg + q of the tile
        }
        if (amax_out) {                                 // (uniform) one value per wave into LDS; ONE atomic per workgroup behind the prologue's barrier
#pragma unroll                                          //  (an atomic per wave and row tile: 65 536 on one address, +87 us on the 0.79 ms contraction)
            for (int o = 8; o > 0; o >>= 1) wave_max = fmaxf(wave_max, __shfl_xor(wave_max, o, 64));
            if (lane == 0) bt_wmax[wave] = wave_max;
        }
    }

    // DOT: the accumulators are TRANSPOSED (the W image is the matrix instruction's A operand, the rows its B operand: same registers, swapped):
    // lane (n, lg) holds T[column 16 ct + 4 lg + q, row 16 rt + n].  The field factor x0[r, j] is then ONE value per lane and row tile, and
    // the dot over the columns runs inside the lane (16 fmas + a two-step fold of the lane groups per row tile and chunk; in the forward's
    // layout it was a 16-lane reduction of four values per row tile: ~80 of the chunk's ~160 VALU instructions).
    f32x4 out[RT][CT], T[1][RT][CT];
#pragma unroll
    for (int rt = 0; rt < RT; ++rt)
#pragma unroll
        for (int ct = 0; ct < CT; ++ct) {
            out[rt][ct] = (f32x4){0.f, 0.f, 0.f, 0.f};
            T[0][rt][ct] = (f32x4){0.f, 0.f, 0.f, 0.f};
        }
    op_t a[KS][RT][NP];                                                              // the half's A operands: [k-step][row tile][piece]
    // DOT: y in the accumulators' (transposed) layout: row 16*rt + n, columns 16*ct + 4*lg + q
    f32x4 yv[DOT ? RT : 1][DOT ? CT : 1];
    if constexpr (DOT) {
#pragma unroll
        for (int rt = 0; rt < RT; ++rt) {
            const int64_t gr = row0 + wave * WR + rt * 16 + n;
            const int64_t grc = gr < R ? gr : R - 1;
            const float* yr = y + ((grc >> dshift) * H) * D + (grc & (D - 1));
            // (clamped, unconditional loads, selected afterwards: a load inside the branch of its guard is waited for there, one round trip each)
#pragma unroll
            for (int ct = 0; ct < CT; ++ct)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int h = hbase + 16 * ct + 4 * lg + q;
                    yv[rt][ct][q] = yr[(int64_t)(h < H ? h : H - 1) * D];
                }
#pragma unroll
            for (int ct = 0; ct < CT; ++ct) {
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int h = hbase + 16 * ct + 4 * lg + q;
                    asm volatile("" : "+v"(yv[rt][ct][q]));
                    if (!(h < H && gr < R)) yv[rt][ct][q] = 0.f;
                }
                if constexpr (RS) yv[rt][ct] *= rowinv[rt];          // T carries the rows' scales: taken out where it is consumed
            }
        }
    }

    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if constexpr (RS) {
        if (amax_out && cbi == 0 && tid == 0) {
            float wm = bt_wmax[0];
#pragma unroll
            for (int q = 1; q < 8; ++q) wm = fmaxf(wm, bt_wmax[q]);
            atomicMax(amax_out, __builtin_bit_cast(unsigned int, wm));
        }
    }

    if constexpr (RS && !PAIRS) {
        // The rows' inverse scales go INTO the LDS-resident x0 slice (x0s[j][r] *= 2^-(k_r + kw), exact): the field factor a chunk reads is
        // then already the one that takes T's scales out, and the main loop is the unscaled kernel's instruction for instruction.  (A
        // multiply on the freshly read factor at the top of every chunk cost 8 % of the layer: the wait for that LDS read sat in front of
        // the chunk's first operand reads.)  A row lives in ONE wave (rows WR * wave ..): its lanes (n, lg) share the fields, no barrier.
#pragma unroll
        for (int rt = 0; rt < RT; ++rt) {
            float* xr = x0s + wave * WR + rt * 16 + n;
            for (int j = lg; j < mx; j += 4) xr[j * BT_ROWS] *= rowinv[rt];
        }
    }
    if constexpr (RS && PAIRS) {
        // The first layer on scaled fp16 x 2: the A operand of pair (i, j) is the PRODUCT x0[r,i] x0[r,j], so every row r of the LDS-resident x0
        // slice is scaled by 2^k with its largest |element| in [2^6, 2^7) -- products below 2^14 -- and 2^-2k (times the W image's scale)
        // is the "field factor" of the one pseudo-field.  A row lives in ONE wave (rows WR * wave ..), its 16 x 4 lanes (n, lg) share the
        // fields j = lg, lg + 4, ..: no barrier (a wave's LDS operations complete in order).
#pragma unroll
        for (int rt = 0; rt < RT; ++rt) {
            float* xr = x0s + wave * WR + rt * 16 + n;
            float mxr = 0.f;
            for (int j = lg; j < mx; j += 4) mxr = fmaxf(mxr, fabsf(xr[j * BT_ROWS]));
            mxr = fmaxf(mxr, __shfl_xor(mxr, 16, 64));
            mxr = fmaxf(mxr, __shfl_xor(mxr, 32, 64));
            int k = 133 - (int)((__builtin_bit_cast(unsigned int, mxr) >> 23) & 0xffu);
            k = k > 50 ? 50 : (k < -50 ? -50 : k);
            k = 2 * k + kw > 126 ? (126 - kw) / 2 : (2 * k + kw < -126 ? -((126 + kw) / 2) : k);      // 2^-(2k + kw) stays a normal number
            const float sc = pow2f(k);
            for (int j = lg; j < mx; j += 4) xr[j * BT_ROWS] *= sc;
            const float inv2 = pow2f(-(2 * k + kw));
#pragma unroll
            for (int q = 0; q < 4; ++q) rinv[rt][q] = __shfl(inv2, 4 * lg + q, 64);
        }
    }

    if constexpr (DOT) dotp += (int64_t)cbi * nkh * (R >> dshift) * m * D;     // this column block's partials
    const unsigned char* wlane = Wb + lane * 16;
    const float* x0lane = x0s + wave * WR + (DOT ? n : 4 * lg);      // + j*BT_ROWS + 16*rt: the 4 rows of accumulator registers 0..3 of tile rt
                                                                      //   (DOT: the one row of the lane's transposed accumulators)
    // DOT: the finished dot of one chunk (field jd of half khd): a lane's four partial sums, then the four lane groups; group 0 stores the
    // 16 rows of the tile (16 consecutive floats when D = 16)
    // DOT: the dot partials' address of the row this LANE stores: lane group 0 stores row tile 0's sums, group 1 row tile 1's (below);
    // + ((half * rows) * m + field) * D, a uniform offset.  nullptr: the lane stores to the sink word.
    float* dsel = nullptr;
    if constexpr (DOT) {
        static_assert(RT == 2, "the dot form's reduction folds exactly two row tiles");
        const int64_t gr = row0 + wave * WR + (lg & 1) * 16 + n;
        if (lg < 2 && gr < R) dsel = dotp + ((gr >> dshift) * m) * D + (gr & (D - 1));
    }
    auto store_dot = [&](const f32x4 (&sd)[RT], int khd, int jd, bool live = true) {
        const int64_t uoff = ((int64_t)khd * (R >> dshift) * m + jd) * D;
        // The sums over the four lane groups (rows of 16 lanes) of BOTH row tiles in five VALU instructions, no LDS round trip:
        // v_permlane16_swap exchanges a's odd rows with b's even rows, so a + b holds tile 0's pair sums in rows 0 / 2 and tile 1's in rows
        // 1 / 3; v_permlane32_swap of that with a copy of itself brings the other pair across.  (Two ds_bpermute per tile, each waited for
        // with lgkmcnt(0) behind the chunk's matrix instructions: four exposed LDS round trips per chunk -- 0.3 of the chunk's MFMA time.)
        // The additions are the ones of the butterfly, (g0 + g1) + (g2 + g3): the same bits.
        float a = (sd[0][0] + sd[0][1]) + (sd[0][2] + sd[0][3]);
        float b = (sd[RT - 1][0] + sd[RT - 1][1]) + (sd[RT - 1][2] + sd[RT - 1][3]);
        asm volatile("s_nop 1\n\tv_permlane16_swap_b32 %0, %1\n\ts_nop 1" : "+v"(a), "+v"(b));
        float c = a + b, d = c;
        asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1\n\ts_nop 1" : "+v"(c), "+v"(d));
        const float v = c + d;
        // NO branch: every lane stores; lane groups 2 and 3 (copies), a row past R, or the call in front of the first chunk, go to the sink
        // word.  A predicated store is a basic-block boundary, and the compiler sinks the fmas whose results are only needed later (all of
        // `out`, the second row tile's dot) behind it -- out of the matrix instructions' shadow.
        float* dst = (live && dsel) ? dsel + uoff : sink;
#ifdef CIN_ABL
        if (CIN_ABL & 2) { if (v == 12345.678f) *dst = v; } else
#endif
        *dst = v;
    };
    // one tile of a chunk goes into `out` (and the dot); xf: the chunk's field factors
    auto consume = [&](int rt, int ct, const f32x4& t, f32x4 (&sd)[RT], const f32x4 (&xf)[RT]) {
#ifdef CIN_ABL
        if (CIN_ABL_ON && (CIN_ABL & 8)) { asm volatile("" :: "v"(t)); return; }          // no consume fmas; the tile stays "used"
#endif
        if constexpr (DOT) {
            // explicit pairs (v_pk_fma_f32 on register-adjacent halves): left to itself the compiler pairs the dot's fmas ACROSS tiles, with two
            // v_mov per packed fma, and moves them behind the chunk
            typedef float f32x2 __attribute__((ext_vector_type(2)));
            const f32x2 xb = {xf[rt][0], xf[rt][0]};
            const f32x2 tl = __builtin_shufflevector(t, t, 0, 1), th = __builtin_shufflevector(t, t, 2, 3);
            const f32x4 o = out[rt][ct], y4 = yv[rt][ct], s4 = sd[rt];
            const f32x2 ol = __builtin_elementwise_fma(xb, tl, __builtin_shufflevector(o, o, 0, 1));
            const f32x2 oh = __builtin_elementwise_fma(xb, th, __builtin_shufflevector(o, o, 2, 3));
            const f32x2 sl = __builtin_elementwise_fma(__builtin_shufflevector(y4, y4, 0, 1), tl, __builtin_shufflevector(s4, s4, 0, 1));
            const f32x2 sh = __builtin_elementwise_fma(__builtin_shufflevector(y4, y4, 2, 3), th, __builtin_shufflevector(s4, s4, 2, 3));
            out[rt][ct] = (f32x4){ol[0], ol[1], oh[0], oh[1]};
            sd[rt] = (f32x4){sl[0], sl[1], sh[0], sh[1]};
        } else {
#pragma unroll
            for (int q = 0; q < 4; ++q) out[rt][ct][q] = __builtin_fmaf(xf[rt][q], t[q], out[rt][ct][q]);
        }
    };

    int u = 0;                                  // staged chunk index (its LDS buffer: u & 1)
#ifdef CIN_ABL
    if (CIN_ABL_ON && (CIN_ABL & 1)) nkh = 0;          // timing ablation: no main loop
#endif
    for (int kh = 0; kh < nkh; ++kh) {
        // ---- A operands of this half: xk[r, KS*32*kh + 32*ks + 8*lg + e], split once, used by all m fields
#pragma unroll
        for (int ks = 0; ks < KS; ++ks)
#pragma unroll
            for (int rt = 0; rt < RT; ++rt) {
                float v[8];
                if constexpr (PAIRS) {
                    // the lane's 8 consecutive pairs of this k-step (the table is padded with (0, 0) up to the halves' end: their weights are zero)
                    const u32x4_t pv = *reinterpret_cast<const u32x4_t*>(ptab + (KS * 32 * kh + 32 * ks + 8 * lg));
                    const float* xr = x0s + wave * WR + rt * 16 + n;               // this lane's row of the x0 slice: + field * BT_ROWS
#pragma unroll
                    for (int e = 0; e < 8; ++e) {
                        const unsigned int ent = (pv[e >> 1] >> (16 * (e & 1))) & 0xffffu;
                        v[e] = xr[(ent & 255u) * BT_ROWS] * xr[(ent >> 8) * BT_ROWS];
                    }
                } else {
                // All eight loads are issued before any value is used, and issued UNCONDITIONALLY (clamped index): left alone the compiler sinks
                // each load into the `i < Hp` branch of its only use and waits for it there -- 16 to 32 dependent memory round trips per half
                // and workgroup, with nothing else running on the CU (one workgroup per CU).
                float x[8];
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const int i = KS * 32 * kh + 32 * ks + 8 * lg + e;
                    x[e] = xsrc[rt][(int64_t)(i < Hp ? i : Hp - 1) * D];
                }
#pragma unroll
                for (int e = 0; e < 8; ++e) asm volatile("" : "+v"(x[e]));
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const int i = KS * 32 * kh + 32 * ks + 8 * lg + e;
                    v[e] = i < Hp ? ((RS && !PAIRS) ? x[e] * rscale[rt] : x[e]) : 0.f;    // the W image is zero there; 0 * garbage must stay 0
                }
                }
                unsigned int w[NP][4];
#pragma unroll
                for (int pr = 0; pr < 4; ++pr) {
                    unsigned int pw[NP];
                    Pc::split(v[2 * pr], v[2 * pr + 1], pw);
#pragma unroll
                    for (int p = 0; p < NP; ++p) w[p][pr] = pw[p];
                }
#pragma unroll
                for (int p = 0; p < NP; ++p) a[ks][rt][p] = __builtin_bit_cast(op_t, (u32x4_t){w[p][0], w[p][1], w[p][2], w[p][3]});
            }
        for (int j0 = 0; j0 < m; j0 += FJ, ++u) {      // one staged chunk = (this half, fields j0 .. j0 + nf - 1)
          const int buf = u & 1;
          const int nf = min(FJ, m - j0);
          {
              const int cn = kh * m + j0 + nf;        // first field of the next chunk (the next half starts at a chunk boundary)
#ifdef CIN_ABL
              if (!(CIN_ABL_ON && (CIN_ABL & 4)))
#endif
              if (cn < nchunk) stage_w(cn, min(FJ, m - (j0 + nf < m ? j0 + nf : 0)), buf ^ 1);
          }
          // field f of the staged chunk
          auto field = [&](auto fc) __attribute__((always_inline)) {
            constexpr int f = decltype(fc)::value;
            constexpr int KSN = KS;
            const int j = j0 + f;
            const unsigned char* wl = wlane + buf * (FJ * CHB) + f * CHB;
            f32x4 xcur[RT], sd[RT];
#pragma unroll
            for (int rt = 0; rt < RT; ++rt) {
                if constexpr (PAIRS) xcur[rt] = (f32x4){1.f, 1.f, 1.f, 1.f};
                else if constexpr (DOT) xcur[rt] = (f32x4){x0lane[j * BT_ROWS + 16 * rt], 0.f, 0.f, 0.f};
                else xcur[rt] = *reinterpret_cast<const f32x4*>(x0lane + j * BT_ROWS + 16 * rt);
                if constexpr (RS && PAIRS) xcur[rt] *= rinv[rt];          // (not PAIRS: the slice in LDS already carries the rows' inverse scales)
                sd[rt] = (f32x4){0.f, 0.f, 0.f, 0.f};
            }
            {
                // A chunk's matrix instructions as ONE block -- column tile outermost, both k-steps inside, so that a tile's
                // KS * NMF instructions are one dependent chain on its accumulator --, then its accumulate fmas as one block.  The SIMD's
                // arbiter serves the oldest / highest-priority READY wave: a wave inside a dependent chain is not ready three cycles out of
                // four, and only then does its partner's VALU work get issued (tools/coexec_probe.hip: beside a stream of INDEPENDENT MFMAs,
                // or under a prioritised VALU stream, the partner gets nothing: the times add).  With the fmas interleaved between the chunk's
                // own matrix instructions both waves of a SIMD were in the same mixed phase all the time; as two blocks the partner's
                // block of fmas falls beside this wave's chains once the two drift apart: cin_backward 7.60 -> 7.46 ms on one box, the same
                // with s_setprio 0 / 1 / 2 / 3 around the chains, less (7.52-7.57) with the fmas first or with the halves of the workgroup
                // in opposite orders (profiles/r05_ab_dot_blocks.txt).
                op_t bq[2][KSN][NP];
#pragma unroll
                for (int ks = 0; ks < KSN; ++ks)
#pragma unroll
                    for (int p = 0; p < NP; ++p) bq[0][ks][p] = *reinterpret_cast<const op_t*>(wl + ks * STEPB + p * CT * 1024);
                __builtin_amdgcn_s_setprio(1);
#pragma unroll
                for (int ct = 0; ct < CT; ++ct) {
                    if (ct + 1 < CT) {
#ifdef CIN_ABL
                        if (!(CIN_ABL_ON && (CIN_ABL & 64)))          // no LDS reads behind the chunk's first
#endif
#pragma unroll
                        for (int ks = 0; ks < KSN; ++ks)
#pragma unroll
                            for (int p = 0; p < NP; ++p)
                                bq[(ct + 1) & 1][ks][p] = *reinterpret_cast<const op_t*>(wl + ks * STEPB + (ct + 1) * 1024 + p * CT * 1024);
                    }
#pragma unroll
                    for (int rt = 0; rt < RT; ++rt) {
                        f32x4 t = (f32x4){0.f, 0.f, 0.f, 0.f};
#ifdef CIN_ABL
                        if (CIN_ABL_ON && (CIN_ABL & 32)) asm volatile("" : "+v"(t) : "v"(bq[ct & 1][0][0]), "v"(bq[ct & 1][KSN - 1][NP - 1]), "v"(a[0][rt][0]), "v"(a[KSN - 1][rt][NP - 1])); else   // no MFMAs
#endif
#pragma unroll
                        for (int ks = 0; ks < KSN; ++ks) {
                            if constexpr (DOT) t = Pc::mma(bq[ct & 1][ks], a[ks][rt], t);     // W image x rows: the transposed tile
                            else t = Pc::mma(a[ks][rt], bq[ct & 1][ks], t);
                        }
                        T[0][rt][ct] = t;
                        __builtin_amdgcn_sched_barrier(0);
                    }
                }
                __builtin_amdgcn_s_setprio(0);
#pragma unroll
                for (int ct = 0; ct < CT; ++ct)
#pragma unroll
                    for (int rt = 0; rt < RT; ++rt) consume(rt, ct, T[0][rt][ct], sd, xcur);
                if constexpr (DOT) store_dot(sd, kh, j);
            }
          };
          field(std::integral_constant<int, 0>{});
          if constexpr (FJ == 2) {
              if (nf > 1) field(std::integral_constant<int, 1>{});
          }
          if constexpr (DOT) {
              // The interval's dot stores (one per field) are YOUNGER than the W pieces staged at its top: vector memory operations retire in
              // order, so "at most nf outstanding" means the pieces have landed -- without waiting for the stores' write acknowledgements
              // (vmcnt(0), also the fence inside __syncthreads(), cost a memory round trip per interval).
              if (nf > 1) asm volatile("s_waitcnt vmcnt(2) lgkmcnt(0)" ::: "memory");
              else asm volatile("s_waitcnt vmcnt(1) lgkmcnt(0)" ::: "memory");
#ifdef CIN_ABL
              if (!(CIN_ABL & 128))
#endif
              __builtin_amdgcn_s_barrier();
              asm volatile("" ::: "memory");
          } else {
          asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // this wave's W pieces of the next chunk have landed in LDS
          __syncthreads();
          }
        }
    }
    if constexpr (!RS && NP == 2 && !DOT && !PAIRS) {
        if (vwant == 1) {                                           // (uniform) the plain kernel under the verdict: the image carries W's tensor scale
            const float winv = pow2f(-w_scale_exp(wpart));
#pragma unroll
            for (int rt = 0; rt < RT; ++rt)
#pragma unroll
                for (int ct = 0; ct < CT; ++ct) out[rt][ct] *= winv;
        }
    }
    // ---- epilogue: C/D map of 16x16x32: col = lane & 15, row = 4*(lane >> 4) + reg
    if constexpr (DOT) {
        // transposed accumulators: the lane's row 16*rt + n, columns 16*ct + 4*lg + q (a store instruction writes 16 consecutive d of four columns)
#pragma unroll
        for (int rt = 0; rt < RT; ++rt) {
            const int64_t gr = row0 + wave * WR + rt * 16 + n;
            const int64_t b = (gr < R ? gr : R - 1) >> dshift;
            const int d = (int)(gr & (D - 1));
            float rmx = 0.f;
            if (addp) {                                              // (uniform) all of the lane's pooled-gradient terms in one batch of loads
                f32x4 ap[CT];
#pragma unroll
                for (int ct = 0; ct < CT; ++ct)
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const int h = hbase + 16 * ct + 4 * lg + q;
                        ap[ct][q] = addp[b * addp_ld + (h < H ? h : H - 1)];
                    }
#pragma unroll
                for (int ct = 0; ct < CT; ++ct)
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        asm volatile("" : "+v"(ap[ct][q]));
                        out[rt][ct][q] += ap[ct][q];
                    }
            }
#pragma unroll
            for (int ct = 0; ct < CT; ++ct) {
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int h = hbase + 16 * ct + 4 * lg + q;
                    if (xout && h < H && gr < R) {
                        const float v = out[rt][ct][q];
#ifdef CIN_ABL
                        if ((CIN_ABL & 256) && v != 12345.678f) {} else      // timing ablation: no data-gradient stores
#endif
                        xout[(b * H + h) * D + d] = v;
                        rmx = fmaxf(rmx, fabsf(v));
                    }
                }
            }
            if (xout_bits) {
                rmx = fmaxf(rmx, __shfl_xor(rmx, 16, 64));
                rmx = fmaxf(rmx, __shfl_xor(rmx, 32, 64));
                if (lg == 0 && gr < R) xout_bits[gr] = __builtin_bit_cast(unsigned int, rmx);
            }
        }
    } else {
#pragma unroll
    for (int rt = 0; rt < RT; ++rt) {
        const int64_t gr = row0 + wave * WR + rt * 16 + 4 * lg;     // first of the lane's 4 consecutive rows (same sample: D >= 4)
        const int64_t b = gr >> dshift;
        const int d = (int)(gr & (D - 1));
        f32x4 rmx = (f32x4){0.f, 0.f, 0.f, 0.f};                   // xout_bits: the largest |output| of the lane's columns, per row
#pragma unroll
        for (int ct = 0; ct < CT; ++ct) {
            const int h = hbase + 16 * ct + n;
            f32x4 v = out[rt][ct];
            if (xout && h < H && gr < R) {
                // (the backward of a stack: dL/dxout of the layer below = this data gradient + that layer's pooled gradient, broadcast over d)
                if (addp) v += addp[b * addp_ld + h];
                *reinterpret_cast<f32x4*>(xout + (b * H + h) * D + d) = v;
#pragma unroll
                for (int q = 0; q < 4; ++q) rmx[q] = fmaxf(rmx[q], fabsf(v[q]));
            }
        }
        if (xout_bits) {                                            // (uniform) the rows' maxima for the next layer's row scales: its prologue need not scan xk
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                float mq = rmx[q];
#pragma unroll
                for (int o = 1; o < 16; o <<= 1) mq = fmaxf(mq, __shfl_xor(mq, o, 64));
                if (n == 0 && gr + q < R) xout_bits[gr + q] = __builtin_bit_cast(unsigned int, mq);
            }
        }
    }
    }
    if (pooled) {
        static_assert(RT == 2 || DOT, "the pooled sums are written by the forward configuration (two row tiles per wave)");
        if constexpr (RT == 2 && !DOT) {
#pragma unroll
        for (int ct = 0; ct < CT; ++ct) {
            const int h = hbase + 16 * ct + n;
            float s[2];
#pragma unroll
            for (int rt = 0; rt < 2; ++rt) {
                const f32x4 v = out[rt][ct];
                float p = (v[0] + v[1]) + (v[2] + v[3]);          // rows 4*lg .. 4*lg+3 of the tile
                if (D >= 8) p += __shfl_xor(p, 16, 64);           // lane groups of one sample
                if (D >= 16) p += __shfl_xor(p, 32, 64);
                s[rt] = p;
            }
            if (D == 32) { s[0] += s[1]; }
            const bool writer = D == 4 ? true : (D == 8 ? (lg & 1) == 0 : lg == 0);
#pragma unroll
            for (int rt = 0; rt < 2; ++rt) {
                if (D == 32 && rt == 1) continue;
                const int64_t gr = row0 + wave * 32 + rt * 16 + 4 * lg;
                if (writer && h < H && gr < R) pooled[(gr >> dshift) * pooled_ld + h] = s[rt];
            }
        }
        }
    }
}

}  // namespace dir

using namespace dir;

// Shape plan shared by the workspace query and the launcher: halves of i, column blocks
struct Bf3Plan { int KS, nkh, bw, nfull, ctl; int64_t chunks, bytes_full, bytes_last; };
static Bf3Plan bf3_plan(int m, int Hp, int H, bool dot, int np = 3) {
    Bf3Plan p;
    p.KS = Hp <= 32 ? 1 : 2;
    p.nkh = (Hp + p.KS * 32 - 1) / (p.KS * 32);
    p.bw = dot ? 4 : 8;                                                       // tiles per full column block: 128 columns, 64 in the dot form
    p.nfull = H / (16 * p.bw);
    const int r = H - 16 * p.bw * p.nfull;
    p.ctl = r ? ((r + 15) / 16 + 1) / 2 * 2 : 0;                              // tiles of the last block: 2, 4, 6 or 8 (dot form: 2 or 4)
    p.chunks = (int64_t)p.nkh * m;
    p.bytes_full = (int64_t)p.nfull * p.chunks * p.KS * np * p.bw * 1024;
    p.bytes_last = p.chunks * p.KS * np * p.ctl * 1024;
    return p;
}

extern "C" int64_t dir_cin_bf16x3_workspace_bytes(int m, int Hp, int H) {
    if (m <= 0 || Hp <= 0 || H <= 0) return 0;
    const Bf3Plan p = bf3_plan(m, Hp, H, false), q = bf3_plan(m, Hp, H, true);      // either form of the layer
    const int64_t a = p.bytes_full + p.bytes_last, b = q.bytes_full + q.bytes_last;
    return (a > b ? a : b) + 128 + 4 * (WPARTS + XPARTS) + 64;      // + the partial maxima of W (and of xk's row maxima: the verdict) behind the image, + the dot form's sink word
}

// Shared launcher of the forward (y == nullptr) and the data-gradient form (y, dotp given: 64-column blocks, two fields per chunk, dot partials)
static int bf3_run(const char* name, const float* x0, const float* xk, const float* W, int m, int Hp, int H, int D, int64_t B, float* xout,
                   float* pooled, int64_t pooled_ld, const float* y, float* dotp, void* workspace, int64_t workspace_bytes, dir_stream_t stream,
                   const float* addp = nullptr, int64_t addp_ld = 0, int np = 3 /* 2: fp16 x 2 */,
                   bool rs = false /* fp16 x 2 with the left operand scaled per row (a gradient) */, unsigned int* amax_out = nullptr,
                   const unsigned int* xk_bits = nullptr, unsigned int* xout_bits = nullptr,
                   const unsigned int* verdict_bits = nullptr /* rs forward: xk's row maxima [B * D] -> the device-side plain / row-scaled verdict */,
                   const int64_t* x0_inv = nullptr /* x0 is a row list read through these positions [B, m] */) {
    DIR_CHECK_ARG(m > 0 && Hp > 0 && H > 0 && D > 0 && B >= 0, "%s: m=%d Hp=%d H=%d D=%d", name, m, Hp, H, D);
    if (B == 0) return DIR_OK;                      // nothing to compute or write (empty tensors have no storage: their pointers may be null)
    DIR_CHECK_ARG(x0 && xk && W && (xout || pooled) && workspace, "%s: null pointer", name);
    DIR_CHECK_ARG(!pooled || pooled_ld >= H, "%s: pooled_ld=%lld < H=%d", name, (long long)pooled_ld, H);
    if (!(D == 4 || D == 8 || D == 16 || D == 32)) return fail(DIR_E_UNSUPPORTED, "%s: D=%d (supported: 4, 8, 16, 32)", name, D);
    if (m > 40) return fail(DIR_E_UNSUPPORTED, "%s: field count m=%d exceeds 40 (LDS-resident x0 slice)", name, m);
    if (xout && !aligned16(xout)) return fail(DIR_E_BADARG, "%s: xout must be 16-byte aligned", name);
    DIR_CHECK_ARG(aligned16(workspace) && workspace_bytes >= dir_cin_bf16x3_workspace_bytes(m, Hp, H),
                  "%s: workspace must be 16-byte aligned and hold dir_cin_bf16x3_workspace_bytes(m, Hp, H) bytes", name);
    int dshift = 0;
    while ((1 << dshift) < D) ++dshift;
    const int64_t R = B * D;
    hipStream_t st = as_stream(stream);
    const bool dot = y != nullptr;
    if (amax_out && zero_async(amax_out, sizeof(unsigned int), st) != hipSuccess) return fail(DIR_E_HIP, "%s: zeroing failed", name);
    if (dot && np != 3 && !rs) return fail(DIR_E_UNSUPPORTED, "%s: the data-gradient form on fp16 x 2 needs the row-scaled left operand", name);
    if (rs && np != 2) return fail(DIR_E_UNSUPPORTED, "%s: row scaling belongs to fp16 x 2", name);
    const Bf3Plan pl = bf3_plan(m, Hp, H, dot, np);
    if (xout_bits && (pl.nfull + (pl.ctl ? 1 : 0) != 1 || !xout))
        return fail(DIR_E_UNSUPPORTED, "%s: xout_row_bits needs xout and ONE column block (H <= %d)", name, 16 * pl.bw);
    unsigned char* img = static_cast<unsigned char*>(workspace);
    // rs: W is scaled too (one power of two for the tensor): its partial maxima sit behind the image
    float* wtail = reinterpret_cast<float*>(img + ((pl.bytes_full + pl.bytes_last + 127) & ~(int64_t)127));
    float* wpart = rs ? wtail : nullptr;
    float* sink = wtail + WPARTS + XPARTS;
    if (rs) hipLaunchKernelGGL(cin_w_absmax_k, dim3(WPARTS), dim3(256), 0, st, W, (int64_t)H * Hp * m, wpart);
    const int verdict = (rs && verdict_bits && !dot) ? 1 : 0;
    if (verdict) hipLaunchKernelGGL(cin_bits_absmax_k, dim3(XBLOCKS), dim3(256), 0, st, verdict_bits, R, wpart + WPARTS);
    auto pack = [&](int ncb, int CT, int hoff, unsigned char* dst) {
        const int64_t threads = (int64_t)ncb * pl.chunks * pl.KS * CT * 64 * 4;
        if (np == 2)
            hipLaunchKernelGGL(cin_bf3_pack_w_k<2>, dim3((unsigned)((threads + 255) / 256)), dim3(256), 0, st, W, m, Hp, H, pl.KS, pl.nkh, ncb, CT, hoff,
                               reinterpret_cast<unsigned int*>(dst), wpart, verdict);
        else
            hipLaunchKernelGGL(cin_bf3_pack_w_k<3>, dim3((unsigned)((threads + 255) / 256)), dim3(256), 0, st, W, m, Hp, H, pl.KS, pl.nkh, ncb, CT, hoff,
                               reinterpret_cast<unsigned int*>(dst));
    };
    if (pl.nfull) pack(pl.nfull, pl.bw, 0, img);
    if (pl.ctl) pack(1, pl.ctl, 16 * pl.bw * pl.nfull, img + pl.bytes_full);
    const unsigned nrb = (unsigned)((R + 255) / 256);
    const int64_t dot_block = (int64_t)pl.nkh * B * m * D;         // floats of dot partials per column block
#define BT_LAUNCH(K, C, DOT_, FJ_, NP_, RS_, NCB, HOFF, IMG, DOTP)                                                                    \
    do {                                                                                                                              \
        static LdsOnce once;                                                                                                      \
        (void)lds_limit(once, 160 * 1024, &cin_bf3_k<K, C, 2, DOT_, FJ_, false, NP_, RS_>);                                       \
        const size_t shmem = 2 * (size_t)FJ_ * K * NP_ * C * 1024 + sizeof(float) * (size_t)m * 256 + 32;                             \
        hipLaunchKernelGGL((cin_bf3_k<K, C, 2, DOT_, FJ_, false, NP_, RS_>), dim3(nrb, (unsigned)(NCB)), dim3(512), shmem, st, x0, xk, IMG, m, Hp, H, D, \
                           dshift, pl.nkh, HOFF, R, xout, pooled, pooled_ld, (DOT_) ? y : reinterpret_cast<const float*>(x0_inv), DOTP, addp, addp_ld, nullptr, m, (HOFF) == 0 ? amax_out : nullptr, wpart, \
                           xk_bits, xout_bits, verdict ? ((RS_) ? 0 : 1) : -1, sink);                                                 \
    } while (0)
#define BT_LAUNCH_KS(C, DOT_, FJ_, NP_, RS_, NCB, HOFF, IMG, DOTP)                       \
    do {                                                                                 \
        if (pl.KS == 1) BT_LAUNCH(1, C, DOT_, FJ_, NP_, RS_, NCB, HOFF, IMG, DOTP);      \
        else BT_LAUNCH(2, C, DOT_, FJ_, NP_, RS_, NCB, HOFF, IMG, DOTP);                 \
    } while (0)
#define BT_LAUNCH_FWD(C, NCB, HOFF, IMG)                                                 \
    do {                                                                                 \
        if (verdict) BT_LAUNCH_KS(C, false, 1, 2, false, NCB, HOFF, IMG, nullptr);       \
        if (rs) BT_LAUNCH_KS(C, false, 1, 2, true, NCB, HOFF, IMG, nullptr);             \
        else if (np == 2) BT_LAUNCH_KS(C, false, 1, 2, false, NCB, HOFF, IMG, nullptr);  \
        else BT_LAUNCH_KS(C, false, 1, 3, false, NCB, HOFF, IMG, nullptr);               \
    } while (0)
#define BT_LAUNCH_DOT(C, NCB, HOFF, IMG, DOTP)                                      \
    do {                                                                            \
        if (rs) BT_LAUNCH_KS(C, true, 2, 2, true, NCB, HOFF, IMG, DOTP);            \
        else BT_LAUNCH_KS(C, true, 2, 3, false, NCB, HOFF, IMG, DOTP);              \
    } while (0)
    const unsigned char* li = img + pl.bytes_full;
    const int lo = 16 * pl.bw * pl.nfull;
    if (!dot) {
        if (pl.nfull) BT_LAUNCH_FWD(8, pl.nfull, 0, img);
        switch (pl.ctl) {
            case 0: break;
            case 2: BT_LAUNCH_FWD(2, 1, lo, li); break;
            case 4: BT_LAUNCH_FWD(4, 1, lo, li); break;
            case 6: BT_LAUNCH_FWD(6, 1, lo, li); break;
            default: BT_LAUNCH_FWD(8, 1, lo, li); break;
        }
    } else {
        float* ld_ = dotp + pl.nfull * dot_block;
        if (pl.nfull) BT_LAUNCH_DOT(4, pl.nfull, 0, img, dotp);
        switch (pl.ctl) {
            case 0: break;
            case 2: BT_LAUNCH_DOT(2, 1, lo, li, ld_); break;
            default: BT_LAUNCH_DOT(4, 1, lo, li, ld_); break;
        }
    }
#undef BT_LAUNCH_DOT
#undef BT_LAUNCH_FWD
#undef BT_LAUNCH_KS
#undef BT_LAUNCH
    DIR_CHECK_LAUNCH(name);
    return DIR_OK;
}

extern "C" int dir_cin_layer_bf16x3_f32(const float* x0, const float* xk, const float* W, int m, int Hp, int H, int D, int64_t B,
                                        float* xout, float* pooled, int64_t pooled_ld, void* workspace, int64_t workspace_bytes,
                                        dir_stream_t stream) {
    return bf3_run("dir_cin_layer_bf16x3_f32", x0, xk, W, m, Hp, H, D, B, xout, pooled, pooled_ld, nullptr, nullptr, workspace, workspace_bytes,
                   stream);
}

extern "C" int dir_cin_layer_f16x2_f32(const float* x0, const float* xk, const float* W, int m, int Hp, int H, int D, int64_t B,
                                       float* xout, float* pooled, int64_t pooled_ld, void* workspace, int64_t workspace_bytes,
                                       dir_stream_t stream) {
    return bf3_run("dir_cin_layer_f16x2_f32", x0, xk, W, m, Hp, H, D, B, xout, pooled, pooled_ld, nullptr, nullptr, workspace, workspace_bytes,
                   stream, nullptr, 0, 2);
}

// The forward contraction with a left operand of unknown magnitude (a gradient: the backward's forward-form contractions): fp16 x 2 with
// every row of xk scaled by a power of two inside the kernel (template parameter RS)
extern "C" int dir_cin_layer_grad_f16x2_f32(const float* x0, const float* xk, const float* W, int m, int Hp, int H, int D, int64_t B,
                                            float* xout, float* pooled, int64_t pooled_ld, void* workspace, int64_t workspace_bytes,
                                            unsigned int* xk_absmax_bits_out, dir_stream_t stream) {
    return bf3_run("dir_cin_layer_grad_f16x2_f32", x0, xk, W, m, Hp, H, D, B, xout, pooled, pooled_ld, nullptr, nullptr, workspace,
                   workspace_bytes, stream, nullptr, 0, 2, true, xk_absmax_bits_out);
}

// The FORWARD layer on scaled fp16 x 2 (round 5: what "auto" runs for every layer but the first): as the entry above, and
//   xk_row_bits   (optional, [B * D], row r = b * D + d): the bit pattern of max_i |xk[b, i, d]| left by the layer that produced xk (then the
//                 prologue reads one word per row instead of the row);
//   xout_row_bits (optional, [B * D]): the same of this layer's output, for the next layer (needs xout and H <= 128: one column block).
extern "C" int dir_cin_layer_rows_f16x2_f32(const float* x0, const float* xk, const float* W, int m, int Hp, int H, int D, int64_t B,
                                            float* xout, float* pooled, int64_t pooled_ld, void* workspace, int64_t workspace_bytes,
                                            const unsigned int* xk_row_bits, unsigned int* xout_row_bits, dir_stream_t stream) {
    return bf3_run("dir_cin_layer_rows_f16x2_f32", x0, xk, W, m, Hp, H, D, B, xout, pooled, pooled_ld, nullptr, nullptr, workspace,
                   workspace_bytes, stream, nullptr, 0, 2, true, nullptr, xk_row_bits, xout_row_bits);
}

// ... with the DEVICE-SIDE verdict (see cin_plain_verdict): xk_row_bits is required (the producer's row maxima); inside the window the plain
// fp16 x 2 kernel does the work, outside it the row-scaled one -- both are launched, one of them leaves at once.  xout_row_bits as above.
extern "C" int dir_cin_layer_auto_f16x2_f32(const float* x0, const float* xk, const float* W, int m, int Hp, int H, int D, int64_t B,
                                            float* xout, float* pooled, int64_t pooled_ld, void* workspace, int64_t workspace_bytes,
                                            const unsigned int* xk_row_bits, unsigned int* xout_row_bits, dir_stream_t stream) {
    DIR_CHECK_ARG(xk_row_bits || B == 0, "dir_cin_layer_auto_f16x2_f32: xk_row_bits is null (dir_cin_layer_rows_f16x2_f32 takes no verdict)");
    return bf3_run("dir_cin_layer_auto_f16x2_f32", x0, xk, W, m, Hp, H, D, B, xout, pooled, pooled_ld, nullptr, nullptr, workspace,
                   workspace_bytes, stream, nullptr, 0, 2, true, nullptr, nullptr, xout_row_bits, xk_row_bits);
}

// ---- the first layer over field pairs (PAIRS) -----------------------------------------------------------------------------------------------
namespace dir {
__host__ __device__ __forceinline__ int l1_pair_index(int a, int b, int m) { return a * m - a * (a - 1) / 2 + (b - a); }      // a <= b, row-major over a

// W [H, m*m] -> W2 [H, np] (np = m (m + 1) / 2): W2[h, p(i <= j)] = W[h,i,j] + W[h,j,i] (W[h,i,i] on the diagonal);
// ptab[p] = i | j << 8 for p < np, 0 up to npad
__global__ __launch_bounds__(256) void cin_l1_pairs_k(const float* __restrict__ W, int m, int H, int np, int npad, float* __restrict__ W2,
                                                       unsigned short* __restrict__ ptab) {
    const int64_t total = (int64_t)H * m * m;
    for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < total; e += (int64_t)gridDim.x * 256) {
        const int j = (int)(e % m), i = (int)((e / m) % m), h = (int)(e / ((int64_t)m * m));
        if (i > j) continue;
        const float w = i == j ? W[e] : W[e] + W[((int64_t)h * m + j) * m + i];
        W2[(int64_t)h * np + l1_pair_index(i, j, m)] = w;
        if (h == 0) ptab[l1_pair_index(i, j, m)] = (unsigned short)(i | (j << 8));
    }
    for (int64_t p = (int64_t)blockIdx.x * 256 + threadIdx.x; p < npad - np; p += (int64_t)gridDim.x * 256) ptab[np + p] = 0;
}
}  // namespace dir

struct L1Plan { int np, npad; Bf3Plan pl; int64_t off_w2, off_tab, off_part, total; };
static L1Plan l1_plan(int m, int H, int pieces = 3) {
    L1Plan q;
    q.np = m * (m + 1) / 2;
    q.pl = bf3_plan(1, q.np, H, false, pieces);               // one "field", the pairs as the reduction channels
    q.npad = q.pl.nkh * q.pl.KS * 32;
    int64_t off = q.pl.bytes_full + q.pl.bytes_last;
    off = (off + 255) & ~(int64_t)255;
    q.off_w2 = off;
    off += ((int64_t)H * q.np * 4 + 255) & ~(int64_t)255;
    q.off_tab = off;
    off += ((int64_t)q.npad * 2 + 255) & ~(int64_t)255;
    q.off_part = off;                                          // WPARTS partial maxima of W2 (the scaled fp16 x 2 form)
    off += 256;
    q.total = off;
    return q;
}

extern "C" int64_t dir_cin_layer1_bf16x3_workspace_bytes(int m, int H) {
    if (m <= 0 || H <= 0) return 0;
    return l1_plan(m, H).total;
}

static int l1_run(const char* name, int pieces, const float* x0, const float* W, int m, int H, int D, int64_t B, float* xout, float* pooled,
                  int64_t pooled_ld, void* workspace, int64_t workspace_bytes, dir_stream_t stream, unsigned int* xout_bits = nullptr,
                  const int64_t* x0_inv = nullptr) {
    DIR_CHECK_ARG(m > 0 && H > 0 && D > 0 && B >= 0, "%s: m=%d H=%d D=%d", name, m, H, D);
    if (B == 0) return DIR_OK;
    DIR_CHECK_ARG(x0 && W && (xout || pooled) && workspace, "%s: null pointer", name);
    DIR_CHECK_ARG(!pooled || pooled_ld >= H, "%s: pooled_ld=%lld < H=%d", name, (long long)pooled_ld, H);
    if (!(D == 4 || D == 8 || D == 16 || D == 32)) return fail(DIR_E_UNSUPPORTED, "%s: D=%d (supported: 4, 8, 16, 32)", name, D);
    if (m < 8 || m > 40) return fail(DIR_E_UNSUPPORTED, "%s: m=%d (supported: 8..40; use dir_cin_layer_bf16x3_f32)", name, m);
    if (xout && !aligned16(xout)) return fail(DIR_E_BADARG, "%s: xout must be 16-byte aligned", name);
    const L1Plan q = l1_plan(m, H, pieces);
    DIR_CHECK_ARG((reinterpret_cast<uintptr_t>(workspace) & 255u) == 0 && workspace_bytes >= q.total,
                  "%s: workspace must be 256-byte aligned and hold dir_cin_layer1_bf16x3_workspace_bytes(m, H) bytes", name);
    const Bf3Plan& pl = q.pl;                                  // KS = 2 (m >= 8: more than 32 pairs)
    if (xout_bits && (pl.nfull + (pl.ctl ? 1 : 0) != 1 || !xout))
        return fail(DIR_E_UNSUPPORTED, "%s: xout_row_bits needs xout and ONE column block (H <= 128)", name);
    int dshift = 0;
    while ((1 << dshift) < D) ++dshift;
    const int64_t R = B * D;
    hipStream_t st = as_stream(stream);
    unsigned char* img = static_cast<unsigned char*>(workspace);
    float* W2 = reinterpret_cast<float*>(img + q.off_w2);
    unsigned short* ptab = reinterpret_cast<unsigned short*>(img + q.off_tab);
    hipLaunchKernelGGL(cin_l1_pairs_k, dim3(grid_for(((int64_t)H * m * m + 255) / 256)), dim3(256), 0, st, W, m, H, q.np, q.npad, W2, ptab);
    // fp16 x 2: both operands scaled by exact powers of two (rows of the x0 slice inside the kernel, the pair weights as a tensor): no
    // promise about the magnitudes of x0 or W is needed
    float* wpart = pieces == 2 ? reinterpret_cast<float*>(img + q.off_part) : nullptr;
    if (wpart) hipLaunchKernelGGL(cin_w_absmax_k, dim3(WPARTS), dim3(256), 0, st, W2, (int64_t)H * q.np, wpart);
    auto pack = [&](int ncb, int CT, int hoff, unsigned char* dst) {
        const int64_t threads = (int64_t)ncb * pl.chunks * pl.KS * CT * 64 * 4;
        if (pieces == 2)
            hipLaunchKernelGGL(cin_bf3_pack_w_k<2>, dim3((unsigned)((threads + 255) / 256)), dim3(256), 0, st, W2, 1, q.np, H, pl.KS, pl.nkh, ncb, CT,
                               hoff, reinterpret_cast<unsigned int*>(dst), wpart);
        else
            hipLaunchKernelGGL(cin_bf3_pack_w_k<3>, dim3((unsigned)((threads + 255) / 256)), dim3(256), 0, st, W2, 1, q.np, H, pl.KS, pl.nkh, ncb, CT,
                               hoff, reinterpret_cast<unsigned int*>(dst));
    };
    if (pl.nfull) pack(pl.nfull, pl.bw, 0, img);
    if (pl.ctl) pack(1, pl.ctl, 16 * pl.bw * pl.nfull, img + pl.bytes_full);
    const unsigned nrb = (unsigned)((R + 255) / 256);
#define L1_LAUNCH_NP(C, NP_, NCB, HOFF, IMG)                                                                                            \
    do {                                                                                                                                \
        static LdsOnce once;                                                                                                      \
        (void)lds_limit(once, 160 * 1024, &cin_bf3_k<2, C, 2, false, 1, true, NP_, NP_ == 2>);                                    \
        const size_t shmem = 2 * (size_t)2 * NP_ * C * 1024 + sizeof(float) * (size_t)m * 256 + 32;                                     \
        hipLaunchKernelGGL((cin_bf3_k<2, C, 2, false, 1, true, NP_, NP_ == 2>), dim3(nrb, (unsigned)(NCB)), dim3(512), shmem, st, x0, x0, IMG, 1, q.np, H, D, \
                           dshift, pl.nkh, HOFF, R, xout, pooled, pooled_ld, reinterpret_cast<const float*>(x0_inv), nullptr, nullptr, 0, ptab, m, nullptr, wpart, nullptr, xout_bits); \
    } while (0)
#define L1_LAUNCH(C, NCB, HOFF, IMG)                           \
    do {                                                       \
        if (pieces == 2) L1_LAUNCH_NP(C, 2, NCB, HOFF, IMG);   \
        else L1_LAUNCH_NP(C, 3, NCB, HOFF, IMG);               \
    } while (0)
    const unsigned char* li = img + pl.bytes_full;
    const int lo = 16 * pl.bw * pl.nfull;
    if (pl.nfull) L1_LAUNCH(8, pl.nfull, 0, img);
    switch (pl.ctl) {
        case 0: break;
        case 2: L1_LAUNCH(2, 1, lo, li); break;
        case 4: L1_LAUNCH(4, 1, lo, li); break;
        case 6: L1_LAUNCH(6, 1, lo, li); break;
        default: L1_LAUNCH(8, 1, lo, li); break;
    }
#undef L1_LAUNCH
#undef L1_LAUNCH_NP
    DIR_CHECK_LAUNCH(name);
    return DIR_OK;
}

extern "C" int dir_cin_layer1_bf16x3_f32(const float* x0, const float* W, int m, int H, int D, int64_t B, float* xout, float* pooled,
                                         int64_t pooled_ld, void* workspace, int64_t workspace_bytes, dir_stream_t stream) {
    return l1_run("dir_cin_layer1_bf16x3_f32", 3, x0, W, m, H, D, B, xout, pooled, pooled_ld, workspace, workspace_bytes, stream);
}

extern "C" int dir_cin_layer1_f16x2_f32(const float* x0, const float* W, int m, int H, int D, int64_t B, float* xout, float* pooled,
                                        int64_t pooled_ld, void* workspace, int64_t workspace_bytes, dir_stream_t stream) {
    return l1_run("dir_cin_layer1_f16x2_f32", 2, x0, W, m, H, D, B, xout, pooled, pooled_ld, workspace, workspace_bytes, stream);
}

// ... leaving the rows' maxima of xout ([B * D] words, row r = b * D + d: the bit pattern of max_h |xout[b, h, d]|) for the next layer's
// row scales (dir_cin_layer_rows_f16x2_f32); needs xout and H <= 128
extern "C" int dir_cin_layer1_bits_f16x2_f32(const float* x0, const float* W, int m, int H, int D, int64_t B, float* xout, float* pooled,
                                             int64_t pooled_ld, void* workspace, int64_t workspace_bytes, unsigned int* xout_row_bits,
                                             dir_stream_t stream) {
    return l1_run("dir_cin_layer1_bits_f16x2_f32", 2, x0, W, m, H, D, B, xout, pooled, pooled_ld, workspace, workspace_bytes, stream, xout_row_bits);
}

// The two forward layers of an inference stack with x0 READ THROUGH INVERSE POSITIONS (round 6: the sharded lookup without its finish pass,
// ShardedTables.lookup_rows): x0_rows [n, D] is the row list the exchange left, x0_inv [B, m] int64 the position of (sample, field)'s row in it
// (< 0: a zero row -- a pruned or out-of-range id).  Only the staging of the workgroup's x0 slice differs from the entries above (one int64
// per (sample, field) more, the rows come from where they are): results are bit for bit those of the plain entries on the materialised x0.
//   layer 1 (xk = x0, field pairs, rows scaled inside the kernel; xout_row_bits optional as in dir_cin_layer1_bits_f16x2_f32)
extern "C" int dir_cin_layer1_f16x2_gather_f32(const float* x0_rows, const int64_t* x0_inv, const float* W, int m, int H, int D, int64_t B, float* xout,
                                               float* pooled, int64_t pooled_ld, void* workspace, int64_t workspace_bytes, unsigned int* xout_row_bits,
                                               dir_stream_t stream) {
    DIR_CHECK_ARG(x0_inv || B == 0, "dir_cin_layer1_f16x2_gather_f32: x0_inv is null");
    return l1_run("dir_cin_layer1_f16x2_gather_f32", 2, x0_rows, W, m, H, D, B, xout, pooled, pooled_ld, workspace, workspace_bytes, stream, xout_row_bits,
                  x0_inv);
}

//   a later layer (xk [B, Hp, D] contiguous as the previous layer wrote it): dir_cin_layer_auto_f16x2_f32 when xk_row_bits is given (the
//   device-side plain / row-scaled verdict), dir_cin_layer_rows_f16x2_f32 otherwise
extern "C" int dir_cin_layer_f16x2_gather_f32(const float* x0_rows, const int64_t* x0_inv, const float* xk, const float* W, int m, int Hp, int H, int D,
                                              int64_t B, float* xout, float* pooled, int64_t pooled_ld, void* workspace, int64_t workspace_bytes,
                                              const unsigned int* xk_row_bits, unsigned int* xout_row_bits, dir_stream_t stream) {
    DIR_CHECK_ARG(x0_inv || B == 0, "dir_cin_layer_f16x2_gather_f32: x0_inv is null");
    return bf3_run("dir_cin_layer_f16x2_gather_f32", x0_rows, xk, W, m, Hp, H, D, B, xout, pooled, pooled_ld, nullptr, nullptr, workspace, workspace_bytes,
                   stream, nullptr, 0, 2, true, nullptr, nullptr, xout_row_bits, xk_row_bits, x0_inv);
}

extern "C" int dir_cin_bf16x3_dot_partials(int m, int Hp, int H) {
    if (m <= 0 || Hp <= 0 || H <= 0) return 0;
    const Bf3Plan p = bf3_plan(m, Hp, H, true);
    return p.nkh * (p.nfull + (p.ctl ? 1 : 0));
}

extern "C" int dir_cin_layer_dot_bf16x3_f32(const float* x0, const float* xk, const float* W, const float* y, int m, int Hp, int H, int D,
                                            int64_t B, float* xout, float* dot_partials, void* workspace, int64_t workspace_bytes,
                                            dir_stream_t stream) {
    const char* name = "dir_cin_layer_dot_bf16x3_f32";
    if (B == 0) return DIR_OK;
    DIR_CHECK_ARG(y && dot_partials && xout, "%s: null pointer", name);
    DIR_CHECK_ARG(aligned16(y) && aligned16(dot_partials), "%s: y and dot_partials must be 16-byte aligned", name);
    return bf3_run(name, x0, xk, W, m, Hp, H, D, B, xout, nullptr, 0, y, dot_partials, workspace, workspace_bytes, stream);
}

extern "C" int dir_cin_layer_dot_add_bf16x3_f32(const float* x0, const float* xk, const float* W, const float* y, int m, int Hp, int H, int D,
                                                int64_t B, const float* add_pooled, int64_t add_pooled_ld, float* xout, float* dot_partials,
                                                void* workspace, int64_t workspace_bytes, dir_stream_t stream) {
    const char* name = "dir_cin_layer_dot_add_bf16x3_f32";
    if (B == 0) return DIR_OK;
    DIR_CHECK_ARG(y && dot_partials && xout, "%s: null pointer", name);
    DIR_CHECK_ARG(aligned16(y) && aligned16(dot_partials), "%s: y and dot_partials must be 16-byte aligned", name);
    DIR_CHECK_ARG(!add_pooled || add_pooled_ld >= H, "%s: add_pooled_ld=%lld < H=%d", name, (long long)add_pooled_ld, H);
    return bf3_run(name, x0, xk, W, m, Hp, H, D, B, xout, nullptr, 0, y, dot_partials, workspace, workspace_bytes, stream, add_pooled, add_pooled_ld);
}

extern "C" int dir_cin_layer_dot_add_f16x2_f32(const float* x0, const float* xk, const float* W, const float* y, int m, int Hp, int H, int D,
                                               int64_t B, const float* add_pooled, int64_t add_pooled_ld, float* xout, float* dot_partials,
                                               void* workspace, int64_t workspace_bytes, unsigned int* xk_absmax_bits_out, dir_stream_t stream) {
    const char* name = "dir_cin_layer_dot_add_f16x2_f32";
    if (B == 0) return DIR_OK;
    DIR_CHECK_ARG(y && dot_partials && xout, "%s: null pointer", name);
    DIR_CHECK_ARG(aligned16(y) && aligned16(dot_partials), "%s: y and dot_partials must be 16-byte aligned", name);
    DIR_CHECK_ARG(!add_pooled || add_pooled_ld >= H, "%s: add_pooled_ld=%lld < H=%d", name, (long long)add_pooled_ld, H);
    return bf3_run(name, x0, xk, W, m, Hp, H, D, B, xout, nullptr, 0, y, dot_partials, workspace, workspace_bytes, stream, add_pooled, add_pooled_ld,
                   2, true, xk_absmax_bits_out);
}

// out[e] (+)= sum over p of parts[p][e] in p order, e < n (n % 4 == 0, 16-byte aligned): the dot partials of dir_cin_layer_dot_*_f32 -> dx0,
// accumulated over the layers of a stack in one pass instead of a library reduction plus an add
__global__ __launch_bounds__(256) void sum_partials_k(const float* __restrict__ parts, int P, int64_t n4, int accumulate, float* __restrict__ out) {
    for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < n4; e += (int64_t)gridDim.x * 256) {
        f32x4 s = accumulate ? reinterpret_cast<const f32x4*>(out)[e] : (f32x4){0.f, 0.f, 0.f, 0.f};
        for (int p = 0; p < P; ++p) s += reinterpret_cast<const f32x4*>(parts)[(int64_t)p * n4 + e];
        reinterpret_cast<f32x4*>(out)[e] = s;
    }
}

extern "C" int dir_sum_partials_f32(const float* parts, int P, int64_t n, int accumulate, float* out, dir_stream_t stream) {
    const char* name = "dir_sum_partials_f32";
    DIR_CHECK_ARG(P >= 0 && n >= 0, "%s: P=%d n=%lld", name, P, (long long)n);
    if (n == 0 || (P == 0 && accumulate)) return DIR_OK;
    DIR_CHECK_ARG(out && (parts || P == 0), "%s: null pointer", name);
    if (n % 4 || !aligned16(parts) || !aligned16(out)) return fail(DIR_E_UNSUPPORTED, "%s: n must be a multiple of 4, parts / out 16-byte aligned", name);
    hipLaunchKernelGGL(sum_partials_k, dim3(grid_for((n / 4 + 255) / 256, 16)), dim3(256), 0, as_stream(stream), parts, P, n / 4, accumulate, out);
    DIR_CHECK_LAUNCH(name);
    return DIR_OK;
}
