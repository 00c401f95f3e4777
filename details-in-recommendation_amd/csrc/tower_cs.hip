// tower_cs.hip -- the one-launch DNN tower (dnn_logit_fn, models/DeepFM/deepFM.py:284-319: concat -> [dense(units, act) ->
// batch_normalization]* -> dense(units=1); with GATHER the whole DeepFM inference graph: lookups, FM and first-order terms too,
// deepFM.py:217-223,321-335), COLUMN-SPLIT (round 6).  Same interface and arithmetic family as tower_bf3.hip's fp16 x 2 form (two fp16
// pieces per operand, the three products of weight >= 2^-11 on v_mfma_f32_16x16x32_f16, fp32 accumulate; 1e-5 against float64); what
// changes is who owns what inside the workgroup.
//
// Why.  tower_bf3_k gives every wave 16 batch rows and ALL output columns: the activations never leave registers, but each of the eight
// waves reads the whole weight stage from LDS -- eight times the bytes the workgroup needs (26 KB per wave and stage: 1 625 LDS cycles per
// stage against 1 248 matrix cycles per SIMD), and the weight stream, the barrier per stage and those reads add up (profiles/NOTES.md R6.5:
// 0.206 ms for three 400-wide layers at B = 65 536, matrix instructions alone 0.070).  Halving the reads by giving a wave two row tiles
// needs 312 registers.
//
// How.  One workgroup = 8 waves = 64 batch rows.  The layer's INPUT sits in LDS, pre-split into fp16 hi / lo pieces ([2][64][424] halves,
// 106 KB: one workgroup per CU); wave w owns 3-4 of the <= 26 OUTPUT column tiles for all four row tiles.  Per k-step of 32 a wave reads
//   * its own column tiles' weight fragments straight from the L2-resident image into registers (8 x 16 bytes per lane; nobody else in the
//     workgroup needs them: no LDS copy, no barrier per stage), one k-step ahead;
//   * the four row tiles' input fragments from LDS (8 x ds_read_b128: 8 KB per wave, 64 KB per k-step and workgroup instead of 416 KB);
// and issues 4 x 4 x 3 matrix instructions (D = W-piece (A: 16 output columns x 32 k) x X-piece (B: 32 k x 16 batch rows), transposed
// like tower_bf3_k: a lane holds 4 consecutive output columns of one batch row).  Between layers: barrier, bias / ReLU / affine, split,
// ds_write_b64 of the hi / lo pieces in place, barrier -- two barriers per LAYER instead of one per stage (6 instead of 78 per row tile).
// The units = 1 head: every wave's partial dot over its columns -> LDS -> the wave that owns the row adds the eight partials in wave order.
//
// GATHER (dir_deepfm_tower_cs_f16x2_f32): waves 0..3 look the 64 rows' F table rows up (lane (r, g): dims 4g..4g+3 of every slot, exactly
// tower_bf3_k<GATHER>'s lanes, so the FM and first-order terms are formed in the same order and are bit for bit gather_packed_rows_k's),
// split them and store them as layer 1's input.
//
// Row scaling (RS, the default; DIR_TOWER_RS=0 switches it off for A/B timing): every stored layer input row carries its own power-of-two
// scale (see tc_row_sft below), so that activations of any magnitude -- not only the caller's inputs and weights, which the host checks --
// survive the split into fp16 pieces.  Costs 5-6 % of the launch at B = 65 536 (profiles/NOTES.md R6.17).
//
// This file holds matrix instructions and is compiled without packed fp32 VALU instructions (build.py; isa_check.py).
#include <type_traits>

#include "common.hpp"

namespace dir {
namespace {

typedef float tc_f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 tc_f16x8 __attribute__((ext_vector_type(8)));
typedef unsigned int tc_u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int tc_u32x2 __attribute__((ext_vector_type(2)));
typedef __attribute__((address_space(3))) void* tc_lds_ptr;
typedef const __attribute__((address_space(1))) void* tc_glb_ptr;

constexpr int TC_ROWS = 64;             // batch rows per workgroup
constexpr int TC_NW = 8;                // waves
constexpr int TC_RT = TC_ROWS / 16;     // row tiles (all of them in every wave)
constexpr int TC_CT = 4;                // column tiles a wave owns at most (26 tiles over 8 waves: 4 4 3 3 3 3 3 3)
constexpr int TC_NT = 26;               // column tiles of 16: widths up to 416
constexpr int TC_MAXL = 4;
#ifndef TC_WIN_MAX_E
#define TC_WIN_MAX_E -3
#endif
constexpr int TC_WIN_MAX = TC_WIN_MAX_E;          // RS: a row whose exact max |element| is in [2^-3, 2^15) is stored unscaled (second piece's resolution 2^-24 <= 2^-21 max)
constexpr int TC_WIN_BOUND = 2;         // RS: ... whose BOUND is in [2^2, 2^15): the bound overestimates by 10-100 (<= 2^7: the same 2^-21)
#ifndef TC_PAD
#define TC_PAD 8
#endif
constexpr int TC_STRIDE = 16 * TC_NT + TC_PAD;       // halves per LDS row: 848 bytes = 212 words, 212 mod 32 = 20 -> eight consecutive rows' 16-byte pieces
                                                // fall into eight different bank quads (conflict-free ds_read_b128 per eight lanes)
constexpr int TC_PIECE = TC_ROWS * TC_STRIDE;   // halves per piece plane
constexpr int TC_ACT_BYTES = 2 * TC_PIECE * 2;  // 108 544
constexpr int TC_DUMP = TC_ACT_BYTES + TC_NW * TC_ROWS * 4;      // 256 bytes the input prefetch's LDS-DMA lands in (never read)
constexpr int TC_SLOTS = TC_DUMP + 256;                          // GATHER: every slot's vocabulary bound and table base (16 bytes each), staged once
constexpr int TC_ROWF = TC_SLOTS + 16 * TC_NT;                   // RS: [layer parity][batch row] s (an int): the stored layer input row holds x * 2^-s
constexpr int TC_ROWMAX = TC_ROWF + 2 * 4 * TC_ROWS;             // RS: [layer parity][batch row] the bit pattern of max |layer input| before scaling (LDS atomic max)
constexpr int TC_CST = TC_ROWMAX + 2 * 4 * TC_ROWS;              // RS: per layer (a, b): max |output| <= a * max |input| + b (the comment above tc_row_sft)
constexpr int TC_SMEM = TC_CST + 8 * TC_MAXL;                    // act + the head's partial dots [wave][row] + the dump + the slot table + the rows' scales

struct TowerCsParams {
    const float* X;
    int64_t x_ld, M;
    int Kd, L;
    int N[TC_MAXL];
    const unsigned char* img[TC_MAXL];
    const float* wnorm[TC_MAXL];     // the image's trailer: 64 floats whose maximum is max over output columns of sum_k |W[n][k]|
    const float* bias[TC_MAXL];
    const float* scale[TC_MAXL];
    const float* shift[TC_MAXL];
    int relu[TC_MAXL];
    const float* head_w;
    const float* head_b;
    const float* add0;
    const float* add1;
    float* out;
    int64_t out_ld;
    const float* const* tables;      // GATHER (see tower_bf3.hip: TowerParams)
    const int64_t* vocab;
    const int64_t* ids;
    int64_t ids_sb, ids_sf, row_ld;
    int F, lin_col, want_fm;
    const float* lin_bias;
#ifdef TC_STAMP
    unsigned long long* stamps;      // development: [wave][64] cycle stamps of workgroup 0's first tile
#endif
};

__device__ __forceinline__ unsigned int tc_pk_h(float a, float b) {     // v_cvt_pk_f16_f32 (round to nearest even), a in the low half
    typedef _Float16 h2_t __attribute__((ext_vector_type(2)));
    const h2_t v = {(_Float16)a, (_Float16)b};
    unsigned int w = __builtin_bit_cast(unsigned int, v);
    asm("" : "+v"(w));
    return w;
}
// (a, b) -> the packed hi pair and the packed lo pair (the residuals after rounding to fp16)
__device__ __forceinline__ void tc_split(float a, float b, unsigned int& hi, unsigned int& lo) {
    hi = tc_pk_h(a, b);
#if defined(__HIP_DEVICE_COMPILE__) && !defined(TC_NO_FMA_MIX)
    // the residual as one mixed-precision fma per element (din_pack.hip: dp_split2; exact, bit for bit the convert + subtract): the epilogue
    // between two layers is bound by VALU issue (two waves per SIMD, ~25 instructions per four values)
    float ra, rb;
    asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel_hi:[1,0,0]" : "=v"(ra) : "v"(hi), "v"(a));
    asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(rb) : "v"(hi), "v"(b));
    lo = tc_pk_h(ra, rb);
#else
    typedef _Float16 h2_t __attribute__((ext_vector_type(2)));
    const h2_t h = __builtin_bit_cast(h2_t, hi);
    lo = tc_pk_h(a - (float)h[0], b - (float)h[1]);
#endif
}
// Row scaling (RS): a stored row is multiplied by a power of two 2^-s -- exact -- and the layer's result by 2^s again in the epilogue (exact):
// whatever the activations' magnitudes, the fp16 pieces neither overflow (65 504) nor lose their second piece to fp16's subnormals.
// Layer 1's input: s puts the row's largest |element| into [2^13, 2^14).  A later layer's input is stored by eight waves that each see their own
// columns only; instead of a second pass behind a barrier, s comes from a BOUND the waves all know before they store:
//     max_n |out[n]| <= a * max_k |in[k]| + b,    a = max_n |scale[n]| * max_n sum_k |W[n][k]|,   b = max_n |scale[n]| * max_n |bias[n]| + max_n |shift[n]|
// (the weight norm from the image's trailer, pack time; the bias / affine maxima once per workgroup; max |in| is the ACTUAL maximum of the row,
// collected with LDS atomics while the previous layer stored it, so the bounds do not compound).  The bound puts the row's largest element
// at or below 2^14; it may overestimate by 2^17 before the second piece's resolution (2^-24 absolute) exceeds 2^-22 of the row's maximum --
// sum |w| against a dot product's sqrt(K) is a factor 10-100.
// tc_row_sft -> s from the bit pattern of the row's maximum or bound (0 / subnormal: 0).
// A maximum in [2^lo, 2^15) leaves the row as it is (s = 0): the workgroup's ordinary case, which skips the multiplications.
__device__ __forceinline__ int tc_row_sft(unsigned int maxbits, int lo) {
    const int e = (int)((maxbits >> 23) & 0xffu);
    const int sft = (e == 0 || (e - 127 >= lo && e - 127 <= 14)) ? 0 : e - 127 - 13;
    return sft < -100 ? -100 : (sft > 100 ? 100 : sft);
}
__device__ __forceinline__ float tc_pow2(int s) { return __builtin_bit_cast(float, (unsigned int)(127 + s) << 23); }
__device__ __forceinline__ int tc_byte(unsigned int packed, int i) { return (int)(signed char)(packed >> (8 * i)); }
// max(m, |v[0..3]|): two v_max3_f32 with |.| source modifiers (m >= 0; as a bit pattern it orders like the float)
__device__ __forceinline__ float tc_absmax(const tc_f32x4 v, float m) {
    m = fmaxf(fmaxf(m, fabsf(v[0])), fabsf(v[1]));
    return fmaxf(fmaxf(m, fabsf(v[2])), fabsf(v[3]));
}
// four consecutive values of one batch row -> the 8 bytes of each piece plane
__device__ __forceinline__ void tc_store4(_Float16* act, int row, int col, const tc_f32x4 v) {
    unsigned int h0, l0, h1, l1;
    tc_split(v[0], v[1], h0, l0);
    tc_split(v[2], v[3], h1, l1);
    *reinterpret_cast<tc_u32x2*>(act + row * TC_STRIDE + col) = (tc_u32x2){h0, h1};
    *reinterpret_cast<tc_u32x2*>(act + TC_PIECE + row * TC_STRIDE + col) = (tc_u32x2){l0, l1};
}

// W [N, K] fp32 (row stride w_ld) -> image [k-step][column tile][piece][lane][8 halves]: element e of lane l of tile ct in k-step ks = piece of
// W[n = 16*ct + (l & 15)][k = 32*ks + 8*(l >> 4) + e] (the A operand of v_mfma_f32_16x16x32_f16 in its natural k order); zero where n >= N or k >= K.
// Workgroups [npack, npack + TC_NORM_WGS) do not pack: they form the image's TRAILER, TC_NORM_SLOTS floats whose maximum is
// max_n sum_k |W[n][k]| (row scaling's weight norm): workgroup j's rows j, j + TC_NORM_WGS, .. four at a time (a wave per row), its maximum
// into slot j -- no atomics, nothing to zero first, and no second launch (a HIP graph of a small-batch forward re-packs on every replay:
// as a memset and a kernel of their own the norms cost it 11 us per layer).
constexpr int TC_NORM_SLOTS = 64;
constexpr int TC_NORM_WGS = 64;                // one slot each: a wave sums at most two rows of a 416-wide layer
__global__ __launch_bounds__(256) void tower_cs_pack_k(const float* __restrict__ W, int64_t w_ld, int K, int N, int nks, int nct,
                                                       unsigned int* __restrict__ img, int npack, float* __restrict__ trailer) {
    if ((int)blockIdx.x >= npack) {
        __shared__ float wmax[4];
        const int j = blockIdx.x - npack, wv = threadIdx.x >> 6, ln = threadIdx.x & 63;
        float m = 0.f;
        for (int n = 4 * j + wv; n < N; n += 4 * TC_NORM_WGS) {
            const float* w = W + (int64_t)n * w_ld;
            float a = 0.f;
            for (int k = ln; k < K; k += 64) a += fabsf(w[k]);
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) a += __shfl_xor(a, o, 64);
            m = fmaxf(m, a * 1.0001f);                             // (the sum's own rounding: K 2^-24)
        }
        if (ln == 0) wmax[wv] = m;
        __syncthreads();
        if (threadIdx.x == 0) trailer[j] = fmaxf(fmaxf(wmax[0], wmax[1]), fmaxf(wmax[2], wmax[3]));
        return;
    }
    const int64_t total = (int64_t)nks * nct * 64 * 4;            // one thread per pair of e
    for (int64_t q_ = (int64_t)blockIdx.x * 256 + threadIdx.x; q_ < total; q_ += (int64_t)npack * 256) {
        int64_t q = q_;
        const int ep = (int)(q & 3); q >>= 2;
        const int l = (int)(q & 63); q >>= 6;
        const int ct = (int)(q % nct);
        const int ks = (int)(q / nct);
        const int n = 16 * ct + (l & 15);
        const int k = 32 * ks + 8 * (l >> 4) + 2 * ep;
        const float v0 = (n < N && k < K) ? W[(int64_t)n * w_ld + k] : 0.f;
        const float v1 = (n < N && k + 1 < K) ? W[(int64_t)n * w_ld + k + 1] : 0.f;
        unsigned int hi, lo;
        tc_split(v0, v1, hi, lo);
        const int64_t base = ((int64_t)ks * nct + ct) * 512 + l * 4 + ep;       // dwords: a (k-step, tile) block is 2 pieces x 256 dwords
        img[base] = hi;
        img[base + 256] = lo;
    }
}

template <bool GATHER, bool RS>
__global__ __launch_bounds__(64 * TC_NW, 2) void tower_cs_k(const TowerCsParams p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char tc_smem[];
    _Float16* const act = reinterpret_cast<_Float16*>(tc_smem);                 // [2 pieces][64 rows][TC_STRIDE]
    float* const part = reinterpret_cast<float*>(tc_smem + TC_ACT_BYTES);      // [8 waves][64 rows]
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r16 = lane & 15;
    const int g = lane >> 4;
    const int64_t ntiles = (p.M + TC_ROWS - 1) / TC_ROWS;
    uint64_t* const slot_info = reinterpret_cast<uint64_t*>(tc_smem + TC_SLOTS);
    int* const rowsft = reinterpret_cast<int*>(tc_smem + TC_ROWF);                   // RS: [layer parity][row] s of the stored layer input's row (it holds x 2^-s)
    unsigned int* const rowmax = reinterpret_cast<unsigned int*>(tc_smem + TC_ROWMAX);   // RS: [layer parity][row] max |unscaled layer input|
    float* const cst = reinterpret_cast<float*>(tc_smem + TC_CST);                   // RS: [layer] (a, b) of the output bound
    if constexpr (RS) {
        // the layers' bound constants, once per workgroup, by a wave the GATHER input phase leaves idle; read behind the first input barrier.
        // (Tried: the reductions as LDS atomics instead of shuffles, and the whole block moved behind the wave's first k loop -- both slower
        // by 2-4 % of the launch, profiles/NOTES.md R6.17.)
#ifndef TC_ABL_RS_PRO
        if (wave == TC_NW - 1) {
            // every layer's vectors asked for first (unconditional loads from clamped addresses), then the reductions: layer by layer the
            // cold round trips stood one behind the other (a small-batch forward is one tile: 7 us of its 70)
            constexpr int NV = (16 * TC_NT + 63) / 64;
            float bv[TC_MAXL - 1][NV], sv[TC_MAXL - 1][NV], hv[TC_MAXL - 1][NV], wnv[TC_MAXL - 1];
#pragma unroll
            for (int l = 0; l < TC_MAXL - 1; ++l) {
                const bool on = l + 1 < p.L;
                const int N = on ? p.N[l] : 1;
                const float* dummy = reinterpret_cast<const float*>(p.img[0]);
                const float* bias = on && p.bias[l] ? p.bias[l] : dummy;
                const float* sc = on && p.scale[l] ? p.scale[l] : dummy;
                const float* sh = on && p.shift[l] ? p.shift[l] : dummy;
                wnv[l] = (on ? p.wnorm[l] : p.wnorm[0])[lane];           // TC_NORM_SLOTS = 64 partial maxima
#pragma unroll
                for (int i = 0; i < NV; ++i) {
                    const int j = lane + 64 * i;
                    const int jj = j < N ? j : 0;
                    bv[l][i] = bias[jj];
                    sv[l][i] = sc[jj];
                    hv[l][i] = sh[jj];
                }
            }
#pragma unroll
            for (int l = 0; l < TC_MAXL - 1; ++l) {
                const bool on = l + 1 < p.L;
                const int N = on ? p.N[l] : 1;
                float bm = 0.f, sm = 0.f, hm = 0.f, wn = wnv[l];
#pragma unroll
                for (int i = 0; i < NV; ++i) {
                    if (lane + 64 * i < N) {
                        bm = fmaxf(bm, fabsf(bv[l][i]));
                        sm = fmaxf(sm, fabsf(sv[l][i]));
                        hm = fmaxf(hm, fabsf(hv[l][i]));
                    }
                }
#pragma unroll
                for (int o = 32; o > 0; o >>= 1) {
                    bm = fmaxf(bm, __shfl_xor(bm, o, 64));
                    sm = fmaxf(sm, __shfl_xor(sm, o, 64));
                    hm = fmaxf(hm, __shfl_xor(hm, o, 64));
                    wn = fmaxf(wn, __shfl_xor(wn, o, 64));
                }
                if (!on || !p.bias[l]) bm = 0.f;
                if (!on || !p.scale[l]) { sm = 1.f; hm = 0.f; }
                if (on && lane == 0) {
                    cst[2 * l] = sm * wn;
                    cst[2 * l + 1] = sm * bm + hm;
                }
            }
        }
#endif
    }
    if constexpr (GATHER) {          // (as global loads of p.vocab[ct] / p.tables[ct] each was a waited-for round trip in front of the slot's row read)
        if (tid < p.F) {
            slot_info[2 * tid] = p.vocab ? (uint64_t)p.vocab[tid] : (uint64_t)1 << 63;
            slot_info[2 * tid + 1] = reinterpret_cast<uint64_t>(p.tables[tid]);
        }
        __syncthreads();
    }

#ifdef TC_STAMP
    int stamp_n = 0;
#define TC_MARK() do { if (blockIdx.x == 0 && lane == 0 && stamp_n < 64) p.stamps[wave * 64 + stamp_n++] = __builtin_readcyclecounter(); } while (0)
#else
#define TC_MARK() do {} while (0)
#endif
    for (int64_t t = blockIdx.x; t < ntiles; t += gridDim.x) {
        const int64_t row0 = t * TC_ROWS;
        TC_MARK();                                     // 0: tile start
        // ---- layer 1's input -> LDS (split).  Columns [Kd, 32 * ceil(Kd / 32)) are zero-filled: the matrix instructions read whole k-steps.
        float fm_r = 0.f, lin_r = 0.f;                 // GATHER, waves 0..3: the FM and first-order terms of row 16 * wave + r16 (valid in every lane)
        const int kd32 = (p.Kd + 31) & ~31;
        __syncthreads();                               // the previous tile's last reads of `act` and `part` are done
        if constexpr (GATHER) {
            if (wave < 4) {
                const int lr = 16 * wave + r16;
                const int64_t r = row0 + lr;
                const int64_t rr = r < p.M ? r : p.M - 1;
                const int64_t* idp = p.ids + rr * p.ids_sb;
                tc_f32x4 sum = {0.f, 0.f, 0.f, 0.f}, sq = sum;
                float lin = 0.f;
                // every slot's id first, then every row read (clamped, unconditional), then the sums and stores: one loop would serialise
                // 2 x 26 dependent round trips (cin_bf3.hip: the x0 staging through inverse positions)
                int64_t idv[TC_NT];
#pragma unroll
                for (int ct = 0; ct < TC_NT; ++ct) idv[ct] = ct < p.F ? idp[(int64_t)ct * p.ids_sf] : -1;
                tc_f32x4 vv[TC_NT];
                float lwv[TC_NT];
#pragma unroll
                for (int ct = 0; ct < TC_NT; ++ct) {
                    const int cc = ct < p.F ? ct : 0;
                    const uint64_t bound = slot_info[2 * cc];
                    const bool ok = ct < p.F && (uint64_t)idv[ct] < bound;
                    const float* tr = reinterpret_cast<const float*>(slot_info[2 * cc + 1]) + (ok ? idv[ct] : 0) * p.row_ld;
                    vv[ct] = *reinterpret_cast<const tc_f32x4*>(tr + 4 * g);
                    lwv[ct] = p.lin_col >= 0 ? tr[g == 0 ? p.lin_col : 0] : 0.f;
                    if (g != 0) lwv[ct] = 0.f;
                    if (!ok) {
                        vv[ct] = (tc_f32x4){0.f, 0.f, 0.f, 0.f};
                        lwv[ct] = 0.f;
                    }
                }
                float mf = 0.f;
#pragma unroll
                for (int ct = 0; ct < TC_NT; ++ct) {
                    const tc_f32x4 v = vv[ct];
                    sum += v;                          // f-ascending fp32 sums, as gather_packed_rows_k
                    sq += v * v;
                    lin = lin + lwv[ct];
                    if constexpr (RS) mf = tc_absmax(v, mf);
                    if (16 * ct < kd32) tc_store4(act, lr, 16 * ct + 4 * g, v);      // as the rows arrive
                }
#ifndef TC_ABL_RS_IN
                if constexpr (RS) {                    // the row's largest |element| over its four lanes -> its power-of-two scale; a row outside
                    mf = fmaxf(mf, __shfl_xor(mf, 16, 64));                        // the window is stored again, scaled (the rare case)
                    mf = fmaxf(mf, __shfl_xor(mf, 32, 64));
                    const unsigned int mb = __builtin_bit_cast(unsigned int, mf);
                    const int sft = tc_row_sft(mb, TC_WIN_MAX);
                    const float down = tc_pow2(-sft);
                    if (g == 0) {
                        rowsft[lr] = sft;
                        rowmax[lr] = mb;
                    }
                    if (sft != 0) {
#pragma unroll
                        for (int ct = 0; ct < TC_NT; ++ct)
                            if (16 * ct < kd32) tc_store4(act, lr, 16 * ct + 4 * g, vv[ct] * down);
                    }
                }
#else
                if (g == 0) { rowsft[lr] = 0; rowmax[lr] = 0x3f800000u; }
#endif
                if (p.want_fm) {                       // 0.5 * sum_k (sum^2 - sq), k ascending through the row's four lanes: fm_tail<4>'s chain
                    const tc_f32x4 d = sum * sum - sq;
                    float acc_fm = 0.f;
#pragma unroll
                    for (int cc = 0; cc < 4; ++cc) {
                        const float carry = __shfl(acc_fm, r16 + 16 * (cc > 0 ? cc - 1 : 0), 64);
                        if (g == cc) acc_fm = ((((cc == 0 ? 0.f : carry) + d[0]) + d[1]) + d[2]) + d[3];
                    }
                    fm_r = __shfl(0.5f * acc_fm, r16 + 48, 64);
                }
                lin_r = __shfl(lin, r16, 64) + (p.lin_bias ? p.lin_bias[0] : 0.f);
            }
        } else {
            // thread (row = tid / 8, seg = tid % 8): 16-byte pieces seg, seg + 8, ... of the row
            const int lr = tid >> 3, seg = tid & 7;
            const int64_t r = row0 + lr;
            const float* xr = p.X + (r < p.M ? r : p.M - 1) * p.x_ld;
            tc_f32x4 vv[13];
#pragma unroll
            for (int i = 0; i < 13; ++i) {
                const int k = 4 * (seg + 8 * i);
#ifdef TC_ABL_IN           // timing ablation: no input read
                vv[i] = (tc_f32x4){0.25f, 0.5f, 0.125f, 1.f};
#else
                vv[i] = *reinterpret_cast<const tc_f32x4*>(xr + (k < p.Kd ? k : 0));
#endif
                if (k >= p.Kd) vv[i] = (tc_f32x4){0.f, 0.f, 0.f, 0.f};
            }
            float mf = 0.f;
#pragma unroll
            for (int i = 0; i < 13; ++i) {             // stored as they arrive
                const int k = 4 * (seg + 8 * i);
                if constexpr (RS) mf = tc_absmax(vv[i], mf);
                if (k < kd32) tc_store4(act, lr, k, vv[i]);
            }
#ifndef TC_ABL_RS_IN
            if constexpr (RS) {                        // eight consecutive lanes hold a row; a row outside the window is stored again, scaled
                // (DPP: lanes 1 and 2 apart inside the quad, then the other quad of the eight -- a shuffle is an LDS round trip each)
                int mi = __builtin_bit_cast(int, mf);
                mf = fmaxf(mf, __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(mi, 0xB1, 0xf, 0xf, true)));     // quad_perm [1,0,3,2]
                mi = __builtin_bit_cast(int, mf);
                mf = fmaxf(mf, __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(mi, 0x4E, 0xf, 0xf, true)));     // quad_perm [2,3,0,1]
                mi = __builtin_bit_cast(int, mf);
                mf = fmaxf(mf, __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(mi, 0x141, 0xf, 0xf, true)));    // row_half_mirror
                const unsigned int mb = __builtin_bit_cast(unsigned int, mf);
                const int sft = tc_row_sft(mb, TC_WIN_MAX);
                const float down = tc_pow2(-sft);
                if (seg == 0) {
                    rowsft[lr] = sft;
                    rowmax[lr] = mb;
                }
                if (sft != 0) {
#pragma unroll
                    for (int i = 0; i < 13; ++i) {
                        const int k = 4 * (seg + 8 * i);
                        if (k < kd32) tc_store4(act, lr, k, vv[i] * down);
                    }
                }
            }
#else
            if (seg == 0) { rowsft[lr] = 0; rowmax[lr] = 0x3f800000u; }
#endif
        }
        TC_MARK();                                     // 1: input stored
        __syncthreads();
        TC_MARK();                                     // 2: after the input barrier
        // The NEXT tile's input rows are asked for now (plain form): one word per 128-byte line, by LDS-DMA into a dump area nobody reads, from
        // waves 2 and 3 only.  All 256 workgroups reach their input phase together, so a tile's 106 KB used to arrive as a 27 MB burst with
        // the matrix pipe idle (15 000 cycles per tile); asked for here it streams in under the first layer and the input phase finds it in
        // L2 / the Infinity Cache.  Why LDS-DMA: no registers.  Why two waves: vmcnt counts in order, so the asking wave's first k-step waits
        // for HBM -- waves 2 and 3 own three column tiles and run ahead of their SIMD partners (they idle ~9 000 cycles at the layer barrier).
        if constexpr (!GATHER) {
            const int64_t tn = t + gridDim.x;
            if (tn < ntiles && (wave == 2 || wave == 3)) {              // wave-uniform
                const int nline = (p.Kd + 31) >> 5;                       // 128-byte lines per row (<= 13)
#pragma unroll
                for (int j = 0; j < 7; ++j) {
                    const int line = lane + 64 * j;                       // of this wave's 32 rows
                    const int lr = line / 13, sg = line - 13 * lr;
                    const int64_t rn = tn * TC_ROWS + 32 * (wave - 2) + (lr < 32 ? lr : 31);
                    const float* src = p.X + (rn < p.M ? rn : p.M - 1) * p.x_ld + (sg < nline ? 32 * sg : 0);
                    __builtin_amdgcn_global_load_lds((tc_glb_ptr)src, (tc_lds_ptr)(tc_smem + TC_DUMP), 4, 0, 0);
                }
            }
        }

        for (int l = 0; l < p.L; ++l) {
            const int K = l ? p.N[l - 1] : p.Kd;
            const int N = p.N[l];
            const int nks = (K + 31) >> 5, nct = (N + 15) >> 4;
            const bool last = l + 1 == p.L;
            // the layer's column tiles, padded to whole k-steps of the NEXT layer (its input columns [N, 32 * ceil(N / 32)) must read as zeros:
            // a tile behind the last real one has no matrix work and stores zeros), dealt to the waves as evenly as whole tiles go
            const int nctp = last ? nct : 2 * ((N + 31) >> 5);
            const int base = nctp >> 3, rem = nctp & 7;
            const int c0 = wave * base + (wave < rem ? wave : rem);
            const int cnt = base + (wave < rem ? 1 : 0);                 // tiles this wave stores (<= TC_CT)
            const int creal = nct - c0 < cnt ? (nct - c0 > 0 ? nct - c0 : 0) : cnt;      // ... of which these have matrix work
            if constexpr (RS) {     // the maxima this layer's epilogue collects: last read one layer ago (a barrier back), first written a barrier ahead
                if (tid < TC_ROWS) rowmax[((l + 1) & 1) * TC_ROWS + tid] = 0u;
            }

            tc_f32x4 acc[TC_CT][TC_RT];
#pragma unroll
            for (int c = 0; c < TC_CT; ++c)
#pragma unroll
                for (int rt = 0; rt < TC_RT; ++rt) acc[c][rt] = (tc_f32x4){0.f, 0.f, 0.f, 0.f};

            // this lane's pieces: weights at img + ((ks * nct + tile) * 2 + piece) * 1024 + lane * 16 (a tile the wave does not have: its last real one
            // again -- the loads are unconditional so that the waits can be counted); input at act[piece][16 rt + r16][32 ks + 8 g ..]
            const unsigned char* wl[TC_CT];
#pragma unroll
            for (int c = 0; c < TC_CT; ++c) {
                int tile = c0 + (c < creal ? c : (creal > 0 ? creal - 1 : 0));
                tile = tile < nct ? tile : nct - 1;
                wl[c] = p.img[l] + (int64_t)tile * 2048 + lane * 16;
            }
            const int64_t wstep = (int64_t)nct * 2048;
            const _Float16* xl = act + r16 * TC_STRIDE + 8 * g;

            // The k loop once per tile count (a compile-time NC: straight-line blocks of 12 NC matrix instructions; with the count as a run-time
            // guard inside the loop the compiler emitted a branch around every four of them and the matrix pipe ran at half rate).
            // (TC_ABL_*: timing ablations, WRONG results -- development builds through tools/ab_variant.sh only)
            // RS: the exponents of this layer's input rows (s) and of the rows it stores (s'), one byte per row tile, and whether any differs
            // from 0 -- LDS reads, the bound, a vote: a dependent chain, run behind the first k-step's loads (two registers instead of twelve
            // across the k loop)
            unsigned int pin = 0u, pout = 0u;
            bool scaled = false;                       // wave-uniform
            auto rs_chain = [&]() {
#ifndef TC_ABL_RS_CHAIN
                if constexpr (RS) {
                    const int cur = (l & 1) * TC_ROWS;
#pragma unroll
                    for (int rt = 0; rt < TC_RT; ++rt) pin |= ((unsigned int)rowsft[cur + 16 * rt + r16] & 0xffu) << (8 * rt);
                    if (!last) {
                        const float ca = cst[2 * l], cb = cst[2 * l + 1];
#pragma unroll
                        for (int rt = 0; rt < TC_RT; ++rt) {
                            const float bound = ca * __builtin_bit_cast(float, rowmax[cur + 16 * rt + r16]) + cb;
                            pout |= ((unsigned int)tc_row_sft(__builtin_bit_cast(unsigned int, bound), TC_WIN_BOUND) & 0xffu) << (8 * rt);
                        }
                    }
                    scaled = __any((pin | pout) != 0u);
                }
#endif
            };
            auto mainloop = [&](auto nc_tag) {
                constexpr int NC = decltype(nc_tag)::value;
                auto load_w = [&](int ks, tc_u32x4 (&w)[TC_CT][2]) {
#pragma unroll
                    for (int c = 0; c < NC; ++c) {
                        w[c][0] = *reinterpret_cast<const tc_u32x4*>(wl[c] + ks * wstep);
                        w[c][1] = *reinterpret_cast<const tc_u32x4*>(wl[c] + ks * wstep + 1024);
                    }
                };
                auto load_x = [&](int ks, tc_u32x4 (&x)[TC_RT][2]) {
#pragma unroll
                    for (int rt = 0; rt < TC_RT; ++rt) {
                        x[rt][0] = *reinterpret_cast<const tc_u32x4*>(xl + rt * 16 * TC_STRIDE + 32 * ks);
                        x[rt][1] = *reinterpret_cast<const tc_u32x4*>(xl + TC_PIECE + rt * 16 * TC_STRIDE + 32 * ks);
                    }
                };
                auto compute = [&](const tc_u32x4 (&w)[TC_CT][2], const tc_u32x4 (&x)[TC_RT][2]) {
                    // the three products, smallest first; one product over all of the wave's tiles before the next (4 NC independent accumulators)
#pragma unroll
                    for (int pr = 0; pr < 3; ++pr)
#pragma unroll
                        for (int c = 0; c < NC; ++c) {
                            const tc_f16x8 wv = __builtin_bit_cast(tc_f16x8, w[c][pr == 0 ? 1 : 0]);
#pragma unroll
                            for (int rt = 0; rt < TC_RT; ++rt) {
                                const tc_f16x8 xv = __builtin_bit_cast(tc_f16x8, x[rt][pr == 1 ? 1 : 0]);
                                acc[c][rt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wv, xv, acc[c][rt], 0, 0, 0);
                            }
                        }
                };
                tc_u32x4 wA[TC_CT][2], wB[TC_CT][2], xA[TC_RT][2], xB[TC_RT][2];
                load_w(0, wA);
                load_x(0, xA);
                rs_chain();
#ifdef TC_ABL_W
#define TC_LOAD_W(k, w)
                load_w(0, wB);
#else
#define TC_LOAD_W(k, w) load_w(k, w)
#endif
#ifdef TC_ABL_X
#define TC_LOAD_X(k, x)
                load_x(0, xB);
#else
#define TC_LOAD_X(k, x) load_x(k, x)
#endif
#ifdef TC_ABL_MFMA
#define TC_COMPUTE(w, x) asm volatile("" :: "v"(w[0][0]), "v"(w[NC - 1][0]), "v"(w[0][1]), "v"(w[NC - 1][1]), "v"(x[0][0]), "v"(x[1][0]), "v"(x[2][0]), "v"(x[3][0]), "v"(x[0][1]), "v"(x[1][1]), "v"(x[2][1]), "v"(x[3][1]))
#else
#define TC_COMPUTE(w, x) compute(w, x)
#endif
                for (int ks = 0; ks < nks; ks += 2) {
                    const int k1 = ks + 1 < nks ? ks + 1 : nks - 1;   // (clamped: the loads stay unconditional)
                    // sched_barrier: the next k-step's loads are ISSUED before this k-step's matrix instructions.  Left alone the scheduler sinks
                    // each load to just behind the last use of the register it reuses (one register set instead of two) and the next block opens
                    // with vmcnt(0): the weights' L2 latency, exposed once per k-step
                    TC_LOAD_W(k1, wB);
                    TC_LOAD_X(k1, xB);
                    __builtin_amdgcn_sched_barrier(0);
                    TC_COMPUTE(wA, xA);
                    __builtin_amdgcn_sched_barrier(0);
                    const int k2 = ks + 2 < nks ? ks + 2 : nks - 1;
                    TC_LOAD_W(k2, wA);
                    TC_LOAD_X(k2, xA);
                    __builtin_amdgcn_sched_barrier(0);
                    if (ks + 1 < nks) TC_COMPUTE(wB, xB);
                    __builtin_amdgcn_sched_barrier(0);
                }
#undef TC_LOAD_W
#undef TC_LOAD_X
#undef TC_COMPUTE
            };
            switch (creal) {                           // wave-uniform
                case 4: mainloop(std::integral_constant<int, 4>{}); break;
                case 3: mainloop(std::integral_constant<int, 3>{}); break;
                case 2: mainloop(std::integral_constant<int, 2>{}); break;
                case 1: mainloop(std::integral_constant<int, 1>{}); break;
                default: rs_chain(); break;
            }

            TC_MARK();                                 // 3 + 4 l: main loop done
            // the epilogue's per-column vectors, requested BEFORE the barrier (unconditional loads from clamped addresses; a vector that does
            // not exist reads the weight image and is zeroed): their latency passes while the workgroup's slowest wave finishes its k loop
            // (issued inside the epilogue they cost it ~2 000 cycles per layer: profiles/NOTES.md R6.9)
            const float* bias = p.bias[l];
            const float* sc = p.scale[l];
            const float* sh = p.shift[l];
            const float* hw = last ? p.head_w : nullptr;
            const float* dummy = reinterpret_cast<const float*>(p.img[l]);
            tc_f32x4 b4v[TC_CT], s4v[TC_CT], h4v[TC_CT], w4v[TC_CT];
#pragma unroll
            for (int c = 0; c < TC_CT; ++c) {
                const int col = 16 * (c0 + c) + 4 * g;
                const int cc = (c < cnt && col < N) ? col : 0;
                b4v[c] = *reinterpret_cast<const tc_f32x4*>((bias ? bias : dummy) + cc);
                s4v[c] = *reinterpret_cast<const tc_f32x4*>((sc ? sc : dummy) + cc);
                h4v[c] = *reinterpret_cast<const tc_f32x4*>((sh ? sh : dummy) + cc);
                w4v[c] = *reinterpret_cast<const tc_f32x4*>((hw ? hw : dummy) + cc);
            }
            __syncthreads();                           // everybody has read this layer's input: the output may overwrite it
            TC_MARK();                                 // 4 + 4 l: after the read barrier

            // ---- epilogue: bias, activation, inference batch-norm affine; the result is the next layer's input (split, in place),
            //      the caller's output, or the head's dot product
            const int relu = p.relu[l];
            float hp[TC_RT] = {0.f, 0.f, 0.f, 0.f};    // head: this lane's share of the logit of row 16 rt + r16
            float fin[TC_RT], down[TC_RT];             // RS: 2^s of this layer's input rows (what the accumulators are multiplied by first), 2^-s' of the rows it stores
            float mxf[TC_RT] = {0.f, 0.f, 0.f, 0.f};   // RS: max |output| of the lane's values per row
#pragma unroll
            for (int rt = 0; rt < TC_RT; ++rt) {
                fin[rt] = RS ? tc_pow2(tc_byte(pin, rt)) : 1.f;
                down[rt] = RS ? tc_pow2(-tc_byte(pout, rt)) : 1.f;
            }
            // The body once per combination of flags: as run-time tests inside the loops every group of four values crossed ~10 scalar branches
            // and the epilogue was bound by instruction issue (5 300 cycles per layer and tile, two waves per SIMD).  The two shapes the
            // reference's towers have between layers (bias + ReLU, with or without the batch-norm affine) are compile-time; everything else,
            // and the last layer, takes the flags as run-time values.  SCALED = false: every row of the wave has 2^s = 1 on both sides.
            auto epi = [&](auto f_bias, auto f_relu, auto f_sc, auto f_last, auto f_head, auto f_scaled) {
#pragma unroll
                for (int c = 0; c < TC_CT; ++c) {
                    if (c < cnt) {                         // wave-uniform
                        const int col = 16 * (c0 + c) + 4 * g;       // N % 4 == 0: the lane's four columns are inside or outside together
                        const bool in = col < N;
                        const tc_f32x4 b4 = b4v[c], s4 = s4v[c], h4 = h4v[c];
                        const tc_f32x4 w4 = (in && hw) ? w4v[c] : (tc_f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
                        for (int rt = 0; rt < TC_RT; ++rt) {
                            tc_f32x4 v = acc[c][rt];
                            if (RS && f_scaled) {          // (acc * 2^s is exact: the fused form rounds once, like the sum it replaces)
                                if (f_bias) {
#pragma unroll
                                    for (int e = 0; e < 4; ++e) v[e] = __builtin_fmaf(v[e], fin[rt], b4[e]);
                                } else {
                                    v = v * fin[rt];
                                }
                            } else if (f_bias) {
                                v += b4;
                            }
                            if (f_relu) {
#pragma unroll
                                for (int e = 0; e < 4; ++e) v[e] = v[e] > 0.f ? v[e] : 0.f;
                            }
                            if (f_sc) {                    // multiply then add, unfused (dense.hip's affine epilogue)
#pragma unroll
                                for (int e = 0; e < 4; ++e) v[e] = v[e] * s4[e] + h4[e];
                            }
                            if (!in) v = (tc_f32x4){0.f, 0.f, 0.f, 0.f};
                            if (!f_last) {
#ifndef TC_ABL_RS_MAX
                                if constexpr (RS) mxf[rt] = tc_absmax(v, mxf[rt]);
#endif
#ifdef TC_ABL_EPI          // timing ablation: the epilogue's split + LDS stores only where the compiler cannot drop the values
                                if (v[0] == 123.456f) tc_store4(act, 16 * rt + r16, col, v);
#else
                                tc_store4(act, 16 * rt + r16, col, (RS && f_scaled) ? v * down[rt] : v);
#endif
                            } else if (f_head) {
                                hp[rt] += v[0] * w4[0];
                                hp[rt] += v[1] * w4[1];
                                hp[rt] += v[2] * w4[2];
                                hp[rt] += v[3] * w4[3];
                            } else if (in) {
                                const int64_t r = row0 + 16 * rt + r16;
                                if (r < p.M) *reinterpret_cast<tc_f32x4*>(p.out + r * p.out_ld + col) = v;
                            }
                        }
                    }
                }
            };
            {
                constexpr std::true_type T{};
                constexpr std::false_type F{};
                const bool fb = bias != nullptr, fr = relu != 0, fs = sc != nullptr;
                const bool fh = hw != nullptr;
                if (!last && fb && fr && !fs && !scaled) epi(T, T, F, F, F, F);
                else if (!last && fb && fr && fs && !scaled) epi(T, T, T, F, F, F);
                else if (!last && fb && fr && !fs) epi(T, T, F, F, F, T);
                else if (!last && fb && fr) epi(T, T, T, F, F, T);
                else if (!last) epi(fb, fr, fs, F, F, T);
                else if (fb && fr && !fs && fh && !scaled) epi(T, T, F, T, T, F);       // the last hidden layer under the units = 1 head
                else if (fb && fr && !fs && fh) epi(T, T, F, T, T, T);
                else if (fb && fr && !fs && !scaled) epi(T, T, F, T, F, F);            // ... written out
                else if (fb && fr && !fs) epi(T, T, F, T, F, T);
                else epi(fb, fr, fs, T, fh, T);
            }
            if (last && p.head_w) {
#pragma unroll
                for (int rt = 0; rt < TC_RT; ++rt) {
                    float s = hp[rt];
                    s += __shfl_xor(s, 16, 64);        // the wave's other columns live in the other three lane groups
                    s += __shfl_xor(s, 32, 64);
                    if (g == 0) part[wave * TC_ROWS + 16 * rt + r16] = s;
                }
            }
            if constexpr (RS) {
                if (!last) {         // (workgroup-uniform) the stored rows' actual maxima and scales, for the next layer's epilogue (a barrier ahead)
                    const int nxt = ((l + 1) & 1) * TC_ROWS;
#pragma unroll
                    for (int rt = 0; rt < TC_RT; ++rt) {
                        // (from every lane: four lanes per address cost the LDS a few cycles; two shuffles per row cost the wave their latency)
#ifndef TC_ABL_RS_ATOM
                        if (cnt > 0) atomicMax(&rowmax[nxt + 16 * rt + r16], __builtin_bit_cast(unsigned int, mxf[rt]));
#endif
                        if (wave == 0 && g == 0) rowsft[nxt + 16 * rt + r16] = tc_byte(pout, rt);
                    }
                }
            }
            TC_MARK();                                 // 5 + 4 l: epilogue done
            __syncthreads();
            TC_MARK();                                 // 6 + 4 l: after the write barrier
            if (last && p.head_w && wave < 4 && g == 0) {
                const int lr = 16 * wave + r16;
                const int64_t r = row0 + lr;
                if (r < p.M) {
                    float o = part[lr];
#pragma unroll
                    for (int w = 1; w < TC_NW; ++w) o += part[w * TC_ROWS + lr];
                    o += p.head_b[0];
                    if constexpr (GATHER) {
                        if (p.want_fm) o += fm_r;      // the order of ops.tower(..., adds=(fm, lin))
                        if (p.lin_col >= 0) o += lin_r;
                    }
                    if (p.add0) o += p.add0[r];
                    if (p.add1) o += p.add1[r];
                    p.out[r * p.out_ld] = o;
                }
            }
        }
    }
}

inline int64_t tc_image_body(int K, int N) { return (int64_t)((K + 31) / 32) * ((N + 15) / 16) * 2048; }

int tower_cs_fill(const char* name, TowerCsParams& p, int Kd, int L, const int* N, const void* const* images, const float* const* bias,
                  const float* const* post_scale, const float* const* post_shift, const int* act, const float* head_w, const float* head_b,
                  const float* add0, const float* add1, float* out, int64_t out_ld) {
    DIR_CHECK_ARG(Kd > 0 && L >= 1 && L <= TC_MAXL && N && images && act, "%s: bad shape (Kd=%d L=%d)", name, Kd, L);
    if (Kd > 16 * TC_NT || (Kd & 3)) return fail(DIR_E_UNSUPPORTED, "%s: Kd=%d (a multiple of 4, <= %d)", name, Kd, 16 * TC_NT);
    DIR_CHECK_ARG((head_w == nullptr) == (head_b == nullptr), "%s: head_w and head_b come together", name);
    DIR_CHECK_ARG(head_w || (!add0 && !add1), "%s: add0 / add1 are addends of the head's logit", name);
    p.Kd = Kd; p.L = L;
    for (int l = 0; l < TC_MAXL; ++l) {
        p.N[l] = 0; p.img[l] = nullptr; p.wnorm[l] = nullptr; p.bias[l] = p.scale[l] = p.shift[l] = nullptr; p.relu[l] = 0;
    }
    for (int l = 0; l < L; ++l) {
        if (N[l] <= 0 || N[l] > 16 * TC_NT || (N[l] & 3)) return fail(DIR_E_UNSUPPORTED, "%s: layer %d width %d (a multiple of 4, <= %d)", name, l, N[l], 16 * TC_NT);
        DIR_CHECK_ARG(act[l] == DIR_ACT_NONE || act[l] == DIR_ACT_RELU, "%s: act[%d]=%d", name, l, act[l]);
        DIR_CHECK_ARG(images[l] && aligned16(images[l]), "%s: image %d", name, l);
        const float* sc = post_scale ? post_scale[l] : nullptr;
        const float* sh = post_shift ? post_shift[l] : nullptr;
        DIR_CHECK_ARG((sc == nullptr) == (sh == nullptr), "%s: post_scale and post_shift come together (layer %d)", name, l);
        const float* b = bias ? bias[l] : nullptr;
        if ((b && !aligned16(b)) || (sc && (!aligned16(sc) || !aligned16(sh)))) return fail(DIR_E_UNSUPPORTED, "%s: bias / affine vectors must be 16-byte aligned", name);
        p.N[l] = N[l]; p.img[l] = static_cast<const unsigned char*>(images[l]); p.bias[l] = b; p.scale[l] = sc; p.shift[l] = sh;
        p.wnorm[l] = reinterpret_cast<const float*>(p.img[l] + tc_image_body(l ? N[l - 1] : Kd, N[l]));
        p.relu[l] = act[l] == DIR_ACT_RELU;
    }
    if (head_w) {
        if (!aligned16(head_w) || out_ld < 1) return fail(DIR_E_UNSUPPORTED, "%s: head_w must be 16-byte aligned, out_ld >= 1", name);
    } else if ((out_ld & 3) || out_ld < N[L - 1] || !aligned16(out)) {
        return fail(DIR_E_UNSUPPORTED, "%s: out [M, N_last] needs out_ld %% 4 == 0 and a 16-byte aligned base", name);
    }
    p.head_w = head_w; p.head_b = head_b; p.add0 = add0; p.add1 = add1; p.out = out; p.out_ld = out_ld;
    p.X = nullptr; p.x_ld = 0;
    p.tables = nullptr; p.vocab = nullptr; p.ids = nullptr; p.ids_sb = p.ids_sf = p.row_ld = 0; p.F = 0; p.lin_col = -1; p.want_fm = 0; p.lin_bias = nullptr;
    return DIR_OK;
}

template <bool GATHER, bool RS>
int tower_cs_launch_rs(const char* name, const TowerCsParams& p, dir_stream_t stream) {
    static LdsOnce once;
    if (!lds_limit(once, 160 * 1024, &tower_cs_k<GATHER, RS>)) return fail(DIR_E_HIP, "%s: cannot reserve 160 KiB of LDS", name);
    const int64_t ntiles = (p.M + TC_ROWS - 1) / TC_ROWS;
    const int64_t nwg = ntiles < kCUs ? ntiles : kCUs;            // persistent workgroups, one per CU (108 KB of LDS)
#ifdef TC_STAMP
    static unsigned long long* d_st = nullptr;
    static int calls = 0;
    if (!d_st) (void)hipMalloc(&d_st, 8 * 64 * 8);
    TowerCsParams q = p;
    q.stamps = d_st;
    (void)hipMemsetAsync(d_st, 0, 8 * 64 * 8, as_stream(stream));
    hipLaunchKernelGGL((tower_cs_k<GATHER, RS>), dim3((unsigned)nwg), dim3(64 * TC_NW), TC_SMEM, as_stream(stream), q);
    if (++calls == 300) {
        unsigned long long h[8 * 64];
        (void)hipMemcpy(h, d_st, sizeof(h), hipMemcpyDeviceToHost);
        for (int w = 0; w < 8; ++w) {
            fprintf(stderr, "wave %d:", w);
            for (int i = 1; i < 64 && h[w * 64 + i]; ++i) fprintf(stderr, " %llu", h[w * 64 + i] - h[w * 64 + i - 1]);
            fprintf(stderr, "\n");
        }
    }
    DIR_CHECK_LAUNCH(name);
    return DIR_OK;
#endif
    hipLaunchKernelGGL((tower_cs_k<GATHER, RS>), dim3((unsigned)nwg), dim3(64 * TC_NW), TC_SMEM, as_stream(stream), p);
    DIR_CHECK_LAUNCH(name);
    return DIR_OK;
}

// DIR_TOWER_RS = 1 (default) | 0, read per call (an A/B switch): the rows of every layer's input scaled by powers of two (see tc_row_sft)
template <bool GATHER>
int tower_cs_launch(const char* name, const TowerCsParams& p, dir_stream_t stream) {
    const char* e = getenv("DIR_TOWER_RS");
    return (e && e[0] == '0') ? tower_cs_launch_rs<GATHER, false>(name, p, stream) : tower_cs_launch_rs<GATHER, true>(name, p, stream);
}

}  // namespace
}  // namespace dir

using namespace dir;

extern "C" int64_t dir_tower_cs_image_bytes(int K, int N) {
    if (K <= 0 || N <= 0) return 0;
    return tc_image_body(K, N) + 4 * TC_NORM_SLOTS;           // + the trailer: TC_NORM_SLOTS floats whose maximum is max_n sum_k |W[n][k]|
}

extern "C" int dir_tower_cs_f16x2_pack_f32(const float* W, int64_t w_ld, int K, int N, void* image, int64_t image_bytes, dir_stream_t stream) {
    const char* name = "dir_tower_cs_f16x2_pack_f32";
    DIR_CHECK_ARG(W && image && K > 0 && N > 0 && w_ld >= K, "%s: bad argument (K=%d N=%d w_ld=%lld)", name, K, N, (long long)w_ld);
    DIR_CHECK_ARG(K <= 16 * TC_NT && N <= 16 * TC_NT, "%s: K=%d N=%d exceed %d", name, K, N, 16 * TC_NT);
    DIR_CHECK_ARG(aligned16(image) && image_bytes >= dir_tower_cs_image_bytes(K, N), "%s: image must be 16-byte aligned and hold "
                  "dir_tower_cs_image_bytes(K, N) bytes", name);
    const int nks = (K + 31) / 32, nct = (N + 15) / 16;
    const int64_t threads = (int64_t)nks * nct * 64 * 4;
    const int npack = (int)((threads + 255) / 256);
    float* trailer = reinterpret_cast<float*>(static_cast<unsigned char*>(image) + tc_image_body(K, N));
    hipLaunchKernelGGL(tower_cs_pack_k, dim3((unsigned)(npack + TC_NORM_WGS)), dim3(256), 0, as_stream(stream), W, w_ld, K, N, nks, nct,
                       static_cast<unsigned int*>(image), npack, trailer);
    DIR_CHECK_LAUNCH(name);
    return DIR_OK;
}

extern "C" int dir_tower_cs_f16x2_f32(const float* X, int64_t x_ld, int64_t M, int Kd, int L, const int* N, const void* const* images,
                                      const float* const* bias, const float* const* post_scale, const float* const* post_shift, const int* act,
                                      const float* head_w, const float* head_b, const float* add0, const float* add1, float* out, int64_t out_ld,
                                      dir_stream_t stream) {
    const char* name = "dir_tower_cs_f16x2_f32";
    DIR_CHECK_ARG(M >= 0, "%s: M=%lld", name, (long long)M);
    if ((x_ld & 3) || x_ld < Kd) return fail(DIR_E_UNSUPPORTED, "%s: x_ld=%lld (a multiple of 4, >= Kd)", name, (long long)x_ld);
    TowerCsParams p;
    const int rc = tower_cs_fill(name, p, Kd, L, N, images, bias, post_scale, post_shift, act, head_w, head_b, add0, add1, out, out_ld);
    if (rc != DIR_OK) return rc;
    if (M == 0) return DIR_OK;
    DIR_CHECK_ARG(X && out && aligned16(X), "%s: null or unaligned pointer", name);
    p.X = X; p.x_ld = x_ld; p.M = M;
    return tower_cs_launch<false>(name, p, stream);
}

extern "C" int dir_deepfm_tower_cs_f16x2_f32(const float* const* tables, const int64_t* vocab, int F, int K, int64_t ld, int lin_col,
                                             const int64_t* ids, int64_t stride_b, int64_t stride_f, int want_fm, int64_t M, const float* lin_bias,
                                             int L, const int* N, const void* const* images, const float* const* bias,
                                             const float* const* post_scale, const float* const* post_shift, const int* act, const float* head_w,
                                             const float* head_b, const float* add0, const float* add1, float* out, int64_t out_ld,
                                             dir_stream_t stream) {
    const char* name = "dir_deepfm_tower_cs_f16x2_f32";
    DIR_CHECK_ARG(M >= 0 && F > 0, "%s: M=%lld F=%d", name, (long long)M, F);
    if (K != 16 || F > TC_NT) return fail(DIR_E_UNSUPPORTED, "%s: K=%d F=%d (K = 16, F <= %d: one column tile per slot)", name, K, F, TC_NT);
    DIR_CHECK_ARG(want_fm == 0 || want_fm == 1, "%s: want_fm=%d", name, want_fm);
    if (ld < K + (lin_col >= 0 ? 1 : 0) || (ld & 3) || lin_col >= ld) return fail(DIR_E_UNSUPPORTED, "%s: ld=%lld lin_col=%d", name, (long long)ld, lin_col);
    DIR_CHECK_ARG(head_w || (!want_fm && lin_col < 0), "%s: the FM and first-order terms are addends of the head's logit: head_w / head_b are required", name);
    TowerCsParams p;
    const int rc = tower_cs_fill(name, p, F * K, L, N, images, bias, post_scale, post_shift, act, head_w, head_b, add0, add1, out, out_ld);
    if (rc != DIR_OK) return rc;
    if (M == 0) return DIR_OK;
    DIR_CHECK_ARG(tables && ids && out, "%s: null pointer", name);
    p.M = M; p.tables = tables; p.vocab = vocab; p.ids = ids; p.ids_sb = stride_b; p.ids_sf = stride_f; p.row_ld = ld; p.F = F; p.lin_col = lin_col;
    p.want_fm = want_fm; p.lin_bias = lin_bias;
    return tower_cs_launch<true>(name, p, stream);
}
