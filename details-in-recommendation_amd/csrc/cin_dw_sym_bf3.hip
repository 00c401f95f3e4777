// cin_dw_sym_bf3.hip -- weight gradient of the FIRST CIN layer (xk = x0, Hp = m) on the bf16 matrix pipe with fp32-equivalent arithmetic
// ("bf16 x 3", cin_bf3.hip / cin_dw_bf3.hip).
//
// NO REFERENCE CODE (README.md:28 links arXiv:1803.05170); derivative of the definition in include/dir_hip.h (A14) for X^0 on both sides:
//   dW[h, i*m + j] = sum_{b,d} G[b,h,d] * x0[b,i,d] * x0[b,j,d]                 -- symmetric in (i, j)
// cin_dw_bf3.hip puts the field factor into the G operand and pairs it with 128 xk channels: with only m = 26 channels on the xk side four
// fifths of its column tiles would be padding, and the fp32-MFMA kernel (cin_bwd.hip) took 1.5 ms for the layer at the BASELINE shape.
// Here the PAIR is the column: for the m (m + 1) / 2 unordered pairs p = (i <= j) the operand Q_p[r] = x0[r,i] * x0[r,j] is formed and
// split per k-step (as cin_dw_bf3.hip forms G * x0_j), G's 128 x 32 tile is split once per step by the whole workgroup into LDS, and
//   C[h, p] = sum_r G[r,h] * Q_p[r]          (a GEMM over the rows r = (b, d) with 351 columns at m = 26 instead of 676)
// is written to both dW[h, i*m + j] and dW[h, j*m + i] by the reduce kernel.
//
// Work split.  A work item is (block of 128 h, block of 24 pair tiles = 384 pairs, span of rows); a workgroup of 8 waves takes one item:
// wave w owns pair tiles w, w + 8, w + 16 of the block (16 pairs each: lane n = pair) x 8 h tiles = 24 accumulators (96 registers).
// Per k-step every thread splits one octet of G (thread (wave, lane) = row 16 wave + n of the h block, k octet lane >> 4) into LDS
// (double-buffered, one barrier per step); every wave builds the B operands of its pair tiles (x0_i octet x x0_j octet, split) and then,
// per h tile, reads the three A pieces once (one ds_read_b128 each) for the 18 MFMAs of its three pair tiles.
// Spans leave partial sums part[item][span][128 h][384 pairs]; cin_dw_sym_reduce_k adds them in span order (bitwise reproducible).
#include "common.hpp"

namespace dir {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));
typedef unsigned int u32x4_t __attribute__((ext_vector_type(4)));

constexpr int DWS_TPW = 3;                   // pair tiles per wave
constexpr int DWS_TILES = 8 * DWS_TPW;       // pair tiles per work item
constexpr int DWS_PAIRS = 16 * DWS_TILES;    // pairs per work item (384)
constexpr int DWS_MAXM = 64;

__device__ __forceinline__ unsigned int dws_pk(float a, float b) {      // v_cvt_pk_bf16_f32 (round to nearest even), a in the low half
    typedef __bf16 pk2_t __attribute__((ext_vector_type(2)));
    const pk2_t v = {(__bf16)a, (__bf16)b};
    unsigned int w = __builtin_bit_cast(unsigned int, v);
    asm("" : "+v"(w));      // (empty: hides w's origin so that float(bf16(a)) is formed by a shift, not a second convert; see cin_bf3.hip)
    return w;
}
// 8 values (four pairs) -> three bf16x8 operands that sum to them
__device__ __forceinline__ void dws_split8(f32x2 v0, f32x2 v1, f32x2 v2, f32x2 v3, bf16x8_t (&p)[3]) {
    const f32x2 v[4] = {v0, v1, v2, v3};
    unsigned int w[3][4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        w[0][i] = dws_pk(v[i][0], v[i][1]);
        const f32x2 r = v[i] - (f32x2){__builtin_bit_cast(float, w[0][i] << 16), __builtin_bit_cast(float, w[0][i] & 0xffff0000u)};
        w[1][i] = dws_pk(r[0], r[1]);
        const f32x2 t = r - (f32x2){__builtin_bit_cast(float, w[1][i] << 16), __builtin_bit_cast(float, w[1][i] & 0xffff0000u)};
        w[2][i] = dws_pk(t[0], t[1]);
    }
#pragma unroll
    for (int q = 0; q < 3; ++q) p[q] = __builtin_bit_cast(bf16x8_t, (u32x4_t){w[q][0], w[q][1], w[q][2], w[q][3]});
}

// ---- round 4: fp16 x 2 (cin_bf3.hip states the arithmetic; cin_dw_bf3.hip the tensor scale): G times ONE power of two from the bit pattern of
// max |G| (largest element into [2^14, 2^15): G is an operand on its own here, the pair products x0_i x0_j are the other one and are split as
// in the fp16 x 2 forward's PAIRS form); the reduce pass takes the scale out again
typedef _Float16 f16x8_t __attribute__((ext_vector_type(8)));
__device__ __forceinline__ unsigned int dws_pk_h(float a, float b) {
    typedef _Float16 h2_t __attribute__((ext_vector_type(2)));
    const h2_t v = {(_Float16)a, (_Float16)b};
    unsigned int w = __builtin_bit_cast(unsigned int, v);
    asm("" : "+v"(w));
    return w;
}
__device__ __forceinline__ void dws_split8h(f32x2 v0, f32x2 v1, f32x2 v2, f32x2 v3, f16x8_t (&p)[2]) {
    typedef _Float16 h2_t __attribute__((ext_vector_type(2)));
    const f32x2 v[4] = {v0, v1, v2, v3};
    unsigned int w[2][4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        w[0][i] = dws_pk_h(v[i][0], v[i][1]);
        const h2_t h = __builtin_bit_cast(h2_t, w[0][i]);
        const f32x2 r = v[i] - (f32x2){(float)h[0], (float)h[1]};
        w[1][i] = dws_pk_h(r[0], r[1]);
    }
#pragma unroll
    for (int q = 0; q < 2; ++q) p[q] = __builtin_bit_cast(f16x8_t, (u32x4_t){w[q][0], w[q][1], w[q][2], w[q][3]});
}
template <int NP> struct DwsPc;
template <> struct DwsPc<3> {
    using op_t = bf16x8_t;
    __device__ static __forceinline__ void split8(f32x2 a, f32x2 b, f32x2 c, f32x2 d, op_t (&p)[3]) { dws_split8(a, b, c, d, p); }
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
template <> struct DwsPc<2> {
    using op_t = f16x8_t;
    __device__ static __forceinline__ void split8(f32x2 a, f32x2 b, f32x2 c, f32x2 d, op_t (&p)[2]) { dws_split8h(a, b, c, d, p); }
    __device__ static __forceinline__ f32x4 mma(const op_t (&a)[2], const op_t (&b)[2], f32x4 c) {
        c = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[1], b[0], c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[0], b[1], c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[0], b[0], c, 0, 0, 0);
        return c;
    }
};
__device__ __forceinline__ float dws_scale(unsigned int bits, bool inverse) {     // largest |element| into [2^14, 2^15); k clamped to +-100
    int k = 141 - (int)((bits >> 23) & 0xffu);
    k = k > 100 ? 100 : (k < -100 ? -100 : k);
    return __builtin_bit_cast(float, (unsigned int)(inverse ? 127 - k : 127 + k) << 23);
}

// pair index of (a <= b) among the m (m + 1) / 2 unordered pairs, row-major over a
__host__ __device__ __forceinline__ int dws_pair_index(int a, int b, int m) { return a * m - a * (a - 1) / 2 + (b - a); }

struct DwsPlan { int nhb, npb, npairs, items, nspan; int64_t steps, steps_per_span; };
static DwsPlan dws_plan(int m, int H, int D, int64_t B) {
    DwsPlan p;
    p.nhb = (H + 127) / 128;
    p.npairs = m * (m + 1) / 2;
    p.npb = (p.npairs + DWS_PAIRS - 1) / DWS_PAIRS;
    p.items = p.nhb * p.npb;
    p.steps = (B * D + 31) / 32;
    int ns = kCUs / p.items;
    if (ns < 1) ns = 1;
    if ((int64_t)ns > p.steps) ns = (int)(p.steps > 0 ? p.steps : 1);
    p.nspan = ns;
    p.steps_per_span = (p.steps + ns - 1) / ns;
    return p;
}

template <int NP>
__global__ __launch_bounds__(512, 1) void cin_dw_sym_bf3_k(const float* __restrict__ x0, const float* __restrict__ G, int m, int H, int D, int dshift,
                                                           int npb, int npairs, int nspan, int64_t steps_per_span, int64_t steps, int64_t R,
                                                           float* __restrict__ part, const unsigned int* __restrict__ gbits /* NP == 2 */) {
    using Pc = DwsPc<NP>;
    using op_t = typename Pc::op_t;
    __shared__ __attribute__((aligned(16))) unsigned int Ap[2][NP][8][64][4];     // G pieces of one k-step: [buffer][piece][h tile][lane][8 halves]
    float gs = 1.f;
    if constexpr (NP == 2) gs = dws_scale(*gbits, false);
    __shared__ unsigned char pair_i[DWS_PAIRS], pair_j[DWS_PAIRS];                 // this item's pairs (padding: 255)

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, n = lane & 15, lg = lane >> 4;
    // work item: blockIdx.x = ((hb * npb + pb) * nspan + span
    int q = blockIdx.x;
    const int span = q % nspan; q /= nspan;
    const int pb = q % npb;
    const int hb = q / npb;
    const int64_t s_begin = (int64_t)span * steps_per_span;
    int64_t s_end = s_begin + steps_per_span;
    if (s_end > steps) s_end = steps;

    // the item's pairs: p = DWS_PAIRS * pb + local, enumerated row-major over i <= j
    if (tid < DWS_PAIRS) {
        const int p = DWS_PAIRS * pb + tid;
        int pi = 255, pj = 255;
        if (p < npairs) {
            int a = 0, rem = p;
            while (rem >= m - a) { rem -= m - a; ++a; }      // (at most m iterations, once per workgroup)
            pi = a;
            pj = a + rem;
        }
        pair_i[tid] = (unsigned char)pi;
        pair_j[tid] = (unsigned char)pj;
    }
    __syncthreads();
    // this lane's pairs: tile t of the wave is item tile wave + 8 t, the lane's pair is 16 * tile + n.  A padded pair computes the (0, 0)
    // product into a column of the partial sums that the reduce kernel never reads.
    int fi[DWS_TPW], fj[DWS_TPW];
#pragma unroll
    for (int t = 0; t < DWS_TPW; ++t) {
        const int pl = 16 * (wave + 8 * t) + n;
        fi[t] = pair_i[pl] == 255 ? 0 : pair_i[pl];
        fj[t] = pair_j[pl] == 255 ? 0 : pair_j[pl];
    }

    const int hrow = min(128 * hb + 16 * wave + n, H - 1);          // the G row this THREAD splits (h tile = its wave index); clamped rows are never reduced
    const bool hrow_in = 128 * hb + 16 * wave + n < H;

    f32x4 acc[DWS_TPW][8];
#pragma unroll
    for (int t = 0; t < DWS_TPW; ++t)
#pragma unroll
        for (int ht = 0; ht < 8; ++ht) acc[t][ht] = (f32x4){0.f, 0.f, 0.f, 0.f};

    auto octet = [&](const float* base, int ch, int C, int64_t s, f32x4 (&v)[2]) {     // 8 consecutive d of channel ch of the lane's sample
        const int64_t r = 32 * s + 8 * lg;
        if (r < R) {                                                                  // R % 8 == 0 (D >= 8): an octet is inside or outside
            const float* p = base + (((r >> dshift) * C + ch) << dshift) + (r & (D - 1));
            v[0] = *reinterpret_cast<const f32x4*>(p);
            v[1] = *reinterpret_cast<const f32x4*>(p + 4);
        } else {
            v[0] = v[1] = (f32x4){0.f, 0.f, 0.f, 0.f};
        }
    };
    // raw operands travel one step ahead of their use (x0 octets) or two (G, which goes through LDS)
    f32x4 rg[2], ri[DWS_TPW][2], rj[DWS_TPW][2];
    auto load_b = [&](int64_t s) {
#pragma unroll
        for (int t = 0; t < DWS_TPW; ++t) {
            octet(x0, fi[t], m, s, ri[t]);
            octet(x0, fj[t], m, s, rj[t]);
        }
    };
    auto load_a = [&](int64_t s) {
        octet(G, hrow, H, s, rg);
        if (!hrow_in) rg[0] = rg[1] = (f32x4){0.f, 0.f, 0.f, 0.f};
    };
    auto split_a = [&](int buf) {                 // this thread's octet of G -> its pieces in LDS
        op_t p[NP];
        f32x2 g0 = {rg[0][0], rg[0][1]}, g1 = {rg[0][2], rg[0][3]}, g2 = {rg[1][0], rg[1][1]}, g3 = {rg[1][2], rg[1][3]};
        if constexpr (NP == 2) { g0 *= gs; g1 *= gs; g2 *= gs; g3 *= gs; }
        Pc::split8(g0, g1, g2, g3, p);
#pragma unroll
        for (int pc = 0; pc < NP; ++pc) *reinterpret_cast<op_t*>(&Ap[buf][pc][wave][lane][0]) = p[pc];
    };

    if (s_begin < s_end) {
        load_a(s_begin);
        load_b(s_begin);
        split_a(0);
        load_a(s_begin + 1);
    }
    __syncthreads();
    int buf = 0;
    for (int64_t s = s_begin; s < s_end; ++s, buf ^= 1) {
        // B operands: the x0_i octet times the x0_j octet of each of the wave's pair tiles, split
        op_t b[DWS_TPW][NP];
#pragma unroll
        for (int t = 0; t < DWS_TPW; ++t)
            Pc::split8((f32x2){ri[t][0][0], ri[t][0][1]} * (f32x2){rj[t][0][0], rj[t][0][1]},
                       (f32x2){ri[t][0][2], ri[t][0][3]} * (f32x2){rj[t][0][2], rj[t][0][3]},
                       (f32x2){ri[t][1][0], ri[t][1][1]} * (f32x2){rj[t][1][0], rj[t][1][1]},
                       (f32x2){ri[t][1][2], ri[t][1][3]} * (f32x2){rj[t][1][2], rj[t][1][3]}, b[t]);
        if (s + 1 < s_end) split_a(buf ^ 1);
        load_b(s + 1);                                   // (rows past R read as zeros; a step past the span is loaded and not used)
        load_a(s + 2);
#pragma unroll
        for (int ht = 0; ht < 8; ++ht) {
            op_t a[NP];
#pragma unroll
            for (int pc = 0; pc < NP; ++pc) a[pc] = *reinterpret_cast<const op_t*>(&Ap[buf][pc][ht][lane][0]);
#pragma unroll
            for (int t = 0; t < DWS_TPW; ++t) {
                acc[t][ht] = Pc::mma(a, b[t], acc[t][ht]);
                __builtin_amdgcn_sched_barrier(0);      // one dependent chain per accumulator (dense_bf3.hip: DB3_CHAIN)
            }
        }
        __syncthreads();
    }
    // partial sums: part[blockIdx.x][h = 16 ht + 4 lg + q][pair = 16 (wave + 8 t) + n]   (C/D map: col = lane & 15, row = 4 (lane >> 4) + reg)
    float* dst = part + (int64_t)blockIdx.x * 128 * DWS_PAIRS;
#pragma unroll
    for (int t = 0; t < DWS_TPW; ++t)
#pragma unroll
        for (int ht = 0; ht < 8; ++ht)
#pragma unroll
            for (int qq = 0; qq < 4; ++qq) dst[(int64_t)(16 * ht + 4 * lg + qq) * DWS_PAIRS + 16 * (wave + 8 * t) + n] = acc[t][ht][qq];
}

// dW[h, i*m + j] (+)= sum over spans of C[h, pair(min(i,j), max(i,j))], in span order
__global__ __launch_bounds__(256) void cin_dw_sym_reduce_k(const float* __restrict__ part, int m, int H, int npb, int nspan, int accumulate,
                                                          float* __restrict__ dW, const unsigned int* __restrict__ gbits /* fp16 x 2: the scale to take out */) {
    const int64_t total = (int64_t)H * m * m;
    const float inv = gbits ? dws_scale(*gbits, true) : 1.f;
    for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < total; e += (int64_t)gridDim.x * 256) {
        const int j = (int)(e % m);
        const int i = (int)((e / m) % m);
        const int h = (int)(e / ((int64_t)m * m));
        const int p = i <= j ? dws_pair_index(i, j, m) : dws_pair_index(j, i, m);
        const int pb = p / DWS_PAIRS, pl = p - pb * DWS_PAIRS, hb = h >> 7;
        const int64_t item = (int64_t)hb * npb + pb;
        const float* src = part + (item * nspan) * 128 * DWS_PAIRS + (int64_t)(h & 127) * DWS_PAIRS + pl;
        float s = 0.f;
        for (int sp = 0; sp < nspan; ++sp) s += src[(int64_t)sp * 128 * DWS_PAIRS];
        s *= inv;
        dW[e] = accumulate ? dW[e] + s : s;
    }
}

}  // namespace dir

using namespace dir;

extern "C" int64_t dir_cin_dw_sym_bf16x3_workspace_bytes(int m, int H, int D, int64_t B) {
    if (m <= 0 || H <= 0 || D <= 0 || B < 0) return 0;
    const DwsPlan p = dws_plan(m, H, D, B);
    return (int64_t)p.items * p.nspan * 128 * DWS_PAIRS * (int64_t)sizeof(float);
}

static int dws_run(const char* name, int np, const float* x0, const float* G, int m, int H, int D, int64_t B, int accumulate, float* dW,
                   void* workspace, int64_t workspace_bytes, const unsigned int* gbits, dir_stream_t stream) {
    DIR_CHECK_ARG(np == 3 || gbits, "%s: g_absmax_bits is null", name);
    DIR_CHECK_ARG(dW, "%s: null pointer", name);
    DIR_CHECK_ARG(m > 0 && H > 0 && D > 0 && B >= 0, "%s: m=%d H=%d D=%d", name, m, H, D);
    if (!(D == 8 || D == 16 || D == 32) || m > DWS_MAXM)
        return fail(DIR_E_UNSUPPORTED, "%s: m=%d D=%d (supported: m <= %d, D in 8, 16, 32; use dir_cin_dw_f32)", name, m, D, DWS_MAXM);
    hipStream_t st = as_stream(stream);
    const int64_t n = (int64_t)H * m * m;
    if (B == 0) {                                    // an empty batch has a zero gradient (empty operands have no storage: null allowed)
        if (!accumulate && zero_async(dW, n * sizeof(float), st) != hipSuccess) return fail(DIR_E_HIP, "%s: memset failed", name);
        return DIR_OK;
    }
    DIR_CHECK_ARG(x0 && G && workspace, "%s: null pointer", name);
    if (!(aligned16(x0) && aligned16(G) && aligned16(workspace)))
        return fail(DIR_E_BADARG, "%s: x0 / G / workspace must be 16-byte aligned", name);
    DIR_CHECK_ARG(workspace_bytes >= dir_cin_dw_sym_bf16x3_workspace_bytes(m, H, D, B), "%s: workspace smaller than "
                  "dir_cin_dw_sym_bf16x3_workspace_bytes(m, H, D, B)", name);
    int dshift = 0;
    while ((1 << dshift) < D) ++dshift;
    const DwsPlan p = dws_plan(m, H, D, B);
    if (np == 2)
        hipLaunchKernelGGL(cin_dw_sym_bf3_k<2>, dim3((unsigned)(p.items * p.nspan)), dim3(512), 0, st, x0, G, m, H, D, dshift, p.npb, p.npairs, p.nspan,
                           p.steps_per_span, p.steps, B * D, static_cast<float*>(workspace), gbits);
    else
        hipLaunchKernelGGL(cin_dw_sym_bf3_k<3>, dim3((unsigned)(p.items * p.nspan)), dim3(512), 0, st, x0, G, m, H, D, dshift, p.npb, p.npairs, p.nspan,
                           p.steps_per_span, p.steps, B * D, static_cast<float*>(workspace), nullptr);
    DIR_CHECK_LAUNCH(name);
    hipLaunchKernelGGL(cin_dw_sym_reduce_k, dim3(grid_for((n + 255) / 256)), dim3(256), 0, st, static_cast<const float*>(workspace), m, H, p.npb,
                       p.nspan, accumulate, dW, np == 2 ? gbits : nullptr);
    DIR_CHECK_LAUNCH("cin_dw_sym reduce");
    return DIR_OK;
}

extern "C" int dir_cin_dw_sym_bf16x3_f32(const float* x0, const float* G, int m, int H, int D, int64_t B, int accumulate, float* dW, void* workspace,
                                         int64_t workspace_bytes, dir_stream_t stream) {
    return dws_run("dir_cin_dw_sym_bf16x3_f32", 3, x0, G, m, H, D, B, accumulate, dW, workspace, workspace_bytes, nullptr, stream);
}

// fp16 x 2: G scaled by one power of two from g_absmax_bits (DEVICE: the bit pattern of an upper bound of max |G|, e.g. what
// dir_cin_layer_grad_f16x2_f32 leaves), the pair products x0_i x0_j split as in the fp16 x 2 forward (|x0| of O(1)).  Workspace as the bf16 x 3 entry.
extern "C" int dir_cin_dw_sym_f16x2_f32(const float* x0, const float* G, int m, int H, int D, int64_t B, int accumulate, float* dW, void* workspace,
                                        int64_t workspace_bytes, const unsigned int* g_absmax_bits, dir_stream_t stream) {
    return dws_run("dir_cin_dw_sym_f16x2_f32", 2, x0, G, m, H, D, B, accumulate, dW, workspace, workspace_bytes, g_absmax_bits, stream);
}
