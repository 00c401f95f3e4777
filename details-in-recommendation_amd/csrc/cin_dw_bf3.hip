// cin_dw_bf3.hip -- weight gradient of the CIN layer on the bf16 matrix pipe with fp32-equivalent arithmetic ("bf16 x 3", cin_bf3.hip).
//
// NO REFERENCE CODE (README.md:28 links arXiv:1803.05170); derivative of the definition in include/dir_hip.h (A14):
//   dW[h, i*m + j] = sum_{b,d} G[b,h,d] * xk[b,i,d] * x0[b,j,d]
// For one field j this is a GEMM whose REDUCTION runs over the rows r = (b,d):  dW_j = (G * x0_j)^T . xk  -- the field factor sits
// inside an operand, so unlike the forward (cin_bf3.hip) the operand A_j = G * x0_j has to be formed and split per field: 52 VALU
// instructions per field and k-step in front of the 48 MFMAs of a wave's 8 column tiles (about 1.3 per MFMA with xk's split).
// Both operands are read along the reduction: a lane's 8 k-slots of a 32-row step are 8 consecutive d of one (sample, channel) --
// contiguous in memory for D = 8, 16, 32 -- so G, xk and x0 come straight from global memory, 32 bytes per lane and operand.
//
// Work split.  A work item is (block of 128 h, block of 128 i, block of JB = 4 fields, span of rows); a workgroup of 8 waves (two per
// SIMD) takes one item: wave w owns h tile w (16 rows of the output) x 8 i tiles x 4 fields = 32 accumulators (128 registers).
// Per k-step: xk's 128 x 32 tile is split once by the whole workgroup (thread = one lane-octet) into LDS as three bf16 pieces
// [piece][i tile][lane][8] (double-buffered, one barrier per step); every wave builds its four A_j (G octet x x0_j octet, split) and
// then, per i tile, reads the three B pieces once (one ds_read_b128 each) for the 24 MFMAs of the four fields.
// Spans leave partial sums part[item][span][JB][128][128]; cin_dw_bf3_reduce_k adds them in span order (bitwise reproducible).
#include "common.hpp"

namespace dir {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));
typedef unsigned int u32x4_t __attribute__((ext_vector_type(4)));

constexpr int DWB_JB = 4;
#ifndef DWB_CHAIN
#define DWB_CHAIN 1      // see dense_bf3.hip (DB3_CHAIN): a 128 x 128 layer 4.14 -> 3.9 ms
#endif

__device__ __forceinline__ unsigned int dwb_pk(float a, float b) {      // v_cvt_pk_bf16_f32 (round to nearest even), a in the low half
    typedef __bf16 pk2_t __attribute__((ext_vector_type(2)));
    const pk2_t v = {(__bf16)a, (__bf16)b};
    unsigned int w = __builtin_bit_cast(unsigned int, v);
    asm("" : "+v"(w));      // (empty: hides w's origin so that float(bf16(a)) is formed by a shift, not a second convert; see cin_bf3.hip)
    return w;
}
typedef float f32x2 __attribute__((ext_vector_type(2)));
// 8 values (four pairs) -> three bf16x8 operands that sum to them; residuals on pairs (one v_pk_add_f32 per two elements)
__device__ __forceinline__ void dwb_split8(f32x2 v0, f32x2 v1, f32x2 v2, f32x2 v3, bf16x8_t (&p)[3]) {
    const f32x2 v[4] = {v0, v1, v2, v3};
    unsigned int w[3][4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        w[0][i] = dwb_pk(v[i][0], v[i][1]);
        const f32x2 r = v[i] - (f32x2){__builtin_bit_cast(float, w[0][i] << 16), __builtin_bit_cast(float, w[0][i] & 0xffff0000u)};
        w[1][i] = dwb_pk(r[0], r[1]);
        const f32x2 t = r - (f32x2){__builtin_bit_cast(float, w[1][i] << 16), __builtin_bit_cast(float, w[1][i] & 0xffff0000u)};
        w[2][i] = dwb_pk(t[0], t[1]);
    }
#pragma unroll
    for (int q = 0; q < 3; ++q) p[q] = __builtin_bit_cast(bf16x8_t, (u32x4_t){w[q][0], w[q][1], w[q][2], w[q][3]});
}

// ---- round 4: "fp16 x 2" (cin_bf3.hip states the arithmetic) for the weight gradient -----------------------------------------------------
// G is an operand here and its magnitude is unknown, so it is scaled by ONE power of two for the whole tensor (its largest |element| lands
// in [2^12, 2^13): A_j = (s G) x0_j stays below fp16's 65 504 for |x0| < 8), found by a max pass over G (gabs_k; a caller that already
// knows max |G| passes its bit pattern instead) and taken out again by the reduce pass.  Elements more than 2^-15 below the largest keep an
// ABSOLUTE error of 2^-25 of the scaled tensor -- the reduction runs over all rows, the large elements decide its size.  xk is split as in
// the fp16 x 2 forward (same precondition: activations of O(1)).
typedef _Float16 f16x8_t __attribute__((ext_vector_type(8)));
__device__ __forceinline__ unsigned int dwb_pk_h(float a, float b) {
    typedef _Float16 h2_t __attribute__((ext_vector_type(2)));
    const h2_t v = {(_Float16)a, (_Float16)b};
    unsigned int w = __builtin_bit_cast(unsigned int, v);
    asm("" : "+v"(w));
    return w;
}
__device__ __forceinline__ void dwb_split8h(f32x2 v0, f32x2 v1, f32x2 v2, f32x2 v3, f16x8_t (&p)[2]) {
    typedef _Float16 h2_t __attribute__((ext_vector_type(2)));
    const f32x2 v[4] = {v0, v1, v2, v3};
    unsigned int w[2][4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        w[0][i] = dwb_pk_h(v[i][0], v[i][1]);
        const h2_t h = __builtin_bit_cast(h2_t, w[0][i]);
        const f32x2 r = v[i] - (f32x2){(float)h[0], (float)h[1]};
        w[1][i] = dwb_pk_h(r[0], r[1]);
    }
#pragma unroll
    for (int q = 0; q < 2; ++q) p[q] = __builtin_bit_cast(f16x8_t, (u32x4_t){w[q][0], w[q][1], w[q][2], w[q][3]});
}
template <int NP> struct DwbPc;
template <> struct DwbPc<3> {
    using op_t = bf16x8_t;
    __device__ static __forceinline__ void split8(f32x2 a, f32x2 b, f32x2 c, f32x2 d, op_t (&p)[3]) { dwb_split8(a, b, c, d, p); }
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
template <> struct DwbPc<2> {
    using op_t = f16x8_t;
    __device__ static __forceinline__ void split8(f32x2 a, f32x2 b, f32x2 c, f32x2 d, op_t (&p)[2]) { dwb_split8h(a, b, c, d, p); }
    __device__ static __forceinline__ f32x4 mma(const op_t (&a)[2], const op_t (&b)[2], f32x4 c) {
        c = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[1], b[0], c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[0], b[1], c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[0], b[0], c, 0, 0, 0);
        return c;
    }
};
// scale 2^k / its inverse for a tensor whose largest |element| has the bit pattern `bits`: the largest lands in [2^12, 2^13)
__device__ __forceinline__ float dwb_scale(unsigned int bits, bool inverse) {
    int k = 139 - (int)((bits >> 23) & 0xffu);
    k = k > 100 ? 100 : (k < -100 ? -100 : k);
    return __builtin_bit_cast(float, (unsigned int)(inverse ? 127 - k : 127 + k) << 23);
}
// max |G| as a bit pattern (non-negative floats order like their bits): one atomic per workgroup
__global__ __launch_bounds__(256) void gabs_k(const float* __restrict__ G, int64_t n4, unsigned int* __restrict__ out) {
    float mx = 0.f;
    for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < n4; e += (int64_t)gridDim.x * 256) {
        const f32x4 v = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(G) + e);
        mx = fmaxf(fmaxf(mx, fmaxf(fabsf(v[0]), fabsf(v[1]))), fmaxf(fabsf(v[2]), fabsf(v[3])));
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o, 64));
    __shared__ float wm[4];
    if ((threadIdx.x & 63) == 0) wm[threadIdx.x >> 6] = mx;
    __syncthreads();
    if (threadIdx.x == 0) atomicMax(out, __builtin_bit_cast(unsigned int, fmaxf(fmaxf(wm[0], wm[1]), fmaxf(wm[2], wm[3]))));
}

struct DwbPlan { int nhb, nib, njb, items, nspan; int64_t steps, steps_per_span; };
static DwbPlan dwb_plan(int m, int Hp, int H, int D, int64_t B) {
    DwbPlan p;
    p.nhb = (H + 127) / 128;
    p.nib = (Hp + 127) / 128;
    p.njb = (m + DWB_JB - 1) / DWB_JB;
    p.items = p.nhb * p.nib * p.njb;
    p.steps = (B * D + 31) / 32;
    int ns = kCUs / p.items;
    if (ns < 1) ns = 1;
    if ((int64_t)ns > p.steps) ns = (int)(p.steps > 0 ? p.steps : 1);
    p.nspan = ns;
    p.steps_per_span = (p.steps + ns - 1) / ns;
    return p;
}

template <int NP>
__global__ __launch_bounds__(512, 1) void cin_dw_bf3_k(const float* __restrict__ x0, const float* __restrict__ xk, const float* __restrict__ G,
                                                       int m, int Hp, int H, int D, int dshift, int nib, int njb, int nspan,
                                                       int64_t steps_per_span, int64_t steps, int64_t R, float* __restrict__ part,
                                                       const unsigned int* __restrict__ gmax /* NP == 2: bit pattern of max |G| */) {
    using Pc = DwbPc<NP>;
    using op_t = typename Pc::op_t;
    __shared__ __attribute__((aligned(16))) unsigned int Bp[2][NP][8][64][4];     // xk pieces of one k-step: [buffer][piece][i tile][lane][8 halves]
    float gs = 1.f;
    if constexpr (NP == 2) gs = dwb_scale(*gmax, false);

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, n = lane & 15, lg = lane >> 4;
    // work item: blockIdx.x = ((hb * nib + ib) * njb + jb) * nspan + span
    int q = blockIdx.x;
    const int span = q % nspan; q /= nspan;
    const int jb = q % njb; q /= njb;
    const int ib = q % nib;
    const int hb = q / nib;
    const int64_t s_begin = (int64_t)span * steps_per_span;
    int64_t s_end = s_begin + steps_per_span;
    if (s_end > steps) s_end = steps;

    // a lane's 8 k-slots of step s: rows r = 32 s + 8 lg + e  ->  one (sample, 8 consecutive d) for D >= 8
    const int hrow = min(128 * hb + 16 * wave + n, H - 1);          // this lane's output row h (A operand); clamped rows are never reduced
    const int icol = min(128 * ib + 16 * wave + n, Hp - 1);         // the xk channel this THREAD splits (i tile = its wave index)
    const int j0 = DWB_JB * jb;

    f32x4 acc[DWB_JB][8];
#pragma unroll
    for (int j = 0; j < DWB_JB; ++j)
#pragma unroll
        for (int it = 0; it < 8; ++it) acc[j][it] = (f32x4){0.f, 0.f, 0.f, 0.f};

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
    // Raw operands travel one step ahead of their use: the loads of step s+1 (G, x0) and s+2 (xk) are issued after step s's splits
    // and land under its 192 MFMAs, in the registers the splits have just freed.
    f32x4 rg[2], rx[DWB_JB][2], rb[2];
    auto load_a = [&](int64_t s) {
        octet(G, hrow, H, s, rg);
#pragma unroll
        for (int j = 0; j < DWB_JB; ++j) {
            if (j0 + j < m) octet(x0, j0 + j, m, s, rx[j]);
            else rx[j][0] = rx[j][1] = (f32x4){0.f, 0.f, 0.f, 0.f};
        }
    };
    const bool icol_in = 128 * ib + 16 * wave + n < Hp;
    auto load_b = [&](int64_t s) {
        octet(xk, icol, Hp, s, rb);
        if (!icol_in) rb[0] = rb[1] = (f32x4){0.f, 0.f, 0.f, 0.f};
    };
    auto split_b = [&](int buf) {                 // this thread's octet of xk -> its pieces in LDS
        op_t p[NP];
        Pc::split8((f32x2){rb[0][0], rb[0][1]}, (f32x2){rb[0][2], rb[0][3]}, (f32x2){rb[1][0], rb[1][1]}, (f32x2){rb[1][2], rb[1][3]}, p);
#pragma unroll
        for (int pc = 0; pc < NP; ++pc) *reinterpret_cast<op_t*>(&Bp[buf][pc][wave][lane][0]) = p[pc];
    };

    if (s_begin < s_end) {
        load_b(s_begin);
        load_a(s_begin);
        split_b(0);
        load_b(s_begin + 1);
    }
    __syncthreads();
    int buf = 0;
    for (int64_t s = s_begin; s < s_end; ++s, buf ^= 1) {
        // A operands: G octet of row h times the x0 octet of each of the JB fields, split
        op_t a[DWB_JB][NP];
        f32x2 g0 = {rg[0][0], rg[0][1]}, g1 = {rg[0][2], rg[0][3]}, g2 = {rg[1][0], rg[1][1]}, g3 = {rg[1][2], rg[1][3]};
        if constexpr (NP == 2) { g0 *= gs; g1 *= gs; g2 *= gs; g3 *= gs; }
#pragma unroll
        for (int j = 0; j < DWB_JB; ++j)
            Pc::split8(g0 * (f32x2){rx[j][0][0], rx[j][0][1]}, g1 * (f32x2){rx[j][0][2], rx[j][0][3]}, g2 * (f32x2){rx[j][1][0], rx[j][1][1]},
                       g3 * (f32x2){rx[j][1][2], rx[j][1][3]}, a[j]);
        if (s + 1 < s_end) split_b(buf ^ 1);
        load_a(s + 1);                                   // (rows past R read as zeros; a step past the span is loaded and not used)
        load_b(s + 2);
        // The xk pieces of i tile `it + 1` are read while tile `it` multiplies: by hand (inline asm + a counted lgkmcnt; LDS operations return
        // in order, so "at most NP outstanding" = tile it's pieces are here).  The compiler read a tile's pieces, waited for them with
        // lgkmcnt(0) and only then issued its 12 matrix instructions -- an LDS round trip exposed per tile, eight per step (round 5).
        {
            typedef unsigned int dwb_u32x4 __attribute__((ext_vector_type(4)));
            const unsigned int bl32 = (unsigned int)(size_t)&Bp[buf][0][0][lane][0];
            dwb_u32x4 bq[2][NP];
#define DWB_DS_READ(dst, off) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst) : "v"(bl32), "n"(off))
#pragma unroll
            for (int pc = 0; pc < NP; ++pc) DWB_DS_READ(bq[0][pc], pc * 8 * 1024);
#pragma unroll
            for (int it = 0; it < 8; ++it) {
                if (it + 1 < 8) {
#pragma unroll
                    for (int pc = 0; pc < NP; ++pc) DWB_DS_READ(bq[(it + 1) & 1][pc], (pc * 8 + it + 1) * 1024);
                    if constexpr (NP == 2) asm volatile("s_waitcnt lgkmcnt(2)" : "+v"(bq[it & 1][0]), "+v"(bq[it & 1][1]));
                    else asm volatile("s_waitcnt lgkmcnt(3)" : "+v"(bq[it & 1][0]), "+v"(bq[it & 1][1]), "+v"(bq[it & 1][NP - 1]));
                } else {
                    if constexpr (NP == 2) asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(bq[it & 1][0]), "+v"(bq[it & 1][1]));
                    else asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(bq[it & 1][0]), "+v"(bq[it & 1][1]), "+v"(bq[it & 1][NP - 1]));
                }
                __builtin_amdgcn_sched_barrier(0);
                op_t b[NP];
#pragma unroll
                for (int pc = 0; pc < NP; ++pc) b[pc] = __builtin_bit_cast(op_t, bq[it & 1][pc]);
#pragma unroll
                for (int j = 0; j < DWB_JB; ++j) {
                    acc[j][it] = Pc::mma(a[j], b, acc[j][it]);
#if DWB_CHAIN
                    __builtin_amdgcn_sched_barrier(0);      // one dependent chain per accumulator (dense_bf3.hip: DB3_CHAIN)
#endif
                }
                __builtin_amdgcn_sched_barrier(0);
            }
#undef DWB_DS_READ
        }
        __syncthreads();
    }
    // partial sums: part[blockIdx.x][j][h = 16 wave + 4 lg + q][i = 16 it + n]   (C/D map: col = lane & 15, row = 4 (lane >> 4) + reg)
    float* dst = part + (int64_t)blockIdx.x * DWB_JB * 128 * 128;
#pragma unroll
    for (int j = 0; j < DWB_JB; ++j)
#pragma unroll
        for (int it = 0; it < 8; ++it)
#pragma unroll
            for (int qq = 0; qq < 4; ++qq) dst[((int64_t)j * 128 + 16 * wave + 4 * lg + qq) * 128 + 16 * it + n] = acc[j][it][qq];
}

// dW[h, i*m + j] (+)= sum over spans, in span order
__global__ __launch_bounds__(256) void cin_dw_bf3_reduce_k(const float* __restrict__ part, int m, int Hp, int H, int nib, int njb, int nspan,
                                                          int accumulate, float* __restrict__ dW,
                                                          const unsigned int* __restrict__ gmax /* fp16 x 2: the scale to take out, or null */) {
    const int64_t total = (int64_t)H * Hp * m;
    const float inv = gmax ? dwb_scale(*gmax, true) : 1.f;
    for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < total; e += (int64_t)gridDim.x * 256) {
        const int j = (int)(e % m);
        const int i = (int)((e / m) % Hp);
        const int h = (int)(e / ((int64_t)m * Hp));
        const int hb = h >> 7, ib = i >> 7, jb = j / DWB_JB;
        const int64_t item = ((int64_t)hb * nib + ib) * njb + jb;
        const float* p = part + ((item * nspan) * DWB_JB + (j - jb * DWB_JB)) * 128 * 128 + (int64_t)(h & 127) * 128 + (i & 127);
        float s = 0.f;
        for (int sp = 0; sp < nspan; ++sp) s += p[(int64_t)sp * DWB_JB * 128 * 128];
        s *= inv;
        dW[e] = accumulate ? dW[e] + s : s;
    }
}

}  // namespace dir

using namespace dir;

extern "C" int64_t dir_cin_dw_bf16x3_workspace_bytes(int m, int Hp, int H, int D, int64_t B) {
    if (m <= 0 || Hp <= 0 || H <= 0 || D <= 0 || B < 0) return 0;
    const DwbPlan p = dwb_plan(m, Hp, H, D, B);
    return (int64_t)p.items * p.nspan * DWB_JB * 128 * 128 * (int64_t)sizeof(float);
}

static int dwb_run(const char* name, int np, const float* x0, const float* xk, const float* G, int m, int Hp, int H, int D, int64_t B,
                   int accumulate, float* dW, void* workspace, int64_t workspace_bytes, const unsigned int* gmax_bits, dir_stream_t stream) {
    DIR_CHECK_ARG(dW, "%s: null pointer", name);
    DIR_CHECK_ARG(m > 0 && Hp > 0 && H > 0 && D > 0 && B >= 0, "%s: m=%d Hp=%d H=%d D=%d", name, m, Hp, H, D);
    if (!(D == 8 || D == 16 || D == 32)) return fail(DIR_E_UNSUPPORTED, "%s: D=%d (supported: 8, 16, 32; use dir_cin_dw_f32)", name, D);
    hipStream_t st = as_stream(stream);
    const int64_t n = (int64_t)H * Hp * m;
    if (B == 0) {                                    // an empty batch has a zero gradient (empty operands have no storage: null allowed)
        if (!accumulate && zero_async(dW, n * sizeof(float), st) != hipSuccess) return fail(DIR_E_HIP, "%s: memset failed", name);
        return DIR_OK;
    }
    DIR_CHECK_ARG(x0 && xk && G && workspace, "%s: null pointer", name);
    if (!(aligned16(x0) && aligned16(xk) && aligned16(G) && aligned16(workspace)))
        return fail(DIR_E_BADARG, "%s: x0 / xk / G / workspace must be 16-byte aligned", name);
    const int64_t part_bytes = dir_cin_dw_bf16x3_workspace_bytes(m, Hp, H, D, B);
    DIR_CHECK_ARG(workspace_bytes >= part_bytes + (np == 2 ? 256 : 0), "%s: workspace smaller than its workspace_bytes query", name);
    int dshift = 0;
    while ((1 << dshift) < D) ++dshift;
    const DwbPlan p = dwb_plan(m, Hp, H, D, B);
    float* part = static_cast<float*>(workspace);
    if (np == 2) {
        if (!gmax_bits) {                      // max |G| by a pass of our own, into the workspace's tail
            unsigned int* slot = reinterpret_cast<unsigned int*>(static_cast<unsigned char*>(workspace) + part_bytes);
            if (zero_async(slot, 256, st) != hipSuccess) return fail(DIR_E_HIP, "%s: zeroing failed", name);
            const int64_t n4 = B * H * D / 4;                      // D % 8 == 0: whole vectors
            hipLaunchKernelGGL(gabs_k, dim3(grid_for((n4 + 255) / 256, 8)), dim3(256), 0, st, G, n4, slot);
            DIR_CHECK_LAUNCH("cin_dw gabs");
            gmax_bits = slot;
        }
        hipLaunchKernelGGL(cin_dw_bf3_k<2>, dim3((unsigned)(p.items * p.nspan)), dim3(512), 0, st, x0, xk, G, m, Hp, H, D, dshift, p.nib, p.njb,
                           p.nspan, p.steps_per_span, p.steps, B * D, part, gmax_bits);
    } else {
        hipLaunchKernelGGL(cin_dw_bf3_k<3>, dim3((unsigned)(p.items * p.nspan)), dim3(512), 0, st, x0, xk, G, m, Hp, H, D, dshift, p.nib, p.njb,
                           p.nspan, p.steps_per_span, p.steps, B * D, part, nullptr);
    }
    DIR_CHECK_LAUNCH(name);
    hipLaunchKernelGGL(cin_dw_bf3_reduce_k, dim3(grid_for((n + 255) / 256)), dim3(256), 0, st, static_cast<const float*>(workspace), m, Hp, H, p.nib,
                       p.njb, p.nspan, accumulate, dW, np == 2 ? gmax_bits : nullptr);
    DIR_CHECK_LAUNCH("cin_dw reduce");
    return DIR_OK;
}

extern "C" int dir_cin_dw_bf16x3_f32(const float* x0, const float* xk, const float* G, int m, int Hp, int H, int D, int64_t B, int accumulate,
                                     float* dW, void* workspace, int64_t workspace_bytes, dir_stream_t stream) {
    return dwb_run("dir_cin_dw_bf16x3_f32", 3, x0, xk, G, m, Hp, H, D, B, accumulate, dW, workspace, workspace_bytes, nullptr, stream);
}

extern "C" int64_t dir_cin_dw_f16x2_workspace_bytes(int m, int Hp, int H, int D, int64_t B) {
    if (m <= 0 || Hp <= 0 || H <= 0 || D <= 0 || B < 0) return 0;
    return dir_cin_dw_bf16x3_workspace_bytes(m, Hp, H, D, B) + 256;
}

extern "C" int dir_cin_dw_f16x2_f32(const float* x0, const float* xk, const float* G, int m, int Hp, int H, int D, int64_t B, int accumulate,
                                    float* dW, void* workspace, int64_t workspace_bytes, const unsigned int* g_absmax_bits, dir_stream_t stream) {
    return dwb_run("dir_cin_dw_f16x2_f32", 2, x0, xk, G, m, Hp, H, D, B, accumulate, dW, workspace, workspace_bytes, g_absmax_bits, stream);
}
