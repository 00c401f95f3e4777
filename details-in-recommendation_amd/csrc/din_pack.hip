// din_pack.hip -- DIN local activation unit + pooling, forward, fp16 x 2 arithmetic, for the (K = 64, H1 <= 80, H2 <= 48) shape class of
// BASELINE.json configs[3]: the PACKED form of din_wave.hip's kernel (round 6).  NO REFERENCE CODE (README.md:27 links arXiv:1706.06978);
// the definition is include/dir_hip.h (A13) and oracle/dir_oracle.c.
//
// What din_wave_k pays per SAMPLE, and this kernel does not:
//   * tile padding.  A wave there owns one sample and computes its history in 16-row MFMA tiles: lengths U{1..50} cost 33.3 rows for
//     25.5.  Here a wave owns a RUN of consecutive samples and lays their valid rows end to end: row q of a block of 16 samples is row
//     (q & 15) of tile (q >> 4) whatever sample it belongs to, and only the last tile of a block is partly empty (~2 %).
//   * the per-sample term c = (Wa - Wd)^T a + b1 on the VALU (80 fmas + 20 conflicting LDS reads + 10 shuffles per sample: the
//     5.6 M LDS bank conflicts of profiles/r05_pmc_din.json).  Here the 16 candidate rows of a block are ONE MFMA operand
//     (n = sample): 30 matrix instructions per 16 samples from a pre-split image of (Wa - Wd)^T.
//   * a queue ticket, descriptor loads and four alternative pass bodies per sample.  Samples are dealt to the waves statically, in
//     contiguous ranges of EQUAL WEIGHT (rows + a per-sample constant), found from per-chunk weight sums a small kernel leaves in the
//     caller's workspace: no atomics, the same result bit for bit on every run.
// What stays: everything is computed transposed so that layers chain in registers (lane (kk, r) holds features {16 i + 4 kk + e} of ITS
// OWN row r straight from HBM -- the history never goes through LDS --, the MFMA result is the next layer's operand), weight images in
// MFMA A-operand order in LDS, two waves per SIMD, passes of two row tiles, loads a pass (rows) and two passes (ids) ahead.
// Rows of different samples in one tile need their own sample's candidate row (for h * a) and term c (the accumulators' start): both sit
// in a per-wave LDS slot per block and are read with per-lane addresses (rows of one tile belong to 1-3 samples: broadcasts; the slot
// strides 72 / 84 floats keep neighbouring samples on different banks).  The masked softmax and the pooling run per SEGMENT (the rows of
// one sample inside a pass; wave-uniform loop, 2.25 segments per pass at the BASELINE length mix) with the online form across passes.
// THIS FILE IS COMPILED WITHOUT PACKED fp32 VALU INSTRUCTIONS (build.py; the hazard is described in din_wave.hip).
#include <cstring>

#include "common.hpp"

namespace dir {
namespace {

typedef float dp_f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 dp_f16x8 __attribute__((ext_vector_type(8)));
typedef unsigned int dp_u32x4 __attribute__((ext_vector_type(4)));

constexpr int DP_K = 64, DP_H1P = 80, DP_H2P = 48;
constexpr int DP_WAVES = 8;            // waves per workgroup (two per SIMD), one workgroup per CU
constexpr int DP_BLK = 16;             // samples per block (the n of the per-sample term's MFMA)
constexpr int DP_AVS = 72, DP_CVS = 84;      // floats between samples in the candidate-row / per-sample-term slots
constexpr int DP_ACT_S1 = 96, DP_ACT_S2 = 48;
constexpr int DP_ACT_FLOATS = 3 * DP_ACT_S1 + 3 * DP_ACT_S2;
constexpr float DP_NLOG2E = -1.4426950408889634f;

// The weight image: what a workgroup keeps in LDS for the whole launch.  Built ONCE per weight version by din_pack_image_k into global
// memory (the first build of this kernel staged it from W1 / W2 in every workgroup's prologue: ~20 us of a 230 us launch), copied into LDS
// with 16-byte loads by every workgroup.
struct DpImg {
    unsigned int whd3[2 * 5 * 2 * 256];     // (Wh + Wd)^T: [k-step][m tile][piece][lane][4 dwords]
    unsigned int wp3[2 * 5 * 2 * 256];      // Wp^T
    unsigned int wc3[2 * 5 * 2 * 256];      // (Wa - Wd)^T
    unsigned int w23[3 * 3 * 2 * 256];      // W2^T; hidden 80..95 of the third k-step are zero
    float b1[DP_H1P], b2[DP_H2P], w3[DP_H2P];
    float act[DP_ACT_FLOATS];               // PReLU / Dice: [3][96] layer 1, [3][48] layer 2 (alpha, -log2 e scale, -log2 e shift); sigmoid: unused
};
static_assert(sizeof(DpImg) % 16 == 0, "16-byte copies");
struct DpSh {
    DpImg w;
    float av[DP_WAVES][DP_BLK * DP_AVS];    // per wave: the candidate rows of the block being computed
    float cv[DP_WAVES][DP_BLK * DP_CVS];    // per wave: their terms c (+ b1); before the sample loop: scratch of the range search
};
static_assert(sizeof(DpSh) <= 160 * 1024, "LDS");

__device__ __forceinline__ float4 dp_ld4(const float* p) { return *reinterpret_cast<const float4*>(p); }
__device__ __forceinline__ float dp_sigmoid_pre(float y) { return __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(y)); }
// ACT 0 = sigmoid (the weight images carry -log2 e: the MFMA result IS the exponent), 1 = PReLU, 2 = Dice (din_wave.hip: dw_act)
template <int ACT>
__device__ __forceinline__ float dp_act(float pre, float alpha, float nscale, float nshift) {
    if constexpr (ACT == 0) return dp_sigmoid_pre(pre);
    else if constexpr (ACT == 1) return pre > 0.f ? pre : alpha * pre;
    else {
        const float pgate = dp_sigmoid_pre(fmaf(pre, nscale, nshift));
        return pre * fmaf(pgate, 1.0f - alpha, alpha);
    }
}
template <int ACT>
__device__ __forceinline__ void dp_act_params(const float* rows, int stride, int h, float (&al)[4], float (&ns)[4], float (&nt)[4]) {
#pragma unroll
    for (int g = 0; g < 4; ++g) al[g] = ns[g] = nt[g] = 0.f;
    if constexpr (ACT != 0) {
        const float4 a = dp_ld4(rows + h);
        al[0] = a.x; al[1] = a.y; al[2] = a.z; al[3] = a.w;
    }
    if constexpr (ACT == 2) {
        const float4 b = dp_ld4(rows + stride + h), c = dp_ld4(rows + 2 * stride + h);
        ns[0] = b.x; ns[1] = b.y; ns[2] = b.z; ns[3] = b.w;
        nt[0] = c.x; nt[1] = c.y; nt[2] = c.z; nt[3] = c.w;
    }
}

// fp16 x 2 (cin_bf3.hip explains the arithmetic): an fp32 operand is the sum of two fp16 pieces by round-to-nearest, the three products
// of weight >= 2^-11 accumulate in fp32 on v_mfma_f32_16x16x32_f16
__device__ __forceinline__ unsigned int dp_pk_h(float a, float b) {
    typedef _Float16 h2_t __attribute__((ext_vector_type(2)));
    const h2_t v = {(_Float16)a, (_Float16)b};
    unsigned int w = __builtin_bit_cast(unsigned int, v);
    asm("" : "+v"(w));      // keeps the compiler from folding the split away (dense_bf3.hip: db3_pk)
    return w;
}
__device__ __forceinline__ void dp_split2(float a, float b, unsigned int& hi, unsigned int& lo) {
    hi = dp_pk_h(a, b);
#if defined(__HIP_DEVICE_COMPILE__) && !defined(DP_NO_FMA_MIX)
    // The residual a - float(fp16(a)) as ONE mixed-precision fma per element (v_fma_mix_f32 takes the fp16 half straight from the packed
    // word; exact, bit for bit the convert + subtract it replaces).  The compiler does not select it (profiles/NOTES.md R5.8: in din_wave_k
    // the asm cost more than it saved); THIS kernel is bound by VALU issue, and 583 -> 481 VALU instructions per 32-row pass of the MLP
    // are 0.208 -> 0.199 ms at config 4 (R6.10).
    float ra, rb;
    asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel_hi:[1,0,0]" : "=v"(ra) : "v"(hi), "v"(a));
    asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(rb) : "v"(hi), "v"(b));
    lo = dp_pk_h(ra, rb);
#else
    typedef _Float16 h2_t __attribute__((ext_vector_type(2)));
    const h2_t h = __builtin_bit_cast(h2_t, hi);
    lo = dp_pk_h(a - (float)h[0], b - (float)h[1]);
#endif
}
__device__ __forceinline__ void dp_split8(const float4 s0, const float4 s1, dp_f16x8 (&x)[2]) {
    unsigned int w[2][4];
    dp_split2(s0.x, s0.y, w[0][0], w[1][0]);
    dp_split2(s0.z, s0.w, w[0][1], w[1][1]);
    dp_split2(s1.x, s1.y, w[0][2], w[1][2]);
    dp_split2(s1.z, s1.w, w[0][3], w[1][3]);
#pragma unroll
    for (int pc = 0; pc < 2; ++pc) x[pc] = __builtin_bit_cast(dp_f16x8, (dp_u32x4){w[pc][0], w[pc][1], w[pc][2], w[pc][3]});
}
__device__ __forceinline__ dp_f32x4 dp_mma(const dp_f16x8 (&a)[2], const dp_f16x8 (&x)[2], dp_f32x4 c) {      // three products, smallest first
    c = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[1], x[0], c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[0], x[1], c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[0], x[0], c, 0, 0, 0);
    return c;
}
__device__ __forceinline__ void dp_lda(const unsigned int* img, int tile, int lane4, dp_f16x8 (&a)[2]) {
#pragma unroll
    for (int pc = 0; pc < 2; ++pc) a[pc] = __builtin_bit_cast(dp_f16x8, *reinterpret_cast<const dp_u32x4*>(img + (tile * 2 + pc) * 256 + lane4));
}

__device__ __forceinline__ int dp_readlane(int v, int l) { return __builtin_amdgcn_readlane(v, __builtin_amdgcn_readfirstlane(l)); }
__device__ __forceinline__ int dp_uni(int v) { return __builtin_amdgcn_readfirstlane(v); }
__device__ __forceinline__ long long dp_uni64(long long v) {
    const unsigned int lo = (unsigned int)__builtin_amdgcn_readfirstlane((int)(unsigned int)(unsigned long long)v);
    const unsigned int hi = (unsigned int)__builtin_amdgcn_readfirstlane((int)(unsigned int)((unsigned long long)v >> 32));
    return (long long)(((unsigned long long)hi << 32) | lo);
}
// inclusive prefix sum over the 16 lanes of a DPP row (row_shr:n, out-of-row sources read 0)
__device__ __forceinline__ int dp_row_scan(int v) {
    v += __builtin_amdgcn_update_dpp(0, v, 0x111, 0xF, 0xF, true);
    v += __builtin_amdgcn_update_dpp(0, v, 0x112, 0xF, 0xF, true);
    v += __builtin_amdgcn_update_dpp(0, v, 0x114, 0xF, 0xF, true);
    v += __builtin_amdgcn_update_dpp(0, v, 0x118, 0xF, 0xF, true);
    return v;
}

// rows of two tiles: this lane's 16 features {16 i + 4 kk + e} of its own rows.  The loads are UNCONDITIONAL (a lane without a row, or with a
// masked position, reads row 0: such rows are in no softmax and pooled with weight 0, so what they hold does not matter as long as it is
// finite): a load under a branch makes the compiler's count of outstanding loads imprecise, and every wait behind it becomes vmcnt(0) --
// which, issued after a pass's row loads, exposes their whole HBM latency in every pass (the first build of this kernel did exactly that).
__device__ __forceinline__ void dp_load_rows(const float* __restrict__ table, const int kk, const unsigned int r0, const unsigned int r1,
                                             float4 (&hv)[2][4]) {
    const float* const p0 = table + (size_t)r0 * DP_K + 4 * kk;
    const float* const p1 = table + (size_t)r1 * DP_K + 4 * kk;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        hv[0][i] = dp_ld4(p0 + 16 * i);
        hv[1][i] = dp_ld4(p1 + 16 * i);
    }
}

// One pass of a block: which rows it holds.  The wave-uniform members live in scalar registers; the ids the loads ahead of the pass
// fetched travel beside it (idA / idB, cidA / cidB in the kernel) so that a descriptor costs two vector registers to hand on.
constexpr int DP_MASKED = 0x40000000;      // sj: the row's history id is < 0 (a masked position inside the length): in no softmax, pooled as zeros
struct DpDesc {
    int valid;              // wave-uniform: 0 beyond the wave's last pass
    int first;              // the first pass of its block
    int ns;                 // samples in the block
    int rowbase;            // block-relative index of the pass's first row
    int nrows;              // rows of the block inside the pass (0 .. 32; 0 only for a block of empty samples)
    long long sb;           // the block's first sample
    int sj[2];              // per lane and tile: (sample in block << 16) | history position (| DP_MASKED once the id has landed); -1: no row
};

// The three layers for the NT row tiles of a pass -> the rows' scores sc (b3 included), identical in the four lane groups of a row.
template <int NT, int ACT>
__device__ __forceinline__ void dp_mlp(const DpSh& sh, const float* actl, const float* avs, const float* cvs, const int r16, const int kk,
                                       const int (&so)[2], const float4 (&hv)[2][4], const float b3, float (&sc)[NT]) {
    const int lane4 = 4 * (16 * kk + r16);          // dword offset of this lane's 16 bytes inside a 1 KB operand tile
    // ---- layer 1: pre1^T = (Wh+Wd)^T h^T + Wp^T (h*a)^T + c 1^T, the accumulators start at the row's own sample's c ---------------------
    dp_f32x4 acc1[5][NT];
#pragma unroll
    for (int mt = 0; mt < 5; ++mt)
#pragma unroll
        for (int rt = 0; rt < NT; ++rt) {
            const float4 c4 = dp_ld4(cvs + so[rt] * DP_CVS + 16 * mt + 4 * kk);
            acc1[mt][rt] = (dp_f32x4){c4.x, c4.y, c4.z, c4.w};
        }
#pragma unroll
    for (int part = 0; part < 2; ++part) {
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            dp_f16x8 xb[NT][2];
#pragma unroll
            for (int rt = 0; rt < NT; ++rt) {
                float4 s0 = hv[rt][2 * ks], s1 = hv[rt][2 * ks + 1];
                if (part) {
                    const float4 a0 = dp_ld4(avs + so[rt] * DP_AVS + 16 * (2 * ks) + 4 * kk);
                    const float4 a1 = dp_ld4(avs + so[rt] * DP_AVS + 16 * (2 * ks + 1) + 4 * kk);
                    s0 = make_float4(s0.x * a0.x, s0.y * a0.y, s0.z * a0.z, s0.w * a0.w);
                    s1 = make_float4(s1.x * a1.x, s1.y * a1.y, s1.z * a1.z, s1.w * a1.w);
                }
                dp_split8(s0, s1, xb[rt]);
            }
            const unsigned int* img = part ? sh.w.wp3 : sh.w.whd3;
#pragma unroll
            for (int mt = 0; mt < 5; ++mt) {
                dp_f16x8 a[2];
                dp_lda(img, ks * 5 + mt, lane4, a);
#pragma unroll
                for (int rt = 0; rt < NT; ++rt) acc1[mt][rt] = dp_mma(a, xb[rt], acc1[mt][rt]);
            }
        }
    }
    // z1 = act(pre1) in place: hidden 16 mt + 4 kk + g of row r -- element (mt & 1) * 4 + g of layer 2's k-step mt >> 1
#pragma unroll
    for (int mt = 0; mt < 5; ++mt) {
        float al[4], ns[4], nt[4];
        dp_act_params<ACT>(actl, DP_ACT_S1, 16 * mt + 4 * kk, al, ns, nt);
#pragma unroll
        for (int rt = 0; rt < NT; ++rt)
#pragma unroll
            for (int g = 0; g < 4; ++g) acc1[mt][rt][g] = dp_act<ACT>(acc1[mt][rt][g], al[g], ns[g], nt[g]);
    }
    // ---- layer 2: pre2^T, three k-steps (hidden 80..95 are zeros on both sides) -------------------------------------------------------------
    dp_f32x4 acc2[3][NT];
#pragma unroll
    for (int m2 = 0; m2 < 3; ++m2) {
        const float4 c4 = dp_ld4(&sh.w.b2[16 * m2 + 4 * kk]);
#pragma unroll
        for (int rt = 0; rt < NT; ++rt) acc2[m2][rt] = (dp_f32x4){c4.x, c4.y, c4.z, c4.w};
    }
#pragma unroll
    for (int ks = 0; ks < 3; ++ks) {
        dp_f16x8 xb[NT][2];
#pragma unroll
        for (int rt = 0; rt < NT; ++rt) {
            const dp_f32x4 z0 = acc1[2 * ks][rt];
            const dp_f32x4 z1 = 2 * ks + 1 < 5 ? acc1[2 * ks + 1 < 5 ? 2 * ks + 1 : 0][rt] : (dp_f32x4){0.f, 0.f, 0.f, 0.f};
            dp_split8(make_float4(z0[0], z0[1], z0[2], z0[3]), make_float4(z1[0], z1[1], z1[2], z1[3]), xb[rt]);
        }
#pragma unroll
        for (int m2 = 0; m2 < 3; ++m2) {
            dp_f16x8 a[2];
            dp_lda(sh.w.w23, ks * 3 + m2, lane4, a);
#pragma unroll
            for (int rt = 0; rt < NT; ++rt) acc2[m2][rt] = dp_mma(a, xb[rt], acc2[m2][rt]);
        }
    }
    // ---- layer 3 -----------------------------------------------------------------------------------------------------------------------------
    float wv[3][4];
#pragma unroll
    for (int m2 = 0; m2 < 3; ++m2) {
        const float4 t4 = dp_ld4(&sh.w.w3[16 * m2 + 4 * kk]);
        wv[m2][0] = t4.x; wv[m2][1] = t4.y; wv[m2][2] = t4.z; wv[m2][3] = t4.w;
    }
    float sp[NT];
#pragma unroll
    for (int rt = 0; rt < NT; ++rt) {
        sp[rt] = 0.f;
#pragma unroll
        for (int m2 = 0; m2 < 3; ++m2) {
            float al[4], ns[4], nt[4];
            dp_act_params<ACT>(actl + 3 * DP_ACT_S1, DP_ACT_S2, 16 * m2 + 4 * kk, al, ns, nt);
#pragma unroll
            for (int g = 0; g < 4; ++g) sp[rt] = fmaf(dp_act<ACT>(acc2[m2][rt][g], al[g], ns[g], nt[g]), wv[m2][g], sp[rt]);
        }
    }
    // The four lane groups hold the four quarters of a row's H2 sum.  The sums over the groups of BOTH tiles inside the VALU (cin_bf3.hip:
    // store_dot): v_permlane16_swap exchanges a's odd rows of 16 lanes with b's even rows, v_permlane32_swap a's upper half with b's lower
    // half -- no ds_bpermute round trips at the end of the pass's matrix work.  (g0 + g1) + (g2 + g3), in every lane group.
    float a = sp[0], b = sp[NT - 1];
    asm volatile("s_nop 1\n\tv_permlane16_swap_b32 %0, %1\n\ts_nop 1" : "+v"(a), "+v"(b));
    float c = a + b, d = c;          // rows 0 / 2: tile 0's pair sums, rows 1 / 3: the last tile's
    asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1\n\ts_nop 1" : "+v"(c), "+v"(d));
    float t0 = c + d, t1 = t0;       // rows 0 / 2: tile 0's total, rows 1 / 3: the last tile's
    if constexpr (NT == 2) {
        asm volatile("s_nop 1\n\tv_permlane16_swap_b32 %0, %1\n\ts_nop 1" : "+v"(t0), "+v"(t1));      // t0: tile 0's total everywhere, t1: tile 1's
        sc[0] = t0 + b3;
        sc[1] = t1 + b3;
    } else {
        sc[0] = t0 + b3;
    }
}

// per-chunk weight sums: weight of a sample = its valid history positions + w0 (what a sample costs beyond its rows).  One wave per chunk.
__global__ __launch_bounds__(256) void din_pack_sums_k(const int32_t* __restrict__ hist_len, int T, long long B, long long chunk, int nchunk, int w0,
                                                       int* __restrict__ csum) {
    const int c = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (c >= nchunk) return;
    const long long base = (long long)c * chunk;
    const long long endb = base + chunk < B ? base + chunk : B;
    int s = 0;
    for (long long b = base + lane; b < endb; b += 64) {
        const int len = hist_len ? hist_len[b] : T;
        s += (len < 0 ? 0 : len > T ? T : len) + w0;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
    if (lane == 0) csum[c] = s;
}

// The weight image (see DpImg) from W1 [4K, H1], b1, W2 [H1, H2], b2, W3 and the PReLU / Dice parameters.  A-operand images: dword jp of lane
// l of tile (ks, mt) holds elements j = 2 jp, 2 jp + 1 = W[m = 16 mt + (l & 15)][k], k + 1, k = 16 (2 ks + (j >> 2)) + 4 (l >> 4) + (j & 3).
template <int ACT>
__global__ __launch_bounds__(256) void din_pack_image_k(const float* __restrict__ W1, const float* __restrict__ b1, int H1,
                                                        const float* __restrict__ W2, const float* __restrict__ b2, int H2,
                                                        const float* __restrict__ W3, const float* __restrict__ act_params, DpImg* __restrict__ img) {
    constexpr float SC = ACT == 0 ? DP_NLOG2E : 1.0f;      // sigmoid: the MFMA result IS the exponent of 2
    const int gtid = blockIdx.x * 256 + threadIdx.x, gsz = gridDim.x * 256;
    for (int idx = gtid; idx < 2 * 5 * 64 * 4; idx += gsz) {
        const int jp = idx & 3, l = (idx >> 2) & 63, t = idx >> 8;
        const int ks = t / 5, mt = t - 5 * ks;
        const int m = 16 * mt + (l & 15);
        const int f = 16 * (2 * ks + (jp >> 1)) + 4 * (l >> 4) + 2 * (jp & 1);
        float hd[2] = {0.f, 0.f}, pp[2] = {0.f, 0.f}, cc[2] = {0.f, 0.f};
        if (m < H1) {
#pragma unroll
            for (int e = 0; e < 2; ++e) {
                const float vh = W1[(size_t)(f + e) * H1 + m], va = W1[(size_t)(DP_K + f + e) * H1 + m];
                const float vd = W1[(size_t)(2 * DP_K + f + e) * H1 + m], vp = W1[(size_t)(3 * DP_K + f + e) * H1 + m];
                hd[e] = (vh + vd) * SC;
                pp[e] = vp * SC;
                cc[e] = (va - vd) * SC;
            }
        }
        unsigned int hi, lo;
        dp_split2(hd[0], hd[1], hi, lo);
        img->whd3[(t * 2 + 0) * 256 + l * 4 + jp] = hi;
        img->whd3[(t * 2 + 1) * 256 + l * 4 + jp] = lo;
        dp_split2(pp[0], pp[1], hi, lo);
        img->wp3[(t * 2 + 0) * 256 + l * 4 + jp] = hi;
        img->wp3[(t * 2 + 1) * 256 + l * 4 + jp] = lo;
        dp_split2(cc[0], cc[1], hi, lo);
        img->wc3[(t * 2 + 0) * 256 + l * 4 + jp] = hi;
        img->wc3[(t * 2 + 1) * 256 + l * 4 + jp] = lo;
    }
    for (int idx = gtid; idx < 3 * 3 * 64 * 4; idx += gsz) {
        const int jp = idx & 3, l = (idx >> 2) & 63, t = idx >> 8;
        const int ks = t / 3, m2 = t - 3 * ks;
        const int h2 = 16 * m2 + (l & 15);
        const int hid = 16 * (2 * ks + (jp >> 1)) + 4 * (l >> 4) + 2 * (jp & 1);
        float v[2];
#pragma unroll
        for (int e = 0; e < 2; ++e)      // a padded hidden unit is sigmoid(0) = 0.5: its weights are zero
            v[e] = (hid + e < H1 && h2 < H2) ? W2[(size_t)(hid + e) * H2 + h2] * SC : 0.f;
        unsigned int hi, lo;
        dp_split2(v[0], v[1], hi, lo);
        img->w23[(t * 2 + 0) * 256 + l * 4 + jp] = hi;
        img->w23[(t * 2 + 1) * 256 + l * 4 + jp] = lo;
    }
    for (int idx = gtid; idx < DP_H1P; idx += gsz) img->b1[idx] = idx < H1 ? b1[idx] * SC : 0.f;
    for (int idx = gtid; idx < DP_H2P; idx += gsz) {
        img->b2[idx] = idx < H2 ? b2[idx] * SC : 0.f;
        img->w3[idx] = idx < H2 ? W3[idx] : 0.f;
    }
    for (int idx = gtid; idx < DP_ACT_FLOATS; idx += gsz) {
        float v = 0.f;
        if constexpr (ACT != 0) {
            if (idx < 3 * DP_ACT_S1) {
                const int row = idx / DP_ACT_S1, h = idx - row * DP_ACT_S1;
                if (h < H1) v = act_params[row * H1 + h] * (row == 0 ? 1.0f : DP_NLOG2E);
            } else {
                const int j = idx - 3 * DP_ACT_S1, row = j / DP_ACT_S2, h = j - row * DP_ACT_S2;
                if (h < H2) v = act_params[3 * H1 + row * H2 + h] * (row == 0 ? 1.0f : DP_NLOG2E);
            }
        }
        img->act[idx] = v;
    }
}

// scores written raw by the main kernel -> attention weights: softmax weights exp(x - m_b) / l_b (normalize), zeros beyond the length
__global__ __launch_bounds__(256) void din_pack_scores_k(float* __restrict__ scores, const int32_t* __restrict__ hist_len, int T, long long B,
                                                         int normalize, const float* __restrict__ ml) {
    const long long n = B * T;
    for (long long e = (long long)blockIdx.x * 256 + threadIdx.x; e < n; e += (long long)gridDim.x * 256) {
        const long long b = e / T;
        const int j = (int)(e - b * T);
        const int len = hist_len ? hist_len[b] : T;
        float v = 0.f;
        if (j < len) {
            const float x = scores[e];
            v = normalize ? (x > -INFINITY ? __builtin_amdgcn_exp2f(x - ml[2 * b]) * ml[2 * b + 1] : 0.f) : x;      // x, m: log2 domain
        }
        scores[e] = v;
    }
}

template <int ACT, bool SCORES, bool NORM>
__global__ __launch_bounds__(64 * DP_WAVES, 2) void din_pack_k(const float* __restrict__ table, const int64_t* __restrict__ hist,
                                                               const int32_t* __restrict__ hist_len, const int64_t* __restrict__ cand, int T,
                                                               const DpImg* __restrict__ img, const float* __restrict__ b3,
                                                               long long B, float* __restrict__ out, float* __restrict__ scores,
                                                               float* __restrict__ ml, const int* __restrict__ csum, int nchunk,
                                                               long long chunk, int w0) {
    extern __shared__ __attribute__((aligned(16))) unsigned char dp_smem[];
    DpSh& sh = *reinterpret_cast<DpSh*>(dp_smem);
    const float* const actl = sh.w.act;
    const int tid = threadIdx.x, lane = tid & 63, r16 = lane & 15, kk = lane >> 4;
    const int w = dp_uni(tid >> 6);
    // ---- the chunk sums -> their exclusive prefix (the range search's first level), in the cv slots ------------------------------------------
    long long* const cpre = reinterpret_cast<long long*>(&sh.cv[0][0]);      // [nchunk + 1], nchunk <= 1024
    for (int idx = tid; idx < nchunk; idx += 64 * DP_WAVES) cpre[idx + 1] = csum[idx];
    if (tid == 0) cpre[0] = 0;
    // ---- the weight image: global -> LDS, 16 bytes per thread and trip -----------------------------------------------------------------------
    {
        const dp_u32x4* const src = reinterpret_cast<const dp_u32x4*>(img);
        dp_u32x4* const dst = reinterpret_cast<dp_u32x4*>(&sh.w);
        constexpr int N16 = (int)(sizeof(DpImg) / 16);
#pragma unroll 4
        for (int idx = tid; idx < N16; idx += 64 * DP_WAVES) dst[idx] = src[idx];
    }
    __syncthreads();
    if (w == 0) {                // inclusive scan of cpre[1 ..] in place: per = entries per lane, then the lane totals
        const int per = (nchunk + 63) >> 6;
        long long run = 0;
        for (int i = 0; i < per; ++i) {
            const int c = lane * per + i;
            if (c < nchunk) {
                run += cpre[c + 1];
                cpre[c + 1] = run;
            }
        }
        long long off = run;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const long long t = __shfl_up(off, o, 64);
            if (lane >= o) off += t;
        }
        off -= run;              // exclusive prefix of the lane totals
        for (int i = 0; i < per; ++i) {
            const int c = lane * per + i;
            if (c < nchunk) cpre[c + 1] += off;
        }
    }
    __syncthreads();
    // ---- this wave's samples [s_lo, s_hi): wave g of G takes the samples whose exclusive weight prefix P(b) lies in [ceil(g W / G), ceil((g + 1) W / G)) ----
    auto len_of = [&](const long long b) {
        const int len = hist_len ? hist_len[b] : T;
        return len < 0 ? 0 : len > T ? T : len;
    };
    auto first_at = [&](const long long t) -> long long {      // min b with P(b) >= t (B when there is none); wave-uniform
        int lo_c = 0, hi_c = nchunk;                          // the last chunk whose start prefix is < t (chunk 0 when t == 0)
        while (hi_c - lo_c > 1) {
            const int mid = (lo_c + hi_c) >> 1;
            if (cpre[mid] < t) lo_c = mid; else hi_c = mid;
        }
        const long long base = (long long)lo_c * chunk, endb = base + chunk < B ? base + chunk : B;
        const int tt = (int)(t - cpre[lo_c]);                  // samples of the chunk with a chunk-relative prefix < tt come before the answer
        int carry = 0, cnt = 0;                                // (a chunk's weight fits 31 bits: din_pack_chunks)
        for (long long sub = base; sub < endb && carry < tt; sub += 64) {      // 64 consecutive samples per trip, one per lane
            const long long b = sub + lane;
            const int wgt = b < endb ? len_of(b) + w0 : 0;
            int incl = wgt;
#pragma unroll
            for (int o = 1; o < 64; o <<= 1) {
                const int u = __shfl_up(incl, o, 64);
                if (lane >= o) incl += u;
            }
            cnt += (b < endb && carry + incl - wgt < tt) ? 1 : 0;
            carry += __shfl(incl, 63, 64);
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) cnt += __shfl_xor(cnt, o, 64);
        return dp_uni64(base + cnt);
    };
    const long long Wtot = cpre[nchunk];
    const long long G = (long long)gridDim.x * DP_WAVES, g = (long long)blockIdx.x * DP_WAVES + w;
    const long long s_lo = first_at((g * Wtot + G - 1) / G);
    const long long s_hi = g + 1 == G ? B : first_at(((g + 1) * Wtot + G - 1) / G);
    __syncthreads();             // the cv slots are scratch no longer
    const float bias3 = b3[0];
    const float inv_sqrt_k = 1.0f / sqrtf((float)DP_K);
    float* const avs = sh.av[w];
    float* const cvs = sh.cv[w];
    const int lane4 = 4 * (16 * kk + r16);

    // ---- the cursor two passes ahead of the computation: which rows a pass holds, their history ids ------------------------------------------
    // Every load below is unconditional (clamped addresses, the result selected afterwards): see dp_load_rows.
    long long a_sb = 0;
    int a_ns = 0, a_np = 0, a_pass = -1, a_R = 0, a_cur = 0, a_endv = 0;
    bool a_started = false, a_done = s_lo >= s_hi;
    // lane r's length in the block at sb: the load (raw, from a clamped address) now, the clamp when it is consumed -- a value that is
    // touched right behind its load makes the wave wait for that load and for everything issued before it (the pass's row loads)
    auto block_len_raw = [&](const long long sb) {
        const long long b = sb + r16;
        return hist_len ? hist_len[b < B ? b : B - 1] : T;
    };
    auto block_len_of = [&](const long long sb, const int raw) {      // 0 beyond the wave's range
        return sb + r16 < s_hi ? (raw < 0 ? 0 : raw > T ? T : raw) : 0;
    };
    int a_lens_next = block_len_raw(s_lo);
    auto advance = [&](DpDesc& d, long long (&id)[2], long long& cid) __attribute__((always_inline)) {
        d.valid = 0;
        d.first = 0;
        d.sj[0] = d.sj[1] = -1;
        d.ns = 0; d.rowbase = 0; d.nrows = 0; d.sb = 0;
        bool live = !a_done;
        if (live) {
            ++a_pass;
            if (a_pass >= a_np) {                              // the next block
                const long long nb = a_started ? a_sb + DP_BLK : s_lo;
                a_started = true;
                if (nb >= s_hi) {
                    a_done = true;
                    live = false;
                } else {
                    a_sb = nb;
                    a_ns = (int)(s_hi - nb < DP_BLK ? s_hi - nb : DP_BLK);
                    a_endv = dp_row_scan(block_len_of(nb, a_lens_next));
                    a_R = dp_readlane(a_endv, 15);
                    a_np = a_R > 0 ? (a_R + 31) >> 5 : 1;
                    a_pass = 0;
                    a_cur = 0;
                    d.first = 1;
                }
            }
        }
        unsigned int off0 = 0, off1 = 0;
        if (live) {
            d.valid = 1;
            d.sb = a_sb;
            d.ns = a_ns;
            d.rowbase = a_pass * 32;
            d.nrows = a_R - d.rowbase > 32 ? 32 : a_R - d.rowbase;
            // row q belongs to the first sample whose end (inclusive prefix) is > q: samples before a_cur ended before this pass
            const int q0 = d.rowbase + r16, q1 = q0 + 16, qlast = d.rowbase + 31;
            const int base = a_cur ? dp_readlane(a_endv, a_cur - 1) : 0;
            int s0 = a_cur, s1 = a_cur, p0 = base, p1 = base;
            int i = a_cur;
            while (i < a_ns) {
                const int e = dp_readlane(a_endv, i);
                if (e > qlast) break;
                s0 += q0 >= e ? 1 : 0;
                p0 = q0 >= e ? e : p0;
                s1 += q1 >= e ? 1 : 0;
                p1 = q1 >= e ? e : p1;
                ++i;
            }
            a_cur = i;
            if (q0 < a_R) {
                d.sj[0] = (s0 << 16) | (q0 - p0);
                off0 = (unsigned int)(s0 * T + (q0 - p0));
            }
            if (q1 < a_R) {
                d.sj[1] = (s1 << 16) | (q1 - p1);
                off1 = (unsigned int)(s1 * T + (q1 - p1));
            }
        }
        // the loads of the step, all of them every time: the two history ids (entry 0 of the block for a lane without a row), the
        // candidate id of lane r's sample and the NEXT block's length of lane r's sample (consumed when the cursor enters a block)
        const int64_t* const hb = hist + a_sb * T;              // wave-uniform base: the loads take a 32-bit per-lane offset
        id[0] = hb[off0];
        id[1] = hb[off1];
        const long long bc = a_sb + r16;
        cid = cand[bc < B ? bc : B - 1];
        a_lens_next = block_len_raw(a_sb + DP_BLK);
    };

    // ---- state of the computation: the block's sample ends, the first unfinished sample, the open sample's online softmax --------------------
    // NORM: scores and running maximum in the log2 domain (x log2 e / sqrt K), so that a weight is one v_exp
    int c_endv = 0, c_cur = 0, c_lens = 0, c_R = 0;
    long long c_sb_f = 0;
    bool c_has = false;            // lane r's sample of the block being fetched has a candidate row (id >= 0): otherwise a = 0
    float m_run = -INFINITY, l_run = 0.f;
    float o[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) o[i] = 0.f;
    float4 hvA[2][4], hvB[2][4], an[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) an[i] = hvA[0][i] = hvA[1][i] = hvB[0][i] = hvB[1][i] = make_float4(0.f, 0.f, 0.f, 0.f);
    const float xscale = NORM ? inv_sqrt_k * 1.4426950408889634f : 1.0f;
    const bool hi8 = (r16 & 8) != 0, hi4 = (r16 & 4) != 0, hi2 = (r16 & 2) != 0, hi1 = (r16 & 1) != 0;

    auto fetch = [&](DpDesc& d, const long long (&id)[2], const long long cid, float4 (&h)[2][4]) __attribute__((always_inline)) {
        // a pass's rows; with a block's first pass its lengths and candidate rows.  (Row ids fit 32 bits: 2^31 rows of 256 bytes are 512 GB.)
        const bool neg0 = id[0] < 0, neg1 = id[1] < 0;
        dp_load_rows(table, kk, neg0 ? 0u : (unsigned int)id[0], neg1 ? 0u : (unsigned int)id[1], h);
        if (d.sj[0] >= 0 && neg0) d.sj[0] |= DP_MASKED;
        if (d.sj[1] >= 0 && neg1) d.sj[1] |= DP_MASKED;
        if (d.first) {          // (the only loads under a branch: the LAST ones a step issues, and nothing waits before the next step's start)
            const bool has = r16 < d.ns && cid >= 0;
            c_lens = block_len_raw(d.sb);       // (raw: clamped at the block switch)
            c_sb_f = d.sb;
            const float* const pa = table + (size_t)(has ? (unsigned int)cid : 0u) * DP_K + 4 * kk;
#pragma unroll
            for (int i = 0; i < 4; ++i) an[i] = dp_ld4(pa + 16 * i);
            c_has = has;
        }
    };
    // The open sample is complete: its pooled output.  The 16 sums over the 16 rows of a lane group by a transposing butterfly (each step
    // halves the values a lane keeps: 15 x (2 selects + 1 DPP add) instead of 16 x 4 DPP adds); lane r ends with value r = 4 i + e, feature
    // 16 i + 4 kk + e.
    auto finalize = [&](const long long b) __attribute__((always_inline)) {
        const float inv_l = NORM ? (l_run > 0.f ? __builtin_amdgcn_rcpf(l_run) : 0.f) : 1.0f;
        float n8[8], n4[4], n2[2];
#pragma unroll
        for (int j = 0; j < 8; ++j) n8[j] = (hi8 ? o[8 + j] : o[j]) + dpp_mov<0x140>(hi8 ? o[j] : o[8 + j]);          // row_mirror: r <-> 15 - r
#pragma unroll
        for (int j = 0; j < 4; ++j) n4[j] = (hi4 ? n8[4 + j] : n8[j]) + dpp_mov<0x141>(hi4 ? n8[j] : n8[4 + j]);      // row_half_mirror: r <-> 7 - r
#pragma unroll
        for (int j = 0; j < 2; ++j) n2[j] = (hi2 ? n4[2 + j] : n4[j]) + dpp_mov<0x4E>(hi2 ? n4[j] : n4[2 + j]);       // r <-> r ^ 2
        const float tot = (hi1 ? n2[1] : n2[0]) + dpp_mov<0xB1>(hi1 ? n2[0] : n2[1]);                                   // r <-> r ^ 1
        out[b * DP_K + 16 * (r16 >> 2) + 4 * kk + (r16 & 3)] = tot * inv_l;
        if (SCORES && NORM && lane == 0) {
            ml[2 * b] = m_run;
            ml[2 * b + 1] = inv_l;
        }
        m_run = -INFINITY;
        l_run = 0.f;
        if (!NORM) {
#pragma unroll
            for (int i = 0; i < 16; ++i) o[i] = 0.f;        // (NORM: the next sample's first rescale factor exp2(-inf) = 0 wipes them)
        }
    };

    DpDesc dC, dB, dA;
    long long idB[2], idA[2], cidB = -1, cidA = -1;
    advance(dC, idB, cidB);
    fetch(dC, idB, cidB, hvA);
    advance(dB, idB, cidB);

    auto step = [&](float4 (&hv)[2][4], float4 (&hvn)[2][4]) __attribute__((always_inline)) {
        // Everything outstanding was issued at least a pass ago: one wait here, and no load this step issues is consumed before the next
        // step's wait -- the computation below never waits for memory.
        __builtin_amdgcn_s_waitcnt(0x0F70);      // vmcnt(0)
        if (dC.first) {
            if (!c_has) {            // (a sample without a candidate row, or a lane beyond the block's samples: a = 0)
#pragma unroll
                for (int i = 0; i < 4; ++i) an[i] = make_float4(0.f, 0.f, 0.f, 0.f);
            }
            // ---- a new block: its candidate rows -> the av slot; c[m] = sum_f a[f] (Wa - Wd)[f][m] + b1[m] for the 16 samples at once -> the cv slot --
            c_endv = dp_row_scan(block_len_of(c_sb_f, c_lens));
            c_R = dp_readlane(c_endv, 15);
            c_cur = 0;
#pragma unroll
            for (int i = 0; i < 4; ++i) *reinterpret_cast<float4*>(avs + r16 * DP_AVS + 16 * i + 4 * kk) = an[i];
            dp_f32x4 accc[5];
#pragma unroll
            for (int mt = 0; mt < 5; ++mt) {
                const float4 c4 = dp_ld4(&sh.w.b1[16 * mt + 4 * kk]);
                accc[mt] = (dp_f32x4){c4.x, c4.y, c4.z, c4.w};
            }
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                dp_f16x8 xb[2];
                dp_split8(an[2 * ks], an[2 * ks + 1], xb);
#pragma unroll
                for (int mt = 0; mt < 5; ++mt) {
                    dp_f16x8 a[2];
                    dp_lda(sh.w.wc3, ks * 5 + mt, lane4, a);
                    accc[mt] = dp_mma(a, xb, accc[mt]);
                }
            }
#pragma unroll
            for (int mt = 0; mt < 5; ++mt)
                *reinterpret_cast<float4*>(cvs + r16 * DP_CVS + 16 * mt + 4 * kk) = make_float4(accc[mt][0], accc[mt][1], accc[mt][2], accc[mt][3]);
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");   // same-wave LDS hand-off: the DS queue is in order, the fences
            __builtin_amdgcn_wave_barrier();                         // only keep the compiler from moving the reads above the writes
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        }
        fetch(dB, idB, cidB, hvn);          // the next pass's rows (its ids were issued a pass ago)
        advance(dA, idA, cidA);             // the ids of the pass after it
        // ---- this pass: the three layers for its rows ---------------------------------------------------------------------------------------
        const int srow[2] = {dC.sj[0] < 0 ? -1 : (dC.sj[0] >> 16) & 0xff, dC.sj[1] < 0 ? -1 : (dC.sj[1] >> 16) & 0xff};      // the row's sample, -1: no row
        const int smem[2] = {(dC.sj[0] & DP_MASKED) ? -1 : srow[0], (dC.sj[1] & DP_MASKED) ? -1 : srow[1]};                  // ... -1 also for a masked position
        const int so[2] = {srow[0] < 0 ? 0 : srow[0], srow[1] < 0 ? 0 : srow[1]};
        float sc[2] = {0.f, 0.f};
        // the wave inside its matrix work outranks its SIMD partner (whose scalar-heavy bookkeeping fills the gaps): -2.6 % at config 4
        __builtin_amdgcn_s_setprio(1);
        if (dC.nrows > 16) {
            dp_mlp<2, ACT>(sh, actl, avs, cvs, r16, kk, so, hv, bias3, sc);
        } else if (dC.nrows > 0) {
            float s1_[1];
            dp_mlp<1, ACT>(sh, actl, avs, cvs, r16, kk, so, hv, bias3, s1_);
            sc[0] = s1_[0];
        }
        __builtin_amdgcn_s_setprio(0);
        sc[0] *= xscale;
        sc[1] *= xscale;
        // ---- its segments: masked (online) softmax + pooling per sample; a sample that ends inside the pass is written out ------------------------
        {
            const int pend = dC.rowbase + 32;
            const bool last = pend >= c_R;               // the block's last pass: every sample still open or not begun (an empty one) ends here
            int i = c_cur;
            while (i < dC.ns) {
                const int st = i ? dp_readlane(c_endv, i - 1) : 0;
                if (st >= pend && !last) break;
                const int en = dp_readlane(c_endv, i);
                if (en > st && en > dC.rowbase) {
                    const bool mem0 = smem[0] == i, mem1 = smem[1] == i;
                    float w0_, w1_;                      // the two rows' pooling weights
                    if constexpr (NORM) {
                        const float x0 = mem0 ? sc[0] : -INFINITY, x1 = mem1 ? sc[1] : -INFINITY;
                        const float mx = fmaxf(m_run, row16_max(fmaxf(x0, x1)));
                        const float mxs = mx > -INFINITY ? mx : 0.f;           // (nothing to pool yet: every factor below is exp2(-inf) = 0)
                        const float rescale = __builtin_amdgcn_exp2f(m_run - mxs);
                        w0_ = __builtin_amdgcn_exp2f(x0 - mxs);
                        w1_ = __builtin_amdgcn_exp2f(x1 - mxs);
                        l_run = l_run * rescale + row16_sum(w0_ + w1_);
#pragma unroll
                        for (int q = 0; q < 16; ++q) o[q] *= rescale;
                        m_run = mx;
                        if (SCORES && kk == 0) {
                            const long long row = (dC.sb + i) * T;
                            if (srow[0] == i) scores[row + (dC.sj[0] & 0xffff)] = x0;
                            if (srow[1] == i) scores[row + (dC.sj[1] & 0xffff)] = x1;
                        }
                    } else {
                        w0_ = mem0 ? sc[0] : 0.f;
                        w1_ = mem1 ? sc[1] : 0.f;
                        if (SCORES && kk == 0) {
                            const long long row = (dC.sb + i) * T;
                            if (srow[0] == i) scores[row + (dC.sj[0] & 0xffff)] = w0_;
                            if (srow[1] == i) scores[row + (dC.sj[1] & 0xffff)] = w1_;
                        }
                    }
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        o[4 * q + 0] = fmaf(w0_, hv[0][q].x, o[4 * q + 0]); o[4 * q + 1] = fmaf(w0_, hv[0][q].y, o[4 * q + 1]);
                        o[4 * q + 2] = fmaf(w0_, hv[0][q].z, o[4 * q + 2]); o[4 * q + 3] = fmaf(w0_, hv[0][q].w, o[4 * q + 3]);
                        o[4 * q + 0] = fmaf(w1_, hv[1][q].x, o[4 * q + 0]); o[4 * q + 1] = fmaf(w1_, hv[1][q].y, o[4 * q + 1]);
                        o[4 * q + 2] = fmaf(w1_, hv[1][q].z, o[4 * q + 2]); o[4 * q + 3] = fmaf(w1_, hv[1][q].w, o[4 * q + 3]);
                    }
                }
                if (en > pend) break;                    // the sample goes on in the next pass
                finalize(dC.sb + i);
                ++i;
            }
            c_cur = i;
        }
        dC = dB;
        dB = dA;
        idB[0] = idA[0]; idB[1] = idA[1];
        cidB = cidA;
    };
    while (true) {               // two passes per trip: the row buffers swap roles (no copies)
        if (!dC.valid) break;
        step(hvA, hvB);
        if (!dC.valid) break;
        step(hvB, hvA);
    }
}

}  // namespace

bool din_pack_covers(int K, int T, int H1, int H2) {
    return K == DP_K && T >= 1 && T <= 65535 && H1 > 0 && H2 > 0 && H1 <= DP_H1P && H2 <= DP_H2P && !(H1 & 3) && !(H2 & 3);
}

static void din_pack_chunks(int64_t B, int64_t& chunk, int& nchunk) {
    chunk = 64;                  // the range search scans one chunk per boundary: small chunks, at most 1024 of them
    while ((B + chunk - 1) / chunk > 1024) chunk *= 2;
    nchunk = (int)((B + chunk - 1) / chunk);
    if (nchunk < 1) nchunk = 1;
}
static int64_t round256(int64_t n) { return (n + 255) & ~(int64_t)255; }

int64_t din_pack_image_bytes() { return round256((int64_t)sizeof(DpImg)); }

// workspace: [chunk sums | (scores: the samples' softmax maxima / sums) | a weight image for callers that pass none]
int64_t din_pack_workspace_bytes(int64_t B, int want_scores) {
    int64_t chunk;
    int nchunk;
    din_pack_chunks(B < 1 ? 1 : B, chunk, nchunk);
    return round256((int64_t)nchunk * 4) + (want_scores ? round256(B * 8) : 0) + din_pack_image_bytes();
}

int launch_din_pack_image(hipStream_t st, const float* W1, const float* b1, int H1, const float* W2, const float* b2, int H2, const float* W3,
                          int activation, const float* act_params, void* image) {
    if (activation < 0 || activation > 2 || (activation != 0 && !act_params))
        return fail(DIR_E_UNSUPPORTED, "din_pack_image_k: activation %d (0 sigmoid, 1 PReLU, 2 Dice)", activation);
    DpImg* img = static_cast<DpImg*>(image);
    const dim3 grid(11), block(256);
    if (activation == 0) hipLaunchKernelGGL(din_pack_image_k<0>, grid, block, 0, st, W1, b1, H1, W2, b2, H2, W3, act_params, img);
    else if (activation == 1) hipLaunchKernelGGL(din_pack_image_k<1>, grid, block, 0, st, W1, b1, H1, W2, b2, H2, W3, act_params, img);
    else hipLaunchKernelGGL(din_pack_image_k<2>, grid, block, 0, st, W1, b1, H1, W2, b2, H2, W3, act_params, img);
    return DIR_OK;
}

int launch_din_pack(hipStream_t st, const float* table, const int64_t* hist, const int32_t* hist_len, const int64_t* cand, int T,
                    const void* image, const float* b3, int normalize, int64_t B, float* out, float* scores, int activation, void* workspace) {
    int64_t chunk;
    int nchunk;
    din_pack_chunks(B, chunk, nchunk);
    int* csum = static_cast<int*>(workspace);
    float* ml = reinterpret_cast<float*>(static_cast<unsigned char*>(workspace) + round256((int64_t)nchunk * 4));
    // the weight of a sample beyond its rows, in rows: what its candidate row, its share of a block's term and its output cost
    // (DIR_DIN_PACK_W0: an A/B switch read per call like DIR_DIN_ARITH; any value gives the same results up to the summation order across passes)
    const char* w0e = getenv("DIR_DIN_PACK_W0");
    const int w0 = w0e ? (atoi(w0e) < 0 ? 0 : atoi(w0e) > 64 ? 64 : atoi(w0e)) : 5;
    typedef void (*kern_t)(const float*, const int64_t*, const int32_t*, const int64_t*, int, const DpImg*, const float*, long long, float*, float*,
                           float*, const int*, int, long long, int);
    static const kern_t kerns[3][2][2] = {{{&din_pack_k<0, false, false>, &din_pack_k<0, false, true>}, {&din_pack_k<0, true, false>, &din_pack_k<0, true, true>}},
                                          {{&din_pack_k<1, false, false>, &din_pack_k<1, false, true>}, {&din_pack_k<1, true, false>, &din_pack_k<1, true, true>}},
                                          {{&din_pack_k<2, false, false>, &din_pack_k<2, false, true>}, {&din_pack_k<2, true, false>, &din_pack_k<2, true, true>}}};
    if (activation < 0 || activation > 2) return fail(DIR_E_UNSUPPORTED, "din_pack_k: activation %d (0 sigmoid, 1 PReLU, 2 Dice)", activation);
    static LdsOnce once[3][2][2];
    const int sco = scores ? 1 : 0, nrm = normalize ? 1 : 0;
    const size_t shmem = sizeof(DpSh);
    const kern_t kern = kerns[activation][sco][nrm];
    if (!lds_limit(once[activation][sco][nrm], (int)shmem, kern)) return fail(DIR_E_HIP, "din_pack_k: cannot reserve %zu B of LDS", shmem);
    hipLaunchKernelGGL(din_pack_sums_k, dim3((unsigned)((nchunk + 3) / 4)), dim3(256), 0, st, hist_len, T, (long long)B, (long long)chunk, nchunk, w0, csum);
    const int64_t waves_wanted = (B + 1) / 2;            // a wave should see at least a couple of samples
    int64_t nwg = (waves_wanted + DP_WAVES - 1) / DP_WAVES;
    if (nwg > kCUs) nwg = kCUs;
    if (nwg < 1) nwg = 1;
    hipLaunchKernelGGL(kern, dim3((unsigned)nwg), dim3(64 * DP_WAVES), shmem, st, table, hist, hist_len, cand, T, static_cast<const DpImg*>(image), b3,
                       (long long)B, out, scores, ml, csum, nchunk, (long long)chunk, w0);
    if (scores) {
        const int64_t n = B * T;
        int64_t nb = (n + 255) / 256;
        if (nb > kCUs * 8) nb = kCUs * 8;
        hipLaunchKernelGGL(din_pack_scores_k, dim3((unsigned)nb), dim3(256), 0, st, scores, hist_len, T, (long long)B, normalize, ml);
    }
    return DIR_OK;
}

}  // namespace dir

extern "C" int64_t dir_din_pack_workspace_bytes(int64_t B, int want_scores) { return B < 0 ? 0 : dir::din_pack_workspace_bytes(B, want_scores); }
extern "C" int64_t dir_din_pack_image_bytes(void) { return dir::din_pack_image_bytes(); }

extern "C" int dir_din_pack_weights_f32(const float* W1, const float* b1, int H1, const float* W2, const float* b2, int H2, const float* W3,
                                        int activation, const float* act_params, void* image, dir_stream_t stream) {
    using namespace dir;
    const char* name = "dir_din_pack_weights_f32";
    if (!din_pack_covers(DP_K, 1, H1, H2)) return fail(DIR_E_UNSUPPORTED, "%s: H1 <= 80, H2 <= 48, multiples of 4 (H1=%d H2=%d)", name, H1, H2);
    DIR_CHECK_ARG(W1 && b1 && W2 && b2 && W3 && image, "%s: null pointer", name);
    if (!aligned16(image)) return fail(DIR_E_BADARG, "%s: image must be 16-byte aligned", name);
    const int rc = launch_din_pack_image(as_stream(stream), W1, b1, H1, W2, b2, H2, W3, activation, act_params, image);
    if (rc != DIR_OK) return rc;
    DIR_CHECK_LAUNCH(name);
    return DIR_OK;
}

extern "C" int dir_din_attention_pool_packed_f32(const float* table, int K, const int64_t* hist, const int32_t* hist_len, const int64_t* cand,
                                                 int T, const float* W1, const float* b1, int H1, const float* W2, const float* b2, int H2,
                                                 const float* W3, const float* b3, int normalize, int activation, const float* act_params,
                                                 const void* image, int64_t B, float* out, float* scores, void* workspace,
                                                 int64_t workspace_bytes, dir_stream_t stream) {
    using namespace dir;
    const char* name = "dir_din_attention_pool_packed_f32";
    DIR_CHECK_ARG(K > 0 && T > 0 && H1 > 0 && H2 > 0 && B >= 0, "%s: K=%d T=%d H1=%d H2=%d", name, K, T, H1, H2);
    if (!din_pack_covers(K, T, H1, H2))
        return fail(DIR_E_UNSUPPORTED, "%s: covers K = 64, H1 <= 80, H2 <= 48 (multiples of 4), T <= 65535 (K=%d T=%d H1=%d H2=%d)", name, K, T, H1, H2);
    if (B == 0) return DIR_OK;
    DIR_CHECK_ARG(table && hist && cand && b3 && out && workspace && (image || (W1 && b1 && W2 && b2 && W3)), "%s: null pointer", name);
    if (!aligned16(table) || !aligned16(workspace) || !aligned16(out) || !aligned16(image))
        return fail(DIR_E_BADARG, "%s: table / out / workspace / image must be 16-byte aligned", name);
    const int64_t need = din_pack_workspace_bytes(B, scores != nullptr);
    if (workspace_bytes < need) return fail(DIR_E_BADARG, "%s: workspace needs %lld bytes (dir_din_pack_workspace_bytes)", name, (long long)need);
    hipStream_t st = as_stream(stream);
    if (!image) {                // no image of the caller's: one built in the workspace's tail, every call
        void* own = static_cast<unsigned char*>(workspace) + need - din_pack_image_bytes();
        const int rc = launch_din_pack_image(st, W1, b1, H1, W2, b2, H2, W3, activation, act_params, own);
        if (rc != DIR_OK) return rc;
        image = own;
    }
    const int rc = launch_din_pack(st, table, hist, hist_len, cand, T, image, b3, normalize, B, out, scores, activation, workspace);
    if (rc != DIR_OK) return rc;
    DIR_CHECK_LAUNCH(name);
    return DIR_OK;
}
