// din_wave.hip -- DIN local activation unit + pooling, forward, for the (K = 64, H1 <= 80, H2 <= 48, T <= 64) shape class
// (BASELINE.json configs[3]).  NO REFERENCE CODE (README.md:27 links arXiv:1706.06978); the definition is the one in
// include/dir_hip.h (A13) and oracle/dir_oracle.c.
//
// ONE WAVE OWNS ONE SAMPLE -- no workgroup barrier anywhere in the sample loop (the round-1 kernel spent 60 % of a sample's
// time outside MFMA issue: seven barriers per sample, single-wave softmax / pooling phases, LDS round trips between layers).
//
// Everything is computed TRANSPOSED so that one layer's MFMA result is directly the next layer's MFMA operand:
//   pre1^T [H1 x rows] = (Wh+Wd)^T [H1 x K] . h^T [K x rows] + Wp^T . (h*a)^T + c 1^T,     c = (Wa-Wd)^T a + b1   (per sample)
//   pre2^T [H2 x rows] = W2^T [H2 x H1] . sigmoid(pre1^T)                                  + b2 1^T
// v_mfma_f32_16x16x4_f32: lane (kk = l >> 4, r = l & 15) supplies A[m = r][k = kk] and B[k = kk][n = r] and holds
// D[m = 4 kk + g][n = r].  With N = history rows a lane's B operand is ITS OWN row r: the 16 features {16 i + 4 kk + e} of
// row (16 t + r) sit in 16 registers straight from HBM (four 16-byte loads; a row's four lane groups read one contiguous 64-byte
// segment per load) -- the history never goes through LDS.  The result D holds hidden units {16 mt + 4 kk + g} of row r, which is
// exactly a B operand of layer 2 if layer 2's reduction index is enumerated as (mt, g) <-> hidden 16 mt + 4 kk + g: the weights
// (A operands, from LDS) are laid out to match.  So layers chain in registers; the LDS only holds weights (85 KB, filled once
// per workgroup) and is read with one ds_read_b128 per four MFMA steps, shared by the row tiles of a pass.
// A sample is processed in passes of <= 2 row tiles (<= 256 VGPRs: two waves per SIMD hide each other's load latencies);
// the masked softmax runs ONLINE over the passes (running max / sum, rescaled pooled accumulator), so a pass's rows are
// pooled while they are still in registers.
// Loads are software-pipelined across passes and samples: the rows (and candidate row) of the NEXT pass are issued before the
// current pass's MFMAs, the NEXT sample's scalars and history ids one sample ahead, its queue ticket two ahead -- a wave never
// waits for HBM inside a sample.  The current candidate row lives in a per-wave LDS slot (its registers hold the prefetch).
// Samples are handed out by a device-side queue (one atomic per sample, issued a whole sample ahead): history lengths vary
// 1..50, and a static split of 32 samples per wave would leave the slowest wave ~25 % behind the mean.  The forward has no
// cross-sample reduction, so the result does not depend on the order.
#include <atomic>
#include <cstring>

#include "common.hpp"

namespace dir {

typedef float f32x4w __attribute__((ext_vector_type(4)));

constexpr int DW_K = 64, DW_H1P = 80, DW_H2P = 48;
constexpr int DW_WS = DW_K + 4;        // row stride of the [hidden][feature] weight images
constexpr int DW_W2S = DW_H1P + 4;     // row stride of the [h2][hidden] image
constexpr int DW_WAVES = 8;            // waves per workgroup (two per SIMD), one workgroup per CU
constexpr int DW_GROUPS = 8;           // queue shards (sample ranges)
constexpr int DW_SLOTS = 64;           // queue records for launches in flight

typedef __bf16 dw_bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned int dw_u32x4 __attribute__((ext_vector_type(4)));

struct DinWaveSh {
    static constexpr bool kBf3 = false;
    static constexpr int kNP = 3;          // (unused by the fp32 path)
    float whd[DW_H1P * DW_WS];         // (Wh + Wd)^T
    float wp[DW_H1P * DW_WS];          // Wp^T
    float wc[DW_H1P * DW_WS];          // (Wa - Wd)^T
    float w2[DW_H2P * DW_W2S];         // W2^T
    float b1[DW_H1P], b2[DW_H2P], w3[DW_H2P];
    float cvec[DW_WAVES][DW_H1P];      // per wave: the per-sample term c (+ b1)
    float av[DW_WAVES][DW_K];          // per wave: the candidate row of the sample being computed
};

// The bf16x3 variant (the recipe of cin_bf3.hip / tower_bf3.hip: an fp32 operand is the sum of three bf16 pieces by round-to-nearest,
// the six piece products of weight >= 2^-16 accumulate in fp32 on v_mfma_f32_16x16x32_bf16 -- 6 x 16 cycles per 32 k instead of
// 8 x 32 on the fp32 pipe).  The operand / result layouts are those of the fp32 kernel with two of its k-groups per MFMA:
// element j of lane (kk, r) in k-step ks is reduction index 16 (2 ks + (j >> 2)) + 4 kk + (j & 3), which for layer 1 is the lane's
// own registers hv[2 ks], hv[2 ks + 1] and for layer 2 its accumulator tiles 2 ks, 2 ks + 1 -- layers still chain in registers.
// The weight images are split once per workgroup and stored in MFMA A-operand order: [k-step][m tile][piece][lane][8] bf16,
// one ds_read_b128 per piece per (k-step, m tile), shared by the row tiles of a pass.
// THIS FILE IS COMPILED WITHOUT PACKED fp32 VALU INSTRUCTIONS (build.py: -target-feature -packed-fp32-ops).  On gfx950 a
// v_pk_add_f32 / v_pk_fma_f32 whose op_sel sends src1's HIGH register to the LOW result loses that low result in lanes 48..63 when a
// v_mfma_*_16x16x32_{bf16,f16} is issued on the same SIMD right behind it -- by the same wave (half of all executions) or by the other
// wave of the SIMD (1-2 %); s_nop in between does not help, the compiler (ROCm 7.2) does not know the hazard (tools/pk_mfma_probe.hip,
// profiles/NOTES.md R3.6).  The round-2 attempt at this kernel failed on exactly that: the packed (row tile 0, row tile 1) sums of
// layer 3 lost one sigmoid(pre2) * w3 term of the FIRST row tile in 30-150 samples per launch.  tools/check_pk_mfma.py (run by build.py)
// rejects any kernel that holds both instruction kinds.
// Round 4: the same kernel on "fp16 x 2" (NP = 2; cin_bf3.hip explains the arithmetic): two fp16 pieces per operand, the three products
// of weight >= 2^-11 on v_mfma_f32_16x16x32_f16 -- half the matrix instructions and 6 instead of 11 split instructions per operand
// pair.  The unit's operands are table rows, their products with the candidate row, sigmoid / PReLU / Dice outputs and weights:
// embedding-scale numbers, well inside fp16's range; the result stays within 1e-5 of the oracle with a factor 10-20 to spare
// (tests/test_gpu_parity.py: the 26-shape DIN test runs every arithmetic).  DIR_DIN_ARITH = f16x2 (default) | bf16x3 | f32.
template <int NP>
struct DinWaveShP {
    static constexpr bool kBf3 = true;
    static constexpr int kNP = NP;
    unsigned int whd3[2 * 5 * NP * 64 * 4];    // (Wh + Wd)^T pieces
    unsigned int wp3[2 * 5 * NP * 64 * 4];     // Wp^T pieces
    unsigned int w23[3 * 3 * NP * 64 * 4];     // W2^T pieces; hidden 80..95 of the third k-step are zero
    float wc[DW_H1P * DW_WS];                  // (Wa - Wd)^T (the per-sample term stays on the VALU in fp32)
    float b1[DW_H1P], b2[DW_H2P], w3[DW_H2P];
    float cvec[DW_WAVES][DW_H1P];
    float av[DW_WAVES][DW_K];
};
using DinWaveSh3 = DinWaveShP<3>;
using DinWaveSh2 = DinWaveShP<2>;

// [slot][0..7] next sample of each range, [slot][8] waves finished.  All zero between launches (the last wave of a launch
// clears its record).
__device__ unsigned int dw_queue[DW_SLOTS][16];

// sigmoid(x) = 1 / (1 + 2^(-x log2 e)).  The factor -log2 e is folded into the weight / bias images (the MFMA result IS the
// exponent), so a sigmoid is v_exp + v_add + v_rcp: VALU work is not hidden behind fp32 MFMAs, every instruction saved counts.
constexpr float DW_NLOG2E = -1.4426950408889634f;
__device__ __forceinline__ float dw_sigmoid_pre(float y) { return __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(y)); }
// The hidden activations of the unit (round 4): ACT 0 = sigmoid (the default; what the public DIN code uses in its attention MLP),
// 1 = PReLU  f(s) = s > 0 ? s : alpha s, 2 = Dice  f(s) = p s + (1 - p) alpha s with p = sigmoid((s - E[s]) / sqrt(Var[s] + eps))
// (arXiv:1706.06978 section 5.3; inference form: the moving statistics folded into scale / shift).  For ACT != 0 the weight images carry
// NO -log2 e factor (the MFMA result is the pre-activation itself) and the per-unit parameters sit in LDS behind the weight images:
// row 0 = alpha, row 1 = -log2 e * scale, row 2 = -log2 e * shift.  A padded hidden unit has pre-activation 0: f(0) = 0 in both.
template <int ACT>
__device__ __forceinline__ float dw_act(float pre, float alpha, float nscale, float nshift) {
    if constexpr (ACT == 0) return dw_sigmoid_pre(pre);
    else if constexpr (ACT == 1) return pre > 0.f ? pre : alpha * pre;
    else {
        const float pgate = dw_sigmoid_pre(fmaf(pre, nscale, nshift));
        return pre * fmaf(pgate, 1.0f - alpha, alpha);
    }
}
constexpr int DW_ACT_S1 = 96, DW_ACT_S2 = 48;          // strides of the staged parameter rows (padded units: alpha = scale = shift = 0)
constexpr int DW_ACT_FLOATS = 3 * DW_ACT_S1 + 3 * DW_ACT_S2;
// the three parameters of the four hidden units 4-aligned at h (one ds_read_b128 each; nothing is read for the sigmoid)
template <int ACT>
__device__ __forceinline__ void dw_act_params(const float* rows, int stride, int h, float (&al)[4], float (&ns)[4], float (&nt)[4]) {
#pragma unroll
    for (int g = 0; g < 4; ++g) al[g] = ns[g] = nt[g] = 0.f;
    if constexpr (ACT != 0) {
        const float4 a = *reinterpret_cast<const float4*>(rows + h);
        al[0] = a.x; al[1] = a.y; al[2] = a.z; al[3] = a.w;
    }
    if constexpr (ACT == 2) {
        const float4 b = *reinterpret_cast<const float4*>(rows + stride + h), c = *reinterpret_cast<const float4*>(rows + 2 * stride + h);
        ns[0] = b.x; ns[1] = b.y; ns[2] = b.z; ns[3] = b.w;
        nt[0] = c.x; nt[1] = c.y; nt[2] = c.z; nt[3] = c.w;
    }
}
__device__ __forceinline__ float4 dw_ld4(const float* p) { return *reinterpret_cast<const float4*>(p); }
__device__ __forceinline__ float dw_dot4(float4 a, float4 b, float acc) {
    acc = fmaf(a.x, b.x, acc); acc = fmaf(a.y, b.y, acc); acc = fmaf(a.z, b.z, acc); acc = fmaf(a.w, b.w, acc);
    return acc;
}
#define DW_MFMA(a, b, c) __builtin_amdgcn_mfma_f32_16x16x4f32((a), (b), (c), 0, 0, 0)
#define DW_MFMA3(a, b, c) __builtin_amdgcn_mfma_f32_16x16x32_bf16((a), (b), (c), 0, 0, 0)

__device__ __forceinline__ unsigned int dw_pk(float a, float b) {      // v_cvt_pk_bf16_f32 (round to nearest even), a in the low half
    typedef __bf16 pk2_t __attribute__((ext_vector_type(2)));
    const pk2_t v = {(__bf16)a, (__bf16)b};
    unsigned int w = __builtin_bit_cast(unsigned int, v);
    asm("" : "+v"(w));      // see dense_bf3.hip (db3_pk): keeps the compiler from folding the split away
    return w;
}
__device__ __forceinline__ void dw_split_pair(float a, float b, unsigned int& w0, unsigned int& w1, unsigned int& w2) {
    w0 = dw_pk(a, b);
    const float ra = a - __builtin_bit_cast(float, w0 << 16), rb = b - __builtin_bit_cast(float, w0 & 0xffff0000u);
    w1 = dw_pk(ra, rb);
    const float sa = ra - __builtin_bit_cast(float, w1 << 16), sb = rb - __builtin_bit_cast(float, w1 & 0xffff0000u);
    w2 = dw_pk(sa, sb);
}
typedef _Float16 dw_f16x8 __attribute__((ext_vector_type(8)));
__device__ __forceinline__ unsigned int dw_pk_h(float a, float b) {     // v_cvt_pk_f16_f32 (round to nearest even), a in the low half
    typedef _Float16 h2_t __attribute__((ext_vector_type(2)));
    const h2_t v = {(_Float16)a, (_Float16)b};
    unsigned int w = __builtin_bit_cast(unsigned int, v);
    asm("" : "+v"(w));
    return w;
}
template <int NP> struct DwPc;
template <> struct DwPc<3> {
    using op_t = dw_bf16x8;
    __device__ static __forceinline__ void split(float a, float b, unsigned int (&w)[3]) { dw_split_pair(a, b, w[0], w[1], w[2]); }
    __device__ static __forceinline__ f32x4w mma(const op_t (&a)[3], const op_t (&x)[3], f32x4w c) {      // six products, smallest first
        c = DW_MFMA3(a[0], x[2], c);
        c = DW_MFMA3(a[2], x[0], c);
        c = DW_MFMA3(a[1], x[1], c);
        c = DW_MFMA3(a[0], x[1], c);
        c = DW_MFMA3(a[1], x[0], c);
        c = DW_MFMA3(a[0], x[0], c);
        return c;
    }
};
template <> struct DwPc<2> {
    using op_t = dw_f16x8;
    __device__ static __forceinline__ void split(float a, float b, unsigned int (&w)[2]) {
        typedef _Float16 h2_t __attribute__((ext_vector_type(2)));
        w[0] = dw_pk_h(a, b);
        const h2_t h = __builtin_bit_cast(h2_t, w[0]);
        w[1] = dw_pk_h(a - (float)h[0], b - (float)h[1]);
    }
    __device__ static __forceinline__ f32x4w mma(const op_t (&a)[2], const op_t (&x)[2], f32x4w c) {      // three products, smallest first
        c = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[1], x[0], c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[0], x[1], c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[0], x[0], c, 0, 0, 0);
        return c;
    }
};
// eight fp32 values (two float4) -> the NP operands that sum to them
template <int NP>
__device__ __forceinline__ void dw_split8p(const float4 s0, const float4 s1, typename DwPc<NP>::op_t (&x)[NP]) {
    unsigned int w[NP][4];
    const float v[8] = {s0.x, s0.y, s0.z, s0.w, s1.x, s1.y, s1.z, s1.w};
#pragma unroll
    for (int pr = 0; pr < 4; ++pr) {
        unsigned int pw[NP];
        DwPc<NP>::split(v[2 * pr], v[2 * pr + 1], pw);
#pragma unroll
        for (int pc = 0; pc < NP; ++pc) w[pc][pr] = pw[pc];
    }
#pragma unroll
    for (int pc = 0; pc < NP; ++pc) x[pc] = __builtin_bit_cast(typename DwPc<NP>::op_t, (dw_u32x4){w[pc][0], w[pc][1], w[pc][2], w[pc][3]});
}
// eight fp32 values (two float4) -> the three bf16x8 operands that sum to them
__device__ __forceinline__ void dw_split8(const float4 s0, const float4 s1, dw_bf16x8 (&x)[3]) {
    unsigned int w[3][4];
    dw_split_pair(s0.x, s0.y, w[0][0], w[1][0], w[2][0]);
    dw_split_pair(s0.z, s0.w, w[0][1], w[1][1], w[2][1]);
    dw_split_pair(s1.x, s1.y, w[0][2], w[1][2], w[2][2]);
    dw_split_pair(s1.z, s1.w, w[0][3], w[1][3], w[2][3]);
#pragma unroll
    for (int pc = 0; pc < 3; ++pc) x[pc] = __builtin_bit_cast(dw_bf16x8, (dw_u32x4){w[pc][0], w[pc][1], w[pc][2], w[pc][3]});
}
// acc += A . B with A = a[0] + a[1] + a[2], B = x[0] + x[1] + x[2]: the six products of weight >= 2^-16, smallest first
__device__ __forceinline__ f32x4w dw_mfma6(const dw_bf16x8 (&a)[3], const dw_bf16x8 (&x)[3], f32x4w c) {
    c = DW_MFMA3(a[0], x[2], c);
    c = DW_MFMA3(a[2], x[0], c);
    c = DW_MFMA3(a[1], x[1], c);
    c = DW_MFMA3(a[0], x[1], c);
    c = DW_MFMA3(a[1], x[0], c);
    c = DW_MFMA3(a[0], x[0], c);
    return c;
}

// The sample queue.  A wave belongs to one of DW_GROUPS sample ranges (blockIdx & 7) and draws tickets of DW_CH consecutive samples
// from that range's counter.  The atomic of the NEXT ticket is issued when the current one is opened and is only waited for when
// its samples are needed, DW_CH samples (> 10 us) later: the ticket round trip (1-3 us under load; it cost 0.1-0.16 ms of a 0.7 ms
// launch when every sample waited for its own ticket) is off the critical path.  q == nullptr (DIR_DIN_STATIC=1, an A/B switch):
// a static stride over the samples instead.
constexpr int DW_CH = 2;
struct DwQueue {
    unsigned int* q;
    long long lo, hi, next;
    int left;
    bool dead;
    unsigned int ticket;       // lane 0: the in-flight atomic's result
    long long stride_next, stride;   // static mode

    __device__ __forceinline__ void issue() {
        ticket = 0;
        if ((threadIdx.x & 63) == 0) ticket = atomicAdd(q, (unsigned int)DW_CH);
    }
    __device__ __forceinline__ void init(unsigned int* qrec, long long B) {
        const int G = (int)gridDim.x < DW_GROUPS ? (int)gridDim.x : DW_GROUPS;   // every range needs at least one workgroup
        const int g = (int)(blockIdx.x % (unsigned)G);
        q = qrec ? qrec + g : nullptr;
        lo = (long long)g * B / G;
        hi = (long long)(g + 1) * B / G;
        left = 0;
        dead = false;
        next = 0;
        stride = (long long)gridDim.x * DW_WAVES;
        stride_next = (long long)blockIdx.x * DW_WAVES + (threadIdx.x >> 6);
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
            left = (int)((hi - base) < DW_CH ? (hi - base) : DW_CH);
            issue();
        }
        --left;
        return next++;
    }
    __device__ __forceinline__ void drain() {            // the outstanding ticket must have landed before the record is cleared
        if (q && !dead) (void)__builtin_amdgcn_readfirstlane((int)ticket);
    }
};

// rows of up to two tiles: this lane's 16 features {16 i + 4 kk + e} of its own row (16 t + r), straight from HBM; a masked
// row (id < 0) is zeros
__device__ __forceinline__ void dw_load_rows(const float* __restrict__ table, const int kk, const long long id0, const long long id1,
                                             float4 (&hv)[2][4]) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        hv[0][i] = make_float4(0.f, 0.f, 0.f, 0.f);
        hv[1][i] = make_float4(0.f, 0.f, 0.f, 0.f);
        if (id0 >= 0) hv[0][i] = dw_ld4(table + id0 * DW_K + 16 * i + 4 * kk);
        if (id1 >= 0) hv[1][i] = dw_ld4(table + id1 * DW_K + 16 * i + 4 * kk);
    }
}

// One pass: NT row tiles whose rows are hv[rt] and whose history ids are id[rt] (-1: masked row).  Updates the online-softmax state.
// rec[rt] != nullptr (training forward): the row's z1 and z2 go into its record (common.hpp: kDinRec*).
template <int NT, int ACT, typename Sh>
__device__ __forceinline__ void dw_pass(const Sh& sh, const float* actl, const int w, const int r16, const int kk,
                                        const long long (&id)[NT], const float4 (&hv)[2][4], const float b3, const bool normalize,
                                        const float inv_sqrt_k, float& m_run, float& l_run, float4 (&o)[4], float (&xs)[NT],
                                        float* const (&rec)[NT]) {
    constexpr int NA = NT == 1 ? 2 : 1;      // a single row tile alternates two accumulators (dependent MFMAs need 40 cycles)
    float4 hp[NT][4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const float4 a4 = dw_ld4(&sh.av[w][16 * i + 4 * kk]);
#pragma unroll
        for (int rt = 0; rt < NT; ++rt)
            hp[rt][i] = make_float4(hv[rt][i].x * a4.x, hv[rt][i].y * a4.y, hv[rt][i].z * a4.z, hv[rt][i].w * a4.w);
    }
    f32x4w acc2[3][NT];
#pragma unroll
    for (int m2 = 0; m2 < 3; ++m2) {
        const float4 c4 = dw_ld4(&sh.b2[16 * m2 + 4 * kk]);
#pragma unroll
        for (int rt = 0; rt < NT; ++rt) acc2[m2][rt] = (f32x4w){c4.x, c4.y, c4.z, c4.w};
    }
    if constexpr (Sh::kBf3) {
        const int lane4 = 4 * (16 * kk + r16);          // dword offset of this lane's 16 bytes inside a 1 KB operand tile
        // ---- layer 1: pre1^T = (Wh+Wd)^T h^T + Wp^T (h*a)^T + c 1^T, four k-steps of 32 ------------------------------------------
        f32x4w acc1[5][NT];
#pragma unroll
        for (int mt = 0; mt < 5; ++mt) {
            const float4 c4 = dw_ld4(&sh.cvec[w][16 * mt + 4 * kk]);
#pragma unroll
            for (int rt = 0; rt < NT; ++rt) acc1[mt][rt] = (f32x4w){c4.x, c4.y, c4.z, c4.w};
        }
#pragma unroll
        for (int part = 0; part < 2; ++part) {
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                constexpr int NP = Sh::kNP;
                using op_t = typename DwPc<NP>::op_t;
                op_t xb[NT][NP];
#pragma unroll
                for (int rt = 0; rt < NT; ++rt)
                    dw_split8p<NP>(part ? hp[rt][2 * ks] : hv[rt][2 * ks], part ? hp[rt][2 * ks + 1] : hv[rt][2 * ks + 1], xb[rt]);
                const unsigned int* img = part ? sh.wp3 : sh.whd3;
#pragma unroll
                for (int mt = 0; mt < 5; ++mt) {
                    op_t a[NP];
#pragma unroll
                    for (int pc = 0; pc < NP; ++pc)
                        a[pc] = __builtin_bit_cast(op_t, *reinterpret_cast<const dw_u32x4*>(img + ((ks * 5 + mt) * NP + pc) * 256 + lane4));
#pragma unroll
                    for (int rt = 0; rt < NT; ++rt) acc1[mt][rt] = DwPc<NP>::mma(a, xb[rt], acc1[mt][rt]);
                }
            }
        }
        // z1 = sigmoid(pre1) in place: hidden 16 mt + 4 kk + g of row r -- element (mt & 1) * 4 + g of layer 2's k-step mt >> 1
#pragma unroll
        for (int mt = 0; mt < 5; ++mt) {
            float al[4], ns[4], nt[4];
            dw_act_params<ACT>(actl, DW_ACT_S1, 16 * mt + 4 * kk, al, ns, nt);
#pragma unroll
            for (int rt = 0; rt < NT; ++rt)
#pragma unroll
                for (int g = 0; g < 4; ++g) acc1[mt][rt][g] = dw_act<ACT>(acc1[mt][rt][g], al[g], ns[g], nt[g]);
        }
#pragma unroll
        for (int rt = 0; rt < NT; ++rt)
            if (rec[rt]) {
#pragma unroll
                for (int mt = 0; mt < 5; ++mt)
                    din_rec_store(rec[rt] + kDinRecZ1 + 16 * mt + 4 * kk, acc1[mt][rt][0], acc1[mt][rt][1], acc1[mt][rt][2], acc1[mt][rt][3]);
            }
        // ---- layer 2: pre2^T, three k-steps (hidden 80..95 are zeros on both sides) -------------------------------------------------
#pragma unroll
        for (int ks = 0; ks < 3; ++ks) {
            constexpr int NP = Sh::kNP;
            using op_t = typename DwPc<NP>::op_t;
            op_t xb[NT][NP];
#pragma unroll
            for (int rt = 0; rt < NT; ++rt) {
                const f32x4w z0 = acc1[2 * ks][rt];
                const f32x4w z1 = 2 * ks + 1 < 5 ? acc1[2 * ks + 1 < 5 ? 2 * ks + 1 : 0][rt] : (f32x4w){0.f, 0.f, 0.f, 0.f};
                dw_split8p<NP>(make_float4(z0[0], z0[1], z0[2], z0[3]), make_float4(z1[0], z1[1], z1[2], z1[3]), xb[rt]);
            }
#pragma unroll
            for (int m2 = 0; m2 < 3; ++m2) {
                op_t a[NP];
#pragma unroll
                for (int pc = 0; pc < NP; ++pc)
                    a[pc] = __builtin_bit_cast(op_t, *reinterpret_cast<const dw_u32x4*>(sh.w23 + ((ks * 3 + m2) * NP + pc) * 256 + lane4));
#pragma unroll
                for (int rt = 0; rt < NT; ++rt) acc2[m2][rt] = DwPc<NP>::mma(a, xb[rt], acc2[m2][rt]);
            }
        }
    } else {
    // ---- layer 1: pre1^T, accumulators start at the per-sample term -------------------------------------------------------------
    f32x4w acc1[5][NT][NA];
#pragma unroll
    for (int mt = 0; mt < 5; ++mt) {
        const float4 c4 = dw_ld4(&sh.cvec[w][16 * mt + 4 * kk]);
#pragma unroll
        for (int rt = 0; rt < NT; ++rt) {
            acc1[mt][rt][0] = (f32x4w){c4.x, c4.y, c4.z, c4.w};
            if (NA == 2) acc1[mt][rt][NA - 1] = (f32x4w){0.f, 0.f, 0.f, 0.f};
        }
    }
#pragma unroll
    for (int mt = 0; mt < 5; ++mt) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const float4 ah = dw_ld4(&sh.whd[(16 * mt + r16) * DW_WS + 16 * i + 4 * kk]);
            const float4 ap = dw_ld4(&sh.wp[(16 * mt + r16) * DW_WS + 16 * i + 4 * kk]);
            const float ahv[4] = {ah.x, ah.y, ah.z, ah.w}, apv[4] = {ap.x, ap.y, ap.z, ap.w};
#pragma unroll
            for (int e = 0; e < 4; ++e) {
#pragma unroll
                for (int rt = 0; rt < NT; ++rt) {
                    const float hve = e == 0 ? hv[rt][i].x : e == 1 ? hv[rt][i].y : e == 2 ? hv[rt][i].z : hv[rt][i].w;
                    acc1[mt][rt][0] = DW_MFMA(ahv[e], hve, acc1[mt][rt][0]);
                }
#pragma unroll
                for (int rt = 0; rt < NT; ++rt) {
                    const float hpe = e == 0 ? hp[rt][i].x : e == 1 ? hp[rt][i].y : e == 2 ? hp[rt][i].z : hp[rt][i].w;
                    acc1[mt][rt][NA - 1] = DW_MFMA(apv[e], hpe, acc1[mt][rt][NA - 1]);
                }
            }
        }
    }
    // z1 = sigmoid(pre1), in place: lane (kk, r) holds hidden 16 mt + 4 kk + g of row r -- a B operand of layer 2
#pragma unroll
    for (int mt = 0; mt < 5; ++mt) {
        float al[4], ns[4], nt[4];
        dw_act_params<ACT>(actl, DW_ACT_S1, 16 * mt + 4 * kk, al, ns, nt);
#pragma unroll
        for (int rt = 0; rt < NT; ++rt)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const float pre = NA == 2 ? acc1[mt][rt][0][g] + acc1[mt][rt][NA - 1][g] : acc1[mt][rt][0][g];
                acc1[mt][rt][0][g] = dw_act<ACT>(pre, al[g], ns[g], nt[g]);
            }
    }
#pragma unroll
    for (int rt = 0; rt < NT; ++rt)
        if (rec[rt]) {
#pragma unroll
            for (int mt = 0; mt < 5; ++mt)
                din_rec_store(rec[rt] + kDinRecZ1 + 16 * mt + 4 * kk, acc1[mt][rt][0][0], acc1[mt][rt][0][1], acc1[mt][rt][0][2], acc1[mt][rt][0][3]);
        }
    // ---- layer 2: pre2^T; the reduction walks (mt, g) <-> hidden 16 mt + 4 kk + g --------------------------------------------------
#pragma unroll
    for (int mt = 0; mt < 5; ++mt) {
        float aw[3][4];
#pragma unroll
        for (int m2 = 0; m2 < 3; ++m2) {
            const float4 t4 = dw_ld4(&sh.w2[(16 * m2 + r16) * DW_W2S + 16 * mt + 4 * kk]);
            aw[m2][0] = t4.x; aw[m2][1] = t4.y; aw[m2][2] = t4.z; aw[m2][3] = t4.w;
        }
#pragma unroll
        for (int g = 0; g < 4; ++g)
#pragma unroll
            for (int m2 = 0; m2 < 3; ++m2)
#pragma unroll
                for (int rt = 0; rt < NT; ++rt) acc2[m2][rt] = DW_MFMA(aw[m2][g], acc1[mt][rt][0][g], acc2[m2][rt]);
    }
    }
    // ---- layer 3 + mask + online softmax + pooling of this pass's rows -----------------------------------------------------------------
    float wv[3][4];
#pragma unroll
    for (int m2 = 0; m2 < 3; ++m2) {
        const float4 t4 = dw_ld4(&sh.w3[16 * m2 + 4 * kk]);
        wv[m2][0] = t4.x; wv[m2][1] = t4.y; wv[m2][2] = t4.z; wv[m2][3] = t4.w;
    }
    float sc[NT];
#pragma unroll
    for (int rt = 0; rt < NT; ++rt) {
        float sp = 0.f;
#pragma unroll
        for (int m2 = 0; m2 < 3; ++m2) {
            float zz[4];
            float al[4], ns[4], nt[4];
            dw_act_params<ACT>(actl + 3 * DW_ACT_S1, DW_ACT_S2, 16 * m2 + 4 * kk, al, ns, nt);
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                zz[g] = dw_act<ACT>(acc2[m2][rt][g], al[g], ns[g], nt[g]);
                sp = fmaf(zz[g], wv[m2][g], sp);
            }
            if (rec[rt]) din_rec_store(rec[rt] + kDinRecZ2 + 16 * m2 + 4 * kk, zz[0], zz[1], zz[2], zz[3]);
        }
        sp += __shfl_xor(sp, 16, 64);        // the four lane groups hold the four quarters of the H2 sum of row r
        sp += __shfl_xor(sp, 32, 64);
        sc[rt] = sp + b3;
    }
    if (normalize) {
        float mx = m_run;
#pragma unroll
        for (int rt = 0; rt < NT; ++rt) {
            xs[rt] = id[rt] >= 0 ? sc[rt] * inv_sqrt_k : -INFINITY;
            mx = fmaxf(mx, row16_max(xs[rt]));
        }
        if (mx > -INFINITY) {                // wave-uniform (row maxima are identical in the four lane groups)
            const float rescale = __expf(m_run - mx);      // m_run = -inf: 0
            float psum = 0.f;
            float p[NT];
#pragma unroll
            for (int rt = 0; rt < NT; ++rt) {
                p[rt] = id[rt] >= 0 ? __expf(xs[rt] - mx) : 0.f;
                psum += row16_sum(p[rt]);
            }
            l_run = l_run * rescale + psum;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                float4 acc = make_float4(o[i].x * rescale, o[i].y * rescale, o[i].z * rescale, o[i].w * rescale);
#pragma unroll
                for (int rt = 0; rt < NT; ++rt) {
                    acc.x = fmaf(p[rt], hv[rt][i].x, acc.x); acc.y = fmaf(p[rt], hv[rt][i].y, acc.y);
                    acc.z = fmaf(p[rt], hv[rt][i].z, acc.z); acc.w = fmaf(p[rt], hv[rt][i].w, acc.w);
                }
                o[i] = acc;
            }
            m_run = mx;
        }
    } else {
#pragma unroll
        for (int rt = 0; rt < NT; ++rt) {
            xs[rt] = id[rt] >= 0 ? sc[rt] : 0.f;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                o[i].x = fmaf(xs[rt], hv[rt][i].x, o[i].x); o[i].y = fmaf(xs[rt], hv[rt][i].y, o[i].y);
                o[i].z = fmaf(xs[rt], hv[rt][i].z, o[i].z); o[i].w = fmaf(xs[rt], hv[rt][i].w, o[i].w);
            }
        }
    }
}

template <typename Sh, bool SAVE, int ACT>
__global__ __launch_bounds__(64 * DW_WAVES, 2) void din_wave_k(const float* __restrict__ table, const int64_t* __restrict__ hist,
                                                               const int32_t* __restrict__ hist_len, const int64_t* __restrict__ cand,
                                                               int T, const float* __restrict__ W1, const float* __restrict__ b1, int H1,
                                                               const float* __restrict__ W2, const float* __restrict__ b2, int H2,
                                                               const float* __restrict__ W3, const float* __restrict__ b3, int normalize,
                                                               long long B, float* __restrict__ out, float* __restrict__ scores, int slot,
                                                               const int64_t* __restrict__ tile_off, float* __restrict__ saved,
                                                               const float* __restrict__ act_params /* ACT != 0: [3 H1 + 3 H2] */) {
    extern __shared__ __attribute__((aligned(16))) unsigned char dw_smem[];
    Sh& sh = *reinterpret_cast<Sh*>(dw_smem);
    float* const actl = reinterpret_cast<float*>(dw_smem + ((sizeof(Sh) + 15) & ~(size_t)15));      // ACT != 0: [3][96] layer 1, [3][48] layer 2
    constexpr float SC = ACT == 0 ? DW_NLOG2E : 1.0f;      // what the weight / bias images are multiplied by
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, r16 = lane & 15, kk = lane >> 4;
    if constexpr (ACT != 0) {
        for (int idx = tid; idx < DW_ACT_FLOATS; idx += 64 * DW_WAVES) {
            float v = 0.f;
            if (idx < 3 * DW_ACT_S1) {
                const int row = idx / DW_ACT_S1, h = idx - row * DW_ACT_S1;
                if (h < H1) v = act_params[row * H1 + h] * (row == 0 ? 1.0f : DW_NLOG2E);
            } else {
                const int j = idx - 3 * DW_ACT_S1, row = j / DW_ACT_S2, h = j - row * DW_ACT_S2;
                if (h < H2) v = act_params[3 * H1 + row * H2 + h] * (row == 0 ? 1.0f : DW_NLOG2E);
            }
            actl[idx] = v;
        }
    }
    // ---- weight images, once per workgroup ---------------------------------------------------------------------------------------------------
#pragma unroll 3
    for (int idx = tid; idx < (DW_H1P / 4) * DW_K; idx += 64 * DW_WAVES) {      // 16-byte loads along m (H1 % 4 == 0, W1 16-byte aligned)
        const int f = idx / (DW_H1P / 4), m = 4 * (idx - f * (DW_H1P / 4));
        float4 vh = make_float4(0.f, 0.f, 0.f, 0.f), va = vh, vd = vh, vp = vh;
        if (m < H1) {
            vh = dw_ld4(W1 + (size_t)f * H1 + m);
            va = dw_ld4(W1 + (size_t)(DW_K + f) * H1 + m);
            vd = dw_ld4(W1 + (size_t)(2 * DW_K + f) * H1 + m);
            vp = dw_ld4(W1 + (size_t)(3 * DW_K + f) * H1 + m);
        }
        const float sc_ = SC;      // pre-activations come out of the MFMAs already multiplied by -log2 e
        const float h4[4] = {(vh.x + vd.x) * sc_, (vh.y + vd.y) * sc_, (vh.z + vd.z) * sc_, (vh.w + vd.w) * sc_};
        const float p4[4] = {vp.x * sc_, vp.y * sc_, vp.z * sc_, vp.w * sc_};
        const float c4[4] = {(va.x - vd.x) * sc_, (va.y - vd.y) * sc_, (va.z - vd.z) * sc_, (va.w - vd.w) * sc_};
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            if constexpr (!Sh::kBf3) {
                sh.whd[(m + e) * DW_WS + f] = h4[e];
                sh.wp[(m + e) * DW_WS + f] = p4[e];
            }
            sh.wc[(m + e) * DW_WS + f] = c4[e];
        }
    }
    if constexpr (Sh::kBf3) {
        // A-operand images: dword jp of lane l of tile (ks, mt) holds elements j = 2 jp, 2 jp + 1 = W[m = 16 mt + (l & 15)][k], k + 1,
        // k = 16 (2 ks + (j >> 2)) + 4 (l >> 4) + (j & 3)
        for (int idx = tid; idx < 2 * 5 * 64 * 4; idx += 64 * DW_WAVES) {
            const int jp = idx & 3, l = (idx >> 2) & 63, t = idx >> 8;
            const int ks = t / 5, mt = t - 5 * ks;
            const int m = 16 * mt + (l & 15);
            const int f = 16 * (2 * ks + (jp >> 1)) + 4 * (l >> 4) + 2 * (jp & 1);
            float hd[2] = {0.f, 0.f}, pp[2] = {0.f, 0.f};
            if (m < H1) {
#pragma unroll
                for (int e = 0; e < 2; ++e) {
                    hd[e] = (W1[(size_t)(f + e) * H1 + m] + W1[(size_t)(2 * DW_K + f + e) * H1 + m]) * SC;
                    pp[e] = W1[(size_t)(3 * DW_K + f + e) * H1 + m] * SC;
                }
            }
            constexpr int NP = Sh::kNP;
            unsigned int qw[NP];
            DwPc<NP>::split(hd[0], hd[1], qw);
#pragma unroll
            for (int pc = 0; pc < NP; ++pc) sh.whd3[(t * NP + pc) * 256 + l * 4 + jp] = qw[pc];
            DwPc<NP>::split(pp[0], pp[1], qw);
#pragma unroll
            for (int pc = 0; pc < NP; ++pc) sh.wp3[(t * NP + pc) * 256 + l * 4 + jp] = qw[pc];
        }
        for (int idx = tid; idx < 3 * 3 * 64 * 4; idx += 64 * DW_WAVES) {
            const int jp = idx & 3, l = (idx >> 2) & 63, t = idx >> 8;
            const int ks = t / 3, m2 = t - 3 * ks;
            const int h2 = 16 * m2 + (l & 15);
            const int hid = 16 * (2 * ks + (jp >> 1)) + 4 * (l >> 4) + 2 * (jp & 1);
            float v[2];
#pragma unroll
            for (int e = 0; e < 2; ++e)      // a padded hidden unit is sigmoid(0) = 0.5: its weights are zero
                v[e] = (hid + e < H1 && h2 < H2) ? W2[(size_t)(hid + e) * H2 + h2] * SC : 0.f;
            constexpr int NP = Sh::kNP;
            unsigned int qw[NP];
            DwPc<NP>::split(v[0], v[1], qw);
#pragma unroll
            for (int pc = 0; pc < NP; ++pc) sh.w23[(t * NP + pc) * 256 + l * 4 + jp] = qw[pc];
        }
    } else {
        for (int idx = tid; idx < DW_H2P * DW_H1P; idx += 64 * DW_WAVES) {
            const int hid = idx / DW_H2P, h2 = idx - hid * DW_H2P;
            sh.w2[h2 * DW_W2S + hid] = (hid < H1 && h2 < H2) ? W2[(size_t)hid * H2 + h2] * SC : 0.f;   // a padded hidden unit is sigmoid(0) = 0.5:
        }                                                                                           // its weights are zero
    }
    for (int idx = tid; idx < DW_H1P; idx += 64 * DW_WAVES) sh.b1[idx] = idx < H1 ? b1[idx] * SC : 0.f;
    for (int idx = tid; idx < DW_H2P; idx += 64 * DW_WAVES) {
        sh.b2[idx] = idx < H2 ? b2[idx] * SC : 0.f;
        sh.w3[idx] = idx < H2 ? W3[idx] : 0.f;
    }
    __syncthreads();
    const float bias3 = b3[0];
    const float inv_sqrt_k = 1.0f / sqrtf((float)DW_K);
    const int ntile_t = (T + 15) >> 4;
    unsigned int* q = slot >= 0 ? dw_queue[slot] : nullptr;
    DwQueue dq;
    dq.init(q, B);
    // Sample descriptors travel through three stages, each consumed one sample after it was issued (nothing blocks):
    //   index (queue ticket) -> scalars (length, candidate id; the index is wave-uniform: scalar loads) -> the 4 x 16 history ids
    auto uniform64 = [](long long v) {
        const unsigned int lo = (unsigned int)__builtin_amdgcn_readfirstlane((int)(unsigned int)(unsigned long long)v);
        const unsigned int hi = (unsigned int)__builtin_amdgcn_readfirstlane((int)(unsigned int)((unsigned long long)v >> 32));
        return (long long)(((unsigned long long)hi << 32) | lo);
    };
    auto load_scalars = [&](const long long bb, int& len, long long& cid, long long& toff) {
        len = 0;
        cid = -1;
        toff = 0;
        if (bb >= 0) {
            const long long bs = uniform64(bb);
            len = hist_len ? min((int)hist_len[bs], T) : T;
            cid = cand[bs];
            if (SAVE) toff = tile_off[bs];
        }
    };
    auto load_ids = [&](const long long bb, const int len, long long (&id)[4]) {
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const int j = 16 * t + r16;
            id[t] = j < len ? hist[bb * T + j] : -1;           // len == 0 beyond the last sample
        }
    };
    auto load_cand = [&](const long long cid, float4 (&a)[4]) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            a[i] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (cid >= 0) a[i] = dw_ld4(table + cid * DW_K + 16 * i + 4 * kk);
        }
    };
    long long b = dq.take();
    long long bn = b >= 0 ? dq.take() : -1;
    long long bt = bn >= 0 ? dq.take() : -1;                     // scalars in flight
    long long bq = bt >= 0 ? dq.take() : -1;                     // index only
    int len, len_n, len_t;
    long long cid, cid_n, cid_t, toff, toff_n, toff_t, id[4], id_n[4];
    load_scalars(b, len, cid, toff);
    load_scalars(bn, len_n, cid_n, toff_n);
    load_scalars(bt, len_t, cid_t, toff_t);
    load_ids(b, len, id);
    load_ids(bn, len_n, id_n);
    float4 hv[2][4], hvn[2][4], an[4];
    load_cand(cid, an);
    dw_load_rows(table, kk, id[0], id[1], hv);
    while (b >= 0) {
        const int RT = (len + 15) >> 4;
        // ---- this sample's candidate row -> its LDS slot; per-sample term c[m] = sum_f a[f] (Wa - Wd)[f][m] + b1[m] ----------------------------
        if (r16 == 0) {
#pragma unroll
            for (int i = 0; i < 4; ++i) *reinterpret_cast<float4*>(&sh.av[w][16 * i + 4 * kk]) = an[i];
        }
#pragma unroll
        for (int mt = 0; mt < 5; ++mt) {          // lane (kk, r) sums its 16 features for m = 16 mt + r
            float part = 0.f;
#pragma unroll
            for (int i = 0; i < 4; ++i) part = dw_dot4(an[i], dw_ld4(&sh.wc[(16 * mt + r16) * DW_WS + 16 * i + 4 * kk]), part);
            part += __shfl_xor(part, 16, 64);
            part += __shfl_xor(part, 32, 64);
            if (kk == 0) sh.cvec[w][16 * mt + r16] = part + sh.b1[16 * mt + r16];
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");   // same-wave LDS hand-off: the DS queue is in order, the fences
        __builtin_amdgcn_wave_barrier();                         // only keep the compiler from moving the reads above the writes
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        float m_run = -INFINITY, l_run = 0.f;
        float4 o[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) o[i] = make_float4(0.f, 0.f, 0.f, 0.f);
        float xs[4] = {0.f, 0.f, 0.f, 0.f};
        if (normalize) xs[0] = xs[1] = xs[2] = xs[3] = -INFINITY;
        auto record = [&](const int t) -> float* {               // training forward: the record of this lane's row of tile t (rows inside the length)
            return (SAVE && 16 * t + r16 < len) ? saved + ((toff + t) * 16 + r16) * kDinRecRow : nullptr;
        };
        // ---- pass 0 (tiles 0, 1), with the next pass's loads in flight under it -------------------------------------------------------------------
        if (RT > 2) {
            dw_load_rows(table, kk, id[2], id[3], hvn);
        } else {
            dw_load_rows(table, kk, id_n[0], id_n[1], hvn);
            load_cand(cid_n, an);
        }
        if (RT >= 2) {
            const long long idp[2] = {id[0], id[1]};
            float xp[2];
            float* const recp[2] = {record(0), record(1)};
            dw_pass<2, ACT>(sh, actl, w, r16, kk, idp, hv, bias3, normalize != 0, inv_sqrt_k, m_run, l_run, o, xp, recp);
            xs[0] = xp[0]; xs[1] = xp[1];
        } else if (RT == 1) {
            const long long idp[1] = {id[0]};
            float xp[1];
            float* const recp[1] = {record(0)};
            dw_pass<1, ACT>(sh, actl, w, r16, kk, idp, hv, bias3, normalize != 0, inv_sqrt_k, m_run, l_run, o, xp, recp);
            xs[0] = xp[0];
        }
#pragma unroll
        for (int rt = 0; rt < 2; ++rt)
#pragma unroll
            for (int i = 0; i < 4; ++i) hv[rt][i] = hvn[rt][i];
        // ---- pass 1 (tiles 2, 3) -----------------------------------------------------------------------------------------------------------------
        if (RT > 2) {
            dw_load_rows(table, kk, id_n[0], id_n[1], hvn);
            load_cand(cid_n, an);
            if (RT >= 4) {
                const long long idp[2] = {id[2], id[3]};
                float xp[2];
                float* const recp[2] = {record(2), record(3)};
                dw_pass<2, ACT>(sh, actl, w, r16, kk, idp, hv, bias3, normalize != 0, inv_sqrt_k, m_run, l_run, o, xp, recp);
                xs[2] = xp[0]; xs[3] = xp[1];
            } else {
                const long long idp[1] = {id[2]};
                float xp[1];
                float* const recp[1] = {record(2)};
                dw_pass<1, ACT>(sh, actl, w, r16, kk, idp, hv, bias3, normalize != 0, inv_sqrt_k, m_run, l_run, o, xp, recp);
                xs[2] = xp[0];
            }
#pragma unroll
            for (int rt = 0; rt < 2; ++rt)
#pragma unroll
                for (int i = 0; i < 4; ++i) hv[rt][i] = hvn[rt][i];
        }
        // ---- pooled output: sum the 16 rows of the lane group, lane r = 0 stores the group's 16 features --------------------------------------
        const float inv_l = (normalize && l_run > 0.f) ? 1.0f / l_run : (normalize ? 0.f : 1.0f);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            float4 t4;
            t4.x = row16_sum(o[i].x) * inv_l; t4.y = row16_sum(o[i].y) * inv_l;
            t4.z = row16_sum(o[i].z) * inv_l; t4.w = row16_sum(o[i].w) * inv_l;
            if (r16 == 0) *reinterpret_cast<float4*>(out + b * DW_K + 16 * i + 4 * kk) = t4;
        }
        if (scores && kk == 0) {
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                const int j = 16 * t + r16;
                if (t < ntile_t && j < T) {
                    float v = 0.f;
                    if (t < RT) v = normalize ? (xs[t] > -INFINITY ? __expf(xs[t] - m_run) * inv_l : 0.f) : xs[t];
                    scores[b * T + j] = v;
                }
            }
        }
        // ---- advance: the next sample's rows / candidate are in hv / an; fetch the descriptor after it and a new ticket -----------------------------
        b = bn; len = len_n; cid = cid_n; toff = toff_n;
#pragma unroll
        for (int t = 0; t < 4; ++t) id[t] = id_n[t];
        bn = bt; len_n = len_t; cid_n = cid_t; toff_n = toff_t;  // its scalars were issued a sample ago
        load_ids(bn, len_n, id_n);
        bt = bq;
        load_scalars(bt, len_t, cid_t, toff_t);
        bq = bt >= 0 ? dq.take() : -1;
    }
    // ---- leave the queue record clean for the next launch that draws this slot -------------------------------------------------------------------
    dq.drain();
    if (lane == 0 && q) {
        const unsigned int done = atomicAdd(&q[DW_GROUPS], 1u);
        if (done == gridDim.x * DW_WAVES - 1) {
#pragma unroll
            for (int i = 0; i <= DW_GROUPS; ++i) atomicExch(&q[i], 0u);
        }
    }
}

static std::atomic<unsigned int> dw_next_slot{0};
// The arithmetic of the call in progress on this thread when an `_arith` entry asked for one by argument (-1: DIR_DIN_ARITH / the default).
// Set and restored inside that one call (din.hip: DwArithScope): no state survives it.
thread_local int dw_arith_override = -1;

bool din_wave_covers(int K, int T, int H1, int H2) { return K == DW_K && T <= 64 && H1 <= DW_H1P && H2 <= DW_H2P; }

int launch_din_wave(hipStream_t st, const float* table, const int64_t* hist, const int32_t* hist_len, const int64_t* cand, int T,
                    const float* W1, const float* b1, int H1, const float* W2, const float* b2, int H2, const float* W3,
                    const float* b3, int normalize, int64_t B, float* out, float* scores, const int64_t* tile_off, float* saved, int activation,
                    const float* act_params) {
    // DIR_DIN_ARITH = f16x2 (default) | bf16x3 | f32: the arithmetic of the two MFMA layers; DIR_DIN_STATIC = 1 | 0: a static stride over the
    // samples instead of the device-side queue (default: static for bf16x3 -- its per-sample time is short enough that the ticket
    // atomics cost more than the imbalance they remove, 0.45 vs 0.50 ms at config 4 -- and the queue for fp32).  Both are read per call
    // (A/B runs and tests flip them inside one process).  saved != nullptr: the training forward (z1, z2 records for the backward).
    const char* arith = getenv("DIR_DIN_ARITH");
    const int ar = dw_arith_override >= 0 ? dw_arith_override
                                          : (arith && strcmp(arith, "f32") == 0) ? 0 : (arith && strcmp(arith, "bf16x3") == 0) ? 1 : 2;      // default: fp16 x 2
    const bool bf3 = ar != 0;
    const bool save = saved != nullptr;
    typedef void (*kern_t)(const float*, const int64_t*, const int32_t*, const int64_t*, int, const float*, const float*, int, const float*,
                           const float*, int, const float*, const float*, int, long long, float*, float*, int, const int64_t*, float*, const float*);
    // [arithmetic][save | PReLU | Dice]: the training forward (SAVE) exists for the sigmoid unit only
    static const kern_t kerns[3][4] = {{&din_wave_k<DinWaveSh, false, 0>, &din_wave_k<DinWaveSh, true, 0>, &din_wave_k<DinWaveSh, false, 1>,
                                        &din_wave_k<DinWaveSh, false, 2>},
                                       {&din_wave_k<DinWaveSh3, false, 0>, &din_wave_k<DinWaveSh3, true, 0>, &din_wave_k<DinWaveSh3, false, 1>,
                                        &din_wave_k<DinWaveSh3, false, 2>},
                                       {&din_wave_k<DinWaveSh2, false, 0>, &din_wave_k<DinWaveSh2, true, 0>, &din_wave_k<DinWaveSh2, false, 1>,
                                        &din_wave_k<DinWaveSh2, false, 2>}};
    if (activation < 0 || activation > 2 || (activation != 0 && (save || !act_params)))
        return fail(DIR_E_UNSUPPORTED, "din_wave_k: activation %d (0 sigmoid, 1 PReLU, 2 Dice; the training forward covers the sigmoid unit only)", activation);
    static LdsOnce once[3][4];
    const int which = activation ? 1 + activation : (save ? 1 : 0);
    const size_t shbytes = ar == 0 ? sizeof(DinWaveSh) : ar == 1 ? sizeof(DinWaveSh3) : sizeof(DinWaveSh2);
    const size_t shmem = ((shbytes + 15) & ~(size_t)15) + (activation ? sizeof(float) * DW_ACT_FLOATS : 0);
    const kern_t kern = kerns[ar][which];
    if (!lds_limit(once[ar][which], (int)shmem, kern)) return fail(DIR_E_HIP, "din_wave_k: cannot reserve %zu B of LDS", shmem);
    // Up to DW_SLOTS launches may be in flight at once (distinct streams); a record is reused only after DW_SLOTS further launches.
    const char* stat = getenv("DIR_DIN_STATIC");
    const bool static_split = stat ? atoi(stat) != 0 : bf3;
    const int slot = static_split ? -1 : (int)(dw_next_slot.fetch_add(1) % DW_SLOTS);
    const int64_t waves_wanted = (B + 1) / 2;            // a wave should see at least a couple of samples
    int64_t nwg = (waves_wanted + DW_WAVES - 1) / DW_WAVES;
    if (nwg > kCUs) nwg = kCUs;
    if (nwg < 1) nwg = 1;
    hipLaunchKernelGGL(kern, dim3((unsigned)nwg), dim3(64 * DW_WAVES), shmem, st, table, hist, hist_len, cand, T, W1, b1, H1, W2, b2, H2, W3, b3,
                       normalize, (long long)B, out, scores, slot, tile_off, saved, act_params);
    return DIR_OK;
}

}  // namespace dir
