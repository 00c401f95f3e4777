// cin_bwd.hip -- weight gradient of one xDeepFM CIN layer on fp32 MFMA for gfx950.
//
// NO REFERENCE CODE (the reference has no CIN, README.md:28); definition: include/dir_hip.h (A14 backward),
// oracle/dir_oracle.c (orc_cin_dw_f32).  With rows r = (b,d), G = dL/dxout (the pooled gradient already
// broadcast over d by the caller):
//
//   dW[h, i*m + j] = sum_r G[r,h] * xk[r,i] * x0[r,j]
//
// GEMM view: M = kk = (i,j) (Hp*m of them), N = h, reduction over the B*D rows.  The left operand
// Z^T[kk, r] = xk[r,i]*x0[r,j] is never stored (as in the forward, csrc/cin.hip).  The other two gradients
// (d/dxk, d/dx0) ARE the forward contraction with permuted weights and are run through dir_cin_layer_f32 by the
// host mirror (autograd.py: CinLayer).
//
// No LDS, no barriers.  v_mfma_f32_32x32x2_f32: lane l holds A[row l&31][k = l>>5] and B[k = l>>5][col l&31].
// The reduction index can be paired freely between the two half-waves, so a group of 8 consecutive rows is split
// as "half 0 takes rows 0..3, half 1 takes rows 4..7": one global dwordx4 along d (contiguous in [B, C, D]
// tensors) gives a lane its operand for FOUR k-steps.  Per group of 8 rows and per wave (2 kk tiles x 4 h tiles =
// 8 accumulators): 4 loads for A (xk and x0 of both tiles) + 4 loads for B (G of the 4 column tiles) + 4 packed
// multiplies feed 32 MFMAs -- VALU/VMEM work is not hidden behind fp32 MFMAs, so the count per
// MFMA is what matters.  Loads run one group ahead of their MFMAs.
//
// Work split: the work items are (kk block of 256, group of 8 rows), linearised kk-block-major.  Each of the nwg
// workgroups (one per CU: a single full round, no partial last round) takes an equal contiguous span; a span that
// crosses into the next kk block ends one output segment and starts another.  Segment s of workgroup w goes to
// part[w][s][128 h][256 kk]; cin_dw_reduce_k finds the (w, s) pairs of a kk block analytically and adds them in
// workgroup order (bitwise reproducible).
#include "common.hpp"

namespace dir {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

// 16-byte buffer load: address = resource base + lane byte offset (VGPR) + wave-uniform byte offset (SGPR) -- the
// split this kernel needs, with no address arithmetic on the VALU; out-of-range lanes read 0 instead of faulting.
__device__ __forceinline__ float4 buf_load4(__amdgpu_buffer_rsrc_t r, uint32_t lane_off, uint32_t wave_off) {
    const auto raw = __builtin_amdgcn_raw_buffer_load_b128(r, (int)lane_off, (int)wave_off, 0);
    const f32x4 v = __builtin_bit_cast(f32x4, raw);   // (assigning the builtin's result to an int vector type splats .x)
    return make_float4(v.x, v.y, v.z, v.w);
}

constexpr int CIN_DW_NB = 4, CIN_DW_DIST = 2;   // operand ring: buffers, prefetch distance in groups of 8 rows

#ifdef CIN_DW_STAMP   // tools/cin_dw_probe.hip only: [0] shader cycles, [1] 100 MHz ticks, [2] groups, [3] waves
__device__ unsigned long long cin_dw_stamp[4];
#endif

template <int HTN>
struct CinDwOps {   // one group's operands of a lane
    float4 xk[2], x0[2], g[HTN];
};

template <int HTN /* 32-column tiles of h this launch covers: 1..4 */>
__global__ __launch_bounds__(256, 1) void cin_dw_k(const float* __restrict__ x0, const float* __restrict__ xk,
                                                   const float* __restrict__ G, int m, int Hp, int H, int hb /* first h */, int D, int dshift,
                                                   int64_t R /* B*D, a multiple of 4 */, int64_t NQ /* row groups, incl. a partial last one */,
                                                   int64_t L /* work items per workgroup */, int nslot,
                                                   float* __restrict__ part /* [nwg][nslot][128][256] per h block */) {
    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int n = lane & 31, hh = lane >> 5;
    const int Kd = Hp * m;
    const int ncb = (Kd + 255) >> 8;
    const int64_t T = (int64_t)ncb * NQ;
    int64_t lin = (int64_t)blockIdx.x * L;
    int64_t lin_end = lin + L < T ? lin + L : T;
    float* pw = part + (int64_t)blockIdx.x * nslot * (128 * 256);

    // Addressing: the 8 rows of group q are rows 8q + 4*hh + (0..3).  For D >= 8 they lie in ONE sample, so the
    // address is a wave-uniform base (sample, first d: scalar ALU, free next to MFMAs) plus a lane-constant byte offset
    // (channel, half-wave); for D = 4 the two half-waves are consecutive samples, again a lane constant.  No VALU.
    const uint32_t sx = (uint32_t)(Hp * D), s0 = (uint32_t)(m * D), sg = (uint32_t)(H * D);   // sample strides
    const uint32_t hx = D >= 8 ? 4u * hh : hh * sx, h0 = D >= 8 ? 4u * hh : hh * s0, hg = D >= 8 ? 4u * hh : hh * sg;
    uint32_t log_[HTN];
#pragma unroll
    for (int cc = 0; cc < HTN; ++cc) {
        int h = hb + 32 * cc + n;
        h = h < H ? h : H - 1;
        log_[cc] = ((uint32_t)(h * D) + hg) * 4u;
    }
    const int64_t nq_full = R >> 3;                       // groups whose 8 rows all exist
    // buffer resources over the three tensors (host: each below 2^32 bytes); 0x00020000 = raw 32-bit data format
    const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(xk), 0, (int)(uint32_t)(R * Hp * 4), 0x00020000);
    const __amdgpu_buffer_rsrc_t r0 = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(x0), 0, (int)(uint32_t)(R * m * 4), 0x00020000);
    const __amdgpu_buffer_rsrc_t rg = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(G), 0, (int)(uint32_t)(R * H * 4), 0x00020000);

#ifdef CIN_DW_STAMP
    const unsigned long long st_c0 = __builtin_readcyclecounter(), st_r0 = wall_clock64();
    const int64_t st_items = lin_end - lin;
#endif
    for (int seg = 0; lin < lin_end; ++seg) {
    const int cb = (int)(lin / NQ);
    const int64_t qa = lin - (int64_t)cb * NQ;
    int64_t qe = qa + (lin_end - lin);                    // end of this segment: the span's end or the kk block's end
    if (qe > NQ) qe = NQ;
    lin += qe - qa;
    int64_t qb = qe < nq_full ? qe : nq_full;

    // lane constants of this kk block: byte offsets of the lane's (i, j) channels
    uint32_t lox[2], lo0[2];
    int kk0[2];
#pragma unroll
    for (int t = 0; t < 2; ++t) {
        kk0[t] = wave * 64 + t * 32;                      // inside the kk block
        int kk = cb * 256 + kk0[t] + n;
        kk = kk < Kd ? kk : Kd - 1;          // clamped: rows kk >= Kd are computed from valid data and never used
        const int i = kk / m, j = kk - i * m;
        lox[t] = ((uint32_t)(i * D) + hx) * 4u;
        lo0[t] = ((uint32_t)(j * D) + h0) * 4u;
    }

    f32x16 acc[2][HTN];
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int cc = 0; cc < HTN; ++cc)
#pragma unroll
            for (int q = 0; q < 16; ++q) acc[t][cc][q] = 0.f;

    auto load_ops = [&](int64_t q, CinDwOps<HTN>& o) {
        const int64_t r8 = q * 8, bq = r8 >> dshift, dq = r8 & (int64_t)(D - 1);      // wave-uniform
        const uint32_t wx = (uint32_t)((bq * sx + dq) * 4), w0 = (uint32_t)((bq * s0 + dq) * 4), wg = (uint32_t)((bq * sg + dq) * 4);
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            o.xk[t] = buf_load4(rx, lox[t], wx);
            o.x0[t] = buf_load4(r0, lo0[t], w0);
        }
#pragma unroll
        for (int cc = 0; cc < HTN; ++cc) o.g[cc] = buf_load4(rg, log_[cc], wg);
    };
    auto run_group = [&](const CinDwOps<HTN>& o) {
        float a[2][4];
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            const f32x2 xa = {o.xk[t].x, o.xk[t].y}, xb = {o.xk[t].z, o.xk[t].w};
            const f32x2 ya = {o.x0[t].x, o.x0[t].y}, yb = {o.x0[t].z, o.x0[t].w};
            f32x2 pa, pb;
            asm("v_pk_mul_f32 %0, %1, %2" : "=v"(pa) : "v"(xa), "v"(ya));
            asm("v_pk_mul_f32 %0, %1, %2" : "=v"(pb) : "v"(xb), "v"(yb));
            a[t][0] = pa.x; a[t][1] = pa.y; a[t][2] = pb.x; a[t][3] = pb.y;
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int s = 0; s < 4; ++s) {
#pragma unroll
            for (int t = 0; t < 2; ++t) {
#pragma unroll
                for (int cc = 0; cc < HTN; ++cc) {
                    const float bv = s == 0 ? o.g[cc].x : s == 1 ? o.g[cc].y : s == 2 ? o.g[cc].z : o.g[cc].w;
                    acc[t][cc] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[t][s], bv, acc[t][cc], 0, 0, 0);
                }
            }
        }
        __builtin_amdgcn_sched_barrier(0);
    };

    {
        // NB operand buffers in a ring, loads DIST groups ahead of their MFMAs (DIST * 32 MFMAs ~ DIST * 0.9 us of
        // issue against an HBM/MALL round trip); the hot loop has no branch but its own
        constexpr int NB = CIN_DW_NB, DIST = CIN_DW_DIST;
        CinDwOps<HTN> ring[NB];
        int64_t q = qa;
        if (q + NB <= qb) {
            const int64_t ql = qb - 1;                         // prefetches past the end re-read the last group
#pragma unroll
            for (int k = 0; k < DIST; ++k) load_ops(q + k, ring[k]);
            for (; q + NB <= qb; q += NB) {
#pragma unroll
                for (int k = 0; k < NB; ++k) {
                    const int64_t qn = q + k + DIST;
                    load_ops(qn < ql ? qn : ql, ring[(k + DIST) % NB]);
                    run_group(ring[k]);
                }
            }
        }
        for (; q < qb; ++q) {   // up to NB-1 left-over groups
            load_ops(q, ring[0]);
            run_group(ring[0]);
        }
    }
    // tail: R % 8 == 4 -- the last group has rows for half-wave 0 only
    if (qe > nq_full) {   // this segment ends with the partial group
        CinDwOps<HTN> o;
        // rows R-4 .. R-1 for BOTH halves (valid addresses); half 1 contributes zero
        const int64_t r4 = R - 4, bq = r4 >> dshift, dq = r4 & (int64_t)(D - 1);
        const uint32_t wx = (uint32_t)((bq * sx + dq) * 4), w0 = (uint32_t)((bq * s0 + dq) * 4), wg = (uint32_t)((bq * sg + dq) * 4);
        // undo the half-wave part of the lane offsets
        const uint32_t cx = hh ? (D >= 8 ? 16u : sx * 4u) : 0u, c0 = hh ? (D >= 8 ? 16u : s0 * 4u) : 0u, cg = hh ? (D >= 8 ? 16u : sg * 4u) : 0u;
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            float4 v = buf_load4(rx, lox[t] - cx, wx);
            if (hh) v = make_float4(0.f, 0.f, 0.f, 0.f);
            o.xk[t] = v;
            o.x0[t] = buf_load4(r0, lo0[t] - c0, w0);
        }
#pragma unroll
        for (int cc = 0; cc < HTN; ++cc) o.g[cc] = buf_load4(rg, log_[cc] - cg, wg);
        run_group(o);
    }

    // ---- store the segment: C/D map col = lane&31 (h), row = (reg&3) + 8*(reg>>2) + 4*(lane>>5) (kk) ----------
    float* pb = pw + (int64_t)seg * (128 * 256);
#pragma unroll
    for (int t = 0; t < 2; ++t) {
#pragma unroll
        for (int cc = 0; cc < HTN; ++cc) {
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int kk = kk0[t] + 8 * g + 4 * hh;
                *reinterpret_cast<float4*>(pb + (32 * cc + n) * 256 + kk) =
                    make_float4(acc[t][cc][4 * g], acc[t][cc][4 * g + 1], acc[t][cc][4 * g + 2], acc[t][cc][4 * g + 3]);
            }
        }
    }
    }   // segments
#ifdef CIN_DW_STAMP
    if (lane == 0) {
        atomicAdd(&cin_dw_stamp[0], __builtin_readcyclecounter() - st_c0);
        atomicAdd(&cin_dw_stamp[1], wall_clock64() - st_r0);
        atomicAdd(&cin_dw_stamp[2], (unsigned long long)st_items);
        atomicAdd(&cin_dw_stamp[3], 1ULL);
    }
#endif
}

// dW[h][kk] (+)= sum of the segments that cover kk's block, in workgroup order
__global__ __launch_bounds__(256) void cin_dw_reduce_k(const float* __restrict__ part, int nwg, int nslot, int64_t NQ, int64_t L,
                                                       int hb /* first h of this launch's block */, int nh /* its valid columns */, int Kd,
                                                       int accumulate, float* __restrict__ dW) {
    const int64_t n = (int64_t)nh * Kd;
    const int64_t T = (int64_t)((Kd + 255) >> 8) * NQ;
    for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < n; e += (int64_t)gridDim.x * 256) {
        const int hl = (int)(e / Kd), kk = (int)(e - (int64_t)hl * Kd);
        const int cb = kk >> 8, kl = kk & 255;
        float* dst = dW + ((int64_t)(hb + hl) * Kd + kk);
        const int64_t lo = (int64_t)cb * NQ, hi = lo + NQ - 1;          // this block's work items (NQ >= 1)
        const int w0 = (int)(lo / L);
        int w1 = (int)(hi / L);
        const int wl = (int)((T - 1) / L);
        if (w1 > wl) w1 = wl;
        float s = accumulate ? *dst : 0.f;
        for (int w = w0; w <= w1; ++w) {
            const int slot = cb - (int)(((int64_t)w * L) / NQ);
            s += part[((int64_t)w * nslot + slot) * (128 * 256) + hl * 256 + kl];
        }
        *dst = s;
    }
}

// ------------------------------------------------------------------------------------------------------------------
// Data gradients of one CIN layer, both from ONE pass:
//   T_j[r, i]  = sum_h G[r,h] * W[h, i*m + j]          plain GEMM per j: A = G (row tile x H), B = W_j (H x Hp)
//   dxk[r, i] += x0[r, j] * T_j[r, i]                   FMA epilogue, accumulated over j in registers
//   dx0[r, j]  = sum_i xk[r, i] * T_j[r, i]             reduction over the columns = over the 32 lanes of a half-wave
// A is stationary: a wave keeps its 32-row slice of G in HT/2 registers (lane = row l&31, k parity l>>5) for the
// whole kernel; W_j streams through LDS, double buffered, one barrier per j (HT/2 * CT MFMAs per wave).  The host
// passes W already permuted to the LDS image Wp[j][h][n][cc] = W[h, (32*cc + n)*m + j] (zero for 32*cc + n >= Hp),
// so staging is 16-byte loads and 16-byte LDS stores at the same offsets and a lane's CT B operands of a k-step are one
// ds_read_b128.  MFMA count = rows/32 * ceil(Hp/32) * m * HT/2: no padding of the 26 fields (the forward kernel
// run on permuted weights pads the 26 output columns to 32 and needs two passes: 21 ms -> ~7 ms per 128-wide layer).
// Limits (the host mirror falls back to the forward-kernel formulation otherwise): H <= 128, Hp <= 128, m <= 64.
// ------------------------------------------------------------------------------------------------------------------
// Sums of 16 per-lane values over the 32 lanes of each half-wave.  v_permlane16_swap (gfx950) exchanges the odd rows of
// one register with the even rows of another, so ONE add folds the two 16-lane rows of a half for TWO values at once:
// afterwards rows 0/2 carry values 0..7 and rows 1/3 values 8..15, and only 8 registers go through the in-row DPP steps.
__device__ __forceinline__ void half32_sum16(const float (&v)[16], float (&u)[8]) {
#pragma unroll
    for (int q = 0; q < 8; ++q) {
        // inline asm: with this compiler the builtin's two results fold to the same register when both feed one add
        float a = v[q], b = v[q + 8];
        asm volatile("s_nop 1\n\tv_permlane16_swap_b32 %0, %1\n\ts_nop 1" : "+v"(a), "+v"(b));
        u[q] = row16_sum(a + b);
    }
}

template <int V>
struct CinInt { static constexpr int value = V; };

template <int CT /* tile slots per column index in the W image (the host's layout): 1, 2, 4 */, int CC /* column tiles COMPUTED: <= CT */,
          int HT /* rows of H slice 0: 32, 64, 96, 128 */, int HL /* rows of slice 1 when NHC == 2 (else == HT) */,
          bool FULLH /* H fills every slice: the W staging needs no per-element predicate (49 exec-mask branches per j otherwise) */,
          int NHC /* slices of H: 1, or 2 for 128 < H <= 256 (HT = 128) */>
__global__ __launch_bounds__(256, 1) void cin_dx_k(const float* __restrict__ x0, const float* __restrict__ xk,
                                                   const float* __restrict__ Wp /* [column block][m][H][32][CT] */,
                                                   const float* __restrict__ G, int m, int Hp, int H, int D, int dshift,
                                                   int yoff /* first column block of this launch */, int multi /* > 1 column block in total */,
                                                   int ks_last /* k-steps of the LAST slice that touch rows h < H (even) */,
                                                   int64_t R, float* __restrict__ dxk, float* __restrict__ dx0) {
    static_assert(CC <= CT && (NHC == 2 || HL == HT) && (NHC == 1 || HT == 128), "cin_dx_k shape");
    constexpr int WJ = HT * 32 * CT;           // floats of one LDS buffer (slice 0 is the larger slice)
    constexpr int NGA = HT / 2 + (NHC == 2 ? HL / 2 : 0);
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* Wb = smem;                          // [2][WJ]
    float* x0s = smem + 2 * WJ;                // [m][128]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, n = lane & 31, hh = lane >> 5;
    const int64_t row0 = (int64_t)blockIdx.x * 128 + wave * 32;     // this wave's row tile
    const int by = (int)blockIdx.y + yoff;
    const int cbase = by * (32 * CT);          // first column (i) of this column block
    Wp += (int64_t)by * m * H * (32 * CT);
    auto slice_floats = [&](int hc) {          // valid floats of slice hc of a W_j image
        int rows = H - hc * HT;
        const int cap = hc == 0 ? HT : HL;
        rows = rows < 0 ? 0 : (rows > cap ? cap : rows);
        return rows * 32 * CT;
    };

    // ---- one-time loads ---------------------------------------------------------------------------------------------
    // A operand: ga[s] = G[row0 + n, h = 2s + hh]  (slice 1's k-steps follow slice 0's: h = HT + 2(s - HT/2) + hh = 2s + hh)
    float ga[NGA];
    {
        const int64_t r = row0 + n;
        const bool ok = r < R;
        const int64_t rc = ok ? r : R - 1;
        const float* src = G + ((rc >> dshift) * H) * D + (rc & (D - 1));
#pragma unroll
        for (int s = 0; s < NGA; ++s) {
            const int h = 2 * s + hh;
            ga[s] = (ok && h < H) ? src[(int64_t)h * D] : 0.f;
        }
    }
    // C/D rows of this lane: rr(reg) = (reg&3) + 8*(reg>>2) + 4*hh; 4 consecutive regs = 4 consecutive rows (same sample)
    float xkv[CC][16];
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        const int64_t r = row0 + 8 * g + 4 * hh;
        const bool ok = r < R;
        const int64_t rc = ok ? r : R - 4;
        const float* src = xk + ((rc >> dshift) * Hp) * D + (rc & (D - 1));
#pragma unroll
        for (int cc = 0; cc < CC; ++cc) {
            const int i = cbase + 32 * cc + n;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (ok && i < Hp) v = *reinterpret_cast<const float4*>(src + (int64_t)i * D);
            xkv[cc][4 * g] = v.x; xkv[cc][4 * g + 1] = v.y; xkv[cc][4 * g + 2] = v.z; xkv[cc][4 * g + 3] = v.w;
        }
    }
    // x0 slice of the workgroup's 128 rows, [j][row]
    for (int e = tid; e < m * 128; e += 256) {
        const int j = e >> 7, rl = e & 127;
        const int64_t r = (int64_t)blockIdx.x * 128 + rl;
        x0s[e] = r < R ? x0[((r >> dshift) * m + j) * D + (r & (D - 1))] : 0.f;
    }
    // zero, once, the rows the staging never writes: buffer b always holds slice b when NHC == 2, slice 0 otherwise
#pragma unroll
    for (int bq = 0; bq < 2; ++bq) {
        const int hc = NHC == 2 ? bq : 0;
        const int cap = (hc == 0 ? HT : HL) * 32 * CT;
        for (int e = slice_floats(hc) + tid; e < cap; e += 256) Wb[bq * WJ + e] = 0.f;
    }

    // The next slice is staged in NP parts (fewer live registers; the second part's loads sit mid-way in the MFMA stream).
    // Per slice: NV float4 per thread; two parts when that count is even.
    constexpr int NV0 = HT * CT / 32, NV1 = HL * CT / 32;
    constexpr int NP0 = (NV0 >= 2 && NV0 % 2 == 0) ? 2 : 1, NP1 = (NV1 >= 2 && NV1 % 2 == 0) ? 2 : 1;
    constexpr int NVP0 = NV0 / NP0, NVP1 = NV1 / NP1, NVPM = NVP0 > NVP1 ? NVP0 : NVP1;
    float4 wst[NVPM];
    auto w_load = [&](int j, auto HC, int part) {
        constexpr int hc = decltype(HC)::value;
        constexpr int NVP = hc == 0 ? NVP0 : NVP1;
        const float4* src = reinterpret_cast<const float4*>(Wp + ((int64_t)j * H + hc * HT) * (32 * CT));
        const int wr = slice_floats(hc);
#pragma unroll
        for (int q = 0; q < NVP; ++q) {
            const int e4 = tid + 256 * (part * NVP + q);
            wst[q] = (FULLH || e4 * 4 < wr) ? src[e4] : make_float4(0.f, 0.f, 0.f, 0.f);
        }
    };
    auto w_store = [&](int buf, auto HC, int part) {
        constexpr int hc = decltype(HC)::value;
        constexpr int NVP = hc == 0 ? NVP0 : NVP1;
        float4* dst = reinterpret_cast<float4*>(Wb + buf * WJ);
        const int wr = slice_floats(hc);
#pragma unroll
        for (int q = 0; q < NVP; ++q) {
            const int e4 = tid + 256 * (part * NVP + q);
            if (FULLH || e4 * 4 < wr) dst[e4] = wst[q];
        }
    };
#pragma unroll
    for (int part = 0; part < NP0; ++part) {
        w_load(0, CinInt<0>{}, part);
        w_store(0, CinInt<0>{}, part);
    }
    __syncthreads();

    f32x16 dk[CC];
#pragma unroll
    for (int cc = 0; cc < CC; ++cc)
#pragma unroll
        for (int q = 0; q < 16; ++q) dk[cc][q] = 0.f;
    // dx0 store targets of this lane (lanes with (lane & 15) == 0 store two runs of 4 rows per j): element offsets for j = 0,
    // computed once -- inside the j loop the address is this plus j * D
    int64_t dx0_off[2];
    bool dx0_ok[2];
    {
        const int gsel = (lane >> 4) & 1;
#pragma unroll
        for (int e = 0; e < 2; ++e) {
            const int64_t r = row0 + 8 * (2 * gsel + e) + 4 * hh;
            dx0_ok[e] = r < R;
            dx0_off[e] = ((r >> dshift) * m) * D + (r & (D - 1));
        }
    }

    for (int j = 0; j < m; ++j) {
        f32x16 T[CC];
        // one H slice of stage j: its k-steps, with the NEXT stage's W slice staged around them
        auto do_slice = [&](auto HC) {
            constexpr int hc = decltype(HC)::value;
            constexpr int KS = (hc == 0 ? HT : HL) / 2;             // k-steps of this slice
            constexpr int S0 = hc == 0 ? 0 : HT / 2;                // its first k-step in ga[]
            constexpr int nhc = hc + 1 < NHC ? hc + 1 : 0;          // the stage after (j, hc)
            constexpr int NPn = nhc == 0 ? NP0 : NP1;
            const int buf = NHC == 2 ? hc : (j & 1);
            const int nj = hc + 1 < NHC ? j : j + 1;
            const bool more = nj < m;
            if (more) w_load(nj, CinInt<nhc>{}, 0);
            const float* wb = Wb + buf * WJ + (hh * 32 + n) * CT;      // [h][n][cc]: h = 2s + hh within the slice
            // bursts of 2 k-steps = 2*CC MFMAs; operand reads run one burst ahead (cin.hip)
            float bw[2][CC];
            auto read_b = [&](int s, float (&o)[CC]) {
                const float* src = wb + s * (2 * 32 * CT);
                if (CT == 4) {
                    const float4 v = *reinterpret_cast<const float4*>(src);
                    o[0] = v.x;
                    if (CC > 1) o[1 % CC] = v.y;
                    if (CC > 2) o[2 % CC] = v.z;
                    if (CC > 3) o[3 % CC] = v.w;
                } else if (CT == 2) {
                    const float2 v = *reinterpret_cast<const float2*>(src);
                    o[0] = v.x;
                    if (CC > 1) o[1 % CC] = v.y;
                } else {
                    o[0] = src[0];
                }
            };
            read_b(0, bw[0]);
            read_b(1, bw[1]);
#pragma unroll
            for (int s = 0; s < KS; s += 2) {
                // rows h >= H of the last slice are zero in both operands: their k-steps are skipped (a wave-uniform branch per
                // burst, after the mid-way staging point so that the next slice is always staged)
                if (hc == NHC - 1 && s > (KS / 4) * 2 && s >= ks_last) break;
                float bn[2][CC];
                if (s + 2 < KS) {
                    read_b(s + 2, bn[0]);
                    read_b(s + 3, bn[1]);
                } else {
#pragma unroll
                    for (int cc = 0; cc < CC; ++cc) { bn[0][cc] = bw[0][cc]; bn[1][cc] = bw[1][cc]; }
                }
                if (NPn == 2 && s == (KS / 4) * 2 && more) {   // mid-way: first part to LDS, second part's loads
                    w_store(buf ^ 1, CinInt<nhc>{}, 0);
                    w_load(nj, CinInt<nhc>{}, 1);
                }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int e = 0; e < 2; ++e)
#pragma unroll
                    for (int cc = 0; cc < CC; ++cc) {
                        if (hc == 0 && s + e == 0) {
                            f32x16 z;
#pragma unroll
                            for (int q = 0; q < 16; ++q) z[q] = 0.f;
                            T[cc] = __builtin_amdgcn_mfma_f32_32x32x2f32(ga[0], bw[0][cc], z, 0, 0, 0);
                        } else {
                            T[cc] = __builtin_amdgcn_mfma_f32_32x32x2f32(ga[S0 + s + e], bw[e][cc], T[cc], 0, 0, 0);
                        }
                    }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int cc = 0; cc < CC; ++cc) { bw[0][cc] = bn[0][cc]; bw[1][cc] = bn[1][cc]; }
            }
            if (hc == NHC - 1) {
                // ---- epilogue of j --------------------------------------------------------------------------------------
                float x0v[16];
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const float4 v = *reinterpret_cast<const float4*>(x0s + j * 128 + wave * 32 + 8 * g + 4 * hh);
                    x0v[4 * g] = v.x; x0v[4 * g + 1] = v.y; x0v[4 * g + 2] = v.z; x0v[4 * g + 3] = v.w;
                }
                float p[16], u[8];
#pragma unroll
                for (int q = 0; q < 16; ++q) {
                    float acc = T[0][q] * xkv[0][q];
#pragma unroll
                    for (int cc = 1; cc < CC; ++cc) acc = __builtin_fmaf(T[cc][q], xkv[cc][q], acc);
                    p[q] = acc;
#pragma unroll
                    for (int cc = 0; cc < CC; ++cc) dk[cc][q] = __builtin_fmaf(x0v[q], T[cc][q], dk[cc][q]);
                }
                half32_sum16(p, u);
                if ((lane & 15) == 0) {   // rows 0/2 hold the totals of C/D regs 0..7, rows 1/3 of regs 8..15: 2 runs of 4 rows (d) each
#pragma unroll
                    for (int e = 0; e < 2; ++e)
                        if (dx0_ok[e]) {
                            float* dst = dx0 + dx0_off[e] + (int64_t)j * D;
                            if (multi) {
#pragma unroll
                                for (int q = 0; q < 4; ++q) atomicAdd(dst + q, u[4 * e + q]);
                            } else {
                                *reinterpret_cast<float4*>(dst) = make_float4(u[4 * e], u[4 * e + 1], u[4 * e + 2], u[4 * e + 3]);
                            }
                        }
                }
            }
            if (more) w_store(buf ^ 1, CinInt<nhc>{}, NPn - 1);
            __syncthreads();
        };
        do_slice(CinInt<0>{});
        if constexpr (NHC == 2) do_slice(CinInt<1>{});
    }

    // ---- dxk: C/D map col = lane&31 (i within the column tile), rows as above ------------------------------------------
#pragma unroll
    for (int cc = 0; cc < CC; ++cc) {
        const int i = cbase + 32 * cc + n;
        if (i >= Hp) continue;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const int64_t r = row0 + 8 * g + 4 * hh;
            if (r < R)
                *reinterpret_cast<float4*>(dxk + ((r >> dshift) * Hp + i) * D + (r & (D - 1))) =
                    make_float4(dk[cc][4 * g], dk[cc][4 * g + 1], dk[cc][4 * g + 2], dk[cc][4 * g + 3]);
        }
    }
}

struct CinDxArgs {
    const float *x0, *xk, *Wp, *G;
    int m, Hp, H, D, dshift, multi, ks_last;
    int64_t R;
    float *dxk, *dx0;
    size_t shmem;
    hipStream_t st;
};

template <int CT, int CC, int HT, int HL, int NHC>
static void launch_cin_dx(const CinDxArgs& a, dim3 grid, int yoff) {
    static LdsOnce once;
    (void)lds_limit(once, 160 * 1024, &cin_dx_k<CT, CC, HT, HL, true, NHC>, &cin_dx_k<CT, CC, HT, HL, false, NHC>);
    if (a.H == (NHC == 2 ? HT + HL : HT))
        hipLaunchKernelGGL((cin_dx_k<CT, CC, HT, HL, true, NHC>), grid, dim3(256), a.shmem, a.st, a.x0, a.xk, a.Wp, a.G, a.m, a.Hp, a.H, a.D, a.dshift, yoff,
                           a.multi, a.ks_last, a.R, a.dxk, a.dx0);
    else
        hipLaunchKernelGGL((cin_dx_k<CT, CC, HT, HL, false, NHC>), grid, dim3(256), a.shmem, a.st, a.x0, a.xk, a.Wp, a.G, a.m, a.Hp, a.H, a.D, a.dshift, yoff,
                           a.multi, a.ks_last, a.R, a.dxk, a.dx0);
}

// ht: rows of slice 0 (a multiple of 32), hl: rows of slice 1 (nhc == 2) -- both rounded up to the instantiated 32 / 64 / 96 / 128
template <int CT, int CC>
static void launch_cin_dx_cc(const CinDxArgs& a, int ht, int hl, int nhc, dim3 grid, int yoff) {
    if (nhc == 2) {
        if (hl == 32) launch_cin_dx<CT, CC, 128, 32, 2>(a, grid, yoff);
        else if (hl == 64) launch_cin_dx<CT, CC, 128, 64, 2>(a, grid, yoff);
        else if (hl == 96) launch_cin_dx<CT, CC, 128, 96, 2>(a, grid, yoff);
        else launch_cin_dx<CT, CC, 128, 128, 2>(a, grid, yoff);
    } else if (ht == 32) launch_cin_dx<CT, CC, 32, 32, 1>(a, grid, yoff);
    else if (ht == 64) launch_cin_dx<CT, CC, 64, 64, 1>(a, grid, yoff);
    else if (ht == 96) launch_cin_dx<CT, CC, 96, 96, 1>(a, grid, yoff);
    else launch_cin_dx<CT, CC, 128, 128, 1>(a, grid, yoff);
}

struct CinDwPlan { int nwg, nslot; int64_t NQ, L; };
static CinDwPlan cin_dw_plan(int m, int Hp, int H, int D, int64_t B) {
    CinDwPlan p;
    const int64_t Kd = (int64_t)Hp * m;
    const int64_t ncb = (Kd + 255) / 256;
    p.NQ = (B * D + 7) / 8;
    if (p.NQ < 1) p.NQ = 1;
    const int64_t T = ncb * p.NQ;
    int64_t nwg = kCUs;                                           // one resident workgroup per CU, a single round per h block
    if (nwg < 1) nwg = 1;
    if (nwg > T) nwg = T;
    p.L = (T + nwg - 1) / nwg;
    p.nwg = (int)((T + p.L - 1) / p.L);                           // no empty workgroups
    p.nslot = (int)((p.L + p.NQ - 1) / p.NQ) + 1;                 // kk blocks a span can touch
    return p;
}

}  // namespace dir

using namespace dir;

extern "C" int64_t dir_cin_dw_workspace_bytes(int m, int Hp, int H, int D, int64_t B) {
    if (m <= 0 || Hp <= 0 || H <= 0 || D <= 0 || B < 0) return 0;
    const CinDwPlan p = cin_dw_plan(m, Hp, H, D, B);
    return (int64_t)p.nwg * p.nslot * 128 * 256 * (int64_t)sizeof(float);       // one h block at a time (blocks run back to back)
}

extern "C" int dir_cin_dw_f32(const float* x0, const float* xk, const float* G, int m, int Hp, int H, int D, int64_t B,
                              int accumulate, float* dW, void* workspace, dir_stream_t stream) {
    DIR_CHECK_ARG(x0 && xk && G && dW && workspace, "dir_cin_dw_f32: null pointer");
    DIR_CHECK_ARG(m > 0 && Hp > 0 && H > 0 && D > 0 && B >= 0, "dir_cin_dw_f32: m=%d Hp=%d H=%d D=%d", m, Hp, H, D);
    if (!(D == 4 || D == 8 || D == 16 || D == 32)) return fail(DIR_E_UNSUPPORTED, "dir_cin_dw_f32: D=%d (supported: 4, 8, 16, 32)", D);
    if (!(aligned16(x0) && aligned16(xk) && aligned16(G) && aligned16(workspace)))
        return fail(DIR_E_BADARG, "dir_cin_dw_f32: x0 / xk / G / workspace must be 16-byte aligned");
    const int64_t cmax = Hp > H ? (Hp > m ? Hp : m) : (H > m ? H : m);
    if (B * D * cmax * 4 >= ((int64_t)1 << 32)) return fail(DIR_E_UNSUPPORTED, "dir_cin_dw_f32: x0 / xk / G must each stay below 4 GiB (32-bit buffer offsets)");
    const int64_t Kd = (int64_t)Hp * m, n = (int64_t)H * Kd;
    if (Kd >= ((int64_t)1 << 30)) return fail(DIR_E_UNSUPPORTED, "dir_cin_dw_f32: Hp*m too large");
    hipStream_t st = as_stream(stream);
    if (B == 0) {
        if (!accumulate && zero_async(dW, n * sizeof(float), st) != hipSuccess) return fail(DIR_E_HIP, "dir_cin_dw_f32: memset failed");
        return DIR_OK;
    }
    int dshift = 0;
    while ((1 << dshift) < D) ++dshift;
    const int64_t R = B * D;
    const CinDwPlan p = cin_dw_plan(m, Hp, H, D, B);
    // h blocks of up to four 32-column tiles, one after the other through the same workspace (each block's reduce runs before the
    // next block's partial sums are written: stream order); the block kernel is instantiated per tile count, so H = 200 costs
    // 4 + 3 tiles and H = 64 two, not a padded 128 each.
    float* ws = static_cast<float*>(workspace);
    for (int hb = 0; hb < H; hb += 128) {
        const int nh = H - hb < 128 ? H - hb : 128;
        const dim3 grid((unsigned)p.nwg);
        switch ((nh + 31) / 32) {
            case 1: hipLaunchKernelGGL(cin_dw_k<1>, grid, dim3(256), 0, st, x0, xk, G, m, Hp, H, hb, D, dshift, R, p.NQ, p.L, p.nslot, ws); break;
            case 2: hipLaunchKernelGGL(cin_dw_k<2>, grid, dim3(256), 0, st, x0, xk, G, m, Hp, H, hb, D, dshift, R, p.NQ, p.L, p.nslot, ws); break;
            case 3: hipLaunchKernelGGL(cin_dw_k<3>, grid, dim3(256), 0, st, x0, xk, G, m, Hp, H, hb, D, dshift, R, p.NQ, p.L, p.nslot, ws); break;
            default: hipLaunchKernelGGL(cin_dw_k<4>, grid, dim3(256), 0, st, x0, xk, G, m, Hp, H, hb, D, dshift, R, p.NQ, p.L, p.nslot, ws); break;
        }
        DIR_CHECK_LAUNCH("cin_dw");
        hipLaunchKernelGGL(cin_dw_reduce_k, dim3(grid_for(((int64_t)nh * Kd + 255) / 256)), dim3(256), 0, st, ws, p.nwg, p.nslot, p.NQ, p.L,
                           hb, nh, (int)Kd, accumulate, dW);
        DIR_CHECK_LAUNCH("cin_dw_reduce");
    }
    return DIR_OK;
}

extern "C" int dir_cin_dx_f32(const float* x0, const float* xk, const float* Wp, const float* G, int m, int Hp, int H, int D,
                              int64_t B, float* dxk, float* dx0, dir_stream_t stream) {
    DIR_CHECK_ARG(x0 && xk && Wp && G && dxk && dx0, "dir_cin_dx_f32: null pointer");
    DIR_CHECK_ARG(m > 0 && Hp > 0 && H > 0 && D > 0 && B >= 0, "dir_cin_dx_f32: m=%d Hp=%d H=%d D=%d", m, Hp, H, D);
    if (!(D == 4 || D == 8 || D == 16 || D == 32)) return fail(DIR_E_UNSUPPORTED, "dir_cin_dx_f32: D=%d (supported: 4, 8, 16, 32)", D);
    if (H > 256 || Hp > 256 || m > 64)
        return fail(DIR_E_UNSUPPORTED, "dir_cin_dx_f32: needs H <= 256, Hp <= 256, m <= 64 (H=%d Hp=%d m=%d): use the forward formulation", H, Hp, m);
    if (!(aligned16(x0) && aligned16(xk) && aligned16(Wp) && aligned16(G) && aligned16(dxk) && aligned16(dx0)))
        return fail(DIR_E_BADARG, "dir_cin_dx_f32: all tensors must be 16-byte aligned");
    if (B == 0) return DIR_OK;
    int dshift = 0;
    while ((1 << dshift) < D) ++dshift;
    const int64_t R = B * D;
    const int ct = Hp <= 32 ? 1 : Hp <= 64 ? 2 : 4;           // tile slots of the host's W image
    const int ncb = (Hp + 32 * ct - 1) / (32 * ct);            // column blocks: 2 when 128 < Hp <= 256
    int cc_last = (Hp - (ncb - 1) * 32 * ct + 31) / 32;        // column tiles the last block really has
    if (ct == 4 && cc_last < 3) cc_last = 3;                   // instantiated: 3 or 4 of a 4-slot image
    const int nhc = H > 128 ? 2 : 1;
    const int ht = nhc == 2 ? 128 : ((H + 31) / 32) * 32;      // slice rows follow H in steps of 32: H = 200 -> 128 + 96, not 2 x 128
    const int hl = nhc == 2 ? ((H - 128 + 31) / 32) * 32 : ht;
    hipStream_t st = as_stream(stream);
    if (ncb > 1 && zero_async(dx0, sizeof(float) * (size_t)B * m * D, st) != hipSuccess)
        return fail(DIR_E_HIP, "dir_cin_dx_f32: memset failed");          // the two column blocks ADD their dx0 shares
    const int rows_last = nhc == 2 ? H - 128 : H;
    CinDxArgs a{x0, xk, Wp, G, m, Hp, H, D, dshift, ncb > 1 ? 1 : 0, ((rows_last + 3) / 4) * 2, R, dxk, dx0,
                sizeof(float) * (2 * (size_t)ht * 32 * ct + (size_t)m * 128), st};
    const unsigned rows = (unsigned)((R + 127) / 128);
    if (ct == 1) launch_cin_dx_cc<1, 1>(a, ht, hl, nhc, dim3(rows, 1), 0);
    else if (ct == 2) launch_cin_dx_cc<2, 2>(a, ht, hl, nhc, dim3(rows, 1), 0);
    else {
        const int nfull = cc_last == 4 ? ncb : ncb - 1;        // blocks computing all four tiles
        if (nfull > 0) launch_cin_dx_cc<4, 4>(a, ht, hl, nhc, dim3(rows, (unsigned)nfull), 0);
        if (nfull < ncb) launch_cin_dx_cc<4, 3>(a, ht, hl, nhc, dim3(rows, 1), nfull);
    }
    DIR_CHECK_LAUNCH("cin_dx");
    return DIR_OK;
}
