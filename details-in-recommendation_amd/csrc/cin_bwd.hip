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

struct CinDwOps {   // one group's operands of a lane
    float4 xk[2], x0[2], g[4];
};

__global__ __launch_bounds__(256, 1) void cin_dw_k(const float* __restrict__ x0, const float* __restrict__ xk,
                                                   const float* __restrict__ G, int m, int Hp, int H, int D, int dshift,
                                                   int64_t R /* B*D, a multiple of 4 */, int64_t NQ /* row groups, incl. a partial last one */,
                                                   int64_t L /* work items per workgroup */, int nslot,
                                                   float* __restrict__ part /* [nwg][nslot][128][256] per h block */) {
    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int n = lane & 31, hh = lane >> 5;
    const int Kd = Hp * m;
    const int ncb = (Kd + 255) >> 8;
    const int hb = blockIdx.y * 128;
    const int64_t T = (int64_t)ncb * NQ;
    int64_t lin = (int64_t)blockIdx.x * L;
    int64_t lin_end = lin + L < T ? lin + L : T;
    float* pw = part + ((int64_t)blockIdx.y * gridDim.x + blockIdx.x) * nslot * (128 * 256);

    // Addressing: the 8 rows of group q are rows 8q + 4*hh + (0..3).  For D >= 8 they lie in ONE sample, so the
    // address is a wave-uniform base (sample, first d: scalar ALU, free next to MFMAs) plus a lane-constant byte offset
    // (channel, half-wave); for D = 4 the two half-waves are consecutive samples, again a lane constant.  No VALU.
    const uint32_t sx = (uint32_t)(Hp * D), s0 = (uint32_t)(m * D), sg = (uint32_t)(H * D);   // sample strides
    const uint32_t hx = D >= 8 ? 4u * hh : hh * sx, h0 = D >= 8 ? 4u * hh : hh * s0, hg = D >= 8 ? 4u * hh : hh * sg;
    uint32_t log_[4];
#pragma unroll
    for (int cc = 0; cc < 4; ++cc) {
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

    f32x16 acc[2][4];
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int cc = 0; cc < 4; ++cc)
#pragma unroll
            for (int q = 0; q < 16; ++q) acc[t][cc][q] = 0.f;

    auto load_ops = [&](int64_t q, CinDwOps& o) {
        const int64_t r8 = q * 8, bq = r8 >> dshift, dq = r8 & (int64_t)(D - 1);      // wave-uniform
        const uint32_t wx = (uint32_t)((bq * sx + dq) * 4), w0 = (uint32_t)((bq * s0 + dq) * 4), wg = (uint32_t)((bq * sg + dq) * 4);
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            o.xk[t] = buf_load4(rx, lox[t], wx);
            o.x0[t] = buf_load4(r0, lo0[t], w0);
        }
#pragma unroll
        for (int cc = 0; cc < 4; ++cc) o.g[cc] = buf_load4(rg, log_[cc], wg);
    };
    auto run_group = [&](const CinDwOps& o) {
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
                for (int cc = 0; cc < 4; ++cc) {
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
        CinDwOps ring[NB];
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
        CinDwOps o;
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
        for (int cc = 0; cc < 4; ++cc) o.g[cc] = buf_load4(rg, log_[cc] - cg, wg);
        run_group(o);
    }

    // ---- store the segment: C/D map col = lane&31 (h), row = (reg&3) + 8*(reg>>2) + 4*(lane>>5) (kk) ----------
    float* pb = pw + (int64_t)seg * (128 * 256);
#pragma unroll
    for (int t = 0; t < 2; ++t) {
#pragma unroll
        for (int cc = 0; cc < 4; ++cc) {
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
                                                       int H, int Kd, int accumulate, float* __restrict__ dW) {
    const int64_t n = (int64_t)H * Kd;
    const int64_t T = (int64_t)((Kd + 255) >> 8) * NQ;
    for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < n; e += (int64_t)gridDim.x * 256) {
        const int h = (int)(e / Kd), kk = (int)(e - (int64_t)h * Kd);
        const int cb = kk >> 8, kl = kk & 255;
        const int hbk = h >> 7, hl = h & 127;
        const int64_t lo = (int64_t)cb * NQ, hi = lo + NQ - 1;          // this block's work items (NQ >= 1)
        const int w0 = (int)(lo / L);
        int w1 = (int)(hi / L);
        const int wl = (int)((T - 1) / L);
        if (w1 > wl) w1 = wl;
        float s = accumulate ? dW[e] : 0.f;
        for (int w = w0; w <= w1; ++w) {
            const int slot = cb - (int)(((int64_t)w * L) / NQ);
            s += part[(((int64_t)hbk * nwg + w) * nslot + slot) * (128 * 256) + hl * 256 + kl];
        }
        dW[e] = s;
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

template <int CT /* column tiles per column block: 1, 2, 4 */, int HT /* rows of one H slice: 32, 64, 128 */,
          bool FULLH /* H == NHC * HT: the W staging needs no per-element predicate (49 exec-mask branches per j otherwise) */,
          int NHC /* slices of H: 1, or 2 for 128 < H <= 256 (HT = 128) */>
__global__ __launch_bounds__(256, 1) void cin_dx_k(const float* __restrict__ x0, const float* __restrict__ xk,
                                                   const float* __restrict__ Wp /* [column block][m][H][32][CT] */,
                                                   const float* __restrict__ G, int m, int Hp, int H, int D, int dshift,
                                                   int64_t R, float* __restrict__ dxk, float* __restrict__ dx0) {
    constexpr int KS = HT / 2;                 // k-steps per slice
    constexpr int WJ = HT * 32 * CT;           // floats of one W slice image (rows h >= H stay zero)
    constexpr int NV = WJ / 4 / 256;           // float4 per thread per slice
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* Wb = smem;                          // [2][WJ]
    float* x0s = smem + 2 * WJ;                // [m][128]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, n = lane & 31, hh = lane >> 5;
    const int64_t row0 = (int64_t)blockIdx.x * 128 + wave * 32;     // this wave's row tile
    const int cbase = blockIdx.y * (32 * CT);  // first column (i) of this column block
    const bool multi = gridDim.y > 1;          // two column blocks: each adds its dx0 contribution (exactly two addends: order-free)
    Wp += (int64_t)blockIdx.y * m * H * (32 * CT);
    auto slice_floats = [&](int hc) {          // valid floats of slice hc of a W_j image
        int rows = H - hc * HT;
        rows = rows < 0 ? 0 : (rows > HT ? HT : rows);
        return rows * 32 * CT;
    };

    // ---- one-time loads ---------------------------------------------------------------------------------------------
    // A operand: ga[hc*KS + s] = G[row0 + n, h = hc*HT + 2s + hh]
    float ga[NHC * KS];
    {
        const int64_t r = row0 + n;
        const bool ok = r < R;
        const int64_t rc = ok ? r : R - 1;
        const float* src = G + ((rc >> dshift) * H) * D + (rc & (D - 1));
#pragma unroll
        for (int s = 0; s < NHC * KS; ++s) {
            const int h = 2 * s + hh;
            ga[s] = (ok && h < H) ? src[(int64_t)h * D] : 0.f;
        }
    }
    // C/D rows of this lane: rr(reg) = (reg&3) + 8*(reg>>2) + 4*hh; 4 consecutive regs = 4 consecutive rows (same sample)
    float xkv[CT][16];
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        const int64_t r = row0 + 8 * g + 4 * hh;
        const bool ok = r < R;
        const int64_t rc = ok ? r : R - 4;
        const float* src = xk + ((rc >> dshift) * Hp) * D + (rc & (D - 1));
#pragma unroll
        for (int cc = 0; cc < CT; ++cc) {
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
    for (int bq = 0; bq < 2; ++bq)
        for (int e = slice_floats(NHC == 2 ? bq : 0) + tid; e < WJ; e += 256) Wb[bq * WJ + e] = 0.f;

    // the next slice is staged in NP parts (fewer live registers; the second part's loads sit mid-way in the MFMA stream)
    constexpr int NP = NV >= 2 ? 2 : 1, NVP = NV / NP;
    float4 wst[NVP];
    auto w_load = [&](int j, int hc, int part) {
        const float4* src = reinterpret_cast<const float4*>(Wp + ((int64_t)j * H + hc * HT) * (32 * CT));
        const int wr = slice_floats(hc);
#pragma unroll
        for (int q = 0; q < NVP; ++q) {
            const int e4 = tid + 256 * (part * NVP + q);
            wst[q] = (FULLH || e4 * 4 < wr) ? src[e4] : make_float4(0.f, 0.f, 0.f, 0.f);
        }
    };
    auto w_store = [&](int buf, int hc, int part) {
        float4* dst = reinterpret_cast<float4*>(Wb + buf * WJ);
        const int wr = slice_floats(hc);
#pragma unroll
        for (int q = 0; q < NVP; ++q) {
            const int e4 = tid + 256 * (part * NVP + q);
            if (FULLH || e4 * 4 < wr) dst[e4] = wst[q];
        }
    };
#pragma unroll
    for (int part = 0; part < NP; ++part) {
        w_load(0, 0, part);
        w_store(0, 0, part);
    }
    __syncthreads();

    f32x16 dk[CT];
#pragma unroll
    for (int cc = 0; cc < CT; ++cc)
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
        f32x16 T[CT];
#pragma unroll
        for (int hc = 0; hc < NHC; ++hc) {
            const int buf = NHC == 2 ? hc : (j & 1);
            // the stage after (j, hc)
            const int nj = hc + 1 < NHC ? j : j + 1, nhc = hc + 1 < NHC ? hc + 1 : 0;
            const bool more = nj < m;
            if (more) w_load(nj, nhc, 0);
            const float* wb = Wb + buf * WJ + (hh * 32 + n) * CT;      // [h][n][cc]: h = 2s + hh within the slice
            // bursts of 2 k-steps = 2*CT MFMAs; operand reads run one burst ahead (cin.hip)
            float bw[2][CT];
            auto read_b = [&](int s, float (&o)[CT]) {
                const float* src = wb + s * (2 * 32 * CT);
                if (CT == 4) {
                    const float4 v = *reinterpret_cast<const float4*>(src);
                    o[0] = v.x; o[1 % CT] = v.y; o[2 % CT] = v.z; o[3 % CT] = v.w;
                } else if (CT == 2) {
                    const float2 v = *reinterpret_cast<const float2*>(src);
                    o[0] = v.x; o[1 % CT] = v.y;
                } else {
                    o[0] = src[0];
                }
            };
            read_b(0, bw[0]);
            read_b(1, bw[1]);
#pragma unroll
            for (int s = 0; s < KS; s += 2) {
                float bn[2][CT];
                if (s + 2 < KS) {
                    read_b(s + 2, bn[0]);
                    read_b(s + 3, bn[1]);
                } else {
#pragma unroll
                    for (int cc = 0; cc < CT; ++cc) { bn[0][cc] = bw[0][cc]; bn[1][cc] = bw[1][cc]; }
                }
                if (NP == 2 && s == (KS / 4) * 2 && more) {   // mid-way: first part to LDS, second part's loads
                    w_store(buf ^ 1, nhc, 0);
                    w_load(nj, nhc, 1);
                }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int e = 0; e < 2; ++e)
#pragma unroll
                    for (int cc = 0; cc < CT; ++cc) {
                        if (hc == 0 && s + e == 0) {
                            f32x16 z;
#pragma unroll
                            for (int q = 0; q < 16; ++q) z[q] = 0.f;
                            T[cc] = __builtin_amdgcn_mfma_f32_32x32x2f32(ga[0], bw[0][cc], z, 0, 0, 0);
                        } else {
                            T[cc] = __builtin_amdgcn_mfma_f32_32x32x2f32(ga[hc * KS + s + e], bw[e][cc], T[cc], 0, 0, 0);
                        }
                    }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int cc = 0; cc < CT; ++cc) { bw[0][cc] = bn[0][cc]; bw[1][cc] = bn[1][cc]; }
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
                    for (int cc = 1; cc < CT; ++cc) acc = __builtin_fmaf(T[cc][q], xkv[cc][q], acc);
                    p[q] = acc;
#pragma unroll
                    for (int cc = 0; cc < CT; ++cc) dk[cc][q] = __builtin_fmaf(x0v[q], T[cc][q], dk[cc][q]);
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
            if (more) w_store(buf ^ 1, nhc, NP - 1);
            __syncthreads();
        }
    }

    // ---- dxk: C/D map col = lane&31 (i within the column tile), rows as above ------------------------------------------
#pragma unroll
    for (int cc = 0; cc < CT; ++cc) {
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

template <int CT, int HT, int NHC>
static void launch_cin_dx(dim3 grid, size_t shmem, hipStream_t st, const float* x0, const float* xk, const float* Wp,
                          const float* G, int m, int Hp, int H, int D, int dshift, int64_t R, float* dxk, float* dx0) {
    static bool set = false;
    if (!set) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&cin_dx_k<CT, HT, true, NHC>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&cin_dx_k<CT, HT, false, NHC>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        set = true;
    }
    if (H == HT * NHC)
        hipLaunchKernelGGL((cin_dx_k<CT, HT, true, NHC>), grid, dim3(256), shmem, st, x0, xk, Wp, G, m, Hp, H, D, dshift, R, dxk, dx0);
    else
        hipLaunchKernelGGL((cin_dx_k<CT, HT, false, NHC>), grid, dim3(256), shmem, st, x0, xk, Wp, G, m, Hp, H, D, dshift, R, dxk, dx0);
}

template <int CT>
static void launch_cin_dx_ct(int ht, int nhc, dim3 grid, size_t shmem, hipStream_t st, const float* x0, const float* xk, const float* Wp,
                             const float* G, int m, int Hp, int H, int D, int dshift, int64_t R, float* dxk, float* dx0) {
    if (nhc == 2) launch_cin_dx<CT, 128, 2>(grid, shmem, st, x0, xk, Wp, G, m, Hp, H, D, dshift, R, dxk, dx0);
    else if (ht == 32) launch_cin_dx<CT, 32, 1>(grid, shmem, st, x0, xk, Wp, G, m, Hp, H, D, dshift, R, dxk, dx0);
    else if (ht == 64) launch_cin_dx<CT, 64, 1>(grid, shmem, st, x0, xk, Wp, G, m, Hp, H, D, dshift, R, dxk, dx0);
    else launch_cin_dx<CT, 128, 1>(grid, shmem, st, x0, xk, Wp, G, m, Hp, H, D, dshift, R, dxk, dx0);
}

struct CinDwPlan { int nwg, nslot, nhb; int64_t NQ, L; };
static CinDwPlan cin_dw_plan(int m, int Hp, int H, int D, int64_t B) {
    CinDwPlan p;
    const int64_t Kd = (int64_t)Hp * m;
    const int64_t ncb = (Kd + 255) / 256;
    p.nhb = (H + 127) / 128;
    p.NQ = (B * D + 7) / 8;
    if (p.NQ < 1) p.NQ = 1;
    const int64_t T = ncb * p.NQ;
    int64_t nwg = kCUs / p.nhb;                                   // one resident workgroup per CU, a single round
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
    return (int64_t)p.nhb * p.nwg * p.nslot * 128 * 256 * (int64_t)sizeof(float);
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
        if (!accumulate && hipMemsetAsync(dW, 0, n * sizeof(float), st) != hipSuccess) return fail(DIR_E_HIP, "dir_cin_dw_f32: memset failed");
        return DIR_OK;
    }
    int dshift = 0;
    while ((1 << dshift) < D) ++dshift;
    const int64_t R = B * D;
    const CinDwPlan p = cin_dw_plan(m, Hp, H, D, B);
    hipLaunchKernelGGL(cin_dw_k, dim3((unsigned)p.nwg, (unsigned)p.nhb), dim3(256), 0, st, x0, xk, G, m, Hp, H, D, dshift, R,
                       p.NQ, p.L, p.nslot, static_cast<float*>(workspace));
    DIR_CHECK_LAUNCH("cin_dw");
    hipLaunchKernelGGL(cin_dw_reduce_k, dim3(grid_for((n + 255) / 256)), dim3(256), 0, st, static_cast<const float*>(workspace),
                       p.nwg, p.nslot, p.NQ, p.L, H, (int)Kd, accumulate, dW);
    DIR_CHECK_LAUNCH("cin_dw_reduce");
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
    const int ct = Hp <= 32 ? 1 : Hp <= 64 ? 2 : 4;
    const int ncb = (Hp + 32 * ct - 1) / (32 * ct);            // column blocks: 2 when 128 < Hp <= 256
    const int nhc = H > 128 ? 2 : 1;
    const int ht = nhc == 2 ? 128 : (H <= 32 ? 32 : H <= 64 ? 64 : 128);
    const size_t shmem = sizeof(float) * (2 * (size_t)ht * 32 * ct + (size_t)m * 128);
    dim3 grid((unsigned)((R + 127) / 128), (unsigned)ncb);
    hipStream_t st = as_stream(stream);
    if (ncb > 1 && hipMemsetAsync(dx0, 0, sizeof(float) * (size_t)B * m * D, st) != hipSuccess)
        return fail(DIR_E_HIP, "dir_cin_dx_f32: memset failed");          // the two column blocks ADD their dx0 shares
    if (ct == 1) launch_cin_dx_ct<1>(ht, nhc, grid, shmem, st, x0, xk, Wp, G, m, Hp, H, D, dshift, R, dxk, dx0);
    else if (ct == 2) launch_cin_dx_ct<2>(ht, nhc, grid, shmem, st, x0, xk, Wp, G, m, Hp, H, D, dshift, R, dxk, dx0);
    else launch_cin_dx_ct<4>(ht, nhc, grid, shmem, st, x0, xk, Wp, G, m, Hp, H, D, dshift, R, dxk, dx0);
    DIR_CHECK_LAUNCH("cin_dx");
    return DIR_OK;
}
