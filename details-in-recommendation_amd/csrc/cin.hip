// cin.hip -- xDeepFM compressed-interaction-network layer on fp32 MFMA for gfx950.
//
// NO REFERENCE CODE: /root/reference/README.md:28 only links arXiv:1803.05170 (eq. 6).  Definition:
// include/dir_hip.h / oracle/dir_oracle.c (orc_cin_layer_f32).
//
//   xout[b,h,d] = sum_{i<Hp} sum_{j<m} W[h, i*m+j] * xk[b,i,d] * x0[b,j,d]
//
// GEMM view: rows r = (b,d) (B*D of them), columns h, reduction kk = (i,j) of length Hp*m.  The
// left operand Z[r, kk] = xk[b,i,d] * x0[b,j,d] would be B*D x Hp*m floats (14 GB per layer at
// B = 65536); it is never stored: each lane forms its one A element per MFMA with a single v_mul.
//
// v_mfma_f32_32x32x2_f32 (exact fp32, k-ordered fma chain): lane l holds A[row l&31][k = l>>5] and
// B[k = l>>5][col l&31]; the two k of one instruction are the field pair j = 2*jp + (l>>5).
// A wave owns RT = 2 row tiles (64 rows = 4 samples at D = 16) x CT <= 4 column tiles (all H <= 128)
// = 8 accumulators (128 VGPRs).  A 256-thread workgroup (one wave per SIMD) owns 256 rows.  H's 32-column tiles are cut into
// 4-tile blocks plus one smaller block (dir_cin_layer_f32: 200 columns = 4 + 3 tiles, one launch per block size); a 3-tile block
// stages the 4-slot image and computes three tiles; a 1-tile block (H <= 32) has a quarter-size image and runs TWO workgroups
// per CU, so that one's staging and operand reads issue under the other's MFMAs.
//
// LDS (one array, 16-byte aligned carve):
//   x0s [mp][256]            the workgroup's x0 slice, transposed so that lanes read consecutive rows
//   Ws  [2][IC*mp][32][CS]   W chunk of IC values of i, transposed to [kk][n][cc] with h = 32*cc + n (CS = 1, 2 or 4 slots): the four
//                            column-tile operands of a lane are ONE ds_read_b128 (LDS reads and VALU work are
//                            not hidden behind fp32 MFMAs -- tools/mfma_probe2.hip -- so their count matters)
//   xks [2][IC][256]         xk chunk
// Chunk c+1 is fetched from global into registers before chunk c's MFMAs and written to the other LDS
// buffer after them: one barrier per chunk of IC*mp/2*RT*CT = 416 MFMAs.
#include "common.hpp"

namespace dir {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

// Diagnostic build only (DIR_CIN_STAMP=1): per-chunk cycle shares, never used by the product path.
// [0] cycles from chunk start to the end of its MFMA stream, [1] cycles from there to past the barrier,
// [2] chunks counted, [3] cycles of the prologue (x0 staging + first chunk), [4] epilogue cycles, [5] workgroups
__device__ unsigned long long cin_stamp_acc[8];
__device__ __forceinline__ unsigned long long cin_now() {
    unsigned long long t;
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)::"memory");
    return t;
}

constexpr int CIN_RT = 2;          // row tiles per wave
constexpr int CIN_ROWS = 256;      // rows per workgroup = 4 waves * RT * 32
constexpr int cin_ws(int CT) { return 32 * (CT == 3 ? 4 : CT); }   // floats per kk row of the LDS W image: [32 n][1 | 2 | 4 cc]
constexpr int CIN_KS = 2;          // k-steps per MFMA burst (one VALU/LDS cluster per burst)
constexpr int cin_ic(int MT) { return MT > 26 ? 2 : 4; }   // i values per chunk (LDS: 2*IC*mp*129*4 B of W)

template <int MT /* field count padded to an instantiated size: register arrays, unrolling */, int CT /* column tiles */,
          bool FP /* interleaved fast staging; needs m == MT */, bool STAMP = false /* diagnostic cycle stamps */,
          int ICT = 0 /* i values per chunk; 0: cin_ic(MT) */>
__global__ __launch_bounds__(256, CT == 1 ? 2 : 1) void cin_k(const float* __restrict__ x0, const float* __restrict__ xk,
                                                const float* __restrict__ W, int m /* actual fields, <= MT */, int Hp, int H,
                                                int D, int dshift, int hoff /* first output column of this launch */,
                                                int64_t R /* B*D */,
                                                float* __restrict__ xout,
                                                float* __restrict__ pooled, int64_t pooled_ld) {
    if (FP) m = MT;   // the fast path is only selected when m == MT: keep the divisions compile-time there
    constexpr int mp = (MT + 1) & ~1;
    constexpr int MP2 = mp / 2;
    constexpr int IC = ICT > 0 ? ICT : cin_ic(MT);
    constexpr int CS = CT == 3 ? 4 : CT;  // column tiles STAGED: the W image holds 4 tile slots per column index, so a 3-tile block
                                          // stages like a 4-tile one (same contiguous runs, same b128 operand reads) and computes 3
    constexpr int WS = cin_ws(CT);
    constexpr int WCH = IC * mp * WS;     // floats per W buffer
    constexpr int XCH = IC * CIN_ROWS;    // floats per xk buffer
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* x0s = smem;                       // [mp][256]
    float* Ws = x0s + mp * CIN_ROWS;         // [2][WCH]
    float* xks = Ws + 2 * WCH;               // [2][XCH]

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int n = lane & 31;
    const int hh = lane >> 5;
    unsigned long long st_a = 0, st_b = 0, st_pro = 0, st_t0 = 0, st_cnt = 0;
    if (STAMP) st_t0 = cin_now();
    const int64_t row0 = (int64_t)blockIdx.x * CIN_ROWS;   // first (b,d) row of this workgroup
    const int hbase = hoff + blockIdx.y * (32 * CT);       // first output column of this workgroup
    const int Kd = Hp * m;

    // ---- stage x0 slice: x0s[j][r] = x0[b, j, d] with (b,d) = row0 + r; thread tid owns row r = tid.  All loads
    // are issued before the first store (and before chunk 0's loads below) so that the prologue pays one memory
    // round trip, not one per field.
    float x0v[mp];
    {
        const int64_t gr = row0 + tid;
        const bool ok = gr < R;
        const int64_t grc = ok ? gr : R - 1;
        const float* src = x0 + ((grc >> dshift) * m) * D + (grc & (D - 1));
#pragma unroll
        for (int j = 0; j < mp; ++j) x0v[j] = (j < m && ok) ? src[(int64_t)j * D] : 0.f;
    }

    // per-thread staging registers for the next chunk
    constexpr int WE = (IC * MT * 32 * CS + 255) / 256;   // W elements per thread per chunk
    constexpr int XE = XCH / 256;                         // xk elements per thread per chunk (= IC)
    float wreg[WE];
    float xreg[XE];

    auto fetch_chunk = [&](int c) {
        const int i0 = c * IC;
#pragma unroll
        for (int q = 0; q < WE; ++q) {
            const int e = tid + 256 * q;           // e = hl * (IC*m) + kkl   (runtime m: generic path)
            const int hl = e / (IC * m);
            const int kkl = e - hl * (IC * m);
            const int il = kkl / m;
            const int h = hbase + hl;
            float v = 0.f;
            if (hl < 32 * CS && h < H && i0 + il < Hp) v = W[(int64_t)h * Kd + (int64_t)i0 * m + kkl];
            wreg[q] = v;
        }
#pragma unroll
        for (int q = 0; q < XE; ++q) {
            const int e = tid + 256 * q;           // e = il * 256 + r
            const int il = e >> 8, r = e & 255;
            const int64_t gr = row0 + r;
            float v = 0.f;
            if (i0 + il < Hp && gr < R) {
                const int64_t b = gr >> dshift;
                const int d = (int)(gr & (D - 1));
                v = xk[(b * Hp + (i0 + il)) * D + d];
            }
            xreg[q] = v;
        }
    };
    // store the staged registers of chunk c+1 into LDS buffer `buf`; `part` selects one IC-th of the W elements
    // (all parts when part < 0) so that the stores can be interleaved between the MFMA blocks of chunk c
    auto store_chunk = [&](int buf, int part) {
        float* wb = Ws + buf * WCH;
        constexpr int WPP = (WE + IC - 1) / IC;
#pragma unroll
        for (int q = 0; q < WE; ++q) {
            if (part >= 0 && q / WPP != part) continue;
            const int e = tid + 256 * q;
            const int hl = e / (IC * m);
            const int kkl = e - hl * (IC * m);
            const int il = kkl / m;
            const int j = kkl - il * m;
            if (hl < 32 * CS) wb[(il * mp + j) * WS + (hl & 31) * CS + (hl >> 5)] = wreg[q];
        }
        if (part < 0 || part == IC - 1) {
            float* xb = xks + buf * XCH;
#pragma unroll
            for (int q = 0; q < XE; ++q) xb[tid + 256 * q] = xreg[q];
        }
    };

    // ---- fast staging of the NEXT chunk, spread over the k-steps of the current one -------------------------
    // TPR threads share one W row of the chunk (IC*m contiguous floats in global memory); a thread owns RL
    // consecutive floats of it: 16-byte global loads, and LDS destinations that differ from a per-thread base by
    // compile-time offsets only (no per-element address arithmetic inside the MFMA stream).
    constexpr int TPR = 256 / (32 * CS);
    constexpr bool FAST = FP;                          // the host only selects FP when cin_fast_shape<MT, CT>() holds
    constexpr int RL = FAST ? (IC * MT) / TPR : 4;
    constexpr bool VEC4 = FAST && (RL % 4 == 0) && ((IC * MT) % 4 == 0);
    constexpr bool VEC2 = FAST && !VEC4 && (RL % 2 == 0) && ((IC * MT) % 2 == 0);   // 8-byte loads (IC = 2 at m = 26)
    constexpr int NLD = VEC4 ? RL / 4 : (VEC2 ? RL / 2 : RL);            // load slots
    static_assert(!FAST || RL <= WE, "staging registers are shared with the generic path");
    const int srow = tid / TPR, spart = tid - srow * TPR;
    const bool srow_ok = hbase + srow < H;
    // loads are unconditional from clamped (always valid) addresses.  No select is needed: a W row h >= H only
    // feeds output column h and an xk row >= R only feeds output row r, neither of which is ever stored.
    const float* wsrc = W + (int64_t)(srow_ok ? hbase + srow : 0) * Kd + spart * RL;
    const int sil0 = (spart * RL) / MT, sj0 = spart * RL - sil0 * MT;
    const int wdst0 = (sil0 * mp + sj0) * WS + (srow & 31) * CS + (srow >> 5);
    constexpr bool fast_ok = FAST;
    auto fast_i0 = [&](int c) {   // first i of chunk c, clamped so that the loads stay inside W (a clamped
        const int i0 = c * IC;    // chunk is re-staged by the generic path afterwards)
        return i0 + IC <= Hp ? i0 : Hp - IC;
    };
    auto fast_load = [&](int slot, int i0) {
        const float* src = wsrc + (int64_t)i0 * MT;
        if (VEC4) {
            const float4 v = *reinterpret_cast<const float4*>(src + 4 * slot);
            wreg[4 * slot] = v.x; wreg[4 * slot + 1] = v.y; wreg[4 * slot + 2] = v.z; wreg[4 * slot + 3] = v.w;
        } else if (VEC2) {
            const float2 v = *reinterpret_cast<const float2*>(src + 2 * slot);
            wreg[2 * slot] = v.x; wreg[2 * slot + 1] = v.y;
        } else {
            wreg[slot] = src[slot];
        }
    };
    const int64_t xrow = (row0 + tid < R) ? row0 + tid : R - 1;          // XE == IC: element q is (il = q, r = tid)
    const float* xsrc = xk + ((xrow >> dshift) * Hp) * D + (xrow & (D - 1));
    auto fast_load_xk = [&](int i0) {
#pragma unroll
        for (int q = 0; q < XE; ++q) xreg[q] = xsrc[(int64_t)(i0 + q) * D];
    };
    auto fast_store = [&](int idx, int buf) {   // idx < RL: W element; RL <= idx < RL + XE: xk element
        if (idx < RL) {
            const int off = (RL % MT == 0) ? ((idx / MT) * mp + (idx % MT)) * WS : idx * WS;
            Ws[buf * WCH + wdst0 + off] = wreg[idx];
        } else if (idx < RL + XE) {
            xks[buf * XCH + tid + 256 * (idx - RL)] = xreg[idx - RL];
        }
    };

    // zero the pad rows j in [m, mp) of both W buffers once (m < MT, or odd MT); the staging never writes them
    if (mp != m) {
        const int npad = mp - m;
        for (int e = tid; e < 2 * IC * npad * WS; e += 256) {
            const int buf = e / (IC * npad * WS);
            int rem = e - buf * (IC * npad * WS);
            const int il = rem / (npad * WS);
            rem -= il * (npad * WS);
            const int jp_ = rem / WS, hl = rem - jp_ * WS;
            Ws[buf * WCH + (il * mp + m + jp_) * WS + hl] = 0.f;
        }
    }

    const int nchunk = (Hp + IC - 1) / IC;
    auto store_x0 = [&]() {
#pragma unroll
        for (int j = 0; j < mp; ++j) x0s[j * CIN_ROWS + tid] = x0v[j];
    };
    if (FAST) {   // chunk 0 through the same contiguous-run staging (the host guarantees Hp >= IC on this path)
#pragma unroll
        for (int slot = 0; slot < NLD; ++slot) fast_load(slot, 0);
        fast_load_xk(0);
        store_x0();
#pragma unroll
        for (int idx = 0; idx < RL + XE; ++idx) fast_store(idx, 0);
    } else {
        fetch_chunk(0);
        store_x0();
        store_chunk(0, -1);
    }
    __syncthreads();

    // this lane's x0 operands: x0p[jp][t] = x0s[2*jp + hh][wave*64 + t*32 + n]
    f32x2 x0p[MP2];   // {row tile 0, row tile 1}: register pairs for the packed multiply
#pragma unroll
    for (int jp = 0; jp < MP2; ++jp) {
        const float* src = x0s + (2 * jp + hh) * CIN_ROWS + wave * 64 + n;
        x0p[jp] = f32x2{src[0], src[32]};
    }

    f32x16 acc[CIN_RT][CT];
#pragma unroll
    for (int t = 0; t < CIN_RT; ++t)
#pragma unroll
        for (int c = 0; c < CT; ++c)
#pragma unroll
            for (int q = 0; q < 16; ++q) acc[t][c][q] = 0.f;

    if (STAMP) st_pro = cin_now() - st_t0;
    for (int c = 0; c < nchunk; ++c) {
        unsigned long long st_c0 = 0;
        if (STAMP) {
            __builtin_amdgcn_sched_barrier(0);
            st_c0 = cin_now();
            __builtin_amdgcn_sched_barrier(0);
        }
        const int buf = c & 1;
        const bool more = c + 1 < nchunk;
        // the next chunk is staged by the interleaved fast path when it is a whole chunk; a partial (last) or
        // irregular one by the generic bulk path after this chunk's MFMAs
        const bool next_bulk = more && (!fast_ok || (c + 2) * IC > Hp);
        const int i0n = fast_ok ? fast_i0(more ? c + 1 : c) : 0;
        const float* wb = Ws + buf * WCH + hh * WS + n * CS;
        const float* xb = xks + buf * XCH + wave * 64 + n;
        // One burst = KS k-steps = KS*RT*CT MFMAs issued back to back.  Everything else of those steps -- the LDS
        // operand reads for the NEXT burst (latency ~100 cycles vs >= 1000 cycles of MFMA issue), a few staging
        // instructions of the next chunk and the KS*RT products forming this burst's A operands -- sits in ONE
        // cluster in front of the burst: a VALU or LDS instruction in an fp32 MFMA stream costs ~7.5 issue cycles
        // (nothing hides behind v_mfma_f32_32x32x2_f32 except scalar instructions and s_nop), so their COUNT is
        // what is minimised.  Keep the burst sequence ONE straight-line block: with control flow inside the chunk
        // loop the compiler copies all 128 accumulator registers between AGPRs and VGPRs on every edge.
        constexpr int NS = IC * MP2, KS = (NS % CIN_KS == 0) ? CIN_KS : 1, NG = NS / KS;
        float bw[KS][CT], xkv[KS][CIN_RT];
        auto read_ops = [&](int g, float (&bo)[KS][CT], float (&xo)[KS][CIN_RT], const float (&xprev)[CIN_RT]) {
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
                const int s_ = g * KS + ks, il = s_ / MP2, jp = s_ - il * MP2;
                const float* src = wb + (il * mp + 2 * jp) * WS;
                if (CT == 4) {
                    const float4 v = *reinterpret_cast<const float4*>(src);
                    bo[ks][0] = v.x; bo[ks][1 % CT] = v.y; bo[ks][2 % CT] = v.z; bo[ks][3 % CT] = v.w;
                } else if (CT == 3) {
                    const float4 v = *reinterpret_cast<const float4*>(src);
                    bo[ks][0] = v.x; bo[ks][1 % CT] = v.y; bo[ks][2 % CT] = v.z;
                } else if (CT == 2) {
                    const float2 v = *reinterpret_cast<const float2*>(src);
                    bo[ks][0] = v.x; bo[ks][1 % CT] = v.y;
                } else {
                    bo[ks][0] = src[0];
                }
#pragma unroll
                for (int t = 0; t < CIN_RT; ++t)
                    xo[ks][t] = (jp == 0) ? xb[il * CIN_ROWS + t * 32] : (ks == 0 ? xprev[t] : xo[ks - 1][t]);
            }
        };
        {
            const float none[CIN_RT] = {0.f, 0.f};
            read_ops(0, bw, xkv, none);
        }
#pragma unroll
        for (int g = 0; g < NG; ++g) {
            float bwn[KS][CT], xkn[KS][CIN_RT];
            if (g + 1 < NG) {
                read_ops(g + 1, bwn, xkn, xkv[KS - 1]);
            } else {
#pragma unroll
                for (int ks = 0; ks < KS; ++ks) {
#pragma unroll
                    for (int cc = 0; cc < CT; ++cc) bwn[ks][cc] = bw[ks][cc];
#pragma unroll
                    for (int t = 0; t < CIN_RT; ++t) xkn[ks][t] = xkv[ks][t];
                }
            }
            if (FAST && fast_ok) {   // a few staging instructions of the next chunk per k-step
                constexpr int SS = NS / 2;
                constexpr int SPS = (RL + XE + (NS - SS) - 1) / (NS - SS);
#pragma unroll
                for (int ks = 0; ks < KS; ++ks) {
                    const int step = g * KS + ks;
                    if (step < NLD) fast_load(step, i0n);
                    if (step == (NLD < SS ? NLD : SS - 1)) fast_load_xk(i0n);
                    if (step >= SS) {
#pragma unroll
                        for (int k = 0; k < SPS; ++k) fast_store((step - SS) * SPS + k, buf ^ 1);
                    }
                }
            }
            float a[KS][CIN_RT];
            static_assert(CIN_RT == 2, "the A products of the two row tiles are one packed multiply");
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {   // v_pk_mul_f32: same IEEE products, half the VALU instructions
                const int jp = (g * KS + ks) % MP2;
                const f32x2 xv = {xkv[ks][0], xkv[ks][1]};
                f32x2 pr;
                asm("v_pk_mul_f32 %0, %1, %2" : "=v"(pr) : "v"(xv), "v"(x0p[jp]));
                a[ks][0] = pr.x; a[ks][1] = pr.y;
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int ks = 0; ks < KS; ++ks)
#pragma unroll
                for (int t = 0; t < CIN_RT; ++t)
#pragma unroll
                    for (int cc = 0; cc < CT; ++cc)
                        acc[t][cc] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[ks][t], bw[ks][cc], acc[t][cc], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
#pragma unroll
                for (int cc = 0; cc < CT; ++cc) bw[ks][cc] = bwn[ks][cc];
#pragma unroll
                for (int t = 0; t < CIN_RT; ++t) xkv[ks][t] = xkn[ks][t];
            }
        }
        unsigned long long st_c1 = 0;
        if (STAMP) {
            __builtin_amdgcn_sched_barrier(0);
            st_c1 = cin_now();
            __builtin_amdgcn_sched_barrier(0);
        }
        if (next_bulk) {
            fetch_chunk(c + 1);
            store_chunk(buf ^ 1, -1);
        }
        __syncthreads();
        if (STAMP) {
            __builtin_amdgcn_sched_barrier(0);
            const unsigned long long st_c2 = cin_now();
            st_a += st_c1 - st_c0;
            st_b += st_c2 - st_c1;
            ++st_cnt;
        }
    }
    unsigned long long st_e0 = 0;
    if (STAMP) st_e0 = cin_now();

    // ---- epilogue: C/D map col = lane&31, row = (reg&3) + 8*(reg>>2) + 4*(lane>>5) -------------------
    if (xout) {   // NULL: the caller only wants the pooled sums (last layer of a CIN stack)
#pragma unroll
    for (int t = 0; t < CIN_RT; ++t) {
        const int64_t trow = row0 + wave * 64 + t * 32;
#pragma unroll
        for (int cc = 0; cc < CT; ++cc) {
            const int h = hbase + 32 * cc + n;
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int64_t gr = trow + 8 * g + 4 * hh;   // first of 4 consecutive rows
                if (h < H && gr < R) {
                    const int64_t b = gr >> dshift;
                    const int d = (int)(gr & (D - 1));
                    float4 v = make_float4(acc[t][cc][4 * g], acc[t][cc][4 * g + 1], acc[t][cc][4 * g + 2], acc[t][cc][4 * g + 3]);
                    *reinterpret_cast<float4*>(xout + (b * H + h) * D + d) = v;
                }
            }
        }
    }
    }
    if (pooled) {
        // sum over d: the D rows of one sample are whole g groups (8 rows each, 4 per lane half)
#pragma unroll
        for (int t = 0; t < CIN_RT; ++t) {
            const int64_t trow = row0 + wave * 64 + t * 32;
#pragma unroll
            for (int cc = 0; cc < CT; ++cc) {
                const int h = hbase + 32 * cc + n;
                const float p0 = (acc[t][cc][0] + acc[t][cc][1]) + (acc[t][cc][2] + acc[t][cc][3]);
                const float p1 = (acc[t][cc][4] + acc[t][cc][5]) + (acc[t][cc][6] + acc[t][cc][7]);
                const float p2 = (acc[t][cc][8] + acc[t][cc][9]) + (acc[t][cc][10] + acc[t][cc][11]);
                const float p3 = (acc[t][cc][12] + acc[t][cc][13]) + (acc[t][cc][14] + acc[t][cc][15]);
                if (D == 4) {  // every (g, half) is its own sample
                    const float pv[4] = {p0, p1, p2, p3};
#pragma unroll
                    for (int g = 0; g < 4; ++g) {
                        const int64_t gr = trow + 8 * g + 4 * hh;
                        if (h < H && gr < R) pooled[(gr >> dshift) * pooled_ld + h] = pv[g];
                    }
                } else {
                    // combine g groups of one sample, then the two lane halves
                    float s0, s1, s2, s3;   // sums for the (up to) 4 samples of this tile
                    if (D == 8) { s0 = p0; s1 = p1; s2 = p2; s3 = p3; }
                    else if (D == 16) { s0 = p0 + p1; s1 = p2 + p3; s2 = 0.f; s3 = 0.f; }
                    else { s0 = (p0 + p1) + (p2 + p3); s1 = 0.f; s2 = 0.f; s3 = 0.f; }
                    s0 += __shfl_xor(s0, 32, 64);
                    s1 += __shfl_xor(s1, 32, 64);
                    s2 += __shfl_xor(s2, 32, 64);
                    s3 += __shfl_xor(s3, 32, 64);
                    const int ns = 32 >> dshift;   // samples per tile: 4, 2, 1
                    const float sv[4] = {s0, s1, s2, s3};
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const int64_t gr = trow + (int64_t)q * D;
                        if (q < ns && hh == 0 && h < H && gr < R) pooled[(gr >> dshift) * pooled_ld + h] = sv[q];
                    }
                }
            }
        }
    }
    if (STAMP && lane == 0) {
        atomicAdd(&cin_stamp_acc[0], st_a);
        atomicAdd(&cin_stamp_acc[1], st_b);
        atomicAdd(&cin_stamp_acc[2], st_cnt);
        atomicAdd(&cin_stamp_acc[3], st_pro);
        atomicAdd(&cin_stamp_acc[4], cin_now() - st_e0);
        atomicAdd(&cin_stamp_acc[5], 1ULL);
    }
}

template <int MT, int CT, int IC = 0>
constexpr bool cin_fast_shape() {   // a thread's contiguous run of a W row must be whole 16-byte (or 8-byte) loads
    constexpr int TPR = 256 / (32 * (CT == 3 ? 4 : CT));
    constexpr int N = (IC > 0 ? IC : cin_ic(MT)) * MT;
    constexpr int RL = N / TPR;
    // (a 1-tile block takes the interleaved staging with scalar loads too: its two workgroups per CU hide them)
    return (N % TPR == 0) && ((RL % MT == 0) || (MT % RL == 0)) && ((RL % 4 == 0 && N % 4 == 0) || (RL % 2 == 0 && N % 2 == 0) || CT == 1);
}

template <int MT, int CT, bool FP>
static void launch_cin_one(dim3 grid, size_t shmem, hipStream_t st, const float* x0, const float* xk, const float* W,
                           int m, int Hp, int H, int D, int dshift, int hoff, int64_t R, float* xout, float* pooled,
                           int64_t pooled_ld) {
    static LdsOnce once;
    (void)lds_limit(once, 160 * 1024, &cin_k<MT, CT, FP>);
    hipLaunchKernelGGL((cin_k<MT, CT, FP>), grid, dim3(256), shmem, st, x0, xk, W, m, Hp, H, D, dshift, hoff, R, xout, pooled, pooled_ld);
}

template <int MT, int CT>
static void launch_cin_ct(dim3 grid, size_t shmem, hipStream_t st, const float* x0, const float* xk, const float* W,
                          int m, int Hp, int H, int D, int dshift, int hoff, int64_t R, float* xout, float* pooled,
                          int64_t pooled_ld) {
    // interleaved fast staging needs: a compatible (m, column-tile) shape, whole 16-byte aligned W rows and at
    // least one whole chunk; everything else takes the generic bulk staging
    static const int fast_env = getenv("DIR_CIN_FAST") ? atoi(getenv("DIR_CIN_FAST")) : 1;
    const bool wvec = m == MT && ((int64_t)Hp * MT) % 4 == 0 && aligned16(W);
    if constexpr (MT == 26 && CT == 4) {
        static const int stamp_env = getenv("DIR_CIN_STAMP") ? atoi(getenv("DIR_CIN_STAMP")) : 0;
        if (stamp_env && fast_env && wvec && Hp >= cin_ic(MT)) {   // diagnostic build: cycle stamps, same arithmetic
            static LdsOnce once;
            (void)lds_limit(once, 160 * 1024, &cin_k<26, 4, true, true>);
            hipLaunchKernelGGL((cin_k<26, 4, true, true>), grid, dim3(256), shmem, st, x0, xk, W, m, Hp, H, D, dshift, hoff, R, xout, pooled, pooled_ld);
            return;
        }
    }
    // Hp a multiple of 2 but not of the chunk size (layer 1 of the BASELINE stack: Hp = m = 26): chunks of 2 values of i divide
    // it exactly, instead of a zero-padded tail chunk that also falls off the interleaved staging
    if constexpr (cin_ic(MT) == 4 && cin_fast_shape<MT, CT, 2>()) {
        const bool w8 = m == MT && ((int64_t)Hp * MT) % 2 == 0 && (reinterpret_cast<uintptr_t>(W) & 7u) == 0;
        if (fast_env && w8 && Hp % 4 != 0 && Hp % 2 == 0) {
            constexpr int mp = (MT + 1) & ~1;
            const size_t sh2 = sizeof(float) * ((size_t)mp * CIN_ROWS + 2 * (size_t)2 * mp * cin_ws(CT) + 2 * (size_t)2 * CIN_ROWS);
            static LdsOnce once;
            (void)lds_limit(once, 160 * 1024, &cin_k<MT, CT, true, false, 2>);
            hipLaunchKernelGGL((cin_k<MT, CT, true, false, 2>), grid, dim3(256), sh2, st, x0, xk, W, m, Hp, H, D, dshift, hoff, R, xout, pooled, pooled_ld);
            return;
        }
    }
    if constexpr (cin_fast_shape<MT, CT>()) {
        if (fast_env && wvec && Hp >= cin_ic(MT)) {
            launch_cin_one<MT, CT, true>(grid, shmem, st, x0, xk, W, m, Hp, H, D, dshift, hoff, R, xout, pooled, pooled_ld);
            return;
        }
    }
    launch_cin_one<MT, CT, false>(grid, shmem, st, x0, xk, W, m, Hp, H, D, dshift, hoff, R, xout, pooled, pooled_ld);
}

template <int MT>
static int launch_cin(int ct, dim3 grid, hipStream_t st, const float* x0, const float* xk, const float* W, int m,
                      int Hp, int H, int D, int dshift, int hoff, int64_t R, float* xout, float* pooled, int64_t pooled_ld) {
    constexpr int mp = (MT + 1) & ~1;
    constexpr int IC = cin_ic(MT);
    const size_t shmem = sizeof(float) * ((size_t)mp * CIN_ROWS + 2 * (size_t)IC * mp * cin_ws(ct) + 2 * (size_t)IC * CIN_ROWS);
    switch (ct) {
        case 1: launch_cin_ct<MT, 1>(grid, shmem, st, x0, xk, W, m, Hp, H, D, dshift, hoff, R, xout, pooled, pooled_ld); break;
        case 2: launch_cin_ct<MT, 2>(grid, shmem, st, x0, xk, W, m, Hp, H, D, dshift, hoff, R, xout, pooled, pooled_ld); break;
        case 3: launch_cin_ct<MT, 3>(grid, shmem, st, x0, xk, W, m, Hp, H, D, dshift, hoff, R, xout, pooled, pooled_ld); break;
        default: launch_cin_ct<MT, 4>(grid, shmem, st, x0, xk, W, m, Hp, H, D, dshift, hoff, R, xout, pooled, pooled_ld); break;
    }
    return 0;
}

}  // namespace dir

using namespace dir;

extern "C" int dir_cin_layer_f32(const float* x0, const float* xk, const float* W, int m, int Hp, int H, int D,
                                 int64_t B, float* xout, float* pooled, int64_t pooled_ld, dir_stream_t stream) {
    DIR_CHECK_ARG(x0 && xk && W && (xout || pooled), "dir_cin_layer_f32: null pointer");
    DIR_CHECK_ARG(m > 0 && Hp > 0 && H > 0 && D > 0 && B >= 0, "dir_cin_layer_f32: m=%d Hp=%d H=%d D=%d", m, Hp, H, D);
    DIR_CHECK_ARG(!pooled || pooled_ld >= H, "dir_cin_layer_f32: pooled_ld=%lld < H=%d", (long long)pooled_ld, H);
    if (!(D == 4 || D == 8 || D == 16 || D == 32))
        return fail(DIR_E_UNSUPPORTED, "dir_cin_layer_f32: D=%d (supported: 4, 8, 16, 32)", D);
    if (xout && !aligned16(xout)) return fail(DIR_E_BADARG, "dir_cin_layer_f32: xout must be 16-byte aligned");
    if (B == 0) return DIR_OK;
    int dshift = 0;
    while ((1 << dshift) < D) ++dshift;
    const int64_t R = B * D;
    // Column blocks: a workgroup computes 1..4 tiles of 32 output columns.  H's tiles are split into 4-tile blocks plus ONE smaller
    // block for the remainder (a lone remainder tile borrows from a 4-block: 4 + 1 -> 3 + 2), each block size its own launch, so
    // H = 200 computes 224 columns (4 + 3 tiles), not 256.
    const int nt = (H + 31) / 32;
    int cnt[5] = {0, 0, 0, 0, 0};            // cnt[c] = number of c-tile blocks
    if (nt <= 4) cnt[nt] = 1;
    else {
        cnt[4] = nt / 4;
        const int r = nt % 4;
        if (r == 1) { --cnt[4]; cnt[3] = 1; cnt[2] = 1; }
        else if (r) cnt[r] = 1;
    }
    hipStream_t st = as_stream(stream);
    int hoff = 0;
    for (int ct = 4; ct >= 1; --ct) {
        if (!cnt[ct]) continue;
        dim3 grid((unsigned)((R + CIN_ROWS - 1) / CIN_ROWS), (unsigned)cnt[ct]);
        // the kernel is instantiated for padded field counts 8, 16, 26, 40; fields beyond m are zero operands
        if (m <= 8) launch_cin<8>(ct, grid, st, x0, xk, W, m, Hp, H, D, dshift, hoff, R, xout, pooled, pooled_ld);
        else if (m <= 16) launch_cin<16>(ct, grid, st, x0, xk, W, m, Hp, H, D, dshift, hoff, R, xout, pooled, pooled_ld);
        else if (m <= 26) launch_cin<26>(ct, grid, st, x0, xk, W, m, Hp, H, D, dshift, hoff, R, xout, pooled, pooled_ld);
        else if (m <= 40) launch_cin<40>(ct, grid, st, x0, xk, W, m, Hp, H, D, dshift, hoff, R, xout, pooled, pooled_ld);
        else return fail(DIR_E_UNSUPPORTED, "dir_cin_layer_f32: field count m=%d exceeds 40", m);
        hoff += 32 * ct * cnt[ct];
    }
    DIR_CHECK_LAUNCH("cin_layer");
    return DIR_OK;
}

// Diagnostic only: read (and clear) the cycle stamps accumulated by the DIR_CIN_STAMP=1 build.
extern "C" int dir_debug_cin_stamps(unsigned long long* out8) {
    if (!out8) return dir::fail(DIR_E_BADARG, "dir_debug_cin_stamps: null pointer");
    unsigned long long zero[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    if (hipDeviceSynchronize() != hipSuccess) return dir::fail(DIR_E_HIP, "dir_debug_cin_stamps: sync failed");
    if (hipMemcpyFromSymbol(out8, HIP_SYMBOL(dir::cin_stamp_acc), sizeof(zero)) != hipSuccess) return dir::fail(DIR_E_HIP, "dir_debug_cin_stamps: copy failed");
    if (hipMemcpyToSymbol(HIP_SYMBOL(dir::cin_stamp_acc), zero, sizeof(zero)) != hipSuccess) return dir::fail(DIR_E_HIP, "dir_debug_cin_stamps: clear failed");
    return DIR_OK;
}
