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
// = 8 accumulators (128 VGPRs).  A 256-thread workgroup (one wave per SIMD) owns 256 rows.
//
// LDS (one array, 16-byte aligned carve):
//   x0s [mp][256]            the workgroup's x0 slice, transposed so that lanes read consecutive rows
//   Ws  [2][IC*mp][HP]       W chunk of IC values of i, transposed to [kk][h] (HP = 129: conflict-free
//                            for both the transposing store and the per-column read)
//   xks [2][IC][256]         xk chunk
// Chunk c+1 is fetched from global into registers before chunk c's MFMAs and written to the other LDS
// buffer after them: one barrier per chunk of IC*mp/2*RT*CT = 416 MFMAs.
#include "common.hpp"

namespace dir {

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int CIN_RT = 2;          // row tiles per wave
constexpr int CIN_ROWS = 256;      // rows per workgroup = 4 waves * RT * 32
constexpr int CIN_HP = 129;        // padded H stride of the LDS W image
constexpr int CIN_IC = 4;          // i values per chunk

template <int MT /* m, compile-time */, int CT /* column tiles */>
__global__ __launch_bounds__(256, 1) void cin_k(const float* __restrict__ x0, const float* __restrict__ xk,
                                                const float* __restrict__ W, int Hp, int H, int D, int dshift,
                                                int64_t R /* B*D */, float* __restrict__ xout,
                                                float* __restrict__ pooled, int64_t pooled_ld) {
    constexpr int m = MT;
    constexpr int mp = (MT + 1) & ~1;
    constexpr int MP2 = mp / 2;
    constexpr int IC = CIN_IC;
    constexpr int HP = CIN_HP;
    constexpr int WCH = IC * mp * HP;     // floats per W buffer
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
    const int64_t row0 = (int64_t)blockIdx.x * CIN_ROWS;   // first (b,d) row of this workgroup
    const int hbase = blockIdx.y * (32 * CT);              // first output column of this workgroup
    const int Kd = Hp * m;

    // ---- stage x0 slice: x0s[j][r] = x0[b, j, d] with (b,d) = row0 + r ------------------------------
    for (int e = tid; e < mp * CIN_ROWS; e += 256) {
        const int j = e / CIN_ROWS, r = e - j * CIN_ROWS;
        const int64_t gr = row0 + r;
        float v = 0.f;
        if (j < m && gr < R) {
            const int64_t b = gr >> dshift;
            const int d = (int)(gr & (D - 1));
            v = x0[(b * m + j) * D + d];
        }
        x0s[e] = v;
    }

    // per-thread staging registers for the next chunk
    constexpr int WE = (IC * MT * 32 * CT + 255) / 256;   // W elements per thread per chunk
    constexpr int XE = XCH / 256;                         // xk elements per thread per chunk (= IC)
    float wreg[WE];
    float xreg[XE];

    auto fetch_chunk = [&](int c) {
        const int i0 = c * IC;
#pragma unroll
        for (int q = 0; q < WE; ++q) {
            const int e = tid + 256 * q;           // e = hl * (IC*m) + kkl
            const int hl = e / (IC * MT);
            const int kkl = e - hl * (IC * MT);
            const int il = kkl / MT;
            const int h = hbase + hl;
            float v = 0.f;
            if (hl < 32 * CT && h < H && i0 + il < Hp) v = W[(int64_t)h * Kd + (int64_t)i0 * MT + kkl];
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
    auto store_chunk = [&](int buf) {
        float* wb = Ws + buf * WCH;
#pragma unroll
        for (int q = 0; q < WE; ++q) {
            const int e = tid + 256 * q;
            const int hl = e / (IC * MT);
            const int kkl = e - hl * (IC * MT);
            const int il = kkl / MT;
            const int j = kkl - il * MT;
            if (hl < 32 * CT) wb[(il * mp + j) * HP + hl] = wreg[q];
        }
        float* xb = xks + buf * XCH;
#pragma unroll
        for (int q = 0; q < XE; ++q) xb[tid + 256 * q] = xreg[q];
    };

    // zero the j = m pad rows of both W buffers once (odd m only)
    if (mp != m) {
        for (int e = tid; e < 2 * IC * HP; e += 256) {
            const int buf = e / (IC * HP);
            const int rem = e - buf * (IC * HP);
            const int il = rem / HP, hl = rem - il * HP;
            Ws[buf * WCH + (il * mp + m) * HP + hl] = 0.f;
        }
    }

    const int nchunk = (Hp + IC - 1) / IC;
    fetch_chunk(0);
    store_chunk(0);
    __syncthreads();

    // this lane's x0 operands: x0r[t][jp] = x0s[2*jp + hh][wave*64 + t*32 + n]
    float x0r[CIN_RT][MP2];
#pragma unroll
    for (int t = 0; t < CIN_RT; ++t)
#pragma unroll
        for (int jp = 0; jp < MP2; ++jp) x0r[t][jp] = x0s[(2 * jp + hh) * CIN_ROWS + wave * 64 + t * 32 + n];

    f32x16 acc[CIN_RT][CT];
#pragma unroll
    for (int t = 0; t < CIN_RT; ++t)
#pragma unroll
        for (int c = 0; c < CT; ++c)
#pragma unroll
            for (int q = 0; q < 16; ++q) acc[t][c][q] = 0.f;

    for (int c = 0; c < nchunk; ++c) {
        const int buf = c & 1;
        if (c + 1 < nchunk) fetch_chunk(c + 1);
        const float* wb = Ws + buf * WCH + hh * HP + n;
        const float* xb = xks + buf * XCH + wave * 64 + n;
#pragma unroll 1
        for (int il = 0; il < IC; ++il) {
            float xkv[CIN_RT];
#pragma unroll
            for (int t = 0; t < CIN_RT; ++t) xkv[t] = xb[il * CIN_ROWS + t * 32];
            const float* wr = wb + il * mp * HP;
#pragma unroll
            for (int jp = 0; jp < MP2; ++jp) {
                float bw[CT];
#pragma unroll
                for (int cc = 0; cc < CT; ++cc) bw[cc] = wr[(2 * jp) * HP + 32 * cc];
                float a[CIN_RT];
#pragma unroll
                for (int t = 0; t < CIN_RT; ++t) a[t] = xkv[t] * x0r[t][jp];
#pragma unroll
                for (int t = 0; t < CIN_RT; ++t)
#pragma unroll
                    for (int cc = 0; cc < CT; ++cc)
                        acc[t][cc] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[t], bw[cc], acc[t][cc], 0, 0, 0);
            }
        }
        if (c + 1 < nchunk) store_chunk(buf ^ 1);
        __syncthreads();
    }

    // ---- epilogue: C/D map col = lane&31, row = (reg&3) + 8*(reg>>2) + 4*(lane>>5) -------------------
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
}

template <int MT>
static int launch_cin(int ct, dim3 grid, size_t shmem, hipStream_t st, const float* x0, const float* xk,
                      const float* W, int Hp, int H, int D, int dshift, int64_t R, float* xout, float* pooled,
                      int64_t pooled_ld) {
#define DIR_GO(CT)                                                                                         \
    do {                                                                                                   \
        static bool set = false;                                                                           \
        if (!set) {                                                                                        \
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&cin_k<MT, CT>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); \
            set = true;                                                                                    \
        }                                                                                                  \
        hipLaunchKernelGGL((cin_k<MT, CT>), grid, dim3(256), shmem, st, x0, xk, W, Hp, H, D, dshift, R, xout, pooled, pooled_ld); \
    } while (0)
    switch (ct) {
        case 1: DIR_GO(1); break;
        case 2: DIR_GO(2); break;
        default: DIR_GO(4); break;
    }
#undef DIR_GO
    return 0;
}

}  // namespace dir

using namespace dir;

extern "C" int dir_cin_layer_f32(const float* x0, const float* xk, const float* W, int m, int Hp, int H, int D,
                                 int64_t B, float* xout, float* pooled, int64_t pooled_ld, dir_stream_t stream) {
    DIR_CHECK_ARG(x0 && xk && W && xout, "dir_cin_layer_f32: null pointer");
    DIR_CHECK_ARG(m > 0 && Hp > 0 && H > 0 && D > 0 && B >= 0, "dir_cin_layer_f32: m=%d Hp=%d H=%d D=%d", m, Hp, H, D);
    DIR_CHECK_ARG(!pooled || pooled_ld >= H, "dir_cin_layer_f32: pooled_ld=%lld < H=%d", (long long)pooled_ld, H);
    if (!(D == 4 || D == 8 || D == 16 || D == 32))
        return fail(DIR_E_UNSUPPORTED, "dir_cin_layer_f32: D=%d (supported: 4, 8, 16, 32)", D);
    if (!aligned16(xout)) return fail(DIR_E_BADARG, "dir_cin_layer_f32: xout must be 16-byte aligned");
    if (B == 0) return DIR_OK;
    int dshift = 0;
    while ((1 << dshift) < D) ++dshift;
    const int64_t R = B * D;
    int ct = (H + 31) / 32;
    ct = ct >= 3 ? 4 : ct;                   // column tiles per workgroup: 1, 2 or 4
    const int colblocks = (H + 32 * ct - 1) / (32 * ct);
    const int mp = (m + 1) & ~1;
    const size_t shmem = sizeof(float) * ((size_t)mp * CIN_ROWS + 2 * (size_t)CIN_IC * mp * CIN_HP + 2 * (size_t)CIN_IC * CIN_ROWS);
    if (shmem > 160 * 1024) return fail(DIR_E_UNSUPPORTED, "dir_cin_layer_f32: m=%d needs %zu B of LDS", m, shmem);
    dim3 grid((unsigned)((R + CIN_ROWS - 1) / CIN_ROWS), (unsigned)colblocks);
    hipStream_t st = as_stream(stream);
    switch (m) {
        case 26: launch_cin<26>(ct, grid, shmem, st, x0, xk, W, Hp, H, D, dshift, R, xout, pooled, pooled_ld); break;
        case 8: launch_cin<8>(ct, grid, shmem, st, x0, xk, W, Hp, H, D, dshift, R, xout, pooled, pooled_ld); break;
        case 5: launch_cin<5>(ct, grid, shmem, st, x0, xk, W, Hp, H, D, dshift, R, xout, pooled, pooled_ld); break;
        default:
            return fail(DIR_E_UNSUPPORTED, "dir_cin_layer_f32: field count m=%d is not among the built instantiations (5, 8, 26)", m);
    }
    DIR_CHECK_LAUNCH("cin_layer");
    return DIR_OK;
}
