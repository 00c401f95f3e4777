// cin_bf3.hip -- the CIN layer of cin.hip on the bf16 matrix pipe with fp32-equivalent arithmetic ("bf16 x 3").
//
// NO REFERENCE CODE (README.md:28 links arXiv:1803.05170); definition as in cin.hip / include/dir_hip.h:
//   xout[b,h,d] = sum_{i<Hp} sum_{j<m} W[h, i*m+j] * xk[b,i,d] * x0[b,j,d]
//
// Arithmetic.  Every fp32 operand of the GEMM view (A[r,(i,j)] = fl(xk[b,i,d] * x0[b,j,d]) -- the same rounded product cin.hip feeds
// its fp32 MFMA -- and B = W) is split into three bf16 pieces by round-to-nearest: v = v0 + v1 + v2 exactly (3 x 8 significant bits
// plus the pieces' signs cover fp32's 24).  Of the nine piece products the six with weight >= 2^-16 are accumulated in fp32 by
// v_mfma_f32_32x32x16_bf16 (a bf16 x bf16 product is exact in fp32); the three dropped ones are <= 2^-24 relative each and of
// random sign.  The bf16 pipe runs 16 x the fp32 MFMA rate, so the six products cost 6/16 of cin.hip's MFMA time.
//
// Reduction order.  One MFMA step covers 16 reduction indices: lane half g = lane >> 5 supplies the 8 values i = 8*ib + e (e < 8)
// of field j = 2*t + g.  Steps run s = ib * MP2 + t (MP2 = ceil(m / 2)); four steps are a chunk (one barrier per 192 MFMAs).
//
// LDS (one workgroup of 4 waves = 256 rows (b,d) x 128 columns h, one wave per SIMD, accumulators in AGPRs):
//   x0s [mp][256] f32         the workgroup's x0 slice (whole kernel)
//   xks [2][2][256][4] f32    xk for one block of 8 values of i, [parity of ib][e >> 2][row][e & 3]: two conflict-free ds_read_b128
//   Wb  [2][4 steps][3 planes][4 cc][2 g][32 n][8 e] bf16   W chunk, in exactly the order cin_bf3_pack_w_k writes the global image,
//                             so a chunk arrives by 48 global_load_lds_dwordx4 per workgroup (no staging registers, no ds_write)
//                             and the 8 bf16 of one (plane, column tile) operand are ONE ds_read_b128.
#include "common.hpp"

namespace dir {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
typedef unsigned int u32x4_t __attribute__((ext_vector_type(4)));

constexpr int BF3_ROWS = 256;                 // rows per workgroup
constexpr int BF3_CH = 4;                     // steps per chunk
constexpr int BF3_STEP_BYTES = 3 * 4 * 2 * 32 * 16;   // 12 KB of W image per step
constexpr int BF3_CHUNK_BYTES = BF3_CH * BF3_STEP_BYTES;

__host__ __device__ inline int bf3_steps(int m, int Hp) {      // padded to whole chunks
    const int s = ((Hp + 7) / 8) * ((m + 1) / 2);
    return (s + BF3_CH - 1) / BF3_CH * BF3_CH;
}

__device__ __forceinline__ unsigned int bf3_pk_c(float a, float b) {
    const bf16x2_t v = {(__bf16)a, (__bf16)b};
    return __builtin_bit_cast(unsigned int, v);
}
__device__ __forceinline__ unsigned int bf3_pk(float a, float b) {      // v_cvt_pk_bf16_f32: round to nearest even; a in the low half
    unsigned int w;   // (asm: written as a cast the compiler converts `a` a second time, alone, to form float(bf16(a)) by a shift)
    asm("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(w) : "v"(a), "v"(b));
    return w;
}

// W [H, Hp*m] fp32 -> image [column block of 128][step][plane][cc][g][n][8 e] bf16 (zero where h >= H, i >= Hp, j >= m or the step is padding)
__global__ __launch_bounds__(256) void cin_bf3_pack_w_k(const float* __restrict__ W, int m, int Hp, int H, int nsteps, int ncb,
                                                       unsigned int* __restrict__ img) {
    const int MP2 = (m + 1) / 2;
    const int nblk = (Hp + 7) / 8;
    const int64_t total = (int64_t)ncb * nsteps * (4 * 2 * 32 * 4);   // one thread per (cb, step, cc, g, n, pair of e) -> 3 dwords
    for (int64_t e_ = (int64_t)blockIdx.x * 256 + threadIdx.x; e_ < total; e_ += (int64_t)gridDim.x * 256) {
        int64_t q = e_;
        const int ep = (int)(q & 3); q >>= 2;
        const int n = (int)(q & 31); q >>= 5;
        const int g = (int)(q & 1); q >>= 1;
        const int cc = (int)(q & 3); q >>= 2;
        const int s = (int)(q % nsteps);
        const int cb = (int)(q / nsteps);
        const int ib = s / MP2, t = s - ib * MP2;
        const int h = cb * 128 + 32 * cc + n;
        const int j = 2 * t + g;
        float v[2];
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int i = 8 * ib + 2 * ep + u;
            v[u] = (ib < nblk && h < H && i < Hp && j < m) ? W[(int64_t)h * Hp * m + (int64_t)i * m + j] : 0.f;
        }
        const unsigned int p0 = bf3_pk(v[0], v[1]);
        const float r0 = v[0] - __builtin_bit_cast(float, p0 << 16), r1 = v[1] - __builtin_bit_cast(float, p0 & 0xffff0000u);
        const unsigned int p1 = bf3_pk(r0, r1);
        const float s0 = r0 - __builtin_bit_cast(float, p1 << 16), s1 = r1 - __builtin_bit_cast(float, p1 & 0xffff0000u);
        const unsigned int p2 = bf3_pk(s0, s1);
        // dword index inside the step: ((plane*4 + cc)*2 + g)*32*4 + n*4 + ep
        const int64_t base = ((int64_t)cb * nsteps + s) * (BF3_STEP_BYTES / 4) + ((cc * 2 + g) * 32 + n) * 4 + ep;
        img[base] = p0;
        img[base + 1 * (4 * 2 * 32 * 4)] = p1;
        img[base + 2 * (4 * 2 * 32 * 4)] = p2;
    }
}

// 8 products of one row tile -> three bf16x8 operands whose sum is the products
__device__ __forceinline__ void bf3_split(const float (&x)[8], bf16x8_t& a0, bf16x8_t& a1, bf16x8_t& a2) {
    unsigned int w0[4], w1[4], w2[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const float a = x[2 * i], b = x[2 * i + 1];
        w0[i] = bf3_pk(a, b);
        const float ra = a - __builtin_bit_cast(float, w0[i] << 16), rb = b - __builtin_bit_cast(float, w0[i] & 0xffff0000u);
        w1[i] = bf3_pk(ra, rb);
        const float sa = ra - __builtin_bit_cast(float, w1[i] << 16), sb = rb - __builtin_bit_cast(float, w1[i] & 0xffff0000u);
        w2[i] = bf3_pk(sa, sb);
    }
    a0 = __builtin_bit_cast(bf16x8_t, (u32x4_t){w0[0], w0[1], w0[2], w0[3]});
    a1 = __builtin_bit_cast(bf16x8_t, (u32x4_t){w1[0], w1[1], w1[2], w1[3]});
    a2 = __builtin_bit_cast(bf16x8_t, (u32x4_t){w2[0], w2[1], w2[2], w2[3]});
}

// Issue order of one region of the step loop (12 MFMAs, 26 VALU instructions of the next step's A operands, the LDS reads of the
// next region's B operands; the first region of a step also reads the A inputs and starts with three bare MFMAs that cover the
// latency of those reads): MFMA : VALU = 1 : 2-3.
template <bool FIRST>
__device__ __forceinline__ void bf3_region_order() {
    if constexpr (FIRST) {
        __builtin_amdgcn_sched_group_barrier(0x100, 9, 0);
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
#pragma unroll
        for (int q = 0; q < 9; ++q) {
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x002, 3, 0);
        }
    } else {
        __builtin_amdgcn_sched_group_barrier(0x100, 3, 0);
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x002, 2, 0);
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x002, 2, 0);
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x002, 3, 0);
        }
    }
}

typedef __attribute__((address_space(3))) void* lds_ptr_t;
typedef const __attribute__((address_space(1))) void* glb_ptr_t;

template <int VAR>
__global__ __launch_bounds__(256, 1) void cin_bf3_k(const float* __restrict__ x0, const float* __restrict__ xk,
                                                    const unsigned char* __restrict__ img /* packed W image */, int m, int Hp, int H,
                                                    int D, int dshift, int nsteps, int64_t R, float* __restrict__ xout,
                                                    float* __restrict__ pooled, int64_t pooled_ld) {
    extern __shared__ __attribute__((aligned(16))) unsigned char bf3_smem[];
    const int mp = (m + 1) & ~1, MP2 = mp >> 1;
    const int nblk = (Hp + 7) >> 3;
    unsigned char* Wb = bf3_smem;                                                   // [2][BF3_CHUNK_BYTES]
    float* xks = reinterpret_cast<float*>(bf3_smem + 2 * BF3_CHUNK_BYTES);          // [2][2][256][4]
    float* x0s = xks + 2 * 2 * BF3_ROWS * 4;                                        // [mp][256]

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int n = lane & 31;
    const int g = lane >> 5;
    const int64_t row0 = (int64_t)blockIdx.x * BF3_ROWS;
    const int hbase = blockIdx.y * 128;
    const unsigned char* gimg = img + (int64_t)blockIdx.y * nsteps * BF3_STEP_BYTES;
    const int nchunk = nsteps / BF3_CH;

    // this thread's staging row (b,d) = row0 + tid, clamped (a row >= R only feeds output rows that are never stored)
    const int64_t srow = (row0 + tid < R) ? row0 + tid : R - 1;
    const float* x0src = x0 + ((srow >> dshift) * m) * D + (srow & (D - 1));
    const float* xksrc = xk + ((srow >> dshift) * Hp) * D + (srow & (D - 1));

    auto stage_piece = [&](int c, int buf, int q) {   // piece q of this wave's 12 x 1 KB pieces of chunk c, lane-linear
        const int piece = q * 4 + wave;
        const unsigned char* src = gimg + (int64_t)((VAR & 128) ? 0 : c) * BF3_CHUNK_BYTES + piece * 1024 + lane * 16;
        unsigned char* dst = Wb + buf * BF3_CHUNK_BYTES + piece * 1024;
        __builtin_amdgcn_global_load_lds((glb_ptr_t)src, (lds_ptr_t)dst, 16, 0, 0);
    };
    auto stage_w = [&](int c, int buf) {
#pragma unroll
        for (int q = 0; q < BF3_CHUNK_BYTES / 1024 / 4; ++q) stage_piece(c, buf, q);
    };
    float xreg[8];
    auto load_xk = [&](int ib) {             // block ib of 8 values of i (zeros past Hp: the W image is zero there, 0 * garbage must stay 0)
        const int ibc = ib < nblk ? ib : nblk - 1;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const int i = 8 * ibc + e;
            const float v = xksrc[(int64_t)(i < Hp ? i : Hp - 1) * D];
            xreg[e] = i < Hp ? v : 0.f;
        }
    };
    auto store_xk = [&](int ib) {
        const int ibc = ib < nblk ? ib : nblk - 1;
        float* dst = xks + (ibc & 1) * (2 * BF3_ROWS * 4) + tid * 4;
        *reinterpret_cast<float4*>(dst) = make_float4(xreg[0], xreg[1], xreg[2], xreg[3]);
        *reinterpret_cast<float4*>(dst + BF3_ROWS * 4) = make_float4(xreg[4], xreg[5], xreg[6], xreg[7]);
    };

    // ---- prologue: W chunk 0, x0 slice, xk block 0
    stage_w(0, 0);
    load_xk(0);
    for (int j = 0; j < mp; ++j) x0s[j * BF3_ROWS + tid] = j < m ? x0src[(int64_t)j * D] : 0.f;
    store_xk(0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();

    f32x16 acc[2][4];
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int c = 0; c < 4; ++c)
#pragma unroll
            for (int q = 0; q < 16; ++q) acc[t][c][q] = 0.f;

    const int rl = wave * 64 + n;                       // this lane's row in the workgroup (tile 0; tile 1 = +32)
    // Inputs of one step's A operands: xk[r, 8*ib .. 8*ib+7] and x0[r, 2*t + g] of both row tiles (six LDS reads) ...
    struct AIn { float4 lo[2], hi[2]; float xv[2]; };
    auto read_in = [&](int ib_, int t_, AIn& in) {
        const float* xb = xks + (ib_ & 1) * (2 * BF3_ROWS * 4) + rl * 4;
        const float* x0b = x0s + (2 * t_ + g) * BF3_ROWS + rl;
#pragma unroll
        for (int tl = 0; tl < 2; ++tl) {
            in.lo[tl] = *reinterpret_cast<const float4*>(xb + tl * 32 * 4);
            in.hi[tl] = *reinterpret_cast<const float4*>(xb + BF3_ROWS * 4 + tl * 32 * 4);
            in.xv[tl] = x0b[tl * 32];
        }
    };
    // ... and one "unit" u = 0..7 of the build: the two products e = 2*(u&3), 2*(u&3)+1 of row tile u >> 2, split into the three
    // bf16 pieces (13 VALU instructions; a step's 8 units are spread over its 48 MFMAs)
    auto build_unit = [&](const AIn& in, int u, unsigned int (&w)[2][3][4]) {
        const int tl = u >> 2, pr = u & 3;
        const float4 src = pr < 2 ? in.lo[tl] : in.hi[tl];
        const float a = ((pr & 1) ? src.z : src.x) * in.xv[tl], b = ((pr & 1) ? src.w : src.y) * in.xv[tl];
        const unsigned int w0 = (VAR & 2) ? bf3_pk(a, b) : bf3_pk_c(a, b);
        const float ra = a - __builtin_bit_cast(float, w0 << 16), rb = b - __builtin_bit_cast(float, w0 & 0xffff0000u);
        const unsigned int w1 = (VAR & 2) ? bf3_pk(ra, rb) : bf3_pk_c(ra, rb);
        const float sa = ra - __builtin_bit_cast(float, w1 << 16), sb = rb - __builtin_bit_cast(float, w1 & 0xffff0000u);
        w[tl][0][pr] = w0; w[tl][1][pr] = w1; w[tl][2][pr] = (VAR & 2) ? bf3_pk(sa, sb) : bf3_pk_c(sa, sb);
    };
    auto as_op = [](const unsigned int (&w)[4]) { return __builtin_bit_cast(bf16x8_t, (u32x4_t){w[0], w[1], w[2], w[3]}); };

    int ib = 0, t = 0;                                   // (ib, t) of the step whose A operands are in `aw`
    unsigned int aw[2][3][4];
    {
        AIn in;
        read_in(0, 0, in);
#pragma unroll
        for (int u = 0; u < 8; ++u) build_unit(in, u, aw);
    }
    int ib8 = 8 / MP2, t8 = 8 - ib8 * MP2;               // (block, field pair) of step 4c + 8
    bf16x8_t b[3], bn[3];                                // B operands (three planes) of the current / next region
#pragma unroll
    for (int p = 0; p < 3; ++p) b[p] = *reinterpret_cast<const bf16x8_t*>(Wb + (g * 32 + n) * 16 + (p * 4) * 1024);

    for (int c = 0; c < nchunk; ++c) {
        const int buf = c & 1;
        if (c + 1 < nchunk && !(VAR & 16) && !(VAR & 64)) stage_w(c + 1, buf ^ 1);
        const int cn = c + 1 < nchunk ? c + 1 : c;    // (VAR & 64: pieces issued region by region; the last chunk re-stages itself into the idle buffer)
        // xk staging rule (needs MP2 >= 8, checked by the host): chunk c stages the block of step 4c + 8.  Chunk c reads the blocks of
        // steps 4c .. 4c + 4 (its own steps and the inputs of step 4c + 4's A operands, built under its last step); those were staged
        // by chunk c-1 (block of step 4c + 4) or earlier and published by a barrier.  The staged block is block(4c) or block(4c) + 1:
        // re-staging a block that is being read writes identical values, and a new block goes to the buffer of the other parity,
        // whose previous content (block(4c) - 1) has no reader left.
        const int ibs = ib8;
        if (!(VAR & 32)) load_xk(ibs);
        t8 += BF3_CH;
        if (t8 >= MP2) { t8 -= MP2; ++ib8; }
        const unsigned char* wl = Wb + buf * BF3_CHUNK_BYTES + (g * 32 + n) * 16;
        if (!(VAR & 1)) {
#pragma unroll
            for (int p = 0; p < 3; ++p) b[p] = *reinterpret_cast<const bf16x8_t*>(wl + (p * 4) * 1024);
        }
#pragma unroll
        for (int st = 0; st < BF3_CH; ++st) {
            int tn = t + 1, ibn = ib;
            if (tn == MP2) { tn = 0; ibn = ib + 1; }
            const unsigned char* ws = wl + st * BF3_STEP_BYTES;
            unsigned int an[2][3][4];
            AIn in;
            bf16x8_t a[2][3];
#pragma unroll
            for (int tl = 0; tl < 2; ++tl)
#pragma unroll
                for (int p = 0; p < 3; ++p) a[tl][p] = as_op(aw[tl][p]);
            // Four regions per step, one per column tile cc: 12 MFMAs + two units of the next step's A operands + the LDS reads of
            // the next region's B operands (region 0 also reads the inputs of the units).  Inside a region the MFMAs and the VALU work
            // are interleaved 1 : 2-3 (one wave per SIMD: a VALU instruction only hides in the shadow of an MFMA of the same wave).
#pragma unroll
            for (int cc = 0; cc < 4; ++cc) {
                if ((VAR & 64) && st * 4 + cc < BF3_CHUNK_BYTES / 1024 / 4) stage_piece(cn, buf ^ 1, st * 4 + cc);
                if (cc == 0 && !(VAR & 4)) read_in(ibn < nblk ? ibn : nblk - 1, tn, in);
                const bool last = (cc == 3 && st == BF3_CH - 1);
                if (!last && !(VAR & 8)) {
                    const unsigned char* wn = cc < 3 ? ws + (cc + 1) * 1024 : ws + BF3_STEP_BYTES;
#pragma unroll
                    for (int p = 0; p < 3; ++p) bn[p] = *reinterpret_cast<const bf16x8_t*>(wn + (p * 4) * 1024);
                }
                if (last && (VAR & 1)) {
                    // The chunk's barrier sits IN FRONT of its last region: the 12 MFMAs below only need registers, and they cover
                    // the latency of the first LDS reads from the next chunk's buffer (one wave per SIMD: nothing else would).
                    store_xk(ibs);
                    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // this wave's W pieces of chunk c + 1 have landed in LDS
                    __syncthreads();
                    const unsigned char* wn = Wb + (buf ^ 1) * BF3_CHUNK_BYTES + (g * 32 + n) * 16;
#pragma unroll
                    for (int p = 0; p < 3; ++p) bn[p] = *reinterpret_cast<const bf16x8_t*>(wn + (p * 4) * 1024);
                    __builtin_amdgcn_sched_barrier(0);
                }
#pragma unroll
                for (int tl = 0; tl < 2; ++tl) {
                    f32x16 v = acc[tl][cc];
                    v = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[tl][0], b[2], v, 0, 0, 0);
                    v = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[tl][2], b[0], v, 0, 0, 0);
                    v = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[tl][1], b[1], v, 0, 0, 0);
                    v = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[tl][0], b[1], v, 0, 0, 0);
                    v = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[tl][1], b[0], v, 0, 0, 0);
                    v = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[tl][0], b[0], v, 0, 0, 0);
                    acc[tl][cc] = v;
                }
                if (!(VAR & 4)) {
                    build_unit(in, 2 * cc, an);
                    build_unit(in, 2 * cc + 1, an);
                }
                // the region's issue order: LDS reads first, then MFMA : VALU = 1 : 2 (region 0 starts with three bare MFMAs that
                // cover the latency of the input reads)
                if (cc == 0) bf3_region_order<true>(); else bf3_region_order<false>();
                __builtin_amdgcn_sched_barrier(0);
                if (!(VAR & 8) && !(last && !(VAR & 1))) {
#pragma unroll
                    for (int p = 0; p < 3; ++p) b[p] = bn[p];
                }
            }
#pragma unroll
            for (int tl = 0; tl < 2; ++tl)
#pragma unroll
                for (int p = 0; p < 3; ++p)
#pragma unroll
                    for (int q = 0; q < 4; ++q) aw[tl][p][q] = (VAR & 4) ? aw[tl][p][q] : an[tl][p][q];
            t = tn; ib = ibn;
        }
        if (!(VAR & 1) && !(VAR & 32)) {
            store_xk(ibs);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
        }
    }

    // ---- epilogue: C/D map col = lane&31, row = (reg&3) + 8*(reg>>2) + 4*(lane>>5)  (as cin.hip)
    if (xout) {
#pragma unroll
        for (int tl = 0; tl < 2; ++tl) {
            const int64_t trow = row0 + wave * 64 + tl * 32;
#pragma unroll
            for (int cc = 0; cc < 4; ++cc) {
                const int h = hbase + 32 * cc + n;
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int64_t gr = trow + 8 * q + 4 * g;
                    if (h < H && gr < R) {
                        const int64_t b = gr >> dshift;
                        const int d = (int)(gr & (D - 1));
                        float4 v = make_float4(acc[tl][cc][4 * q], acc[tl][cc][4 * q + 1], acc[tl][cc][4 * q + 2], acc[tl][cc][4 * q + 3]);
                        *reinterpret_cast<float4*>(xout + (b * H + h) * D + d) = v;
                    }
                }
            }
        }
    }
    if (pooled) {
#pragma unroll
        for (int tl = 0; tl < 2; ++tl) {
            const int64_t trow = row0 + wave * 64 + tl * 32;
#pragma unroll
            for (int cc = 0; cc < 4; ++cc) {
                const int h = hbase + 32 * cc + n;
                const float p0 = (acc[tl][cc][0] + acc[tl][cc][1]) + (acc[tl][cc][2] + acc[tl][cc][3]);
                const float p1 = (acc[tl][cc][4] + acc[tl][cc][5]) + (acc[tl][cc][6] + acc[tl][cc][7]);
                const float p2 = (acc[tl][cc][8] + acc[tl][cc][9]) + (acc[tl][cc][10] + acc[tl][cc][11]);
                const float p3 = (acc[tl][cc][12] + acc[tl][cc][13]) + (acc[tl][cc][14] + acc[tl][cc][15]);
                if (D == 4) {
                    const float pv[4] = {p0, p1, p2, p3};
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const int64_t gr = trow + 8 * q + 4 * g;
                        if (h < H && gr < R) pooled[(gr >> dshift) * pooled_ld + h] = pv[q];
                    }
                } else {
                    float s0, s1, s2, s3;
                    if (D == 8) { s0 = p0; s1 = p1; s2 = p2; s3 = p3; }
                    else if (D == 16) { s0 = p0 + p1; s1 = p2 + p3; s2 = 0.f; s3 = 0.f; }
                    else { s0 = (p0 + p1) + (p2 + p3); s1 = 0.f; s2 = 0.f; s3 = 0.f; }
                    s0 += __shfl_xor(s0, 32, 64);
                    s1 += __shfl_xor(s1, 32, 64);
                    s2 += __shfl_xor(s2, 32, 64);
                    s3 += __shfl_xor(s3, 32, 64);
                    const int ns = 32 >> dshift;
                    const float sv[4] = {s0, s1, s2, s3};
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const int64_t gr = trow + (int64_t)q * D;
                        if (q < ns && g == 0 && h < H && gr < R) pooled[(gr >> dshift) * pooled_ld + h] = sv[q];
                    }
                }
            }
        }
    }
}

}  // namespace dir

using namespace dir;

extern "C" int64_t dir_cin_bf16x3_workspace_bytes(int m, int Hp, int H) {
    if (m <= 0 || Hp <= 0 || H <= 0) return 0;
    const int64_t a = (int64_t)((H + 127) / 128) * bf3_steps(m, Hp) * BF3_STEP_BYTES, b = cin_bf3t_workspace_bytes(m, Hp, H);
    return a > b ? a : b;
}

extern "C" int dir_cin_layer_bf16x3_f32(const float* x0, const float* xk, const float* W, int m, int Hp, int H, int D, int64_t B,
                                        float* xout, float* pooled, int64_t pooled_ld, void* workspace, int64_t workspace_bytes,
                                        dir_stream_t stream) {
    const char* name = "dir_cin_layer_bf16x3_f32";
    DIR_CHECK_ARG(x0 && xk && W && (xout || pooled) && workspace, "%s: null pointer", name);
    DIR_CHECK_ARG(m > 0 && Hp > 0 && H > 0 && D > 0 && B >= 0, "%s: m=%d Hp=%d H=%d D=%d", name, m, Hp, H, D);
    DIR_CHECK_ARG(!pooled || pooled_ld >= H, "%s: pooled_ld=%lld < H=%d", name, (long long)pooled_ld, H);
    if (!(D == 4 || D == 8 || D == 16 || D == 32)) return fail(DIR_E_UNSUPPORTED, "%s: D=%d (supported: 4, 8, 16, 32)", name, D);
    const int kern = getenv("DIR_BF3_KERNEL") ? atoi(getenv("DIR_BF3_KERNEL")) : 1;   // development switch: 1 = cin_bf3_k, 2 = cin_bf3t_k
    if (m > 40 || (kern != 2 && m < 15)) return fail(DIR_E_UNSUPPORTED, "%s: field count m=%d (supported: 15..40; use dir_cin_layer_f32)", name, m);
    if (xout && !aligned16(xout)) return fail(DIR_E_BADARG, "%s: xout must be 16-byte aligned", name);
    DIR_CHECK_ARG(aligned16(workspace) && workspace_bytes >= dir_cin_bf16x3_workspace_bytes(m, Hp, H),
                  "%s: workspace must be 16-byte aligned and hold dir_cin_bf16x3_workspace_bytes(m, Hp, H) bytes", name);
    if (B == 0) return DIR_OK;
    int dshift = 0;
    while ((1 << dshift) < D) ++dshift;
    const int64_t R = B * D;
    const int nsteps = bf3_steps(m, Hp);
    const int ncb = (H + 127) / 128;
    hipStream_t st = as_stream(stream);
    if (kern == 2) {
        launch_cin_bf3t(st, x0, xk, W, m, Hp, H, D, dshift, R, xout, pooled, pooled_ld, workspace);
        DIR_CHECK_LAUNCH("cin_layer_bf16x3");
        return DIR_OK;
    }
    const int64_t pack_threads = (int64_t)ncb * nsteps * 1024;
    hipLaunchKernelGGL(cin_bf3_pack_w_k, dim3((unsigned)((pack_threads + 255) / 256)), dim3(256), 0, st, W, m, Hp, H, nsteps, ncb,
                       static_cast<unsigned int*>(workspace));
    const int mp = (m + 1) & ~1;
    const size_t shmem = 2 * (size_t)BF3_CHUNK_BYTES + sizeof(float) * (2 * 2 * BF3_ROWS * 4 + (size_t)mp * BF3_ROWS);
    dim3 grid((unsigned)((R + BF3_ROWS - 1) / BF3_ROWS), (unsigned)ncb);
    const int var = getenv("DIR_BF3_VAR") ? atoi(getenv("DIR_BF3_VAR")) : 2;   // development switch (tools/cin_bf3_var.py)
    const unsigned char* img = static_cast<const unsigned char*>(workspace);
#define BF3_LAUNCH(V)                                                                                                              \
    do {                                                                                                                           \
        static bool set = false;                                                                                                   \
        if (!set) {                                                                                                                \
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&cin_bf3_k<V>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); \
            set = true;                                                                                                            \
        }                                                                                                                          \
        hipLaunchKernelGGL(cin_bf3_k<V>, grid, dim3(256), shmem, st, x0, xk, img, m, Hp, H, D, dshift, nsteps, R, xout, pooled, pooled_ld); \
    } while (0)
    switch (var) {
        case 0: BF3_LAUNCH(0); break;
        case 1: BF3_LAUNCH(1); break;
        case 2: BF3_LAUNCH(2); break;
        case 66: BF3_LAUNCH(66); break;    // W pieces issued region by region
        case 130: BF3_LAUNCH(130); break;  // ablation: W staged from chunk 0 every time (L1/L2-hot source)
        case 6: BF3_LAUNCH(6); break;      // timing ablations (results wrong by construction): no A build
        case 10: BF3_LAUNCH(10); break;    // no B reads
        case 18: BF3_LAUNCH(18); break;    // no W staging after chunk 0
        case 34: BF3_LAUNCH(34); break;    // no barrier / xk staging
        case 62: BF3_LAUNCH(62); break;    // MFMAs only
        case 3: BF3_LAUNCH(3); break;
        default: BF3_LAUNCH(2); break;
    }
#undef BF3_LAUNCH
    DIR_CHECK_LAUNCH("cin_layer_bf16x3");
    return DIR_OK;
}
