// cin_pooled.hip -- the LAST layer of a CIN stack in inference, fused: pooled[b,h] = sum_{i,j} W[h,i,j] Z[b,i,j], Z[b,i,j] = sum_d xk[b,i,d] x0[b,j,d]
// without Z ever reaching memory (round 6).
//
// NO REFERENCE CODE (README.md:28 links arXiv:1803.05170); the definition is include/dir_hip.h (A14) and oracle/dir_oracle.c.
//
// Why.  csrc/cin_pool.hip takes the sum over the embedding dimension first (1/D of the layer's matrix work), but as TWO passes: cin_pool_z_k
// writes Z [B, Hp m] (872 MB at the BASELINE shape: 0.39 ms of streaming) and the dense kernel reads it back and splits it (0.22-0.27 ms).
// Training needs Z (the weight gradient is g^T Z); inference does not.  Here a lane forms the Z values it is about to feed the matrix pipe:
//   D[h][sample] += A[h][k] . B[k][sample] with v_mfma_f32_16x16x32_bf16, k = (channel i, field j): ONE k-step per channel, its 32 k slots the
//   fields (26 at the BASELINE shape, padded with zero weights).  Lane (kk, r) supplies B[k = 8 kk + j'][n = r] = Z[sample r][i][field 8 kk + j']:
//   the x0 rows of ITS eight fields of ITS sample stay in 128 registers for the whole sample tile, a channel's 16 xk values arrive with four
//   16-byte LDS reads, 128 fmas give the eight sums, three-way bf16 splits give the operand (Z is a general input: bf16 x 3 has fp32's exponent range and needs no row maxima -- the
//   row-scaled fp16 x 2 form of the two-pass path gets its scale from a max over the row this kernel never holds).
//   A = the weights of channel i: [H / 16 tiles][3 pieces] of 1 KB in MFMA operand order, streamed through LDS by LDS-DMA in a ring of three
//   stages; the xk rows come the same way (a wave's 16 samples x 64 bytes per channel: ONE LDS-DMA instruction), so that every load of the
//   channel loop is counted by ONE in-order counter and the wait in front of a channel's barrier can leave channel i + 2's loads in flight;
//   one barrier per channel; a workgroup = 8 waves x 16 samples.
// Bytes: x0 + xk + pooled (0.68 GB at the BASELINE shape) from HBM, the 3 MB image once per 128 samples from L2.
// THIS FILE IS COMPILED WITHOUT PACKED fp32 VALU INSTRUCTIONS (build.py; the gfx950 hazard beside 16x16x32 MFMAs: din_wave.hip).
#include "common.hpp"

namespace dir {
namespace {

typedef float cq_f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 cq_bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned int cq_u32x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) void* cq_lds_ptr;
typedef const __attribute__((address_space(1))) void* cq_glb_ptr;

constexpr int CQ_D = 16;          // embedding dimension
constexpr int CQ_MP = 32;         // fields per k-step (m <= 32; the slots behind m carry zero weights)
constexpr int CQ_NW = 8;          // waves per workgroup
constexpr int CQ_ROWS = 16 * CQ_NW;

__device__ __forceinline__ unsigned int cq_pk(float a, float b) {      // v_cvt_pk_bf16_f32 (round to nearest even), a in the low half
    typedef __bf16 pk2_t __attribute__((ext_vector_type(2)));
    const pk2_t v = {(__bf16)a, (__bf16)b};
    unsigned int w = __builtin_bit_cast(unsigned int, v);
    asm("" : "+v"(w));      // keeps the compiler from folding the split away (dense_bf3.hip: db3_pk)
    return w;
}
__device__ __forceinline__ void cq_split_pair(float a, float b, unsigned int& w0, unsigned int& w1, unsigned int& w2) {
    w0 = cq_pk(a, b);
    const float ra = a - __builtin_bit_cast(float, w0 << 16), rb = b - __builtin_bit_cast(float, w0 & 0xffff0000u);
    w1 = cq_pk(ra, rb);
    const float sa = ra - __builtin_bit_cast(float, w1 << 16), sb = rb - __builtin_bit_cast(float, w1 & 0xffff0000u);
    w2 = cq_pk(sa, sb);
}

// W [H, Hp * m] (column i m + j) -> image [channel i][tile ht][piece][lane l][4 dwords]: element e of lane l = piece of
// W[h = 16 ht + (l & 15)][i][field 8 (l >> 4) + e]; zero where h >= H or the field is >= m.
__global__ __launch_bounds__(256) void cin_pooled_pack_k(const float* __restrict__ W, int m, int Hp, int H, int HT, unsigned int* __restrict__ img) {
    const int64_t total = (int64_t)Hp * HT * 64 * 4;              // one thread per pair of e
    for (int64_t e_ = (int64_t)blockIdx.x * 256 + threadIdx.x; e_ < total; e_ += (int64_t)gridDim.x * 256) {
        int64_t q = e_;
        const int ep = (int)(q & 3); q >>= 2;
        const int l = (int)(q & 63); q >>= 6;
        const int ht = (int)(q % HT);
        const int i = (int)(q / HT);
        const int h = 16 * ht + (l & 15);
        const int j = 8 * (l >> 4) + 2 * ep;
        const float v0 = (h < H && j < m) ? W[(int64_t)h * Hp * m + (int64_t)i * m + j] : 0.f;
        const float v1 = (h < H && j + 1 < m) ? W[(int64_t)h * Hp * m + (int64_t)i * m + j + 1] : 0.f;
        unsigned int p0, p1, p2;
        cq_split_pair(v0, v1, p0, p1, p2);
        const int64_t base = (((int64_t)i * HT + ht) * 3) * 256 + l * 4 + ep;
        img[base] = p0;
        img[base + 256] = p1;
        img[base + 512] = p2;
    }
}

constexpr int CQ_RING = 3;        // stages of weights / of xk rows in LDS: what channel i + 2 needs is in flight while channel i is computed
template <int HT>
__global__ __launch_bounds__(64 * CQ_NW, 2) void cin_pooled_k(const float* __restrict__ x0, const float* __restrict__ xk,
                                                              const unsigned char* __restrict__ img, int m, int Hp, int H, int64_t B,
                                                              float* __restrict__ pooled, int64_t pooled_ld,
                                                              const int64_t* __restrict__ x0_inv /* nullable: x0 is then a row list [n, 16] and
                                                                 x0_inv [B, m] the position of (sample, field)'s row in it, < 0 = a zero row */) {
    constexpr int STAGE = HT * 3 * 1024;                          // bytes of one channel's weights
    constexpr int NS = (HT * 3 + CQ_NW - 1) / CQ_NW;              // weight pieces a wave brings per stage (a slot behind the last is the last again)
    constexpr int XSLOT = 1024;                                   // a wave's 16 samples x 64 bytes of one channel
    extern __shared__ __attribute__((aligned(16))) unsigned char cq_smem[];      // [RING][STAGE] weights, then [NW][RING][XSLOT] xk rows
    unsigned char* const xring = cq_smem + CQ_RING * STAGE;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, r16 = lane & 15, kk = lane >> 4;
    const int wave_u = __builtin_amdgcn_readfirstlane(wave);
    const int64_t ntiles = (B + CQ_ROWS - 1) / CQ_ROWS;
    if ((int64_t)blockIdx.x >= ntiles) return;
    // Every load of the channel loop is an LDS-DMA, the SAME number per wave and channel (NS weight pieces + one piece of xk rows): the wait
    // in front of a channel's barrier is then s_waitcnt vmcnt(NS + 1) -- what channel i + 2 needs may stay in flight, everything older has
    // landed (vmcnt counts in order: a register load of xk in between would have to be waited for with it, HBM latency and all).
    auto fetch = [&](const int i, const int slot, const float* xkb) {   // channel i -> ring slot `slot`
        const unsigned char* src = img + (int64_t)i * STAGE + lane * 16;
        unsigned char* dst = cq_smem + slot * STAGE + lane * 16;
#pragma unroll
        for (int k = 0; k < NS; ++k) {
            const int s_ = wave_u + CQ_NW * k < HT * 3 ? wave_u + CQ_NW * k : HT * 3 - 1;
            __builtin_amdgcn_global_load_lds((cq_glb_ptr)(src + s_ * 1024), (cq_lds_ptr)(dst + s_ * 1024), 16, 0, 0);
        }
        // lane (kk, r) brings floats 4 kk .. 4 kk + 3 of its sample's channel row: the slot holds [piece kk][sample r][16 bytes]
        __builtin_amdgcn_global_load_lds((cq_glb_ptr)(xkb + (int64_t)i * CQ_D + 4 * kk),
                                         (cq_lds_ptr)(xring + (wave_u * CQ_RING + slot) * XSLOT + lane * 16), 16, 0, 0);
    };
    for (int64_t t = blockIdx.x; t < ntiles; t += gridDim.x) {
        const int64_t b = t * CQ_ROWS + wave * 16 + r16;
        const int64_t bb = b < B ? b : B - 1;
        const float* xkb = xk + bb * Hp * CQ_D;
        __syncthreads();                                          // the previous tile's last channels have been read
        fetch(0, 0, xkb);
        fetch(Hp > 1 ? 1 : 0, 1, xkb);
        // the x0 rows of this lane's eight fields of its sample (a field >= m: zeros, and its weights are zeros too)
        float4 xr[8][4];
        int64_t xe[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) xe[j] = bb * m + (8 * kk + j < m ? 8 * kk + j : 0);
        if (x0_inv) {                                              // the sharded lookup's received rows, read through the inverse positions
#pragma unroll
            for (int j = 0; j < 8; ++j) xe[j] = x0_inv[xe[j]];     // (all eight positions in flight before the first row read)
        }
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const bool live = 8 * kk + j < m && xe[j] >= 0;
            const float* p = x0 + (xe[j] >= 0 ? xe[j] : 0) * CQ_D;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const float4 v = *reinterpret_cast<const float4*>(p + 4 * q);
                xr[j][q] = live ? v : make_float4(0.f, 0.f, 0.f, 0.f);
            }
        }
        cq_f32x4 acc[HT];
#pragma unroll
        for (int ht = 0; ht < HT; ++ht) acc[ht] = (cq_f32x4){0.f, 0.f, 0.f, 0.f};
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        int slot = 0;
        for (int i = 0; i < Hp; ++i) {
            {   // channel i + 2 (the last channel again beyond the end: the same number of loads every time)
                const int nslot = slot + 2 < CQ_RING ? slot + 2 : slot + 2 - CQ_RING;
                fetch(i + 2 < Hp ? i + 2 : Hp - 1, nslot, xkb);
            }
            // this channel's 16 xk values of the lane's sample, from the wave's ring
            // (LDS reads by hand: behind a compiler-visible read of LDS the compiler waits for EVERY outstanding LDS-DMA -- vmcnt(0) right behind
            // the loads this channel has just issued, their whole latency exposed in every channel: 636 -> 2xx us.  What orders these reads
            // behind the DMA that filled the slot is the vmcnt(NS + 1) + barrier at the end of the previous channel.)
            const unsigned int xl = (unsigned int)(size_t)(xring + (wave_u * CQ_RING + slot) * XSLOT + r16 * 16);
            cq_f32x4 cu[4];
            asm volatile("ds_read_b128 %0, %4\n\tds_read_b128 %1, %4 offset:256\n\tds_read_b128 %2, %4 offset:512\n\tds_read_b128 %3, %4 offset:768\n\t"
                         "s_waitcnt lgkmcnt(0)"
                         : "=&v"(cu[0]), "=&v"(cu[1]), "=&v"(cu[2]), "=&v"(cu[3]) : "v"(xl));
            float4 cur[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) cur[q] = make_float4(cu[q][0], cu[q][1], cu[q][2], cu[q][3]);
            // Z[sample][i][field 8 kk + j] = sum_d xk[i][d] x0[field][d]: one fma chain per field, d ascending (eight independent chains: the
            // kernel is bound by its vector instructions -- 16 fmas per Z value are the floor, cin_pool_z_k's four partial sums cost three adds more)
            float z[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                float a_ = cur[0].x * xr[j][0].x;
                a_ = fmaf(cur[0].y, xr[j][0].y, a_); a_ = fmaf(cur[0].z, xr[j][0].z, a_); a_ = fmaf(cur[0].w, xr[j][0].w, a_);
#pragma unroll
                for (int q = 1; q < 4; ++q) {
                    a_ = fmaf(cur[q].x, xr[j][q].x, a_); a_ = fmaf(cur[q].y, xr[j][q].y, a_);
                    a_ = fmaf(cur[q].z, xr[j][q].z, a_); a_ = fmaf(cur[q].w, xr[j][q].w, a_);
                }
                z[j] = a_;
            }
            unsigned int w[3][4];
#pragma unroll
            for (int pr = 0; pr < 4; ++pr) cq_split_pair(z[2 * pr], z[2 * pr + 1], w[0][pr], w[1][pr], w[2][pr]);
            cq_bf16x8 x[3];
#pragma unroll
            for (int pc = 0; pc < 3; ++pc) x[pc] = __builtin_bit_cast(cq_bf16x8, (cq_u32x4){w[pc][0], w[pc][1], w[pc][2], w[pc][3]});
            // The weight pieces are read by hand ONE TILE AHEAD into two register sets used alternately (tower_bf3.hip: left to itself the compiler
            // reads a tile's three pieces, waits, issues its six matrix instructions and only then reads the next tile's -- every tile's LDS
            // latency exposed: 2 600 cycles per channel and wave for 770 of matrix work).  ds_read_b128 results return in order: lgkmcnt(3) =
            // "this tile's three pieces are here, the next tile's may still be in flight".
            const unsigned int wl = (unsigned int)(size_t)(cq_smem + slot * STAGE + lane * 16);
            cq_u32x4 wq[2][3];
#define CQ_DS_READ(dst, off) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst) : "v"(wl), "n"(off))
            CQ_DS_READ(wq[0][0], 0);
            CQ_DS_READ(wq[0][1], 1024);
            CQ_DS_READ(wq[0][2], 2048);
#pragma unroll
            for (int ht = 0; ht < HT; ++ht) {
                if (ht + 1 < HT) {
                    CQ_DS_READ(wq[(ht + 1) & 1][0], ((ht + 1) * 3 + 0) * 1024);
                    CQ_DS_READ(wq[(ht + 1) & 1][1], ((ht + 1) * 3 + 1) * 1024);
                    CQ_DS_READ(wq[(ht + 1) & 1][2], ((ht + 1) * 3 + 2) * 1024);
                    asm volatile("s_waitcnt lgkmcnt(3)" : "+v"(wq[ht & 1][0]), "+v"(wq[ht & 1][1]), "+v"(wq[ht & 1][2]));
                } else {
                    asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(wq[ht & 1][0]), "+v"(wq[ht & 1][1]), "+v"(wq[ht & 1][2]));
                }
                __builtin_amdgcn_sched_barrier(0);
                const cq_bf16x8 a0 = __builtin_bit_cast(cq_bf16x8, wq[ht & 1][0]);
                const cq_bf16x8 a1 = __builtin_bit_cast(cq_bf16x8, wq[ht & 1][1]);
                const cq_bf16x8 a2 = __builtin_bit_cast(cq_bf16x8, wq[ht & 1][2]);
                cq_f32x4 c = acc[ht];                               // the six products of weight >= 2^-16, smallest first
                c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a0, x[2], c, 0, 0, 0);
                c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a2, x[0], c, 0, 0, 0);
                c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a1, x[1], c, 0, 0, 0);
                c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a0, x[1], c, 0, 0, 0);
                c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a1, x[0], c, 0, 0, 0);
                c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a0, x[0], c, 0, 0, 0);
                acc[ht] = c;
                __builtin_amdgcn_sched_barrier(0);
            }
#undef CQ_DS_READ
            // channel i + 1's pieces (issued a channel ago) have landed; the NS + 1 loads of channel i + 2 may stay in flight
            if constexpr (NS == 1) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
            else if constexpr (NS == 2) asm volatile("s_waitcnt vmcnt(3)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
            __syncthreads();                                      // ... everyone's have, and everyone is done reading this slot
            slot = slot + 1 < CQ_RING ? slot + 1 : 0;
        }
        // D layout: lane (kk, r) holds pooled[sample r][16 ht + 4 kk + g]
        if (b < B) {
#pragma unroll
            for (int ht = 0; ht < HT; ++ht) {
                const int col = 16 * ht + 4 * kk;                 // H % 4 == 0: the lane's four columns are inside or outside together
                if (col < H) *reinterpret_cast<float4*>(pooled + b * pooled_ld + col) = make_float4(acc[ht][0], acc[ht][1], acc[ht][2], acc[ht][3]);
            }
        }
    }
}

}  // namespace

bool cin_pooled_fused_covers(int m, int Hp, int H, int D) { return D == CQ_D && m >= 1 && m <= CQ_MP && Hp >= 1 && H >= 4 && H <= 128 && !(H & 3); }

}  // namespace dir

using namespace dir;

extern "C" int64_t dir_cin_pooled_image_bytes(int m, int Hp, int H, int D) {
    if (!cin_pooled_fused_covers(m, Hp, H, D)) return 0;
    return (int64_t)Hp * ((H + 15) / 16) * 3 * 1024;
}

extern "C" int dir_cin_pooled_pack_f32(const float* W, int m, int Hp, int H, int D, void* image, int64_t image_bytes, dir_stream_t stream) {
    const char* name = "dir_cin_pooled_pack_f32";
    if (!cin_pooled_fused_covers(m, Hp, H, D))
        return fail(DIR_E_UNSUPPORTED, "%s: covers D = 16, m <= 32, H <= 128 (a multiple of 4) (m=%d Hp=%d H=%d D=%d)", name, m, Hp, H, D);
    DIR_CHECK_ARG(W && image, "%s: null pointer", name);
    DIR_CHECK_ARG(aligned16(image) && image_bytes >= dir_cin_pooled_image_bytes(m, Hp, H, D), "%s: image must be 16-byte aligned and hold "
                  "dir_cin_pooled_image_bytes(m, Hp, H, D) bytes", name);
    const int HT = (H + 15) / 16;
    const int64_t threads = (int64_t)Hp * HT * 64 * 4;
    hipLaunchKernelGGL(cin_pooled_pack_k, dim3((unsigned)((threads + 255) / 256)), dim3(256), 0, as_stream(stream), W, m, Hp, H, HT,
                       static_cast<unsigned int*>(image));
    DIR_CHECK_LAUNCH(name);
    return DIR_OK;
}

static int cin_pooled_run(const char* name, const float* x0, const int64_t* x0_inv, const float* xk, const void* image, int m, int Hp, int H, int D,
                          int64_t B, float* pooled, int64_t pooled_ld, dir_stream_t stream) {
    DIR_CHECK_ARG(B >= 0 && m > 0 && Hp > 0 && H > 0, "%s: B=%lld m=%d Hp=%d H=%d", name, (long long)B, m, Hp, H);
    if (!cin_pooled_fused_covers(m, Hp, H, D))
        return fail(DIR_E_UNSUPPORTED, "%s: covers D = 16, m <= 32, H <= 128 (a multiple of 4) (m=%d Hp=%d H=%d D=%d)", name, m, Hp, H, D);
    if (B == 0) return DIR_OK;
    DIR_CHECK_ARG(x0 && xk && image && pooled, "%s: null pointer", name);
    if (!aligned16(x0) || !aligned16(xk) || !aligned16(image) || !aligned16(pooled) || (pooled_ld & 3) || pooled_ld < H)
        return fail(DIR_E_BADARG, "%s: x0 / xk / image / pooled must be 16-byte aligned, pooled_ld a multiple of 4 and >= H", name);
    const int HT = (H + 15) / 16;
    const int64_t ntiles = (B + CQ_ROWS - 1) / CQ_ROWS;
    const int64_t nwg = ntiles < kCUs ? ntiles : kCUs;            // persistent workgroups, one per CU (8 waves x <= 256 registers)
    const size_t shmem = (size_t)CQ_RING * ((size_t)HT * 3 * 1024 + CQ_NW * 1024);
    hipStream_t st = as_stream(stream);
    const unsigned char* im = static_cast<const unsigned char*>(image);
#define DIR_CQ(HT_) hipLaunchKernelGGL((cin_pooled_k<HT_>), dim3((unsigned)nwg), dim3(64 * CQ_NW), shmem, st, x0, xk, im, m, Hp, H, B, pooled, pooled_ld, x0_inv)
    switch (HT) {
        case 1: DIR_CQ(1); break;
        case 2: DIR_CQ(2); break;
        case 3: DIR_CQ(3); break;
        case 4: DIR_CQ(4); break;
        case 5: DIR_CQ(5); break;
        case 6: DIR_CQ(6); break;
        case 7: DIR_CQ(7); break;
        default: DIR_CQ(8); break;
    }
#undef DIR_CQ
    DIR_CHECK_LAUNCH(name);
    return DIR_OK;
}

extern "C" int dir_cin_pooled_last_bf16x3_f32(const float* x0, const float* xk, const void* image, int m, int Hp, int H, int D, int64_t B,
                                              float* pooled, int64_t pooled_ld, dir_stream_t stream) {
    return cin_pooled_run("dir_cin_pooled_last_bf16x3_f32", x0, nullptr, xk, image, m, Hp, H, D, B, pooled, pooled_ld, stream);
}

// ... with x0 read through inverse positions (the sharded lookup without its finish pass, include/dir_hip.h): x0_rows [n, D] row list,
// x0_inv [B, m] int64 positions into it (< 0: a zero row -- a pruned / out-of-range id)
extern "C" int dir_cin_pooled_last_bf16x3_gather_f32(const float* x0_rows, const int64_t* x0_inv, const float* xk, const void* image, int m, int Hp,
                                                     int H, int D, int64_t B, float* pooled, int64_t pooled_ld, dir_stream_t stream) {
    DIR_CHECK_ARG(x0_inv || B == 0, "dir_cin_pooled_last_bf16x3_gather_f32: x0_inv is null");
    return cin_pooled_run("dir_cin_pooled_last_bf16x3_gather_f32", x0_rows, x0_inv, xk, image, m, Hp, H, D, B, pooled, pooled_ld, stream);
}

