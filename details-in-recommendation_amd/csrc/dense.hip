// dense.hip -- Y = act(X . Wt^T + bias) on fp32 MFMA for gfx950: the hidden layers of the reference's DNN towers.
//
// Reference: dnn_logit_fn, models/DeepFM/deepFM.py:295-300 (tf.layers.dense(units, activation)), _deep_architecture,
// models/DeepCrossNetwork/DeepCrossNetwork.py:394-399, _base_model, models/ESMM/ESMM.py:139-142.  [TF-upstream] dense =
// matmul + bias add + activation; fp32 throughout.  Wt is the [N, Kd] transpose of the TF kernel (torch's nn.Linear.weight).
//
// M = batch rows (65 536), Kd and N a few hundred: too skinny for the library's tiles (rocBLAS picks 32x256 macro tiles and
// reaches 0.56 of the fp32 MFMA peak at 65 536 x 416 x 400, with the activation as a second pass over Y).  Here a 256-thread
// workgroup owns a 128 x NT tile of Y (NT = 80 divides the 400-wide layers exactly; 128 otherwise); wave w owns rows
// [32w, 32w+32) x all NT columns as 2 x NT/16 accumulators of v_mfma_f32_16x16x4_f32.  X and Wt stream through LDS in 16- or 32-wide
// k chunks, double-buffered (the next chunk's global loads are issued before the current chunk's MFMAs, stored after them: one
// barrier per chunk).  Both operands are k-contiguous in LDS (row stride chunk + 4 words: the sixteen rows of a 16-byte fragment read
// land in sixteen different bank quads), so a lane's four k-steps come from ONE ds_read_b128: 14 LDS reads per 80 MFMAs at
// NT = 80.  Bias and ReLU are applied to the accumulators.  Workgroup ids are remapped so that the N-blocks of one row block run
// on the same XCD (they share the X tile through that XCD's L2).
#include <type_traits>

#include "common.hpp"

namespace dir {

typedef float f32x4d __attribute__((ext_vector_type(4)));

// 16-byte buffer load: resource base + lane byte offset (VGPR) + wave-uniform byte offset (SGPR); lanes past the resource's size
// read zeros
__device__ __forceinline__ float4 buf_load4(__amdgpu_buffer_rsrc_t r, uint32_t lane_off, uint32_t wave_off) {
    const auto raw = __builtin_amdgcn_raw_buffer_load_b128(r, (int)lane_off, (int)wave_off, 0);
    const f32x4d v = __builtin_bit_cast(f32x4d, raw);   // (assigning the builtin's result to an int vector type splats .x)
    return make_float4(v.x, v.y, v.z, v.w);
}

constexpr int DN_MT = 128;          // rows per workgroup

template <int NT, bool RELU, int KC>
__global__ __launch_bounds__(256, KC == 16 ? 4 : 2) void dense_k(const float* __restrict__ X, int64_t x_ld, const float* __restrict__ Wt,
                                                    int64_t w_ld, const float* __restrict__ bias, int64_t M, int Kd, int N,
                                                    float* __restrict__ Y, int64_t y_ld, int nb, int remap, int vec_out,
                                                    const float* __restrict__ gate, int64_t gate_ld,
                                                    const float* __restrict__ pscale, const float* __restrict__ pshift) {
    constexpr int NTILES = NT / 16;
    constexpr int DN_KC = KC, DN_LS = KC + 4;      // k chunk, LDS row stride (floats)
    constexpr int TPR = KC / 4;                    // threads (16-byte pieces) per staged row
    constexpr int AJ = DN_MT * TPR / 256, BJ = (NT * TPR + 255) / 256;
    extern __shared__ __attribute__((aligned(16))) float dense_smem[];
    float (*As)[DN_MT * DN_LS] = reinterpret_cast<float (*)[DN_MT * DN_LS]>(dense_smem);
    float (*Bs)[NT * DN_LS] = reinterpret_cast<float (*)[NT * DN_LS]>(dense_smem + 2 * DN_MT * DN_LS);
    const int tid = threadIdx.x;
    const int lane = tid & 63, w = tid >> 6, r16 = lane & 15, kk = lane >> 4;

    int id = blockIdx.x;
    if (remap) id = (id % kXCDs) * ((int)gridDim.x / kXCDs) + id / kXCDs;      // round-robin XCD dispatch -> contiguous ids per XCD
    const int nblk = id % nb;
    const int64_t m0 = (int64_t)(id / nb) * DN_MT;
    const int n0 = nblk * NT;

    // Staging: thread <-> (row, 16-byte column c) of the k chunk.  Rows beyond M / N are clamped to the last valid row (their
    // results are dropped by the epilogue), so the full chunks need no predicate; only a k tail (Kd % 32) is masked.
    // Staging: thread <-> (row, 16-byte column c) of the k chunk, as buffer loads: address = resource base + per-lane byte offset
    // (fixed for the whole kernel) + wave-uniform k offset in an SGPR -- no address arithmetic on the VALU, which fp32 MFMAs do
    // not hide -- and rows beyond M / N read as zeros through the resource's range check.  Only a k tail (Kd % KC) is masked.
    const int64_t rows_x = (M - m0) < DN_MT ? (M - m0) : DN_MT;
    const int rows_w = (N - n0) < NT ? (N - n0) : NT;
    const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(X + m0 * x_ld), 0,
                                                                        (int)(uint32_t)(((rows_x - 1) * x_ld + Kd) * 4), 0x00020000);
    const __amdgpu_buffer_rsrc_t rw = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(Wt + (int64_t)n0 * w_ld), 0,
                                                                        (int)(uint32_t)((((int64_t)rows_w - 1) * w_ld + Kd) * 4), 0x00020000);
    uint32_t xo[AJ], wo[BJ];
#pragma unroll
    for (int j = 0; j < AJ; ++j) {
        const int i = tid + 256 * j, row = i / TPR, c = i % TPR;
        xo[j] = (uint32_t)((row * x_ld + 4 * c) * 4);
    }
#pragma unroll
    for (int j = 0; j < BJ; ++j) {
        const int i = tid + 256 * j, row = i / TPR, c = i % TPR;
        wo[j] = row < NT ? (uint32_t)((row * w_ld + 4 * c) * 4) : 0xfffffff0u;
    }
    const int nfull = Kd / DN_KC;
    const int nchunks = (Kd + DN_KC - 1) / DN_KC;
    float4 ar[AJ], br[BJ];
    auto gload_full = [&](int ch) {
        const uint32_t kb = (uint32_t)(ch * DN_KC * 4);
#pragma unroll
        for (int j = 0; j < AJ; ++j) ar[j] = buf_load4(rx, xo[j], kb);
#pragma unroll
        for (int j = 0; j < BJ; ++j) br[j] = buf_load4(rw, wo[j], kb);
    };
    auto gload_masked = [&](int ch) {                      // any chunk; lanes past Kd get an out-of-range offset (-> zeros)
        const uint32_t kb = (uint32_t)(ch * DN_KC * 4);
        const uint32_t big = (ch * DN_KC + 4 * (tid % TPR) < Kd) ? 0u : 0xfffffff0u;
#pragma unroll
        for (int j = 0; j < AJ; ++j) ar[j] = buf_load4(rx, xo[j] | big, kb);
#pragma unroll
        for (int j = 0; j < BJ; ++j) br[j] = buf_load4(rw, wo[j] | big, kb);
    };
    auto lstore = [&](int buf) {
#pragma unroll
        for (int j = 0; j < AJ; ++j) {
            const int i = tid + 256 * j, row = i / TPR, c = i % TPR;
            *reinterpret_cast<float4*>(&As[buf][row * DN_LS + 4 * c]) = ar[j];
        }
#pragma unroll
        for (int j = 0; j < BJ; ++j) {
            const int i = tid + 256 * j, row = i / TPR, c = i % TPR;
            if (row < NT) *reinterpret_cast<float4*>(&Bs[buf][row * DN_LS + 4 * c]) = br[j];
        }
    };

    f32x4d acc[2][NTILES];
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
        for (int nt = 0; nt < NTILES; ++nt) acc[mt][nt] = (f32x4d){0.f, 0.f, 0.f, 0.f};

    auto compute = [&](int buf) {
        const float* ab = &As[buf][(32 * w + r16) * DN_LS + kk * (KC / 4)];
        const float* bb = &Bs[buf][r16 * DN_LS + kk * (KC / 4)];
#pragma unroll
        for (int q = 0; q < KC / 16; ++q) {
            const float4 a0 = *reinterpret_cast<const float4*>(ab + 4 * q);
            const float4 a1 = *reinterpret_cast<const float4*>(ab + 16 * DN_LS + 4 * q);
            const float av0[4] = {a0.x, a0.y, a0.z, a0.w}, av1[4] = {a1.x, a1.y, a1.z, a1.w};
#pragma unroll
            for (int nt = 0; nt < NTILES; ++nt) {
                const float4 b4 = *reinterpret_cast<const float4*>(bb + nt * 16 * DN_LS + 4 * q);
                const float bv[4] = {b4.x, b4.y, b4.z, b4.w};
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    acc[0][nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(av0[e], bv[e], acc[0][nt], 0, 0, 0);
                    acc[1][nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(av1[e], bv[e], acc[1][nt], 0, 0, 0);
                }
            }
        }
    };

    gload_masked(0);
    lstore(0);
    __syncthreads();
    int ch = 0;
    for (; ch + 2 < nfull; ch += 2) {       // the steady state, two chunks per trip so that every LDS address is base + immediate:
        gload_full(ch + 1);                 // the next chunk is a full one (no predicate anywhere in the loop)
        compute(0);
        lstore(1);
        __syncthreads();
        gload_full(ch + 2);
        compute(1);
        lstore(0);
        __syncthreads();
    }
    for (; ch + 1 < nfull; ++ch) {
        gload_full(ch + 1);
        compute(ch & 1);
        lstore((ch + 1) & 1);
        __syncthreads();
    }
    for (; ch < nchunks; ++ch) {            // at most two iterations: the last full chunk (prefetching a k tail) and the tail
        if (ch + 1 < nchunks) gload_masked(ch + 1);
        compute(ch & 1);
        if (ch + 1 < nchunks) lstore((ch + 1) & 1);
        __syncthreads();
    }

    // epilogue: bias + activation on the accumulators (lane: rows 4 kk + g of its row tile, column r16 of its column tile).
    // vec_out: the wave's 32 x NT block goes through LDS (the operand buffers are dead after the loop's last barrier) and leaves
    // as whole rows, 16 bytes per lane -- 4-byte stores in 64-byte pieces made the Y write the largest fixed cost of a tile.
    if (vec_out) {
        constexpr int EPS = NT + 4;
        float* ep = dense_smem + w * (16 * EPS);                   // 16 rows at a time: the staging fits the smallest operand image
        // Rows leave through buffer stores: resource = this wave's 32 rows of Y, lane offset fixed per pass (rows past M and columns
        // past N get an out-of-range offset and are dropped), row-group offset in an SGPR -- no 64-bit address arithmetic and no
        // integer division in the loop (the previous form spent 470 VALU instructions per tile here).  A row's NT/4 16-byte pieces
        // are covered as whole groups of 16 lanes (pass A: lanes <-> 4 rows x 16 pieces) plus, for NT = 80, the last 4 pieces
        // (pass B: lanes <-> 16 rows x 4 pieces).
        const int64_t wrow0 = m0 + 32 * w;
        const int64_t wrows = (M - wrow0) < 32 ? (M - wrow0) : 32;
        if (wrows <= 0) return;                                    // wave-uniform
        const __amdgpu_buffer_rsrc_t ry = __builtin_amdgcn_make_buffer_rsrc(Y + wrow0 * y_ld + n0, 0,
                                                                            (int)(uint32_t)(((wrows - 1) * y_ld + (N - n0)) * 4), 0x00020000);
        const __amdgpu_buffer_rsrc_t rg = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(gate ? gate + wrow0 * gate_ld + n0 : Y), 0,
                                                                            gate ? (int)(uint32_t)(((wrows - 1) * gate_ld + (N - n0)) * 4) : 0, 0x00020000);
        auto put = [&](int lrow, int c4, int srow) {               // row srow + lrow of the wave's block (srow wave-uniform), 16-byte piece c4
            float4 v = *reinterpret_cast<const float4*>(ep + (lrow + (srow & 15)) * EPS + 4 * c4);
            const bool ok = (n0 + 4 * c4 < N) && (lrow + srow < (int)wrows);   // explicit: the SGPR part of the offset is not relied on for range checks
            if (gate) {      // backward through the previous layer's ReLU: pass the value where that layer's output was positive
                const float4 gt = buf_load4(rg, ok ? (uint32_t)((lrow * gate_ld + 4 * c4) * 4) : 0xfffffff0u, (uint32_t)(srow * gate_ld * 4));
                v.x = gt.x > 0.f ? v.x : 0.f;
                v.y = gt.y > 0.f ? v.y : 0.f;
                v.z = gt.z > 0.f ? v.z : 0.f;
                v.w = gt.w > 0.f ? v.w : 0.f;
            }
            const f32x4d vv = {v.x, v.y, v.z, v.w};
            __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(decltype(__builtin_amdgcn_raw_buffer_load_b128(ry, 0, 0, 0)), vv), ry,
                                                   ok ? (int)((lrow * y_ld + 4 * c4) * 4) : (int)0xfffffff0u, (int)(srow * y_ld * 4), 0);
        };
        constexpr int C4 = NT / 4, FULL = C4 / 16, REST = C4 % 16;
        static_assert(REST == 0 || REST == 4, "NT/4 must be a multiple of 16, or 4 more");
#pragma unroll
        for (int mt = 0; mt < 2; ++mt) {       // (a wave's LDS operations execute in order: the second half's writes follow the first half's reads)
#pragma unroll
            for (int nt = 0; nt < NTILES; ++nt) {
                const int col = n0 + 16 * nt + r16;
                const float bcol = (bias && col < N) ? bias[col] : 0.f;
                const float sc = (pscale && col < N) ? pscale[col] : 1.f, sf = (pscale && col < N) ? pshift[col] : 0.f;
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    float v = acc[mt][nt][g] + bcol;
                    if (RELU) v = fmaxf(v, 0.f);
                    if (pscale) v = v * sc + sf;     // the inference batch-norm that follows the activation, per column
                    ep[(4 * kk + g) * EPS + 16 * nt + r16] = v;
                }
            }
#pragma unroll
            for (int cb = 0; cb < FULL; ++cb)
#pragma unroll
                for (int i = 0; i < 4; ++i) put(lane >> 4, 16 * cb + (lane & 15), 16 * mt + 4 * i);
            if (REST == 4) put(lane >> 2, 16 * FULL + (lane & 3), 16 * mt);
        }
        return;
    }
#pragma unroll
    for (int nt = 0; nt < NTILES; ++nt) {
        const int col = n0 + 16 * nt + r16;
        const float bcol = (bias && col < N) ? bias[col] : 0.f;
        const float sc = (pscale && col < N) ? pscale[col] : 1.f, sf = (pscale && col < N) ? pshift[col] : 0.f;
#pragma unroll
        for (int mt = 0; mt < 2; ++mt)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int64_t row = m0 + 32 * w + 16 * mt + 4 * kk + g;
                float v = acc[mt][nt][g] + bcol;
                if (RELU) v = fmaxf(v, 0.f);
                if (pscale) v = v * sc + sf;
                if (row < M && col < N) {
                    if (gate && !(gate[row * gate_ld + col] > 0.f)) v = 0.f;
                    Y[row * y_ld + col] = v;
                }
            }
    }
}

template <int NT, int KC>
static void launch_dense(hipStream_t st, const float* X, int64_t x_ld, const float* Wt, int64_t w_ld, const float* bias, int act, int64_t M,
                         int Kd, int N, float* Y, int64_t y_ld, const float* gate, int64_t gate_ld, const float* ps, const float* psh) {
    const int nb = (N + NT - 1) / NT;
    const int64_t mb = (M + DN_MT - 1) / DN_MT;
    const int64_t total = mb * nb;
    const int remap = (total % kXCDs) == 0 ? 1 : 0;
    const int vec_out = ((N & 3) == 0 && (y_ld & 3) == 0 && aligned16(Y) && (!gate || ((gate_ld & 3) == 0 && aligned16(gate)))) ? 1 : 0;
    size_t shmem = sizeof(float) * 2 * (DN_MT + NT) * (KC + 4);       // 60 KB (NT = 80) / 74 KB (NT = 128) at KC = 32
    const size_t ep_bytes = sizeof(float) * 4 * 16 * (NT + 4);             // the epilogue's row staging (16 rows per wave at a time)
    if (shmem < ep_bytes) shmem = ep_bytes;
    static LdsOnce once;           // (the limit is the chip's: the first call's own size would be too small for a later, larger one)
    (void)lds_limit(once, 160 * 1024, &dense_k<NT, true, KC>, &dense_k<NT, false, KC>);
    if (act)
        hipLaunchKernelGGL((dense_k<NT, true, KC>), dim3((unsigned)total), dim3(256), shmem, st, X, x_ld, Wt, w_ld, bias, M, Kd, N, Y, y_ld, nb, remap, vec_out, gate, gate_ld, ps, psh);
    else
        hipLaunchKernelGGL((dense_k<NT, false, KC>), dim3((unsigned)total), dim3(256), shmem, st, X, x_ld, Wt, w_ld, bias, M, Kd, N, Y, y_ld, nb, remap, vec_out, gate, gate_ld, ps, psh);
}


// ---- small batches (round 5): the reference's own operating point, batch 100 / 256 (models/DeepCrossNetwork/train.py:16-17) ----------------
// At a few hundred rows a layer is a few tens of MFLOP: nothing is bound but the LENGTH of the longest dependent chain.  dense_k gives a
// 128-row workgroup the whole reduction (26 staged chunks, a barrier each: ~30 us per 400-wide layer whatever M is) and the fused tower one
// workgroup per 128 rows for all its layers (~110 us).  Here ONE WAVE owns a 16 x 16 tile of Y: its operands come straight from L2 /
// HBM as 16-byte loads (lane (kk, r) reads X[row r][16 j + 4 kk ..+3] and Wt[col r][16 j + 4 kk ..+3]: element e of both feeds the j-th
// step's e-th v_mfma_f32_16x16x4_f32, whose four k slots are then {e, 4 + e, 8 + e, 12 + e} + 16 j on both sides), no LDS, no barrier; four
// independent accumulators (one per e) keep the matrix pipe's 8-pass latency out of the chain, and the loads of eight steps are in flight
// at once; the four waves of a workgroup split the reduction of ONE tile (partials through LDS, added in wave order).  (M / 16) x (N / 16)
// workgroups: 400 at 256 x 400 -- a layer is a few microseconds, and a forward captured in a HIP graph is the sum of them.  fp32-input
// MFMA: exact fp32 products, fp32 accumulation.
template <bool RELU, int RT /* row tiles of 16 per workgroup: 2 halves the weight traffic of the wide layers (1024 x 1024 at M = 256) */>
__global__ __launch_bounds__(256) void dense_small_k(const float* __restrict__ X, int64_t x_ld, const float* __restrict__ Wt, int64_t w_ld,
                                                     const float* __restrict__ bias, int64_t M, int Kd, int N, float* __restrict__ Y, int64_t y_ld,
                                                     const float* __restrict__ pscale, const float* __restrict__ pshift) {
    // the workgroup's four waves share ONE 16 RT x 16 tile: wave w takes the w-th quarter of the k steps (the chain is a quarter as long),
    // waves 1..3 leave their partial tiles in LDS and wave 0 adds them in wave order (a fixed order: bitwise reproducible)
    __shared__ float4 part[3][RT][64];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, r = lane & 15, kk = lane >> 4;
    const int64_t row0 = (int64_t)blockIdx.y * (16 * RT);
    const int col0 = blockIdx.x * 16;
    const int wr = col0 + r < N ? col0 + r : N - 1;                     // rows / columns past the end: clamped loads, results dropped
    const float* wp = Wt + (int64_t)wr * w_ld + 4 * kk;
    const float* xp[RT];
#pragma unroll
    for (int t = 0; t < RT; ++t) {
        const int64_t xr = row0 + 16 * t + r < M ? row0 + 16 * t + r : M - 1;
        xp[t] = X + xr * x_ld + 4 * kk;
    }
    f32x4d acc[RT][4];
#pragma unroll
    for (int t = 0; t < RT; ++t)
#pragma unroll
        for (int e = 0; e < 4; ++e) acc[t][e] = (f32x4d){0.f, 0.f, 0.f, 0.f};
    const int steps = (Kd + 15) / 16;
    const int jb = (steps * w) / 4, je = (steps * (w + 1)) / 4;          // this wave's k steps
    constexpr int U = 8;
    for (int j0 = jb; j0 < je; j0 += U) {
        float4 xa[U][RT], wb[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int k = 16 * (j0 + u) + 4 * kk;
            const bool ok = j0 + u < je && k < Kd;                       // Kd % 4 == 0: a 16-byte piece is inside the row or outside it
            wb[u] = ok ? *reinterpret_cast<const float4*>(wp + 16 * (j0 + u)) : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
            for (int t = 0; t < RT; ++t)
                xa[u][t] = ok ? *reinterpret_cast<const float4*>(xp[t] + 16 * (j0 + u)) : make_float4(0.f, 0.f, 0.f, 0.f);
        }
#pragma unroll
        for (int u = 0; u < U; ++u)
#pragma unroll
            for (int t = 0; t < RT; ++t) {
                acc[t][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(xa[u][t].x, wb[u].x, acc[t][0], 0, 0, 0);
                acc[t][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(xa[u][t].y, wb[u].y, acc[t][1], 0, 0, 0);
                acc[t][2] = __builtin_amdgcn_mfma_f32_16x16x4f32(xa[u][t].z, wb[u].z, acc[t][2], 0, 0, 0);
                acc[t][3] = __builtin_amdgcn_mfma_f32_16x16x4f32(xa[u][t].w, wb[u].w, acc[t][3], 0, 0, 0);
            }
    }
    float4 tv[RT];
#pragma unroll
    for (int t = 0; t < RT; ++t) {
        tv[t].x = (acc[t][0][0] + acc[t][1][0]) + (acc[t][2][0] + acc[t][3][0]);
        tv[t].y = (acc[t][0][1] + acc[t][1][1]) + (acc[t][2][1] + acc[t][3][1]);
        tv[t].z = (acc[t][0][2] + acc[t][1][2]) + (acc[t][2][2] + acc[t][3][2]);
        tv[t].w = (acc[t][0][3] + acc[t][1][3]) + (acc[t][2][3] + acc[t][3][3]);
        if (w > 0) part[w - 1][t][lane] = tv[t];
    }
    __syncthreads();
    if (w > 0) return;
    // D layout: lane (kk, r) holds Y[row0 + 16 t + 4 kk + g][col0 + r]
    const int col = col0 + r;
    if (col >= N) return;
    const float b = bias ? bias[col] : 0.f;
    const float sc = pscale ? pscale[col] : 1.f, sf = pscale ? pshift[col] : 0.f;
#pragma unroll
    for (int t = 0; t < RT; ++t) {
        float4 s4 = tv[t];
#pragma unroll
        for (int q = 0; q < 3; ++q) {
            const float4 p = part[q][t][lane];
            s4.x += p.x; s4.y += p.y; s4.z += p.z; s4.w += p.w;
        }
        const float sv[4] = {s4.x, s4.y, s4.z, s4.w};
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const int64_t row = row0 + 16 * t + 4 * kk + g;
            float v = sv[g] + b;
            if (RELU) v = v > 0.f ? v : 0.f;
            if (pscale) v = v * sc + sf;
            if (row < M) Y[row * y_ld + col] = v;
        }
    }
}


// ---- mid-size batches (round 6): 512 < M < ~6 000 rows, the only operating range the reference documents beyond its own 100 / 256
// (models/DeepCrossNetwork/train.py:16-17 sets the batch by flag).  dense_small_k's one-tile workgroups re-read both operands from L2 for
// every 16 x 16 outputs (170 MB of L2 traffic at 2048 x 400 x 416: 23 us), dense_k's 128-row workgroups leave most of the chip idle (16 row
// blocks at 2048 rows: 27 us whatever N is) -- the library's 14 us was what ran there.  Here a workgroup owns a (32 RT) x 64 tile of Y: its four
// waves sit 2 x 2 on it (RT x 2 accumulator tiles of 16 x 16 each), the reduction comes through LDS in chunks of 32 / 64 (registers -> LDS,
// double-buffered: the next chunk's global loads are in flight under this chunk's matrix instructions; one barrier per chunk), operands are
// read with one ds_read_b128 per tile and 16 k (row stride chunk + 8 floats: conflict-free for the 16-lane groups of a b128 read), element e of a
// lane's four floats feeding the e-th v_mfma_f32_16x16x4_f32 of the step on both sides (dense_small_k's enumeration of the k slots).
// fp32-input MFMA: exact products, fp32 accumulation.  224 workgroups at 2048 x 400: 8 us.
template <bool RELU, int RT /* 16-row tiles per wave: the workgroup's tile is 32 RT rows x 64 columns */, int KC /* reduction chunk: 32 | 64 */>
__global__ __launch_bounds__(256) void dense_mid_k(const float* __restrict__ X, int64_t x_ld, const float* __restrict__ Wt, int64_t w_ld,
                                                   const float* __restrict__ bias, int64_t M, int Kd, int N, float* __restrict__ Y, int64_t y_ld,
                                                   const float* __restrict__ pscale, const float* __restrict__ pshift) {
    constexpr int ROWS = 32 * RT;
    constexpr int LD = KC + 8;                               // row stride 40 / 72 floats: conflict-free for the 16-lane groups of a ds_read_b128
    constexpr int TPR = KC / 4;                              // threads per staged row (16-byte pieces)
    constexpr int RPT = 256 / TPR;                           // rows one trip of the 256 threads covers: 32 / 16
    constexpr int XT = ROWS / RPT, WT = 64 / RPT;            // staging trips for the x / weight tile
    __shared__ __attribute__((aligned(16))) float xs[2][ROWS * LD];
    __shared__ __attribute__((aligned(16))) float ws[2][64 * LD];
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, r = lane & 15, kk = lane >> 4;
    const int wm = w >> 1, wn = w & 1;                       // the wave's place on the tile: rows 16 RT wm .., columns 32 wn ..
    const int64_t row0 = (int64_t)blockIdx.y * ROWS;
    const int col0 = blockIdx.x * 64;
    // staging: thread t moves the 16-byte piece (row t / TPR [+ RPT per trip], floats 4 (t % TPR) ..) of a chunk; rows / columns past the end
    // are clamped (their results are dropped), pieces past Kd are zeros
    const int sr = tid / TPR, sk = 4 * (tid % TPR);
    const float* xg[XT];
    const float* wg[WT];
#pragma unroll
    for (int t = 0; t < XT; ++t) {
        const int64_t xr = row0 + sr + RPT * t < M ? row0 + sr + RPT * t : M - 1;
        xg[t] = X + xr * x_ld + sk;
    }
#pragma unroll
    for (int t = 0; t < WT; ++t) {
        const int wc = col0 + sr + RPT * t < N ? col0 + sr + RPT * t : N - 1;
        wg[t] = Wt + (int64_t)wc * w_ld + sk;
    }
    const int nchunk = (Kd + KC - 1) / KC;
    // the next chunk's global loads (registers) are in flight under this chunk's matrix instructions; one barrier per chunk
    float4 xr_[XT], wr_[WT];
    auto gload = [&](int c) {
        const int k = KC * c + sk;
        const bool ok = k < Kd;                              // Kd % 4 == 0: a piece is inside the row or outside it
#pragma unroll
        for (int t = 0; t < XT; ++t) xr_[t] = ok ? *reinterpret_cast<const float4*>(xg[t] + KC * c) : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
        for (int t = 0; t < WT; ++t) wr_[t] = ok ? *reinterpret_cast<const float4*>(wg[t] + KC * c) : make_float4(0.f, 0.f, 0.f, 0.f);
    };
    auto lstore = [&](int b) {
#pragma unroll
        for (int t = 0; t < XT; ++t) *reinterpret_cast<float4*>(&xs[b][(sr + RPT * t) * LD + sk]) = xr_[t];
#pragma unroll
        for (int t = 0; t < WT; ++t) *reinterpret_cast<float4*>(&ws[b][(sr + RPT * t) * LD + sk]) = wr_[t];
    };
    f32x4d acc[RT][2];
#pragma unroll
    for (int t = 0; t < RT; ++t)
#pragma unroll
        for (int u = 0; u < 2; ++u) acc[t][u] = (f32x4d){0.f, 0.f, 0.f, 0.f};
    gload(0);
    lstore(0);
    __syncthreads();
    for (int c = 0; c < nchunk; ++c) {
        const int b = c & 1;
        if (c + 1 < nchunk) gload(c + 1);
#pragma unroll
        for (int j = 0; j < KC / 16; ++j) {                  // 16-wide steps
            float4 a[RT], bq[2];
#pragma unroll
            for (int t = 0; t < RT; ++t) a[t] = *reinterpret_cast<const float4*>(&xs[b][(16 * RT * wm + 16 * t + r) * LD + 16 * j + 4 * kk]);
#pragma unroll
            for (int u = 0; u < 2; ++u) bq[u] = *reinterpret_cast<const float4*>(&ws[b][(32 * wn + 16 * u + r) * LD + 16 * j + 4 * kk]);
            // element e of the lanes' four floats feeds the step's e-th instruction on every tile: the tiles' chains interleave (a dependent
            // v_mfma_f32_16x16x4_f32 needs 40 cycles, issue is 32)
#pragma unroll
            for (int e = 0; e < 4; ++e)
#pragma unroll
                for (int t = 0; t < RT; ++t)
#pragma unroll
                    for (int u = 0; u < 2; ++u) {
                        const float av = e == 0 ? a[t].x : e == 1 ? a[t].y : e == 2 ? a[t].z : a[t].w;
                        const float bv = e == 0 ? bq[u].x : e == 1 ? bq[u].y : e == 2 ? bq[u].z : bq[u].w;
                        acc[t][u] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, bv, acc[t][u], 0, 0, 0);
                    }
        }
        if (c + 1 < nchunk) lstore(b ^ 1);                   // (the buffer the previous trip read: every wave passed the barrier below since)
        __syncthreads();
    }
    // D layout: lane (kk, r) holds Y[row 4 kk + g][col r] of its tile
#pragma unroll
    for (int u = 0; u < 2; ++u) {
        const int col = col0 + 32 * wn + 16 * u + r;
        if (col >= N) continue;
        const float bv = bias ? bias[col] : 0.f;
        const float sc = pscale ? pscale[col] : 1.f, sf = pscale ? pshift[col] : 0.f;
#pragma unroll
        for (int t = 0; t < RT; ++t)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int64_t row = row0 + 16 * RT * wm + 16 * t + 4 * kk + g;
                float v = acc[t][u][g] + bv;
                if (RELU) v = v > 0.f ? v : 0.f;
                if (pscale) v = v * sc + sf;
                if (row < M) Y[row * y_ld + col] = v;
            }
    }
}

}  // namespace dir

using namespace dir;

extern "C" int dir_dense_mid_f32(const float* X, int64_t x_ld, const float* Wt, int64_t w_ld, const float* bias, int act, const float* post_scale,
                                 const float* post_shift, int64_t M, int Kd, int N, float* Y, int64_t y_ld, dir_stream_t stream) {
    const char* name = "dir_dense_mid_f32";
    DIR_CHECK_ARG(M >= 0 && Kd > 0 && N > 0 && x_ld >= Kd && w_ld >= Kd && y_ld >= N, "%s: M=%lld Kd=%d N=%d x_ld=%lld w_ld=%lld y_ld=%lld", name,
                  (long long)M, Kd, N, (long long)x_ld, (long long)w_ld, (long long)y_ld);
    DIR_CHECK_ARG(act == DIR_ACT_NONE || act == DIR_ACT_RELU, "%s: act=%d", name, act);
    DIR_CHECK_ARG((post_scale == nullptr) == (post_shift == nullptr), "%s: post_scale and post_shift come together", name);
    if (M == 0) return DIR_OK;
    DIR_CHECK_ARG(X && Wt && Y, "%s: null pointer", name);
    if ((Kd & 3) || (x_ld & 3) || (w_ld & 3) || !aligned16(X) || !aligned16(Wt))
        return fail(DIR_E_UNSUPPORTED, "%s: Kd, x_ld and w_ld must be multiples of 4 and X / Wt 16-byte aligned (Kd=%d x_ld=%lld w_ld=%lld)", name, Kd,
                    (long long)x_ld, (long long)w_ld);
    if ((M + 31) / 32 > 65535) return fail(DIR_E_UNSUPPORTED, "%s: M=%lld (this entry is for batches of a few thousand rows)", name, (long long)M);
    // 64-row tiles where they still give every CU a workgroup, 32-row tiles below that
    const int64_t cb = (N + 63) / 64;
    const bool rt2 = ((M + 63) / 64) * cb >= kCUs;
    const int rows = rt2 ? 64 : 32;
    const dim3 grid((unsigned)cb, (unsigned)((M + rows - 1) / rows));
    hipStream_t st = as_stream(stream);
    const bool kc64 = Kd >= 128 && !rt2;                     // 64-wide chunks (half the barriers) for the 32-row tiles; the 64-row tiles keep four workgroups per CU
#define DIR_DM(RELU_, RT_, KC_) hipLaunchKernelGGL((dense_mid_k<RELU_, RT_, KC_>), grid, dim3(256), 0, st, X, x_ld, Wt, w_ld, bias, M, Kd, N, Y, y_ld, post_scale, post_shift)
#define DIR_DM2(RELU_, RT_) do { if (kc64) DIR_DM(RELU_, RT_, 64); else DIR_DM(RELU_, RT_, 32); } while (0)
    if (act) { if (rt2) DIR_DM2(true, 2); else DIR_DM2(true, 1); }
    else { if (rt2) DIR_DM2(false, 2); else DIR_DM2(false, 1); }
#undef DIR_DM2
#undef DIR_DM
    DIR_CHECK_LAUNCH(name);
    return DIR_OK;
}

extern "C" int dir_dense_small_f32(const float* X, int64_t x_ld, const float* Wt, int64_t w_ld, const float* bias, int act, const float* post_scale,
                                   const float* post_shift, int64_t M, int Kd, int N, float* Y, int64_t y_ld, dir_stream_t stream) {
    const char* name = "dir_dense_small_f32";
    DIR_CHECK_ARG(M >= 0 && Kd > 0 && N > 0 && x_ld >= Kd && w_ld >= Kd && y_ld >= N, "%s: M=%lld Kd=%d N=%d x_ld=%lld w_ld=%lld y_ld=%lld", name,
                  (long long)M, Kd, N, (long long)x_ld, (long long)w_ld, (long long)y_ld);
    DIR_CHECK_ARG(act == DIR_ACT_NONE || act == DIR_ACT_RELU, "%s: act=%d", name, act);
    DIR_CHECK_ARG((post_scale == nullptr) == (post_shift == nullptr), "%s: post_scale and post_shift come together", name);
    if (M == 0) return DIR_OK;
    DIR_CHECK_ARG(X && Wt && Y, "%s: null pointer", name);
    if ((Kd & 3) || (x_ld & 3) || (w_ld & 3) || !aligned16(X) || !aligned16(Wt))
        return fail(DIR_E_UNSUPPORTED, "%s: Kd, x_ld and w_ld must be multiples of 4 and X / Wt 16-byte aligned (Kd=%d x_ld=%lld w_ld=%lld)", name, Kd,
                    (long long)x_ld, (long long)w_ld);
    if ((M + 15) / 16 > 65535) return fail(DIR_E_UNSUPPORTED, "%s: M=%lld (this entry is for batches of at most ~1 M rows; dir_dense_f32 beyond a few thousand)", name, (long long)M);
    // two row tiles per workgroup where the weight matrix is large and the batch has rows to pair (halves the weight reads from L2)
    const bool rt2 = M > 64 && (int64_t)N * Kd >= 512 * 1024;
    const int rows = rt2 ? 32 : 16;
    const dim3 grid((unsigned)((N + 15) / 16), (unsigned)((M + rows - 1) / rows));
    hipStream_t st = as_stream(stream);
#define DIR_DS(RELU_, RT_) hipLaunchKernelGGL((dense_small_k<RELU_, RT_>), grid, dim3(256), 0, st, X, x_ld, Wt, w_ld, bias, M, Kd, N, Y, y_ld, post_scale, post_shift)
    if (act) { if (rt2) DIR_DS(true, 2); else DIR_DS(true, 1); }
    else { if (rt2) DIR_DS(false, 2); else DIR_DS(false, 1); }
#undef DIR_DS
    DIR_CHECK_LAUNCH(name);
    return DIR_OK;
}

static int dense_entry(const char* name, const float* X, int64_t x_ld, const float* Wt, int64_t w_ld, const float* bias, int act, int64_t M, int Kd,
                       int N, float* Y, int64_t y_ld, const float* gate, int64_t gate_ld, dir_stream_t stream, const float* ps = nullptr,
                       const float* psh = nullptr) {
    DIR_CHECK_ARG(M >= 0 && Kd > 0 && N > 0 && x_ld >= Kd && w_ld >= Kd && y_ld >= N, "%s: M=%lld Kd=%d N=%d x_ld=%lld w_ld=%lld y_ld=%lld", name,
                  (long long)M, Kd, N, (long long)x_ld, (long long)w_ld, (long long)y_ld);
    DIR_CHECK_ARG(act == DIR_ACT_NONE || act == DIR_ACT_RELU, "%s: act=%d", name, act);
    if (M == 0) return DIR_OK;
    DIR_CHECK_ARG(X && Wt && Y, "%s: null pointer", name);
    if ((Kd & 3) || (x_ld & 3) || (w_ld & 3) || !aligned16(X) || !aligned16(Wt))
        return fail(DIR_E_UNSUPPORTED, "%s: Kd and x_ld must be multiples of 4 and X / Wt 16-byte aligned (Kd=%d x_ld=%lld)", name, Kd,
                    (long long)x_ld);
    const int64_t mb = (M + DN_MT - 1) / DN_MT;
    if (mb * ((N + 79) / 80) > 0x7fffffffLL) return fail(DIR_E_UNSUPPORTED, "%s: M=%lld too large", name, (long long)M);
    hipStream_t st = as_stream(stream);
    // k chunk 16: 33 KB (NT = 80) / 41 KB (NT = 128) of LDS per workgroup, so four / three workgroups share a CU and hide each other's
    // per-chunk latencies: 0.80 vs 0.72 of the fp32 MFMA peak at 400-wide layers, 0.89 vs 0.81 at 1024 x 1024 (32-wide chunks: two
    // workgroups per CU).  DIR_DENSE_KC = 32 selects the wide chunks (tools/dense_sweep.py).
    static const int kc_env = getenv("DIR_DENSE_KC") ? atoi(getenv("DIR_DENSE_KC")) : 16;
    if (N % 80 == 0 && N % 128 != 0) {
        if (kc_env == 32) launch_dense<80, 32>(st, X, x_ld, Wt, w_ld, bias, act, M, Kd, N, Y, y_ld, gate, gate_ld, ps, psh);
        else launch_dense<80, 16>(st, X, x_ld, Wt, w_ld, bias, act, M, Kd, N, Y, y_ld, gate, gate_ld, ps, psh);
    } else {
        if (kc_env == 32) launch_dense<128, 32>(st, X, x_ld, Wt, w_ld, bias, act, M, Kd, N, Y, y_ld, gate, gate_ld, ps, psh);
        else launch_dense<128, 16>(st, X, x_ld, Wt, w_ld, bias, act, M, Kd, N, Y, y_ld, gate, gate_ld, ps, psh);
    }
    DIR_CHECK_LAUNCH(name);
    return DIR_OK;
}

extern "C" int dir_dense_f32(const float* X, int64_t x_ld, const float* Wt, int64_t w_ld, const float* bias, int act, int64_t M, int Kd, int N,
                             float* Y, int64_t y_ld, dir_stream_t stream) {
    return dense_entry("dir_dense_f32", X, x_ld, Wt, w_ld, bias, act, M, Kd, N, Y, y_ld, nullptr, 0, stream);
}

extern "C" int dir_dense_gated_f32(const float* X, int64_t x_ld, const float* Wt, int64_t w_ld, const float* gate, int64_t gate_ld, int64_t M,
                                   int Kd, int N, float* Y, int64_t y_ld, dir_stream_t stream) {
    DIR_CHECK_ARG(M == 0 || (gate && gate_ld >= N), "dir_dense_gated_f32: gate [M, N] with gate_ld >= N (gate_ld=%lld N=%d)", (long long)gate_ld, N);
    return dense_entry("dir_dense_gated_f32", X, x_ld, Wt, w_ld, nullptr, DIR_ACT_NONE, M, Kd, N, Y, y_ld, gate, gate_ld, stream);
}

extern "C" int dir_dense_affine_f32(const float* X, int64_t x_ld, const float* Wt, int64_t w_ld, const float* bias, int act, const float* post_scale,
                                    const float* post_shift, int64_t M, int Kd, int N, float* Y, int64_t y_ld, dir_stream_t stream) {
    DIR_CHECK_ARG((post_scale == nullptr) == (post_shift == nullptr), "dir_dense_affine_f32: post_scale and post_shift come together");
    return dense_entry("dir_dense_affine_f32", X, x_ld, Wt, w_ld, bias, act, M, Kd, N, Y, y_ld, nullptr, 0, stream, post_scale, post_shift);
}
