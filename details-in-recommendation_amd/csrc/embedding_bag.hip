// embedding_bag.hip -- multi-slot sparse embedding gather (+ fused FM second-order) for gfx950.
//
// Replaces (reference, /root/reference):
//   myself_input_layer                                   models/DeepFM/deepFM.py:363-400
//   tf.feature_column.input_layer                        models/DeepCrossNetwork/DeepCrossNetwork.py:126
//   fm_logit_fn (when fused)                             models/DeepFM/deepFM.py:321-335
//   [TF-upstream] safe_embedding_lookup_sparse / embedding_lookup_sparse bag semantics
//
// Work decomposition (HBM-bound, no reuse, so no LDS): LPS = K/4 lanes own one sample; lane c of
// the group owns the 16-byte chunk c of every row of that sample.  A wave therefore covers 64/LPS
// samples, walks the F slots, and per slot issues ONE global_load_dwordx4 per lane = 64/LPS rows of
// 16*LPS bytes.  UF slots are issued back to back before the first use, so each wave keeps
// UF * 1 KiB of row reads in flight.  The FM accumulators (sum_f e, sum_f e^2) are per-lane float4
// registers; the final sum over k runs lane c -> c+1 inside the group so that the result is the
// f-ascending / k-ascending fp32 sum, bit for bit what oracle/dir_oracle.c computes.
// The concat output [B, F*K] is written with the same lane mapping (16 B per lane, 64-B..1-KiB
// contiguous per sample).
#include "common.hpp"

namespace dir {

template <int VEC> struct VecT;
template <> struct VecT<4> { using T = float4; };
template <> struct VecT<1> { using T = float; };

__device__ __forceinline__ float4 ldv(const float* p, float4*) { return *reinterpret_cast<const float4*>(p); }
__device__ __forceinline__ float ldv(const float* p, float*) { return *p; }
// non-temporal (streaming) row loads: a table row is read once per launch, so it should not displace
// the output / id lines in L2 and the Infinity Cache
typedef float f32x4_t __attribute__((ext_vector_type(4)));
__device__ __forceinline__ float4 ldv_nt(const float* p, float4*) {
    f32x4_t v = __builtin_nontemporal_load(reinterpret_cast<const f32x4_t*>(p));
    return make_float4(v.x, v.y, v.z, v.w);
}
__device__ __forceinline__ float ldv_nt(const float* p, float*) { return __builtin_nontemporal_load(p); }
__device__ __forceinline__ void stv(float* p, float4 v) { *reinterpret_cast<float4*>(p) = v; }
__device__ __forceinline__ void stv(float* p, float v) { *p = v; }
__device__ __forceinline__ float4 vzero(float4*) { return make_float4(0.f, 0.f, 0.f, 0.f); }
__device__ __forceinline__ float vzero(float*) { return 0.f; }
__device__ __forceinline__ float4 vadd(float4 a, float4 b) { return make_float4(a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w); }
__device__ __forceinline__ float vadd(float a, float b) { return a + b; }
__device__ __forceinline__ float4 vmul(float4 a, float4 b) { return make_float4(a.x * b.x, a.y * b.y, a.z * b.z, a.w * b.w); }
__device__ __forceinline__ float vmul(float a, float b) { return a * b; }
__device__ __forceinline__ float4 vscale(float4 a, float s) { return make_float4(a.x * s, a.y * s, a.z * s, a.w * s); }
__device__ __forceinline__ float vscale(float a, float s) { return a * s; }
__device__ __forceinline__ float4 vdiv(float4 a, float s) { return make_float4(a.x / s, a.y / s, a.z / s, a.w / s); }
__device__ __forceinline__ float vdiv(float a, float s) { return a / s; }
// ordered horizontal add: acc + v.x + v.y + v.z + v.w, left to right
__device__ __forceinline__ float hadd_into(float acc, float4 v) { return (((acc + v.x) + v.y) + v.z) + v.w; }
__device__ __forceinline__ float hadd_into(float acc, float v) { return acc + v; }

// Exclusive upper bound of the ids of slot f as an unsigned value: ids are looked up iff (uint64_t)id < bound, which prunes
// id < 0 always and id >= vocab_f when the caller passed the vocabulary sizes (DEVICE [F]; NULL = precondition unchecked).
__device__ __forceinline__ uint64_t id_bound(const int64_t* __restrict__ vocab, int f) {
    return vocab ? (uint64_t)vocab[f] : (uint64_t)1 << 63;
}

// FM tail shared by the fused gather and the standalone kernel.  sum/sq hold this lane's chunk of
// sum_f e and sum_f e^2; returns 0.5 * sum_k (sum^2 - sq) in lane LPS-1 of the group.
template <int LPS, typename V>
__device__ __forceinline__ float fm_tail(V sum, V sq, int lane, int c) {
    V sm = vmul(sum, sum);
    V d;
    if constexpr (sizeof(V) == 16) {
        d = make_float4(sm.x - sq.x, sm.y - sq.y, sm.z - sq.z, sm.w - sq.w);
    } else {
        d = sm - sq;
    }
    float acc = 0.f;
    const int gbase = lane & ~(LPS - 1);
#pragma unroll
    for (int cc = 0; cc < LPS; ++cc) {
        float carry = __shfl(acc, gbase + (cc > 0 ? cc - 1 : 0), 64);
        if (c == cc) acc = hadd_into(cc == 0 ? 0.f : carry, d);
    }
    return 0.5f * acc;
}

// ------------------------------------------------------------------------------------------------
// one-hot gather, optional concat write, optional fused FM
// ------------------------------------------------------------------------------------------------
template <int LPS, int VEC, int KT, int UF, bool DO_FM, bool DO_OUT, bool NT>
__global__ __launch_bounds__(256) void gather_onehot_k(const float* const* __restrict__ tables,
                                                       const int64_t* __restrict__ vocab,
                                                       const int64_t* __restrict__ ids, int64_t sb,
                                                       int64_t sf, int F, int Krt, int64_t B,
                                                       float* __restrict__ out, int64_t out_ld,
                                                       float* __restrict__ fm) {
    using V = typename VecT<VEC>::T;
    constexpr int SPW = 64 / LPS;
    const int K = KT > 0 ? KT : Krt;
    const int lane = threadIdx.x & 63;
    const int c = lane & (LPS - 1);
    const int s = lane / LPS;
    const int kv = (K + VEC - 1) / VEC;
    const bool cact = c < kv;
    const int64_t nwave = (int64_t)gridDim.x * (blockDim.x >> 6);
    for (int64_t g = (int64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6); g * SPW < B; g += nwave) {
        const int64_t b = g * SPW + s;
        const bool act = cact && (b < B);
        const int64_t* idp = ids + (act ? b * sb : 0);
        float* op = DO_OUT ? out + (act ? b * out_ld : 0) + c * VEC : nullptr;
        V sum = vzero((V*)nullptr), sq = vzero((V*)nullptr);
        for (int f0 = 0; f0 < F; f0 += UF) {
            int64_t id[UF];
#pragma unroll
            for (int u = 0; u < UF; ++u) {
                const int f = f0 + u;
                id[u] = (act && f < F) ? idp[(int64_t)f * sf] : (int64_t)-1;
            }
            V row[UF];
#pragma unroll
            for (int u = 0; u < UF; ++u) {
                const int f = f0 + u;
                row[u] = vzero((V*)nullptr);
                if (f < F) {
                    const float* t = tables[f];
                    // id < 0 is pruned; with a vocab array id >= vocab_f is pruned too (one unsigned compare, scalar bound)
                    if ((uint64_t)id[u] < id_bound(vocab, f)) row[u] = NT ? ldv_nt(t + id[u] * K + c * VEC, (V*)nullptr) : ldv(t + id[u] * K + c * VEC, (V*)nullptr);
                }
            }
#pragma unroll
            for (int u = 0; u < UF; ++u) {
                const int f = f0 + u;
                if (f < F) {
                    if (DO_OUT && act) stv(op + (int64_t)f * K, row[u]);
                    if (DO_FM) {
                        sum = vadd(sum, row[u]);
                        sq = vadd(sq, vmul(row[u], row[u]));
                    }
                }
            }
        }
        if (DO_FM) {
            float r = fm_tail<LPS>(sum, sq, lane, c);
            if (c == LPS - 1 && b < B) fm[b] = r;
        }
    }
}

// ------------------------------------------------------------------------------------------------
// Packed rows (serving layout): slot f's table is [vocab_f, ld] with the K embedding floats at columns [0, K) and the
// first-order (linear_model) weight at column lin_col.  With ld*4 = 128 B and 128-byte aligned tables one row is one
// memory line: the gather moves the same DRAM bytes as the 64-byte-row layout (a 64-byte request costs a 128-byte
// slot) and the linear term of DeepFM (deepFM.py:255-275) rides along for free.
// Same lane mapping and the same ordered sums as gather_onehot_k: emb / fm / lin are bit-identical to the separate
// gather_fm + linear_sparse_sum path.
// ------------------------------------------------------------------------------------------------
template <int LPS, int UF, bool NT>
__global__ __launch_bounds__(256) void gather_packed_rows_k(const float* const* __restrict__ tables,
                                                            const int64_t* __restrict__ vocab,
                                                            const int64_t* __restrict__ ids, int64_t sb, int64_t sf,
                                                            int F, int K, int64_t ld, int lin_col, int64_t B,
                                                            float* __restrict__ out, int64_t out_ld,
                                                            float* __restrict__ fm, const float* __restrict__ bias,
                                                            float* __restrict__ lin_out,
                                                            float* __restrict__ fsum /* [B, K] field sums S[b] = sum_f e[b,f] or NULL */,
                                                            unsigned int* __restrict__ row_bits = nullptr /* [B]: bit pattern of max |out[b, :]| */,
                                                            unsigned int* __restrict__ all_bits = nullptr /* ... of max |out| (with row_bits) */,
                                                            unsigned int* __restrict__ block_bits = nullptr, unsigned int* __restrict__ ticket = nullptr) {
    constexpr int SPW = 64 / LPS;
    const int lane = threadIdx.x & 63;
    const int c = lane & (LPS - 1);
    const int s = lane / LPS;
    const int kv = K >> 2;
    const bool cact = c < kv;
    const bool want_lin = lin_out != nullptr && lin_col >= 0;
    // Line mode (round 5): the launcher gives a sample enough lanes to cover the first-order column too (K = 16, lin_col = 16: 8 lanes of
    // 16 bytes = the whole 128-byte row), so the weight arrives with the row's ONE request; lane cl holds it and keeps the running sum.
    // (Four lanes + a separate 4-byte load by lane 0 were two requests per row: the second one hits the line in L2 but costs its own slot
    // on the way to the CU -- 76.8 us against the plain gather's 61.)
    const int cl = want_lin ? (lin_col >> 2) : -1;
    const bool line = cl >= kv && cl < LPS;                 // (kernel-uniform)
    const int lw_lane = line ? cl : 0;
    const bool lact = line ? (c * 4 + 4 <= ld) : cact;      // lanes that load a piece of the row
    const int64_t nwave = (int64_t)gridDim.x * (blockDim.x >> 6);
    float wmx = 0.f;
    for (int64_t g = (int64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6); g * SPW < B; g += nwave) {
        const int64_t b = g * SPW + s;
        const bool act = cact && (b < B);
        const bool ldact = lact && (b < B);
        const int64_t* idp = ids + (ldact ? b * sb : 0);
        float* op = out ? out + (act ? b * out_ld : 0) + c * 4 : nullptr;
        float4 sum = make_float4(0.f, 0.f, 0.f, 0.f), sq = sum;
        float lin = 0.f;
        float4 mx4 = make_float4(0.f, 0.f, 0.f, 0.f);
        for (int f0 = 0; f0 < F; f0 += UF) {
            int64_t id[UF];
#pragma unroll
            for (int u = 0; u < UF; ++u) {
                const int f = f0 + u;
                id[u] = (ldact && f < F) ? idp[(int64_t)f * sf] : (int64_t)-1;
            }
            float4 row[UF];
            float lw[UF];
#pragma unroll
            for (int u = 0; u < UF; ++u) {
                const int f = f0 + u;
                row[u] = make_float4(0.f, 0.f, 0.f, 0.f);
                lw[u] = 0.f;
                if (f < F && (uint64_t)id[u] < id_bound(vocab, f)) {
                    const float* t = tables[f] + id[u] * ld;
                    row[u] = NT ? ldv_nt(t + c * 4, (float4*)nullptr) : ldv(t + c * 4, (float4*)nullptr);
                    if (want_lin && !line && c == 0) lw[u] = NT ? ldv_nt(t + lin_col, (float*)nullptr) : t[lin_col];
                }
            }
            if (line) {                                     // the weight is element lin_col & 3 of lane cl's piece; lanes past the embedding add nothing
#pragma unroll
                for (int u = 0; u < UF; ++u) {
                    const int e = lin_col & 3;
                    lw[u] = c == cl ? (e == 0 ? row[u].x : e == 1 ? row[u].y : e == 2 ? row[u].z : row[u].w) : 0.f;
                    if (!cact) row[u] = make_float4(0.f, 0.f, 0.f, 0.f);
                }
            }
#pragma unroll
            for (int u = 0; u < UF; ++u) {
                const int f = f0 + u;
                if (f < F) {
                    if (op && act) stv(op + (int64_t)f * K, row[u]);
                    sum = vadd(sum, row[u]);
                    sq = vadd(sq, vmul(row[u], row[u]));
                    lin = lin + lw[u];                    // slot order, as linear_onehot_k / the oracle
                    if (row_bits)                         // (uniform) rows of lanes past the embedding are zero
                        mx4 = make_float4(fmaxf(mx4.x, fabsf(row[u].x)), fmaxf(mx4.y, fabsf(row[u].y)), fmaxf(mx4.z, fabsf(row[u].z)),
                                          fmaxf(mx4.w, fabsf(row[u].w)));
                }
            }
        }
        if (row_bits) {                                      // (uniform) the sample's LPS lanes hold its whole output row
            float mx = fmaxf(fmaxf(mx4.x, mx4.y), fmaxf(mx4.z, mx4.w));
#pragma unroll
            for (int o = LPS >> 1; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o, 64));
            if (c == 0 && b < B) row_bits[b] = __builtin_bit_cast(unsigned int, mx);
            wmx = fmaxf(wmx, (b < B) ? mx : 0.f);
        }
        if (fsum && act) stv(fsum + b * K + c * 4, sum);     // f-ascending fp32 sums: the S of the FM backward, bit for bit
        if (fm) {
            const float r = fm_tail<LPS>(sum, sq, lane, c);
            if (c == LPS - 1 && b < B) fm[b] = r;
        }
        if (want_lin && c == lw_lane && b < B) lin_out[b] = lin + (bias ? bias[0] : 0.f);
    }
    if (all_bits) grid_max_bits(wmx, all_bits, block_bits, ticket);      // (uniform; every workgroup of the grid gets here)
}

// ------------------------------------------------------------------------------------------------
// multi-hot (CSR) bags: entries reduced in order, then combiner
// ------------------------------------------------------------------------------------------------
// [TF-upstream] embedding_lookup(..., max_norm): every looked-up row is clipped to l2-norm max_norm BEFORE it is weighted
// (clip_ops.clip_by_norm, r1.10+ form):  row * max_norm / max(||row||, max_norm),  ||row|| = sqrt(sum_k row_k^2) (0 when
// the sum is 0).  The sum of squares runs k-ascending through the LPS lanes of the group (lane c -> c+1), so it is the
// oracle's sequential fp32 sum bit for bit.
template <int LPS, typename V>
__device__ __forceinline__ V clip_row(V row, float max_norm, int lane, int c) {
    V sq = vmul(row, row);
    float acc = 0.f;
    const int gbase = lane & ~(LPS - 1);
#pragma unroll
    for (int cc = 0; cc < LPS; ++cc) {
        float carry = __shfl(acc, gbase + (cc > 0 ? cc - 1 : 0), 64);
        if (c == cc) acc = hadd_into(cc == 0 ? 0.f : carry, sq);
    }
    const float l2sum = __shfl(acc, gbase + LPS - 1, 64);
    const float l2norm = l2sum > 0.f ? sqrtf(l2sum) : l2sum;
    return vdiv(vscale(row, max_norm), fmaxf(l2norm, max_norm));
}

template <int LPS, int VEC, bool CLIP>
__global__ __launch_bounds__(256) void bag_csr_k(const float* const* __restrict__ tables,
                                                 const int64_t* __restrict__ vocab,
                                                 const int64_t* __restrict__ ids,
                                                 const int64_t* __restrict__ offsets /* nullptr: one entry per bag */,
                                                 const float* __restrict__ weights, int64_t sb, int64_t sf,
                                                 int F, int K, int64_t B, const int32_t* __restrict__ slot_combiner,
                                                 int combiner, const float* __restrict__ slot_max_norm, float max_norm, int flags,
                                                 float* __restrict__ out, int64_t out_ld) {
    // A bag needs three dependent global reads (offsets -> ids/weights -> rows).  The loop over the fields of a sample is
    // software-pipelined so that only the row reads are on the critical path: at field f the offsets of field f+2 and the
    // first U ids/weights of field f+1 are issued before the rows of field f are summed.  Entries are still reduced in bag
    // order (bit-exact against the oracle); bags longer than U continue with the plain loop.
    using V = typename VecT<VEC>::T;
    constexpr int SPW = 64 / LPS;
    constexpr int U = 8;
    const int lane = threadIdx.x & 63;
    const int c = lane & (LPS - 1);
    const int s = lane / LPS;
    const int kv = (K + VEC - 1) / VEC;
    const bool cact = c < kv;
    const bool prune_w = (flags & DIR_BAG_PRUNE_NONPOSITIVE_WEIGHTS) != 0;
    const bool nt = (flags & DIR_GATHER_STREAM_ROWS) != 0;      // tables far beyond the Infinity Cache: rows bypass the caches
    const int64_t nwave = (int64_t)gridDim.x * (blockDim.x >> 6);
    for (int64_t g = (int64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6); g * SPW < B; g += nwave) {
        const int64_t b = g * SPW + s;
        const bool sact = b < B;              // every lane of a sample's group walks the same bag (clip_row shuffles inside it)
        const bool act = cact && sact;
        auto load_off = [&](int f, int64_t& beg, int64_t& end) {
            beg = 0;
            end = 0;
            if (sact && f < F) {
                const int64_t bag = b * sb + (int64_t)f * sf;
                if (offsets) {
                    beg = offsets[bag];
                    end = offsets[bag + 1];
                } else {          // one-hot ids: the bag index addresses its single entry
                    beg = bag;
                    end = bag + 1;
                }
            }
        };
        auto load_ent = [&](int64_t e0, int64_t end, int f, int64_t (&id)[U], float (&w)[U]) {
            const uint64_t bound = f < F ? id_bound(vocab, f) : 0;
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int64_t e = e0 + u;
                id[u] = e < end ? ids[e] : (int64_t)-1;
                w[u] = (weights && e < end) ? weights[e] : 1.0f;
                if (!((uint64_t)id[u] < bound)) id[u] = -1;                 // id < 0 and id >= vocab_f are pruned
                if (weights && prune_w && !(w[u] > 0.0f)) id[u] = -1;
            }
        };
        int64_t beg0, end0, beg1, end1, beg2, end2;
        int64_t id0[U], id1[U];
        float w0[U], w1[U];
        load_off(0, beg0, end0);
        load_off(1, beg1, end1);
        load_ent(beg0, end0, 0, id0, w0);
        for (int f = 0; f < F; ++f) {
            const float* t = tables[f];
            const int comb = slot_combiner ? slot_combiner[f] : combiner;
            const float mn = slot_max_norm ? slot_max_norm[f] : max_norm;     // every embedding_column carries its own max_norm (0: none)
            load_off(f + 2, beg2, end2);
            load_ent(beg1, end1, f + 1, id1, w1);       // field f+1 (empty range when f+1 == F)
            V acc = vzero((V*)nullptr);
            float wsum = 0.f, w2sum = 0.f;
            int cnt = 0;
            for (int64_t e0 = beg0; e0 < end0; e0 += U) {
                if (e0 != beg0) load_ent(e0, end0, f, id0, w0);   // a bag longer than U
                V row[U];
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    row[u] = vzero((V*)nullptr);
                    if (id0[u] >= 0 && cact) row[u] = nt ? ldv_nt(t + id0[u] * K + c * VEC, (V*)nullptr) : ldv(t + id0[u] * K + c * VEC, (V*)nullptr);
                }
                if (CLIP && mn > 0.f) {                  // (CLIP is a template parameter: the unclipped kernel carries none of this)
#pragma unroll
                    for (int u = 0; u < U; ++u) row[u] = clip_row<LPS>(row[u], mn, lane, c);
                }
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    if (id0[u] >= 0) {
                        acc = weights ? vadd(acc, vscale(row[u], w0[u])) : vadd(acc, row[u]);
                        wsum = wsum + w0[u];
                        w2sum = w2sum + w0[u] * w0[u];
                        ++cnt;
                    }
                }
            }
            if (cnt > 0) {
                if (comb == DIR_COMBINER_MEAN) {
                    acc = vdiv(acc, weights ? wsum : (float)cnt);
                } else if (comb == DIR_COMBINER_SQRTN) {
                    acc = vdiv(acc, weights ? sqrtf(w2sum) : sqrtf((float)cnt));
                }
            }
            if (act) stv(out + b * out_ld + (int64_t)f * K + c * VEC, acc);
            beg0 = beg1; end0 = end1; beg1 = beg2; end1 = end2;
#pragma unroll
            for (int u = 0; u < U; ++u) { id0[u] = id1[u]; w0[u] = w1[u]; }
        }
    }
}

// ------------------------------------------------------------------------------------------------
// standalone FM over a materialised [B, F*K] embedding matrix
// ------------------------------------------------------------------------------------------------
template <int LPS, int VEC>
__global__ __launch_bounds__(256) void fm_k(const float* __restrict__ emb, int64_t ld, int64_t B, int F,
                                            int K, float* __restrict__ out) {
    using V = typename VecT<VEC>::T;
    constexpr int SPW = 64 / LPS;
    const int lane = threadIdx.x & 63;
    const int c = lane & (LPS - 1);
    const int s = lane / LPS;
    const int kv = (K + VEC - 1) / VEC;
    const bool cact = c < kv;
    const int64_t nwave = (int64_t)gridDim.x * (blockDim.x >> 6);
    for (int64_t g = (int64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6); g * SPW < B; g += nwave) {
        const int64_t b = g * SPW + s;
        const bool act = cact && (b < B);
        const float* ep = emb + (act ? b * ld : 0) + c * VEC;
        V sum = vzero((V*)nullptr), sq = vzero((V*)nullptr);
#pragma unroll 8
        for (int f = 0; f < F; ++f) {
            V r = act ? ldv(ep + (int64_t)f * K, (V*)nullptr) : vzero((V*)nullptr);
            sum = vadd(sum, r);
            sq = vadd(sq, vmul(r, r));
        }
        float r = fm_tail<LPS>(sum, sq, lane, c);
        if (c == LPS - 1 && b < B) out[b] = r;
    }
}

// ------------------------------------------------------------------------------------------------
// id validation (debug aid)
// ------------------------------------------------------------------------------------------------
__global__ void check_ids_k(const int64_t* __restrict__ vocab, int F, const int64_t* __restrict__ ids,
                            const int64_t* __restrict__ offsets, int64_t sb, int64_t sf, int64_t B,
                            int32_t* __restrict__ bad) {
    const int64_t n = B * F;
    int local = 0;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t b = i / F;
        const int f = (int)(i - b * F);
        const int64_t v = vocab[f];
        if (!offsets) {
            const int64_t id = ids[b * sb + f * sf];
            if (id >= v || id < -1) ++local;          // -1 is the "missing" marker; anything below it is as invalid as id >= vocab
        } else {
            const int64_t bag = b * sb + f * sf;
            for (int64_t e = offsets[bag]; e < offsets[bag + 1]; ++e)
                if (ids[e] >= v || ids[e] < -1) ++local;
        }
    }
    if (local) atomicAdd(bad, local);
}

// ---- host dispatch -----------------------------------------------------------------------------
static int next_pow2(int v) {
    int p = 1;
    while (p < v) p <<= 1;
    return p;
}

static int env_int(const char* name, int dflt) {
    const char* e = getenv(name);
    return e ? atoi(e) : dflt;
}

template <int LPS, int VEC, int KT, bool DO_FM, bool DO_OUT>
static void launch_onehot_uf(int uf, bool stream_rows, int64_t work_blocks, hipStream_t st, const float* const* tables,
                             const int64_t* vocab, const int64_t* ids, int64_t sb, int64_t sf, int F, int K, int64_t B, float* out,
                             int64_t out_ld, float* fm) {
    static const int nt_force = env_int("DIR_GATHER_NT", -1);   // development override: 0 / 1
    const bool nt_env = nt_force >= 0 ? nt_force != 0 : stream_rows;
#define DIR_GO(UF)                                                                                                  \
    do {                                                                                                            \
        if (nt_env) {                                                                                               \
            dim3 grid(grid_resident(work_blocks, resident_blocks(gather_onehot_k<LPS, VEC, KT, UF, DO_FM, DO_OUT, true>))); \
            hipLaunchKernelGGL((gather_onehot_k<LPS, VEC, KT, UF, DO_FM, DO_OUT, true>), grid, dim3(256), 0, st, tables, vocab, ids, sb, sf, F, K, B, out, out_ld, fm); \
        } else {                                                                                                    \
            dim3 grid(grid_resident(work_blocks, resident_blocks(gather_onehot_k<LPS, VEC, KT, UF, DO_FM, DO_OUT, false>))); \
            hipLaunchKernelGGL((gather_onehot_k<LPS, VEC, KT, UF, DO_FM, DO_OUT, false>), grid, dim3(256), 0, st, tables, vocab, ids, sb, sf, F, K, B, out, out_ld, fm); \
        }                                                                                                           \
    } while (0)
    switch (uf) {
        case 4: DIR_GO(4); break;
        case 13: DIR_GO(13); break;
        case 26: DIR_GO(26); break;
        default: DIR_GO(8); break;
    }
#undef DIR_GO
}

template <bool DO_FM, bool DO_OUT>
static int launch_onehot(const float* const* tables, const int64_t* vocab, int F, int K, const int64_t* ids, int64_t sb, int64_t sf,
                         int flags, int64_t B, float* out, int64_t out_ld, float* fm, hipStream_t st) {
    const bool vec = (K % 4 == 0) && (!DO_OUT || (out_ld % 4 == 0 && aligned16(out)));
    static const int uf_env = env_int("DIR_GATHER_UF", 0);
    int uf = uf_env > 0 ? uf_env : (F == 26 ? 26 : (F % 13 == 0 ? 13 : (F > 48 ? 26 : 8)));   // measured at F = 100: 26 -> 260 us, 8 -> 298 us
    const bool stream_rows = (flags & DIR_GATHER_STREAM_ROWS) != 0;
    const int lps = next_pow2(vec ? K / 4 : K);
    if (lps > 64) return fail(DIR_E_UNSUPPORTED, "embedding row of K=%d floats is wider than one wave covers (max %d)", K, vec ? 256 : 64);
    const int spw = 64 / lps;
    const int64_t waves = (B + spw - 1) / spw;
    const int64_t grid = (waves + 3) / 4;   // work blocks; the launch picks the resident count
#define DIR_CASE(L, V, KT) launch_onehot_uf<L, V, KT, DO_FM, DO_OUT>(uf, stream_rows, grid, st, tables, vocab, ids, sb, sf, F, K, B, out, out_ld, fm)
    if (vec) {
        switch (lps) {
            case 1: DIR_CASE(1, 4, 4); break;
            case 2: if (K == 8) DIR_CASE(2, 4, 8); else DIR_CASE(2, 4, 0); break;
            case 4: if (K == 16) DIR_CASE(4, 4, 16); else DIR_CASE(4, 4, 0); break;
            case 8: if (K == 32) DIR_CASE(8, 4, 32); else DIR_CASE(8, 4, 0); break;
            case 16: if (K == 64) DIR_CASE(16, 4, 64); else DIR_CASE(16, 4, 0); break;
            case 32: DIR_CASE(32, 4, 0); break;
            default: DIR_CASE(64, 4, 0); break;
        }
    } else {
        switch (lps) {
            case 1: DIR_CASE(1, 1, 1); break;
            case 2: DIR_CASE(2, 1, 0); break;
            case 4: DIR_CASE(4, 1, 0); break;
            case 8: DIR_CASE(8, 1, 0); break;
            case 16: DIR_CASE(16, 1, 0); break;
            case 32: DIR_CASE(32, 1, 0); break;
            default: DIR_CASE(64, 1, 0); break;
        }
    }
#undef DIR_CASE
    DIR_CHECK_LAUNCH("gather_onehot");
    return DIR_OK;
}

static int launch_csr(const float* const* tables, const int64_t* vocab, int F, int K, const int64_t* ids, const int64_t* offsets,
                      const float* weights, int64_t sb, int64_t sf, const int32_t* slot_combiner, int combiner, const float* slot_max_norm,
                      float max_norm, int flags, int64_t B, float* out, int64_t out_ld, hipStream_t st) {
    const bool vec = (K % 4 == 0) && (out_ld % 4 == 0) && aligned16(out);
    const int lps = next_pow2(vec ? K / 4 : K);
    if (lps > 64) return fail(DIR_E_UNSUPPORTED, "embedding row of K=%d floats is wider than one wave covers", K);
    const int spw = 64 / lps;
    const int64_t waves = (B + spw - 1) / spw;
    dim3 grid(grid_for((waves + 3) / 4));
#define DIR_CASE(L, V)                                                                                                              \
    do {                                                                                                                            \
        if (max_norm > 0.f || slot_max_norm)                                                                                        \
            hipLaunchKernelGGL((bag_csr_k<L, V, true>), grid, dim3(256), 0, st, tables, vocab, ids, offsets, weights, sb, sf, F, K, B, \
                               slot_combiner, combiner, slot_max_norm, max_norm, flags, out, out_ld);                               \
        else                                                                                                                        \
            hipLaunchKernelGGL((bag_csr_k<L, V, false>), grid, dim3(256), 0, st, tables, vocab, ids, offsets, weights, sb, sf, F, K, B, \
                               slot_combiner, combiner, slot_max_norm, max_norm, flags, out, out_ld);                               \
    } while (0)
    if (vec) {
        switch (lps) {
            case 1: DIR_CASE(1, 4); break;
            case 2: DIR_CASE(2, 4); break;
            case 4: DIR_CASE(4, 4); break;
            case 8: DIR_CASE(8, 4); break;
            case 16: DIR_CASE(16, 4); break;
            case 32: DIR_CASE(32, 4); break;
            default: DIR_CASE(64, 4); break;
        }
    } else {
        switch (lps) {
            case 1: DIR_CASE(1, 1); break;
            case 2: DIR_CASE(2, 1); break;
            case 4: DIR_CASE(4, 1); break;
            case 8: DIR_CASE(8, 1); break;
            case 16: DIR_CASE(16, 1); break;
            case 32: DIR_CASE(32, 1); break;
            default: DIR_CASE(64, 1); break;
        }
    }
#undef DIR_CASE
    DIR_CHECK_LAUNCH("bag_csr");
    return DIR_OK;
}

}  // namespace dir

using namespace dir;

extern "C" int dir_embedding_bag_ex2_f32(const float* const* tables, const int64_t* vocab, int F, int K, const int64_t* ids,
                                         const int64_t* offsets, const float* weights, int64_t stride_b, int64_t stride_f,
                                         const int32_t* slot_combiner, int combiner, const float* slot_max_norm, float max_norm, int flags,
                                         int64_t B, float* out, int64_t out_ld, dir_stream_t stream) {
    DIR_CHECK_ARG(F > 0 && K > 0 && B >= 0, "dir_embedding_bag_f32: F=%d K=%d B=%lld", F, K, (long long)B);
    if (B == 0) return DIR_OK;  // an empty batch carries no buffers
    DIR_CHECK_ARG(tables && ids && out, "dir_embedding_bag_f32: null pointer");
    DIR_CHECK_ARG(out_ld >= (int64_t)F * K, "dir_embedding_bag_f32: out_ld=%lld < F*K=%lld", (long long)out_ld, (long long)F * K);
    DIR_CHECK_ARG(combiner >= DIR_COMBINER_SUM && combiner <= DIR_COMBINER_SQRTN, "dir_embedding_bag_f32: combiner=%d", combiner);
    DIR_CHECK_ARG(offsets || !weights, "dir_embedding_bag_f32: weights need offsets (multi-hot)");
    DIR_CHECK_ARG(!(max_norm < 0.f), "dir_embedding_bag_f32: max_norm=%g", max_norm);
    if (!offsets && !(max_norm > 0.f) && !slot_max_norm)  // a one-entry bag without clipping: every combiner is the identity on it
        return launch_onehot<false, true>(tables, vocab, F, K, ids, stride_b, stride_f, flags, B, out, out_ld, nullptr, as_stream(stream));
    return launch_csr(tables, vocab, F, K, ids, offsets, weights, stride_b, stride_f, slot_combiner, combiner, slot_max_norm, max_norm, flags, B,
                      out, out_ld, as_stream(stream));
}

extern "C" int dir_embedding_bag_ex_f32(const float* const* tables, const int64_t* vocab, int F, int K, const int64_t* ids,
                                        const int64_t* offsets, const float* weights, int64_t stride_b, int64_t stride_f,
                                        const int32_t* slot_combiner, int combiner, float max_norm, int flags, int64_t B,
                                        float* out, int64_t out_ld, dir_stream_t stream) {
    return dir_embedding_bag_ex2_f32(tables, vocab, F, K, ids, offsets, weights, stride_b, stride_f, slot_combiner, combiner, nullptr, max_norm,
                                     flags, B, out, out_ld, stream);
}

extern "C" int dir_embedding_bag_f32(const float* const* tables, int F, int K, const int64_t* ids,
                                     const int64_t* offsets, const float* weights, int64_t stride_b,
                                     int64_t stride_f, int combiner, int flags, int64_t B, float* out,
                                     int64_t out_ld, dir_stream_t stream) {
    return dir_embedding_bag_ex_f32(tables, nullptr, F, K, ids, offsets, weights, stride_b, stride_f, nullptr, combiner, 0.f, flags, B,
                                    out, out_ld, stream);
}

extern "C" int dir_gather_fm_fused_f32(const float* const* tables, const int64_t* vocab, int F, int K, const int64_t* ids,
                                       int64_t stride_b, int64_t stride_f, int flags, int64_t B, float* out,
                                       int64_t out_ld, float* fm, dir_stream_t stream) {
    DIR_CHECK_ARG(F > 0 && K > 0 && B >= 0, "dir_gather_fm_fused_f32: F=%d K=%d B=%lld", F, K, (long long)B);
    if (B == 0) return DIR_OK;
    DIR_CHECK_ARG(tables && ids && (out || fm), "dir_gather_fm_fused_f32: null pointer");
    DIR_CHECK_ARG(!out || out_ld >= (int64_t)F * K, "dir_gather_fm_fused_f32: out_ld=%lld < F*K", (long long)out_ld);
    hipStream_t st = as_stream(stream);
    if (out && fm) return launch_onehot<true, true>(tables, vocab, F, K, ids, stride_b, stride_f, flags, B, out, out_ld, fm, st);
    if (fm) return launch_onehot<true, false>(tables, vocab, F, K, ids, stride_b, stride_f, flags, B, nullptr, 0, fm, st);
    return launch_onehot<false, true>(tables, vocab, F, K, ids, stride_b, stride_f, flags, B, out, out_ld, nullptr, st);
}

extern "C" int dir_fm_second_order_f32(const float* emb, int64_t emb_ld, int64_t B, int F, int K, float* out,
                                       dir_stream_t stream) {
    DIR_CHECK_ARG(emb && out, "dir_fm_second_order_f32: null pointer");
    DIR_CHECK_ARG(F > 0 && K > 0 && B >= 0 && emb_ld >= (int64_t)F * K, "dir_fm_second_order_f32: F=%d K=%d B=%lld ld=%lld", F, K, (long long)B, (long long)emb_ld);
    if (B == 0) return DIR_OK;
    const bool vec = (K % 4 == 0) && (emb_ld % 4 == 0) && aligned16(emb);
    const int lps = next_pow2(vec ? K / 4 : K);
    if (lps > 64) return fail(DIR_E_UNSUPPORTED, "dir_fm_second_order_f32: K=%d too wide", K);
    const int spw = 64 / lps;
    const int64_t waves = (B + spw - 1) / spw;
    dim3 grid(grid_for((waves + 3) / 4));
    hipStream_t st = as_stream(stream);
#define DIR_CASE(L, V) hipLaunchKernelGGL((fm_k<L, V>), grid, dim3(256), 0, st, emb, emb_ld, B, F, K, out)
    if (vec) {
        switch (lps) {
            case 1: DIR_CASE(1, 4); break;
            case 2: DIR_CASE(2, 4); break;
            case 4: DIR_CASE(4, 4); break;
            case 8: DIR_CASE(8, 4); break;
            case 16: DIR_CASE(16, 4); break;
            case 32: DIR_CASE(32, 4); break;
            default: DIR_CASE(64, 4); break;
        }
    } else {
        switch (lps) {
            case 1: DIR_CASE(1, 1); break;
            case 2: DIR_CASE(2, 1); break;
            case 4: DIR_CASE(4, 1); break;
            case 8: DIR_CASE(8, 1); break;
            case 16: DIR_CASE(16, 1); break;
            case 32: DIR_CASE(32, 1); break;
            default: DIR_CASE(64, 1); break;
        }
    }
#undef DIR_CASE
    DIR_CHECK_LAUNCH("fm_second_order");
    return DIR_OK;
}

extern "C" int dir_check_ids(const int64_t* vocab, int F, const int64_t* ids, const int64_t* offsets,
                             int64_t stride_b, int64_t stride_f, int64_t B, int32_t* bad_count,
                             dir_stream_t stream) {
    DIR_CHECK_ARG(vocab && ids && bad_count && F > 0 && B >= 0, "dir_check_ids: bad argument");
    hipStream_t st = as_stream(stream);
    if (zero_async(bad_count, sizeof(int32_t), st) != hipSuccess) return fail(DIR_E_HIP, "dir_check_ids: memset failed");
    if (B == 0) return DIR_OK;
    dim3 grid(grid_for((B * F + 255) / 256));
    hipLaunchKernelGGL(check_ids_k, grid, dim3(256), 0, st, vocab, F, ids, offsets, stride_b, stride_f, B, bad_count);
    DIR_CHECK_LAUNCH("check_ids");
    return DIR_OK;
}

static int gather_packed_launch(const float* const* tables, const int64_t* vocab, int F, int K, int64_t ld, int lin_col,
                                const int64_t* ids, int64_t stride_b, int64_t stride_f, int flags,
                                int64_t B, float* out, int64_t out_ld, float* fm, const float* bias,
                                float* lin_out, float* fsum, dir_stream_t stream, unsigned int* row_bits = nullptr,
                                unsigned int* all_bits = nullptr, unsigned int* bits_ws = nullptr) {
    DIR_CHECK_ARG(F > 0 && K > 0 && B >= 0 && ld >= K && lin_col < ld, "dir_gather_fm_linear_packed_f32: F=%d K=%d ld=%lld lin_col=%d", F, K, (long long)ld, lin_col);
    if (B == 0) return DIR_OK;
    DIR_CHECK_ARG(tables && ids && (out || fm || lin_out || fsum), "dir_gather_fm_linear_packed_f32: null pointer");
    DIR_CHECK_ARG(!out || out_ld >= (int64_t)F * K, "dir_gather_fm_linear_packed_f32: out_ld");
    if ((K & 3) || (ld & 3) || (out && ((out_ld & 3) || !aligned16(out))))
        return fail(DIR_E_UNSUPPORTED, "dir_gather_fm_linear_packed_f32: K, ld, out_ld must be multiples of 4 and out 16-byte aligned");
    int lps = next_pow2(K / 4);
    if (lps > 16) return fail(DIR_E_UNSUPPORTED, "dir_gather_fm_linear_packed_f32: K=%d (supported: up to 64)", K);
    // line mode (see the kernel): twice the lanes per sample when that brings the first-order column into the row's own request
    if (lin_out && lin_col >= K && next_pow2(lin_col / 4 + 1) == 2 * lps && 2 * lps <= 16) lps *= 2;
    const int spw = 64 / lps;
    const int64_t waves = (B + spw - 1) / spw;
    const int64_t work = (waves + 3) / 4;
    const bool nt = (flags & DIR_GATHER_STREAM_ROWS) != 0;
    hipStream_t st = as_stream(stream);
#define DIR_GO(L, NTV)                                                                                          \
    do {                                                                                                        \
        dim3 grid(grid_resident(work, resident_blocks(gather_packed_rows_k<L, 13, NTV>)));                     \
        hipLaunchKernelGGL((gather_packed_rows_k<L, 13, NTV>), grid, dim3(256), 0, st, tables, vocab, ids, stride_b, stride_f, F, K, ld, \
                           lin_col, B, out, out_ld, fm, bias, lin_out, fsum, row_bits, all_bits, bits_ws ? bits_ws + 1 : nullptr, bits_ws); \
    } while (0)
#define DIR_L(L) do { if (nt) DIR_GO(L, true); else DIR_GO(L, false); } while (0)
    switch (lps) {
        case 1: DIR_L(1); break;
        case 2: DIR_L(2); break;
        case 4: DIR_L(4); break;
        case 8: DIR_L(8); break;
        default: DIR_L(16); break;
    }
#undef DIR_L
#undef DIR_GO
    DIR_CHECK_LAUNCH("gather_fm_linear_packed");
    return DIR_OK;
}

extern "C" int dir_gather_fm_linear_packed_f32(const float* const* tables, const int64_t* vocab, int F, int K, int64_t ld, int lin_col,
                                               const int64_t* ids, int64_t stride_b, int64_t stride_f, int flags,
                                               int64_t B, float* out, int64_t out_ld, float* fm, const float* bias,
                                               float* lin_out, dir_stream_t stream) {
    return gather_packed_launch(tables, vocab, F, K, ld, lin_col, ids, stride_b, stride_f, flags, B, out, out_ld, fm, bias, lin_out, nullptr, stream);
}

// dir_gather_fm_rows_f32 that also leaves what the row-scaled fp16 x 2 layer reading `out` needs (dir_row_absmax_bits_f32's outputs, without
// its pass over out): row_bits[b] = bit pattern of max_k |out[b, k]|, *all_bits = of max |out|; workspace as dir_row_absmax_bits_f32's
// (dir_row_absmax_workspace_words() words of device memory, the first one zero before the first call).
extern "C" int dir_gather_fm_rows_bits_f32(const float* const* tables, const int64_t* vocab, int F, int K, int64_t ld, const int64_t* ids,
                                           int64_t stride_b, int64_t stride_f, int flags, int64_t B, float* out, int64_t out_ld,
                                           float* fm, float* fsum, unsigned int* row_bits, unsigned int* all_bits, unsigned int* workspace,
                                           dir_stream_t stream) {
    DIR_CHECK_ARG(row_bits && all_bits && workspace, "dir_gather_fm_rows_bits_f32: null pointer");
    if (B == 0) {
        if (zero_async(all_bits, sizeof(unsigned int), as_stream(stream)) != hipSuccess) return fail(DIR_E_HIP, "dir_gather_fm_rows_bits_f32: zeroing failed");
        return DIR_OK;
    }
    return gather_packed_launch(tables, vocab, F, K, ld, -1, ids, stride_b, stride_f, flags, B, out, out_ld, fm, nullptr, nullptr, fsum, stream,
                                row_bits, all_bits, workspace);
}

extern "C" int dir_gather_fm_rows_f32(const float* const* tables, const int64_t* vocab, int F, int K, int64_t ld, const int64_t* ids,
                                      int64_t stride_b, int64_t stride_f, int flags, int64_t B, float* out, int64_t out_ld,
                                      float* fm, float* fsum, dir_stream_t stream) {
    return gather_packed_launch(tables, vocab, F, K, ld, -1, ids, stride_b, stride_f, flags, B, out, out_ld, fm, nullptr, nullptr, fsum, stream);
}
