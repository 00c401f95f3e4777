// ids.hip -- integer id paths: hashing, bucketising, 'div' shard routing (bit-exact integer work).
//
// Replaces (reference, /root/reference):
//   categorical_column_with_hash_bucket('occupation', 1000)   models/DeepCrossNetwork/train.py:85-86
//     [TF-upstream] string_to_hash_bucket_fast = FarmHash Fingerprint64(bytes) mod buckets
//   bucketized numeric columns (docstring)                    models/DeepFM/deepFM.py:95
//   min_max_variable_partitioner + 'div' lookups              models/DeepFM/deepFM.py:163-167
//
// Fingerprint64 is farmhashna::Hash64 of FarmHash 1.1, restated from the public algorithm; checked
// against published known answers in tests/ (lengths <= 16; longer branches have no published vector).
#include "common.hpp"
#include <type_traits>

namespace dir {
namespace fh {

constexpr uint64_t k0 = 0xc3a5c85c97cb3127ULL;
constexpr uint64_t k1 = 0xb492b66fbe98f273ULL;
constexpr uint64_t k2 = 0x9ae16a3b2f90404fULL;

__host__ __device__ inline uint64_t fetch64(const char* p) {
    uint64_t v = 0;
    for (int i = 0; i < 8; ++i) v |= (uint64_t)(uint8_t)p[i] << (8 * i);
    return v;
}
__host__ __device__ inline uint64_t fetch32(const char* p) {
    uint64_t v = 0;
    for (int i = 0; i < 4; ++i) v |= (uint64_t)(uint8_t)p[i] << (8 * i);
    return v;
}
__host__ __device__ inline uint64_t rot(uint64_t v, int s) { return s == 0 ? v : ((v >> s) | (v << (64 - s))); }
__host__ __device__ inline uint64_t smix(uint64_t v) { return v ^ (v >> 47); }
__host__ __device__ inline uint64_t hash16(uint64_t u, uint64_t v, uint64_t mul) {
    uint64_t a = (u ^ v) * mul;
    a ^= (a >> 47);
    uint64_t b = (v ^ a) * mul;
    b ^= (b >> 47);
    b *= mul;
    return b;
}

__host__ __device__ inline uint64_t len0to16(const char* s, size_t len) {
    if (len >= 8) {
        uint64_t mul = k2 + len * 2;
        uint64_t a = fetch64(s) + k2;
        uint64_t b = fetch64(s + len - 8);
        uint64_t c = rot(b, 37) * mul + a;
        uint64_t d = (rot(a, 25) + b) * mul;
        return hash16(c, d, mul);
    }
    if (len >= 4) {
        uint64_t mul = k2 + len * 2;
        uint64_t a = fetch32(s);
        return hash16(len + (a << 3), fetch32(s + len - 4), mul);
    }
    if (len > 0) {
        uint8_t a = (uint8_t)s[0];
        uint8_t b = (uint8_t)s[len >> 1];
        uint8_t c = (uint8_t)s[len - 1];
        uint32_t y = (uint32_t)a + ((uint32_t)b << 8);
        uint32_t z = (uint32_t)len + ((uint32_t)c << 2);
        return smix(y * k2 ^ z * k0) * k2;
    }
    return k2;
}

__host__ __device__ inline uint64_t len17to32(const char* s, size_t len) {
    uint64_t mul = k2 + len * 2;
    uint64_t a = fetch64(s) * k1;
    uint64_t b = fetch64(s + 8);
    uint64_t c = fetch64(s + len - 8) * mul;
    uint64_t d = fetch64(s + len - 16) * k2;
    return hash16(rot(a + b, 43) + rot(c, 30) + d, a + rot(b + k2, 18) + c, mul);
}

struct U128 { uint64_t first, second; };

__host__ __device__ inline U128 weak32(uint64_t w, uint64_t x, uint64_t y, uint64_t z, uint64_t a, uint64_t b) {
    a += w;
    b = rot(b + a + z, 21);
    uint64_t c = a;
    a += x;
    a += y;
    b += rot(a, 44);
    return U128{a + z, b + c};
}
__host__ __device__ inline U128 weak32(const char* s, uint64_t a, uint64_t b) {
    return weak32(fetch64(s), fetch64(s + 8), fetch64(s + 16), fetch64(s + 24), a, b);
}

__host__ __device__ inline uint64_t len33to64(const char* s, size_t len) {
    uint64_t mul = k2 + len * 2;
    uint64_t a = fetch64(s) * k2;
    uint64_t b = fetch64(s + 8);
    uint64_t c = fetch64(s + len - 8) * mul;
    uint64_t d = fetch64(s + len - 16) * k2;
    uint64_t y = rot(a + b, 43) + rot(c, 30) + d;
    uint64_t z = hash16(y, a + rot(b + k2, 18) + c, mul);
    uint64_t e = fetch64(s + 16) * mul;
    uint64_t f = fetch64(s + 24);
    uint64_t g = (y + fetch64(s + len - 32)) * mul;
    uint64_t h = (z + fetch64(s + len - 24)) * mul;
    return hash16(rot(e + f, 43) + rot(g, 30) + h, e + rot(f + a, 18) + g, mul);
}

// every length (host and device)
__host__ __device__ inline uint64_t fingerprint64(const char* s, size_t len) {
    if (len <= 16) return len0to16(s, len);
    if (len <= 32) return len17to32(s, len);
    if (len <= 64) return len33to64(s, len);
    const uint64_t seed = 81;
    uint64_t x = seed;
    uint64_t y = seed * k1 + 113;
    uint64_t z = smix(y * k2 + 113) * k2;
    U128 v{0, 0}, w{0, 0};
    x = x * k2 + fetch64(s);
    const char* end = s + ((len - 1) / 64) * 64;
    const char* last64 = end + ((len - 1) & 63) - 63;
    do {
        x = rot(x + y + v.first + fetch64(s + 8), 37) * k1;
        y = rot(y + v.second + fetch64(s + 48), 42) * k1;
        x ^= w.second;
        y += v.first + fetch64(s + 40);
        z = rot(z + w.first, 33) * k1;
        v = weak32(s, v.second * k1, x + w.first);
        w = weak32(s + 32, z + w.second, y + fetch64(s + 16));
        uint64_t t = z; z = x; x = t;
        s += 64;
    } while (s != end);
    uint64_t mul = k1 + ((z & 0xff) << 1);
    s = last64;
    w.first += ((len - 1) & 63);
    v.first += w.first;
    w.first += v.first;
    x = rot(x + y + v.first + fetch64(s + 8), 37) * mul;
    y = rot(y + v.second + fetch64(s + 48), 42) * mul;
    x ^= w.second * 9;
    y += v.first * 9 + fetch64(s + 40);
    z = rot(z + w.first, 33) * mul;
    v = weak32(s, v.second * mul, x + w.first);
    w = weak32(s + 32, z + w.second, y + fetch64(s + 16));
    uint64_t t = z; z = x; x = t;
    return hash16(hash16(v.first, w.first, mul) + smix(y) * k0 + z, hash16(v.second, w.second, mul) + x, mul);
}

}  // namespace fh

// Decimal text of an int64 ([TF-upstream] as_string / StrCat) built IN REGISTERS: the <= 20 characters live in three
// little-endian 64-bit words (byte k of the string = byte k of w[0..2]); no private-memory byte array.
struct DecStr {
    uint64_t w0, w1, w2;
    int len;
};

__device__ __forceinline__ DecStr i64_to_dec_regs(int64_t v) {
    DecStr d{0, 0, 0, 0};
    uint64_t u = v < 0 ? (uint64_t)0 - (uint64_t)v : (uint64_t)v;
    do {   // prepend one character: shift the 192-bit string up by a byte, put the new digit at byte 0
        const uint64_t q = u / 10;
        const uint64_t c = (uint64_t)'0' + (u - q * 10);
        d.w2 = (d.w2 << 8) | (d.w1 >> 56);
        d.w1 = (d.w1 << 8) | (d.w0 >> 56);
        d.w0 = (d.w0 << 8) | c;
        ++d.len;
        u = q;
    } while (u);
    if (v < 0) {
        d.w2 = (d.w2 << 8) | (d.w1 >> 56);
        d.w1 = (d.w1 << 8) | (d.w0 >> 56);
        d.w0 = (d.w0 << 8) | (uint64_t)'-';
        ++d.len;
    }
    return d;
}

// 8 (or 4) string bytes starting at byte offset `off` (0 <= off <= 16), as Fetch64 / Fetch32 read them
__device__ __forceinline__ uint64_t dec_fetch64(const DecStr& d, int off) {
    const int wi = off >> 3, sh = (off & 7) * 8;
    const uint64_t lo = wi == 0 ? d.w0 : (wi == 1 ? d.w1 : d.w2);
    const uint64_t hi = wi == 0 ? d.w1 : (wi == 1 ? d.w2 : 0);
    return sh == 0 ? lo : ((lo >> sh) | (hi << (64 - sh)));
}
__device__ __forceinline__ uint64_t dec_fetch32(const DecStr& d, int off) { return dec_fetch64(d, off) & 0xffffffffULL; }

// farmhashna::Hash64 for len <= 32 over a register-resident string (same arithmetic as fh::len0to16 / len17to32)
__device__ __forceinline__ uint64_t fingerprint64_dec(const DecStr& d) {
    using namespace fh;
    const uint64_t len = (uint64_t)d.len;
    if (d.len > 16) {
        const uint64_t mul = k2 + len * 2;
        const uint64_t a = dec_fetch64(d, 0) * k1;
        const uint64_t b = dec_fetch64(d, 8);
        const uint64_t c = dec_fetch64(d, d.len - 8) * mul;
        const uint64_t e = dec_fetch64(d, d.len - 16) * k2;
        return hash16(rot(a + b, 43) + rot(c, 30) + e, a + rot(b + k2, 18) + c, mul);
    }
    if (d.len >= 8) {
        const uint64_t mul = k2 + len * 2;
        const uint64_t a = dec_fetch64(d, 0) + k2;
        const uint64_t b = dec_fetch64(d, d.len - 8);
        const uint64_t c = rot(b, 37) * mul + a;
        const uint64_t e = (rot(a, 25) + b) * mul;
        return hash16(c, e, mul);
    }
    if (d.len >= 4) {
        const uint64_t mul = k2 + len * 2;
        const uint64_t a = dec_fetch32(d, 0);
        return hash16(len + (a << 3), dec_fetch32(d, d.len - 4), mul);
    }
    // 1..3 characters (an int64 always has at least one digit)
    const uint32_t a = (uint32_t)(d.w0 & 0xff);
    const uint32_t b = (uint32_t)((d.w0 >> (8 * (d.len >> 1))) & 0xff);
    const uint32_t c = (uint32_t)((d.w0 >> (8 * (d.len - 1))) & 0xff);
    const uint32_t y = a + (b << 8);
    const uint32_t z = (uint32_t)d.len + (c << 2);
    return smix(y * k2 ^ z * k0) * k2;
}

// keys: flattened [.., F]; buckets_f: nullptr -> one bucket count for all, else device array [F] (entry i uses
// buckets_f[i % F])
__global__ void hash_bucket_i64_k(const int64_t* __restrict__ keys, int64_t n, uint64_t buckets,
                                  const int64_t* __restrict__ buckets_f, int F, int64_t* __restrict__ out) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const DecStr d = i64_to_dec_regs(keys[i]);
        const uint64_t h = fingerprint64_dec(d);
        const uint64_t nb = buckets_f ? (uint64_t)buckets_f[(uint64_t)i % (uint32_t)F] : buckets;
        out[i] = (int64_t)(h % nb);
    }
}

// byte strings in one buffer: string i = bytes[offsets[i] .. offsets[i+1])
__global__ void hash_bucket_bytes_k(const char* __restrict__ bytes, const int64_t* __restrict__ offsets, int64_t n,
                                    uint64_t buckets, int64_t* __restrict__ out) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t b = offsets[i], e = offsets[i + 1];
        out[i] = (int64_t)(fh::fingerprint64(bytes + b, (size_t)(e - b)) % buckets);
    }
}

__global__ void bucketize_k(const float* __restrict__ x, int64_t n, const float* __restrict__ bd, int nb,
                            int64_t* __restrict__ out) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const float v = x[i];
        // upper_bound: number of boundaries <= v  (NaN compares false everywhere -> bucket 0)
        int lo = 0, hi = nb;
        while (lo < hi) {
            const int mid = (lo + hi) >> 1;
            if (bd[mid] <= v) lo = mid + 1; else hi = mid;
        }
        out[i] = lo;
    }
}

__host__ __device__ inline void div_owner(int64_t id, int64_t q, int64_t r, int64_t thr, int* owner, int64_t* local) {
    if (id < thr) {
        const int64_t o = id / (q + 1);
        *owner = (int)o;
        *local = id - o * (q + 1);
    } else {
        const int64_t o = r + (q > 0 ? (id - thr) / q : 0);
        *owner = (int)o;
        *local = id - (thr + (o - r) * q);
    }
}

__global__ void shard_route_k(const int64_t* __restrict__ ids, int64_t n, const int64_t* __restrict__ vocab, int F,
                              int P, int32_t* __restrict__ owner, int64_t* __restrict__ local) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t id = ids[i];
        if (id < 0) {
            owner[i] = (int32_t)(i % P);
            local[i] = -1;
        } else {
            const int64_t V = vocab[i % F];
            const int64_t q = V / P, r = V % P;
            int o;
            int64_t l;
            div_owner(id, q, r, r * (q + 1), &o, &l);
            owner[i] = o;
            local[i] = l;
        }
    }
}

// flat row gather: out[i, :] = tables[slot[i]][row[i], :]  (row < 0 -> zeros).  LPS lanes per row,
// 16 B per lane when K % 4 == 0.
template <int VEC>
__global__ __launch_bounds__(256) void gather_rows_k(const float* const* __restrict__ tables, int K, int lps,
                                                     const int32_t* __restrict__ slot,
                                                     const int64_t* __restrict__ row, int64_t n,
                                                     float* __restrict__ out) {
    const int kv = K / VEC;
    const int64_t tid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t nthr = (int64_t)gridDim.x * blockDim.x;
    for (int64_t q = tid; q < n * lps; q += nthr) {
        const int64_t i = q / lps;
        const int c = (int)(q - i * lps);
        if (c >= kv) continue;
        const int64_t r = row[i];
        float* o = out + i * K + c * VEC;
        if (VEC == 4) {
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (r >= 0) v = *reinterpret_cast<const float4*>(tables[slot ? slot[i] : 0] + r * K + c * 4);
            *reinterpret_cast<float4*>(o) = v;
        } else {
            float v = 0.f;
            if (r >= 0) v = tables[slot ? slot[i] : 0][r * K + c];
            *o = v;
        }
    }
}

// ------------------------------------------------------------------------------------------------
// Bucketing by owner for the sharded lookup: a counting sort over P <= 64 owners in three launches.
//   k1: per-workgroup histogram of owners (LDS atomics)            -> wg_counts[wg][P]
//   k2: one workgroup: exclusive scan over workgroups per owner    -> wg_base[wg][P], counts[P], starts[P]
//   k3: every element gets dst = starts[o] + wg_base[wg][o] + (LDS atomic rank);
//       payload[dst] = local*F + slot (or -1 for a pruned id), inv[i] = dst
// The order inside a bucket is arbitrary (atomics) but inv is its exact inverse, so the values routed back
// through inv are independent of it.
// ------------------------------------------------------------------------------------------------
constexpr int BK_EPB = 4096;   // elements per workgroup
constexpr int BK_MAXF = 256;   // fields whose 'div' constants are cached in LDS (more: read from global)

// per-field 'div' constants q = V / Pf, thr = (V % Pf) * (q + 1), r = V % Pf, staged once per workgroup.  Pf = the number of
// row slices of the table (parts[f]; P when parts == NULL) and first = the rank holding slice 0 (slice j lives on rank
// (first + j) % P): the reference's min_max_variable_partitioner cuts a table into <= P slices of >= min_slice_size bytes
// (models/DeepFM/deepFM.py:163-167) and [TF-upstream] replica_device_setter deals the slices round-robin over the ps tasks.
struct FieldDiv { int64_t q, thr, V; int r; int small; int first; };

__device__ __forceinline__ FieldDiv make_fielddiv(int64_t V, int Pf, int first = 0) {
    FieldDiv d;
    d.q = V / Pf;
    d.r = (int)(V % Pf);
    d.thr = (int64_t)d.r * (d.q + 1);
    d.V = V;
    d.small = V < (int64_t)0x7fffffff ? 1 : 0;   // every quotient fits 32-bit unsigned arithmetic
    d.first = first;
    return d;
}

__device__ __forceinline__ void route_fd(int64_t id, const FieldDiv& d, int* owner, int64_t* local) {
    if (d.small) {   // 32-bit divisions (ids < vocab < 2^31): ~4x cheaper than the 64-bit software division
        const uint32_t u = (uint32_t)id, q = (uint32_t)d.q, thr = (uint32_t)d.thr;
        if (u < thr) {
            const uint32_t o = u / (q + 1);
            *owner = (int)o;
            *local = (int64_t)(u - o * (q + 1));
        } else {
            const uint32_t o = (uint32_t)d.r + (q > 0 ? (u - thr) / q : 0u);
            *owner = (int)o;
            *local = (int64_t)(u - (thr + (o - (uint32_t)d.r) * q));
        }
    } else {
        div_owner(id, d.q, d.r, d.thr, owner, local);
    }
}

// element e of workgroup `wg`: its id, slot f = i % F and owner / local row; pruned ids spread as i % P
#define BK_ROUTE_ELEMENT()                                                        \
    const int64_t i = base + e;                                                   \
    const int f = (int)((uint32_t)(f0 + e) % (uint32_t)F);                        \
    const int64_t id = ids[i];                                                    \
    int oo_;                                                                      \
    int64_t l;                                                                    \
    if (id < 0 || id >= (f < BK_MAXF ? fd[f].V : vocab[f])) {  /* pruned, or out of range: nobody owns it */ \
        oo_ = (int)((uint32_t)(p0 + e) % (uint32_t)P);                            \
        l = -1;                                                                   \
    } else if (f < BK_MAXF) {                                                     \
        route_fd(id, fd[f], &oo_, &l);                                            \
        oo_ += fd[f].first;                                                       \
        if (oo_ >= P) oo_ -= P;                                                   \
    } else {                                                                      \
        FieldDiv dd = make_fielddiv(vocab[f], parts ? parts[f] : P, first ? first[f] : 0); \
        route_fd(id, dd, &oo_, &l);                                               \
        oo_ += dd.first;                                                          \
        if (oo_ >= P) oo_ -= P;                                                   \
    }

// Wave-aggregated LDS counter update: lanes of a wave that target the same owner issue ONE atomic (the
// lowest such lane), the others take consecutive ranks behind it.  Returns this lane's rank in cnt[o].
// `active` lanes take part; the loop runs once per distinct owner present in the wave.
__device__ __forceinline__ int wave_agg_rank(int* cnt, int o, bool active) {
    const unsigned long long lt = (1ull << (threadIdx.x & 63)) - 1ull;
    unsigned long long todo = __ballot(active);
    int rank = 0;
    while (todo) {
        const int leader = __ffsll((long long)todo) - 1;
        const int oo = __shfl(o, leader, 64);
        const unsigned long long same = __ballot(active && o == oo) & todo;
        int basev = 0;
        if ((int)(threadIdx.x & 63) == leader) basev = atomicAdd(&cnt[oo], __popcll(same));
        basev = __shfl(basev, leader, 64);
        if (active && o == oo) rank = basev + __popcll(same & lt);
        todo &= ~same;
    }
    return rank;
}

__global__ __launch_bounds__(256) void bucket_hist_k(const int64_t* __restrict__ ids, int64_t n,
                                                     const int64_t* __restrict__ vocab, const int32_t* __restrict__ parts,
                                                     const int32_t* __restrict__ first, int F, int P,
                                                     int32_t* __restrict__ wg_counts) {
    __shared__ int cnt[64];
    __shared__ FieldDiv fd[BK_MAXF];
    if (threadIdx.x < 64) cnt[threadIdx.x] = 0;
    for (int f = threadIdx.x; f < F && f < BK_MAXF; f += 256) fd[f] = make_fielddiv(vocab[f], parts ? parts[f] : P, first ? first[f] : 0);
    __syncthreads();
    const int64_t base = (int64_t)blockIdx.x * BK_EPB;
    const int f0 = (int)(base % F), p0 = (int)(base % P);
    const int lim = (int)((n - base) < BK_EPB ? (n - base) : BK_EPB);
    for (int e0 = 0; e0 < lim; e0 += 256) {   // uniform trip count: the aggregation uses wave-wide ballots
        const int e = e0 + threadIdx.x;
        const bool active = e < lim;
        int o = 0;
        if (active) {
            BK_ROUTE_ELEMENT()
            (void)l;
            (void)i;
            o = oo_;
        }
        wave_agg_rank(cnt, o, active);
    }
    __syncthreads();
    if (threadIdx.x < P) wg_counts[(int64_t)blockIdx.x * P + threadIdx.x] = cnt[threadIdx.x];
}

// one workgroup of 16 waves; wave w scans owners w, w+16, ...: 64 workgroup counts per step with a
// shuffle prefix scan and a running carry
__global__ __launch_bounds__(1024) void bucket_scan_k(const int32_t* __restrict__ wg_counts, int nwg, int P,
                                                      int32_t* __restrict__ wg_base, int64_t* __restrict__ counts,
                                                      int64_t* __restrict__ starts) {
    __shared__ int64_t tot[64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int o = wave; o < P; o += 16) {
        int run = 0;
        for (int c = 0; c < nwg; c += 64) {
            const int wg = c + lane;
            const int v = wg < nwg ? wg_counts[(int64_t)wg * P + o] : 0;
            int inc = v;
#pragma unroll
            for (int d = 1; d < 64; d <<= 1) {
                const int t = __shfl_up(inc, d, 64);
                if (lane >= d) inc += t;
            }
            if (wg < nwg) wg_base[(int64_t)wg * P + o] = run + inc - v;
            run += __shfl(inc, 63, 64);
        }
        if (lane == 0) {
            tot[o] = run;
            counts[o] = run;
        }
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        int64_t run = 0;
        for (int k = 0; k < P; ++k) {
            starts[k] = run;
            run += tot[k];
        }
    }
}

__global__ __launch_bounds__(256) void bucket_scatter_k(const int64_t* __restrict__ ids, int64_t n,
                                                        const int64_t* __restrict__ vocab, const int32_t* __restrict__ parts,
                                                        const int32_t* __restrict__ first, int F, int P,
                                                        const int32_t* __restrict__ wg_base,
                                                        const int64_t* __restrict__ starts,
                                                        int64_t* __restrict__ payload, int64_t* __restrict__ inv) {
    __shared__ int cnt[64];
    __shared__ int64_t basev[64];
    __shared__ FieldDiv fd[BK_MAXF];
    if (threadIdx.x < 64) cnt[threadIdx.x] = 0;
    if (threadIdx.x < P) basev[threadIdx.x] = starts[threadIdx.x] + wg_base[(int64_t)blockIdx.x * P + threadIdx.x];
    for (int f = threadIdx.x; f < F && f < BK_MAXF; f += 256) fd[f] = make_fielddiv(vocab[f], parts ? parts[f] : P, first ? first[f] : 0);
    __syncthreads();
    const int64_t base = (int64_t)blockIdx.x * BK_EPB;
    const int f0 = (int)(base % F), p0 = (int)(base % P);
    const int lim = (int)((n - base) < BK_EPB ? (n - base) : BK_EPB);
    for (int e0 = 0; e0 < lim; e0 += 256) {
        const int e = e0 + threadIdx.x;
        const bool active = e < lim;
        int o = 0, fsl = 0;
        int64_t ll = -1, ii = 0;
        if (active) {
            BK_ROUTE_ELEMENT()
            o = oo_;
            ll = l;
            ii = i;
            fsl = f;
        }
        const int rank = wave_agg_rank(cnt, o, active);
        if (active) {
            const int64_t dst = basev[o] + rank;
            payload[dst] = ll < 0 ? (int64_t)-1 : ll * F + fsl;
            inv[ii] = dst;
        }
    }
}

// ------------------------------------------------------------------------------------------------
// Fixed-capacity bucketing (the sync-free lookup): owner o gets a SLAB of cap payload slots behind a one-word header,
//   payload[o*(cap+1)] = number of valid slots (written by bucket_cap_fin_k), payload[o*(cap+1) + 1 + pos] = local*F + slot.
// One pass: every workgroup routes its 4096 elements keeping (owner, rank-in-workgroup, payload) in registers, reserves a
// contiguous range of every owner's slab with ONE global atomic per owner, and scatters.  inv[i] = o*cap + pos is the row of
// element i in the [P*cap, K] row buffer that comes back; pruned / out-of-range ids and elements that do not fit the slab
// get inv = -1 (the finish gather turns that into a zero row; an overflow is reported through the flag and the caller
// repeats the lookup on the exact variable-size path).  The order inside a slab is arbitrary (atomics); inv is its exact
// inverse, so the looked-up values do not depend on it.  gcount: P int32 counters, zero on entry, zeroed again by fin.
// ------------------------------------------------------------------------------------------------
// Elements per workgroup of the one-pass kernel = 256 * EPT.  Two opposing costs: few workgroups leave CUs idle (4096 elements
// per workgroup on a 16 384 x 26 micro-batch: 104 workgroups, 24 us), many workgroups queue up on the P slab counters (same-address
// atomics retire at ~90 per us: 1 664 workgroups on the whole 65 536 x 26 batch: 28 us).  The host picks EPT for ~400-800 workgroups.
typedef float f32x4_ids __attribute__((ext_vector_type(4)));

template <int BK_EPT>
__global__ __launch_bounds__(256) void bucket_cap_k(const int64_t* __restrict__ ids, int64_t n,
                                                    const int64_t* __restrict__ vocab, const int32_t* __restrict__ parts,
                                                    const int32_t* __restrict__ first, int F, int P, int64_t cap,
                                                    int32_t* __restrict__ gcount, int64_t* __restrict__ payload,
                                                    int64_t* __restrict__ inv) {
    __shared__ int cnt[64];
    __shared__ int basev[64];
    __shared__ FieldDiv fd[BK_MAXF];
    if (threadIdx.x < 64) cnt[threadIdx.x] = 0;
    for (int f = threadIdx.x; f < F && f < BK_MAXF; f += 256) fd[f] = make_fielddiv(vocab[f], parts ? parts[f] : P, first ? first[f] : 0);
    __syncthreads();
    constexpr int BKC_EPB = 256 * BK_EPT;
    const int64_t base = (int64_t)blockIdx.x * BKC_EPB;
    const int f0 = (int)(base % F), p0 = (int)(base % P);
    const int lim = (int)((n - base) < BKC_EPB ? (n - base) : BKC_EPB);
    int orank[BK_EPT];        // owner << 16 | rank inside the workgroup (rank < 4096); -1: pruned / inactive
    int64_t pv[BK_EPT];
#pragma unroll
    for (int k = 0; k < BK_EPT; ++k) {
        const int e = k * 256 + threadIdx.x;
        const bool active = e < lim;
        int o = 0;
        bool keep = false;
        pv[k] = -1;
        if (active) {
            BK_ROUTE_ELEMENT()
            (void)i;
            o = oo_;
            keep = l >= 0;
            pv[k] = l * F + f;
        }
        const int rank = wave_agg_rank(cnt, o, keep);
        orank[k] = keep ? (o << 16) | rank : -1;
    }
    __syncthreads();
    if (threadIdx.x < P) basev[threadIdx.x] = cnt[threadIdx.x] ? atomicAdd(&gcount[threadIdx.x], cnt[threadIdx.x]) : 0;
    __syncthreads();
#pragma unroll
    for (int k = 0; k < BK_EPT; ++k) {
        const int e = k * 256 + threadIdx.x;
        if (e >= lim) continue;
        int64_t dst = -1;
        if (orank[k] >= 0) {
            const int o = orank[k] >> 16;
            const int64_t pos = (int64_t)basev[o] + (orank[k] & 0xffff);
            if (pos < cap) {
                payload[(int64_t)o * (cap + 1) + 1 + pos] = pv[k];
                dst = (int64_t)o * cap + pos;
            }
        }
        inv[base + e] = dst;
    }
}

// Slab header word: low 32 bits = number of valid slots (<= cap); high 32 bits = the SENDER's largest per-owner demand in this
// micro-batch.  Every rank therefore learns every other rank's demand from the id exchange itself (slab_stat_k below): the global
// overflow verdict needs no collective of its own.
__device__ __forceinline__ int64_t slab_count(int64_t header) { return (int64_t)(uint32_t)header; }

__global__ void bucket_cap_fin_k(int32_t* __restrict__ gcount, int P, int64_t cap, int64_t* __restrict__ payload,
                                 int64_t* __restrict__ counts, int32_t* __restrict__ overflow, int64_t* __restrict__ stat) {
    const int o = threadIdx.x;
    int over = 0;
    int c32 = 0;
    int64_t c = 0;
    if (o < P) {
        c = gcount[o];
        counts[o] = c;                                    // the true demand (may exceed cap): the caller sizes the next cap from it
        gcount[o] = 0;
        over = c > cap ? 1 : 0;
        c32 = (int)(c < 0x7fffffff ? c : 0x7fffffff);
    }
    const unsigned long long any = __ballot(over);
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) c32 = max(c32, __shfl_xor(c32, d, 64));
    if (o < P) payload[(int64_t)o * (cap + 1)] = (c < cap ? c : cap) | ((int64_t)c32 << 32);
    if (o == 0) {
        overflow[0] = any ? 1 : 0;
        if (stat) {          // [overflow, largest per-owner demand]: what the caller reduces over chunks and ranks
            stat[0] = any ? 1 : 0;
            stat[1] = c32;
        }
    }
}

// The finalize step as a device function: wave 0 of the LAST workgroup of a bucket kernel runs it (bucket_cap2_k), so the slab headers,
// the true demands and the verdict come out of the same launch.  Reads the counters with device-scope atomics (they were written by
// other workgroups' atomics; nothing is cached for them) and leaves them zero.
__device__ __forceinline__ void bucket_cap_finalize(int32_t* __restrict__ gcount, int P, int64_t cap, int64_t* __restrict__ payload,
                                                    int64_t* __restrict__ counts, int32_t* __restrict__ overflow, int64_t* __restrict__ stat) {
    const int o = threadIdx.x;       // 0..63
    int over = 0, c32 = 0;
    int64_t c = 0;
    if (o < P) {
        c = atomicExch(&gcount[o], 0);
        counts[o] = c;
        over = c > cap ? 1 : 0;
        c32 = (int)(c < 0x7fffffff ? c : 0x7fffffff);
    }
    const unsigned long long any = __ballot(over);
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) c32 = max(c32, __shfl_xor(c32, d, 64));
    if (o < P) payload[(int64_t)o * (cap + 1)] = (c < cap ? c : cap) | ((int64_t)c32 << 32);
    if (o == 0) {
        overflow[0] = any ? 1 : 0;
        if (stat) {
            stat[0] = any ? 1 : 0;
            stat[1] = c32;
        }
    }
}

// bucket_cap_k, round 4: the same one-pass routing with (a) the finalize step inside (the last workgroup to arrive at a device-scope
// counter writes headers / demands / verdict: one launch, no 64-thread kernel behind it) and (b) workgroups of NT = 1024 threads.
// What bounded bucket_cap_k (24 + 4.8 us for 1.7 M ids against ~8 us of traffic) is the returning atomic every workgroup issues on
// the P slab counters: one address retires ~90 of them per us, all 832 workgroups were resident and issued theirs together, so the
// last one waited ~9 us with nothing else to do.  Four times the threads per workgroup = a quarter of the reservations (208 at the
// BASELINE size: ~2.3 us of queue) at the same number of resident waves.  (Tried first and dropped: persistent 256-thread workgroups
// with the reservation of tile t+1 in flight under the scatter of tile t -- 33-59 us: two register sets per thread cost more
// occupancy than the overlap bought; and a __threadfence() before the arrival counter -- on gfx950 a device-scope release writes
// the XCD's L2 back, 84 us.  Nothing but the counters crosses workgroups inside the launch, and those are device-scope atomics whose
// results the workgroup has consumed before it arrives: no fence is needed; slabs / inv / headers are for LATER kernels.)
template <int BK_EPT, int NT>
__global__ __launch_bounds__(NT) void bucket_cap2_k(const int64_t* __restrict__ ids, int64_t n,
                                                    const int64_t* __restrict__ vocab, const int32_t* __restrict__ parts,
                                                    const int32_t* __restrict__ first, int F, int P, int64_t cap,
                                                    int32_t* __restrict__ gcount, int64_t* __restrict__ payload,
                                                    int64_t* __restrict__ inv, int64_t* __restrict__ counts,
                                                    int32_t* __restrict__ overflow, int64_t* __restrict__ stat) {
    __shared__ int cnt[64];
    __shared__ int basev[64];
    __shared__ FieldDiv fd[BK_MAXF];
    __shared__ int s_last;
    if (threadIdx.x < 64) cnt[threadIdx.x] = 0;
    for (int f = threadIdx.x; f < F && f < BK_MAXF; f += NT) fd[f] = make_fielddiv(vocab[f], parts ? parts[f] : P, first ? first[f] : 0);
    __syncthreads();
    constexpr int BKC_EPB = NT * BK_EPT;
    static_assert(BKC_EPB <= 65536, "the rank inside a tile is kept in 16 bits");
    const int64_t base = (int64_t)blockIdx.x * BKC_EPB;
    const int f0 = (int)(base % F), p0 = (int)(base % P);
    const int lim = (int)((n - base) < BKC_EPB ? (n - base) : BKC_EPB);
    int orank[BK_EPT];        // owner << 16 | rank inside the tile; -1: pruned / inactive
    int64_t pv[BK_EPT];
#pragma unroll
    for (int k = 0; k < BK_EPT; ++k) {
        const int e = k * NT + threadIdx.x;
        const bool active = e < lim;
        int o = 0;
        bool keep = false;
        pv[k] = -1;
        if (active) {
            BK_ROUTE_ELEMENT()
            (void)i;
            o = oo_;
            keep = l >= 0;
            pv[k] = l * F + f;
        }
        const int rank = wave_agg_rank(cnt, o, keep);
        orank[k] = keep ? (o << 16) | rank : -1;
    }
    __syncthreads();
    if (threadIdx.x < P) basev[threadIdx.x] = cnt[threadIdx.x] ? atomicAdd(&gcount[threadIdx.x], cnt[threadIdx.x]) : 0;
    __syncthreads();
#pragma unroll
    for (int k = 0; k < BK_EPT; ++k) {
        const int e = k * NT + threadIdx.x;
        if (e >= lim) continue;
        int64_t dst = -1;
        if (orank[k] >= 0) {
            const int o = orank[k] >> 16;
            const int64_t pos = (int64_t)basev[o] + (orank[k] & 0xffff);
            if (pos < cap) {
                payload[(int64_t)o * (cap + 1) + 1 + pos] = pv[k];
                dst = (int64_t)o * cap + pos;
            }
        }
        inv[base + e] = dst;
    }
    // arrival: the last workgroup finalizes (see the header comment for why there is no fence)
    if (threadIdx.x == 0) s_last = atomicAdd(&gcount[64], 1) == (int)gridDim.x - 1 ? 1 : 0;
    __syncthreads();
    if (s_last && threadIdx.x < 64) {
        bucket_cap_finalize(gcount, P, cap, payload, counts, overflow, stat);
        if (threadIdx.x == 0) __hip_atomic_store(&gcount[64], 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);      // (an atomic like every other access to the word)
    }
}

// After the id exchange: n_slabs received slabs (any number of micro-batches x P senders, `cap + 1` words apart) ->
// stat = {1 iff some sender's demand exceeded cap, the largest demand}: the same two numbers on every rank.
__global__ void slab_stat_k(const int64_t* __restrict__ recv, int n_slabs, int64_t cap, int64_t* __restrict__ stat) {
    int m = 0;
    for (int i = threadIdx.x; i < n_slabs; i += 64) m = max(m, (int)(recv[(int64_t)i * (cap + 1)] >> 32));
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) m = max(m, __shfl_xor(m, d, 64));
    if (threadIdx.x == 0) {
        stat[0] = (int64_t)m > cap ? 1 : 0;
        stat[1] = m;
    }
}

// Fixed-capacity bucketing WITH duplicate removal ([TF-upstream] embedding_lookup_sparse gathers unique ids): one workgroup takes
// 256 * EPT samples of ONE slot f, so a tile holds that many draws from one table and a hot row shows up many times in it.  Every
// entry's row id goes into an LDS hash table (open addressing, 2 slots per entry); the first inserter of a
// value owns it, reserves a slab position like bucket_cap_k does (one global atomic per owner and workgroup) and publishes it in
// the table; duplicates copy the owner's position into inv.  Duplicates that fall into DIFFERENT tiles are sent once per tile:
// the lookup's result is the same, only the exchange carries a few more rows than an exact unique() would (on Zipf(1.05) ids over
// 10^6 rows a 4096-sample tile removes ~49 % of the rows, an exact unique over 65 536 samples ~63 %) for none of a sort's cost.
// ids are read with strides (sb, sf); inv is written FIELD-MAJOR, inv[f * B + b] (coalesced), and handed to the finish gather as a
// strided [B, F] view.  gcount / fin as in bucket_cap_k: the demand the headers carry is the de-duplicated one.
template <int EPT>
__global__ __launch_bounds__(256) void bucket_cap_dedup_k(const int64_t* __restrict__ ids, int64_t sb, int64_t sf, int64_t B,
                                                          const int64_t* __restrict__ vocab, const int32_t* __restrict__ parts,
                                                          const int32_t* __restrict__ first, int F, int P, int64_t cap,
                                                          int32_t* __restrict__ gcount, int64_t* __restrict__ payload,
                                                          int64_t* __restrict__ inv) {
    constexpr int TILE = 256 * EPT, HT = 2 * TILE;
    constexpr uint32_t EMPTY = 0xffffffffu;
    __shared__ uint32_t hkey[HT];
    __shared__ int32_t hpos[HT];          // slab position o * cap + pos of the value in hkey (needs P * cap < 2^31), -1: did not fit
    __shared__ int cnt[64];
    __shared__ int basev[64];
    const int tiles_per_f = (int)((B + TILE - 1) / TILE);
    const int f = (int)(blockIdx.x / tiles_per_f);
    const int64_t b0 = (int64_t)(blockIdx.x - f * tiles_per_f) * TILE;
    const FieldDiv fd = make_fielddiv(vocab[f], parts ? parts[f] : P, first ? first[f] : 0);
    for (int h = threadIdx.x; h < HT; h += 256) hkey[h] = EMPTY;
    if (threadIdx.x < 64) cnt[threadIdx.x] = 0;
    __syncthreads();
    int own[EPT];           // owner, -1: pruned / out of range / inactive
    int slot[EPT];          // hash slot (-1: row id too wide for the 32-bit table: sent without de-duplication)
    int64_t pv[EPT];
    unsigned win = 0;       // bit k: this thread owns its value
#pragma unroll
    for (int k = 0; k < EPT; ++k) {
        const int64_t b = b0 + k * 256 + threadIdx.x;
        own[k] = -1;
        slot[k] = -1;
        pv[k] = -1;
        if (b < B) {
            const int64_t id = ids[b * sb + (int64_t)f * sf];
            if (id >= 0 && id < fd.V) {
                int o;
                int64_t l;
                route_fd(id, fd, &o, &l);
                o += fd.first;
                if (o >= P) o -= P;
                own[k] = o;
                const int64_t p = l * F + f;
                pv[k] = p;
                if (id < (int64_t)EMPTY) {                  // the tile is ONE slot: the row id itself identifies the value (owner included)
                    const uint32_t p32 = (uint32_t)id;
                    uint32_t h = (p32 * 2654435761u) >> (32 - __builtin_ctz(HT));     // multiplicative hash: the top log2(HT) bits
                    for (;;) {
                        const uint32_t old = atomicCAS(&hkey[h], EMPTY, p32);
                        if (old == EMPTY) { win |= 1u << k; break; }
                        if (old == p32) break;
                        h = (h + 1) & (uint32_t)(HT - 1);
                    }
                    slot[k] = (int)h;
                } else {
                    win |= 1u << k;
                }
            }
        }
    }
    int rank[EPT];
#pragma unroll
    for (int k = 0; k < EPT; ++k) rank[k] = wave_agg_rank(cnt, own[k] < 0 ? 0 : own[k], (win >> k) & 1u);
    __syncthreads();
    if (threadIdx.x < P) basev[threadIdx.x] = cnt[threadIdx.x] ? atomicAdd(&gcount[threadIdx.x], cnt[threadIdx.x]) : 0;
    __syncthreads();
    int64_t dst[EPT];
#pragma unroll
    for (int k = 0; k < EPT; ++k) {
        dst[k] = -1;
        if ((win >> k) & 1u) {
            const int o = own[k];
            const int64_t pos = (int64_t)basev[o] + rank[k];
            if (pos < cap) {
                payload[(int64_t)o * (cap + 1) + 1 + pos] = pv[k];
                dst[k] = (int64_t)o * cap + pos;
            }
            if (slot[k] >= 0) hpos[slot[k]] = (int32_t)dst[k];
        }
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < EPT; ++k) {
        const int64_t b = b0 + k * 256 + threadIdx.x;
        if (b >= B) continue;
        int64_t d = dst[k];
        if (own[k] >= 0 && !((win >> k) & 1u)) d = hpos[slot[k]];
        inv[(int64_t)f * B + b] = d;
    }
}

// owner side of the fixed-capacity exchange: recv = P slabs [header | cap slots] as received (slab s from rank s);
// out[(s*cap + j), :] = tables[p % F][p / F, :] for j < header_s.  Slots behind the header are neither read nor written
// (the requester never looks at them); with sanitize the slot itself is overwritten with -1 so that the slab can later be
// walked as a flat pruned-aware payload (the owner side of the sharded backward).
template <int VEC, bool NT>
__global__ __launch_bounds__(256) void gather_slabs_k(const float* const* __restrict__ tables, int K, int lps, int F,
                                                      int64_t* __restrict__ recv, int P, int64_t cap, int sanitize,
                                                      float* __restrict__ out) {
    const int kv = K / VEC;
    const int64_t total = (int64_t)P * cap;
    const int64_t tid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t nthr = (int64_t)gridDim.x * blockDim.x;
    for (int64_t q = tid; q < total * lps; q += nthr) {
        const int64_t i = q / lps;
        const int c = (int)(q - i * lps);
        const int64_t sl = i / cap, j = i - sl * cap;
        int64_t* slab = recv + sl * (cap + 1);
        const int64_t valid = slab_count(slab[0]);
        if (j >= valid) {
            if (sanitize && c == 0) slab[1 + j] = -1;
            continue;
        }
        if (c >= kv) continue;
        const int64_t p = slab[1 + j];
        int slot;
        int64_t row;
        if (p < (int64_t)0x7fffffff) {   // 32-bit division when it fits
            const uint32_t r32 = (uint32_t)p / (uint32_t)F;
            slot = (int)((uint32_t)p - r32 * (uint32_t)F);
            row = r32;
        } else {
            row = p / F;
            slot = (int)(p - row * F);
        }
        float* o = out + i * K + c * VEC;
        const float* src = tables[slot] + row * K + c * VEC;
        if (VEC == 4) {
            float4 v;
            if (NT) {
                const f32x4_ids t = __builtin_nontemporal_load(reinterpret_cast<const f32x4_ids*>(src));
                v = make_float4(t.x, t.y, t.z, t.w);
            } else {
                v = *reinterpret_cast<const float4*>(src);
            }
            *reinterpret_cast<float4*>(o) = v;
        } else {
            *o = NT ? __builtin_nontemporal_load(src) : *src;
        }
    }
}
#undef BK_ROUTE_ELEMENT

// owner side: unpack payload -> (slot, row) and gather; out[i, :] = tables[p % F][p / F, :]  (p < 0 -> zeros)
template <int VEC, bool NT>
__global__ __launch_bounds__(256) void gather_packed_k(const float* const* __restrict__ tables, int K, int lps, int F,
                                                       const int64_t* __restrict__ payload, int64_t n,
                                                       float* __restrict__ out) {
    const int kv = K / VEC;
    const int64_t tid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t nthr = (int64_t)gridDim.x * blockDim.x;
    for (int64_t q = tid; q < n * lps; q += nthr) {
        const int64_t i = q / lps;
        const int c = (int)(q - i * lps);
        if (c >= kv) continue;
        const int64_t p = payload[i];
        float* o = out + i * K + c * VEC;
        int slot = 0;
        int64_t row = 0;
        if (p >= 0) {
            if (p < (int64_t)0x7fffffff) {   // 32-bit division when it fits
                const uint32_t r32 = (uint32_t)p / (uint32_t)F;
                slot = (int)((uint32_t)p - r32 * (uint32_t)F);
                row = r32;
            } else {
                row = p / F;
                slot = (int)(p - row * F);
            }
        }
        if (VEC == 4) {
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (p >= 0) {
                const float* src = tables[slot] + row * K + c * 4;
                if (NT) {
                    const f32x4_ids t = __builtin_nontemporal_load(reinterpret_cast<const f32x4_ids*>(src));
                    v = make_float4(t.x, t.y, t.z, t.w);
                } else {
                    v = *reinterpret_cast<const float4*>(src);
                }
            }
            *reinterpret_cast<float4*>(o) = v;
        } else {
            float v = 0.f;
            if (p >= 0) v = NT ? __builtin_nontemporal_load(tables[slot] + row * K + c) : tables[slot][row * K + c];
            *o = v;
        }
    }
}

}  // namespace dir

using namespace dir;

extern "C" uint64_t dir_fingerprint64(const char* s, int64_t len) {
    if (!s || len < 0) return fh::k2;
    return fh::fingerprint64(s, (size_t)len);
}

extern "C" int dir_hash_bucket_fast(const char* const* strs, const int64_t* lens, int64_t n, int64_t num_buckets,
                                    int64_t* out) {
    DIR_CHECK_ARG(strs && lens && out && n >= 0 && num_buckets > 0, "dir_hash_bucket_fast: bad argument");
    for (int64_t i = 0; i < n; ++i) {
        DIR_CHECK_ARG(strs[i] || lens[i] == 0, "dir_hash_bucket_fast: null string %lld", (long long)i);
        out[i] = (int64_t)(fh::fingerprint64(strs[i], (size_t)lens[i]) % (uint64_t)num_buckets);
    }
    return DIR_OK;
}

extern "C" int dir_hash_bucket_i64_device(const int64_t* keys, int64_t n, int64_t num_buckets, int64_t* out,
                                          dir_stream_t stream) {
    DIR_CHECK_ARG(keys && out && n >= 0 && num_buckets > 0, "dir_hash_bucket_i64_device: bad argument");
    if (n == 0) return DIR_OK;
    hipLaunchKernelGGL(hash_bucket_i64_k, dim3(grid_for((n + 255) / 256)), dim3(256), 0, as_stream(stream), keys, n,
                       (uint64_t)num_buckets, (const int64_t*)nullptr, 1, out);
    DIR_CHECK_LAUNCH("hash_bucket_i64");
    return DIR_OK;
}

extern "C" int dir_hash_bucket_i64_fields_device(const int64_t* keys, int64_t n, const int64_t* buckets_f, int F,
                                                 int64_t* out, dir_stream_t stream) {
    DIR_CHECK_ARG(n >= 0 && F > 0, "dir_hash_bucket_i64_fields_device: bad argument");
    if (n == 0) return DIR_OK;
    DIR_CHECK_ARG(keys && buckets_f && out, "dir_hash_bucket_i64_fields_device: null pointer");
    hipLaunchKernelGGL(hash_bucket_i64_k, dim3(grid_for((n + 255) / 256)), dim3(256), 0, as_stream(stream), keys, n,
                       (uint64_t)1, buckets_f, F, out);
    DIR_CHECK_LAUNCH("hash_bucket_i64_fields");
    return DIR_OK;
}

extern "C" int dir_hash_bucket_bytes_device(const char* bytes, const int64_t* offsets, int64_t n, int64_t num_buckets,
                                            int64_t* out, dir_stream_t stream) {
    DIR_CHECK_ARG(n >= 0 && num_buckets > 0, "dir_hash_bucket_bytes_device: bad argument");
    if (n == 0) return DIR_OK;
    DIR_CHECK_ARG(offsets && out, "dir_hash_bucket_bytes_device: null pointer");
    hipLaunchKernelGGL(hash_bucket_bytes_k, dim3(grid_for((n + 255) / 256)), dim3(256), 0, as_stream(stream), bytes, offsets,
                       n, (uint64_t)num_buckets, out);
    DIR_CHECK_LAUNCH("hash_bucket_bytes");
    return DIR_OK;
}

extern "C" int dir_bucketize_f32(const float* x, int64_t n, const float* boundaries, int nb, int64_t* out,
                                 dir_stream_t stream) {
    DIR_CHECK_ARG(x && out && n >= 0 && nb >= 0 && (nb == 0 || boundaries), "dir_bucketize_f32: bad argument");
    if (n == 0) return DIR_OK;
    hipLaunchKernelGGL(bucketize_k, dim3(grid_for((n + 255) / 256)), dim3(256), 0, as_stream(stream), x, n, boundaries,
                       nb, out);
    DIR_CHECK_LAUNCH("bucketize");
    return DIR_OK;
}

extern "C" void dir_shard_div_owner(int64_t id, int64_t vocab, int P, int* owner, int64_t* local) {
    const int64_t q = vocab / P, r = vocab % P;
    div_owner(id, q, r, r * (q + 1), owner, local);
}

extern "C" int dir_shard_route(const int64_t* ids, int64_t n, const int64_t* vocab, int F, int P, int32_t* owner,
                               int64_t* local, dir_stream_t stream) {
    DIR_CHECK_ARG(ids && vocab && owner && local && n >= 0 && F > 0 && P > 0, "dir_shard_route: bad argument");
    if (n == 0) return DIR_OK;
    hipLaunchKernelGGL(shard_route_k, dim3(grid_for((n + 255) / 256)), dim3(256), 0, as_stream(stream), ids, n, vocab, F,
                       P, owner, local);
    DIR_CHECK_LAUNCH("shard_route");
    return DIR_OK;
}

extern "C" int dir_gather_rows_f32(const float* const* tables, int K, const int32_t* slot, const int64_t* row,
                                   int64_t n, float* out, dir_stream_t stream) {
    DIR_CHECK_ARG(tables && row && out && K > 0 && n >= 0, "dir_gather_rows_f32: bad argument");
    if (n == 0) return DIR_OK;
    const bool vec = (K % 4 == 0) && aligned16(out);
    int lps = 1;
    while (lps < (vec ? K / 4 : K)) lps <<= 1;
    dim3 grid(grid_for((n * lps + 255) / 256));
    if (vec) hipLaunchKernelGGL((gather_rows_k<4>), grid, dim3(256), 0, as_stream(stream), tables, K, lps, slot, row, n, out);
    else hipLaunchKernelGGL((gather_rows_k<1>), grid, dim3(256), 0, as_stream(stream), tables, K, lps, slot, row, n, out);
    DIR_CHECK_LAUNCH("gather_rows");
    return DIR_OK;
}

extern "C" int64_t dir_shard_bucket_workspace_bytes(int64_t n, int P) {
    const int64_t nwg = (n + BK_EPB - 1) / BK_EPB;
    return 2 * nwg * (int64_t)P * (int64_t)sizeof(int32_t);
}

extern "C" int dir_shard_bucket(const int64_t* ids, int64_t n, const int64_t* vocab, const int32_t* parts, const int32_t* first,
                                int F, int P, int64_t* payload,
                                int64_t* inv, int64_t* counts, int64_t* starts, void* workspace, dir_stream_t stream) {
    DIR_CHECK_ARG(n >= 0 && F > 0 && P > 0 && P <= 64, "dir_shard_bucket: n=%lld F=%d P=%d (P <= 64)", (long long)n, F, P);
    DIR_CHECK_ARG(counts && starts, "dir_shard_bucket: null pointer");
    hipStream_t st = as_stream(stream);
    if (n == 0) {
        if (zero_async(counts, sizeof(int64_t) * P, st) != hipSuccess || zero_async(starts, sizeof(int64_t) * P, st) != hipSuccess)
            return fail(DIR_E_HIP, "dir_shard_bucket: memset failed");
        return DIR_OK;
    }
    DIR_CHECK_ARG(ids && vocab && payload && inv && workspace, "dir_shard_bucket: null pointer");
    const int nwg = (int)((n + BK_EPB - 1) / BK_EPB);
    int32_t* wg_counts = static_cast<int32_t*>(workspace);
    int32_t* wg_base = wg_counts + (int64_t)nwg * P;
    hipLaunchKernelGGL(bucket_hist_k, dim3(nwg), dim3(256), 0, st, ids, n, vocab, parts, first, F, P, wg_counts);
    hipLaunchKernelGGL(bucket_scan_k, dim3(1), dim3(1024), 0, st, wg_counts, nwg, P, wg_base, counts, starts);
    hipLaunchKernelGGL(bucket_scatter_k, dim3(nwg), dim3(256), 0, st, ids, n, vocab, parts, first, F, P, wg_base, starts, payload, inv);
    DIR_CHECK_LAUNCH("shard_bucket");
    return DIR_OK;
}

// 64 slab counters + the arrival counter of bucket_cap2_k (+ padding): all zero before the first call, left zero by every call
extern "C" int64_t dir_shard_bucket_cap_workspace_bytes(int P) { return P > 0 && P <= 64 ? 128 * (int64_t)sizeof(int32_t) : 0; }

extern "C" int dir_shard_bucket_cap(const int64_t* ids, int64_t n, const int64_t* vocab, const int32_t* parts, const int32_t* first,
                                    int F, int P, int64_t cap, int64_t* payload, int64_t* inv, int64_t* counts, int32_t* overflow,
                                    int64_t* stat, void* workspace, dir_stream_t stream) {
    DIR_CHECK_ARG(n >= 0 && F > 0 && P > 0 && P <= 64 && cap > 0, "dir_shard_bucket_cap: n=%lld F=%d P=%d cap=%lld (P <= 64)", (long long)n, F, P, (long long)cap);
    DIR_CHECK_ARG(vocab && payload && counts && overflow && workspace && (n == 0 || (ids && inv)), "dir_shard_bucket_cap: null pointer");
    if ((int64_t)P * cap >= (int64_t)1 << 40) return fail(DIR_E_UNSUPPORTED, "dir_shard_bucket_cap: P*cap too large");
    hipStream_t st = as_stream(stream);
    int32_t* gcount = static_cast<int32_t*>(workspace);
    static const int legacy = getenv("DIR_BUCKET_LEGACY") ? atoi(getenv("DIR_BUCKET_LEGACY")) : 0;      // development A/B switch
    if (n > 0 && !legacy) {
        static const int ept_env = dev_env_int("DIR_BUCKET_EPT", 0);
        static const int nt_env = dev_env_int("DIR_BUCKET_NT", 0);
        const int nt = nt_env ? nt_env : (n <= 256 * 1024 ? 256 : 1024);
        const int ept = ept_env ? ept_env : 4;            // (1024 x 4: 16.1 us, x 8: 18.5, x 16: 30.3 at 1.7 M ids)
#define DIR_BK2(EPT, NT)                                                                                                              \
    hipLaunchKernelGGL((bucket_cap2_k<EPT, NT>), dim3((unsigned)((n + NT * EPT - 1) / (NT * EPT))), dim3(NT), 0, st, ids, n, vocab, parts, first, F, \
                       P, cap, gcount, payload, inv, counts, overflow, stat)
        if (nt != 256 && nt != 512 && nt != 1024) return fail(DIR_E_BADARG, "dir_shard_bucket_cap: DIR_BUCKET_NT must be 256, 512 or 1024");
        if (nt == 512 && ept != 8) return fail(DIR_E_BADARG, "dir_shard_bucket_cap: DIR_BUCKET_NT=512 is instantiated for DIR_BUCKET_EPT=8 only");
        if (nt == 256 && ept <= 4) DIR_BK2(4, 256);
        else if (nt == 256) DIR_BK2(8, 256);
        else if (nt == 512) DIR_BK2(8, 512);
        else if (ept <= 4) DIR_BK2(4, 1024);
        else if (ept <= 8) DIR_BK2(8, 1024);
        else DIR_BK2(16, 1024);
#undef DIR_BK2
        DIR_CHECK_LAUNCH("shard_bucket_cap");
        return DIR_OK;
    }
    if (n > 0) {
#define DIR_BK(EPT)                                                                                                            \
    hipLaunchKernelGGL((bucket_cap_k<EPT>), dim3((unsigned)((n + 256 * EPT - 1) / (256 * EPT))), dim3(256), 0, st, ids, n, vocab, \
                       parts, first, F, P, cap, gcount, payload, inv)
        if (n <= 768 * 1024) DIR_BK(4);
        else if (n <= 2560 * 1024) DIR_BK(8);
        else DIR_BK(16);
#undef DIR_BK
    }
    hipLaunchKernelGGL(bucket_cap_fin_k, dim3(1), dim3(64), 0, st, gcount, P, cap, payload, counts, overflow, stat);
    DIR_CHECK_LAUNCH("shard_bucket_cap");
    return DIR_OK;
}

extern "C" int dir_shard_bucket_cap_dedup(const int64_t* ids, int64_t stride_b, int64_t stride_f, int64_t B, const int64_t* vocab,
                                          const int32_t* parts, const int32_t* first, int F, int P, int64_t cap, int64_t* payload,
                                          int64_t* inv, int64_t* counts, int32_t* overflow, int64_t* stat, void* workspace,
                                          dir_stream_t stream) {
    DIR_CHECK_ARG(B >= 0 && F > 0 && P > 0 && P <= 64 && cap > 0, "dir_shard_bucket_cap_dedup: B=%lld F=%d P=%d cap=%lld (P <= 64)", (long long)B, F, P, (long long)cap);
    DIR_CHECK_ARG(vocab && payload && counts && overflow && workspace && (B == 0 || (ids && inv)), "dir_shard_bucket_cap_dedup: null pointer");
    if ((int64_t)P * cap >= (int64_t)0x7fffffff) return fail(DIR_E_UNSUPPORTED, "dir_shard_bucket_cap_dedup: P*cap must fit int32");
    hipStream_t st = as_stream(stream);
    int32_t* gcount = static_cast<int32_t*>(workspace);
    if (B > 0) {
        if (B * F <= ((int64_t)1 << 20)) {
            const int64_t tiles = (B + 2047) / 2048;
            if (tiles * F >= (int64_t)0x7fffffff) return fail(DIR_E_UNSUPPORTED, "dir_shard_bucket_cap_dedup: too many tiles");
            hipLaunchKernelGGL((bucket_cap_dedup_k<8>), dim3((unsigned)(tiles * F)), dim3(256), 0, st, ids, stride_b, stride_f, B, vocab, parts, first, F, P,
                               cap, gcount, payload, inv);
        } else {
            const int64_t tiles = (B + 4095) / 4096;
            if (tiles * F >= (int64_t)0x7fffffff) return fail(DIR_E_UNSUPPORTED, "dir_shard_bucket_cap_dedup: too many tiles");
            hipLaunchKernelGGL((bucket_cap_dedup_k<16>), dim3((unsigned)(tiles * F)), dim3(256), 0, st, ids, stride_b, stride_f, B, vocab, parts, first, F, P,
                               cap, gcount, payload, inv);
        }
    }
    hipLaunchKernelGGL(bucket_cap_fin_k, dim3(1), dim3(64), 0, st, gcount, P, cap, payload, counts, overflow, stat);
    DIR_CHECK_LAUNCH("shard_bucket_cap_dedup");
    return DIR_OK;
}

extern "C" int dir_shard_slab_stat(const int64_t* recv, int n_slabs, int64_t cap, int64_t* stat, dir_stream_t stream) {
    DIR_CHECK_ARG(recv && stat && n_slabs > 0 && cap > 0, "dir_shard_slab_stat: bad argument");
    hipLaunchKernelGGL(slab_stat_k, dim3(1), dim3(64), 0, as_stream(stream), recv, n_slabs, cap, stat);
    DIR_CHECK_LAUNCH("shard_slab_stat");
    return DIR_OK;
}

extern "C" int dir_gather_slabs_f32(const float* const* tables, int F, int K, int64_t* recv, int P, int64_t cap, int flags,
                                    float* out, dir_stream_t stream) {
    DIR_CHECK_ARG(F > 0 && K > 0 && P > 0 && cap > 0, "dir_gather_slabs_f32: bad argument");
    DIR_CHECK_ARG(tables && recv && out, "dir_gather_slabs_f32: null pointer");
    const bool vec = (K % 4 == 0) && aligned16(out);
    int lps = 1;
    while (lps < (vec ? K / 4 : K)) lps <<= 1;
    const int64_t n = (int64_t)P * cap;
    dim3 grid(grid_for((n * lps + 255) / 256));
    const bool nt = (flags & DIR_GATHER_STREAM_ROWS) != 0;
    const int sanitize = (flags & DIR_SLAB_SANITIZE) ? 1 : 0;
    hipStream_t st = as_stream(stream);
    // (round 4: a 2-D grid form without per-element divisions and with 8 row loads in flight per lane -- gather_slabs16_k -- measured
    // 70.0 us against this kernel's 66.1 on the same box and was removed: the kernel sits at the request floor of 64-byte rows, NOTES R4.2)
    if (vec && nt) hipLaunchKernelGGL((gather_slabs_k<4, true>), grid, dim3(256), 0, st, tables, K, lps, F, recv, P, cap, sanitize, out);
    else if (vec) hipLaunchKernelGGL((gather_slabs_k<4, false>), grid, dim3(256), 0, st, tables, K, lps, F, recv, P, cap, sanitize, out);
    else if (nt) hipLaunchKernelGGL((gather_slabs_k<1, true>), grid, dim3(256), 0, st, tables, K, lps, F, recv, P, cap, sanitize, out);
    else hipLaunchKernelGGL((gather_slabs_k<1, false>), grid, dim3(256), 0, st, tables, K, lps, F, recv, P, cap, sanitize, out);
    DIR_CHECK_LAUNCH("gather_slabs");
    return DIR_OK;
}

extern "C" int dir_gather_packed_f32(const float* const* tables, int F, int K, const int64_t* payload, int64_t n,
                                     int flags, float* out, dir_stream_t stream) {
    DIR_CHECK_ARG(F > 0 && K > 0 && n >= 0, "dir_gather_packed_f32: bad argument");
    if (n == 0) return DIR_OK;
    DIR_CHECK_ARG(tables && payload && out, "dir_gather_packed_f32: null pointer");
    const bool vec = (K % 4 == 0) && aligned16(out);
    int lps = 1;
    while (lps < (vec ? K / 4 : K)) lps <<= 1;
    dim3 grid(grid_for((n * lps + 255) / 256));
    const bool nt = (flags & DIR_GATHER_STREAM_ROWS) != 0;
    hipStream_t st = as_stream(stream);
    if (vec && nt) hipLaunchKernelGGL((gather_packed_k<4, true>), grid, dim3(256), 0, st, tables, K, lps, F, payload, n, out);
    else if (vec) hipLaunchKernelGGL((gather_packed_k<4, false>), grid, dim3(256), 0, st, tables, K, lps, F, payload, n, out);
    else if (nt) hipLaunchKernelGGL((gather_packed_k<1, true>), grid, dim3(256), 0, st, tables, K, lps, F, payload, n, out);
    else hipLaunchKernelGGL((gather_packed_k<1, false>), grid, dim3(256), 0, st, tables, K, lps, F, payload, n, out);
    DIR_CHECK_LAUNCH("gather_packed");
    return DIR_OK;
}
