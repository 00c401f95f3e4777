// radix_sort.hip -- stable LSD radix sort of (uint32 key, uint32 value) pairs for gfx950, in-tree, graph-capturable.
//
// What it is for: the (row, entry) sort in front of every sorted sparse update (csrc/backward.hip: Adagrad / FTRL / Adam on the
// embedding tables -- the reference's optimisers, models/DeepFM/deepFM.py:58,61, models/DeepCrossNetwork/DeepCrossNetwork.py:264-290).
// Rounds 1-3 called rocPRIM's onesweep sort there.  Two things were wrong with that call at 1.7 M pairs (B 65 536 x 26 slots):
//   * it issues 7 hipMemsetAsync per sort (34 us of a 335 us step), and
//   * those memsets make a captured HIP graph unusable: a hipGraph holding memset nodes faults on replay once enough EAGER memsets
//     (another optimiser's sorts) have run in between (profiles/NOTES.md R4.3: tools/graph_part_probe.py isolates it) -- so no
//     training step that sorts could be replayed from a graph.
// This sort launches kernels only: init (zero the histograms, tile states and tickets), one histogram pass over the keys for ALL
// digit positions, then one kernel per digit (8 or 9 bits; 25-bit keys: 3 x 9) with a decoupled look-back over the tiles
// (Merrill & Garland's single-pass scan, as "onesweep" uses it):
//   tile = 1024 threads x 8 keys (16 waves, each owning 512 consecutive keys, lane-strided: coalesced loads, memory order = (i, lane));
//   rank inside the wave by match-any (RB ballots per key) against a per-wave LDS histogram, exclusive scan over the waves per digit,
//   publish the tile's digit counts (flag | count in ONE 32-bit word: no fence needed), sum the predecessors' counts until a tile
//   with an inclusive prefix is met, publish the inclusive prefix, scatter.
// Tiles take their index from a ticket (atomic counter), so a tile only ever waits for tiles that started before it.
// Stable: equal keys keep their input order (the sparse updates sum a row's entries in entry order: bitwise reproducible results).
// One-hot entries ids [B, F] with B >= 4096 take the SLOT-MAJOR form further down (radix_slot_sort_entries): the slot part of the key is
// the entry's position, so the pairs are written slot-major by an LDS transposition and each slot's segment is sorted by its local row,
// 10 bits per launch -- 2 sorting launches instead of 3 at the BASELINE shape, look-backs of 8 tiles instead of 208 (64 us against 95).
#include <string.h>

#include "common.hpp"

namespace dir {

namespace {

constexpr int RS_NT = 1024;                 // threads per tile
constexpr int RS_ITEMS = 8;                 // keys per thread
constexpr int RS_TILE = RS_NT * RS_ITEMS;   // 8192
constexpr int RS_NW = RS_NT / 64;           // 16 waves
constexpr int RS_MAXPASS = 4;
constexpr int RS_WIN = 16;                 // predecessors read per look-back round trip
constexpr uint32_t RS_AGG = 1u << 30, RS_INCL = 2u << 30, RS_VAL = (1u << 30) - 1u;

// development timing masks (results are wrong under them): 1 no scatter, 2 no look-back, 4 no ranking.  Without ranks the scatter
// addresses are garbage, so 4 switches the scatter off as well (a DIR_RS_DBG=4 run once wrote out of bounds; without the look-back a
// digit's keys still land inside that digit's range).
inline int rs_dbg_mask() {
    int d = dev_env_int("DIR_RS_DBG", 0);          // (-DDIR_DEVELOPMENT builds only; radix_sort_dbg_refused() fails the entries otherwise)
    if (d & 4) d |= 1;
    return d;
}

struct RsLayout {                           // offsets into the temp storage (bytes)
    size_t hist, ticket, status, total;
    int rb, passes, bins;
    int64_t ntiles;
};

RsLayout rs_layout(size_t n, unsigned bits) {
    RsLayout L;
    if (bits < 1) bits = 1;
    if (bits > 32) bits = 32;
    const int p8 = (int)((bits + 7) / 8), p9 = (int)((bits + 8) / 9);
    L.rb = p9 < p8 ? 9 : 8;
    L.passes = L.rb == 9 ? p9 : p8;
    L.bins = 1 << L.rb;
    L.ntiles = (int64_t)((n + RS_TILE - 1) / RS_TILE);
    if (L.ntiles < 1) L.ntiles = 1;
    size_t off = 0;
    auto take = [&](size_t b) { const size_t o = off; off += (b + 255) & ~(size_t)255; return o; };
    L.hist = take((size_t)RS_MAXPASS * 512 * 4);
    L.ticket = take(RS_MAXPASS * 4);
    L.status = take((size_t)L.passes * (size_t)L.ntiles * (size_t)L.bins * 4);
    L.total = off;
    return L;
}

__global__ __launch_bounds__(256) void rs_init_k(uint32_t* __restrict__ w, int64_t nwords) {
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < nwords; i += (int64_t)gridDim.x * 256) w[i] = 0u;
}

// digit histograms of every pass in one read of the keys: LDS histograms per workgroup, then one non-returning atomic per non-empty bin
template <int RB>
__global__ __launch_bounds__(1024) void rs_hist_k(const uint32_t* __restrict__ keys, int64_t n, int passes, unsigned bits, uint32_t* __restrict__ ghist,
                                                  uint32_t* __restrict__ status, int64_t status_words) {
    constexpr int BINS = 1 << RB;
    __shared__ uint32_t h[RS_MAXPASS][BINS];
    // the tile states of every pass are zeroed here (nothing reads them before the first pass kernel): rs_init_k only clears the 8 KB
    // of histograms and tickets this kernel's atomics need zero
    for (int64_t i = (int64_t)blockIdx.x * 1024 + threadIdx.x; i < status_words; i += (int64_t)gridDim.x * 1024) status[i] = 0u;
    for (int i = threadIdx.x; i < RS_MAXPASS * BINS; i += 1024) (&h[0][0])[i] = 0u;
    __syncthreads();
    const uint32_t kmask = bits >= 32 ? 0xffffffffu : ((1u << bits) - 1u);
    for (int64_t i = (int64_t)blockIdx.x * 1024 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 1024) {
        const uint32_t k = keys[i] & kmask;
#pragma unroll
        for (int p = 0; p < RS_MAXPASS; ++p)
            if (p < passes) atomicAdd(&h[p][(k >> (p * RB)) & (BINS - 1)], 1u);
    }
    __syncthreads();
    for (int i = threadIdx.x; i < passes * BINS; i += 1024) {
        const uint32_t c = (&h[0][0])[i];
        if (c) atomicAdd(&ghist[(i / BINS) * 512 + (i % BINS)], c);
    }
}

__device__ __forceinline__ uint32_t ld_status(const uint32_t* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void st_status(uint32_t* p, uint32_t v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

// The slot-major form (SLOT = true, see radix_slot_sort_entries below): the pairs are F segments of seg_len entries (slot f's entries of the
// batch, local row as the key), a tile lies inside one segment, histograms / tile states / the look-back are per segment, a segment whose
// vocabulary needs fewer digits than the launch count sits out the first launches, and the last launch writes global rows.
struct RsSlot {
    const int64_t* row_base;     // [F] device
    int64_t seg_len;             // B
    int tps, F, P, pi;           // tiles per segment, slots, launches, this launch
    uint32_t total_rows;
};

// vocabulary of slot f, its first global row and how many 10-bit digits its local keys (0 .. vocab, vocab = pruned) need
__device__ __forceinline__ void rs_slot_info(const int64_t* __restrict__ row_base, int f, int F, uint32_t total_rows, int P, int64_t& base,
                                             uint32_t& vf, int& np) {
    base = row_base[f];
    vf = (uint32_t)((f + 1 < F ? row_base[f + 1] : (int64_t)total_rows) - base);
    const int L = vf ? 32 - __builtin_clz(vf) : 0;
    np = L <= 10 ? 1 : (L + 9) / 10;
    if (np > P) np = P;          // (bit_length(vocab) <= bit_length(total_rows): cannot happen)
}

template <int RB, bool SLOT = false>
__global__ __launch_bounds__(RS_NT) void rs_pass_k(const uint32_t* __restrict__ kin, const uint32_t* __restrict__ vin, uint32_t* __restrict__ kout,
                                                   uint32_t* __restrict__ vout, int64_t n, int shift, unsigned bits,
                                                   const uint32_t* __restrict__ ghist /* [512] of this pass; SLOT: [F][P][BINS] */,
                                                   uint32_t* __restrict__ status /* [ntiles][BINS] of this pass */,
                                                   uint32_t* __restrict__ ticket, int dbg /* development: 1 no scatter, 2 no look-back, 4 no ranking */,
                                                   RsSlot sl) {
    constexpr int BINS = 1 << RB;
    extern __shared__ __attribute__((aligned(16))) uint32_t rs_lds[];
    uint32_t (*whist)[BINS] = reinterpret_cast<uint32_t (*)[BINS]>(rs_lds);     // [RS_NW][BINS] per-wave digit counts, then each wave's exclusive offset inside the tile
    uint32_t* gbase = rs_lds + RS_NW * BINS;                                      // [BINS] where this tile's keys of digit d start in the output
    uint32_t* toff = gbase + BINS;                                                // [BINS] where digit d starts inside the tile (digit-sorted order)
    uint32_t (*scan)[BINS] = reinterpret_cast<uint32_t (*)[BINS]>(toff + BINS);   // scan[0..1][wave]: the waves' totals of the two digit scans (the rest of the [4][BINS] block is unused)
    uint32_t* lkey = toff + BINS + 4 * BINS;                                      // [RS_TILE] the tile's keys in digit order
    uint32_t* lval = lkey + RS_TILE;                                              // [RS_TILE]
    uint32_t& s_tile = lval[RS_TILE];                                             // (all LDS is dynamic: the 160 KiB limit is set for the kernel)
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if (tid == 0) s_tile = atomicAdd(ticket, 1u);
    for (int i = tid; i < RS_NW * BINS; i += RS_NT) (&whist[0][0])[i] = 0u;
    __syncthreads();
    const int64_t tile = s_tile;
    int64_t seg_base = 0, first_tile = 0, tloc = tile, row0 = 0;      // the segment's first pair / first tile, the tile inside it
    uint32_t vf = 0;
    if constexpr (SLOT) {
        const int f = (int)(tile / sl.tps);
        int np;
        rs_slot_info(sl.row_base, f, sl.F, sl.total_rows, sl.P, row0, vf, np);
        if (sl.pi < sl.P - np) return;                   // this segment's keys need fewer digits: it starts at a later launch (whole tile)
        shift = RB * (sl.pi - (sl.P - np));
        ghist += ((int64_t)f * sl.P + (sl.pi - (sl.P - np))) * BINS;
        seg_base = (int64_t)f * sl.seg_len;
        first_tile = (int64_t)f * sl.tps;
        tloc = tile - first_tile;
        n = sl.seg_len;
        bits = 32;
    }
    const int64_t cbase = tloc * RS_TILE + (int64_t)wave * (64 * RS_ITEMS);
    const uint32_t kmask = bits >= 32 ? 0xffffffffu : ((1u << bits) - 1u);
    const uint32_t gh = tid < BINS ? ghist[tid] : 0u;      // (asked for here: its latency runs under the key loads and the ranking)
    uint32_t key[RS_ITEMS], val[RS_ITEMS];
    uint32_t rank[RS_ITEMS];                     // digit << 16 | rank inside the wave's chunk (< 512)
#pragma unroll
    for (int i = 0; i < RS_ITEMS; ++i) {
        const int64_t idx = cbase + i * 64 + lane;
        key[i] = idx < n ? kin[seg_base + idx] : 0xffffffffu;
    }
#pragma unroll
    for (int i = 0; i < RS_ITEMS; ++i) {         // (the values ride along from here: loaded where they are used, their latency sat between
        const int64_t idx = cbase + i * 64 + lane;   //  the scans and the look-back of every tile)
        val[i] = idx < n ? vin[seg_base + idx] : 0u;
    }
    const unsigned long long lt = (1ull << lane) - 1ull;
#pragma unroll
    for (int i = 0; i < RS_ITEMS; ++i) {
        const int64_t idx = cbase + i * 64 + lane;
        const bool valid = idx < n;
        const uint32_t d = ((key[i] & kmask) >> shift) & (BINS - 1);
        unsigned long long peers = __ballot(valid);
        if (!(dbg & 4))
#pragma unroll
        for (int b = 0; b < RB; ++b) {
            const bool bit = (d >> b) & 1u;
            const unsigned long long bal = __ballot(bit);
            peers &= bit ? bal : ~bal;
        }
        // every peer reads the wave's running count of its digit, then the lowest peer adds the group (a wave's LDS operations execute
        // in program order, and no other wave touches this row of whist)
        uint32_t old = 0;
        if (valid) old = whist[wave][d];
        if (valid && (peers & lt) == 0ull) whist[wave][d] = old + (uint32_t)__popcll(peers);
        rank[i] = (d << 16) | (old + (uint32_t)__popcll(peers & lt));
    }
    __syncthreads();
    // per digit: exclusive scan over the waves, the tile's count, the look-back over earlier tiles
    uint32_t cnt = 0;
    if (tid < BINS) {
        uint32_t run = 0;
#pragma unroll
        for (int w = 0; w < RS_NW; ++w) {
            const uint32_t c = whist[w][tid];
            whist[w][tid] = run;
            run += c;
        }
        cnt = run;
        st_status(status + tile * BINS + tid, (tloc == 0 ? RS_INCL : RS_AGG) | cnt);
    }
    // two scans over the digits at once -- the pass's global histogram (where digit d starts in the output) and this tile's counts (where
    // digit d starts inside the tile): inside a wave by shuffles (64 digits per wave), the waves' totals through LDS.  (Ten Hillis-Steele
    // steps over LDS with a workgroup barrier each were ~1.5 us of every tile's path.)
    uint32_t gincl = gh, tincl = cnt;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const uint32_t ua = __shfl_up(gincl, o), ub = __shfl_up(tincl, o);
        if (lane >= o) {
            gincl += ua;
            tincl += ub;
        }
    }
    if (lane == 63) {
        scan[0][wave] = gincl;
        scan[1][wave] = tincl;
    }
    __syncthreads();
    {
        uint32_t pa = 0, pb = 0;
#pragma unroll
        for (int w = 0; w < RS_NW; ++w) {
            const uint32_t ta = scan[0][w], tb = scan[1][w];       // (broadcast reads)
            if (w < wave) {
                pa += ta;
                pb += tb;
            }
        }
        gincl += pa;
        tincl += pb;
    }
    if (tid < BINS) toff[tid] = tincl - cnt;
    __syncthreads();
    // the tile in digit order, in LDS (needs nothing from other tiles: it runs while they publish their counts).  Direct stores from the
    // ranked registers were 12 of a pass's 28 us: every lane of a store hit its own cache line
#pragma unroll
    for (int i = 0; i < RS_ITEMS; ++i) {
        const int64_t idx = cbase + i * 64 + lane;
        if (idx < n) {
            const uint32_t d = rank[i] >> 16;
            const uint32_t lp = toff[d] + whist[wave][d] + (rank[i] & 0xffffu);
            lkey[lp] = key[i];
            lval[lp] = val[i];
        }
    }
    if (tid < BINS) {
        const uint32_t gexcl = gincl - gh;
        uint32_t prefix = 0;
        if (tloc > 0 && !(dbg & 2)) {
            // Windowed look-back: RS_WIN predecessors' words are loaded at once (independent loads: one latency per window, not per
            // tile).  The tiles of a launch start together: a walk that reads ONE predecessor per round trip needs ~sqrt(2 t) round
            // trips of a device-scope load for tile t; with a window of 16 the first 16 k (k + 1) / 2 tiles are done after k round trips.
            bool done = false;
            for (int64_t j = tile - 1; !done; j -= RS_WIN) {
                uint32_t s[RS_WIN];
#pragma unroll
                for (int u = 0; u < RS_WIN; ++u) s[u] = j - u >= first_tile ? ld_status(status + (j - u) * BINS + tid) : RS_INCL;      // "tile -1": inclusive, 0
#pragma unroll
                for (int u = 0; u < RS_WIN; ++u) {
                    if (done) continue;
                    while ((s[u] >> 30) == 0u) {                       // not published yet (rare: the tiles rank in lockstep)
                        __builtin_amdgcn_s_sleep(1);
                        s[u] = ld_status(status + (j - u) * BINS + tid);
                    }
                    prefix += s[u] & RS_VAL;
                    if ((s[u] >> 30) == 2u) done = true;
                }
            }
            st_status(status + tile * BINS + tid, RS_INCL | (prefix + cnt));
        }
        gbase[tid] = (uint32_t)seg_base + gexcl + prefix - toff[tid];          // output position of the tile's digit-sorted element j of digit d: gbase[d] + j
    }
    __syncthreads();
    const int64_t left = n - tloc * RS_TILE;
    const int nvalid = (int)(left < RS_TILE ? left : RS_TILE);
    if (!(dbg & 1)) {
#pragma unroll
        for (int i = 0; i < RS_ITEMS; ++i) {
            const int j = i * RS_NT + tid;                  // consecutive lanes: consecutive elements of a digit's run: consecutive addresses
            if (j < nvalid) {
                const uint32_t k = lkey[j];
                const uint32_t d = ((k & kmask) >> shift) & (BINS - 1);
                const uint32_t pos = gbase[d] + (uint32_t)j;
                if constexpr (SLOT) kout[pos] = sl.pi + 1 < sl.P ? k : (k < vf ? (uint32_t)(row0 + k) : sl.total_rows);     // last launch: global rows
                else kout[pos] = k;
                vout[pos] = lval[j];
            }
        }
    }
}

// ---- slot-major sort of one-hot entries (round 4) ----------------------------------------------------------------------------------------
// The sparse updates sort (global row, entry) pairs of ids [B, F]: entry e = b F + f belongs to slot f, and global rows of different slots
// never compare equal -- the most significant part of the key is known from the entry's POSITION.  So: write the pairs slot-major (segment
// f = slot f's B entries in batch order: an LDS transposition, no sort pass), then sort every segment by the LOCAL row alone, 10 bits per
// launch.  A 10^6-row vocabulary needs 2 launches where the 25-bit global keys needed 3, and a tile's look-back stays inside its
// segment (8 tiles at B = 65 536 instead of 208).  The host only knows total_rows, so it launches P = ceil(bits(total_rows) / 10) passes;
// a segment whose vocabulary needs np < P digits starts in the buffer (np - 1) & 1 and its tiles sit out the first P - np launches --
// every segment's sorted pairs end in k1 / v1 whatever its np.  Output: the same keys (global row, total_rows for a pruned id) the
// global sort produces, ordered by (slot, local row, batch position); pruned entries close their SEGMENT instead of the whole array
// (the update kernels skip keys >= total_rows wherever they stand).
constexpr int RSS_RB = 10, RSS_BINS = 1 << RSS_RB, RSS_FC = 32, RSS_MIN_B = 4096;

constexpr int RSS_KB = 64;        // batch rows per transposition step (256 of them left one workgroup per CU and 18 us; 64: four per CU)
__global__ __launch_bounds__(256) void rss_keys_k(const int64_t* __restrict__ ids, int64_t sb, int64_t sf, int F, int64_t B,
                                                  const int64_t* __restrict__ row_base, uint32_t total_rows, int P, uint32_t* __restrict__ kA,
                                                  uint32_t* __restrict__ kB, uint32_t* __restrict__ vA, uint32_t* __restrict__ vB,
                                                  uint32_t* __restrict__ zero_words, int64_t nzero) {
    __shared__ uint32_t s[RSS_FC][RSS_KB + 1];
    __shared__ uint32_t svf[RSS_FC];
    __shared__ int sbuf[RSS_FC];
    // the histograms and tickets of this sort (nothing reads them before the histogram kernel, the next launch on the stream)
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < nzero; i += (int64_t)gridDim.x * 256) zero_words[i] = 0u;
    for (int fc = 0; fc < F; fc += RSS_FC) {
        const int fw = F - fc < RSS_FC ? F - fc : RSS_FC;
        __syncthreads();
        if ((int)threadIdx.x < fw) {
            int64_t base;
            uint32_t vf;
            int np;
            rs_slot_info(row_base, fc + threadIdx.x, F, total_rows, P, base, vf, np);
            svf[threadIdx.x] = vf;
            sbuf[threadIdx.x] = (np - 1) & 1;
        }
        for (int64_t b0 = (int64_t)blockIdx.x * RSS_KB; b0 < B; b0 += (int64_t)gridDim.x * RSS_KB) {
            const int nb = (int)(B - b0 < RSS_KB ? B - b0 : RSS_KB);
            __syncthreads();
            for (int e = threadIdx.x; e < nb * fw; e += 256) {
                const int i = e / fw, j = e - i * fw;
                const int64_t id = ids[(b0 + i) * sb + (int64_t)(fc + j) * sf];
                const uint32_t vf = svf[j];
                s[j][i] = (uint64_t)id < (uint64_t)vf ? (uint32_t)id : vf;        // pruned ids (id < 0 or >= vocab) sort behind the slot's rows
            }
            __syncthreads();
            for (int e = threadIdx.x; e < fw * RSS_KB; e += 256) {
                const int j = e / RSS_KB, i = e % RSS_KB;
                if (i < nb) {
                    const int64_t pos = (int64_t)(fc + j) * B + b0 + i;
                    (sbuf[j] ? kB : kA)[pos] = s[j][i];
                    (sbuf[j] ? vB : vA)[pos] = (uint32_t)((b0 + i) * F + fc + j);
                }
            }
        }
    }
}

// per segment: the histograms of every digit position in one read of the keys (and the tile states of every launch zeroed)
__global__ __launch_bounds__(1024) void rss_hist_k(const uint32_t* __restrict__ kA, const uint32_t* __restrict__ kB, int64_t B, int F,
                                                   const int64_t* __restrict__ row_base, uint32_t total_rows, int P, uint32_t* __restrict__ ghist,
                                                   uint32_t* __restrict__ status, int64_t status_words) {
    __shared__ uint32_t h[RS_MAXPASS][RSS_BINS];
    const int f = blockIdx.y;
    const int64_t bid = (int64_t)blockIdx.y * gridDim.x + blockIdx.x, nblk = (int64_t)gridDim.x * gridDim.y;
    for (int64_t i = bid * 1024 + threadIdx.x; i < status_words; i += nblk * 1024) status[i] = 0u;
    for (int i = threadIdx.x; i < RS_MAXPASS * RSS_BINS; i += 1024) (&h[0][0])[i] = 0u;
    __syncthreads();
    int64_t base;
    uint32_t vf;
    int np;
    rs_slot_info(row_base, f, F, total_rows, P, base, vf, np);
    const uint32_t* __restrict__ keys = (((np - 1) & 1) ? kB : kA) + (int64_t)f * B;
    for (int64_t i = (int64_t)blockIdx.x * 1024 + threadIdx.x; i < B; i += (int64_t)gridDim.x * 1024) {
        const uint32_t k = keys[i];
#pragma unroll
        for (int p = 0; p < RS_MAXPASS; ++p)
            if (p < np) atomicAdd(&h[p][(k >> (p * RSS_RB)) & (RSS_BINS - 1)], 1u);
    }
    __syncthreads();
    for (int i = threadIdx.x; i < np * RSS_BINS; i += 1024) {
        const uint32_t c = (&h[0][0])[i];
        if (c) atomicAdd(&ghist[((int64_t)f * P + i / RSS_BINS) * RSS_BINS + (i % RSS_BINS)], c);
    }
}

struct RssLayout { size_t hist, ticket, status, total; int P; int64_t tps, ntiles; };
RssLayout rss_layout(int64_t B, int F, unsigned gbits) {
    RssLayout L;
    L.P = (int)((gbits + RSS_RB - 1) / RSS_RB);
    if (L.P < 1) L.P = 1;
    L.tps = (B + RS_TILE - 1) / RS_TILE;
    L.ntiles = L.tps * F;
    size_t off = 0;
    auto take = [&](size_t b) { const size_t o = off; off += (b + 255) & ~(size_t)255; return o; };
    L.hist = take((size_t)F * L.P * RSS_BINS * 4);
    L.ticket = take(RS_MAXPASS * 4);
    L.status = take((size_t)L.P * (size_t)L.ntiles * RSS_BINS * 4);
    L.total = off;
    return L;
}

__global__ __launch_bounds__(256) void rs_copy2_k(const uint32_t* __restrict__ a, uint32_t* __restrict__ b, const uint32_t* __restrict__ c,
                                                  uint32_t* __restrict__ d, int64_t n) {
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        b[i] = a[i];
        d[i] = c[i];
    }
}

}  // namespace

size_t radix_sort_temp_bytes(size_t n, unsigned bits) { return rs_layout(n, bits).total; }

int radix_sort_input_buffer(size_t n, unsigned bits) { return (rs_layout(n, bits).passes & 1) ? 0 : 1; }

// Sorts n pairs by the low `bits` bits of the key.  The input pairs must be in buffer radix_sort_input_buffer(n, bits) (0: k0 / v0,
// 1: k1 / v1); the sorted pairs end up in k1 / v1; the other buffer is scratch.  tmp: radix_sort_temp_bytes(n, bits) bytes, 256-byte
// aligned, any content.  n < 2^30.  Kernel launches only (no memset / memcpy nodes): safe inside a HIP-graph capture.
hipError_t radix_sort_pairs_u32(void* tmp, uint32_t* k0, uint32_t* k1, uint32_t* v0, uint32_t* v1, size_t n, unsigned bits, hipStream_t st) {
    if (dev_env_ignored("DIR_RS_DBG")) {           // a timing mask that corrupts the sort: never silently ignored, never honoured in a production build
        fprintf(stderr, "dir_hip: DIR_RS_DBG is set but this library was not built with -DDIR_DEVELOPMENT: refusing to sort\n");
        return hipErrorInvalidValue;
    }
    if (n == 0) return hipSuccess;
    if (n >= ((size_t)1 << 30)) return hipErrorInvalidValue;
    const RsLayout L = rs_layout(n, bits);
    char* base = static_cast<char*>(tmp);
    uint32_t* ghist = reinterpret_cast<uint32_t*>(base + L.hist);
    uint32_t* ticket = reinterpret_cast<uint32_t*>(base + L.ticket);
    uint32_t* status = reinterpret_cast<uint32_t*>(base + L.status);
    const int64_t nwords = (int64_t)(L.status / 4);          // histograms + tickets (they sit in front of the tile states)
    const int64_t status_words = (int64_t)((L.total - L.status) / 4);
    hipLaunchKernelGGL(rs_init_k, dim3((unsigned)((nwords + 255) / 256)), dim3(256), 0, st, reinterpret_cast<uint32_t*>(base), nwords);
    uint32_t* kb[2] = {k0, k1};
    uint32_t* vb[2] = {v0, v1};
    int cur = (L.passes & 1) ? 0 : 1;
    static const int dbg = rs_dbg_mask();
    const int64_t hb = ((int64_t)n + 8191) / 8192;
    const unsigned hgrid = (unsigned)(hb < kCUs * 2 ? hb : kCUs * 2);
    if (L.rb == 9) hipLaunchKernelGGL((rs_hist_k<9>), dim3(hgrid), dim3(1024), 0, st, kb[cur], (int64_t)n, L.passes, bits, ghist, status, status_words);
    else hipLaunchKernelGGL((rs_hist_k<8>), dim3(hgrid), dim3(1024), 0, st, kb[cur], (int64_t)n, L.passes, bits, ghist, status, status_words);
    const size_t lds = sizeof(uint32_t) * ((size_t)RS_NW * L.bins + 6 * (size_t)L.bins + 2 * (size_t)RS_TILE + 4);
    static LdsOnce once8, once9;
    if (!(L.rb == 9 ? lds_limit(once9, 160 * 1024, &rs_pass_k<9>) : lds_limit(once8, 160 * 1024, &rs_pass_k<8>))) return hipErrorInvalidValue;
    for (int p = 0; p < L.passes; ++p) {
        uint32_t* stp = status + (size_t)p * (size_t)L.ntiles * (size_t)L.bins;
        if (L.rb == 9)
            hipLaunchKernelGGL((rs_pass_k<9>), dim3((unsigned)L.ntiles), dim3(RS_NT), lds, st, kb[cur], vb[cur], kb[cur ^ 1], vb[cur ^ 1], (int64_t)n,
                               p * 9, bits, ghist + p * 512, stp, ticket + p, dbg, RsSlot{});
        else
            hipLaunchKernelGGL((rs_pass_k<8>), dim3((unsigned)L.ntiles), dim3(RS_NT), lds, st, kb[cur], vb[cur], kb[cur ^ 1], vb[cur ^ 1], (int64_t)n,
                               p * 8, bits, ghist + p * 512, stp, ticket + p, dbg, RsSlot{});
        cur ^= 1;
    }
    return hipGetLastError();
}

// The slot-major sort (see rss_keys_k): eligible when the entries are ids [B, F] with B large enough to fill tiles.
bool radix_slot_sort_ok(int64_t B, int F, unsigned gbits) {
    static const bool off = dev_env("DIR_SORT") && !strcmp(dev_env("DIR_SORT"), "global");      // development A/B switch
    return !off && B >= RSS_MIN_B && F >= 1 && F <= 65535 && gbits >= 1 && gbits <= 32 && B * F < ((int64_t)1 << 30);
}

size_t radix_slot_sort_temp_bytes(int64_t B, int F, unsigned gbits) { return rss_layout(B, F, gbits).total; }

// ids [B, F] (strides sb, sf) -> the (global row, entry b F + f) pairs sorted by (slot, local row, b) in k1 / v1; k0 / v0 scratch.
// gbits = bit_length(total_rows).  tmp: radix_slot_sort_temp_bytes bytes, 256-byte aligned, any content.  Kernel launches only.
hipError_t radix_slot_sort_entries(void* tmp, const int64_t* ids, int64_t sb, int64_t sf, int F, int64_t B, const int64_t* row_base,
                                   uint32_t total_rows, unsigned gbits, uint32_t* k0, uint32_t* k1, uint32_t* v0, uint32_t* v1, hipStream_t st) {
    if (dev_env_ignored("DIR_RS_DBG")) {
        fprintf(stderr, "dir_hip: DIR_RS_DBG is set but this library was not built with -DDIR_DEVELOPMENT: refusing to sort\n");
        return hipErrorInvalidValue;
    }
    if (!radix_slot_sort_ok(B, F, gbits)) return hipErrorInvalidValue;
    const RssLayout L = rss_layout(B, F, gbits);
    char* base = static_cast<char*>(tmp);
    uint32_t* ghist = reinterpret_cast<uint32_t*>(base + L.hist);
    uint32_t* ticket = reinterpret_cast<uint32_t*>(base + L.ticket);
    uint32_t* status = reinterpret_cast<uint32_t*>(base + L.status);
    const int64_t nzero = (int64_t)(L.status / 4), status_words = (int64_t)((L.total - L.status) / 4);
    const int64_t kb = (B + RSS_KB - 1) / RSS_KB;
    hipLaunchKernelGGL(rss_keys_k, dim3((unsigned)(kb < kCUs * 8 ? kb : kCUs * 8)), dim3(256), 0, st, ids, sb, sf, F, B, row_base, total_rows, L.P, k0, k1,
                       v0, v1, reinterpret_cast<uint32_t*>(base), nzero);
    hipLaunchKernelGGL(rss_hist_k, dim3((unsigned)L.tps, (unsigned)F), dim3(1024), 0, st, k0, k1, B, F, row_base, total_rows, L.P, ghist, status,
                       status_words);
    const size_t lds = sizeof(uint32_t) * ((size_t)RS_NW * RSS_BINS + 6 * (size_t)RSS_BINS + 2 * (size_t)RS_TILE + 4);
    static LdsOnce once;
    if (!lds_limit(once, 160 * 1024, &rs_pass_k<RSS_RB, true>)) return hipErrorInvalidValue;
    static const int dbg = rs_dbg_mask();
    uint32_t* kbuf[2] = {k0, k1};
    uint32_t* vbuf[2] = {v0, v1};
    for (int pi = 0; pi < L.P; ++pi) {
        const int cur = (L.P - 1 - pi) & 1;              // the last launch reads buffer 0 and writes buffer 1
        const RsSlot sl{row_base, B, (int)L.tps, F, L.P, pi, total_rows};
        hipLaunchKernelGGL((rs_pass_k<RSS_RB, true>), dim3((unsigned)L.ntiles), dim3(RS_NT), lds, st, kbuf[cur], vbuf[cur], kbuf[cur ^ 1], vbuf[cur ^ 1],
                           B, 0, 32u, ghist, status + (size_t)pi * (size_t)L.ntiles * RSS_BINS, ticket + pi, dbg, sl);
    }
    return hipGetLastError();
}

// ---- zero fills as kernels (hipMemsetAsync becomes a memset NODE under graph capture, see the header comment) ------------------------
namespace {
__global__ __launch_bounds__(256) void zero_words_k(uint32_t* __restrict__ p, int64_t n) {
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) p[i] = 0u;
}
__global__ __launch_bounds__(256) void zero_2d_k(uint32_t* __restrict__ p, int64_t pitch_words, int64_t width_words, int64_t rows) {
    const int64_t total = width_words * rows;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int64_t r = i / width_words;
        p[r * pitch_words + (i - r * width_words)] = 0u;
    }
}
}  // namespace

hipError_t zero_async(void* p, size_t bytes, hipStream_t st) {
    if (bytes == 0) return hipSuccess;
    if ((reinterpret_cast<uintptr_t>(p) & 3u) || (bytes & 3u)) return hipErrorInvalidValue;
    const int64_t n = (int64_t)(bytes / 4), b = (n + 255) / 256;
    hipLaunchKernelGGL(zero_words_k, dim3((unsigned)(b < kCUs * 8 ? b : kCUs * 8)), dim3(256), 0, st, static_cast<uint32_t*>(p), n);
    return hipGetLastError();
}

hipError_t zero_2d_async(void* p, size_t pitch_bytes, size_t width_bytes, size_t rows, hipStream_t st) {
    if (width_bytes == 0 || rows == 0) return hipSuccess;
    if ((reinterpret_cast<uintptr_t>(p) & 3u) || (pitch_bytes & 3u) || (width_bytes & 3u)) return hipErrorInvalidValue;
    const int64_t total = (int64_t)(width_bytes / 4) * (int64_t)rows, b = (total + 255) / 256;
    hipLaunchKernelGGL(zero_2d_k, dim3((unsigned)(b < kCUs * 8 ? b : kCUs * 8)), dim3(256), 0, st, static_cast<uint32_t*>(p), (int64_t)(pitch_bytes / 4),
                       (int64_t)(width_bytes / 4), (int64_t)rows);
    return hipGetLastError();
}

}  // namespace dir

using namespace dir;

extern "C" int64_t dir_debug_radix_sort_workspace_bytes(int64_t n, int bits) {
    if (n < 0 || n >= ((int64_t)1 << 30) || bits < 1 || bits > 32) return 0;
    return (int64_t)radix_sort_temp_bytes((size_t)n, (unsigned)bits) + 256;
}

extern "C" int dir_debug_radix_sort_pairs_u32(uint32_t* keys_in, uint32_t* vals_in, int64_t n, int bits, uint32_t* keys_out, uint32_t* vals_out,
                                              void* workspace, int64_t workspace_bytes, dir_stream_t stream) {
    DIR_CHECK_ARG(n >= 0 && n < ((int64_t)1 << 30) && bits >= 1 && bits <= 32, "dir_debug_radix_sort_pairs_u32: n=%lld bits=%d", (long long)n, bits);
    if (n == 0) return DIR_OK;
    DIR_CHECK_ARG(keys_in && vals_in && keys_out && vals_out && workspace, "dir_debug_radix_sort_pairs_u32: null pointer");
    char* ws = static_cast<char*>(workspace);
    ws += (256 - (reinterpret_cast<uintptr_t>(ws) & 255u)) & 255u;
    if ((int64_t)radix_sort_temp_bytes((size_t)n, (unsigned)bits) + (ws - static_cast<char*>(workspace)) > workspace_bytes)
        return fail(DIR_E_BADARG, "dir_debug_radix_sort_pairs_u32: workspace too small");
    hipStream_t st = as_stream(stream);
    if (radix_sort_input_buffer((size_t)n, (unsigned)bits) == 1) {      // an even number of digit passes starts from the output buffers
        const int64_t b = (n + 255) / 256;
        hipLaunchKernelGGL(rs_copy2_k, dim3((unsigned)(b < kCUs * 8 ? b : kCUs * 8)), dim3(256), 0, st, keys_in, keys_out, vals_in, vals_out, n);
    }
    if (radix_sort_pairs_u32(ws, keys_in, keys_out, vals_in, vals_out, (size_t)n, (unsigned)bits, st) != hipSuccess)
        return fail(DIR_E_HIP, "dir_debug_radix_sort_pairs_u32: launch failed");
    return DIR_OK;
}

extern "C" int64_t dir_debug_slot_sort_workspace_bytes(int64_t B, int F, int64_t total_rows) {
    if (B <= 0 || F <= 0 || total_rows <= 0 || total_rows >= 0xffffffffll) return 0;
    unsigned bits = 1;
    while (bits < 32 && (((uint64_t)1 << bits) <= (uint64_t)total_rows)) ++bits;
    if (!radix_slot_sort_ok(B, F, bits)) return 0;
    return (int64_t)radix_slot_sort_temp_bytes(B, F, bits) + 256 + 2 * (((int64_t)B * F * 4 + 255) & ~(int64_t)255);
}

// test entry: ids [B, F] -> the sorted (global row, entry) pairs the sparse updates consume (0 workspace bytes: shape not covered)
extern "C" int dir_debug_slot_sort_entries(const int64_t* ids, int64_t stride_b, int64_t stride_f, int F, int64_t B, const int64_t* row_base,
                                           int64_t total_rows, uint32_t* keys_out, uint32_t* vals_out, void* workspace, int64_t workspace_bytes,
                                           dir_stream_t stream) {
    const int64_t need = dir_debug_slot_sort_workspace_bytes(B, F, total_rows);
    DIR_CHECK_ARG(need > 0 && ids && row_base && keys_out && vals_out && workspace && workspace_bytes >= need, "dir_debug_slot_sort_entries: bad argument");
    unsigned bits = 1;
    while (bits < 32 && (((uint64_t)1 << bits) <= (uint64_t)total_rows)) ++bits;
    char* ws = static_cast<char*>(workspace);
    ws += (256 - (reinterpret_cast<uintptr_t>(ws) & 255u)) & 255u;
    const size_t nb = ((size_t)B * F * 4 + 255) & ~(size_t)255;
    uint32_t* k0 = reinterpret_cast<uint32_t*>(ws);
    uint32_t* v0 = reinterpret_cast<uint32_t*>(ws + nb);
    if (radix_slot_sort_entries(ws + 2 * nb, ids, stride_b, stride_f, F, B, row_base, (uint32_t)total_rows, bits, k0, keys_out, v0, vals_out,
                                as_stream(stream)) != hipSuccess)
        return fail(DIR_E_HIP, "dir_debug_slot_sort_entries: launch failed");
    return DIR_OK;
}
