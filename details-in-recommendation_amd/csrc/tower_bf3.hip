// tower_bf3.hip -- a whole DNN tower in ONE launch: up to four hidden layers of width <= 416 and, optionally, the units = 1 logit layer
// behind them, with the activations of a 128-row tile never leaving the chip.
//
// Reference: dnn_logit_fn (models/DeepFM/deepFM.py:284-319: concat -> [dense(units, act) -> batch_normalization]* -> dense(units=1)),
// the tower of _base_model (models/ESMM/ESMM.py:139-146) and the 400-wide DNN beside the CIN.  Same arithmetic as dense_bf3.hip (every
// fp32 operand split into three bf16 pieces by round-to-nearest, the six piece products of weight >= 2^-16 accumulated in fp32 on the
// bf16 matrix pipe), same 1e-5 bar against float64; what changes is where the data lives.
//
// Why.  Layer by layer (dense_bf3_k), a 400-wide layer at batch 65 536 writes a 105 MB activation and the next one reads it back and
// splits it again: 3 x 117 us + a library GEMV for the head, 0.45 of the bf16 pipe, at the HBM ridge.  Here only the layer-1 input
// (the [B, 416] concat) is read and only the [B] logit (or the last activation) is written.
//
// How.  One workgroup = 8 waves (two per SIMD) = 128 rows; wave w owns rows [16w, 16w+16) through every layer.  The product is
// evaluated transposed with v_mfma_f32_16x16x32_bf16, D = W-piece (A: 16 output columns x 32 k) x X-piece (B: 32 k x 16 batch rows), so
// an accumulator tile has the batch row on the lane and 4 consecutive output columns in its registers: register r of lane (row, group
// g) is column 4*g + r of the tile.  That IS a valid B-operand layout for the next layer when the next layer's k index is enumerated
// the same way (k-step s = registers 0..3 of input tiles 2s and 2s+1): the packed weight images are written in that k order, so bias +
// ReLU (+ the inference batch-norm affine) turn the <= 26 accumulator tiles into the next layer's input IN PLACE -- no LDS round trip,
// no cross-lane traffic, no HBM.  Layer 1 reads X straight into the same register layout (one 16-byte piece per 16-column tile and
// lane), so all layers run one loop body.  Per k-step of 32 a lane splits 8 fp32 values into three bf16x8 operands (52 VALU
// instructions in front of 150 MFMAs).  Registers: 104 (input) + 104 (accumulators) of the 256 a wave has at two waves per SIMD.
// W streams through LDS: a k-step's image for 13 column tiles (3 pieces x 13 x 1 KB) per stage, double-buffered, filled by
// global_load_lds from a 1 MB-per-layer image that stays in L2 (every workgroup re-reads all of it per tile: 128 rows per workgroup
// keep that at ~40 GB/s per CU); one barrier per stage of 78 MFMAs per wave.  The k loop is fully unrolled (register arrays need
// static indices): 13 k-steps x 26 tiles x 6 MFMAs of straight-line code, one copy shared by all layers.
//
// The units = 1 head (deepFM.py:311-317) is a dot product of the last activation with the weight row: in registers, two shuffles.
#include "common.hpp"

namespace dir {

typedef float tw_f32x4 __attribute__((ext_vector_type(4)));
typedef float tw_f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 tw_bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned int tw_u32x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) void* tw_lds_ptr;
typedef const __attribute__((address_space(1))) void* tw_glb_ptr;

// Round 6: RT row tiles of 16 per wave.  RT = 1: eight waves of 16 rows (two per SIMD, <= 256 registers each); RT = 2: FOUR waves of 32 rows
// (one per SIMD, the whole 512-register file): a weight piece read from LDS feeds the matrix instructions of two row tiles, so the
// workgroup reads HALF the LDS bytes per row (VERDICT r5 item 3's experiment; measured SLOWER, see tower_launch: kept as an A/B switch).  A
// workgroup is 128 rows either way.
constexpr int TW_ROWS = 128;      // rows per workgroup
constexpr int TW_NT = 26;         // column tiles of 16: widths up to 416
constexpr int TW_ST = 13;         // column tiles per stage (one LDS buffer: 3 pieces x 13 KB)
constexpr int TW_MAXL = 4;
constexpr int TW_BUFB = 3 * TW_ST * 1024;
constexpr int tw_bufb(int np) { return np * TW_ST * 1024; }      // bytes of one stage: np pieces x 13 KB (np = 3: bf16 x 3, np = 2: fp16 x 2)

struct TowerParams {
    const float* X;
    int64_t x_ld, M;
    int Kd, L;
    int N[TW_MAXL];
    const unsigned char* img[TW_MAXL];
    const float* bias[TW_MAXL];
    const float* scale[TW_MAXL];
    const float* shift[TW_MAXL];
    int relu[TW_MAXL];
    const float* head_w;
    const float* head_b;
    const float* add0;
    const float* add1;
    float* out;
    int64_t out_ld;
    // GATHER variant (dir_deepfm_tower_bf16x3_f32): layer 1's input row is the concatenation of F packed serving rows looked up here
    // (slot f's table [vocab_f, row_ld], the K = 16 embedding floats first, the first-order weight at column lin_col); X / x_ld unused
    const float* const* tables;
    const int64_t* vocab;
    const int64_t* ids;
    int64_t ids_sb, ids_sf, row_ld;
    int F, lin_col, want_fm;
    const float* lin_bias;
};

__device__ __forceinline__ unsigned int tw_pk(float a, float b) {      // v_cvt_pk_bf16_f32 (round to nearest even), a in the low half
    typedef __bf16 pk2_t __attribute__((ext_vector_type(2)));
    const pk2_t v = {(__bf16)a, (__bf16)b};
    unsigned int w = __builtin_bit_cast(unsigned int, v);
    asm("" : "+v"(w));      // hides where w came from (see dense_bf3.hip: db3_pk); the convert stays a compiler-generated instruction
    return w;
}
__device__ __forceinline__ void tw_split_pair(float a, float b, unsigned int& w0, unsigned int& w1, unsigned int& w2) {
    w0 = tw_pk(a, b);
    const float ra = a - __builtin_bit_cast(float, w0 << 16), rb = b - __builtin_bit_cast(float, w0 & 0xffff0000u);
    w1 = tw_pk(ra, rb);
    const float sa = ra - __builtin_bit_cast(float, w1 << 16), sb = rb - __builtin_bit_cast(float, w1 & 0xffff0000u);
    w2 = tw_pk(sa, sb);
}

// Round 4: "fp16 x 2" (NP = 2; csrc/cin_bf3.hip explains the arithmetic and its preconditions): two fp16 pieces per operand, three products
// on v_mfma_f32_16x16x32_f16.  For this kernel it is above all LDS traffic: a stage is 2 x 13 KB instead of 3 x 13 KB and every wave
// reads two 1 KB W pieces per tile instead of three (the kernel was LDS-read bound at ~70 % LDS busy), next to half the matrix
// instructions and 6 instead of 11 split instructions per operand pair.
typedef _Float16 tw_f16x8 __attribute__((ext_vector_type(8)));
__device__ __forceinline__ unsigned int tw_pk_h(float a, float b) {     // v_cvt_pk_f16_f32 (round to nearest even), a in the low half
    typedef _Float16 h2_t __attribute__((ext_vector_type(2)));
    const h2_t v = {(_Float16)a, (_Float16)b};
    unsigned int w = __builtin_bit_cast(unsigned int, v);
    asm("" : "+v"(w));
    return w;
}
template <int NP> struct TwPc;
template <> struct TwPc<3> {
    using op_t = tw_bf16x8;
    __device__ static __forceinline__ void split(float a, float b, unsigned int (&w)[3]) { tw_split_pair(a, b, w[0], w[1], w[2]); }
};
template <> struct TwPc<2> {
    using op_t = tw_f16x8;
    __device__ static __forceinline__ void split(float a, float b, unsigned int (&w)[2]) {
        typedef _Float16 h2_t __attribute__((ext_vector_type(2)));
        w[0] = tw_pk_h(a, b);
        const h2_t h = __builtin_bit_cast(h2_t, w[0]);
        w[1] = tw_pk_h(a - (float)h[0], b - (float)h[1]);
    }
};

// input column (k index) of element j of lane group g in k-step ks: the accumulator-register enumeration of the header comment
__host__ __device__ inline int tw_kmap(int ks, int g, int j) { return 32 * ks + 16 * (j >> 2) + 4 * g + (j & 3); }

// W [N, K] fp32 (row stride w_ld) -> image [k-step][stage][piece][13 tile slots][lane][8 e] bf16 (stage st = column tiles
// [13*st, 13*st + 13)): element e of lane l of tile ct in k-step ks = piece of
// W[n = 16*ct + (l & 15)][k = tw_kmap(ks, l >> 4, e)]; zero where n >= N or k >= K.
template <int NP>
__global__ __launch_bounds__(256) void tower_bf3_pack_k(const float* __restrict__ W, int64_t w_ld, int K, int N, int nks, int nct,
                                                        unsigned int* __restrict__ img) {
    const int64_t total = (int64_t)nks * nct * 64 * 4;            // one thread per pair of e
    for (int64_t e_ = (int64_t)blockIdx.x * 256 + threadIdx.x; e_ < total; e_ += (int64_t)gridDim.x * 256) {
        int64_t q = e_;
        const int ep = (int)(q & 3); q >>= 2;
        const int l = (int)(q & 63); q >>= 6;
        const int ct = (int)(q % nct);
        const int ks = (int)(q / nct);
        const int n = 16 * ct + (l & 15);
        const int k = tw_kmap(ks, l >> 4, 2 * ep);               // e = 2*ep and 2*ep + 1 are neighbours in k
        const float v0 = (n < N && k < K) ? W[(int64_t)n * w_ld + k] : 0.f;
        const float v1 = (n < N && k + 1 < K) ? W[(int64_t)n * w_ld + k + 1] : 0.f;
        unsigned int pw[NP];
        TwPc<NP>::split(v0, v1, pw);
        const int st = ct / TW_ST, cs = ct - st * TW_ST;
        const int nstg = (nct + TW_ST - 1) / TW_ST;
        // dwords: every stage is a full NP x 13 KB block (slots behind the layer's last tile are never loaded): constant piece stride
        const int64_t base = ((int64_t)ks * nstg + st) * (NP * TW_ST * 256) + (cs * 64 + l) * 4 + ep;
#pragma unroll
        for (int q = 0; q < NP; ++q) img[base + q * TW_ST * 256] = pw[q];
    }
}

// GATHER: DeepFM inference in one launch.  The lane that would load X[row][16 ct + 4 g .. + 3] loads the same four floats from slot ct's
// table row of sample `row` instead (an id outside [0, vocab) reads as a zero row, as in gather_packed_rows_k), and the lookups' two other
// consumers ride along in the order the gather kernel uses, so the logit is bit for bit the two-launch path's: the FM second-order term
// 0.5 sum_k ((sum_f e)^2 - sum_f e^2) (field sums f-ascending, then k-ascending across the row's four lanes) and the first-order term
// sum_f w_f + bias (f-ascending), both added to the head's logit in the epilogue.  The 109 MB concat is never written or read.
template <bool GATHER, int NP = 3, int RT = 1>
__global__ __launch_bounds__(512 / RT, RT == 1 ? 2 : 1) void tower_bf3_k(const TowerParams p) {
    constexpr int TW_NW = 8 / RT;             // waves per workgroup
    constexpr int BUFB = tw_bufb(NP);
    extern __shared__ __attribute__((aligned(16))) unsigned char tw_smem[];      // [2][TW_BUFB]: the W image of one stage
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int wave_u = __builtin_amdgcn_readfirstlane(wave);      // provably wave-uniform: scalar loop control around the LDS-DMA issue
    const int r16 = lane & 15;
    const int g = lane >> 4;
    const int64_t ntiles = (p.M + TW_ROWS - 1) / TW_ROWS;
    if ((int64_t)blockIdx.x >= ntiles) return;

    // A stage = the W image of (layer l, k-step ks, column tiles [13*st, 13*st + 13)): 3 pieces x nst slots of 1 KB, lane-linear, in a fixed
    // [piece][13 slots] layout.  Wave w brings slots w and w + 8 of every piece (up to six LDS-DMA instructions per stage).
    auto stage = [&](int l, int ks, int st, int buf) {
#ifdef TW_ABL_DMA          // timing ablation (WRONG results): the W image is streamed into LDS once per row tile only
        if (l | ks | st) return;
#endif
        const int nct = (p.N[l] + 15) >> 4;
        const int nstg = nct > TW_ST ? 2 : 1;
        const int nst = nct - st * TW_ST < TW_ST ? nct - st * TW_ST : TW_ST;
        const unsigned char* src = p.img[l] + ((int64_t)ks * nstg + st) * BUFB + lane * 16;
        unsigned char* dst = tw_smem + buf * BUFB;
#pragma unroll
        for (int pc = 0; pc < NP; ++pc)
            for (int cs = wave_u; cs < nst; cs += TW_NW) {        // wave-uniform trip count
                const int off = (pc * TW_ST + cs) * 1024;
                __builtin_amdgcn_global_load_lds((tw_glb_ptr)(src + off), (tw_lds_ptr)(dst + off), 16, 0, 0);
            }
    };

    // GATHER: every slot's vocabulary bound and table base, once per workgroup, behind the two W stage buffers
    uint64_t* const slot_info = reinterpret_cast<uint64_t*>(tw_smem + 2 * BUFB);
    if constexpr (GATHER) {
        if (tid < p.F) {
            slot_info[2 * tid] = p.vocab ? (uint64_t)p.vocab[tid] : (uint64_t)1 << 63;
            slot_info[2 * tid + 1] = reinterpret_cast<uint64_t>(p.tables[tid]);
        }
        __syncthreads();
    }

    tw_f32x4 act[RT][TW_NT], acc[RT][TW_NT];
    int buf = 0;
    stage(0, 0, 0, 0);
    for (int64_t t = blockIdx.x; t < ntiles; t += gridDim.x) {
        int64_t rrow[RT];
        float fm_r[RT], lin_r[RT];                     // GATHER: the rows' FM and first-order terms, both valid in lane group g == 0
#pragma unroll
        for (int rt = 0; rt < RT; ++rt) {
        const int64_t r = t * TW_ROWS + (wave * RT + rt) * 16 + r16;
        rrow[rt] = r;
        fm_r[rt] = 0.f;
        lin_r[rt] = 0.f;
        if constexpr (GATHER) {
            const int64_t rr = r < p.M ? r : p.M - 1;
            const int64_t* idp = p.ids + rr * p.ids_sb;
            tw_f32x4 sum = {0.f, 0.f, 0.f, 0.f}, sq = sum;
            float lin = 0.f;
            // Every slot's id first, then every row read (unconditional, from a clamped address), then the sums: written as one loop with the
            // loads under their guards, each slot cost a chain of waited-for round trips (id -> bound -> table pointer -> row: 60 x
            // s_waitcnt vmcnt(0) in the prologue's machine code, profiles/NOTES.md R6.11); the arithmetic and its order are unchanged.
            int64_t idv[TW_NT];
#pragma unroll
            for (int ct = 0; ct < TW_NT; ++ct) idv[ct] = idp[(int64_t)(ct < p.F ? ct : 0) * p.ids_sf];
            float lwv[TW_NT];
#pragma unroll
            for (int ct = 0; ct < TW_NT; ++ct) {
                const int cc = ct < p.F ? ct : 0;
                // the slot's bound and table base from LDS (staged once per workgroup: as global loads of p.vocab[cc] / p.tables[cc] each was a
                // waited-for round trip in front of the slot's row read)
                const uint64_t bound = slot_info[2 * cc];
                const bool ok = ct < p.F && (uint64_t)idv[ct] < bound;
                const float* t = reinterpret_cast<const float*>(slot_info[2 * cc + 1]) + (ok ? idv[ct] : 0) * p.row_ld;
                act[rt][ct] = *reinterpret_cast<const tw_f32x4*>(t + 4 * g);
                lwv[ct] = p.lin_col >= 0 ? t[g == 0 ? p.lin_col : 0] : 0.f;
                idv[ct] = ok ? 1 : 0;                  // (from here on: the slot's validity)
            }
#pragma unroll
            for (int ct = 0; ct < TW_NT; ++ct) {
                tw_f32x4 v = act[rt][ct];
                float lw = g == 0 ? lwv[ct] : 0.f;
                if (!idv[ct]) {
                    v = (tw_f32x4){0.f, 0.f, 0.f, 0.f};
                    lw = 0.f;
                }
                act[rt][ct] = v;
                sum += v;                              // f-ascending fp32 sums, as gather_packed_rows_k
                sq += v * v;
                lin = lin + lw;
            }
            // 0.5 * sum_k (sum^2 - sq), k ascending through the row's four lanes (g = 0..3 hold k = 4 g .. 4 g + 3): fm_tail<4>'s chain
            if (p.want_fm) {                                       // (kernel-uniform)
                const tw_f32x4 d = sum * sum - sq;
                float acc_fm = 0.f;
#pragma unroll
                for (int cc = 0; cc < 4; ++cc) {
                    const float carry = __shfl(acc_fm, r16 + 16 * (cc > 0 ? cc - 1 : 0), 64);
                    if (g == cc) acc_fm = ((((cc == 0 ? 0.f : carry) + d[0]) + d[1]) + d[2]) + d[3];
                }
                fm_r[rt] = __shfl(0.5f * acc_fm, r16 + 48, 64);
            }
            lin_r[rt] = lin + (p.lin_bias ? p.lin_bias[0] : 0.f);
        } else {   // X -> act, accumulator layout: register e of tile ct = X[row][16*ct + 4*g + e] (Kd % 4 == 0: a piece is in or out)
            const float* xr = p.X + (r < p.M ? r : p.M - 1) * p.x_ld + 4 * g;
#pragma unroll
            for (int ct = 0; ct < TW_NT; ++ct)
                act[rt][ct] = 16 * ct + 4 * g < p.Kd ? *reinterpret_cast<const tw_f32x4*>(xr + 16 * ct) : (tw_f32x4){0.f, 0.f, 0.f, 0.f};
        }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");          // the first stage's W pieces (issued before the loop / by the previous tile)
        __syncthreads();

        for (int l = 0; l < p.L; ++l) {
            const int K = l ? p.N[l - 1] : p.Kd;
            const int N = p.N[l];
            const int nks = (K + 31) >> 5, nct = (N + 15) >> 4;
            const int nstg = nct > TW_ST ? 2 : 1;
            const bool last = l + 1 == p.L;
#pragma unroll
            for (int rt = 0; rt < RT; ++rt)
#pragma unroll
                for (int ct = 0; ct < TW_NT; ++ct) acc[rt][ct] = (tw_f32x4){0.f, 0.f, 0.f, 0.f};

#pragma unroll
            for (int ks = 0; ks < TW_NT / 2; ++ks) {
                if (ks < nks) {                                   // workgroup-uniform
                    // this k-step's B operands: the 4 registers of input tiles 2ks and 2ks+1, split three ways
                    using op_t = typename TwPc<NP>::op_t;
                    op_t xa[RT][NP];
#pragma unroll
                    for (int rt = 0; rt < RT; ++rt) {
                        const tw_f32x4 a0 = act[rt][2 * ks], a1 = act[rt][2 * ks + 1];
                        const float av[8] = {a0[0], a0[1], a0[2], a0[3], a1[0], a1[1], a1[2], a1[3]};
                        unsigned int w[NP][4];
#pragma unroll
                        for (int pr = 0; pr < 4; ++pr) {
                            unsigned int pw[NP];
                            TwPc<NP>::split(av[2 * pr], av[2 * pr + 1], pw);
#pragma unroll
                            for (int pc = 0; pc < NP; ++pc) w[pc][pr] = pw[pc];
                        }
#pragma unroll
                        for (int pc = 0; pc < NP; ++pc) xa[rt][pc] = __builtin_bit_cast(op_t, (tw_u32x4){w[pc][0], w[pc][1], w[pc][2], w[pc][3]});
                    }
#pragma unroll
                    for (int st = 0; st < 2; ++st) {
                        if (st < nstg) {                          // workgroup-uniform
                            // the next stage: of this k-step, of the next one, of the next layer, or the first of this workgroup's next tile
                            if (st + 1 < nstg) stage(l, ks, st + 1, buf ^ 1);
                            else if (ks + 1 < nks) stage(l, ks + 1, 0, buf ^ 1);
                            else if (!last) stage(l + 1, 0, 0, buf ^ 1);
                            else if (t + gridDim.x < ntiles) stage(0, 0, 0, buf ^ 1);
                            // ALL 13 tile slots of the stage, no branch per tile (a slot behind the layer's last tile holds stale, finite W
                            // pieces; what it adds up in acc[ct >= nct] is discarded by the epilogue).
                            // The W operands are read by hand one tile ahead into two register sets used alternately: left to itself the
                            // (register-starved) compiler reads a tile's three pieces, waits, issues its six MFMAs, and only then reads the
                            // next tile's -- the LDS latency of every tile exposed (411 -> 337 us for three 400-wide layers).
                            // ds_read_b128 results return in order, so lgkmcnt(3) = "this tile's three pieces are here, the next tile's may
                            // still be in flight"; an outstanding scalar load can only make that wait longer, never shorter.
                            // sched_barrier: nothing may be moved across the wait (cdna_hip_programming.md 5.4 rule 18).
                            const unsigned int wl = (unsigned int)(size_t)(tw_smem + buf * BUFB + lane * 16);
                            tw_u32x4 wq[2][NP];
#ifdef TW_ABL_LDS          // timing ablation (WRONG results): the W operands are whatever the registers hold; no LDS read, nothing to wait for
#define TW_DS_READ(dst, off) asm volatile("" : "=v"(dst) : "v"(wl), "n"(off))
#else
#define TW_DS_READ(dst, off) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst) : "v"(wl), "n"(off))
#endif
                            if constexpr (NP == 2) {
                                // the same hand pipelining with two pieces per tile: lgkmcnt(2) = "this tile's two pieces are here"
                                TW_DS_READ(wq[0][0], 0);
                                TW_DS_READ(wq[0][1], TW_ST * 1024);
#pragma unroll
                                for (int cs = 0; cs < TW_ST; ++cs) {
                                    if (cs + 1 < TW_ST) {
                                        TW_DS_READ(wq[(cs + 1) & 1][0], (cs + 1) * 1024);
                                        TW_DS_READ(wq[(cs + 1) & 1][1], (TW_ST + cs + 1) * 1024);
                                        asm volatile("s_waitcnt lgkmcnt(2)" : "+v"(wq[cs & 1][0]), "+v"(wq[cs & 1][1]));
                                    } else {
                                        asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(wq[cs & 1][0]), "+v"(wq[cs & 1][1]));
                                    }
                                    __builtin_amdgcn_sched_barrier(0);
                                    const tw_f16x8 w0 = __builtin_bit_cast(tw_f16x8, wq[cs & 1][0]);
                                    const tw_f16x8 w1 = __builtin_bit_cast(tw_f16x8, wq[cs & 1][1]);
#pragma unroll
                                    for (int rt = 0; rt < RT; ++rt) {
                                        tw_f32x4 tt = acc[rt][st * TW_ST + cs];
                                        tt = __builtin_amdgcn_mfma_f32_16x16x32_f16(w1, xa[rt][0], tt, 0, 0, 0);      // the three products, smallest first
                                        tt = __builtin_amdgcn_mfma_f32_16x16x32_f16(w0, xa[rt][1], tt, 0, 0, 0);
                                        tt = __builtin_amdgcn_mfma_f32_16x16x32_f16(w0, xa[rt][0], tt, 0, 0, 0);
                                        acc[rt][st * TW_ST + cs] = tt;
                                    }
                                    __builtin_amdgcn_sched_barrier(0);
                                }
                            } else {
                            TW_DS_READ(wq[0][0], 0);
                            TW_DS_READ(wq[0][1], TW_ST * 1024);
                            TW_DS_READ(wq[0][2], 2 * TW_ST * 1024);
#pragma unroll
                            for (int cs = 0; cs < TW_ST; ++cs) {
                                if (cs + 1 < TW_ST) {
                                    TW_DS_READ(wq[(cs + 1) & 1][0], (cs + 1) * 1024);
                                    TW_DS_READ(wq[(cs + 1) & 1][1], (TW_ST + cs + 1) * 1024);
                                    TW_DS_READ(wq[(cs + 1) & 1][2], (2 * TW_ST + cs + 1) * 1024);
                                    asm volatile("s_waitcnt lgkmcnt(3)" : "+v"(wq[cs & 1][0]), "+v"(wq[cs & 1][1]), "+v"(wq[cs & 1][2]));
                                } else {
                                    asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(wq[cs & 1][0]), "+v"(wq[cs & 1][1]), "+v"(wq[cs & 1][2]));
                                }
                                __builtin_amdgcn_sched_barrier(0);
                                const tw_bf16x8 w0 = __builtin_bit_cast(tw_bf16x8, wq[cs & 1][0]);
                                const tw_bf16x8 w1 = __builtin_bit_cast(tw_bf16x8, wq[cs & 1][1]);
                                const tw_bf16x8 w2 = __builtin_bit_cast(tw_bf16x8, wq[cs & 1][2]);
#pragma unroll
                                for (int rt = 0; rt < RT; ++rt) {
                                    tw_f32x4 tt = acc[rt][st * TW_ST + cs];
                                    tt = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w0, xa[rt][2], tt, 0, 0, 0);
                                    tt = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w2, xa[rt][0], tt, 0, 0, 0);
                                    tt = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w1, xa[rt][1], tt, 0, 0, 0);
                                    tt = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w0, xa[rt][1], tt, 0, 0, 0);
                                    tt = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w1, xa[rt][0], tt, 0, 0, 0);
                                    tt = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w0, xa[rt][0], tt, 0, 0, 0);
                                    acc[rt][st * TW_ST + cs] = tt;
                                }
                                __builtin_amdgcn_sched_barrier(0);
                            }
                            }
#undef TW_DS_READ
                            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // this wave's pieces of the next stage have landed
#ifndef TW_ABL_BAR         // timing ablation (WRONG results): no barrier between the stages
                            __syncthreads();                                    // ... everyone's have, and everyone is done reading this stage
#endif
                            buf ^= 1;
                        }
                    }
                }
            }

            // ---- epilogue: bias, activation, inference batch-norm affine; the result is the next layer's input, in place
            const float* bias = p.bias[l];
            const float* sc = p.scale[l];
            const float* sh = p.shift[l];
            const int relu = p.relu[l];
#pragma unroll
            for (int rt = 0; rt < RT; ++rt) {
            const int64_t r = rrow[rt];
            float part = 0.f;                                     // head: this lane's share of the logit
            // Three passes over the 26 tiles, each with its per-column vectors requested in two batches of 13 (unconditional loads from clamped
            // addresses; a vector that does not exist reads the weight image and is not used): written as one loop with each load under its
            // guard, every tile waited for its bias (and scale, shift, head weight) in turn -- 26 to 104 L2 round trips per layer and row tile,
            // 144 x s_waitcnt vmcnt(0) in the kernel (profiles/NOTES.md R6.12).  Per element the operations and their order are unchanged.
            const float* dummy = reinterpret_cast<const float*>(p.img[l]);
            constexpr int HB = TW_NT / 2;
            auto vec13 = [&](const float* src, int half, tw_f32x4 (&o)[HB]) {
#pragma unroll
                for (int q = 0; q < HB; ++q) {
                    const int col = 16 * (half * HB + q) + 4 * g;
                    o[q] = *reinterpret_cast<const tw_f32x4*>((src ? src : dummy) + (col < N ? col : 0));
                }
            };
#pragma unroll
            for (int half = 0; half < 2; ++half) {               // bias + activation
                tw_f32x4 b4[HB];
                vec13(bias, half, b4);
#pragma unroll
                for (int q = 0; q < HB; ++q) {
                    const int ct = half * HB + q, col = 16 * ct + 4 * g;      // N % 4 == 0: the lane's four columns are inside or outside together
                    tw_f32x4 v = acc[rt][ct];
                    if (col < N) {
                        if (bias) v += b4[q];
                        if (relu) {
#pragma unroll
                            for (int e = 0; e < 4; ++e) v[e] = v[e] > 0.f ? v[e] : 0.f;
                        }
                    } else {
                        v = (tw_f32x4){0.f, 0.f, 0.f, 0.f};
                    }
                    acc[rt][ct] = v;
                }
            }
            if (sc) {                                            // (kernel-uniform) the inference batch-norm affine: multiply then add, unfused (dense.hip's affine epilogue)
#pragma unroll
                for (int half = 0; half < 2; ++half) {
                    tw_f32x4 s4[HB], h4[HB];
                    vec13(sc, half, s4);
                    vec13(sh, half, h4);
#pragma unroll
                    for (int q = 0; q < HB; ++q) {
                        const int ct = half * HB + q, col = 16 * ct + 4 * g;
                        if (col < N) {
                            tw_f32x4 v = acc[rt][ct];
#pragma unroll
                            for (int e = 0; e < 4; ++e) v[e] = v[e] * s4[q][e] + h4[q][e];
                            acc[rt][ct] = v;
                        }
                    }
                }
            }
            if (last && p.head_w) {
#pragma unroll
                for (int half = 0; half < 2; ++half) {
                    tw_f32x4 w4[HB];
                    vec13(p.head_w, half, w4);
#pragma unroll
                    for (int q = 0; q < HB; ++q) {
                        const int ct = half * HB + q, col = 16 * ct + 4 * g;
                        if (col < N) {
                            const tw_f32x4 v = acc[rt][ct];
                            part += v[0] * w4[q][0];
                            part += v[1] * w4[q][1];
                            part += v[2] * w4[q][2];
                            part += v[3] * w4[q][3];
                        }
                    }
                }
            } else if (last && r < p.M) {
#pragma unroll
                for (int ct = 0; ct < TW_NT; ++ct) {
                    const int col = 16 * ct + 4 * g;
                    if (col < N) *reinterpret_cast<tw_f32x4*>(p.out + r * p.out_ld + col) = acc[rt][ct];
                }
            }
#pragma unroll
            for (int ct = 0; ct < TW_NT; ++ct) act[rt][ct] = acc[rt][ct];
            if (last && p.head_w) {
                part += __shfl_xor(part, 16, 64);                 // the row's other columns live in the other three lane groups
                part += __shfl_xor(part, 32, 64);
                if (g == 0 && r < p.M) {
                    float o = part + p.head_b[0];
                    if constexpr (GATHER) {
                        if (p.want_fm) o += fm_r[rt];             // the order of ops.tower(..., adds=(fm, lin))
                        if (p.lin_col >= 0) o += lin_r[rt];
                    }
                    if (p.add0) o += p.add0[r];
                    if (p.add1) o += p.add1[r];
                    p.out[r * p.out_ld] = o;
                }
            }
            }
        }
    }
}

}  // namespace dir

using namespace dir;

extern "C" int64_t dir_tower_bf16x3_image_bytes(int K, int N) {
    if (K <= 0 || N <= 0) return 0;
    return (int64_t)((K + 31) / 32) * (((N + 15) / 16 + TW_ST - 1) / TW_ST) * TW_BUFB;
}

static int tower_pack(const char* name, int pieces, const float* W, int64_t w_ld, int K, int N, void* image, int64_t image_bytes, dir_stream_t stream) {
    DIR_CHECK_ARG(W && image && K > 0 && N > 0 && w_ld >= K, "%s: bad argument (K=%d N=%d w_ld=%lld)", name, K, N, (long long)w_ld);
    DIR_CHECK_ARG(K <= 16 * TW_NT && N <= 16 * TW_NT, "%s: K=%d N=%d exceed %d", name, K, N, 16 * TW_NT);
    DIR_CHECK_ARG(aligned16(image) && image_bytes >= dir_tower_bf16x3_image_bytes(K, N), "%s: image must be 16-byte aligned and hold "
                  "dir_tower_bf16x3_image_bytes(K, N) bytes", name);
    const int nks = (K + 31) / 32, nct = (N + 15) / 16;
    const int64_t threads = (int64_t)nks * nct * 64 * 4;
    if (pieces == 2)
        hipLaunchKernelGGL(tower_bf3_pack_k<2>, dim3((unsigned)((threads + 255) / 256)), dim3(256), 0, as_stream(stream), W, w_ld, K, N, nks, nct,
                           static_cast<unsigned int*>(image));
    else
        hipLaunchKernelGGL(tower_bf3_pack_k<3>, dim3((unsigned)((threads + 255) / 256)), dim3(256), 0, as_stream(stream), W, w_ld, K, N, nks, nct,
                           static_cast<unsigned int*>(image));
    DIR_CHECK_LAUNCH(name);
    return DIR_OK;
}

extern "C" int dir_tower_bf16x3_pack_f32(const float* W, int64_t w_ld, int K, int N, void* image, int64_t image_bytes, dir_stream_t stream) {
    return tower_pack("dir_tower_bf16x3_pack_f32", 3, W, w_ld, K, N, image, image_bytes, stream);
}

// the fp16 x 2 image of the same weight (two pieces per stage; dir_tower_bf16x3_image_bytes(K, N) bytes hold it): for dir_tower_f16x2_f32 /
// dir_deepfm_tower_f16x2_f32 ONLY -- the two arithmetics' images are not interchangeable
extern "C" int dir_tower_f16x2_pack_f32(const float* W, int64_t w_ld, int K, int N, void* image, int64_t image_bytes, dir_stream_t stream) {
    return tower_pack("dir_tower_f16x2_pack_f32", 2, W, w_ld, K, N, image, image_bytes, stream);
}

// the layers / head part of the argument checks, shared by the two entries
static int tower_fill(const char* name, TowerParams& p, int Kd, int L, const int* N, const void* const* images, const float* const* bias,
                      const float* const* post_scale, const float* const* post_shift, const int* act, const float* head_w, const float* head_b,
                      const float* add0, const float* add1, float* out, int64_t out_ld) {
    DIR_CHECK_ARG(Kd > 0 && L >= 1 && L <= TW_MAXL && N && images && act, "%s: bad shape (Kd=%d L=%d)", name, Kd, L);
    if (Kd > 16 * TW_NT || (Kd & 3)) return fail(DIR_E_UNSUPPORTED, "%s: Kd=%d (a multiple of 4, <= %d)", name, Kd, 16 * TW_NT);
    DIR_CHECK_ARG((head_w == nullptr) == (head_b == nullptr), "%s: head_w and head_b come together", name);
    DIR_CHECK_ARG(head_w || (!add0 && !add1), "%s: add0 / add1 are addends of the head's logit", name);
    p.Kd = Kd; p.L = L;
    for (int l = 0; l < TW_MAXL; ++l) {
        p.N[l] = 0; p.img[l] = nullptr; p.bias[l] = p.scale[l] = p.shift[l] = nullptr; p.relu[l] = 0;
    }
    for (int l = 0; l < L; ++l) {
        if (N[l] <= 0 || N[l] > 16 * TW_NT || (N[l] & 3)) return fail(DIR_E_UNSUPPORTED, "%s: layer %d width %d (a multiple of 4, <= %d)", name, l, N[l], 16 * TW_NT);
        DIR_CHECK_ARG(act[l] == DIR_ACT_NONE || act[l] == DIR_ACT_RELU, "%s: act[%d]=%d", name, l, act[l]);
        DIR_CHECK_ARG(images[l] && aligned16(images[l]), "%s: image %d", name, l);
        const float* sc = post_scale ? post_scale[l] : nullptr;
        const float* sh = post_shift ? post_shift[l] : nullptr;
        DIR_CHECK_ARG((sc == nullptr) == (sh == nullptr), "%s: post_scale and post_shift come together (layer %d)", name, l);
        const float* b = bias ? bias[l] : nullptr;
        if ((b && !aligned16(b)) || (sc && (!aligned16(sc) || !aligned16(sh)))) return fail(DIR_E_UNSUPPORTED, "%s: bias / affine vectors must be 16-byte aligned", name);
        p.N[l] = N[l]; p.img[l] = static_cast<const unsigned char*>(images[l]); p.bias[l] = b; p.scale[l] = sc; p.shift[l] = sh;
        p.relu[l] = act[l] == DIR_ACT_RELU;
    }
    if (head_w) {
        if (!aligned16(head_w) || out_ld < 1) return fail(DIR_E_UNSUPPORTED, "%s: head_w must be 16-byte aligned, out_ld >= 1", name);
    } else if ((out_ld & 3) || out_ld < N[L - 1] || !aligned16(out)) {
        return fail(DIR_E_UNSUPPORTED, "%s: out [M, N_last] needs out_ld %% 4 == 0 and a 16-byte aligned base", name);
    }
    p.head_w = head_w; p.head_b = head_b; p.add0 = add0; p.add1 = add1; p.out = out; p.out_ld = out_ld;
    p.X = nullptr; p.x_ld = 0;
    p.tables = nullptr; p.vocab = nullptr; p.ids = nullptr; p.ids_sb = p.ids_sf = p.row_ld = 0; p.F = 0; p.lin_col = -1; p.want_fm = 0; p.lin_bias = nullptr;
    return DIR_OK;
}

template <bool GATHER, int NP, int RT>
static int tower_launch_rt(const char* name, const TowerParams& p, dir_stream_t stream) {
    static LdsOnce once;
    if (!lds_limit(once, 160 * 1024, &tower_bf3_k<GATHER, NP, RT>)) return fail(DIR_E_HIP, "%s: cannot reserve 160 KiB of LDS", name);
    const int64_t ntiles = (p.M + TW_ROWS - 1) / TW_ROWS;
    const int64_t cap = kCUs;
    const int64_t nwg = ntiles < cap ? ntiles : cap;              // persistent workgroups: one per CU (8 waves x 256 or 4 x 512 registers), 78 / 52 KB of LDS
    hipLaunchKernelGGL((tower_bf3_k<GATHER, NP, RT>), dim3((unsigned)nwg), dim3(512 / RT), 2 * tw_bufb(NP) + (GATHER ? 16 * TW_NT : 0), as_stream(stream), p);
    return DIR_OK;
}
template <bool GATHER, int NP>
static int tower_launch(const char* name, const TowerParams& p, dir_stream_t stream) {
    // DIR_TOWER_RT = 1 | 2 (read per call: an A/B switch): row tiles per wave, see TW_ROWS.  Default 1: with RT = 2 the workgroup reads half the
    // LDS bytes, but ONE wave per SIMD hides nothing -- mlp_dense 0.240 -> 0.286 ms, deepfm_full 0.278 -> 0.371, pipe busy 0.38 -> 0.29, the
    // wave issuing a third of its cycles (profiles/NOTES.md R6.4): the stage's LDS reads (812 cycles per k-step and CU at fp16 x 2) were not
    // what bounds this kernel.
    const char* e = getenv("DIR_TOWER_RT");
    const int rt = e ? atoi(e) : 1;
    const int rc = rt == 2 ? tower_launch_rt<GATHER, NP, 2>(name, p, stream) : tower_launch_rt<GATHER, NP, 1>(name, p, stream);
    if (rc != DIR_OK) return rc;
    DIR_CHECK_LAUNCH(name);
    return DIR_OK;
}

static int tower_run(const char* name, int pieces, const float* X, int64_t x_ld, int64_t M, int Kd, int L, const int* N, const void* const* images,
                     const float* const* bias, const float* const* post_scale, const float* const* post_shift, const int* act,
                     const float* head_w, const float* head_b, const float* add0, const float* add1, float* out, int64_t out_ld,
                     dir_stream_t stream) {
    DIR_CHECK_ARG(M >= 0, "%s: M=%lld", name, (long long)M);
    if ((x_ld & 3) || x_ld < Kd) return fail(DIR_E_UNSUPPORTED, "%s: x_ld=%lld (a multiple of 4, >= Kd)", name, (long long)x_ld);
    TowerParams p;
    const int rc = tower_fill(name, p, Kd, L, N, images, bias, post_scale, post_shift, act, head_w, head_b, add0, add1, out, out_ld);
    if (rc != DIR_OK) return rc;
    if (M == 0) return DIR_OK;
    DIR_CHECK_ARG(X && out && aligned16(X), "%s: null or unaligned pointer", name);
    p.X = X; p.x_ld = x_ld; p.M = M;
    return pieces == 2 ? tower_launch<false, 2>(name, p, stream) : tower_launch<false, 3>(name, p, stream);
}

extern "C" int dir_tower_bf16x3_f32(const float* X, int64_t x_ld, int64_t M, int Kd, int L, const int* N, const void* const* images,
                                    const float* const* bias, const float* const* post_scale, const float* const* post_shift, const int* act,
                                    const float* head_w, const float* head_b, const float* add0, const float* add1, float* out, int64_t out_ld,
                                    dir_stream_t stream) {
    return tower_run("dir_tower_bf16x3_f32", 3, X, x_ld, M, Kd, L, N, images, bias, post_scale, post_shift, act, head_w, head_b, add0, add1, out, out_ld,
                     stream);
}

extern "C" int dir_tower_f16x2_f32(const float* X, int64_t x_ld, int64_t M, int Kd, int L, const int* N, const void* const* images,
                                   const float* const* bias, const float* const* post_scale, const float* const* post_shift, const int* act,
                                   const float* head_w, const float* head_b, const float* add0, const float* add1, float* out, int64_t out_ld,
                                   dir_stream_t stream) {
    return tower_run("dir_tower_f16x2_f32", 2, X, x_ld, M, Kd, L, N, images, bias, post_scale, post_shift, act, head_w, head_b, add0, add1, out, out_ld,
                     stream);
}

static int deepfm_tower_run(const char* name, int pieces, const float* const* tables, const int64_t* vocab, int F, int K, int64_t ld, int lin_col,
                            const int64_t* ids, int64_t stride_b, int64_t stride_f, int want_fm, int64_t M, const float* lin_bias, int L,
                            const int* N, const void* const* images, const float* const* bias, const float* const* post_scale,
                            const float* const* post_shift, const int* act, const float* head_w, const float* head_b,
                            const float* add0, const float* add1, float* out, int64_t out_ld, dir_stream_t stream) {
    DIR_CHECK_ARG(M >= 0 && F > 0, "%s: M=%lld F=%d", name, (long long)M, F);
    if (K != 16 || F > TW_NT) return fail(DIR_E_UNSUPPORTED, "%s: K=%d F=%d (K = 16, F <= %d: one column tile per slot)", name, K, F, TW_NT);
    DIR_CHECK_ARG(want_fm == 0 || want_fm == 1, "%s: want_fm=%d", name, want_fm);
    if (ld < K + (lin_col >= 0 ? 1 : 0) || (ld & 3) || lin_col >= ld) return fail(DIR_E_UNSUPPORTED, "%s: ld=%lld lin_col=%d", name, (long long)ld, lin_col);
    DIR_CHECK_ARG(head_w || (!want_fm && lin_col < 0), "%s: the FM and first-order terms are addends of the head's logit: head_w / head_b are required", name);
    TowerParams p;
    const int rc = tower_fill(name, p, F * K, L, N, images, bias, post_scale, post_shift, act, head_w, head_b, add0, add1, out, out_ld);
    if (rc != DIR_OK) return rc;
    if (M == 0) return DIR_OK;
    DIR_CHECK_ARG(tables && ids && out, "%s: null pointer", name);
    p.M = M; p.tables = tables; p.vocab = vocab; p.ids = ids; p.ids_sb = stride_b; p.ids_sf = stride_f; p.row_ld = ld; p.F = F; p.lin_col = lin_col;
    p.want_fm = want_fm; p.lin_bias = lin_bias;
    return pieces == 2 ? tower_launch<true, 2>(name, p, stream) : tower_launch<true, 3>(name, p, stream);
}

extern "C" int dir_deepfm_tower_bf16x3_f32(const float* const* tables, const int64_t* vocab, int F, int K, int64_t ld, int lin_col,
                                           const int64_t* ids, int64_t stride_b, int64_t stride_f, int want_fm, int64_t M, const float* lin_bias, int L,
                                           const int* N, const void* const* images, const float* const* bias, const float* const* post_scale,
                                           const float* const* post_shift, const int* act, const float* head_w, const float* head_b,
                                           const float* add0, const float* add1, float* out, int64_t out_ld, dir_stream_t stream) {
    return deepfm_tower_run("dir_deepfm_tower_bf16x3_f32", 3, tables, vocab, F, K, ld, lin_col, ids, stride_b, stride_f, want_fm, M, lin_bias, L, N,
                            images, bias, post_scale, post_shift, act, head_w, head_b, add0, add1, out, out_ld, stream);
}

extern "C" int dir_deepfm_tower_f16x2_f32(const float* const* tables, const int64_t* vocab, int F, int K, int64_t ld, int lin_col,
                                          const int64_t* ids, int64_t stride_b, int64_t stride_f, int want_fm, int64_t M, const float* lin_bias, int L,
                                          const int* N, const void* const* images, const float* const* bias, const float* const* post_scale,
                                          const float* const* post_shift, const int* act, const float* head_w, const float* head_b,
                                          const float* add0, const float* add1, float* out, int64_t out_ld, dir_stream_t stream) {
    return deepfm_tower_run("dir_deepfm_tower_f16x2_f32", 2, tables, vocab, F, K, ld, lin_col, ids, stride_b, stride_f, want_fm, M, lin_bias, L, N,
                            images, bias, post_scale, post_shift, act, head_w, head_b, add0, add1, out, out_ld, stream);
}
