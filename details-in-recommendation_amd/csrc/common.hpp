// common.hpp -- shared host/device helpers for libdir_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <cstring>
#include <cstdlib>
#include <stdint.h>
#include <stdio.h>
#include <stdarg.h>
#include <atomic>

#include "../../include/dir_hip.h"

// Cross-workgroup handshakes in this library ("the last workgroup to arrive finalizes": bucket_cap2_k, row_absmax_k, the status words of
// rs_pass_k's decoupled look-back) use RELAXED agent-scope atomics without release / acquire fences.  Under the HIP / LLVM memory model that
// is a data race; on gfx950 it is ordered by the hardware: every value a workgroup publishes before its arrival atomic is itself
// written by a device-scope atomic (or a store that has completed: s_waitcnt vmcnt(0) precedes the arrival), atomics execute at the L2 --
// the single point of coherence of an agent's memory on one XCD partition mode -- and a returning atomic has completed before the wave issues the next one.  A release fence
// (__threadfence(): buffer_wbl2) costs 56 us per 800 workgroups here (profiles/NOTES.md R4.2).  The guarantee is this target's, so nothing
// else may be compiled from these sources:
#if defined(__HIP_DEVICE_COMPILE__) && !defined(__gfx950__)
#error "details-in-recommendation_amd/csrc relies on gfx950's ordering of device-scope atomics (see the comment above); build with --offload-arch=gfx950 only"
#endif

namespace dir {

// thread-local error text behind dir_last_error()
char* err_buf();
int fail(int code, const char* fmt, ...);

inline hipStream_t as_stream(dir_stream_t s) { return reinterpret_cast<hipStream_t>(s); }

#define DIR_CHECK_ARG(cond, ...)                                   \
    do {                                                           \
        if (!(cond)) return ::dir::fail(DIR_E_BADARG, __VA_ARGS__); \
    } while (0)

#define DIR_CHECK_LAUNCH(name)                                                              \
    do {                                                                                    \
        hipError_t e__ = hipGetLastError();                                                 \
        if (e__ != hipSuccess)                                                              \
            return ::dir::fail(DIR_E_HIP, "%s: launch failed: %s", name, hipGetErrorString(e__)); \
    } while (0)

constexpr int kWave = 64;     // gfx950 wavefront
constexpr int kCUs = 256;     // MI355X compute units
constexpr int kXCDs = 8;

inline bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }

// grid size for a grid-stride kernel over `work_blocks` block-sized work items: enough blocks to fill
// every CU several times over, capped so that launch cost stays flat.
inline int grid_for(int64_t work_blocks, int blocks_per_cu = 8) {
    int64_t cap = (int64_t)kCUs * blocks_per_cu;
    int64_t g = work_blocks < cap ? work_blocks : cap;
    return g < 1 ? 1 : (int)g;
}

// Number of 256-thread workgroups of `kernel` that are co-resident on the chip (occupancy x CUs), cached
// per kernel.  A grid-stride kernel launched with exactly this many workgroups has no partial last round.
template <typename K>
inline int resident_blocks(K kernel, size_t shmem = 0, int block = 256) {
    static thread_local struct { const void* k; size_t sh; int n; } cache[64];
    static thread_local int used = 0;
    const void* key = reinterpret_cast<const void*>(kernel);
    for (int i = 0; i < used; ++i)
        if (cache[i].k == key && cache[i].sh == shmem) return cache[i].n;
    int per_cu = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kernel, block, shmem) != hipSuccess || per_cu < 1) per_cu = 1;
    if (per_cu > 8) per_cu = 8;
    const int n = per_cu * kCUs;
    if (used < 64) cache[used++] = {key, shmem, n};
    return n;
}

#if defined(__HIPCC__)
// The maximum (as a bit pattern: non-negative floats order like their bits) over ALL workgroups of a launch of 256-thread blocks, without a
// zeroing launch and without a same-address atomic per row: every workgroup leaves its maximum in block_bits[blockIdx.x] and takes a
// ticket; the LAST one to arrive reduces the block maxima, writes *all_bits and puts the ticket back to zero (the caller's word of
// persistent, initially zero memory; block_bits holds gridDim.x words).  Called by all threads of the block; wmx: the thread's maximum.
__device__ __forceinline__ void grid_max_bits(float wmx, unsigned int* __restrict__ all_bits, unsigned int* __restrict__ block_bits,
                                              unsigned int* __restrict__ ticket) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) wmx = fmaxf(wmx, __shfl_xor(wmx, o, 64));
    __shared__ float gm_wm[4];
    __shared__ unsigned int gm_last;
    if ((threadIdx.x & 63) == 0) gm_wm[threadIdx.x >> 6] = wmx;
    __syncthreads();
    if (threadIdx.x == 0) {
        // No fence (a device-scope release is an L2 write-back on gfx950: NOTES R4.2): the block maximum is left by a RETURNING
        // device-scope atomic, whose result is waited for before the ticket is taken -- it has been performed where the last
        // workgroup's device-scope loads will look.
        const unsigned int old = __hip_atomic_exchange(block_bits + blockIdx.x,
                                                       __builtin_bit_cast(unsigned int, fmaxf(fmaxf(gm_wm[0], gm_wm[1]), fmaxf(gm_wm[2], gm_wm[3]))),
                                                       __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        asm volatile("" ::"v"(old) : "memory");
        gm_last = __hip_atomic_fetch_add(ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == gridDim.x - 1 ? 1u : 0u;
    }
    __syncthreads();
    if (gm_last) {
        unsigned int m = 0u;
        for (unsigned int b = threadIdx.x; b < gridDim.x; b += 256)
            m = max(m, __hip_atomic_load(block_bits + b, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) m = max(m, (unsigned int)__shfl_xor((int)m, o, 64));
        __shared__ unsigned int gm_um[4];
        if ((threadIdx.x & 63) == 0) gm_um[threadIdx.x >> 6] = m;
        __syncthreads();
        if (threadIdx.x == 0) {
            *all_bits = max(max(gm_um[0], gm_um[1]), max(gm_um[2], gm_um[3]));
            __hip_atomic_store(ticket, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
}
#endif

// hipFuncAttributeMaxDynamicSharedMemorySize belongs to (kernel, DEVICE): one LdsOnce per call site, one bit per device, set only after the
// runtime accepted the call (two host threads racing both make the idempotent call; a process that launches on a second GPU sets it there
// too).  -> false if the runtime refused: the launch that follows then fails and DIR_CHECK_LAUNCH reports it.
struct LdsOnce { std::atomic<uint64_t> done{0}; };
template <typename... Ks>
inline bool lds_limit(LdsOnce& o, int bytes, Ks... kernels) {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return false;
    const uint64_t bit = 1ull << (dev & 63);
    if (o.done.load(std::memory_order_acquire) & bit) return true;
    bool ok = true;
    ((ok = (hipFuncSetAttribute(reinterpret_cast<const void*>(kernels), hipFuncAttributeMaxDynamicSharedMemorySize, bytes) == hipSuccess) && ok), ...);
    if (ok) o.done.fetch_or(bit, std::memory_order_release);
    return ok;
}

inline int grid_resident(int64_t work_blocks, int resident) {
    int64_t g = work_blocks < resident ? work_blocks : resident;
    return g < 1 ? 1 : (int)g;
}

// csrc/radix_sort.hip: the in-tree stable LSD radix sort of (uint32 key, uint32 value) pairs -- kernel launches only, so it can sit inside a
// HIP-graph capture (rocPRIM's sort issues hipMemsetAsync: memset nodes, see that file's header) -- and zero fills as kernels.
size_t radix_sort_temp_bytes(size_t n, unsigned bits);
int radix_sort_input_buffer(size_t n, unsigned bits);       // 0: the input pairs go to k0 / v0, 1: to k1 / v1; sorted pairs always land in k1 / v1
hipError_t radix_sort_pairs_u32(void* tmp, uint32_t* k0, uint32_t* k1, uint32_t* v0, uint32_t* v1, size_t n, unsigned bits, hipStream_t st);
// the slot-major form for one-hot entries ids [B, F] (gbits = bit_length(total_rows)): see csrc/radix_sort.hip
bool radix_slot_sort_ok(int64_t B, int F, unsigned gbits);
size_t radix_slot_sort_temp_bytes(int64_t B, int F, unsigned gbits);
hipError_t radix_slot_sort_entries(void* tmp, const int64_t* ids, int64_t sb, int64_t sf, int F, int64_t B, const int64_t* row_base,
                                   uint32_t total_rows, unsigned gbits, uint32_t* k0, uint32_t* k1, uint32_t* v0, uint32_t* v1, hipStream_t st);
// Development A/B switches read from the environment (DIR_RS_DBG, DIR_SORT, DIR_ADA_STAGE_MIN, DIR_BUCKET_EPT, DIR_BUCKET_NT ...) exist only in a
// build with -DDIR_DEVELOPMENT (DIR_DEVELOPMENT=1 python build.py): some of them produce WRONG results by design (timing masks), so the
// production library does not look at them at all; dev_env_set() lets an entry point fail loudly when one is set anyway.
inline const char* dev_env(const char* name) {
#ifdef DIR_DEVELOPMENT
    return getenv(name);
#else
    (void)name;
    return nullptr;
#endif
}
inline int dev_env_int(const char* name, int dflt) { const char* e = dev_env(name); return e ? atoi(e) : dflt; }
inline bool dev_env_ignored(const char* name) {        // the variable is set, and this build ignores it
#ifdef DIR_DEVELOPMENT
    (void)name;
    return false;
#else
    const char* e = getenv(name);
    return e && *e && strcmp(e, "0") != 0;
#endif
}
hipError_t zero_async(void* p, size_t bytes, hipStream_t st);                                                // p, bytes: multiples of 4
hipError_t zero_2d_async(void* p, size_t pitch_bytes, size_t width_bytes, size_t rows, hipStream_t st);      // as hipMemset2DAsync(.., 0, ..)

// csrc/din_wave.hip: the wave-per-sample DIN forward for the (K = 64, H1 <= 80, H2 <= 48, T <= 64) shape class
bool din_wave_covers(int K, int T, int H1, int H2);
int launch_din_wave(hipStream_t st, const float* table, const int64_t* hist, const int32_t* hist_len, const int64_t* cand, int T,
                    const float* W1, const float* b1, int H1, const float* W2, const float* b2, int H2, const float* W3,
                    const float* b3, int normalize, int64_t B, float* out, float* scores, const int64_t* tile_off = nullptr,
                    float* saved = nullptr, int activation = 0 /* 0 sigmoid, 1 PReLU, 2 Dice */, const float* act_params = nullptr);
// The per-row record the DIN training path keeps between kernels (one per history position inside its sample's length, 16 rows per tile,
// a sample's tiles consecutive from tile_off[b]): z1 [80] | dpre1 [80] | z2 [48] | d score | 3 pad.  The training forward
// (dir_din_attention_pool_save_f32) writes z1 and z2; the backward's row pass adds dpre1 and d score; its weight-gradient pass reads all.
constexpr int kDinRecRow = 212, kDinRecZ1 = 0, kDinRecDp1 = 80, kDinRecZ2 = 160, kDinRecDs = 208;
#ifndef DIN_NT
#define DIN_NT 0          // 1: non-temporal record stores / loads.  Measured (tools/build_nt_variants.sh, one box): the step 2.77 -> 2.87 ms,
                          // the row pass 0.88 -> 0.99 ms, the saving forward 0.56 -> 0.64 ms -- the records do profit from L2 / Infinity Cache
#endif

}  // namespace dir

// ---- device helpers ------------------------------------------------------------------------
#if defined(__HIPCC__)
namespace dir {

// Value of `v` from lane `src` of the wave (all lanes active).
__device__ __forceinline__ float shfl(float v, int src) { return __shfl(v, src, 64); }

// 16-byte accesses to the DIN training records (streamed: each byte is written once and read once or twice by LATER kernels)
typedef float din_rec_f4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void din_rec_store(float* p, float a, float b, float c, float d) {
    const din_rec_f4 v = {a, b, c, d};
#if DIN_NT
    __builtin_nontemporal_store(v, reinterpret_cast<din_rec_f4*>(p));
#else
    *reinterpret_cast<din_rec_f4*>(p) = v;
#endif
}
__device__ __forceinline__ float4 din_rec_load(const float* p) {
#if DIN_NT
    const din_rec_f4 v = __builtin_nontemporal_load(reinterpret_cast<const din_rec_f4*>(p));
#else
    const din_rec_f4 v = *reinterpret_cast<const din_rec_f4*>(p);
#endif
    return make_float4(v[0], v[1], v[2], v[3]);
}

// Sum over the LPS consecutive lanes that share a sample (butterfly; result in every lane).
template <int LPS>
__device__ __forceinline__ float group_sum(float v) {
#pragma unroll
    for (int o = LPS / 2; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

__device__ __forceinline__ float wave_sum(float v) { return group_sum<64>(v); }

// ---- DPP reductions: cross-lane adds inside the VALU (no LDS round trip, unlike ds_bpermute shuffles) -----
// A DPP "row" is 16 consecutive lanes.  quad_perm [1,0,3,2] = 0xB1, [2,3,0,1] = 0x4E, row_half_mirror = 0x141,
// row_mirror = 0x140.  After the four steps every lane of a row holds the row's total.
template <int CTRL>
__device__ __forceinline__ float dpp_mov(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xF, 0xF, false));
}
__device__ __forceinline__ float row16_sum(float v) {
    v += dpp_mov<0xB1>(v);
    v += dpp_mov<0x4E>(v);
    v += dpp_mov<0x141>(v);
    v += dpp_mov<0x140>(v);
    return v;
}
__device__ __forceinline__ float row16_max(float v) {
    v = fmaxf(v, dpp_mov<0xB1>(v));
    v = fmaxf(v, dpp_mov<0x4E>(v));
    v = fmaxf(v, dpp_mov<0x141>(v));
    v = fmaxf(v, dpp_mov<0x140>(v));
    return v;
}
__device__ __forceinline__ float lane_bcast(float v, int l) {
    return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), l));
}
// whole-wave reductions: DPP inside the four rows, then four scalar lane reads
__device__ __forceinline__ float wave_sum_dpp(float v) {
    v = row16_sum(v);
    return (lane_bcast(v, 0) + lane_bcast(v, 16)) + (lane_bcast(v, 32) + lane_bcast(v, 48));
}
__device__ __forceinline__ float wave_max_dpp(float v) {
    v = row16_max(v);
    return fmaxf(fmaxf(lane_bcast(v, 0), lane_bcast(v, 16)), fmaxf(lane_bcast(v, 32), lane_bcast(v, 48)));
}

__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}

}  // namespace dir
#endif
