// rmw_probe.hip -- what a random 128-byte read-modify-write stream reaches on this box: the access pattern of the sorted sparse update
// (csrc/backward.hip: adagrad_tile_k -- 1.7 M [embedding | accumulator] rows of 128 bytes read and written in ascending row order, plus a
// random 64-byte gradient row per entry; PMC: 608 MB in 167-174 us = 3.5-3.6 TB/s, profiles/r04_pmc_kernels_train_sparse.json).
// Rows: 26 M x 128 B (3.3 GB); per launch 1.7 M distinct rows, sorted ascending; eight row sets rotate (218 MB of lines each: together
// beyond the 256 MiB Infinity Cache).  Variants: read only, write only, read-modify-write, the same with a random 64-byte read from a
// 109 MB gradient array; U rows in flight per lane group.
// build: hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/rmw_probe.hip -o tools/rmw_probe
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <algorithm>
#include <random>
#include <vector>
typedef float f4 __attribute__((ext_vector_type(4)));

// 8 lanes per row (16 bytes each); MODE 0 read, 1 write, 2 rmw, 3 rmw + gradient read (lanes 0..3 of the group read 64 bytes of g[val])
template <int MODE, int U>
__global__ __launch_bounds__(256) void rows_k(f4* __restrict__ arena, const uint32_t* __restrict__ rows, const uint32_t* __restrict__ val,
                                              const f4* __restrict__ g, int64_t n, float* sink) {
    const int lane8 = threadIdx.x & 7;
    const int64_t grp = ((int64_t)blockIdx.x * 256 + threadIdx.x) >> 3, ngrp = ((int64_t)gridDim.x * 256) >> 3;
    f4 acc = {0, 0, 0, 0};
    for (int64_t i0 = grp * U; i0 < n; i0 += ngrp * U) {
        f4 v[U], gv[U];
        int64_t r[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int64_t i = i0 + u;
            r[u] = i < n ? (int64_t)rows[i] : -1;
            gv[u] = f4{0, 0, 0, 0};
            if (MODE == 3 && r[u] >= 0 && lane8 < 4) gv[u] = g[(int64_t)val[i] * 4 + lane8];
        }
#pragma unroll
        for (int u = 0; u < U; ++u)
            if (r[u] >= 0 && MODE != 1) v[u] = arena[r[u] * 8 + lane8];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            if (r[u] < 0) continue;
            if (MODE == 0) acc += v[u];
            else if (MODE == 1) arena[r[u] * 8 + lane8] = f4{1.f, 2.f, 3.f, (float)lane8};
            else arena[r[u] * 8 + lane8] = v[u] * 1.0001f + gv[u];
        }
    }
    if (acc.x + acc.y + acc.z + acc.w == 1.2345e38f) sink[0] = acc.x;
}

// the same row as TWO 64-byte halves: 4 lanes per row, every lane loads its 16-byte chunk of the first half and of the second half in two
// instructions (how adagrad_tile_k's four-lane groups read [w | accum] rows), the gradient row with all four lanes; stores likewise
// EXTRA: two more L2-resident reads per entry, as the folded FM backward makes them: a 64-byte row of a 4 MB array and a 4-byte scalar,
// both indexed by the entry's sample (val / 26)
template <int U, bool EXTRA = false>
__global__ __launch_bounds__(256) void rows4_k(f4* __restrict__ arena, const uint32_t* __restrict__ rows, const uint32_t* __restrict__ val,
                                               const f4* __restrict__ g, int64_t n, const f4* __restrict__ fs = nullptr,
                                               const float* __restrict__ fg = nullptr) {
    const int lane4 = threadIdx.x & 3;
    const int64_t grp = ((int64_t)blockIdx.x * 256 + threadIdx.x) >> 2, ngrp = ((int64_t)gridDim.x * 256) >> 2;
    for (int64_t i0 = grp * U; i0 < n; i0 += ngrp * U) {
        f4 a[U], b[U], gv[U];
        int64_t r[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int64_t i = i0 + u;
            r[u] = i < n ? (int64_t)rows[i] : -1;
            if (r[u] >= 0) {
                gv[u] = g[(int64_t)val[i] * 4 + lane4];
                if (EXTRA) {
                    const uint32_t bsm = val[i] / 26u;
                    gv[u] = gv[u] + fs[(int64_t)bsm * 4 + lane4] * fg[bsm];
                }
                a[u] = arena[r[u] * 8 + lane4];
                b[u] = arena[r[u] * 8 + 4 + lane4];
            }
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            if (r[u] < 0) continue;
            arena[r[u] * 8 + 4 + lane4] = b[u] + gv[u] * gv[u];
            arena[r[u] * 8 + lane4] = a[u] * 1.0001f + gv[u];
        }
    }
}

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

int main() {
    const int64_t V = 26000000, N = 65536 * 26, SETS = 8;
    f4* arena; uint32_t *rows, *val; f4* g; float* sink;
    CK(hipMalloc(&arena, V * 128)); CK(hipMemset(arena, 0, V * 128));
    CK(hipMalloc(&rows, SETS * N * 4)); CK(hipMalloc(&val, N * 4)); CK(hipMalloc(&g, N * 64)); CK(hipMemset(g, 0, N * 64)); CK(hipMalloc(&sink, 4));
    std::mt19937_64 rng(1);
    std::vector<uint32_t> h(N);
    for (int s = 0; s < SETS; ++s) {
        // distinct rows, as uniform ids over 26 x 1 M rows nearly are: slot f's 65 536 rows inside its own million
        for (int64_t f = 0; f < 26; ++f) {
            std::vector<uint32_t> pick(65536);
            for (auto& x : pick) x = (uint32_t)(f * 1000000 + rng() % 1000000);
            std::sort(pick.begin(), pick.end());
            pick.erase(std::unique(pick.begin(), pick.end()), pick.end());
            while (pick.size() < 65536) pick.push_back(pick.back());      // (a few duplicates close the set: as the real key list has)
            std::copy(pick.begin(), pick.end(), h.begin() + f * 65536);
        }
        CK(hipMemcpy(rows + s * N, h.data(), N * 4, hipMemcpyHostToDevice));
    }
    for (int64_t i = 0; i < N; ++i) h[i] = (uint32_t)i;
    std::shuffle(h.begin(), h.end(), rng);
    CK(hipMemcpy(val, h.data(), N * 4, hipMemcpyHostToDevice));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const char* names[4] = {"read 128 B rows", "write 128 B rows", "read-modify-write 128 B rows", "rmw + random 64 B gradient row"};
    const double bytes[4] = {128.0 * N, 128.0 * N, 256.0 * N, 320.0 * N};
#define RUN(MODE, U, GRID)                                                                                                          \
    do {                                                                                                                            \
        for (int it = 0; it < 3; ++it) hipLaunchKernelGGL((rows_k<MODE, U>), dim3(GRID), dim3(256), 0, 0, arena, rows + (it % SETS) * N, val, g, N, sink); \
        CK(hipEventRecord(e0));                                                                                                     \
        for (int it = 0; it < 24; ++it) hipLaunchKernelGGL((rows_k<MODE, U>), dim3(GRID), dim3(256), 0, 0, arena, rows + (it % SETS) * N, val, g, N, sink); \
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));                                                                        \
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));                                                                             \
        printf("%-34s U=%d grid=%5d: %7.1f us  %5.2f TB/s\n", names[MODE], U, GRID, ms * 1e3 / 24, bytes[MODE] * 24 / (ms * 1e-3) / 1e12); \
    } while (0)
    RUN(0, 1, 2048); RUN(0, 4, 2048); RUN(0, 4, 8192);
    RUN(1, 1, 2048); RUN(1, 4, 2048);
    RUN(2, 1, 2048); RUN(2, 2, 2048); RUN(2, 4, 2048); RUN(2, 4, 8192); RUN(2, 8, 2048);
    RUN(3, 1, 2048); RUN(3, 2, 2048); RUN(3, 4, 2048); RUN(3, 4, 8192); RUN(3, 8, 2048);
#define RUN4(U, GRID)                                                                                                               \
    do {                                                                                                                            \
        for (int it = 0; it < 3; ++it) hipLaunchKernelGGL((rows4_k<U>), dim3(GRID), dim3(256), 0, 0, arena, rows + (it % SETS) * N, val, g, N); \
        CK(hipEventRecord(e0));                                                                                                     \
        for (int it = 0; it < 24; ++it) hipLaunchKernelGGL((rows4_k<U>), dim3(GRID), dim3(256), 0, 0, arena, rows + (it % SETS) * N, val, g, N); \
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));                                                                        \
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));                                                                             \
        printf("%-34s U=%d grid=%5d: %7.1f us  %5.2f TB/s\n", "rmw as two 64 B halves + gradient", U, GRID, ms * 1e3 / 24, 320.0 * N * 24 / (ms * 1e-3) / 1e12); \
    } while (0)
    RUN4(1, 2048); RUN4(2, 2048); RUN4(1, 6656); RUN4(4, 2048);
    f4* fs; float* fg;
    CK(hipMalloc(&fs, 65536 * 64)); CK(hipMemset(fs, 0, 65536 * 64)); CK(hipMalloc(&fg, 65536 * 4)); CK(hipMemset(fg, 0, 65536 * 4));
    for (int grid : {2048, 6656}) {
        for (int it = 0; it < 3; ++it) hipLaunchKernelGGL((rows4_k<1, true>), dim3(grid), dim3(256), 0, 0, arena, rows + (it % SETS) * N, val, g, N, fs, fg);
        CK(hipEventRecord(e0));
        for (int it = 0; it < 24; ++it) hipLaunchKernelGGL((rows4_k<1, true>), dim3(grid), dim3(256), 0, 0, arena, rows + (it % SETS) * N, val, g, N, fs, fg);
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        printf("%-34s U=1 grid=%5d: %7.1f us\n", "... + 64 B and 4 B L2-resident reads", grid, ms * 1e3 / 24);
    }
    return 0;
}
