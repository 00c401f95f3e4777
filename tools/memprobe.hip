// memprobe.hip -- MI355X memory-system probe for the embedding gather (development tool, not product).
// Measures, at the BASELINE byte counts (109 MB of rows out of a 1.66 GB table):
//   R<rb>   random whole rows of rb bytes, 16 B per lane, sum kept in a register (negligible writes)
//   R<rb>nt same with non-temporal loads
//   Wseg    the gather's store pattern: 64-B segments at a 1664-B stride per instruction
//   Wlin    fully coalesced 1-KiB-per-instruction stores of the same bytes
//   RW64    random 64-B rows read + written in the gather pattern (= the gather without ids)
// Build: hipcc --offload-arch=gfx950 -O3 tools/memprobe.hip -o tools/memprobe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
#include <vector>
#include <algorithm>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)

typedef float f32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ float4 ntld(const float4* p) {
    f32x4 v = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(p));
    return make_float4(v.x, v.y, v.z, v.w);
}
__device__ __forceinline__ void ntst(float4 v, float4* p) {
    f32x4 w = {v.x, v.y, v.z, v.w};
    __builtin_nontemporal_store(w, reinterpret_cast<f32x4*>(p));
}
__device__ __forceinline__ uint64_t mix(uint64_t x) {
    x ^= x >> 33; x *= 0xff51afd7ed558ccdULL; x ^= x >> 33; x *= 0xc4ceb9fe1a85ec53ULL; x ^= x >> 33; return x;
}

template <int LPR, bool NT, int UF>
__global__ __launch_bounds__(256) void rd_rows(const float4* __restrict__ tab, const uint32_t* __restrict__ idx,
                                               int64_t nrows, float* __restrict__ sink) {
    const int lane = threadIdx.x & 63, c = lane % LPR, s = lane / LPR;
    constexpr int RPW = 64 / LPR;
    const int64_t nwave = (int64_t)gridDim.x * 4;
    float4 acc = make_float4(0, 0, 0, 0);
    for (int64_t g = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6); g * RPW * UF < nrows; g += nwave) {
        uint32_t id[UF];
        float4 v[UF];
#pragma unroll
        for (int u = 0; u < UF; ++u) { int64_t r = (g * UF + u) * RPW + s; id[u] = r < nrows ? idx[r] : 0; }
#pragma unroll
        for (int u = 0; u < UF; ++u) {
            const float4* p = tab + (int64_t)id[u] * LPR + c;
            v[u] = NT ? ntld(p) : *p;
        }
#pragma unroll
        for (int u = 0; u < UF; ++u) { acc.x += v[u].x; acc.y += v[u].y; acc.z += v[u].z; acc.w += v[u].w; }
    }
    if (acc.x + acc.y + acc.z + acc.w == 12345.678f) sink[0] = acc.x;
}

// gather-shaped: 16 samples per wave, F fields, 64-B rows; optional read / write / nt-store
template <bool RD, bool WR, bool NTS, int UF>
__global__ __launch_bounds__(256) void gshape(const float4* __restrict__ tab, const uint32_t* __restrict__ idx,
                                              int64_t B, int F, int64_t rows_per_field, float4* __restrict__ out,
                                              float* __restrict__ sink) {
    const int lane = threadIdx.x & 63, c = lane & 3, s = lane >> 2;
    const int64_t nwave = (int64_t)gridDim.x * 4;
    float4 acc = make_float4(0, 0, 0, 0);
    for (int64_t g = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6); g * 16 < B; g += nwave) {
        const int64_t b = g * 16 + s;
        for (int f0 = 0; f0 < F; f0 += UF) {
            float4 v[UF];
#pragma unroll
            for (int u = 0; u < UF; ++u) {
                v[u] = make_float4(1.f, 2.f, 3.f, (float)u);
                if (RD && f0 + u < F) {
                    uint32_t id = idx[b * F + f0 + u];
                    v[u] = tab[((int64_t)(f0 + u) * rows_per_field + id) * 4 + c];
                }
            }
#pragma unroll
            for (int u = 0; u < UF; ++u) {
                if (f0 + u < F) {
                    if (WR) {
                        float4* p = out + (b * F + f0 + u) * 4 + c;
                        if (NTS) ntst(v[u], p); else *p = v[u];
                    } else { acc.x += v[u].x; acc.y += v[u].y; acc.z += v[u].z; acc.w += v[u].w; }
                }
            }
        }
    }
    if (!WR && acc.x + acc.y + acc.z + acc.w == 12345.678f) sink[0] = acc.x;
}

__global__ void wlin(float4* __restrict__ out, int64_t n4) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (int64_t)gridDim.x * blockDim.x)
        out[i] = make_float4(1.f, 2.f, 3.f, 4.f);
}
__global__ void rlin(const float4* __restrict__ in, int64_t n4, float* sink) {
    float4 acc = make_float4(0, 0, 0, 0);
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (int64_t)gridDim.x * blockDim.x) {
        float4 v = in[i]; acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
    }
    if (acc.x + acc.y + acc.z + acc.w == 12345.678f) sink[0] = acc.x;
}
__global__ void fill_idx(uint32_t* idx, int64_t n, uint32_t mod, uint64_t seed) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
        idx[i] = (uint32_t)(mix(i * 0x9E3779B97F4A7C15ULL + seed) % mod);
}

// sorted variant: a uniformly spaced, jittered, INCREASING index list (what sorting a uniform random id set gives)
__global__ void fill_idx_sorted(uint32_t* idx, int64_t n, uint32_t mod, uint64_t seed) {
    const double stride = (double)mod / (double)n;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const uint32_t base = (uint32_t)(i * stride);
        const uint32_t jit = (uint32_t)(mix(i * 0x9E3779B97F4A7C15ULL + seed) % (uint32_t)(stride < 1 ? 1 : stride));
        idx[i] = base + jit < mod ? base + jit : mod - 1;
    }
}

template <typename F> static double timeit(F f, int iters = 30) {
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    for (int i = 0; i < 5; ++i) f(i);
    CK(hipDeviceSynchronize());
    std::vector<float> ts;
    for (int i = 0; i < iters; ++i) { CK(hipEventRecord(a)); f(i); CK(hipEventRecord(b)); CK(hipEventSynchronize(b)); float ms; CK(hipEventElapsedTime(&ms, a, b)); ts.push_back(ms); }
    std::sort(ts.begin(), ts.end());
    return ts[ts.size() / 2] * 1e3;  // median us
}

int main(int argc, char** argv) {
    const int64_t TAB_BYTES = 26LL * 1000000 * 64;          // 1.664 GB
    const int64_t ROW_BYTES_TOTAL = 65536LL * 26 * 64;     // 109 MB
    const int NB = 4;                                       // rotated index batches
    float4* tab; float4* out; float* sink; uint32_t* idx;
    CK(hipMalloc(&tab, TAB_BYTES)); CK(hipMalloc(&out, ROW_BYTES_TOTAL)); CK(hipMalloc(&sink, 64));
    CK(hipMalloc(&idx, sizeof(uint32_t) * 65536 * 26 * NB));
    CK(hipMemset(tab, 0, TAB_BYTES));
    int grids[] = {512, 1024, 2048, 4096};
    printf("%-10s %-6s %10s %10s\n", "test", "grid", "us", "GB/s");
    auto report = [&](const char* name, int grid, double us, double bytes) { printf("%-10s %-6d %10.1f %10.0f\n", name, grid, us, bytes / us * 1e-3); fflush(stdout); };
    {   // streaming ceilings
        double us = timeit([&](int) { hipLaunchKernelGGL(wlin, dim3(2048), dim3(256), 0, 0, out, ROW_BYTES_TOTAL / 16); });
        report("Wlin", 2048, us, ROW_BYTES_TOTAL);
        us = timeit([&](int i) { hipLaunchKernelGGL(rlin, dim3(2048), dim3(256), 0, 0, tab + (i % 8) * (ROW_BYTES_TOTAL / 16), ROW_BYTES_TOTAL / 16, sink); });
        report("Rlin", 2048, us, ROW_BYTES_TOTAL);
    }
#define RUN_RD(RB, NT, UF, NAME)                                                                        \
    for (int gi = 0; gi < 4; ++gi) {                                                                      \
        const int64_t nrows = ROW_BYTES_TOTAL / RB;                                                       \
        hipLaunchKernelGGL(fill_idx, dim3(1024), dim3(256), 0, 0, idx, nrows * NB, (uint32_t)(TAB_BYTES / RB), 7ULL); \
        double us = timeit([&](int i) { hipLaunchKernelGGL((rd_rows<RB / 16, NT, UF>), dim3(grids[gi]), dim3(256), 0, 0, tab, idx + (i % NB) * nrows, nrows, sink); }); \
        report(NAME, grids[gi], us, (double)ROW_BYTES_TOTAL + nrows * 4);                                 \
    }
    if (getenv("MEMPROBE_SORTED")) {   // random rows visited in increasing address order (per launch): is the request ceiling a DRAM-page effect?
        for (int gi = 1; gi < 3; ++gi) {
            const int64_t nrows = ROW_BYTES_TOTAL / 64;
            for (int b = 0; b < NB; ++b)
                hipLaunchKernelGGL(fill_idx_sorted, dim3(1024), dim3(256), 0, 0, idx + b * nrows, nrows, (uint32_t)(TAB_BYTES / 64), 7ULL + b);
            double us = timeit([&](int i) { hipLaunchKernelGGL((rd_rows<4, false, 8>), dim3(grids[gi]), dim3(256), 0, 0, tab, idx + (i % NB) * nrows, nrows, sink); });
            report("R64sorted", grids[gi], us, (double)ROW_BYTES_TOTAL + nrows * 4);
            us = timeit([&](int i) { hipLaunchKernelGGL((rd_rows<4, true, 8>), dim3(grids[gi]), dim3(256), 0, 0, tab, idx + (i % NB) * nrows, nrows, sink); });
            report("R64sortNT", grids[gi], us, (double)ROW_BYTES_TOTAL + nrows * 4);
        }
    }
    RUN_RD(64, false, 8, "R64");
    RUN_RD(64, true, 8, "R64nt");
    RUN_RD(64, false, 16, "R64u16");
    RUN_RD(128, false, 8, "R128");
    RUN_RD(128, true, 8, "R128nt");
    RUN_RD(256, false, 8, "R256");
    // gather-shaped
    hipLaunchKernelGGL(fill_idx, dim3(1024), dim3(256), 0, 0, idx, 65536LL * 26 * NB, 1000000u, 11ULL);
    for (int gi = 0; gi < 4; ++gi) {
        const int64_t n = 65536LL * 26;
        double us = timeit([&](int i) { hipLaunchKernelGGL((gshape<true, false, false, 13>), dim3(grids[gi]), dim3(256), 0, 0, tab, idx + (i % NB) * n, 65536, 26, 1000000, out, sink); });
        report("G_rd", grids[gi], us, (double)ROW_BYTES_TOTAL + n * 4);
        us = timeit([&](int i) { hipLaunchKernelGGL((gshape<false, true, false, 13>), dim3(grids[gi]), dim3(256), 0, 0, tab, idx, 65536, 26, 1000000, out, sink); });
        report("G_wr", grids[gi], us, (double)ROW_BYTES_TOTAL);
        us = timeit([&](int i) { hipLaunchKernelGGL((gshape<false, true, true, 13>), dim3(grids[gi]), dim3(256), 0, 0, tab, idx, 65536, 26, 1000000, out, sink); });
        report("G_wr_nt", grids[gi], us, (double)ROW_BYTES_TOTAL);
        us = timeit([&](int i) { hipLaunchKernelGGL((gshape<true, true, false, 13>), dim3(grids[gi]), dim3(256), 0, 0, tab, idx + (i % NB) * n, 65536, 26, 1000000, out, sink); });
        report("G_rw", grids[gi], us, 2.0 * ROW_BYTES_TOTAL + n * 4);
        us = timeit([&](int i) { hipLaunchKernelGGL((gshape<true, true, true, 13>), dim3(grids[gi]), dim3(256), 0, 0, tab, idx + (i % NB) * n, 65536, 26, 1000000, out, sink); });
        report("G_rw_nts", grids[gi], us, 2.0 * ROW_BYTES_TOTAL + n * 4);
    }
    return 0;
}
