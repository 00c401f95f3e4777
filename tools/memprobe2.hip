// memprobe2.hip -- cache-policy probe for the gather shape (development tool, not product).
// Gather-shaped kernel (16 samples per wave, 26 fields, 64-B rows, [B,416] concat written) with the
// row loads and the output stores issued as raw buffer ops carrying cache-policy bits
// (aux: 1 = sc0, 2 = nt, 16 = sc1).  Prints median time per (load aux, store aux) pair.
// All indices are bounded by construction: idx[b*F+f] with b < B, f < F (buffer holds NB*B*F entries);
// row offset = ((f*V + id)*64 + c*16) < F*V*64 = table bytes (< 2^32); output offset < B*F*64.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
#include <vector>
#include <algorithm>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ uint64_t mix(uint64_t x) {
    x ^= x >> 33; x *= 0xff51afd7ed558ccdULL; x ^= x >> 33; x *= 0xc4ceb9fe1a85ec53ULL; x ^= x >> 33; return x;
}
__global__ void fill_idx(uint32_t* idx, int64_t n, uint32_t mod, uint64_t seed) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
        idx[i] = (uint32_t)(mix(i * 0x9E3779B97F4A7C15ULL + seed) % mod);
}

template <int LAUX, int SAUX, bool WR, int UF>
__global__ __launch_bounds__(256) void gshape(const float* __restrict__ tab, uint32_t tab_bytes,
                                              const uint32_t* __restrict__ idx, int B, int F, uint32_t V,
                                              float* __restrict__ out, uint32_t out_bytes, float* __restrict__ sink) {
    const int lane = threadIdx.x & 63, c = lane & 3, s = lane >> 2;
    __amdgpu_buffer_rsrc_t rt = __builtin_amdgcn_make_buffer_rsrc((void*)tab, 0, tab_bytes, 0x00020000);
    __amdgpu_buffer_rsrc_t ro = __builtin_amdgcn_make_buffer_rsrc((void*)out, 0, out_bytes, 0x00020000);
    const int nwave = gridDim.x * 4;
    unsigned acc = 0;
    for (int g = blockIdx.x * 4 + (threadIdx.x >> 6); g * 16 < B; g += nwave) {
        const int b = g * 16 + s;   // B is a multiple of 16
        for (int f0 = 0; f0 < F; f0 += UF) {
            u32x4 v[UF];
#pragma unroll
            for (int u = 0; u < UF; ++u) {
                if (f0 + u < F) {
                    const uint32_t id = idx[b * F + f0 + u];
                    const uint32_t off = ((uint32_t)(f0 + u) * V + id) * 64u + c * 16u;
                    v[u] = __builtin_amdgcn_raw_buffer_load_b128(rt, off, 0, LAUX);
                }
            }
#pragma unroll
            for (int u = 0; u < UF; ++u) {
                if (f0 + u < F) {
                    if (WR) __builtin_amdgcn_raw_buffer_store_b128(v[u], ro, ((uint32_t)(b * F + f0 + u)) * 64u + c * 16u, 0, SAUX);
                    else acc += v[u].x ^ v[u].y ^ v[u].z ^ v[u].w;
                }
            }
        }
    }
    if (!WR && acc == 0x12345678u) sink[0] = 1.f;
}

template <typename Fn> static double timeit(Fn f, int iters = 40) {
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    for (int i = 0; i < 8; ++i) f(i);
    CK(hipDeviceSynchronize());
    std::vector<float> ts;
    for (int i = 0; i < iters; ++i) { CK(hipEventRecord(a)); f(i); CK(hipEventRecord(b)); CK(hipEventSynchronize(b)); float ms; CK(hipEventElapsedTime(&ms, a, b)); ts.push_back(ms); }
    std::sort(ts.begin(), ts.end());
    return ts[ts.size() / 2] * 1e3;
}

int main() {
    const int B = 65536, F = 26; const uint32_t V = 1000000; const int NB = 4;
    const uint32_t TAB_BYTES = (uint32_t)((uint64_t)F * V * 64);       // 1 664 000 000 < 2^32
    const uint32_t OUT_BYTES = (uint32_t)((uint64_t)B * F * 64);      // 109 051 904
    float* tab; float* out; float* sink; uint32_t* idx;
    CK(hipMalloc(&tab, TAB_BYTES)); CK(hipMalloc(&out, OUT_BYTES)); CK(hipMalloc(&sink, 64));
    CK(hipMalloc(&idx, sizeof(uint32_t) * (size_t)B * F * NB));
    CK(hipMemset(tab, 0, TAB_BYTES));
    hipLaunchKernelGGL(fill_idx, dim3(1024), dim3(256), 0, 0, idx, (int64_t)B * F * NB, V, 11ULL);
    CK(hipDeviceSynchronize());
    const int64_t n = (int64_t)B * F;
    printf("%-8s %-8s %-4s %9s\n", "ld_aux", "st_aux", "wr", "us");
#define RUN(L, S, WR) { double us = timeit([&](int i) { hipLaunchKernelGGL((gshape<L, S, WR, 13>), dim3(1024), dim3(256), 0, 0, tab, TAB_BYTES, idx + (i % NB) * n, B, F, V, out, OUT_BYTES, sink); }); \
        printf("%-8d %-8d %-4d %9.1f\n", L, S, (int)WR, us); fflush(stdout); }
    RUN(0, 0, false) RUN(1, 0, false) RUN(2, 0, false) RUN(3, 0, false) RUN(16, 0, false) RUN(17, 0, false) RUN(18, 0, false) RUN(19, 0, false)
    RUN(0, 0, true) RUN(0, 2, true) RUN(0, 16, true) RUN(0, 17, true) RUN(0, 19, true)
    RUN(2, 0, true) RUN(2, 2, true) RUN(2, 16, true) RUN(2, 17, true) RUN(2, 19, true)
    RUN(16, 0, true) RUN(16, 2, true) RUN(18, 0, true) RUN(18, 2, true) RUN(19, 0, true) RUN(19, 2, true)
    RUN(3, 0, true) RUN(3, 2, true) RUN(1, 0, true) RUN(17, 0, true)
    return 0;
}
