// copy_probe.hip -- which form of a plain device copy reaches the guide's 6.29 TB/s on this box (bench.py's
// ceiling kernel, csrc/diag.hip, measured 4.7-4.9 TB/s once its windows could not sit in the Infinity Cache -- BELOW the gather's own
// DRAM-side rate, so it is not a ceiling).  Variants: load / store cache policy, grid-stride vs block-contiguous chunks, loads in
// flight per lane, grid size, and the runtime's own hipMemcpyAsync.  Source and destination windows rotate through 3 GiB each.
// build: hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/copy_probe.hip -o tools/copy_probe
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
typedef float f4 __attribute__((ext_vector_type(4)));

template <int U, bool NTL, bool NTS>
__global__ __launch_bounds__(256) void copy_stride_k(const f4* __restrict__ p, f4* __restrict__ q, int64_t n4) {
    const int64_t stride = (int64_t)gridDim.x * 256;
    int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    for (; i + (U - 1) * stride < n4; i += U * stride) {
        f4 v[U];
#pragma unroll
        for (int u = 0; u < U; ++u) v[u] = NTL ? __builtin_nontemporal_load(p + i + u * stride) : p[i + u * stride];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            if (NTS) __builtin_nontemporal_store(v[u], q + i + u * stride);
            else q[i + u * stride] = v[u];
        }
    }
    for (; i < n4; i += stride) q[i] = p[i];
}

// block-contiguous: block b owns [b*chunk, (b+1)*chunk) and walks it 256*U float4 at a time
template <int U, bool NTL, bool NTS>
__global__ __launch_bounds__(256) void copy_chunk_k(const f4* __restrict__ p, f4* __restrict__ q, int64_t n4) {
    const int64_t chunk = (n4 + gridDim.x - 1) / gridDim.x;
    const int64_t lo = (int64_t)blockIdx.x * chunk, hi = lo + chunk < n4 ? lo + chunk : n4;
    for (int64_t i = lo + threadIdx.x; i < hi; i += 256 * U) {
        f4 v[U];
#pragma unroll
        for (int u = 0; u < U; ++u)
            if (i + u * 256 < hi) v[u] = NTL ? __builtin_nontemporal_load(p + i + u * 256) : p[i + u * 256];
#pragma unroll
        for (int u = 0; u < U; ++u)
            if (i + u * 256 < hi) {
                if (NTS) __builtin_nontemporal_store(v[u], q + i + u * 256);
                else q[i + u * 256] = v[u];
            }
    }
}

template <int U, bool NTL>
__global__ __launch_bounds__(256) void read_stride_k(const f4* __restrict__ p, int64_t n4, float* sink) {
    const int64_t stride = (int64_t)gridDim.x * 256;
    f4 acc = {0, 0, 0, 0};
    int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    for (; i + (U - 1) * stride < n4; i += U * stride) {
#pragma unroll
        for (int u = 0; u < U; ++u) acc += NTL ? __builtin_nontemporal_load(p + i + u * stride) : p[i + u * stride];
    }
    for (; i < n4; i += stride) acc += p[i];
    if (acc.x + acc.y + acc.z + acc.w == 1.2345e38f) sink[0] = acc.x;
}

template <bool NTS>
__global__ __launch_bounds__(256) void write_stride_k(f4* __restrict__ q, int64_t n4) {
    const int64_t stride = (int64_t)gridDim.x * 256;
    const f4 v = {1.f, 2.f, 3.f, 4.f};
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += stride) {
        if (NTS) __builtin_nontemporal_store(v, q + i);
        else q[i] = v;
    }
}

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

int main() {
    const size_t pool = (size_t)3 << 30;
    char *src, *dst;
    float* sink;
    CK(hipMalloc(&src, pool));
    CK(hipMalloc(&dst, pool));
    CK(hipMalloc(&sink, 64));
    CK(hipMemset(src, 1, pool));
    CK(hipMemset(dst, 2, pool));
    hipStream_t st;
    CK(hipStreamCreate(&st));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    const size_t sizes[] = {(size_t)109051904, (size_t)1 << 29, (size_t)1 << 30};
    printf("%-44s %10s %8s %10s %10s\n", "variant", "bytes", "grid", "us", "GB/s moved");
    for (size_t nb : sizes) {
        const int64_t n4 = nb / 16;
        const int nwin = (int)(pool / nb);
        auto time = [&](const char* name, int grid, double moved_per, auto launch) {
            const int iters = nb > ((size_t)1 << 28) ? 6 : 16;
            for (int i = 0; i < 2; ++i) launch(i % nwin, (i * 2 + 1) % nwin);
            CK(hipStreamSynchronize(st));
            CK(hipEventRecord(e0, st));
            for (int i = 0; i < iters; ++i) launch((i + 2) % nwin, ((i + 2) * 2 + 1) % nwin);
            CK(hipEventRecord(e1, st));
            CK(hipStreamSynchronize(st));
            float ms;
            CK(hipEventElapsedTime(&ms, e0, e1));
            const double us = ms * 1e3 / iters;
            printf("%-44s %10zu %8d %10.1f %10.0f\n", name, nb, grid, us, moved_per * nb / (us * 1e-6) / 1e9);
        };
#define SRC(w) reinterpret_cast<const f4*>(src + (size_t)(w) * nb)
#define DST(w) reinterpret_cast<f4*>(dst + (size_t)(w) * nb)
        for (int grid : {1024, 2048, 4096, 8192}) {
            time("read  stride U4 nt", grid, 1, [&](int a, int) { hipLaunchKernelGGL((read_stride_k<4, true>), dim3(grid), dim3(256), 0, st, SRC(a), n4, sink); });
            time("write stride plain", grid, 1, [&](int, int b) { hipLaunchKernelGGL((write_stride_k<false>), dim3(grid), dim3(256), 0, st, DST(b), n4); });
            time("copy stride U4 ntL plainS (bench.py r3)", grid, 2, [&](int a, int b) { hipLaunchKernelGGL((copy_stride_k<4, true, false>), dim3(grid), dim3(256), 0, st, SRC(a), DST(b), n4); });
            time("copy stride U4 plainL plainS", grid, 2, [&](int a, int b) { hipLaunchKernelGGL((copy_stride_k<4, false, false>), dim3(grid), dim3(256), 0, st, SRC(a), DST(b), n4); });
            time("copy stride U4 ntL ntS", grid, 2, [&](int a, int b) { hipLaunchKernelGGL((copy_stride_k<4, true, true>), dim3(grid), dim3(256), 0, st, SRC(a), DST(b), n4); });
            time("copy stride U1 plainL plainS", grid, 2, [&](int a, int b) { hipLaunchKernelGGL((copy_stride_k<1, false, false>), dim3(grid), dim3(256), 0, st, SRC(a), DST(b), n4); });
            time("copy stride U8 ntL plainS", grid, 2, [&](int a, int b) { hipLaunchKernelGGL((copy_stride_k<8, true, false>), dim3(grid), dim3(256), 0, st, SRC(a), DST(b), n4); });
            time("copy chunk  U4 ntL plainS", grid, 2, [&](int a, int b) { hipLaunchKernelGGL((copy_chunk_k<4, true, false>), dim3(grid), dim3(256), 0, st, SRC(a), DST(b), n4); });
            time("copy chunk  U8 plainL plainS", grid, 2, [&](int a, int b) { hipLaunchKernelGGL((copy_chunk_k<8, false, false>), dim3(grid), dim3(256), 0, st, SRC(a), DST(b), n4); });
            time("copy chunk  U4 ntL ntS", grid, 2, [&](int a, int b) { hipLaunchKernelGGL((copy_chunk_k<4, true, true>), dim3(grid), dim3(256), 0, st, SRC(a), DST(b), n4); });
        }
        time("hipMemcpyAsync D2D", 0, 2, [&](int a, int b) { CK(hipMemcpyAsync(dst + (size_t)b * nb, src + (size_t)a * nb, nb, hipMemcpyDeviceToDevice, st)); });
        // same-window copy (src and dst fixed): what a cache-resident destination makes of the number
        time("copy stride U4 ntL plainS, FIXED windows", 2048, 2, [&](int, int) { hipLaunchKernelGGL((copy_stride_k<4, true, false>), dim3(2048), dim3(256), 0, st, SRC(0), DST(0), n4); });
    }
    return 0;
}
