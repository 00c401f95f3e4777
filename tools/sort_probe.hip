// tools/sort_probe.hip -- rocprim radix_sort_pairs configurations on the sparse-Adagrad key stream (development tool).
// build: hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/sort_probe.hip -o tools/sort_probe ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstring>
#include <rocprim/device/device_radix_sort.hpp>
#include <cstdio>
#include <cstdint>
#include <vector>
#include <random>
#include <algorithm>

template <class Cfg>
static void run(const char* name, const uint32_t* k0, uint32_t* k1, const uint32_t* v0, uint32_t* v1, size_t n, unsigned bits,
                const std::vector<uint32_t>& ref) {
    size_t tmp = 0;
    if (rocprim::radix_sort_pairs<Cfg>(nullptr, tmp, k0, k1, v0, v1, n, 0u, bits, (hipStream_t)0) != hipSuccess) { printf("%s: size query failed\n", name); return; }
    void* t = nullptr;
    hipMalloc(&t, tmp ? tmp : 256);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    for (int i = 0; i < 3; ++i) rocprim::radix_sort_pairs<Cfg>(t, tmp, k0, k1, v0, v1, n, 0u, bits, (hipStream_t)0);
    hipEventRecord(e0);
    for (int i = 0; i < 20; ++i) rocprim::radix_sort_pairs<Cfg>(t, tmp, k0, k1, v0, v1, n, 0u, bits, (hipStream_t)0);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    std::vector<uint32_t> out(n);
    hipMemcpy(out.data(), k1, n * 4, hipMemcpyDeviceToHost);
    printf("%-34s %7.1f us  tmp %zu KB  %s\n", name, ms / 20 * 1e3, tmp >> 10, out == ref ? "ok" : "WRONG");
    hipFree(t);
}

template <unsigned BITS, unsigned BS, unsigned IPT, unsigned HBS = 512, unsigned HIPT = 12>
using Cfg = rocprim::radix_sort_config<rocprim::default_config, rocprim::default_config,
                                       rocprim::radix_sort_onesweep_config<rocprim::kernel_config<HBS, HIPT>, rocprim::kernel_config<BS, IPT>, BITS,
                                                                           rocprim::block_radix_rank_algorithm::match>, 0>;

int main() {
    const size_t n = 65536 * 26;
    const unsigned bits = 25;
    std::vector<uint32_t> k(n), v(n);
    std::mt19937 g(1);
    for (size_t i = 0; i < n; ++i) { k[i] = (uint32_t)((i % 26) * 1000000u + g() % 1000000u); v[i] = (uint32_t)i; }
    std::vector<uint32_t> ref = k;
    std::sort(ref.begin(), ref.end());
    uint32_t *k0, *k1, *v0, *v1;
    hipMalloc(&k0, n * 4); hipMalloc(&k1, n * 4); hipMalloc(&v0, n * 4); hipMalloc(&v1, n * 4);
    hipMemcpy(k0, k.data(), n * 4, hipMemcpyHostToDevice);
    hipMemcpy(v0, v.data(), n * 4, hipMemcpyHostToDevice);
    run<rocprim::default_config>("default", k0, k1, v0, v1, n, bits, ref);
    run<Cfg<9, 1024, 6>>("9 bits 1024x6", k0, k1, v0, v1, n, bits, ref);
    run<Cfg<9, 1024, 4>>("9 bits 1024x4", k0, k1, v0, v1, n, bits, ref);
    run<Cfg<9, 1024, 8>>("9 bits 1024x8", k0, k1, v0, v1, n, bits, ref);
    run<Cfg<9, 1024, 10>>("9 bits 1024x10", k0, k1, v0, v1, n, bits, ref);
    run<Cfg<9, 1024, 6, 1024, 6>>("9 bits 1024x6 hist 1024x6", k0, k1, v0, v1, n, bits, ref);
    run<Cfg<9, 1024, 6, 256, 8>>("9 bits 1024x6 hist 256x8", k0, k1, v0, v1, n, bits, ref);
    run<Cfg<9, 1024, 6, 512, 32>>("9 bits 1024x6 hist 512x32", k0, k1, v0, v1, n, bits, ref);
    return 0;
}
