// diag.hip -- measurement aids behind dir_debug_* (never on the product path): the streaming ceilings of THIS box, measured
// in the same process as the kernels they are compared with (bench.py's roofline.frac_of_measured_*).
//   dir_debug_stream_read_f32 : every lane reads 16 B per step, grid-stride over n floats, non-temporal; nothing is written
//                               (the sum is stored only if it equals a value it cannot take), so the time is a pure linear read.
//   dir_debug_stream_copy_f32 : read n + write n, 16 bytes per lane, workgroup-contiguous chunks (see stream_copy_k).
#include "common.hpp"

namespace dir {

typedef float f32x4d __attribute__((ext_vector_type(4)));

__global__ __launch_bounds__(256) void stream_read_k(const f32x4d* __restrict__ p, int64_t n4, float* __restrict__ sink) {
    f32x4d acc = {0.f, 0.f, 0.f, 0.f};
    const int64_t stride = (int64_t)gridDim.x * 256;
    int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    for (; i + 3 * stride < n4; i += 4 * stride) {      // four independent 16-byte loads in flight per lane
        f32x4d a = __builtin_nontemporal_load(p + i), b = __builtin_nontemporal_load(p + i + stride);
        f32x4d c = __builtin_nontemporal_load(p + i + 2 * stride), d = __builtin_nontemporal_load(p + i + 3 * stride);
        acc += (a + b) + (c + d);
    }
    for (; i < n4; i += stride) acc += __builtin_nontemporal_load(p + i);
    const float s = (acc.x + acc.y) + (acc.z + acc.w);
    if (s == 1.2345678e38f) sink[0] = s;                // never true for finite data of this size; keeps the loads alive
}

// The copy's form was picked with tools/copy_probe.hip (profiles/r04_copy_probe.txt: 512 MB - 1 GiB windows that cannot sit in the Infinity
// Cache): every workgroup walks ONE contiguous chunk 16 KiB at a time, non-temporal loads AND stores -- 5.6-6.0 TB/s of bytes moved, against
// 4.4-4.9 TB/s for round 3's grid-stride form with plain stores (whose lanes' next access lies grid x 4 KiB away), 4.95-5.4 for
// hipMemcpyAsync; the guide quotes 6.29 TB/s for its float4 copy.
__global__ __launch_bounds__(256) void stream_copy_k(const f32x4d* __restrict__ p, f32x4d* __restrict__ q, int64_t n4) {
    const int64_t chunk = (n4 + gridDim.x - 1) / gridDim.x;
    const int64_t lo = (int64_t)blockIdx.x * chunk, hi = lo + chunk < n4 ? lo + chunk : n4;
    for (int64_t i = lo + threadIdx.x; i < hi; i += 1024) {
        f32x4d v[4];
#pragma unroll
        for (int u = 0; u < 4; ++u)
            if (i + u * 256 < hi) v[u] = __builtin_nontemporal_load(p + i + u * 256);
#pragma unroll
        for (int u = 0; u < 4; ++u)
            if (i + u * 256 < hi) __builtin_nontemporal_store(v[u], q + i + u * 256);
    }
}

}  // namespace dir

using namespace dir;

extern "C" int dir_debug_stream_read_f32(const float* p, int64_t n, float* sink, dir_stream_t stream) {
    DIR_CHECK_ARG(p && sink && n > 0 && n % 4 == 0 && aligned16(p), "dir_debug_stream_read_f32: n=%lld must be a positive multiple of 4, p 16-byte aligned", (long long)n);
    hipLaunchKernelGGL(stream_read_k, dim3(kCUs * 8), dim3(256), 0, as_stream(stream), reinterpret_cast<const f32x4d*>(p), n / 4, sink);
    DIR_CHECK_LAUNCH("stream_read");
    return DIR_OK;
}

extern "C" int dir_debug_stream_copy_f32(const float* p, float* q, int64_t n, dir_stream_t stream) {
    DIR_CHECK_ARG(p && q && n > 0 && n % 4 == 0 && aligned16(p) && aligned16(q), "dir_debug_stream_copy_f32: bad argument");
    hipLaunchKernelGGL(stream_copy_k, dim3(kCUs * 32), dim3(256), 0, as_stream(stream), reinterpret_cast<const f32x4d*>(p),
                       reinterpret_cast<f32x4d*>(q), n / 4);
    DIR_CHECK_LAUNCH("stream_copy");
    return DIR_OK;
}
