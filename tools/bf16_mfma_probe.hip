// tools/bf16_mfma_probe.hip -- what a wave sustains on v_mfma_f32_16x16x32_bf16: (0) bare, 10 independent accumulators; (1) the dense
// kernel's pattern: 6 dependent MFMAs per accumulator, 2 accumulators interleaved; (2) = (1) + three ds_read_b128 per 12 MFMAs;
// (3) = (2) + the split VALU work per k step.  One or two waves per SIMD.  Build: hipcc --offload-arch=gfx950 -O3 tools/bf16_mfma_probe.hip -o tools/bf16_mfma_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
#define MFMA(a, b, c) __builtin_amdgcn_mfma_f32_16x16x32_bf16((a), (b), (c), 0, 0, 0)

template <int MODE>
__global__ __launch_bounds__(256) void probe(float* out, int iters, unsigned long long* cyc) {
    __shared__ __attribute__((aligned(16))) unsigned int lds[16384];
    for (int i = threadIdx.x; i < 16384; i += 256) lds[i] = 0x3f803f80u;
    __syncthreads();
    const int lane = threadIdx.x & 63;
    bf16x8 a0 = __builtin_bit_cast(bf16x8, (u32x4){0x3f803f80u + lane, 1, 2, 3}), a1 = a0, a2 = a0;
    bf16x8 b0 = a0, b1 = a0, b2 = a0;
    f32x4 acc[10];
    for (int i = 0; i < 10; ++i) acc[i] = (f32x4){0, 0, 0, 0};
    float xv = lane * 0.001f;
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
        if (MODE == 0) {
#pragma unroll
            for (int r = 0; r < 6; ++r)
#pragma unroll
                for (int i = 0; i < 10; ++i) acc[i] = MFMA(a0, b0, acc[i]);
        } else {
            if (MODE >= 3) {           // the split's VALU work: ~44 ops per row tile, two row tiles
#pragma unroll
                for (int q = 0; q < 44; ++q) xv = xv * 1.0001f + 0.5f;
                a1 = __builtin_bit_cast(bf16x8, (u32x4){__builtin_bit_cast(unsigned int, xv), 1, 2, 3});
            }
#pragma unroll
            for (int nt = 0; nt < 5; ++nt) {
                if (MODE >= 2) {
                    const u32x4* p = reinterpret_cast<const u32x4*>(lds) + ((nt * 64 + lane + it) & 1023);
                    b0 = __builtin_bit_cast(bf16x8, p[0]);
                    b1 = __builtin_bit_cast(bf16x8, p[1024]);
                    b2 = __builtin_bit_cast(bf16x8, p[2048]);
                }
                f32x4 c0 = acc[2 * nt], c1 = acc[2 * nt + 1];
                c0 = MFMA(a0, b2, c0); c1 = MFMA(a1, b2, c1);
                c0 = MFMA(a2, b0, c0); c1 = MFMA(a2, b0, c1);
                c0 = MFMA(a1, b1, c0); c1 = MFMA(a1, b1, c1);
                c0 = MFMA(a0, b1, c0); c1 = MFMA(a0, b1, c1);
                c0 = MFMA(a1, b0, c0); c1 = MFMA(a1, b0, c1);
                c0 = MFMA(a0, b0, c0); c1 = MFMA(a0, b0, c1);
                acc[2 * nt] = c0; acc[2 * nt + 1] = c1;
            }
        }
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    float s = 0;
    for (int i = 0; i < 10; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    out[blockIdx.x * 256 + threadIdx.x] = s + xv;
    if (threadIdx.x == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
}

template <int MODE>
static void run(int blocks_per_cu, const char* name) {
    float* out; unsigned long long* cyc;
    hipMalloc(&out, 256 * 2048 * 4); hipMalloc(&cyc, 8);
    const int iters = 2000;
    probe<MODE><<<256 * blocks_per_cu, 256>>>(out, 10, cyc);
    hipDeviceSynchronize();
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0);
    probe<MODE><<<256 * blocks_per_cu, 256>>>(out, iters, cyc);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    unsigned long long c; hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
    const double mf = 60.0 * iters;
    printf("%-34s %d wave(s)/SIMD: %6.1f cycles per MFMA per wave, %7.1f TFLOP/s bf16 chip-wide (%.3f ms)\n", name, blocks_per_cu,
           (double)c / mf, mf * 256.0 * blocks_per_cu * 4 * 16384.0 / (ms * 1e-3) / 1e12, ms);
    hipFree(out); hipFree(cyc);
}

int main() {
    for (int w = 1; w <= 2; ++w) {
        run<0>(w, "bare, 10 independent accumulators");
        run<1>(w, "6-deep chains, 2 interleaved");
        run<2>(w, "+ 3 ds_read_b128 per 12 MFMAs");
        run<3>(w, "+ split VALU per k step");
    }
    return 0;
}
