// tools/bf16_shape_probe.hip -- which bf16 MFMA shape to build the bf16x3 CIN on: chip-wide TFLOP/s of dependent 6-deep chains (the
// six piece products of one accumulator) on RANDOM operands (the clock the chip holds depends on the data), operands in registers.
//   mode 0: v_mfma_f32_32x32x16_bf16, 8 accumulators / wave, 4 waves / CU     mode 1: v_mfma_f32_16x16x32_bf16, 32 accumulators, 4 waves / CU
//   mode 2: 32x32x16, 4 accumulators, 8 waves / CU                            mode 3: 16x16x32, 16 accumulators, 8 waves / CU
// Build: hipcc --offload-arch=gfx950 -O3 tools/bf16_shape_probe.hip -o tools/bf16_shape_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ unsigned int rnd(unsigned int x) {
    x ^= x >> 16; x *= 0x7feb352dU; x ^= x >> 15; x *= 0x846ca68bU; x ^= x >> 16;
    return x;
}
__device__ __forceinline__ bf16x8 rnd_op(unsigned int seed, int zero) {   // 8 random bf16 in +-[0.5, 2)
    u32x4 v;
    for (int i = 0; i < 4; ++i) {
        const unsigned int r = rnd(seed * 4 + i);
        v[i] = zero ? 0u : ((r & 0x80ff80ffu) | 0x3f003f00u);
    }
    return __builtin_bit_cast(bf16x8, v);
}

template <int MODE>
__global__ __launch_bounds__(MODE >= 2 ? 512 : 256) void probe(float* out, int iters, int zero) {
    const int tid = blockIdx.x * blockDim.x + threadIdx.x;
    bf16x8 a[3], b[3];
    for (int i = 0; i < 3; ++i) { a[i] = rnd_op(tid * 8 + i, zero); b[i] = rnd_op(tid * 8 + 4 + i, zero); }
    constexpr bool BIG = (MODE == 0 || MODE == 2);
    constexpr int NACC = MODE == 0 ? 8 : MODE == 1 ? 32 : MODE == 2 ? 4 : 16;
    f32x16 c32[BIG ? NACC : 1];
    f32x4 c16[BIG ? 1 : NACC];
    for (int i = 0; i < (BIG ? NACC : 1); ++i) for (int q = 0; q < 16; ++q) c32[i][q] = 0.f;
    for (int i = 0; i < (BIG ? 1 : NACC); ++i) for (int q = 0; q < 4; ++q) c16[i][q] = 0.f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < NACC; ++i) {
            if (BIG) {
                f32x16 c = c32[i];
                c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0], b[2], c, 0, 0, 0);
                c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[2], b[0], c, 0, 0, 0);
                c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[1], b[1], c, 0, 0, 0);
                c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0], b[1], c, 0, 0, 0);
                c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[1], b[0], c, 0, 0, 0);
                c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0], b[0], c, 0, 0, 0);
                c32[i] = c;
            } else {
                f32x4 c = c16[i];
                c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[0], b[2], c, 0, 0, 0);
                c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[2], b[0], c, 0, 0, 0);
                c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[1], b[1], c, 0, 0, 0);
                c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[0], b[1], c, 0, 0, 0);
                c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[1], b[0], c, 0, 0, 0);
                c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[0], b[0], c, 0, 0, 0);
                c16[i] = c;
            }
        }
        // rotate the operands so that consecutive iterations do not present identical inputs
        const bf16x8 t = a[0]; a[0] = a[1]; a[1] = a[2]; a[2] = t;
        const bf16x8 u = b[0]; b[0] = b[2]; b[2] = b[1]; b[1] = u;
    }
    float s = 0;
    for (int i = 0; i < (BIG ? NACC : 1); ++i) for (int q = 0; q < 16; ++q) s += c32[i][q];
    for (int i = 0; i < (BIG ? 1 : NACC); ++i) for (int q = 0; q < 4; ++q) s += c16[i][q];
    out[tid] = s;
}

template <int MODE>
static void run(float* out, int zero) {
    const int threads = MODE >= 2 ? 512 : 256;
    const int nacc = MODE == 0 ? 8 : MODE == 1 ? 32 : MODE == 2 ? 4 : 16;
    const double flop_per_mfma = 2.0 * 32 * 32 * 16;   // both shapes: 16384 MACs
    const int iters = MODE >= 2 ? 4000 : 2000;          // same MFMA count per SIMD in every mode
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    for (int rep = 0; rep < 3; ++rep) {
        hipEventRecord(e0);
        hipLaunchKernelGGL(probe<MODE>, dim3(256 * 4), dim3(threads), 0, 0, out, iters, zero);   // 4 rounds of one workgroup per CU
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms = 0;
        hipEventElapsedTime(&ms, e0, e1);
        const double mfmas = 256.0 * 4 * (threads / 64) * (double)iters * nacc * 6;
        if (rep == 2) printf("mode %d %s: %.3f ms  %.1f TFLOP/s bf16  (%.1f fp32-equivalent at 6 products)\n", MODE, zero ? "zeros " : "random", ms,
                             mfmas * flop_per_mfma / ms / 1e9, mfmas * flop_per_mfma / ms / 1e9 / 6);
    }
}

int main() {
    float* out;
    hipMalloc(&out, sizeof(float) * 256 * 4 * 512);
    for (int zero = 0; zero < 2; ++zero) {
        run<0>(out, zero); run<1>(out, zero); run<2>(out, zero); run<3>(out, zero);
    }
    return 0;
}
