// tools/mfma_hazard_probe.hip -- does a VALU instruction that reads a v_mfma_f32_16x16x32_bf16 result (VGPR destination, 512-thread
// workgroups: two waves per SIMD) a few instructions after the MFMA always see the finished value?  Each wave runs chains of six
// dependent MFMAs on operands made by the VALU right before (as the bf16x3 kernels do), reads the result on the VALU immediately,
// and the same chain is recomputed with a long s_nop pad as the reference.  Counts mismatches.
// Build: hipcc --offload-arch=gfx950 -O3 tools/mfma_hazard_probe.hip -o tools/mfma_hazard_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ unsigned int rnd(unsigned int x) {
    x ^= x >> 16; x *= 0x7feb352dU; x ^= x >> 15; x *= 0x846ca68bU; x ^= x >> 16;
    return x;
}
__device__ __forceinline__ unsigned int pk(float a, float b) {
    typedef __bf16 pk2_t __attribute__((ext_vector_type(2)));
    const pk2_t v = {(__bf16)a, (__bf16)b};
    return __builtin_bit_cast(unsigned int, v);
}

template <int PAD>
__global__ __launch_bounds__(512, 1) void probe(unsigned int* bad, float* sink, int iters) {
    const int tid = blockIdx.x * 512 + threadIdx.x;
    unsigned int nbad = 0;
    float keep = 0.f;
    for (int it = 0; it < iters; ++it) {
        // operands made by VALU instructions right in front of the MFMAs
        float f[16];
#pragma unroll
        for (int i = 0; i < 16; ++i) f[i] = (float)(int)(rnd(tid * 131 + it * 17 + i) & 0xff) * 0.0078125f - 1.0f;
        bf16x8 a = __builtin_bit_cast(bf16x8, (u32x4){pk(f[0], f[1]), pk(f[2], f[3]), pk(f[4], f[5]), pk(f[6], f[7])});
        bf16x8 b = __builtin_bit_cast(bf16x8, (u32x4){pk(f[8], f[9]), pk(f[10], f[11]), pk(f[12], f[13]), pk(f[14], f[15])});
        f32x4 c0 = {0.f, 0.f, 0.f, 0.f}, c1 = c0, r0 = c0, r1 = c0;
        // test chains: two accumulators interleaved (as two row tiles), results read by the VALU right after
#pragma unroll
        for (int k = 0; k < 6; ++k) {
            c0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c0, 0, 0, 0);
            c1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(b, a, c1, 0, 0, 0);
        }
        if (PAD > 0) {
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int p = 0; p < PAD; ++p) asm volatile("s_nop 15" ::: "memory");
            __builtin_amdgcn_sched_barrier(0);
        }
        const float t0 = (c0[0] + c0[1]) + (c0[2] + c0[3]), t1 = (c1[0] + c1[1]) + (c1[2] + c1[3]);
        // reference: the same chains on opaque copies of the operands (no common-subexpression elimination), read after a long pad
        __builtin_amdgcn_sched_barrier(0);
        u32x4 a2 = __builtin_bit_cast(u32x4, a), b2 = __builtin_bit_cast(u32x4, b);
        asm volatile("" : "+v"(a2), "+v"(b2));
        const bf16x8 ar = __builtin_bit_cast(bf16x8, a2), br = __builtin_bit_cast(bf16x8, b2);
#pragma unroll
        for (int k = 0; k < 6; ++k) {
            r0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ar, br, r0, 0, 0, 0);
            r1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(br, ar, r1, 0, 0, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int p = 0; p < 8; ++p) asm volatile("s_nop 15" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
        const float u0 = (r0[0] + r0[1]) + (r0[2] + r0[3]), u1 = (r1[0] + r1[1]) + (r1[2] + r1[3]);
        nbad += (t0 != u0) + (t1 != u1);
        keep += t0 + t1;
    }
    if (nbad) atomicAdd(bad, nbad);
    sink[tid] = keep;
}

template <int PAD>
static void run(unsigned int* bad, float* sink) {
    hipMemset(bad, 0, 4);
    hipLaunchKernelGGL(probe<PAD>, dim3(256 * 2), dim3(512), 0, 0, bad, sink, 2000);
    unsigned int h = 0;
    hipMemcpy(&h, bad, 4, hipMemcpyDeviceToHost);
    printf("pad %d x s_nop 15 between the last MFMA and the VALU read: %u mismatching reads of %lld\n", PAD, h, 256LL * 2 * 512 * 2000 * 2);
}

int main() {
    unsigned int* bad; float* sink;
    hipMalloc(&bad, 4); hipMalloc(&sink, 4 * 256 * 2 * 512);
    run<0>(bad, sink); run<1>(bad, sink); run<4>(bad, sink);
    run<0>(bad, sink);
    return 0;
}
