// mfma_probe.hip -- what hides behind v_mfma_f32_32x32x2_f32?  (development tool, not product)
// One wave per SIMD (256-thread workgroups, 1 per CU via a 100 KB LDS request), 8 independent accumulators,
// STEPS steps of 8 MFMAs; variants add per-step side work.  Prints shader cycles per MFMA (s_memtime).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)
typedef float f32x16 __attribute__((ext_vector_type(16)));
constexpr int STEPS = 512;

template <int VAR>
__global__ __launch_bounds__(256, 1) void probe(const float* __restrict__ in, float* __restrict__ out, unsigned long long* cyc) {
    extern __shared__ float lds[];
    const int tid = threadIdx.x;
    for (int i = tid; i < 24 * 1024; i += 256) lds[i] = in[i & 1023];
    __syncthreads();
    f32x16 acc[8];
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int q = 0; q < 16; ++q) acc[i][q] = 0.f;
    float a0 = in[tid], a1 = in[tid + 256], b0 = in[tid + 512], b1 = in[tid + 768], b2 = in[tid + 100], b3 = in[tid + 200];
    float x0 = in[tid + 300], x1 = in[tid + 400];
    const float* lp = lds + (tid & 63);
    unsigned long long t0, t1;
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0)::"memory");
#pragma unroll 4
    for (int s = 0; s < STEPS; ++s) {
        float bn0 = b0, bn1 = b1, bn2 = b2, bn3 = b3;
        if (VAR >= 2) {   // operand reads for the next step, as the CIN loop does
            const float* p = lp + ((s & 63) * 256);
            bn0 = p[0]; bn1 = p[32]; bn2 = p[64]; bn3 = p[96];
        }
        if (VAR >= 4) {   // a couple of staging-like LDS stores
            lds[16 * 1024 + ((s & 31) * 256) + tid] = x0;
        }
        if (VAR >= 3) __builtin_amdgcn_sched_barrier(0);
        float m0 = a0, m1 = a1;
        if (VAR >= 1) { m0 = a0 * x0; m1 = a1 * x1; }
        acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(m0, b0, acc[0], 0, 0, 0);
        acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(m0, b1, acc[1], 0, 0, 0);
        acc[2] = __builtin_amdgcn_mfma_f32_32x32x2f32(m0, b2, acc[2], 0, 0, 0);
        acc[3] = __builtin_amdgcn_mfma_f32_32x32x2f32(m0, b3, acc[3], 0, 0, 0);
        acc[4] = __builtin_amdgcn_mfma_f32_32x32x2f32(m1, b0, acc[4], 0, 0, 0);
        acc[5] = __builtin_amdgcn_mfma_f32_32x32x2f32(m1, b1, acc[5], 0, 0, 0);
        acc[6] = __builtin_amdgcn_mfma_f32_32x32x2f32(m1, b2, acc[6], 0, 0, 0);
        acc[7] = __builtin_amdgcn_mfma_f32_32x32x2f32(m1, b3, acc[7], 0, 0, 0);
        if (VAR >= 3) __builtin_amdgcn_sched_barrier(0);
        b0 = bn0; b1 = bn1; b2 = bn2; b3 = bn3;
        if (VAR >= 1) { x0 += 1e-9f; x1 -= 1e-9f; }
    }
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1)::"memory");
    float r = 0.f;
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int q = 0; q < 16; ++q) r += acc[i][q];
    out[blockIdx.x * 256 + tid] = r;
    if ((tid & 63) == 0) atomicAdd(cyc, t1 - t0);
}

template <int VAR> static void run(const char* name, float* in, float* out, unsigned long long* cyc) {
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(&probe<VAR>), hipFuncAttributeMaxDynamicSharedMemorySize, 100 * 1024));
    for (int it = 0; it < 3; ++it) {
        CK(hipMemset(cyc, 0, 8));
        hipLaunchKernelGGL((probe<VAR>), dim3(256), dim3(256), 100 * 1024, 0, in, out, cyc);
        CK(hipDeviceSynchronize());
    }
    unsigned long long c; CK(hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost));
    printf("%-44s %.2f cycles per MFMA\n", name, (double)c / (256.0 * 4) / (STEPS * 8.0));
}

int main() {
    float *in, *out; unsigned long long* cyc;
    CK(hipMalloc(&in, 4096 * 4)); CK(hipMalloc(&out, 256 * 256 * 4)); CK(hipMalloc(&cyc, 8));
    float h[4096]; for (int i = 0; i < 4096; ++i) h[i] = 0.001f * (i % 97) - 0.04f;
    CK(hipMemcpy(in, h, sizeof(h), hipMemcpyHostToDevice));
    run<0>("V0 bare 8 MFMAs per step", in, out, cyc);
    run<1>("V1 + 2 v_mul (+2 v_add) per step", in, out, cyc);
    run<2>("V2 + 4 LDS operand reads per step", in, out, cyc);
    run<3>("V3 + sched_barrier pinning (reads first)", in, out, cyc);
    run<4>("V4 + 1 LDS store per step", in, out, cyc);
    return 0;
}
