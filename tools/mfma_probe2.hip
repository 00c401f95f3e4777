// mfma_probe2.hip -- issue-cost characterisation around v_mfma_f32_32x32x2_f32 (development tool)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)
typedef float f32x16 __attribute__((ext_vector_type(16)));
constexpr int STEPS = 512;
#define SB __builtin_amdgcn_sched_barrier(0)
#define MF(i, A, B) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(A, B, acc[i], 0, 0, 0); SB;
#define VADD(x) asm volatile("v_add_f32 %0, %0, %1" : "+v"(x) : "v"(eps)); SB;
#define SNOP asm volatile("s_nop 0"); SB;
#define SADD asm volatile("s_add_u32 %0, %0, 1" : "+s"(sc)); SB;

template <int VAR>
__global__ __launch_bounds__(256, 1) void probe(const float* __restrict__ in, float* __restrict__ out, unsigned long long* cyc) {
    extern __shared__ float lds[];
    const int tid = threadIdx.x;
    f32x16 acc[8];
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int q = 0; q < 16; ++q) acc[i][q] = 0.f;
    float a0 = in[tid], a1 = in[tid + 256], b0 = in[tid + 512], b1 = in[tid + 768], b2 = in[tid + 100], b3 = in[tid + 200];
    float x0 = in[tid + 300], x1 = in[tid + 400], x2 = in[tid + 500], x3 = in[tid + 600], eps = 1e-9f;
    float m0 = a0 * x0, m1 = a1 * x1;
    unsigned sc = 0;
    unsigned long long t0, t1;
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0)::"memory");
#pragma unroll 2
    for (int s = 0; s < STEPS; ++s) {
        SB;
        if (VAR == 0) {
            MF(0, a0, b0) MF(1, a0, b1) MF(2, a0, b2) MF(3, a0, b3) MF(4, a1, b0) MF(5, a1, b1) MF(6, a1, b2) MF(7, a1, b3)
        } else if (VAR == 1) {          // 1 independent VALU after each group of 4
            MF(0, a0, b0) MF(1, a0, b1) MF(2, a0, b2) MF(3, a0, b3) VADD(x0) MF(4, a1, b0) MF(5, a1, b1) MF(6, a1, b2) MF(7, a1, b3) VADD(x1)
        } else if (VAR == 2) {          // s_nop after each group of 4
            MF(0, a0, b0) MF(1, a0, b1) MF(2, a0, b2) MF(3, a0, b3) SNOP MF(4, a1, b0) MF(5, a1, b1) MF(6, a1, b2) MF(7, a1, b3) SNOP
        } else if (VAR == 3) {          // SALU after each group of 4
            MF(0, a0, b0) MF(1, a0, b1) MF(2, a0, b2) MF(3, a0, b3) SADD MF(4, a1, b0) MF(5, a1, b1) MF(6, a1, b2) MF(7, a1, b3) SADD
        } else if (VAR == 4) {          // 4 independent VALU after each group of 4
            MF(0, a0, b0) MF(1, a0, b1) MF(2, a0, b2) MF(3, a0, b3) VADD(x0) VADD(x1) VADD(x2) VADD(x3)
            MF(4, a1, b0) MF(5, a1, b1) MF(6, a1, b2) MF(7, a1, b3) VADD(x0) VADD(x1) VADD(x2) VADD(x3)
        } else if (VAR == 5) {          // 1 VALU after EVERY mfma
            MF(0, a0, b0) VADD(x0) MF(1, a0, b1) VADD(x1) MF(2, a0, b2) VADD(x2) MF(3, a0, b3) VADD(x3)
            MF(4, a1, b0) VADD(x0) MF(5, a1, b1) VADD(x1) MF(6, a1, b2) VADD(x2) MF(7, a1, b3) VADD(x3)
        } else if (VAR == 6) {          // 8 VALU in one cluster per step
            MF(0, a0, b0) MF(1, a0, b1) MF(2, a0, b2) MF(3, a0, b3) MF(4, a1, b0) MF(5, a1, b1) MF(6, a1, b2) MF(7, a1, b3)
            VADD(x0) VADD(x1) VADD(x2) VADD(x3) VADD(x0) VADD(x1) VADD(x2) VADD(x3)
        } else if (VAR == 7) {          // A operand rewritten in place right after its last reader (WAR on in-flight MFMA)
            MF(0, m0, b0) MF(1, m0, b1) MF(2, m0, b2) MF(3, m0, b3)
            asm volatile("v_mul_f32 %0, %1, %2" : "=v"(m0) : "v"(a0), "v"(x0)); SB;
            MF(4, m1, b0) MF(5, m1, b1) MF(6, m1, b2) MF(7, m1, b3)
            asm volatile("v_mul_f32 %0, %1, %2" : "=v"(m1) : "v"(a1), "v"(x1)); SB;
        } else if (VAR == 8) {          // same products, written one group AHEAD into the other register (no WAR, no adjacent RAW)
            MF(0, m0, b0) MF(1, m0, b1) MF(2, m0, b2) MF(3, m0, b3)
            asm volatile("v_mul_f32 %0, %1, %2" : "=v"(x2) : "v"(a0), "v"(x0)); SB;
            MF(4, m1, b0) MF(5, m1, b1) MF(6, m1, b2) MF(7, m1, b3)
            asm volatile("v_mul_f32 %0, %1, %2" : "=v"(x3) : "v"(a1), "v"(x1)); SB;
            float t = m0; m0 = x2; x2 = t; t = m1; m1 = x3; x3 = t;
        } else if (VAR == 9) {          // 1 LDS read (independent) after each group of 4
            MF(0, a0, b0) MF(1, a0, b1) MF(2, a0, b2) MF(3, a0, b3)
            asm volatile("ds_read_b32 %0, %1" : "=v"(x2) : "v"(tid * 4)); SB;
            MF(4, a1, b0) MF(5, a1, b1) MF(6, a1, b2) MF(7, a1, b3)
            asm volatile("ds_read_b32 %0, %1 offset:1024" : "=v"(x3) : "v"(tid * 4)); SB;
            asm volatile("s_waitcnt lgkmcnt(0)"); SB;
        }
    }
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1)::"memory");
    float r = x0 + x1 + x2 + x3 + m0 + m1 + (float)sc + lds[tid];
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
    printf("%-60s %.2f cycles per MFMA\n", name, (double)c / (256.0 * 4) / (STEPS * 8.0));
}

int main() {
    float *in, *out; unsigned long long* cyc;
    CK(hipMalloc(&in, 4096 * 4)); CK(hipMalloc(&out, 256 * 256 * 4)); CK(hipMalloc(&cyc, 8));
    float h[4096]; for (int i = 0; i < 4096; ++i) h[i] = 0.001f * (i % 97) - 0.04f;
    CK(hipMemcpy(in, h, sizeof(h), hipMemcpyHostToDevice));
    run<0>("P0 bare", in, out, cyc);
    run<1>("P1 1 indep v_add per 4 MFMA", in, out, cyc);
    run<2>("P2 1 s_nop per 4 MFMA", in, out, cyc);
    run<3>("P3 1 s_add per 4 MFMA", in, out, cyc);
    run<4>("P4 4 indep v_add (cluster) per 4 MFMA", in, out, cyc);
    run<5>("P5 1 v_add after every MFMA", in, out, cyc);
    run<6>("P6 8 v_add in one cluster per 8 MFMA", in, out, cyc);
    run<7>("P7 A operand rewritten in place after last reader", in, out, cyc);
    run<8>("P8 A operand written one group ahead (double buffer)", in, out, cyc);
    run<9>("P9 1 ds_read per 4 MFMA + waitcnt per 8", in, out, cyc);
    return 0;
}
