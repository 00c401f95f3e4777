// tools/valu_rate_probe.hip (GPU box) -- issue cost, in cycles per wave instruction, of the VALU instructions the bf16x3 splits are
// made of, and whether VALU work of one wave overlaps the MFMAs of the other wave on the same SIMD.
//   hipcc --offload-arch=gfx950 -O2 -o tools/valu_rate_probe tools/valu_rate_probe.hip && tools/valu_rate_probe
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));

#define REP16(X) X X X X X X X X X X X X X X X X
template <int OP>
__global__ void rate_k(long long* out, int iters, float seed) {
    float a0 = seed + threadIdx.x, a1 = a0 * 1.5f, a2 = a0 + 2.f, a3 = a0 - 3.f;
    unsigned u0 = threadIdx.x * 2654435761u, u1 = u0 ^ 0x55aa55aau, u2 = u0 + 77u, u3 = u1 + 99u;
    f32x2 p0 = {a0, a1}, p1 = {a2, a3}, p2 = {a1, a2}, p3 = {a3, a0};
    long long t0 = clock64();
    for (int i = 0; i < iters; ++i) {
        if (OP == 0) { REP16(asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2\n v_cvt_pk_bf16_f32 %3, %4, %5" : "=v"(u0), "+v"(a0), "+v"(a1), "=v"(u1), "+v"(a2), "+v"(a3));) }
        if (OP == 1) { REP16(asm volatile("v_pk_add_f32 %0, %1, %2\n v_pk_add_f32 %3, %1, %2" : "=v"(p0), "+v"(p1), "+v"(p2), "=v"(p3));) }
        if (OP == 2) { REP16(asm volatile("v_add_f32 %0, %1, %2\n v_add_f32 %3, %1, %2" : "=v"(a0), "+v"(a1), "+v"(a2), "=v"(a3));) }
        if (OP == 3) { REP16(asm volatile("v_perm_b32 %0, %1, %2, %3\n v_and_b32 %4, %1, %2" : "=v"(u0), "+v"(u1), "+v"(u2), "+v"(u3), "=v"(a0));) }
        if (OP == 4) { REP16(asm volatile("v_pk_mul_f32 %0, %1, %2\n v_pk_fma_f32 %3, %1, %2, %1" : "=v"(p0), "+v"(p1), "+v"(p2), "=v"(p3));) }
        if (OP == 5) { REP16(asm volatile("v_lshlrev_b32 %0, 16, %1\n v_add3_u32 %2, %1, %3, %3" : "=v"(u0), "+v"(u1), "=v"(u2), "+v"(u3));) }
    }
    long long t1 = clock64();
    if (threadIdx.x == 0 && blockIdx.x == 0) out[0] = t1 - t0;
    if (a0 + a1 + a2 + a3 + p0[0] + p3[1] == 12345.678f && u0 + u1 + u2 == 77u) out[1] = 1;
}

// two waves per SIMD (512 threads): waves 0..3 issue MFMAs, waves 4..7 issue VALU (mode 1), or nothing (mode 0), or MFMAs too (mode 2)
__global__ __launch_bounds__(512) void overlap_k(long long* out, int iters, int mode) {
    const int wave = threadIdx.x >> 6;
    bf16x8_t a = {}, b = {};
    f32x4 c0 = {0, 0, 0, 0}, c1 = c0, c2 = c0, c3 = c0;
    float a0 = threadIdx.x, a1 = 1.f, a2 = 2.f, a3 = 3.f;
    __syncthreads();
    long long t0 = clock64();
    if (wave < 4 || mode == 2) {
        for (int i = 0; i < iters; ++i) {
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                c0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c0, 0, 0, 0);
                c1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c1, 0, 0, 0);
                c2 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c2, 0, 0, 0);
                c3 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c3, 0, 0, 0);
            }
        }
    } else if (mode == 1) {
        for (int i = 0; i < iters; ++i) { REP16(asm volatile("v_add_f32 %0, %1, %2\n v_add_f32 %3, %1, %2" : "=v"(a0), "+v"(a1), "+v"(a2), "=v"(a3));) }
    }
    long long t1 = clock64();
    if ((threadIdx.x & 63) == 0 && blockIdx.x == 0) out[wave] = t1 - t0;
    if (c0[0] + c1[1] + c2[2] + c3[3] + a0 + a3 == 12345.678f) out[9] = 1;
}

int main() {
    long long* d; hipMalloc(&d, 128); long long h[16];
    const int iters = 2000;
    const char* names[] = {"v_cvt_pk_bf16_f32", "v_pk_add_f32", "v_add_f32", "v_perm_b32 / v_and_b32", "v_pk_mul_f32 / v_pk_fma_f32", "v_lshlrev_b32 / v_add3_u32"};
#define RUN(OP) hipLaunchKernelGGL(rate_k<OP>, dim3(1), dim3(64), 0, 0, d, iters, 1.0f); hipMemcpy(h, d, 128, hipMemcpyDeviceToHost); \
    printf("%-30s %.2f cycles (clock64 ticks) per instruction, one wave\n", names[OP], (double)h[0] / (iters * 32.0));
    RUN(0) RUN(1) RUN(2) RUN(3) RUN(4) RUN(5)
    for (int mode = 0; mode < 3; ++mode) {
        hipLaunchKernelGGL(overlap_k, dim3(1), dim3(512), 0, 0, d, iters, mode); hipMemcpy(h, d, 128, hipMemcpyDeviceToHost);
        printf("overlap mode %d (0: MFMA waves alone, 1: + VALU waves on the same SIMDs, 2: MFMA in all 8 waves): wave0 %lld ticks (%.2f per MFMA), wave4 %lld ticks (%.2f per VALU instr)\n",
               mode, h[0], (double)h[0] / (iters * 32.0), h[4], (double)h[4] / (iters * 32.0));
    }
    return 0;
}
