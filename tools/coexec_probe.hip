// tools/coexec_probe.hip -- does a SIMD run one wave's VALU instructions while the other wave's MFMAs are in the matrix pipe?
// 512-thread workgroups, one per CU: waves 0..3 (first wave of each SIMD) run a loop of v_mfma_f32_16x16x32_bf16, waves 4..7 a loop of
// VALU instructions of one kind; three launches per kind -- matrix waves alone, vector waves alone, both -- timed with HIP events.
// Overlap = (t_mfma + t_valu - t_both) / min(t_mfma, t_valu).
// Build: hipcc --offload-arch=gfx950 -O3 tools/coexec_probe.hip -o tools/coexec_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

template <int KIND, int CHAIN, int PRIO = 0, int SHAPE = 16>      // SHAPE: 16 = v_mfma_f32_16x16x32_bf16 (4 passes), 32 = v_mfma_f32_32x32x16_bf16 (8 passes; round 5); PRIO: s_setprio of the VALU waves (round 5); KIND: 0 v_fma_f32, 1 v_cvt_pk_bf16_f32, 2 v_pk_add_f32, 3 integer shift/and, 4 v_sub_f32; CHAIN: MFMAs on 1 | 4 accumulators
__global__ __launch_bounds__(512) void probe(float* sink, int iters, int run_mfma, int run_valu) {
    const int wave = threadIdx.x >> 6;
    float keep = 0.f;
    if (wave < 4) {
        if (!run_mfma) return;
        u32x4 ma = {0x3f803f80u, 0x3f803f80u, 0x3f803f80u, 0x3f803f80u}, mb = ma;
        asm volatile("" : "+v"(ma), "+v"(mb));
        f32x4 acc[4];
        typedef float f32x16 __attribute__((ext_vector_type(16)));
        f32x16 big[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            acc[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int e = 0; e < 16; ++e) big[i][e] = 0.f;
        }
        for (int it = 0; it < iters; ++it) {
            if constexpr (SHAPE == 16) {
#pragma unroll
                for (int i = 0; i < 24; ++i)
                    asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(acc[CHAIN == 1 ? 0 : (i & 3)]) : "v"(ma), "v"(mb));
            } else {
#pragma unroll
                for (int i = 0; i < 12; ++i)          // the same flops per iteration: 12 x (32 x 32 x 16)
                    asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(big[CHAIN == 1 ? 0 : (i & 3)]) : "v"(ma), "v"(mb));
            }
        }
        asm volatile("s_nop 15\n s_nop 15" ::: "memory");
        keep = acc[0][0] + acc[1][0] + acc[2][0] + acc[3][0] + big[0][0] + big[1][0] + big[2][0] + big[3][0];
    } else {
        if (!run_valu) return;
        if (PRIO == 1) asm volatile("s_setprio 1");
        if (PRIO == 3) asm volatile("s_setprio 3");
        float x[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) x[i] = (float)(threadIdx.x + i) * 0.001f;
        float y = 1.0001f, z = 0.5f;
        asm volatile("" : "+v"(y), "+v"(z));
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int r = 0; r < 12; ++r) {
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    if (KIND == 0) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x[i]) : "v"(y), "v"(z));
                    if (KIND == 1) asm volatile("v_cvt_pk_bf16_f32 %0, %0, %1" : "+v"(x[i]) : "v"(y));
                    if (KIND == 3) asm volatile("v_lshlrev_b32 %0, 1, %0" : "+v"(x[i]));
                    if (KIND == 4) asm volatile("v_sub_f32 %0, %0, %1" : "+v"(x[i]) : "v"(z));
                }
                if (KIND == 2) {
#pragma unroll
                    for (int i = 0; i < 8; i += 2) {
                        f32x2 v = {x[i], x[i + 1]}, w = {y, z};
                        asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(v) : "v"(w));
                        x[i] = v[0]; x[i + 1] = v[1];
                    }
                }
            }
        }
#pragma unroll
        for (int i = 0; i < 8; ++i) keep += x[i];
    }
    if (keep == 12345.678f) sink[threadIdx.x] = keep;
}

template <int KIND, int CHAIN, int PRIO = 0, int SHAPE = 16>
static void run(float* sink) {
    const int iters = 4000;
    float ms[3];
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    const int cfg[3][2] = {{1, 0}, {0, 1}, {1, 1}};
    for (int c = 0; c < 3; ++c) {
        hipLaunchKernelGGL((probe<KIND, CHAIN, PRIO, SHAPE>), dim3(256), dim3(512), 0, 0, sink, 100, cfg[c][0], cfg[c][1]);
        (void)hipEventRecord(e0, 0);
        hipLaunchKernelGGL((probe<KIND, CHAIN, PRIO, SHAPE>), dim3(256), dim3(512), 0, 0, sink, iters, cfg[c][0], cfg[c][1]);
        (void)hipEventRecord(e1, 0);
        (void)hipEventSynchronize(e1);
        (void)hipEventElapsedTime(&ms[c], e0, e1);
    }
    const char* kinds[5] = {"v_fma_f32", "v_cvt_pk_bf16_f32", "v_pk_add_f32", "v_lshlrev_b32", "v_sub_f32"};
    const float mn = ms[0] < ms[1] ? ms[0] : ms[1];
    printf("%-18s (VALU waves at s_setprio %d, MFMA %dx%d) MFMAs on %d accumulator(s): matrix alone %.3f ms, vector alone %.3f ms, both %.3f ms -> %.0f %% of the shorter one overlapped\n",
           kinds[KIND], PRIO, SHAPE, SHAPE, CHAIN == 1 ? 1 : 4, ms[0], ms[1], ms[2], 100.f * (ms[0] + ms[1] - ms[2]) / mn);
    fflush(stdout);
}

int main() {
    float* sink;
    (void)hipMalloc(&sink, 4096);
    run<0, 4>(sink); run<1, 4>(sink); run<2, 4>(sink); run<3, 4>(sink); run<4, 4>(sink);
    run<0, 1>(sink); run<1, 1>(sink); run<4, 1>(sink);
    run<0, 4, 1>(sink); run<0, 4, 3>(sink); run<2, 4, 3>(sink); run<0, 1, 3>(sink);      // round 5: does priority give the VALU wave the slots?
    run<0, 4, 0, 32>(sink); run<0, 1, 0, 32>(sink); run<2, 4, 0, 32>(sink); run<1, 4, 0, 32>(sink);      // round 5: the 8-pass shape
    return 0;
}
