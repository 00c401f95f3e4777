// tools/mfma_valu_mix_probe.hip -- what NV plain VALU instructions cost when ONE wave issues them between its own v_mfma_f32_16x16x32_f16
// (round 5): a loop of 18 x (MFMA, NV x v_fma_f32) -- or tile-sized blocks (6 MFMAs, 6 NV fmas), v_pk_fma_f32, an LDS read per 3 MFMAs --, the MFMAs on one dependent chain or rotating over four accumulators, one or two waves
// per SIMD (256- / 512-thread workgroups, one per CU).  Prints cycles per MFMA slot (GPU time x 2.4 GHz / MFMAs per wave is NOT used: the
// ratio to NV = 0 is what matters).  Build: hipcc --offload-arch=gfx950 -O3 tools/mfma_valu_mix_probe.hip -o tools/mfma_valu_mix_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

template <int NV, int CHAIN, int NT, int PAT>
__global__ __launch_bounds__(NT) void probe(float* sink, int iters) {
    u32x4 ma = {0x3c003c00u, 0x3c003c00u, 0x3c003c00u, 0x3c003c00u}, mb = ma;
    asm volatile("" : "+v"(ma), "+v"(mb));
    f32x4 acc[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) acc[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
    float x[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) x[i] = (float)(threadIdx.x + i) * 0.001f;
    float y = 1.0001f, z = 0.5f;
    asm volatile("" : "+v"(y), "+v"(z));
    typedef float f32x2 __attribute__((ext_vector_type(2)));
    f32x2 xp[4], yp = {y, y}, zp = {z, z};
#pragma unroll
    for (int i = 0; i < 4; ++i) xp[i] = (f32x2){x[2 * i], x[2 * i + 1]};
    asm volatile("" : "+v"(yp), "+v"(zp));
    __shared__ u32x4 lds[NT];
    lds[threadIdx.x] = ma;
    __syncthreads();
    const unsigned la = (unsigned)(threadIdx.x * 16);
    u32x4 rd = ma;
    for (int it = 0; it < iters; ++it) {
        if constexpr (PAT == 1) {
#pragma unroll
            for (int t = 0; t < 18; t += 6) {
#pragma unroll
                for (int i = 0; i < 6; ++i)
                    asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+v"(acc[CHAIN == 1 ? 0 : ((t / 6) & 3)]) : "v"(ma), "v"(mb));
#pragma unroll
                for (int v = 0; v < 6 * NV; ++v) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x[v & 7]) : "v"(y), "v"(z));
            }
        } else {
#pragma unroll
            for (int i = 0; i < 18; ++i) {
                asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+v"(acc[CHAIN == 1 ? 0 : (i & 3)]) : "v"(ma), "v"(mb));
                if constexpr (PAT == 2) {
#pragma unroll
                    for (int v = 0; v < NV; ++v) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(xp[(i * NV + v) & 3]) : "v"(yp), "v"(zp));
                } else {
#pragma unroll
                    for (int v = 0; v < NV; ++v) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x[(i * NV + v) & 7]) : "v"(y), "v"(z));
                }
                if constexpr (PAT == 3)
                    if (i % 3 == 0) asm volatile("ds_read_b128 %0, %1" : "=v"(rd) : "v"(la));
            }
            if constexpr (PAT == 3) asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(rd));
        }
    }
    x[0] += __uint_as_float(rd[0]) + xp[0][0] + xp[1][1] + xp[2][0] + xp[3][1];
    asm volatile("s_nop 15\n s_nop 15" ::: "memory");
    float keep = acc[0][0] + acc[1][0] + acc[2][0] + acc[3][0];
#pragma unroll
    for (int i = 0; i < 8; ++i) keep += x[i];
    if (keep == 12345.678f) sink[threadIdx.x] = keep;
}

template <int NV, int CHAIN, int NT, int PAT>
static float run(float* sink) {
    const int iters = 4000;
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    hipLaunchKernelGGL((probe<NV, CHAIN, NT, PAT>), dim3(256), dim3(NT), 0, 0, sink, 100);
    (void)hipEventRecord(e0, 0);
    hipLaunchKernelGGL((probe<NV, CHAIN, NT, PAT>), dim3(256), dim3(NT), 0, 0, sink, iters);
    (void)hipEventRecord(e1, 0);
    (void)hipEventSynchronize(e1);
    float ms;
    (void)hipEventElapsedTime(&ms, e0, e1);
    return ms;
}

template <int CHAIN, int NT, int PAT>
static void row(float* sink) {
    const float t[6] = {run<0, CHAIN, NT, PAT>(sink), run<1, CHAIN, NT, PAT>(sink), run<2, CHAIN, NT, PAT>(sink), run<3, CHAIN, NT, PAT>(sink), run<4, CHAIN, NT, PAT>(sink),
                        run<6, CHAIN, NT, PAT>(sink)};
    static const char* pats[4] = {"v_fma_f32 after every MFMA", "6 MFMAs, then their 6 NV v_fma_f32", "v_pk_fma_f32 after every MFMA", "v_fma_f32 after every MFMA + a ds_read_b128 per 3 MFMAs"};
    printf("%s; %d wave(s) per SIMD, MFMAs on %d accumulator(s): NV = 0 / 1 / 2 / 3 / 4 / 6 VALU per MFMA: %.3f %.3f %.3f %.3f %.3f %.3f ms  (x %.2f %.2f %.2f %.2f %.2f of NV = 0)\n",
           pats[PAT], NT / 256, CHAIN == 1 ? 1 : 4, t[0], t[1], t[2], t[3], t[4], t[5], t[1] / t[0], t[2] / t[0], t[3] / t[0], t[4] / t[0], t[5] / t[0]);
    fflush(stdout);
}

int main() {
    float* sink;
    (void)hipMalloc(&sink, 4096);
    row<4, 256, 0>(sink); row<1, 256, 0>(sink); row<4, 512, 0>(sink); row<1, 512, 0>(sink);
    row<1, 256, 1>(sink); row<1, 512, 1>(sink); row<4, 512, 1>(sink);
    row<1, 256, 2>(sink); row<1, 512, 2>(sink);
    row<1, 256, 3>(sink); row<1, 512, 3>(sink);
    return 0;
}
