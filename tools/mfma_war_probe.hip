// tools/mfma_war_probe.hip -- can an asynchronous load overwrite a source register of an MFMA that has already been ISSUED but, with
// another wave sharing the SIMD's matrix pipe, has not yet read its operands?
// Each wave: B operand Rb <- LDS (good values), wait; N back-to-back MFMAs reading Rb (independent accumulators, or one dependent
// chain); IMMEDIATELY afterwards a load (LDS / global) or a VALU move writes different values into the SAME register Rb; long pad;
// the N results are compared with a * good.  A result equal to a * evil (or a mix) means that MFMA read Rb after the overwrite.
// 512-thread workgroups = two waves per SIMD, 256 = one.
// Build: hipcc --offload-arch=gfx950 -O3 tools/mfma_war_probe.hip -o tools/mfma_war_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ unsigned int pk(float a, float b) {
    typedef __bf16 pk2_t __attribute__((ext_vector_type(2)));
    const pk2_t v = {(__bf16)a, (__bf16)b};
    return __builtin_bit_cast(unsigned int, v);
}
#define PAD8() asm volatile("s_nop 15\n s_nop 15\n s_nop 15\n s_nop 15\n s_nop 15\n s_nop 15\n s_nop 15\n s_nop 15" ::: "memory")

// MODE 0: LDS load overwrites Rb; 1: VALU move overwrites Rb; 2: global load overwrites Rb; CHAIN: the N MFMAs are one dependent chain
template <int N, int THREADS, int MODE, bool CHAIN>
__global__ __launch_bounds__(THREADS) void probe(const u32x4* __restrict__ gsrc, unsigned int* __restrict__ bad, int iters) {
    __shared__ u32x4 tab[2][64];
    const int lane = threadIdx.x & 63;
    if (threadIdx.x < 64) {
        const float g = (float)(lane % 7) - 3.f, e = (float)(lane % 5) + 4.f;
        tab[0][lane] = (u32x4){pk(g, 1.f), pk(2.f, g), pk(-1.f, 3.f), pk(g, g)};
        tab[1][lane] = (u32x4){pk(e, 9.f), pk(8.f, e), pk(7.f, -9.f), pk(e, 6.f)};
    }
    __syncthreads();
    const float af = (float)(lane % 3) + 1.f;
    u32x4 a = (u32x4){pk(af, 1.f), pk(1.f, 2.f), pk(af, af), pk(0.5f, 1.f)};
    asm volatile("" : "+v"(a));
    // reference through the compiler's own (hazard-checked) path
    const f32x4 ref1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, tab[0][lane]),
                                                              (f32x4){0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
    const unsigned int good_addr = (unsigned int)(size_t)&tab[0][lane], evil_addr = (unsigned int)(size_t)&tab[1][lane];
    const u32x4* gevil = gsrc + lane;
    u32x4 evil_reg = tab[1][lane];
    asm volatile("" : "+v"(evil_reg));
    unsigned int nbad[N];
#pragma unroll
    for (int i = 0; i < N; ++i) nbad[i] = 0;
    for (int it = 0; it < iters; ++it) {
        u32x4 rb;
        asm volatile("ds_read_b128 %0, %1\n s_waitcnt lgkmcnt(0)" : "=v"(rb) : "v"(good_addr) : "memory");
        f32x4 c[N];
        if (CHAIN) {
            f32x4 cc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int i = 0; i < N; ++i) {
                asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(cc) : "v"(a), "v"(rb));
            }
            c[0] = cc;
        } else {
#pragma unroll
            for (int i = 0; i < N; ++i) asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, 0" : "=&v"(c[i]) : "v"(a), "v"(rb));
        }
        if (MODE == 0) asm volatile("ds_read_b128 %0, %1" : "+v"(rb) : "v"(evil_addr) : "memory");
        if (MODE == 1) asm volatile("v_mov_b32 %0, %4\n v_mov_b32 %1, %5\n v_mov_b32 %2, %6\n v_mov_b32 %3, %7"
                                    : "+v"(rb[0]), "+v"(rb[1]), "+v"(rb[2]), "+v"(rb[3]) : "v"(evil_reg[0]), "v"(evil_reg[1]), "v"(evil_reg[2]), "v"(evil_reg[3]));
        if (MODE == 2) asm volatile("global_load_dwordx4 %0, %1, off" : "+v"(rb) : "v"(gevil) : "memory");
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" : "+v"(rb));
        PAD8(); PAD8();
        if (CHAIN) {
            asm volatile("" : "+v"(c[0]));
            const f32x4 want = ref1 * (float)N;
            nbad[0] += (c[0][0] != want[0]) | (c[0][1] != want[1]) | (c[0][2] != want[2]) | (c[0][3] != want[3]);
        } else {
#pragma unroll
            for (int i = 0; i < N; ++i) {
                asm volatile("" : "+v"(c[i]));
                nbad[i] += (c[i][0] != ref1[0]) | (c[i][1] != ref1[1]) | (c[i][2] != ref1[2]) | (c[i][3] != ref1[3]);
            }
        }
        asm volatile("" ::"v"(rb));
    }
#pragma unroll
    for (int i = 0; i < N; ++i)
        if (nbad[i]) atomicAdd(&bad[i], nbad[i]);
}

template <int N, int THREADS, int MODE, bool CHAIN>
static void run(const u32x4* gsrc, unsigned int* bad) {
    hipMemset(bad, 0, 64 * 4);
    const int iters = 2000, nwg = 256 * 2;
    hipLaunchKernelGGL((probe<N, THREADS, MODE, CHAIN>), dim3(nwg), dim3(THREADS), 0, 0, gsrc, bad, iters);
    unsigned int h[64];
    hipMemcpy(h, bad, 64 * 4, hipMemcpyDeviceToHost);
    const char* modes[3] = {"LDS load", "VALU move", "global load"};
    printf("%2d %s MFMAs, %d waves per SIMD, source overwritten by a %-11s: lane-level mismatches per MFMA position (of %lld each):", N,
           CHAIN ? "chained    " : "independent", THREADS / 256, modes[MODE], (long long)nwg * THREADS * iters);
    for (int i = 0; i < (CHAIN ? 1 : N); ++i) printf(" %u", h[i]);
    printf("\n");
    fflush(stdout);
}

int main() {
    unsigned int* bad; u32x4* gsrc;
    hipMalloc(&bad, 64 * 4); hipMalloc(&gsrc, 64 * 16);
    unsigned int hv[256];
    for (int i = 0; i < 256; ++i) hv[i] = 0x41004100u + i;      // bf16 pairs around 8.0: neither the good nor the evil LDS values
    hipMemcpy(gsrc, hv, sizeof(hv), hipMemcpyHostToDevice);
    run<1, 512, 0, false>(gsrc, bad);
    run<2, 512, 0, false>(gsrc, bad);
    run<4, 512, 0, false>(gsrc, bad);
    run<8, 512, 0, false>(gsrc, bad);
    run<12, 512, 0, false>(gsrc, bad);
    run<16, 512, 0, false>(gsrc, bad);
    run<8, 256, 0, false>(gsrc, bad);
    run<16, 256, 0, false>(gsrc, bad);
    run<6, 512, 0, true>(gsrc, bad);
    run<12, 512, 0, true>(gsrc, bad);
    run<8, 512, 1, false>(gsrc, bad);
    run<16, 512, 1, false>(gsrc, bad);
    run<8, 512, 2, false>(gsrc, bad);
    run<16, 512, 2, false>(gsrc, bad);
    run<16, 256, 2, false>(gsrc, bad);
    return 0;
}
