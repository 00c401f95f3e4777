// tools/bpermute_raw_probe.hip -- does ds_bpermute_b32 always see the value a VALU instruction wrote into its data register in the
// instruction right before it?  (din_wave.hip bf16x3 builds: the lane-group reduction of a score, `sp += __shfl_xor(sp, 16)`, compiled to
// v_pk_add_f32 v[98:99] ... ; ds_bpermute_b32 v100, idx, v98 ; ds_bpermute_b32 v101, idx, v99 -- the FIRST of the two came out with the
// register's previous contents in about half of the samples.)
// Each iteration: v[100:101] <- old values; some MFMAs; v_pk_add_f32 v[100:101] <- new values; ds_bpermute of v100 and of v101 right
// behind (NOPS s_nop in between); compare with the xor-16 lane's new values.
// Build: hipcc --offload-arch=gfx950 -O3 tools/bpermute_raw_probe.hip -o tools/bpermute_raw_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

template <int THREADS, int NMFMA, int NOPS, int PRODUCER>      // PRODUCER 0: v_pk_add_f32, 1: two v_add_f32, 2: v_pk_fma_f32, 3: v_fma_f32 x2
__global__ __launch_bounds__(THREADS) void probe(unsigned int* __restrict__ bad, int iters) {
    const int lane = threadIdx.x & 63;
    u32x4 ma = {0x3f803f80u, 0x3f803f80u, 0x3f803f80u, 0x3f803f80u}, mb = ma;
    asm volatile("" : "+v"(ma), "+v"(mb));
    f32x4 acc[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) acc[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
    unsigned int bad0 = 0, bad1 = 0;
    const unsigned int idx = (unsigned int)((lane ^ 16) * 4);
    for (int it = 0; it < iters; ++it) {
        f32x2 a = {(float)(it & 1023), (float)(it & 1023) + 0.25f};
        f32x2 b = {(float)lane, (float)lane + 0.5f};
        f32x2 one = {1.f, 1.f};
        asm volatile("" : "+v"(a), "+v"(b), "+v"(one));
        float g0, g1;
        asm volatile("v_mov_b32 v100, 0\n v_mov_b32 v101, 0" ::: "v100", "v101");
#pragma unroll
        for (int i = 0; i < NMFMA; ++i)
            asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(acc[i & 3]) : "v"(ma), "v"(mb));
#define TAIL "ds_bpermute_b32 %0, %4, v100\n ds_bpermute_b32 %1, %4, v101\n s_waitcnt lgkmcnt(0)"
#define OPS : "=&v"(g0), "=&v"(g1) : "v"(a), "v"(b), "v"(idx), "v"(one) : "memory", "v100", "v101"
        if (PRODUCER == 0) {
            if (NOPS == 0) asm volatile("v_pk_add_f32 v[100:101], %2, %3\n " TAIL OPS);
            if (NOPS == 1) asm volatile("v_pk_add_f32 v[100:101], %2, %3\n s_nop 0\n " TAIL OPS);
            if (NOPS == 2) asm volatile("v_pk_add_f32 v[100:101], %2, %3\n s_nop 1\n " TAIL OPS);
        } else if (PRODUCER == 1) {
            asm volatile("v_add_f32 v100, %2, %3\n v_add_f32 v101, %2, %3\n " TAIL
                         : "=&v"(g0), "=&v"(g1) : "v"(a[0]), "v"(b[0]), "v"(idx), "v"(one) : "memory", "v100", "v101");
        } else if (PRODUCER == 2) {
            asm volatile("v_pk_fma_f32 v[100:101], %2, %5, %3\n " TAIL OPS);
        }
        const float w0 = a[0] + (float)(lane ^ 16), w1 = PRODUCER == 1 ? w0 : a[1] + (float)(lane ^ 16) + 0.5f;
        bad0 += g0 != w0;
        bad1 += g1 != w1;
    }
    asm volatile("s_nop 15\n s_nop 15\n s_nop 15\n s_nop 15" ::: "memory");
    if (bad0) atomicAdd(&bad[0], bad0);
    if (bad1) atomicAdd(&bad[1], bad1);
    if (acc[0][0] + acc[1][0] + acc[2][0] + acc[3][0] == 12345.f) atomicAdd(&bad[2], 1u);
}

template <int THREADS, int NMFMA, int NOPS, int PRODUCER>
static void run(unsigned int* bad) {
    (void)hipMemset(bad, 0, 16);
    const int iters = 4000, nwg = 512;
    hipLaunchKernelGGL((probe<THREADS, NMFMA, NOPS, PRODUCER>), dim3(nwg), dim3(THREADS), 0, 0, bad, iters);
    unsigned int h[4];
    (void)hipMemcpy(h, bad, 16, hipMemcpyDeviceToHost);
    const char* prod[3] = {"v_pk_add_f32", "2 x v_add_f32", "v_pk_fma_f32"};
    printf("%d waves per SIMD, %2d MFMAs in front, producer %-13s, %d s_nop between: stale first bpermute %u, stale second %u (of %lld lane-reads each)\n",
           THREADS / 256, NMFMA, prod[PRODUCER], NOPS, h[0], h[1], (long long)nwg * THREADS * iters);
    fflush(stdout);
}

int main() {
    unsigned int* bad;
    (void)hipMalloc(&bad, 16);
    run<512, 0, 0, 0>(bad); run<512, 6, 0, 0>(bad); run<512, 12, 0, 0>(bad); run<256, 12, 0, 0>(bad);
    run<512, 12, 1, 0>(bad); run<512, 12, 2, 0>(bad); run<512, 12, 0, 1>(bad); run<512, 12, 0, 2>(bad);
    return 0;
}
