// tools/lds_raw_probe.hip -- is a register written by ds_read_b128 always complete when s_waitcnt lgkmcnt(N) lets the wave go on?
// Pattern taken from a failing build of din_wave.hip: VALU writes zeros into R, a burst of MFMAs, two ds_read_b128 (R, then S),
// s_waitcnt lgkmcnt(1), and IMMEDIATELY a v_pk_add_f32 whose two halves both read R's first dword.  Counts halves that saw the zero.
// Build: hipcc --offload-arch=gfx950 -O3 tools/lds_raw_probe.hip -o tools/lds_raw_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

template <int THREADS, int NMFMA, int WAIT>
__global__ __launch_bounds__(THREADS) void probe(unsigned int* __restrict__ bad, int iters) {
    __shared__ f32x4 tab[2][64];
    const int lane = threadIdx.x & 63;
    if (threadIdx.x < 64) {
        tab[0][lane] = (f32x4){1.5f + lane, 2.f, 3.f, 4.f};
        tab[1][lane] = (f32x4){5.f, 6.f, 7.f, 8.f};
    }
    __syncthreads();
    const unsigned int a_addr = (unsigned int)(size_t)&tab[0][lane], b_addr = (unsigned int)(size_t)&tab[1][lane];
    u32x4 ma = {0x3f803f80u, 0x3f803f80u, 0x3f803f80u, 0x3f803f80u}, mb = ma;
    asm volatile("" : "+v"(ma), "+v"(mb));
    f32x4 acc[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) acc[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
    unsigned int lo_bad = 0, hi_bad = 0;
    const float want = 1.5f + lane;
    for (int it = 0; it < iters; ++it) {
        f32x4 s;
        f32x2 d = {0.f, 0.f};
        // R = v[100:103] by name: the packed add reads its first register pair
        asm volatile("v_mov_b32 v100, 0\n v_mov_b32 v101, 0\n v_mov_b32 v102, 0\n v_mov_b32 v103, 0" ::: "v100", "v101", "v102", "v103");
#pragma unroll
        for (int i = 0; i < NMFMA; ++i)
            asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(acc[i & 3]) : "v"(ma), "v"(mb));
        if (WAIT == 1)
            asm volatile("ds_read_b128 v[100:103], %2\n ds_read_b128 %0, %3\n s_waitcnt lgkmcnt(1)\n v_pk_add_f32 %1, %1, v[100:101] op_sel_hi:[1,0]"
                         : "=&v"(s), "+v"(d) : "v"(a_addr), "v"(b_addr) : "memory", "v100", "v101", "v102", "v103");
        else
            asm volatile("ds_read_b128 v[100:103], %2\n ds_read_b128 %0, %3\n s_waitcnt lgkmcnt(0)\n v_pk_add_f32 %1, %1, v[100:101] op_sel_hi:[1,0]"
                         : "=&v"(s), "+v"(d) : "v"(a_addr), "v"(b_addr) : "memory", "v100", "v101", "v102", "v103");
        asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(s));
        lo_bad += d[0] != want;
        hi_bad += d[1] != want;
    }
    asm volatile("s_nop 15\n s_nop 15\n s_nop 15\n s_nop 15" ::: "memory");
    if (lo_bad) atomicAdd(&bad[0], lo_bad);
    if (hi_bad) atomicAdd(&bad[1], hi_bad);
    if (acc[0][0] + acc[1][0] + acc[2][0] + acc[3][0] == 12345.f) atomicAdd(&bad[2], 1u);
}

template <int THREADS, int NMFMA, int WAIT>
static void run(unsigned int* bad) {
    (void)hipMemset(bad, 0, 16);
    const int iters = 4000, nwg = 512;
    hipLaunchKernelGGL((probe<THREADS, NMFMA, WAIT>), dim3(nwg), dim3(THREADS), 0, 0, bad, iters);
    unsigned int h[4];
    (void)hipMemcpy(h, bad, 16, hipMemcpyDeviceToHost);
    printf("%d waves per SIMD, %2d MFMAs in front, s_waitcnt lgkmcnt(%d): stale low half %u, stale high half %u (of %lld lane-reads each)\n",
           THREADS / 256, NMFMA, WAIT, h[0], h[1], (long long)nwg * THREADS * iters);
    fflush(stdout);
}

int main() {
    unsigned int* bad;
    (void)hipMalloc(&bad, 16);
    run<512, 0, 1>(bad); run<512, 4, 1>(bad); run<512, 12, 1>(bad); run<512, 24, 1>(bad);
    run<256, 12, 1>(bad); run<512, 12, 0>(bad); run<1024, 12, 1>(bad);
    return 0;
}
