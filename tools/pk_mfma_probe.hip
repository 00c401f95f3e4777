// tools/pk_mfma_probe.hip -- does a packed fp32 VALU instruction (v_pk_add_f32) issued between MFMAs always write both halves for all 64
// lanes?  (din_wave.hip bf16x3 builds with SLP vectorisation on: `v_pk_add_f32 v[138:139], v[138:139], v[252:253] op_sel:[0,1]` between two
// v_mfma_f32_16x16x32_bf16 lost its LOW-half result in lanes 48..63 in about half of the samples, two waves per SIMD.)
// Each iteration: NB MFMAs; v_mul_f32 x2 write v[10:11]; v_pk_add_f32 v[10:11] += v13 (both halves); NA MFMAs; pad; compare per lane.
// Build: hipcc --offload-arch=gfx950 -O3 tools/pk_mfma_probe.hip -o tools/pk_mfma_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

template <int THREADS, int KIND, int NB, int NA, int OPSEL, int MID = 0>
__global__ __launch_bounds__(THREADS) void probe(unsigned int* __restrict__ bad, int iters) {
    const int lane = threadIdx.x & 63;
    u32x4 ma = {0x3f803f80u, 0x3f803f80u, 0x3f803f80u, 0x3f803f80u}, mb = ma;
    float fa = 1.f, fb = 1.f;
    typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
    u32x2 h2a = {0x3f803f80u, 0x3f803f80u}, h2b = h2a;
    asm volatile("" : "+v"(ma), "+v"(mb), "+v"(fa), "+v"(fb), "+v"(h2a), "+v"(h2b));
    f32x4 acc[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) acc[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
    typedef float f32x16 __attribute__((ext_vector_type(16)));
    f32x16 acc16;
#pragma unroll
    for (int i = 0; i < 16; ++i) acc16[i] = 0.f;
    unsigned int lo_bad = 0, hi_bad = 0;
    for (int it = 0; it < iters; ++it) {
        float x = (float)((it * 7 + lane) & 1023), y = x + 0.5f, z = (float)(lane + 3) * 0.25f, zero = 0.f, r0, r1;
        asm volatile("" : "+v"(x), "+v"(y), "+v"(z), "+v"(zero));
#define MF(i) do { if (KIND == 0) asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(acc[(i) & 3]) : "v"(ma), "v"(mb)); \
                   else if (KIND == 1) asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, %0" : "+v"(acc[(i) & 3]) : "v"(fa), "v"(fb)); \
                   else if (KIND == 2) asm volatile("v_mfma_f32_16x16x16_bf16 %0, %1, %2, %0" : "+v"(acc[(i) & 3]) : "v"(h2a), "v"(h2b)); \
                   else if (KIND == 3) asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+v"(acc[(i) & 3]) : "v"(ma), "v"(mb)); \
                   else asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(acc16) : "v"(ma), "v"(mb)); } while (0)
        const bool split_roles = MID == 9;
        const bool mfma_wave = !split_roles || (threadIdx.x >> 8) != 0;      // wave-uniform
        const bool pk_wave = !split_roles || (threadIdx.x >> 8) == 0;
        if (mfma_wave) {
#pragma unroll
            for (int i = 0; i < NB; ++i) MF(i);
        }
        if (!pk_wave) continue;
        // v10 <- x * 1, v11 <- y * 1 (as the kernel: two v_mul right in front), v13 <- z; then the packed add
#define PRE "v_mov_b32 v13, %2\n v_mov_b32 v12, %3\n v_mul_f32 v10, %0, 1.0\n v_mul_f32 v11, %1, 1.0\n "
#define PKOPS :: "v"(x), "v"(y), "v"(z), "v"(zero) : "v10", "v11", "v12", "v13", "v14"
        // OPSEL 1: op_sel:[0,1] (both results add src1's HIGH register v13 = z; v12 = 0); 0: plain (v12 = v13 = z);
        // 2: op_sel_hi:[1,0] (both results add src1's LOW register: v12 <- z); 3: v_pk_mul_f32 op_sel:[0,1] (z' = 1 + ...: multiply by v13)
        if (OPSEL == 1) asm volatile(PRE "v_pk_add_f32 v[10:11], v[10:11], v[12:13] op_sel:[0,1]" PKOPS);
        if (OPSEL == 0) asm volatile("v_mov_b32 v13, %2\n v_mov_b32 v12, %2\n v_mul_f32 v10, %0, 1.0\n v_mul_f32 v11, %1, 1.0\n v_pk_add_f32 v[10:11], v[10:11], v[12:13]" PKOPS);
        if (OPSEL == 2) asm volatile("v_mov_b32 v13, %3\n v_mov_b32 v12, %2\n v_mul_f32 v10, %0, 1.0\n v_mul_f32 v11, %1, 1.0\n v_pk_add_f32 v[10:11], v[10:11], v[12:13] op_sel_hi:[1,0]" PKOPS);
        if (OPSEL == 4) asm volatile(PRE "v_pk_add_f32 v[10:11], v[12:13], v[10:11] op_sel:[1,0]" PKOPS);
        // 5: neg_lo / neg_hi on src1 (the form of the bf16x3 split's subtraction): v12 = v13 = -z
        if (OPSEL == 5) asm volatile("v_sub_f32 v13, 0, %2\n v_sub_f32 v12, 0, %2\n v_mul_f32 v10, %0, 1.0\n v_mul_f32 v11, %1, 1.0\n v_pk_add_f32 v[10:11], v[10:11], v[12:13] neg_lo:[0,1] neg_hi:[0,1]" PKOPS);
        // 6: v_pk_fma_f32 with op_sel on the addend: v[10:11] = v[10:11] * 1 + v13
        if (OPSEL == 6) asm volatile("v_mov_b32 v13, %2\n v_mov_b32 v12, %3\n v_mul_f32 v10, %0, 1.0\n v_mul_f32 v11, %1, 1.0\n v_mov_b32 v14, 1.0\n v_mov_b32 v15, 1.0\n v_pk_fma_f32 v[10:11], v[10:11], v[14:15], v[12:13] op_sel:[0,0,1]" :: "v"(x), "v"(y), "v"(z), "v"(zero) : "v10", "v11", "v12", "v13", "v14", "v15");
        // 7: op_sel on src0 high -> low AND plain src1:  v[10:11] = (v11, v11) + (z, z): expected low = y + z
        if (MID == 1) asm volatile("s_nop 0");
        if (MID == 2) asm volatile("v_mov_b32 v14, 0" ::: "v14");
        if (MID == 3) asm volatile("s_nop 1");
        if (MID == 4) asm volatile("s_nop 3");
        if (mfma_wave) {
#pragma unroll
            for (int i = 0; i < NA; ++i) MF(i + NB);
        }
        asm volatile("s_nop 15\n s_nop 15\n s_nop 15\n s_nop 15\n v_mov_b32 %0, v10\n v_mov_b32 %1, v11" : "=v"(r0), "=v"(r1) :: "v10", "v11");
        lo_bad += r0 != x + z;
        hi_bad += r1 != y + z;
    }
    asm volatile("s_nop 15\n s_nop 15\n s_nop 15\n s_nop 15" ::: "memory");
    if (lo_bad) atomicAdd(&bad[lane >> 4], lo_bad);
    if (hi_bad) atomicAdd(&bad[4 + (lane >> 4)], hi_bad);
    if (acc[0][0] + acc[1][0] + acc[2][0] + acc[3][0] + acc16[0] + acc16[15] == 12345.f) atomicAdd(&bad[8], 1u);
}

template <int THREADS, int KIND, int NB, int NA, int OPSEL, int MID = 0>
static void run(unsigned int* bad) {
    (void)hipMemset(bad, 0, 64);
    const int iters = 4000, nwg = 512;
    hipLaunchKernelGGL((probe<THREADS, KIND, NB, NA, OPSEL, MID>), dim3(nwg), dim3(THREADS), 0, 0, bad, iters);
    unsigned int h[16];
    (void)hipMemcpy(h, bad, 64, hipMemcpyDeviceToHost);
    printf("%d waves/SIMD, %s MFMAs: %d before, %d after, form %d, between pk and MFMA %d: wrong LOW halves by lane group [%u %u %u %u], wrong HIGH halves [%u %u %u %u] (of %lld per group)\n",
           THREADS / 256, KIND == 0 ? "bf16 16x16x32" : KIND == 1 ? "f32 16x16x4  " : KIND == 2 ? "bf16 16x16x16" : KIND == 3 ? "f16 16x16x32 " : "bf16 32x32x16", NB, NA, OPSEL, MID, h[0], h[1], h[2], h[3], h[4], h[5], h[6], h[7], (long long)nwg * THREADS / 4 * iters);
    fflush(stdout);
}

int main() {
    unsigned int* bad;
    (void)hipMalloc(&bad, 64);
    run<512, 0, 2, 2, 1>(bad);
    run<512, 0, 8, 0, 1, 9>(bad);
    run<512, 4, 8, 0, 1, 9>(bad);
    run<512, 1, 8, 0, 1, 9>(bad);
    run<512, 0, 8, 0, 2, 9>(bad);
    run<512, 0, 8, 0, 6, 9>(bad);
    return 0;
}
