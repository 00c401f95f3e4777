// din_bwd_probe.hip -- cycles per phase of din_bwd_k (development tool, not product).
// build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -Iinclude [-DDIN_STAMP=1] tools/din_bwd_probe.hip \
//        details-in-recommendation_amd/csrc/capi.cpp -o tools/din_bwd_probe
#include "../details-in-recommendation_amd/csrc/din.hip"
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)

int main() {
    const int K = 64, T = 50, H1 = 80, H2 = 40;
    const int64_t B = 65536, V = 2000000;
    std::vector<float> ht((size_t)V * K), hw1(4 * K * H1), hw2(H1 * H2), hw3(H2), hg((size_t)B * K);
    unsigned s = 7u;
    auto rnd = [&]() { s = s * 1664525u + 1013904223u; return ((s >> 8) & 0xffff) / 65536.0f - 0.5f; };
    for (auto& v : ht) v = rnd() * 0.25f;
    for (auto& v : hw1) v = rnd() * 0.1f;
    for (auto& v : hw2) v = rnd() * 0.2f;
    for (auto& v : hw3) v = rnd() * 0.2f;
    for (auto& v : hg) v = rnd() * 0.02f;
    std::vector<int64_t> hh((size_t)B * T), hc(B), hoff(B);
    std::vector<int32_t> hl(B);
    for (auto& v : hh) { s = s * 1664525u + 1013904223u; v = (s >> 4) % V; }
    for (auto& v : hc) { s = s * 1664525u + 1013904223u; v = (s >> 4) % V; }
    const int fixed_len = getenv("DIN_PROBE_LEN") ? atoi(getenv("DIN_PROBE_LEN")) : 0;   // 0: uniform 1..T
    int64_t N = 0;
    for (int64_t b = 0; b < B; ++b) { s = s * 1664525u + 1013904223u; hl[b] = fixed_len ? fixed_len : 1 + (s >> 8) % T; hoff[b] = N; N += hl[b]; }
    float *table, *w1, *b1, *w2, *b2, *w3, *b3, *g, *gh, *ga, *S, *gAP, *gW2, *gb2, *gW3, *gb3; int64_t *hist, *cand, *off; int32_t* len; void* ws;
    CK(hipMalloc(&table, ht.size() * 4)); CK(hipMalloc(&w1, hw1.size() * 4)); CK(hipMalloc(&w2, hw2.size() * 4)); CK(hipMalloc(&w3, hw3.size() * 4));
    CK(hipMalloc(&b1, H1 * 4)); CK(hipMalloc(&b2, H2 * 4)); CK(hipMalloc(&b3, 4)); CK(hipMalloc(&g, hg.size() * 4));
    CK(hipMalloc(&gh, (size_t)N * K * 4)); CK(hipMalloc(&ga, B * K * 4)); CK(hipMalloc(&S, B * H1 * 4));
    CK(hipMalloc(&gAP, 2 * K * H1 * 4)); CK(hipMalloc(&gW2, H1 * H2 * 4)); CK(hipMalloc(&gb2, H2 * 4)); CK(hipMalloc(&gW3, H2 * 4)); CK(hipMalloc(&gb3, 4));
    CK(hipMalloc(&ws, dir_din_backward_workspace_bytes(K, H1, H2)));
    CK(hipMalloc(&hist, hh.size() * 8)); CK(hipMalloc(&cand, B * 8)); CK(hipMalloc(&off, B * 8)); CK(hipMalloc(&len, B * 4));
    CK(hipMemcpy(table, ht.data(), ht.size() * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(w1, hw1.data(), hw1.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(w2, hw2.data(), hw2.size() * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(w3, hw3.data(), hw3.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(g, hg.data(), hg.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemset(b1, 0, H1 * 4)); CK(hipMemset(b2, 0, H2 * 4)); CK(hipMemset(b3, 0, 4));
    CK(hipMemcpy(hist, hh.data(), hh.size() * 8, hipMemcpyHostToDevice)); CK(hipMemcpy(cand, hc.data(), B * 8, hipMemcpyHostToDevice));
    CK(hipMemcpy(off, hoff.data(), B * 8, hipMemcpyHostToDevice)); CK(hipMemcpy(len, hl.data(), B * 4, hipMemcpyHostToDevice));
    for (int it = 0; it < 3; ++it) {
#ifdef DIN_STAMP
        unsigned long long z[12] = {0};
        CK(hipMemcpyToSymbol(HIP_SYMBOL(dir::din_bwd_stamp), z, sizeof(z)));
#endif
        hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
        CK(hipEventRecord(e0, 0));
        if (dir_din_attention_pool_backward_f32(table, K, hist, len, cand, T, w1, b1, H1, w2, b2, H2, w3, b3, 1, B, g, off, gh, ga, S, gAP, gW2,
                                                gb2, gW3, gb3, ws, nullptr) != 0) {
            printf("error: %s\n", dir_last_error()); return 1;
        }
        CK(hipEventRecord(e1, 0)); CK(hipDeviceSynchronize());
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        printf("backward %.3f ms (N = %lld rows)", ms, (long long)N);
#ifdef DIN_STAMP
        CK(hipMemcpyFromSymbol(z, HIP_SYMBOL(dir::din_bwd_stamp), sizeof(z)));
        const double n = (double)z[6];
        printf(" | cycles per sample: stage %.0f  issue-loads %.0f  dw+recompute %.0f  weights/ds %.0f  dpre2 %.0f  dW2/dz1 %.0f  dAP/dX %.0f",
               z[0] / n, z[7] / n, z[1] / n, z[2] / n, z[3] / n, z[4] / n, z[5] / n);
#endif
        printf("\n");
    }
    return 0;
}
