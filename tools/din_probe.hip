// din_probe.hip -- cycles per phase of din_mfma_k (development tool, not product).
// build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -Iinclude tools/din_probe.hip \
//        details-in-recommendation_amd/csrc/capi.cpp -o tools/din_probe      (add -DDIN_PLAIN -o tools/din_probe_plain for the unstamped kernel)
#ifndef DIN_PLAIN
#define DIN_STAMP 1
#endif
#include "../details-in-recommendation_amd/csrc/din.hip"
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)

int main() {
    const int K = 64, T = 50, H1 = 80, H2 = 40;
    const int64_t B = 65536, V = 2000000;
    std::vector<float> ht((size_t)V * K), hw1(4 * K * H1), hw2(H1 * H2), hw3(H2);
    unsigned s = 7u;
    auto rnd = [&]() { s = s * 1664525u + 1013904223u; return ((s >> 8) & 0xffff) / 65536.0f - 0.5f; };
    for (auto& v : ht) v = rnd() * 0.25f;
    for (auto& v : hw1) v = rnd() * 0.1f;
    for (auto& v : hw2) v = rnd() * 0.2f;
    for (auto& v : hw3) v = rnd() * 0.2f;
    std::vector<int64_t> hh((size_t)B * T), hc(B);
    std::vector<int32_t> hl(B);
    for (auto& v : hh) { s = s * 1664525u + 1013904223u; v = (s >> 4) % V; }
    for (auto& v : hc) { s = s * 1664525u + 1013904223u; v = (s >> 4) % V; }
    const int fixed_len = getenv("DIN_PROBE_LEN") ? atoi(getenv("DIN_PROBE_LEN")) : 0;   // 0: uniform 1..T
    for (auto& v : hl) { s = s * 1664525u + 1013904223u; v = fixed_len ? fixed_len : 1 + (s >> 8) % T; }
    float *table, *w1, *b1, *w2, *b2, *w3, *b3, *out; int64_t *hist, *cand; int32_t* len;
    CK(hipMalloc(&table, ht.size() * 4)); CK(hipMalloc(&w1, hw1.size() * 4)); CK(hipMalloc(&w2, hw2.size() * 4)); CK(hipMalloc(&w3, hw3.size() * 4));
    CK(hipMalloc(&b1, H1 * 4)); CK(hipMalloc(&b2, H2 * 4)); CK(hipMalloc(&b3, 4)); CK(hipMalloc(&out, B * K * 4));
    CK(hipMalloc(&hist, hh.size() * 8)); CK(hipMalloc(&cand, B * 8)); CK(hipMalloc(&len, B * 4));
    CK(hipMemcpy(table, ht.data(), ht.size() * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(w1, hw1.data(), hw1.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(w2, hw2.data(), hw2.size() * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(w3, hw3.data(), hw3.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemset(b1, 0, H1 * 4)); CK(hipMemset(b2, 0, H2 * 4)); CK(hipMemset(b3, 0, 4));
    CK(hipMemcpy(hist, hh.data(), hh.size() * 8, hipMemcpyHostToDevice)); CK(hipMemcpy(cand, hc.data(), B * 8, hipMemcpyHostToDevice));
    CK(hipMemcpy(len, hl.data(), B * 4, hipMemcpyHostToDevice));
    for (int it = 0; it < 3; ++it) {
        unsigned long long z[8] = {0};
#ifdef DIN_STAMP
        CK(hipMemcpyToSymbol(HIP_SYMBOL(dir::din_stamp), z, sizeof(z)));
#endif
        hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
        CK(hipEventRecord(e0, 0));
        if (dir_din_attention_pool_f32(table, K, hist, len, cand, T, w1, b1, H1, w2, b2, H2, w3, b3, 1, B, out, nullptr, nullptr) != 0) {
            printf("error: %s\n", dir_last_error()); return 1;
        }
        CK(hipEventRecord(e1, 0)); CK(hipDeviceSynchronize());
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
#ifdef DIN_STAMP
        CK(hipMemcpyFromSymbol(z, HIP_SYMBOL(dir::din_stamp), sizeof(z)));
#else
        z[4] = 1;
#endif
        const double n = (double)z[4];
        printf("launch %.3f ms, %llu workgroups | cycles per sample (stamped build: s_memtime serialises, shares matter): stage %.0f  mlp %.0f  softmax %.0f  pool+out %.0f\n",
               ms, z[5], z[0] / n, z[1] / n, z[2] / n, z[3] / n);
    }
    return 0;
}
