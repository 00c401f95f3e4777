// cin_pool.hip -- the LAST layer of a CIN stack, whose feature map feeds nothing but its pooled sums.
//
// NO REFERENCE CODE (README.md:28 links arXiv:1803.05170); from the definition in include/dir_hip.h (A14):
//   pooled[b,h] = sum_d xout[b,h,d] = sum_d sum_{i,j} W[h,i,j] xk[b,i,d] x0[b,j,d] = sum_{i,j} W[h,i,j] Z[b,i,j],
//   Z[b,i,j] = sum_d xk[b,i,d] * x0[b,j,d]
// The sum over the embedding dimension commutes with the contraction over (i, j): taken first it leaves ONE row of Hp * m products per
// sample instead of D rows -- 1/D of the matrix work (the layer kernel spent 3.3 ms on the last 128-wide layer of the BASELINE stack, this
// form 0.2 ms here + 0.22 ms in the dense kernel).  The backward of that layer given g = dL/dpooled [B, H] has the same shape:
//   dW[h,(i,j)] = sum_b g[b,h] Z[b,i,j]                 (the dense weight-gradient kernel on (g, Z))
//   dZ[b,(i,j)] = sum_h g[b,h] W[h,i,j]                 (the dense kernel on (g, W^T))
//   dxk[b,i,d]  = sum_j dZ[b,i,j] x0[b,j,d]   (+ the pooled gradient of the layer below, broadcast over d: that layer's dL/dxout)
//   dx0[b,j,d] += sum_i dZ[b,i,j] xk[b,i,d]
// This file holds the two per-sample contractions over d / over (i, j): cin_pool_z_k and cin_pool_dx_k.  Both are HBM-bound on the
// [B, Hp * m] matrix (872 MB at the BASELINE shape); a workgroup takes one sample at a time, thread = one xk channel i.
#include "common.hpp"

namespace dir {

constexpr int CP_MAXM = 64;
// NT threads per workgroup: 128 when the layer has at most 128 input channels (one chunk, twice the resident workgroups), else 256

// Z[b, i*m + j] = sum_d xk[b,i,d] x0[b,j,d].  Thread i keeps its channel's D values in registers and walks the fields (x0 rows are LDS
// broadcasts); the Hp x m tile goes through LDS so that the global stores are contiguous 16-byte pieces.
template <int D, int CP_THREADS>
__global__ __launch_bounds__(CP_THREADS) void cin_pool_z_k(const float* __restrict__ x0, const float* __restrict__ xk, int m, int Hp, int64_t B,
                                                            float* __restrict__ Z,
                                                            unsigned int* __restrict__ zbits = nullptr /* optional [B]: bit pattern of max |Z[b, :]| */) {
    extern __shared__ __attribute__((aligned(16))) float cp_smem[];
    float* x0s = cp_smem;                       // [m][D]
    float* zt = cp_smem + ((m * D + 3) & ~3);   // [chunk of 256 channels][m + 1]
    float* zmax_s = zt + CP_THREADS * (m + 1);  // [waves]: behind the tile (all LDS of this kernel is dynamic: the 160 KiB attribute is set for it)
    const int tid = threadIdx.x;
    const int zs = m + 1;
    for (int64_t b = blockIdx.x; b < B; b += gridDim.x) {
        float zmx = 0.f;                        // this thread's largest |Z| of the sample (the row scale of the dense product that follows)
        __syncthreads();                        // the previous sample's tile has been written out
        for (int e = tid; e < m * D; e += CP_THREADS) x0s[e] = x0[b * m * D + e];
        __syncthreads();
        for (int i0 = 0; i0 < Hp; i0 += CP_THREADS) {
            const int i = i0 + tid;
            const int nch = min(CP_THREADS, Hp - i0);
            if (i < Hp) {
                float xv[D];
#pragma unroll
                for (int d = 0; d < D; d += 4) {
                    const float4 v = *reinterpret_cast<const float4*>(xk + (b * Hp + i) * D + d);
                    xv[d] = v.x; xv[d + 1] = v.y; xv[d + 2] = v.z; xv[d + 3] = v.w;
                }
#pragma unroll 2
                for (int j = 0; j < m; ++j) {
                    float4 s = make_float4(0.f, 0.f, 0.f, 0.f);      // four partial sums (d mod 4): four short dependent chains instead of one of D
#pragma unroll
                    for (int d = 0; d < D; d += 4) {                 // (a wave-wide broadcast read of the field's row)
                        const float4 v = *reinterpret_cast<const float4*>(x0s + j * D + d);
                        s.x = fmaf(xv[d], v.x, s.x); s.y = fmaf(xv[d + 1], v.y, s.y); s.z = fmaf(xv[d + 2], v.z, s.z); s.w = fmaf(xv[d + 3], v.w, s.w);
                    }
                    const float zv = (s.x + s.y) + (s.z + s.w);
                    zt[tid * zs + j] = zv;
                    zmx = fmaxf(zmx, fabsf(zv));
                }
            }
            __syncthreads();
            float* dst = Z + (b * Hp + i0) * m;               // the chunk's nch * m values are contiguous in Z
            for (int e = tid; e < nch * m; e += CP_THREADS) dst[e] = zt[(e / m) * zs + (e % m)];
            if (i0 + CP_THREADS < Hp) __syncthreads();
        }
        if (zbits) {                                           // (uniform) one workgroup owns the sample's whole row: a plain store
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) zmx = fmaxf(zmx, __shfl_xor(zmx, o, 64));
            if ((tid & 63) == 0) zmax_s[tid >> 6] = zmx;
            __syncthreads();
            if (tid == 0) {
                float mx = zmax_s[0];
#pragma unroll
                for (int w = 1; w < CP_THREADS / 64; ++w) mx = fmaxf(mx, zmax_s[w]);
                zbits[b] = __builtin_bit_cast(unsigned int, mx);
            }
        }
    }
}

// dxk[b,i,d] = sum_j dZ[b,i,j] x0[b,j,d] (+ addp[b,i]);   dx0[b,j,d] (+)= sum_i dZ[b,i,j] xk[b,i,d]
template <int D, int CP_THREADS>
__global__ __launch_bounds__(CP_THREADS) void cin_pool_dx_k(const float* __restrict__ x0, const float* __restrict__ xk, const float* __restrict__ dZ,
                                                             int m, int Hp, int64_t B, const float* __restrict__ addp, int64_t addp_ld,
                                                             float* __restrict__ dxk, float* __restrict__ dx0, int accumulate) {
    extern __shared__ __attribute__((aligned(16))) float cp_smem[];
    float* x0s = cp_smem;                                   // [m][D]
    float* zt = cp_smem + ((m * D + 3) & ~3);               // [256 channels][m + 1]: the chunk's dZ rows
    float* xks = zt + ((CP_THREADS * (m + 1) + 3) & ~3);    // [256 channels][D]: the chunk's xk rows (for the reduction over i)
    const int tid = threadIdx.x;
    const int zs = m + 1;
    const int nq = m * (D / 4);                             // (field, d-quad) pairs of dx0: one thread each, NR rounds (nq <= 512)
    constexpr int NR = 512 / CP_THREADS;
    for (int64_t b = blockIdx.x; b < B; b += gridDim.x) {
        __syncthreads();
        for (int e = tid; e < m * D; e += CP_THREADS) x0s[e] = x0[b * m * D + e];
        float4 acc0[NR];
#pragma unroll
        for (int r = 0; r < NR; ++r) acc0[r] = make_float4(0.f, 0.f, 0.f, 0.f);
        for (int i0 = 0; i0 < Hp; i0 += CP_THREADS) {
            const int nch = min(CP_THREADS, Hp - i0);
            __syncthreads();                                // x0s staged / the previous chunk's readers are done
            const float* src = dZ + (b * Hp + i0) * m;
            for (int e = tid; e < nch * m; e += CP_THREADS) zt[(e / m) * zs + (e % m)] = src[e];
            const int i = i0 + tid;
            float xv[D];
            if (i < Hp) {
#pragma unroll
                for (int d = 0; d < D; d += 4) {
                    const float4 v = *reinterpret_cast<const float4*>(xk + (b * Hp + i) * D + d);
                    xv[d] = v.x; xv[d + 1] = v.y; xv[d + 2] = v.z; xv[d + 3] = v.w;
                    *reinterpret_cast<float4*>(xks + tid * D + d) = v;
                }
            }
            __syncthreads();
            if (i < Hp) {                                   // dxk row i: walk the fields, x0 rows are LDS broadcasts
                float o[D];
                const float a = addp ? addp[b * addp_ld + i] : 0.f;
#pragma unroll
                for (int d = 0; d < D; ++d) o[d] = a;
#pragma unroll 2
                for (int j = 0; j < m; ++j) {
                    const float z = zt[tid * zs + j];
#pragma unroll
                    for (int d = 0; d < D; d += 4) {
                        const float4 v = *reinterpret_cast<const float4*>(x0s + j * D + d);
                        o[d] = fmaf(z, v.x, o[d]); o[d + 1] = fmaf(z, v.y, o[d + 1]); o[d + 2] = fmaf(z, v.z, o[d + 2]); o[d + 3] = fmaf(z, v.w, o[d + 3]);
                    }
                }
#pragma unroll
                for (int d = 0; d < D; d += 4)
                    *reinterpret_cast<float4*>(dxk + (b * Hp + i) * D + d) = make_float4(o[d], o[d + 1], o[d + 2], o[d + 3]);
            }
            // dx0: thread (j, d-quad) adds this chunk's channels in channel order
#pragma unroll
            for (int r = 0; r < NR; ++r) {
                const int q = tid + r * CP_THREADS;
                if (q < nq) {
                    const int j = q / (D / 4), dq = q - j * (D / 4);
                    float4 s = acc0[r], s2 = make_float4(0.f, 0.f, 0.f, 0.f);
                    int c = 0;
                    for (; c + 8 <= nch; c += 8) {                  // eight channels' LDS reads in flight; even / odd channels on two accumulator sets
                        float z[8];
                        float4 v[8];
#pragma unroll
                        for (int u = 0; u < 8; ++u) {
                            z[u] = zt[(c + u) * zs + j];
                            v[u] = *reinterpret_cast<const float4*>(xks + (c + u) * D + 4 * dq);
                        }
#pragma unroll
                        for (int u = 0; u < 8; u += 2) {
                            s.x = fmaf(z[u], v[u].x, s.x); s.y = fmaf(z[u], v[u].y, s.y); s.z = fmaf(z[u], v[u].z, s.z); s.w = fmaf(z[u], v[u].w, s.w);
                            s2.x = fmaf(z[u + 1], v[u + 1].x, s2.x); s2.y = fmaf(z[u + 1], v[u + 1].y, s2.y);
                            s2.z = fmaf(z[u + 1], v[u + 1].z, s2.z); s2.w = fmaf(z[u + 1], v[u + 1].w, s2.w);
                        }
                    }
                    for (; c < nch; ++c) {
                        const float z = zt[c * zs + j];
                        const float4 v = *reinterpret_cast<const float4*>(xks + c * D + 4 * dq);
                        s.x = fmaf(z, v.x, s.x); s.y = fmaf(z, v.y, s.y); s.z = fmaf(z, v.z, s.z); s.w = fmaf(z, v.w, s.w);
                    }
                    s.x += s2.x; s.y += s2.y; s.z += s2.z; s.w += s2.w;
                    acc0[r] = s;
                }
            }
        }
#pragma unroll
        for (int r = 0; r < NR; ++r) {
            const int q = tid + r * CP_THREADS;
            if (q < nq) {
                float4* p = reinterpret_cast<float4*>(dx0 + b * m * D) + q;       // q = j * (D / 4) + dq: the sample's [m, D] block in order
                float4 s = acc0[r];
                if (accumulate) {
                    const float4 o = *p;
                    s.x += o.x; s.y += o.y; s.z += o.z; s.w += o.w;
                }
                *p = s;
            }
        }
    }
}

// The same for a layer of at most NT channels and at most 32 fields (one chunk per sample), software-pipelined: the NEXT sample's dZ rows,
// xk rows and x0 slice are loaded into registers before the current sample is computed, so that a workgroup's turn no longer begins
// with two dependent round trips to memory (cin_pool_dx_k moved 2.2 GB at 2.8 TB/s).
template <int D, int NT>
__global__ __launch_bounds__(NT) void cin_pool_dx1_k(const float* __restrict__ x0, const float* __restrict__ xk, const float* __restrict__ dZ, int m,
                                                      int Hp, int64_t B, const float* __restrict__ addp, int64_t addp_ld, float* __restrict__ dxk,
                                                      float* __restrict__ dx0, int accumulate) {
    extern __shared__ __attribute__((aligned(16))) float cp_smem[];
    float* x0s = cp_smem;                                   // [m][D]
    float* zt = cp_smem + ((m * D + 3) & ~3);               // [NT channels][m + 1]
    float* xks = zt + ((NT * (m + 1) + 3) & ~3);            // [NT channels][D]
    constexpr int ME = 32, M0 = 32 * 32 / NT;               // dZ elements / x0 elements per thread (m <= 32, D <= 32)
    const int tid = threadIdx.x;
    const int zs = m + 1, nz = Hp * m, n0 = m * D;
    const int nq = n0 / 4;                                  // (field, d-quad) pairs of dx0
    constexpr int NR = 512 / NT;
    int off[ME];                                            // where element tid + k * NT of a sample's [Hp, m] block goes in zt
#pragma unroll
    for (int k = 0; k < ME; ++k) {
        const int e = tid + k * NT;
        off[k] = (e / m) * zs + (e % m);
    }
    float pz[ME], p0[M0], pa = 0.f;
    float4 px[D / 4];
    auto fetch = [&](int64_t bb) {
        const float* src = dZ + bb * nz;
#pragma unroll
        for (int k = 0; k < ME; ++k) {
            const int e = tid + k * NT;
            pz[k] = e < nz ? src[e] : 0.f;
        }
#pragma unroll
        for (int k = 0; k < M0; ++k) {
            const int e = tid + k * NT;
            p0[k] = e < n0 ? x0[bb * n0 + e] : 0.f;
        }
        if (tid < Hp) {
#pragma unroll
            for (int d = 0; d < D; d += 4) px[d / 4] = *reinterpret_cast<const float4*>(xk + (bb * Hp + tid) * D + d);
            pa = addp ? addp[bb * addp_ld + tid] : 0.f;
        }
    };
    int64_t b = blockIdx.x;
    if (b < B) fetch(b);
    for (; b < B; b += gridDim.x) {
        __syncthreads();                                    // the previous sample's LDS readers are done
#pragma unroll
        for (int k = 0; k < ME; ++k)
            if (tid + k * NT < nz) zt[off[k]] = pz[k];
#pragma unroll
        for (int k = 0; k < M0; ++k)
            if (tid + k * NT < n0) x0s[tid + k * NT] = p0[k];
        const float a = pa;
        if (tid < Hp) {
#pragma unroll
            for (int d = 0; d < D; d += 4) *reinterpret_cast<float4*>(xks + tid * D + d) = px[d / 4];
        }
        if (b + gridDim.x < B) fetch(b + gridDim.x);        // in flight under this sample's arithmetic
        __syncthreads();
        if (tid < Hp) {                                     // dxk row: walk the fields, x0 rows are LDS broadcasts
            float o[D];
#pragma unroll
            for (int d = 0; d < D; ++d) o[d] = a;
#pragma unroll 2
            for (int j = 0; j < m; ++j) {
                const float z = zt[tid * zs + j];
#pragma unroll
                for (int d = 0; d < D; d += 4) {
                    const float4 v = *reinterpret_cast<const float4*>(x0s + j * D + d);
                    o[d] = fmaf(z, v.x, o[d]); o[d + 1] = fmaf(z, v.y, o[d + 1]); o[d + 2] = fmaf(z, v.z, o[d + 2]); o[d + 3] = fmaf(z, v.w, o[d + 3]);
                }
            }
#pragma unroll
            for (int d = 0; d < D; d += 4)
                *reinterpret_cast<float4*>(dxk + (b * Hp + tid) * D + d) = make_float4(o[d], o[d + 1], o[d + 2], o[d + 3]);
        }
#pragma unroll
        for (int r = 0; r < NR; ++r) {                      // dx0: thread (j, d-quad) adds the channels in order, eight in flight
            const int q = tid + r * NT;
            if (q < nq) {
                const int j = q / (D / 4), dq = q - j * (D / 4);
                float4 s = make_float4(0.f, 0.f, 0.f, 0.f), s2 = s;
                int c = 0;
                for (; c + 8 <= Hp; c += 8) {
                    float z[8];
                    float4 v[8];
#pragma unroll
                    for (int u = 0; u < 8; ++u) {
                        z[u] = zt[(c + u) * zs + j];
                        v[u] = *reinterpret_cast<const float4*>(xks + (c + u) * D + 4 * dq);
                    }
#pragma unroll
                    for (int u = 0; u < 8; u += 2) {
                        s.x = fmaf(z[u], v[u].x, s.x); s.y = fmaf(z[u], v[u].y, s.y); s.z = fmaf(z[u], v[u].z, s.z); s.w = fmaf(z[u], v[u].w, s.w);
                        s2.x = fmaf(z[u + 1], v[u + 1].x, s2.x); s2.y = fmaf(z[u + 1], v[u + 1].y, s2.y);
                        s2.z = fmaf(z[u + 1], v[u + 1].z, s2.z); s2.w = fmaf(z[u + 1], v[u + 1].w, s2.w);
                    }
                }
                for (; c < Hp; ++c) {
                    const float z = zt[c * zs + j];
                    const float4 v = *reinterpret_cast<const float4*>(xks + c * D + 4 * dq);
                    s.x = fmaf(z, v.x, s.x); s.y = fmaf(z, v.y, s.y); s.z = fmaf(z, v.z, s.z); s.w = fmaf(z, v.w, s.w);
                }
                s.x += s2.x; s.y += s2.y; s.z += s2.z; s.w += s2.w;
                float4* p = reinterpret_cast<float4*>(dx0 + b * n0) + q;
                if (accumulate) {
                    const float4 o = *p;
                    s.x += o.x; s.y += o.y; s.z += o.z; s.w += o.w;
                }
                *p = s;
            }
        }
    }
}

static int cp_check(const char* name, int m, int Hp, int D, int64_t B) {
    DIR_CHECK_ARG(m > 0 && Hp > 0 && D > 0 && B >= 0, "%s: m=%d Hp=%d D=%d", name, m, Hp, D);
    if (!(D == 4 || D == 8 || D == 16 || D == 32) || m > CP_MAXM)
        return fail(DIR_E_UNSUPPORTED, "%s: m=%d D=%d (supported: m <= %d, D in 4, 8, 16, 32)", name, m, D, CP_MAXM);
    return DIR_OK;
}

}  // namespace dir

using namespace dir;

static int cp_z_run(const char* name, const float* x0, const float* xk, int m, int Hp, int D, int64_t B, float* Z, unsigned int* zbits, dir_stream_t stream);
extern "C" int dir_cin_pool_z_f32(const float* x0, const float* xk, int m, int Hp, int D, int64_t B, float* Z, dir_stream_t stream) {
    return cp_z_run("dir_cin_pool_z_f32", x0, xk, m, Hp, D, B, Z, nullptr, stream);
}
// ... and the bit pattern of max |Z[b, :]| per sample (what dir_dense_f16x2_rows_f32 scales the rows of Z by: no max pass over the
// [B, Hp * m] matrix -- 872 MB at the BASELINE shape)
extern "C" int dir_cin_pool_z_bits_f32(const float* x0, const float* xk, int m, int Hp, int D, int64_t B, float* Z, unsigned int* z_row_bits,
                                       dir_stream_t stream) {
    DIR_CHECK_ARG(z_row_bits || B == 0, "dir_cin_pool_z_bits_f32: z_row_bits is null");
    return cp_z_run("dir_cin_pool_z_bits_f32", x0, xk, m, Hp, D, B, Z, z_row_bits, stream);
}
static int cp_z_run(const char* name, const float* x0, const float* xk, int m, int Hp, int D, int64_t B, float* Z, unsigned int* zbits, dir_stream_t stream) {
    if (int rc = cp_check(name, m, Hp, D, B)) return rc;
    if (B == 0) return DIR_OK;
    DIR_CHECK_ARG(x0 && xk && Z, "%s: null pointer", name);
    if (!(aligned16(x0) && aligned16(xk))) return fail(DIR_E_BADARG, "%s: x0 / xk must be 16-byte aligned", name);
    const int nt = Hp <= 128 ? 128 : 256;
    const size_t sh = sizeof(float) * (size_t)(((m * D + 3) & ~3) + nt * (m + 1) + 4);       // + the waves' row maxima
    const dim3 grid((unsigned)(B < 32 * kCUs ? B : 32 * kCUs));        // one sample per workgroup and turn: many small workgroups hide each other's loads
    hipStream_t st = as_stream(stream);
#define DIR_CP_Z(DD)                                                                                              \
    do {                                                                                                          \
        static LdsOnce once;                                                                                                      \
        (void)lds_limit(once, 160 * 1024, &cin_pool_z_k<DD, 128>, &cin_pool_z_k<DD, 256>);                                        \
        if (nt == 128) hipLaunchKernelGGL((cin_pool_z_k<DD, 128>), grid, dim3(128), sh, st, x0, xk, m, Hp, B, Z, zbits); \
        else hipLaunchKernelGGL((cin_pool_z_k<DD, 256>), grid, dim3(256), sh, st, x0, xk, m, Hp, B, Z, zbits);           \
    } while (0)
    switch (D) {
        case 4: DIR_CP_Z(4); break;
        case 8: DIR_CP_Z(8); break;
        case 16: DIR_CP_Z(16); break;
        default: DIR_CP_Z(32); break;
    }
#undef DIR_CP_Z
    DIR_CHECK_LAUNCH(name);
    return DIR_OK;
}

extern "C" int dir_cin_pool_dx_f32(const float* x0, const float* xk, const float* dZ, int m, int Hp, int D, int64_t B, const float* add_pooled,
                                   int64_t add_pooled_ld, float* dxk, float* dx0, int accumulate_dx0, dir_stream_t stream) {
    const char* name = "dir_cin_pool_dx_f32";
    if (int rc = cp_check(name, m, Hp, D, B)) return rc;
    if (B == 0) return DIR_OK;
    DIR_CHECK_ARG(x0 && xk && dZ && dxk && dx0, "%s: null pointer", name);
    DIR_CHECK_ARG(!add_pooled || add_pooled_ld >= Hp, "%s: add_pooled_ld=%lld < Hp=%d", name, (long long)add_pooled_ld, Hp);
    if (!(aligned16(x0) && aligned16(xk) && aligned16(dxk) && aligned16(dx0))) return fail(DIR_E_BADARG, "%s: x0 / xk / dxk / dx0 must be 16-byte aligned", name);
    const int nt = Hp <= 128 ? 128 : 256;
    const size_t sh = sizeof(float) * (size_t)(((m * D + 3) & ~3) + ((nt * (m + 1) + 3) & ~3) + nt * D);
    const dim3 grid((unsigned)(B < 32 * kCUs ? B : 32 * kCUs));
    hipStream_t st = as_stream(stream);
#define DIR_CP_DX(DD)                                                                                             \
    do {                                                                                                          \
        static LdsOnce once;                                                                                                      \
        (void)lds_limit(once, 160 * 1024, &cin_pool_dx_k<DD, 128>, &cin_pool_dx1_k<DD, 128>, &cin_pool_dx_k<DD, 256>);            \
        if (Hp <= nt && m <= 32 && nt == 128)            /* one chunk per sample: the software-pipelined kernel */ \
            hipLaunchKernelGGL((cin_pool_dx1_k<DD, 128>), grid, dim3(128), sh, st, x0, xk, dZ, m, Hp, B, add_pooled, add_pooled_ld, dxk, dx0, \
                               accumulate_dx0);                                                                   \
        else if (nt == 128)                                                                                       \
            hipLaunchKernelGGL((cin_pool_dx_k<DD, 128>), grid, dim3(128), sh, st, x0, xk, dZ, m, Hp, B, add_pooled, add_pooled_ld, dxk, dx0, \
                               accumulate_dx0);                                                                   \
        else                                                                                                      \
            hipLaunchKernelGGL((cin_pool_dx_k<DD, 256>), grid, dim3(256), sh, st, x0, xk, dZ, m, Hp, B, add_pooled, add_pooled_ld, dxk, dx0, \
                               accumulate_dx0);                                                                   \
    } while (0)
    switch (D) {
        case 4: DIR_CP_DX(4); break;
        case 8: DIR_CP_DX(8); break;
        case 16: DIR_CP_DX(16); break;
        default: DIR_CP_DX(32); break;
    }
#undef DIR_CP_DX
    DIR_CHECK_LAUNCH(name);
    return DIR_OK;
}
