// Micro-benchmark for the weight-streaming (skinny, M<=16) GEMM: which load pattern reaches the HBM roof?
//   hipcc --offload-arch=gfx950 -O3 tools/bench_skinny.hip -o gpurun_out/bench_skinny && ./gpurun_out/bench_skinny
// Variants stream NBUF distinct weight matrices round-robin (>> 256 MiB Infinity Cache) and report GB/s.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef _Float16 half_t;
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f4 __attribute__((ext_vector_type(4)));
typedef unsigned int u4 __attribute__((ext_vector_type(4)));

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

// LAYOUT 0: row-major W[N][K], lane (li, g) reads 32 B at row li, k = 64c + 16g  (current kernel)
// LAYOUT 1: tiled: block (panel p = n/16, chunk c) is 2 KB contiguous: [s=0..1][lane][16 B]
template <int LAYOUT, int NT, int U, bool NTLOAD>
__global__ __launch_bounds__(1024) void skinny(const half_t *__restrict__ A, const half_t *__restrict__ W, float *__restrict__ C,
                                               int N, int K) {
    extern __shared__ float red[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nwaves = blockDim.x >> 6;
    const int n0 = blockIdx.x * 16 * NT;
    const int chunks = K >> 6;
    const int c0 = (int)((long)chunks * wave / nwaves), c1 = (int)((long)chunks * (wave + 1) / nwaves);
    const int g = lane >> 4, li = lane & 15;
    const half_t *wp[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        if (LAYOUT == 0) wp[t] = W + (long)(n0 + 16 * t + li) * K + g * 16;
        else wp[t] = W + ((long)(n0 / 16 + t) * chunks) * 1024 + lane * 8;
    }
    const half_t *ap = A + g * 16;
    f4 acc[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) acc[t] = f4{0, 0, 0, 0};
    int c = c0;
    for (; c + U <= c1; c += U) {
        h8 wl[U][NT], wh[U][NT], al[U], ah[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
#pragma unroll
            for (int t = 0; t < NT; ++t) {
                const h8 *p0, *p1;
                if (LAYOUT == 0) { p0 = (const h8 *)(wp[t] + (long)(c + u) * 64); p1 = p0 + 1; }
                else { p0 = (const h8 *)(wp[t] + (long)(c + u) * 1024); p1 = p0 + 64; }
                if (NTLOAD) { wl[u][t] = __builtin_nontemporal_load(p0); wh[u][t] = __builtin_nontemporal_load(p1); }
                else { wl[u][t] = *p0; wh[u][t] = *p1; }
            }
            const h8 *pa = (const h8 *)(ap + (long)(c + u) * 64);
            al[u] = pa[0]; ah[u] = pa[1];
        }
#pragma unroll
        for (int u = 0; u < U; ++u)
#pragma unroll
            for (int t = 0; t < NT; ++t) {
                acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(al[u], wl[u][t], acc[t], 0, 0, 0);
                acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[u], wh[u][t], acc[t], 0, 0, 0);
            }
    }
    for (; c < c1; ++c) {
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            const h8 *p0, *p1;
            if (LAYOUT == 0) { p0 = (const h8 *)(wp[t] + (long)c * 64); p1 = p0 + 1; }
            else { p0 = (const h8 *)(wp[t] + (long)c * 1024); p1 = p0 + 64; }
            const h8 *pa = (const h8 *)(ap + (long)c * 64);
            acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(pa[0], *p0, acc[t], 0, 0, 0);
            acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(pa[1], *p1, acc[t], 0, 0, 0);
        }
    }
    if (nwaves > 1) {
#pragma unroll
        for (int t = 0; t < NT; ++t)
#pragma unroll
            for (int r = 0; r < 4; ++r) red[((wave * NT + t) * 4 + r) * 64 + lane] = acc[t][r];
        __syncthreads();
        if (wave != 0) return;
#pragma unroll
        for (int t = 0; t < NT; ++t)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                float s = acc[t][r];
                for (int w = 1; w < nwaves; ++w) s += red[((w * NT + t) * 4 + r) * 64 + lane];
                acc[t][r] = s;
            }
    }
    if (g == 0)
#pragma unroll
        for (int t = 0; t < NT; ++t) C[n0 + 16 * t + li] = acc[t][0];
}

// pure streaming read of the same bytes (upper bound for this chip / this buffer set)
__global__ __launch_bounds__(256) void stream_read(const u4 *__restrict__ W, float *__restrict__ C, long n16) {
    u4 acc = {0, 0, 0, 0};
    long i = (long)blockIdx.x * 256 + threadIdx.x;
    const long stride = (long)gridDim.x * 256;
    for (; i + 3 * stride < n16; i += 4 * stride) {
        u4 a = __builtin_nontemporal_load(W + i), b = __builtin_nontemporal_load(W + i + stride);
        u4 c = __builtin_nontemporal_load(W + i + 2 * stride), d = __builtin_nontemporal_load(W + i + 3 * stride);
        acc ^= a ^ b ^ c ^ d;
    }
    for (; i < n16; i += stride) acc ^= W[i];
    if ((acc.x ^ acc.y ^ acc.z ^ acc.w) == 0x12345678) C[0] = 1.f;
}

struct Case { const char *name; int N, K; };

template <int LAYOUT, int NT, int U, bool NTLOAD>
double run(const Case &cs, int W, const std::vector<half_t *> &bufs, half_t *A, float *C, int iters) {
    const int groups = cs.N / (16 * NT);
    const size_t lds = W > 1 ? (size_t)W * NT * 4 * 64 * 4 : 0;
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int i = 0; i < 3; ++i)
        hipLaunchKernelGGL((skinny<LAYOUT, NT, U, NTLOAD>), dim3(groups), dim3(64 * W), lds, 0, A, bufs[i % bufs.size()], C, cs.N, cs.K);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    for (int i = 0; i < iters; ++i)
        hipLaunchKernelGGL((skinny<LAYOUT, NT, U, NTLOAD>), dim3(groups), dim3(64 * W), lds, 0, A, bufs[(i + 3) % bufs.size()], C, cs.N, cs.K);
    CK(hipEventRecord(e1));
    CK(hipDeviceSynchronize());
    CK(hipGetLastError());
    float ms;
    CK(hipEventElapsedTime(&ms, e0, e1));
    return 2.0 * cs.N * cs.K * iters / (ms * 1e-3) / 1e9;
}

int main() {
    const Case cases[] = {{"wgu 28672x4096", 28672, 4096}, {"wqkv 6144x4096", 6144, 4096}, {"wo 4096x4096", 4096, 4096},
                          {"wd 4096x14336", 4096, 14336}, {"lm_head 128256x4096", 128256, 4096}, {"sw2 32768x32768", 32768, 32768}};
    half_t *A;
    float *C;
    CK(hipMalloc(&A, 32768 * 2 * 16));
    CK(hipMemset(A, 0, 32768 * 2 * 16));
    CK(hipMalloc(&C, 131072 * 4));
    for (const Case &cs : cases) {
        const size_t bytes = (size_t)cs.N * cs.K * 2;
        int nbuf = (int)((size_t)6e9 / bytes);
        nbuf = nbuf < 2 ? 2 : (nbuf > 48 ? 48 : nbuf);
        std::vector<half_t *> bufs(nbuf);
        for (auto &b : bufs) { CK(hipMalloc(&b, bytes)); CK(hipMemset(b, 0x3c, bytes)); }
        const int iters = nbuf * 3;
        printf("== %s (%.0f MB x %d buffers)\n", cs.name, bytes / 1e6, nbuf);
        {   // streaming-read ceiling on the same buffers
            hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
            for (int i = 0; i < 2; ++i) hipLaunchKernelGGL(stream_read, dim3(2048), dim3(256), 0, 0, (const u4 *)bufs[i % nbuf], C, (long)(bytes / 16));
            CK(hipEventRecord(e0));
            for (int i = 0; i < iters; ++i) hipLaunchKernelGGL(stream_read, dim3(2048), dim3(256), 0, 0, (const u4 *)bufs[(i + 2) % nbuf], C, (long)(bytes / 16));
            CK(hipEventRecord(e1)); CK(hipDeviceSynchronize());
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            printf("   stream_read ceiling                    %7.0f GB/s\n", bytes * (double)iters / (ms * 1e-3) / 1e9);
        }
        const int groups1 = cs.N / 16, chunks = cs.K / 64;
        for (int W : {1, 2, 4, 8, 16}) {
            if (W > chunks) continue;
            printf("   W=%2d waves/blk (%5d blocks NT1):", W, groups1);
            printf(" row/nt/U4 %5.0f", run<0, 1, 4, true>(cs, W, bufs, A, C, iters));
            printf(" | row/plain/U4 %5.0f", run<0, 1, 4, false>(cs, W, bufs, A, C, iters));
            printf(" | tiled/nt/U4 %5.0f", run<1, 1, 4, true>(cs, W, bufs, A, C, iters));
            printf(" | tiled/plain/U4 %5.0f", run<1, 1, 4, false>(cs, W, bufs, A, C, iters));
            printf(" | tiled/nt/U8 %5.0f", run<1, 1, 8, true>(cs, W, bufs, A, C, iters));
            printf(" | tiled/nt/NT2/U4 %5.0f", run<1, 2, 4, true>(cs, W, bufs, A, C, iters));
            printf(" | row/nt/NT2/U4 %5.0f\n", run<0, 2, 4, true>(cs, W, bufs, A, C, iters));
            fflush(stdout);
        }
        for (auto &b : bufs) CK(hipFree(b));
    }
    return 0;
}
