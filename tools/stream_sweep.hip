// How fast can MI355X stream a once-read buffer of the batched decode step's sizes (33.5 .. 235 MB), by launch geometry?
// Every wave reads a contiguous share with UU non-temporal 1-KB wave loads in flight; distinct buffers per launch (32 launches
// in a hipGraph: nothing is re-read from a cache); the time includes the launch seams, as in the decode graph.
//   hipcc --offload-arch=gfx950 -O3 -o gpurun_out/stream_sweep tools/stream_sweep.hip && gpurun_out/stream_sweep
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
typedef unsigned int u4 __attribute__((ext_vector_type(4)));
template <int UU, int NT>
__global__ __launch_bounds__(NT) void sweep_kernel(const u4 *w, long long n16, unsigned *out, int skew) {
    extern __shared__ char sm[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    constexpr int NW = NT / 64;
    const long long kb = n16 / 64, per = kb / ((long long)gridDim.x * NW);
    const long long lo = ((long long)blockIdx.x * NW + wave) * per, hi = lo + per;
    // skew: start each wave's walk at a different place of its share (memory-channel camping: shares are a power-of-two-ish
    // multiple of 1 KB apart, and every wave walks at the same pace)
    const long long rot = skew ? (((long long)blockIdx.x * 7 + wave * 3) % (per / UU)) * UU : 0;
    u4 acc = {0, 0, 0, 0};
    u4 r[UU];
    auto at = [&](long long i) { long long j = i + rot; j = j >= hi ? j - per : j; return w + j * 64 + lane; };
#pragma unroll
    for (int u = 0; u < UU; ++u) r[u] = __builtin_nontemporal_load(at(lo + u));
    for (long long i = lo + UU; i + UU <= hi; i += UU) {
#pragma unroll
        for (int u = 0; u < UU; ++u) {
            acc ^= r[u];
            r[u] = __builtin_nontemporal_load(at(i + u));
        }
    }
#pragma unroll
    for (int u = 0; u < UU; ++u) acc ^= r[u];
    if ((acc[0] ^ acc[1] ^ acc[2] ^ acc[3]) == 0x12345678u) out[blockIdx.x] = 1;
}
int main() {
    hipStream_t s; (void)hipStreamCreate(&s);
    const long long sizes[4] = {33554432, 50331648, 117440512, 234881024};
    const int L = 32;
    u4 *buf[4][L];
    for (int k = 0; k < 4; ++k)
        for (int l = 0; l < L; ++l) { (void)hipMalloc((void **)&buf[k][l], sizes[k]); (void)hipMemset(buf[k][l], 1, sizes[k]); }
    unsigned *out; (void)hipMalloc((void **)&out, 65536);
    (void)hipDeviceSynchronize();
    auto run = [&](auto kern, int nwg, int nt, int uu, int lds, int skew) {
        (void)hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
        printf("  %4d wg x %4d thr, %2d in flight, lds %3d KB, skew %d:", nwg, nt, uu, lds >> 10, skew);
        for (int k = 0; k < 4; ++k) {
            // every wave needs at least two full rounds of its queue inside its share (else the kernel would read past it)
            if (sizes[k] / 1024 / ((long long)nwg * (nt / 64)) < 2 * uu) { printf("       -           "); continue; }
            hipGraph_t g; hipGraphExec_t ge;
            (void)hipStreamBeginCapture(s, hipStreamCaptureModeGlobal);
            for (int l = 0; l < L; ++l) hipLaunchKernelGGL(kern, dim3(nwg), dim3(nt), lds, s, buf[k][l], sizes[k] / 16, out, skew);
            (void)hipStreamEndCapture(s, &g);
            (void)hipGraphInstantiate(&ge, g, nullptr, nullptr, 0);
            (void)hipGraphLaunch(ge, s); (void)hipStreamSynchronize(s);
            double best = 1e30;
            for (int r = 0; r < 5; ++r) {
                auto t0 = std::chrono::steady_clock::now();
                (void)hipGraphLaunch(ge, s); (void)hipStreamSynchronize(s);
                double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() / L;
                if (us < best) best = us;
            }
            printf("  %6.2f us %4.2f TB/s", best, sizes[k] / best / 1e6);
            (void)hipGraphExecDestroy(ge); (void)hipGraphDestroy(g);
        }
        printf("\n");
    };
    printf("columns: 33.5 MB (wo)   50.3 MB (QKV)   117 MB (down)   235 MB (gate/up)\n");
    for (int skew = 0; skew < 2; ++skew) {
        for (int lds : {100 << 10, 0}) {
            for (int nwg : {224, 256, 512, 1024}) {
                run(sweep_kernel<8, 512>, nwg, 512, 8, lds, skew);
                run(sweep_kernel<16, 512>, nwg, 512, 16, lds, skew);
            }
            run(sweep_kernel<8, 1024>, 256, 1024, 8, lds, skew);
            run(sweep_kernel<16, 1024>, 256, 1024, 16, lds, skew);
            run(sweep_kernel<8, 256>, 1024, 256, 8, lds, skew);
            run(sweep_kernel<16, 256>, 2048, 256, 16, lds, skew);
        }
    }
    return 0;
}
