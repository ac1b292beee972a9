// What does a dependent kernel launch cost by workgroup size, LDS size and argument-block size?  (empty bodies, 256 workgroups,
// hipGraph replay of a 1000-launch chain and eager launches; build + run on the GPU box:
//   hipcc --offload-arch=gfx950 -O2 -o gpurun_out/launch_cost tools/launch_cost.hip && gpurun_out/launch_cost)
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <vector>
struct Big { long long pad[40]; float *out; };
template <int NT> __global__ __launch_bounds__(NT) void k_small(float *out) {
    extern __shared__ char sm[];
    if (out && threadIdx.x == 0 && blockIdx.x == 0) out[0] = 1.0f;
}
template <int NT> __global__ __launch_bounds__(NT) void k_big(Big b) {
    extern __shared__ char sm[];
    if (b.out && threadIdx.x == 0 && blockIdx.x == 0) b.out[0] = (float)b.pad[39];
}
template <typename F> static double chain(F launch, int n, bool graph, hipStream_t s) {
    hipGraph_t g; hipGraphExec_t ge;
    if (graph) {
        hipStreamBeginCapture(s, hipStreamCaptureModeGlobal);
        for (int i = 0; i < n; ++i) launch();
        hipStreamEndCapture(s, &g);
        hipGraphInstantiate(&ge, g, nullptr, nullptr, 0);
        hipGraphLaunch(ge, s); hipStreamSynchronize(s);
    } else { for (int i = 0; i < 50; ++i) launch(); hipStreamSynchronize(s); }
    auto t0 = std::chrono::steady_clock::now();
    for (int r = 0; r < 5; ++r) { if (graph) hipGraphLaunch(ge, s); else for (int i = 0; i < n; ++i) launch(); }
    hipStreamSynchronize(s);
    double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() / (5.0 * n);
    if (graph) { hipGraphExecDestroy(ge); hipGraphDestroy(g); }
    return us;
}
int main() {
    hipStream_t s; hipStreamCreate(&s);
    float *d; hipMalloc(&d, 4096);
    Big b{}; b.out = d;
    const int n = 1000;
    hipFuncSetAttribute((const void *)k_small<1024>, hipFuncAttributeMaxDynamicSharedMemorySize, 131072);
    hipFuncSetAttribute((const void *)k_small<512>, hipFuncAttributeMaxDynamicSharedMemorySize, 131072);
    hipFuncSetAttribute((const void *)k_small<256>, hipFuncAttributeMaxDynamicSharedMemorySize, 131072);
    hipFuncSetAttribute((const void *)k_big<1024>, hipFuncAttributeMaxDynamicSharedMemorySize, 131072);
    hipFuncSetAttribute((const void *)k_big<512>, hipFuncAttributeMaxDynamicSharedMemorySize, 131072);
    for (int graph = 0; graph < 2; ++graph) {
        printf("== %s, 256 workgroups, us per dependent launch of an empty kernel\n", graph ? "hipGraph replay" : "eager");
        for (int lds : {0, 65536, 131072}) {
            printf("  LDS %6d B:  256 thr %.2f   512 thr %.2f   1024 thr %.2f   | 328-byte args: 512 thr %.2f   1024 thr %.2f\n", lds,
                   chain([&] { hipLaunchKernelGGL(k_small<256>, dim3(256), dim3(256), lds, s, d); }, n, graph, s),
                   chain([&] { hipLaunchKernelGGL(k_small<512>, dim3(256), dim3(512), lds, s, d); }, n, graph, s),
                   chain([&] { hipLaunchKernelGGL(k_small<1024>, dim3(256), dim3(1024), lds, s, d); }, n, graph, s),
                   chain([&] { hipLaunchKernelGGL(k_big<512>, dim3(256), dim3(512), lds, s, b); }, n, graph, s),
                   chain([&] { hipLaunchKernelGGL(k_big<1024>, dim3(256), dim3(1024), lds, s, b); }, n, graph, s));
        }
        printf("  512 workgroups x 256 thr (attention's grid): %.2f\n", chain([&] { hipLaunchKernelGGL(k_small<256>, dim3(512), dim3(256), 0, s, d); }, n, graph, s));
    }
    return 0;
}
