// Feasibility probe: does hipExtAnyOrderLaunch let kernel N+1 of a stream start (and prefetch) while kernel N is still
// running, with the dependency carried by a device counter?  Prints start/end times of a chain of kernels.
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <cstdio>
#include <vector>

__global__ void work_kernel(unsigned *done_prev, unsigned need_prev, unsigned *done_me, unsigned long long *t, int spin_us) {
    const unsigned long long t_start = wall_clock64();
    if (threadIdx.x == 0) {
        if (done_prev) {
            while (__hip_atomic_load(done_prev, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < need_prev) {
                __builtin_amdgcn_s_sleep(2);
                if (wall_clock64() - t_start > 100000000ull) break;   // 1 s: never hang
            }
        }
    }
    __syncthreads();
    const unsigned long long t_go = wall_clock64();
    while (wall_clock64() - t_go < (unsigned long long)spin_us * 100) {}
    __syncthreads();
    if (threadIdx.x == 0) {
        __hip_atomic_fetch_add(done_me, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (blockIdx.x == 0) { t[0] = t_start; t[1] = t_go; t[2] = wall_clock64(); }
    }
}

int main() {
    const int N = 20, grid = 512, spin = 20;
    unsigned *cnt; unsigned long long *t;
    hipMalloc(&cnt, N * sizeof(unsigned)); hipMalloc(&t, N * 3 * sizeof(unsigned long long));
    hipStream_t s; hipStreamCreate(&s);
    for (int mode = 0; mode < 2; ++mode) {
        hipMemsetAsync(cnt, 0, N * sizeof(unsigned), s);
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        hipEventRecord(e0, s);
        for (int k = 0; k < N; ++k) {
            unsigned *prev = k ? cnt + k - 1 : nullptr;
            hipExtLaunchKernelGGL(work_kernel, dim3(grid), dim3(256), 0, s, nullptr, nullptr, (mode && k) ? hipExtAnyOrderLaunch : 0,
                                  prev, (unsigned)grid, cnt + k, t + 3 * k, spin);
        }
        hipEventRecord(e1, s);
        hipStreamSynchronize(s);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        std::vector<unsigned long long> h(3 * N);
        hipMemcpy(h.data(), t, h.size() * 8, hipMemcpyDeviceToHost);
        printf("mode %d (%s): chain of %d x %d us kernels: %.1f us total, %.2f us per kernel over the spin\n", mode,
               mode ? "any-order" : "in-order", N, spin, ms * 1e3, (ms * 1e3 - N * spin) / N);
        for (int k = 1; k < 6; ++k)
            printf("   k=%d start-prev_end %.2f us, go-prev_end %.2f us\n", k, ((double)h[3 * k] - (double)h[3 * (k - 1) + 2]) * 0.01,
                   ((double)h[3 * k + 1] - (double)h[3 * (k - 1) + 2]) * 0.01);
    }
    hipError_t e = hipGetLastError();
    printf("%s\n", hipGetErrorString(e));
    return 0;
}
