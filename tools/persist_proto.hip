// Does a persistent decode layer pay on MI355X at the batched decode step's sizes?  A byte-moving model of the step (no MFMA: the
// step's GEMMs are HBM-bound): per layer five phases that stream what the real kernels stream - QKV 50.3 MB, K / V cache 29 MB,
// wo 33.5 MB, gate / up 235 MB, down 117 MB, distinct buffers per layer (15 GB for 32 layers: nothing is re-read from a cache) -
// 256 workgroups x 512 threads, one per CU, every wave keeping 8 x 1 KB non-temporal loads in flight, as
//   A  one launch per phase, 160 launches captured in a hipGraph               (what api.cpp decode_step replays today)
//   B  ONE launch per step, a grid barrier between phases                        (decode_stack.hip of round 1, removed in round 3)
//   C  B + the first 8 loads per wave of the NEXT phase requested BEFORE the barrier (weights do not depend on the barrier:
//      128 KB per CU = 32 MB chip-wide in flight while the barrier runs)        (cdna_hip_programming.md 5.6, prefetch-credit)
//   each of B / C with and without the agent-scope release / acquire fences a real hand-off of activations needs;
//   D  B / C with the XCD-hierarchical barrier of MI355X_MICROARCH.md (8 group counters + a top counter) instead of one counter.
// Prints microseconds per layer.  Build + run on the GPU box:
//   hipcc --offload-arch=gfx950 -O3 -o gpurun_out/persist_proto tools/persist_proto.hip && gpurun_out/persist_proto
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <vector>

typedef unsigned int u4 __attribute__((ext_vector_type(4)));
constexpr int NPH = 5, NWG = 256, NT = 512, U = 8;
struct Layer { const u4 *w[NPH]; long long n16[NPH]; };        // n16 = 16-byte pieces of the phase
struct Step { Layer l[32]; int nlayers; };

__device__ __forceinline__ u4 ntload(const u4 *p) { return __builtin_nontemporal_load(p); }

// this wave's share of a phase: pieces [lo, hi) in units of 64 x 16 B (1 KB wave accesses), U in flight
__device__ __forceinline__ void share(long long n16, int wg, int wave, long long &lo, long long &hi) {
    const long long kb = n16 / 64, per = kb / (NWG * 8);
    lo = ((long long)wg * 8 + wave) * per;
    hi = lo + per;
}
__device__ __forceinline__ u4 stream_body(const u4 *w, long long lo, long long hi, int lane, u4 (&pre)[U], bool have_pre) {
    u4 acc = {0, 0, 0, 0};
    u4 r[U];
    long long i = lo;
    if (have_pre) {
#pragma unroll
        for (int u = 0; u < U; ++u) r[u] = pre[u];
    } else {
#pragma unroll
        for (int u = 0; u < U; ++u) r[u] = ntload(w + (i + u) * 64 + lane);
    }
    for (i = lo + U; i + U <= hi; i += U) {
#pragma unroll
        for (int u = 0; u < U; ++u) {
            acc ^= r[u];
            r[u] = ntload(w + (i + u) * 64 + lane);
        }
    }
#pragma unroll
    for (int u = 0; u < U; ++u) acc ^= r[u];
    return acc;
}

__global__ __launch_bounds__(NT) void phase_kernel(const u4 *w, long long n16, unsigned *out) {
    extern __shared__ char sm[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    long long lo, hi;
    share(n16, blockIdx.x, wave, lo, hi);
    u4 dummy[U];
    const u4 a = stream_body(w, lo, hi, lane, dummy, false);
    if ((a[0] ^ a[1] ^ a[2] ^ a[3]) == 0x12345678u) out[blockIdx.x] = 1;       // (never true: keeps the loads alive)
}

// the same stream on a grid of gridDim.x workgroups with UU loads in flight per wave (sweep: does the chip's streaming rate need
// every CU - gemm_wide_kernel runs the gate / up projection on 224 workgroups - and how deep a queue per wave?)
template <int UU>
__global__ __launch_bounds__(NT) void sweep_kernel(const u4 *w, long long n16, unsigned *out) {
    extern __shared__ char sm[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const long long kb = n16 / 64, per = kb / ((long long)gridDim.x * 8);
    const long long lo = ((long long)blockIdx.x * 8 + wave) * per, hi = lo + per;
    u4 acc = {0, 0, 0, 0};
    u4 r[UU];
#pragma unroll
    for (int u = 0; u < UU; ++u) r[u] = ntload(w + (lo + u) * 64 + lane);
    for (long long i = lo + UU; i + UU <= hi; i += UU) {
#pragma unroll
        for (int u = 0; u < UU; ++u) {
            acc ^= r[u];
            r[u] = ntload(w + (i + u) * 64 + lane);
        }
    }
#pragma unroll
    for (int u = 0; u < UU; ++u) acc ^= r[u];
    if ((acc[0] ^ acc[1] ^ acc[2] ^ acc[3]) == 0x12345678u) out[blockIdx.x] = 1;
}

template <bool FENCE>
__device__ __forceinline__ bool grid_barrier(unsigned *cnt, unsigned target, unsigned *err) {
    __syncthreads();
    if (threadIdx.x == 0) {
        if (FENCE) __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __hip_atomic_fetch_add(cnt, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        int spins = 0;
        while (__hip_atomic_load(cnt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
            __builtin_amdgcn_s_sleep(2);
            if (++spins > (1 << 22)) { *err = 1; break; }                      // bounded: never hang the box
        }
        if (FENCE) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    }
    __syncthreads();
    return true;
}

// XCD-hierarchical form (MI355X_MICROARCH.md "barrier-xcd"): workgroups that share blockIdx.x % 8 (one XCD under round-robin
// placement: a speed matter only) count on their own line; the last arriver of a group counts on the top word, waits for the
// other groups there and releases its group through the group's generation word - 32 pollers per line instead of 256 on one.
// hb: [8 groups][32 words: cnt at 0, gen at 16] then the top counter at word 256.
__device__ __forceinline__ void grid_barrier_xcd(unsigned *hb, unsigned epoch, unsigned *err) {
    __syncthreads();
    if (threadIdx.x == 0) {
        const int g = blockIdx.x & 7;
        unsigned *gc = hb + g * 32, *gg = hb + g * 32 + 16, *top = hb + 256;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        const unsigned old = __hip_atomic_fetch_add(gc, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        int spins = 0;
        if (old + 1 == (NWG / 8) * epoch) {                                  // last of its group
            __hip_atomic_fetch_add(top, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            while (__hip_atomic_load(top, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < 8 * epoch) {
                __builtin_amdgcn_s_sleep(1);
                if (++spins > (1 << 22)) { *err = 1; break; }
            }
            __hip_atomic_store(gg, epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        } else {
            while (__hip_atomic_load(gg, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < epoch) {
                __builtin_amdgcn_s_sleep(1);
                if (++spins > (1 << 22)) { *err = 1; break; }
            }
        }
    }
    __syncthreads();
}

template <bool PREFETCH>
__global__ __launch_bounds__(NT) void step_kernel_xcd(Step st, unsigned *hb, unsigned base, unsigned *out, unsigned *err) {
    extern __shared__ char sm[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    u4 acc = {0, 0, 0, 0};
    u4 pre[U];
    bool have = false;
    unsigned nb = 0;
    for (int l = 0; l < st.nlayers; ++l) {
#pragma unroll
        for (int ph = 0; ph < NPH; ++ph) {
            long long lo, hi;
            share(st.l[l].n16[ph], blockIdx.x, wave, lo, hi);
            acc ^= stream_body(st.l[l].w[ph], lo, hi, lane, pre, have);
            have = false;
            if (PREFETCH) {
                const int l2 = ph + 1 < NPH ? l : l + 1, p2 = ph + 1 < NPH ? ph + 1 : 0;
                if (l2 < st.nlayers) {
                    long long lo2, hi2;
                    share(st.l[l2].n16[p2], blockIdx.x, wave, lo2, hi2);
#pragma unroll
                    for (int u = 0; u < U; ++u) pre[u] = ntload(st.l[l2].w[p2] + (lo2 + u) * 64 + lane);
                    have = true;
                }
            }
            ++nb;
            grid_barrier_xcd(hb, base + nb, err);
        }
    }
    if ((acc[0] ^ acc[1] ^ acc[2] ^ acc[3]) == 0x12345678u) out[blockIdx.x] = 1;
}

template <bool PREFETCH, bool FENCE>
__global__ __launch_bounds__(NT) void step_kernel(Step st, unsigned *cnt, unsigned base, unsigned *out, unsigned *err) {
    extern __shared__ char sm[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    u4 acc = {0, 0, 0, 0};
    u4 pre[U];
    bool have = false;
    unsigned nb = 0;
    for (int l = 0; l < st.nlayers; ++l) {
#pragma unroll
        for (int ph = 0; ph < NPH; ++ph) {
            long long lo, hi;
            share(st.l[l].n16[ph], blockIdx.x, wave, lo, hi);
            acc ^= stream_body(st.l[l].w[ph], lo, hi, lane, pre, have);
            have = false;
            if (PREFETCH) {                                                     // next phase's first loads, before the barrier
                const int l2 = ph + 1 < NPH ? l : l + 1, p2 = ph + 1 < NPH ? ph + 1 : 0;
                if (l2 < st.nlayers) {
                    long long lo2, hi2;
                    share(st.l[l2].n16[p2], blockIdx.x, wave, lo2, hi2);
#pragma unroll
                    for (int u = 0; u < U; ++u) pre[u] = ntload(st.l[l2].w[p2] + (lo2 + u) * 64 + lane);
                    have = true;
                }
            }
            ++nb;
            grid_barrier<FENCE>(cnt, base + nb * NWG, err);
        }
    }
    if ((acc[0] ^ acc[1] ^ acc[2] ^ acc[3]) == 0x12345678u) out[blockIdx.x] = 1;
}

int main() {
    hipStream_t s; hipStreamCreate(&s);
    const long long mb[NPH] = {50331648, 29360128, 33554432, 234881024, 117440512};
    const int L = 32;
    Step st{}; st.nlayers = L;
    long long total = 0;
    for (int l = 0; l < L; ++l)
        for (int p = 0; p < NPH; ++p) {
            void *d; if (hipMalloc(&d, mb[p]) != hipSuccess) { printf("alloc failed\n"); return 1; }
            hipMemset(d, 1, mb[p]);
            st.l[l].w[p] = (const u4 *)d; st.l[l].n16[p] = mb[p] / 16; total += mb[p];
        }
    unsigned *cnt, *out, *err; hipMalloc(&cnt, 64); hipMalloc(&out, 4096); hipMalloc(&err, 64);
    hipMemset(cnt, 0, 64); hipMemset(out, 0, 4096); hipMemset(err, 0, 64);
    const int lds = 100 * 1024;           // one workgroup per CU, as the real kernels
    hipFuncSetAttribute((const void *)phase_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    hipFuncSetAttribute((const void *)step_kernel<false, false>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    hipFuncSetAttribute((const void *)step_kernel<false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    hipFuncSetAttribute((const void *)step_kernel<true, false>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    hipFuncSetAttribute((const void *)step_kernel<true, true>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    hipDeviceSynchronize();
    printf("bytes per layer %.1f MB; at 8 TB/s %.1f us, at 5.6 TB/s %.1f us per layer\n", total / L / 1e6, total / L / 8e6, total / L / 5.6e6);
    const int reps = 6;
    auto time_it = [&](auto run) {
        run(); hipStreamSynchronize(s);
        double best = 1e30;
        for (int r = 0; r < reps; ++r) {
            auto t0 = std::chrono::steady_clock::now();
            run(); hipStreamSynchronize(s);
            double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count();
            if (us < best) best = us;
        }
        return best / L;
    };
    // A: launches in a graph
    hipGraph_t g; hipGraphExec_t ge;
    hipStreamBeginCapture(s, hipStreamCaptureModeGlobal);
    for (int l = 0; l < L; ++l)
        for (int p = 0; p < NPH; ++p) hipLaunchKernelGGL(phase_kernel, dim3(NWG), dim3(NT), lds, s, st.l[l].w[p], st.l[l].n16[p], out);
    hipStreamEndCapture(s, &g);
    hipGraphInstantiate(&ge, g, nullptr, nullptr, 0);
    const double a = time_it([&] { hipGraphLaunch(ge, s); });
    printf("A  launches (hipGraph, %d kernels per layer):            %7.2f us per layer\n", NPH, a);
    unsigned base = 0;
    auto persist = [&](auto kern, const char *name) {
        const double t = time_it([&] {
            hipLaunchKernelGGL(kern, dim3(NWG), dim3(NT), lds, s, st, cnt, base, out, err);
            base += (unsigned)(L * NPH * NWG);
        });
        unsigned e = 0; hipMemcpy(&e, err, 4, hipMemcpyDeviceToHost);
        printf("%s %7.2f us per layer  (%+.2f vs A)%s\n", name, t, t - a, e ? "  [BARRIER TIMEOUT]" : "");
    };
    persist(step_kernel<false, false>, "B  one launch, grid barriers, no fences:              ");
    persist(step_kernel<false, true>,  "B' one launch, grid barriers, release/acquire fences: ");
    persist(step_kernel<true, false>,  "C  B + next phase's loads before the barrier:         ");
    persist(step_kernel<true, true>,   "C' B' + next phase's loads before the barrier:        ");
    {   // D: the XCD-hierarchical barrier
        unsigned *hb; hipMalloc(&hb, 4096); hipMemset(hb, 0, 4096);
        hipFuncSetAttribute((const void *)step_kernel_xcd<false>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
        hipFuncSetAttribute((const void *)step_kernel_xcd<true>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
        unsigned ebase = 0;
        auto persist_x = [&](auto kern, const char *name) {
            const double t = time_it([&] {
                hipLaunchKernelGGL(kern, dim3(NWG), dim3(NT), lds, s, st, hb, ebase, out, err);
                ebase += (unsigned)(L * NPH);
            });
            unsigned e = 0; hipMemcpy(&e, err, 4, hipMemcpyDeviceToHost);
            printf("%s %7.2f us per layer  (%+.2f vs A)%s\n", name, t, t - a, e ? "  [BARRIER TIMEOUT]" : "");
        };
        persist_x(step_kernel_xcd<false>, "D  one launch, XCD-hierarchical barriers, no fences:    ");
        persist_x(step_kernel_xcd<true>,  "D' D + next phase's loads before the barrier:          ");
    }
    const double a2 = time_it([&] { hipGraphLaunch(ge, s); });
    printf("A  again:                                                %7.2f us per layer\n", a2);
    // sweep: the gate / up stream alone (235 MB, one launch per layer's buffer, 32 launches in a graph)
    printf("sweep: 235 MB per launch, us per launch and TB/s by workgroups x loads in flight per wave (1 KB each)\n");
    auto sweep = [&](auto kern, int nwg, int uu) {
        hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
        hipGraph_t g2; hipGraphExec_t ge2;
        hipStreamBeginCapture(s, hipStreamCaptureModeGlobal);
        for (int l = 0; l < L; ++l) hipLaunchKernelGGL(kern, dim3(nwg), dim3(NT), lds, s, st.l[l].w[3], st.l[l].n16[3], out);
        hipStreamEndCapture(s, &g2);
        hipGraphInstantiate(&ge2, g2, nullptr, nullptr, 0);
        const double t = time_it([&] { hipGraphLaunch(ge2, s); });
        printf("   %3d workgroups, %2d in flight: %6.2f us  %5.2f TB/s\n", nwg, uu, t, mb[3] / t / 1e6);
        hipGraphExecDestroy(ge2); hipGraphDestroy(g2);
    };
    for (int nwg : {192, 224, 256, 512}) {
        sweep(sweep_kernel<4>, nwg, 4);
        sweep(sweep_kernel<8>, nwg, 8);
        sweep(sweep_kernel<16>, nwg, 16);
    }
    return 0;
}
