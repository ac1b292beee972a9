// Decode step of the Llama stack in ONE launch (rows D3/D4 of the path at batch <= 4): a persistent grid of one
// workgroup per CU walks  [qkv -> attention -> wo -> gate/up -> down] x layers -> lm_head  with a grid-wide
// barrier between phases instead of a kernel boundary.  The step is a pure weight stream (16 GB per token for
// Llama-3-8B, ~2 flop per byte), so what a kernel boundary costs is the drained memory pipe: here every wave
// requests the first chunks of its NEXT phase's weights before it reaches the barrier (weights do not depend on
// activations), so HBM keeps streaming while the barrier, the activation re-staging and the reductions happen.
//
// Work split of a GEMM phase ("stream-K inside the workgroup"): workgroup b owns a contiguous run of weight
// panels (16 output columns each, whole gate/up pairs for the fused SwiGLU phase); because the panel-tiled weight
// layout is [panel][64-k chunk][2 KB], that run is ONE contiguous byte range, which is cut into equal chunk
// ranges over the waves.  A wave accumulates per panel, parks a partial tile in LDS whenever its range crosses
// a panel boundary, and the partials are summed in wave order (fixed order: bitwise reproducible).
// Activations (<= 4 rows) are staged once per phase in LDS by the whole workgroup, the RMSNorm sum of squares
// is taken on the way (norm weights are folded into the GEMM weights at load time).
//
// Reference semantics: transformers LlamaDecoderLayer.forward as driven by model/language_model/opus_llama.py:95-132.
#include "common.h"
#include <algorithm>
#include <cstdlib>
#include <type_traits>
#include <hip/hip_ext.h>

namespace opus {

namespace {

constexpr int MAXM = STACK_MAX_B;  // rows
constexpr unsigned long long SPIN_LIMIT = 200000000ull;   // 2 s of the 100 MHz wall clock

struct GemmDesc {
    const half_t *W;
    int npanels, chunks, pb;       // pb = 2: panels come in gate/up pairs that must stay in one workgroup
};

struct Range {
    const half_t *base;            // chunk i of the workgroup's span at base + i * 1024 (this lane's 16 B)
    int r0, r1;                    // this wave's chunk range inside the span
    int P, pbase;                  // panels of this workgroup, first panel
};

typedef const __attribute__((address_space(1))) h8 *gh8_t;

enum Kind { K_QKV = 0, K_RESID = 1, K_GATEUP = 2, K_LOGITS = 3 };

__device__ __forceinline__ float silu_f(float v) { return v / (1.0f + __expf(-v)); }

}  // namespace

// Activations that cross workgroups inside the launch (x, qkv, ctx, act) are written and read with agent-scope
// relaxed atomics: on gfx950 those are sc1 stores / loads that write through and read past the per-XCD L2, so the
// grid barrier needs no L2 write-back / invalidate (measured at ~4 us per barrier), only "stores retired, then count".
__device__ __forceinline__ unsigned long long ld64(const void *p) {
    return __hip_atomic_load(reinterpret_cast<const unsigned long long *>(p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ unsigned ld32(const void *p) {
    return __hip_atomic_load(reinterpret_cast<const unsigned *>(p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void st32(void *p, unsigned v) {
    __hip_atomic_store(reinterpret_cast<unsigned *>(p), v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ unsigned pack2(float a, float b) {
    const h2 v = h2{(half_t)a, (half_t)b};
    return __builtin_bit_cast(unsigned, v);
}

// NT threads per workgroup (one workgroup per CU); SU 64-k chunks per register set, two sets in flight per wave:
// 16 waves x 2 x 4 chunks or 8 waves x 2 x 8 chunks = 256 KB of weights on the wire per CU either way.
template <int HD, int NT>
__global__ __launch_bounds__(NT) void decode_stack_kernel(StackParams p) {
    constexpr int SU = 4;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    half_t *xs = reinterpret_cast<half_t *>(smem);                   // [M][K] staged activations (or attention scratch)
    float *red = reinterpret_cast<float *>(smem + p.xs_bytes);        // [wave][seg][M][16] partial tiles
    float *sm_small = red + p.red_floats;
    float *wss = sm_small;                                            // [MAXM][16] per-wave sums of squares
    int *wfp = reinterpret_cast<int *>(sm_small + MAXM * 16);         // [16] first panel touched by wave
    int *wlp = wfp + 16;                                              // [16] last panel touched by wave (-1: none)
    float *scr = reinterpret_cast<float *>(wlp + 16);                 // [16] block reductions
    float *rstd = scr + 16;                                           // [MAXM]

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    constexpr int nw = NT >> 6;
    const int NB = gridDim.x, bid = blockIdx.x;
    const int g = lane >> 4, li = lane & 15;
    const int M = p.B, H = p.H, F = p.F;
    const int QD = p.nh * HD, QKV = (p.nh + 2 * p.nkv) * HD;
    const int SEG = p.seg_max;
    unsigned phase = 0;
    bool dead = false;

    h8 wlA[SU], whA[SU], wlB[SU], whB[SU];

    // optional timeline of one workgroup (100 MHz wall clock), for tuning: p.trace[k] = time of mark k
    int tmark = 0;
    bool tracing = false;
    auto mark = [&]() __attribute__((always_inline)) {
        if (tracing && tid == 0 && tmark < 64) p.trace[tmark++] = wall_clock64();
    };

    auto range_of = [&](const GemmDesc &d) __attribute__((always_inline)) {
        const int units = d.npanels / d.pb;
        const int u0 = (int)((int64_t)units * bid / NB), u1 = (int)((int64_t)units * (bid + 1) / NB);
        Range r;
        // (the divisions run on the vector ALU; hand the wave-uniform results back to scalar registers)
        r.P = __builtin_amdgcn_readfirstlane((u1 - u0) * d.pb);
        r.pbase = __builtin_amdgcn_readfirstlane(u0 * d.pb);
        const int S = r.P * d.chunks;
        r.r0 = __builtin_amdgcn_readfirstlane((int)((int64_t)S * wave / nw));
        r.r1 = __builtin_amdgcn_readfirstlane((int)((int64_t)S * (wave + 1) / nw));
        r.base = d.W + ((int64_t)r.pbase * d.chunks) * 1024 + lane * 8;
        return r;
    };
    auto wload = [&](h8 (&wl)[SU], h8 (&wh)[SU], const Range &r, int i) __attribute__((always_inline)) {
#pragma unroll
        for (int u = 0; u < SU; ++u) {
            if (i + u < r.r1) {
                // weight pointers come out of a device table: say "global" so the loads do not go through the
                // flat path (which would also tick the LDS counter)
                const gh8_t ptr = (gh8_t)(uintptr_t)(r.base + (int64_t)(i + u) * 1024);
                wl[u] = __builtin_nontemporal_load(ptr);
                wh[u] = __builtin_nontemporal_load(ptr + 64);
            }
        }
    };
    // the first two register sets of a phase: requested before the grid barrier in front of it
    auto prefetch = [&](const Range &r) __attribute__((always_inline)) {
        wload(wlA, whA, r, r.r0);
        wload(wlB, whB, r, r.r0 + SU);
    };

    // ---- grid barrier: arrive after a phase's stores, wait before reading another workgroup's output ----
    // One counter takes the arrivals; the LAST arriver publishes the phase number to NFLAG "go" words 4 KB apart
    // and every workgroup polls only its own word: 256 pollers on one line slow the HBM channel behind that line
    // enough to make stragglers of the workgroups that are still streaming through it.
    constexpr int NFLAG = 16, FSTRIDE = 1024;                          // words
    unsigned *go = p.bar + 1024 + (bid % NFLAG) * FSTRIDE;
    auto arrive = [&]() __attribute__((always_inline)) {
        __syncthreads();                                              // every wave's (write-through) stores have retired
        ++phase;
        if (tid == 0) {
            const unsigned old = __hip_atomic_fetch_add(&p.bar[0], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (old == phase * (unsigned)NB - 1u) {
                for (int k = 0; k < NFLAG; ++k)
                    __hip_atomic_store(p.bar + 1024 + k * FSTRIDE, phase, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        }
    };
    auto wait = [&]() __attribute__((always_inline)) {
        if (tid == 0 && !dead) {
            const unsigned long long t0 = wall_clock64();
            while (__hip_atomic_load(go, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < phase) {
                __builtin_amdgcn_s_sleep(8);
                // every spin has an exit: a peer that never arrives (not co-resident, faulted) must not hang the GPU
                if (wall_clock64() - t0 > SPIN_LIMIT ||
                    __hip_atomic_load(&p.bar[2], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u) {
                    __hip_atomic_store(&p.bar[2], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    dead = true;
                    break;
                }
            }
        }
        __syncthreads();
    };

    // ---- one GEMM phase: out = epilogue(A[M,K] . W^T) for this workgroup's panels ----
    // on entry wlA/whA/wlB/whB hold chunks [r0, r0 + 2 SU) of THIS phase (requested before the barrier)
    auto gemm_phase = [&](auto kind_tag, const GemmDesc &d, const Range &r, bool need_wait, const void *Asrc,
                          const GemmDesc *nxt, Range *nxt_r) __attribute__((always_inline)) {
        constexpr int KIND = decltype(kind_tag)::value;
        constexpr bool NORM = KIND == K_QKV || KIND == K_GATEUP || KIND == K_LOGITS;
        const int K = d.chunks * 64;
        mark();
        if (need_wait) wait();
        mark();
        // ---- stage the activation rows in LDS (fp16), sum of squares on the way ----
        for (int m = 0; m < M; ++m) {
            if (NORM) {
                const float *src = reinterpret_cast<const float *>(Asrc) + (int64_t)m * K;
                float s = 0.f;
                for (int i = tid; i < (K >> 1); i += NT) {
                    const float2 v = __builtin_bit_cast(float2, ld64(src + 2 * i));
                    s += v.x * v.x + v.y * v.y;
                    *reinterpret_cast<h2 *>(xs + (int64_t)m * K + 2 * i) = h2{(half_t)v.x, (half_t)v.y};
                }
#pragma unroll
                for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
                if (lane == 0) wss[m * 16 + wave] = s;
            } else {
                const half_t *src = reinterpret_cast<const half_t *>(Asrc) + (int64_t)m * K;
                for (int i = tid; i < (K >> 2); i += NT)
                    *reinterpret_cast<unsigned long long *>(xs + (int64_t)m * K + 4 * i) = ld64(src + 4 * i);
            }
        }
        if (lane == 0) {
            wfp[wave] = r.r0 / d.chunks;
            wlp[wave] = r.r1 > r.r0 ? (r.r1 - 1) / d.chunks : -1;
        }
        // finalize geometry: 16 lanes per output, one partial per lane; the residual this thread will add is
        // requested now, a whole phase ahead of its use
        const int nunit = r.P / d.pb, nout = nunit * M * 16;
        constexpr int opp = NT >> 4;   // outputs per pass
        const int fw = tid & 15, fg = tid >> 4;
        float res0 = 0.f;
        if (KIND == K_RESID && fw == 0 && fg < nout) {
            const int col = fg & 15, t = fg >> 4, m = t % M, q = t / M;
            res0 = __builtin_bit_cast(float, ld32(p.x + (int64_t)m * H + (r.pbase + q) * 16 + col));
        }
        __syncthreads();
        if (NORM && tid < M) {
            float ss = 0.f;
            for (int w = 0; w < nw; ++w) ss += wss[tid * 16 + w];
            rstd[tid] = rsqrtf(ss / (float)K + p.eps);
        }
        mark();

        // ---- stream this wave's chunk range ----
        const int mr = li < M ? li : M - 1;                            // rows >= M duplicate the last row: never stored
        const h8 *xr = reinterpret_cast<const h8 *>(xs + (int64_t)mr * K + g * 8);
        f4 acc = f4{0.f, 0.f, 0.f, 0.f};
        int c = __builtin_amdgcn_readfirstlane(r.r0 % d.chunks), seg = 0;
        auto flush = [&]() __attribute__((always_inline)) {
            if (g == 0) {
#pragma unroll
                for (int q = 0; q < MAXM; ++q)
                    if (q < M) red[((wave * SEG + seg) * M + q) * 16 + li] = acc[q];
            }
            acc = f4{0.f, 0.f, 0.f, 0.f};
            ++seg;
        };
        auto step = [&](const h8 &wl, const h8 &wh) __attribute__((always_inline)) {
            const h8 al = xr[c * 8], ah = xr[c * 8 + 4];
            acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(al, wl, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, wh, acc, 0, 0, 0);
            if (++c == d.chunks) {                                     // panel boundary inside the range
                flush();
                c = 0;
            }
        };
        auto compute = [&](const h8 (&wl)[SU], const h8 (&wh)[SU], int i) __attribute__((always_inline)) {
#pragma unroll
            for (int u = 0; u < SU; ++u)
                if (i + u < r.r1) step(wl[u], wh[u]);                  // wave-uniform
        };
        int i = r.r0;
        // steady state: no guards, so the loads of the set that was just refilled stay in flight (counted
        // vmcnt) while the other set is consumed
        for (; i + 4 * SU <= r.r1; i += 2 * SU) {
#pragma unroll
            for (int u = 0; u < SU; ++u) step(wlA[u], whA[u]);
#pragma unroll
            for (int u = 0; u < SU; ++u) {
                const gh8_t ptr = (gh8_t)(uintptr_t)(r.base + (int64_t)(i + 2 * SU + u) * 1024);
                wlA[u] = __builtin_nontemporal_load(ptr);
                whA[u] = __builtin_nontemporal_load(ptr + 64);
            }
#pragma unroll
            for (int u = 0; u < SU; ++u) step(wlB[u], whB[u]);
#pragma unroll
            for (int u = 0; u < SU; ++u) {
                const gh8_t ptr = (gh8_t)(uintptr_t)(r.base + (int64_t)(i + 3 * SU + u) * 1024);
                wlB[u] = __builtin_nontemporal_load(ptr);
                whB[u] = __builtin_nontemporal_load(ptr + 64);
            }
        }
        // tail: fewer than 4 sets left, A and B hold the first two
        compute(wlA, whA, i);
        if (i + 2 * SU < r.r1) wload(wlA, whA, r, i + 2 * SU);
        compute(wlB, whB, i + SU);
        if (i + 3 * SU < r.r1) wload(wlB, whB, r, i + 3 * SU);
        compute(wlA, whA, i + 2 * SU);
        compute(wlB, whB, i + 3 * SU);
        if (c != 0 && r.r1 > r.r0) flush();
        // the next phase's first weights go on the wire now; they land while the partials are reduced, the
        // epilogue is stored and the grid barrier is crossed
        if (nxt) {
            *nxt_r = range_of(*nxt);
            prefetch(*nxt_r);
        }
        mark();
        __syncthreads();
        mark();

        // ---- finalize: 16 lanes per output gather the partial tiles (fixed shuffle tree: bitwise reproducible) ----
        for (int o0 = 0; o0 < nout; o0 += opp) {
            const int o = o0 + fg;
            const bool valid = o < nout;
            const int col = o & 15, t = o >> 4;
            const int m = valid ? t % M : 0, q = valid ? t / M : 0;
            float v0 = 0.f, v1 = 0.f;
            if (valid && fw < nw) {
                const int f0 = wfp[fw], l0 = wlp[fw];
                const int qa = KIND == K_GATEUP ? 2 * q : q;
                if (qa >= f0 && qa <= l0) v0 = red[((fw * SEG + qa - f0) * M + m) * 16 + col];
                if (KIND == K_GATEUP && qa + 1 >= f0 && qa + 1 <= l0) v1 = red[((fw * SEG + qa + 1 - f0) * M + m) * 16 + col];
            }
#pragma unroll
            for (int s = 8; s > 0; s >>= 1) {
                v0 += __shfl_xor(v0, s, 64);
                if (KIND == K_GATEUP) v1 += __shfl_xor(v1, s, 64);
            }
            const float rs = NORM ? rstd[m] : 1.0f;
            float v;
            if (KIND == K_GATEUP) v = silu_f(v0 * rs) * (v1 * rs);
            else v = v0 * rs;
            const int n = (r.pbase / d.pb + q) * 16 + col;            // output column
            if (KIND == K_RESID) {
                if (valid && fw == 0) {
                    const float xo = o0 == 0 ? res0 : __builtin_bit_cast(float, ld32(p.x + (int64_t)m * H + n));
                    st32(p.x + (int64_t)m * H + n, __builtin_bit_cast(unsigned, xo + v));
                }
            } else if (KIND == K_LOGITS) {
                if (valid && fw == 0 && n < p.V) p.logits[(int64_t)m * p.V + n] = v;
            } else {
                const float vn = __shfl_xor(v, 16, 64);               // the neighbouring column (o ^ 1)
                if (valid && fw == 0 && !(col & 1)) {
                    half_t *dst = KIND == K_QKV ? p.qkv + (int64_t)m * QKV + n : p.act + (int64_t)m * F + n;
                    st32(dst, pack2(v, vn));
                }
            }
        }
        mark();
        arrive();
        mark();
    };

    // ---- attention phase: one (row, query head) task per workgroup ----
    auto block_reduce = [&](float v, bool is_max) __attribute__((always_inline)) {
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            const float t = __shfl_xor(v, o, 64);
            v = is_max ? fmaxf(v, t) : v + t;
        }
        __syncthreads();
        if (lane == 0) scr[wave] = v;
        __syncthreads();
        float rr = scr[0];
        for (int w = 1; w < nw; ++w) rr = is_max ? fmaxf(rr, scr[w]) : rr + scr[w];
        return rr;
    };
    // returns true when this workgroup had a task (then it did NOT prefetch the wo weights)
    auto attn_phase = [&](int layer, const Range &r_wo) __attribute__((always_inline)) {
        constexpr int HALF = HD / 2, DV = HD / 8, HC = (DV + 1) / 2;
        constexpr int HH = DV >= 2 ? 2 : 1;                            // lanes per key in the score pass
        constexpr int VP = 2;                                          // value rows per thread requested up front
        constexpr int PARTS = NT / DV;
        const int G = p.nh / p.nkv;
        float *sq = reinterpret_cast<float *>(smem);                   // [HD] rotated query (fp16-rounded)
        float *sk = sq + HD, *sv = sk + HD;                            // rotated new key, new value
        float *sc = sv + HD;                                           // [ctx_cap] scores / probabilities
        float *ared = sc + p.ctx_cap;                                  // [PARTS][HD]
        const int slot = p.T0 + *p.step;
        half_t *kcl = p.kc + (int64_t)layer * p.cache_sl, *vcl = p.vc + (int64_t)layer * p.cache_sl;
        const bool has_task = bid < M * p.nh;
        if (!has_task) {
            prefetch(r_wo);                                            // wo weights stream in under the attention
            mark();
            wait();
            mark();
            return false;
        }
        bool first = true;
        for (int t = bid; t < M * p.nh; t += NB) {
            const int b = t / p.nh, h = t % p.nh, kvh = h / G;
            const bool writer = (h % G) == 0;                          // one workgroup per kv head appends to the cache
            const int kstart = p.kstart[b];
            const int pos = slot - kstart, nkeys = pos + 1;
            const half_t *row = p.qkv + (int64_t)b * QKV;
            half_t *kcb = kcl + b * p.cache_sb + kvh * p.cache_sh;
            half_t *vcb = vcl + b * p.cache_sb + kvh * p.cache_sh;
            // the cached K / V rows were written by earlier launches: request them BEFORE waiting for this step's q
            const int hh = tid % HH, kper = NT / HH;
            const int dv = tid % DV, part = tid / DV;
            h8 kpre[HC], vpre[VP];
            {
                const int j = tid / HH;
                const int jc = j < nkeys - 1 ? j : 0;
                const h8 *kr = reinterpret_cast<const h8 *>(kcb + (int64_t)(kstart + jc) * HD) + hh * HC;
#pragma unroll
                for (int cI = 0; cI < HC; ++cI)
                    if (hh * HC + cI < DV) kpre[cI] = kr[cI];
#pragma unroll
                for (int u = 0; u < VP; ++u) {
                    const int jv = part + u * PARTS;
                    const int jvc = jv < nkeys - 1 ? jv : 0;
                    vpre[u] = *reinterpret_cast<const h8 *>(vcb + (int64_t)(kstart + jvc) * HD + dv * 8);
                }
            }
            if (first) {
                mark();
                wait();
                mark();
                first = false;
            }
            // rotary on q and the new key; stage k, v; append to the cache
            for (int i = tid; i < 2 * HALF; i += NT) {
                const int j = i / HALF, dd = i % HALF;
                const half_t *src = j == 0 ? row + (int64_t)h * HD : row + (int64_t)(p.nh + kvh) * HD;
                const float cc = p.cs[((int64_t)pos * HALF + dd) * 2], sn = p.cs[((int64_t)pos * HALF + dd) * 2 + 1];
                const unsigned short ua = __hip_atomic_load(reinterpret_cast<const unsigned short *>(src + dd), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                const unsigned short ub = __hip_atomic_load(reinterpret_cast<const unsigned short *>(src + dd + HALF), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                const float a = (float)__builtin_bit_cast(half_t, ua), bb = (float)__builtin_bit_cast(half_t, ub);
                float rl, rh;
                rotate_pair(a, bb, cc, sn, rl, rh);
                const half_t lo = (half_t)rl, hi = (half_t)rh;
                if (j == 0) {
                    sq[dd] = (float)lo;
                    sq[dd + HALF] = (float)hi;
                } else {
                    sk[dd] = (float)lo;
                    sk[dd + HALF] = (float)hi;
                    if (writer) {
                        kcb[(int64_t)slot * HD + dd] = lo;
                        kcb[(int64_t)slot * HD + dd + HALF] = hi;
                    }
                }
            }
            for (int dd = tid; dd < HD; dd += NT) {
                const unsigned short uv = __hip_atomic_load(reinterpret_cast<const unsigned short *>(row + (int64_t)(p.nh + p.nkv + kvh) * HD + dd),
                                                            __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                const half_t v = __builtin_bit_cast(half_t, uv);
                sv[dd] = (float)v;
                if (writer) vcb[(int64_t)slot * HD + dd] = v;
            }
            __syncthreads();
            // scores: HH lanes per key (each a slice of the head dim), cached keys then the new one
            for (int j0 = 0; j0 < nkeys; j0 += kper) {
                const int j = j0 + tid / HH;
                float a = 0.f;
                if (j < nkeys - 1) {
                    const h8 *kr = reinterpret_cast<const h8 *>(kcb + (int64_t)(kstart + j) * HD) + hh * HC;
#pragma unroll
                    for (int cI = 0; cI < HC; ++cI) {
                        if (hh * HC + cI < DV) {
                            const h8 kv = j0 == 0 ? kpre[cI] : kr[cI];
#pragma unroll
                            for (int e = 0; e < 8; ++e) a += (float)kv[e] * sq[(hh * HC + cI) * 8 + e];
                        }
                    }
                } else if (j == nkeys - 1) {
#pragma unroll
                    for (int cI = 0; cI < HC; ++cI)
                        if (hh * HC + cI < DV) {
#pragma unroll
                            for (int e = 0; e < 8; ++e) a += sk[(hh * HC + cI) * 8 + e] * sq[(hh * HC + cI) * 8 + e];
                        }
                }
                if (HH == 2) a += __shfl_xor(a, 1, 64);
                if (hh == 0 && j < nkeys) sc[j] = a * p.scale;
            }
            __syncthreads();
            // softmax (fp32), P rounded to fp16 before the PV product like the prefill kernel and HF
            float mx = -INFINITY;
            for (int j = tid; j < nkeys; j += NT) mx = fmaxf(mx, sc[j]);
            mx = block_reduce(mx, true);
            float sum = 0.f;
            for (int j = tid; j < nkeys; j += NT) {
                const float e = __expf(sc[j] - mx);
                sc[j] = (float)(half_t)e;
                sum += e;
            }
            sum = block_reduce(sum, false);
            const float linv = 1.0f / sum;
            // O = P V: thread = (8-wide column slice, key partition)
            float o8[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) o8[e] = 0.f;
#pragma unroll
            for (int u = 0; u < VP; ++u) {
                const int j = part + u * PARTS;
                if (j < nkeys - 1) {
                    const float pj = sc[j];
#pragma unroll
                    for (int e = 0; e < 8; ++e) o8[e] += pj * (float)vpre[u][e];
                }
            }
            for (int j = part + VP * PARTS; j < nkeys - 1; j += PARTS) {
                const h8 vv = *reinterpret_cast<const h8 *>(vcb + (int64_t)(kstart + j) * HD + dv * 8);
                const float pj = sc[j];
#pragma unroll
                for (int e = 0; e < 8; ++e) o8[e] += pj * (float)vv[e];
            }
            if (part == (nkeys - 1) % PARTS) {
                const float pj = sc[nkeys - 1];
#pragma unroll
                for (int e = 0; e < 8; ++e) o8[e] += pj * sv[dv * 8 + e];
            }
#pragma unroll
            for (int e = 0; e < 8; ++e) ared[part * HD + dv * 8 + e] = o8[e];
            __syncthreads();
            for (int dd = 2 * tid; dd < HD; dd += 2 * NT) {
                float s0 = 0.f, s1 = 0.f;
                for (int pp = 0; pp < PARTS; ++pp) {
                    s0 += ared[pp * HD + dd];
                    s1 += ared[pp * HD + dd + 1];
                }
                st32(p.ctx + (int64_t)b * QD + (int64_t)h * HD + dd, pack2(s0 * linv, s1 * linv));
            }
            __syncthreads();
        }
        return true;
    };

    // ---- the step ----
    const int chH = H >> 6, chF = F >> 6, chQ = QD >> 6;
    Range ra, rb;
    {
        const GemmDesc dq{p.layers[0].wqkv, QKV >> 4, chH, 1};
        ra = range_of(dq);
        prefetch(ra);
    }
    for (int l = 0; l < p.n_layers; ++l) {
        const StackLayer L = p.layers[l];
        tracing = p.trace != nullptr && l == 1 && bid == p.trace_block;
        const GemmDesc d_qkv{L.wqkv, QKV >> 4, chH, 1}, d_wo{L.wo, H >> 4, chQ, 1}, d_gu{L.wgu, (2 * F) >> 4, chH, 2},
            d_wd{L.wd, H >> 4, chF, 1};
        gemm_phase(std::integral_constant<int, K_QKV>{}, d_qkv, ra, l > 0, p.x, nullptr, nullptr);
        rb = range_of(d_wo);
        const bool did = attn_phase(l, rb);
        mark();
        arrive();
        if (did) prefetch(rb);
        gemm_phase(std::integral_constant<int, K_RESID>{}, d_wo, rb, true, p.ctx, &d_gu, &ra);
        gemm_phase(std::integral_constant<int, K_GATEUP>{}, d_gu, ra, true, p.x, &d_wd, &rb);
        GemmDesc d_next;
        if (l + 1 < p.n_layers) d_next = GemmDesc{p.layers[l + 1].wqkv, QKV >> 4, chH, 1};
        else d_next = GemmDesc{p.lm_head, (p.V + 15) >> 4, chH, 1};
        gemm_phase(std::integral_constant<int, K_RESID>{}, d_wd, rb, true, p.act, &d_next, &ra);
    }
    {
        const GemmDesc d_lm{p.lm_head, (p.V + 15) >> 4, chH, 1};
        gemm_phase(std::integral_constant<int, K_LOGITS>{}, d_lm, ra, true, p.x, nullptr, nullptr);
    }
    // the last workgroup out re-arms the barrier for the next launch (everyone is past every wait by then)
    if (tid == 0) {
        const unsigned done = __hip_atomic_fetch_add(&p.bar[1], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (done == (unsigned)NB - 1u) {
            for (int k = 0; k < NFLAG; ++k) __hip_atomic_store(p.bar + 1024 + k * FSTRIDE, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(&p.bar[0], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(&p.bar[1], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
}

// Host side ---------------------------------------------------------------------------------------
static int g_cus = 0;

template <int HD, int NT>
static hipError_t launch_hd(StackParams p, hipStream_t s) {
    constexpr int nw = NT / 64;
    if (!g_cus) {
        int dev = 0;
        hipError_t e = hipGetDevice(&dev);
        if (e != hipSuccess) return e;
        e = hipDeviceGetAttribute(&g_cus, hipDeviceAttributeMultiprocessorCount, dev);
        if (e != hipSuccess) return e;
    }
    const int NB = g_cus;
    const int QD = p.nh * HD, QKV = (p.nh + 2 * p.nkv) * HD;
    const int Kmax = std::max(std::max(p.H, p.F), QD);
    // partial-tile slots per wave: a wave's chunk range can straddle this many panels
    auto segs = [&](int npanels, int pb, int chunks) {
        const int units = npanels / pb, pmax = cdiv(units, NB) * pb;
        const int per_wave = cdiv((int64_t)pmax * chunks, nw);
        return (per_wave + chunks - 1) / chunks + 1;
    };
    int seg = segs(QKV >> 4, 1, p.H >> 6);
    seg = std::max(seg, segs(p.H >> 4, 1, QD >> 6));
    seg = std::max(seg, segs((2 * p.F) >> 4, 2, p.H >> 6));
    seg = std::max(seg, segs(p.H >> 4, 1, p.F >> 6));
    seg = std::max(seg, segs((p.V + 15) >> 4, 1, p.H >> 6));
    p.seg_max = seg;
    const int DV = HD / 8, PARTS = NT / DV;
    const size_t attn_bytes = ((size_t)3 * HD + p.ctx_cap + (size_t)PARTS * HD) * sizeof(float);
    size_t xs_bytes = std::max((size_t)p.B * Kmax * sizeof(half_t), attn_bytes);
    xs_bytes = (xs_bytes + 15) & ~(size_t)15;
    p.xs_bytes = (int)xs_bytes;
    p.red_floats = nw * seg * p.B * 16;
    size_t lds = xs_bytes + (size_t)p.red_floats * 4 + (MAXM * 16 + 16 + 16 + 16 + MAXM) * 4;
    if (lds > 160 * 1024) return hipErrorInvalidValue;
    // ask for more than half of a CU's LDS so that the dispatcher cannot stack two workgroups on one CU (they would
    // share that CU's memory pipeline and become the stragglers every barrier waits for)
    lds = std::max(lds, (size_t)96 * 1024);
    hipError_t ea = ensure_dyn_lds(reinterpret_cast<const void *>(&decode_stack_kernel<HD, NT>), lds);
    if (ea != hipSuccess) return ea;
    OPUS_LAUNCH(KC_STACK, (decode_stack_kernel<HD, NT>), dim3(NB), dim3(NT), lds, s, p);
    return hipGetLastError();
}

bool decode_stack_supported(int B, int H, int F, int nh, int nkv, int hd, int ctx_cap) {
    if (B < 1 || B > STACK_MAX_B) return false;
    if (hd != 16 && hd != 32 && hd != 64 && hd != 128) return false;
    if ((H & 63) || (F & 63) || ((nh * hd) & 63) || ((2 * F) & 31)) return false;
    const int Kmax = std::max(std::max(H, F), nh * hd);
    const size_t attn = ((size_t)3 * hd + ctx_cap + (size_t)(1024 / (hd / 8)) * hd) * 4;
    const size_t xs = std::max((size_t)B * Kmax * 2, attn);
    return xs + 16 * 8 * B * 16 * 4 + 1024 <= 150 * 1024;              // generous bound on the partial slots
}

hipError_t launch_decode_stack(const StackParams &p, int hd, hipStream_t s) {
    static const bool wide = getenv("OPUS_STACK_THREADS") && atoi(getenv("OPUS_STACK_THREADS")) == 1024;   // tuning aid
    switch (hd) {
        case 16: return wide ? launch_hd<16, 1024>(p, s) : launch_hd<16, 512>(p, s);
        case 32: return wide ? launch_hd<32, 1024>(p, s) : launch_hd<32, 512>(p, s);
        case 64: return wide ? launch_hd<64, 1024>(p, s) : launch_hd<64, 512>(p, s);
        case 128: return wide ? launch_hd<128, 1024>(p, s) : launch_hd<128, 512>(p, s);
    }
    return hipErrorInvalidValue;
}

}  // namespace opus
