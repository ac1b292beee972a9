// Decode-step attention (row D3/D4): for one new token per batch row,
//   rotary(q, k) at position slot - kstart[b]  ->  append k, v to the cache at `slot`  ->
//   softmax(q K^T * scale over keys kstart[b]..slot) V.
// Workgroup = GP query heads of one kv group of one batch row: GP = 1 at small batch (32 workgroups
// per row instead of 8: the step is latency-bound, so spread it), GP = group at large batch (K/V of a
// kv head are read once per row).  K/V go straight to registers (cdna_hip_programming.md Appendix B
// "Attention decode"): two lanes per key for the scores, one 8-wide column slice per lane for PV.
// The new key/value are used from LDS, so nothing depends on in-launch global visibility.
// slot = T0 + *step is read from device memory so the same launch can be replayed from a hipGraph.
#include "common.h"
#include <cstdlib>

namespace opus {

constexpr int MAXG = 8;

__device__ __forceinline__ float block_reduce(float v, bool is_max, float *scratch) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const float t = __shfl_xor(v, o, 64);
        v = is_max ? fmaxf(v, t) : v + t;
    }
    __syncthreads();
    if (lane == 0) scratch[wave] = v;
    __syncthreads();
    float r = scratch[0];
#pragma unroll
    for (int w = 1; w < 4; ++w) r = is_max ? fmaxf(r, scratch[w]) : r + scratch[w];
    return r;
}

template <int HD, int GP>
__global__ __launch_bounds__(256) void attn_decode_kernel(const half_t *__restrict__ qkv, const float *__restrict__ cs,
                                                          const int32_t *__restrict__ kstart_p,
                                                          const int32_t *__restrict__ step_p, int T0, int nh, int nkv,
                                                          half_t *__restrict__ kc, half_t *__restrict__ vc,
                                                          int64_t cache_sb, int64_t cache_sh, int ctx_cap, float scale,
                                                          half_t *__restrict__ out) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    constexpr int HALF = HD / 2, DV = HD / 8, PARTS = 256 / DV, HC = DV / 2;
    const int G = nh / nkv;
    float *sq = sm;                       // [GP][HD]   rotated query heads (fp16-rounded)
    float *sk = sq + GP * HD;             // [HD]       rotated new key
    float *sv = sk + HD;                  // [HD]       new value
    float *sc = sv + HD;                  // [GP][ctx_cap]
    float *red = sc + GP * ctx_cap;       // [PARTS][GP][HD]
    __shared__ float scratch[4];

    const int b = blockIdx.y, tid = threadIdx.x;
    const int h0 = blockIdx.x * GP;       // first query head of this workgroup
    const int kvh = h0 / G;
    const bool writer = (h0 % G) == 0;    // one workgroup per kv head appends to the cache
    const int slot = T0 + *step_p;
    const int kstart = kstart_p[b];
    const int pos = slot - kstart;
    const int64_t ld = (int64_t)(nh + 2 * nkv) * HD;
    const half_t *row = qkv + (int64_t)b * ld;
    half_t *kcb = kc + b * cache_sb + kvh * cache_sh;
    half_t *vcb = vc + b * cache_sb + kvh * cache_sh;

    // ---- prefetch: the cached K rows of the first 128 keys (two lanes per key) and the first VP value rows of
    // this thread's key partition are requested BEFORE the rotary / staging phase, so the three dependent
    // global-memory round trips of the step (q/k/v row, K, V) overlap into one.
    const int nkeys = slot - kstart + 1;
    const int hh = tid & 1;
    constexpr int VP = 8;
    const int dv = tid % DV, part = tid / DV;
    h8 kpre[HC], vpre[VP];
    {
        const int j = tid >> 1;
        const int jc = j < nkeys - 1 ? j : 0;
        const h8 *kr = reinterpret_cast<const h8 *>(kcb + (int64_t)(kstart + jc) * HD) + hh * HC;
#pragma unroll
        for (int c = 0; c < HC; ++c) kpre[c] = kr[c];
#pragma unroll
        for (int u = 0; u < VP; ++u) {
            const int jv = part + u * PARTS;
            const int jvc = jv < nkeys - 1 ? jv : 0;
            vpre[u] = *reinterpret_cast<const h8 *>(vcb + (int64_t)(kstart + jvc) * HD + dv * 8);
        }
    }

    // ---- rotary on the query heads and the new key; stage k, v ----
    for (int i = tid; i < (GP + 1) * HALF; i += 256) {
        const int j = i / HALF, d = i % HALF;
        const half_t *src = j < GP ? row + (int64_t)(h0 + j) * HD : row + (int64_t)(nh + kvh) * HD;
        const float c = cs[((int64_t)pos * HALF + d) * 2], sn = cs[((int64_t)pos * HALF + d) * 2 + 1];
        const float a = (float)src[d], bb = (float)src[d + HALF];
        const half_t lo = (half_t)(a * c - bb * sn), hi = (half_t)(bb * c + a * sn);
        if (j < GP) {
            sq[j * HD + d] = (float)lo;
            sq[j * HD + d + HALF] = (float)hi;
        } else {
            sk[d] = (float)lo;
            sk[d + HALF] = (float)hi;
            if (writer) {
                kcb[(int64_t)slot * HD + d] = lo;
                kcb[(int64_t)slot * HD + d + HALF] = hi;
            }
        }
    }
    for (int d = tid; d < HD; d += 256) {
        const half_t v = row[(int64_t)(nh + nkv + kvh) * HD + d];
        sv[d] = (float)v;
        if (writer) vcb[(int64_t)slot * HD + d] = v;
    }
    __syncthreads();

    // ---- scores: two lanes per key (each half of the head dim), cached keys then the new one ----
    for (int j0 = 0; j0 < nkeys; j0 += 128) {
        const int j = j0 + (tid >> 1);
        float acc[GP];
#pragma unroll
        for (int gi = 0; gi < GP; ++gi) acc[gi] = 0.f;
        if (j < nkeys - 1) {
            h8 kv[HC];
            if (j0 == 0) {
#pragma unroll
                for (int c = 0; c < HC; ++c) kv[c] = kpre[c];
            } else {
                const h8 *kr = reinterpret_cast<const h8 *>(kcb + (int64_t)(kstart + j) * HD) + hh * HC;
#pragma unroll
                for (int c = 0; c < HC; ++c) kv[c] = kr[c];
            }
#pragma unroll
            for (int c = 0; c < HC; ++c)
#pragma unroll
                for (int gi = 0; gi < GP; ++gi)
#pragma unroll
                    for (int e = 0; e < 8; ++e) acc[gi] += (float)kv[c][e] * sq[gi * HD + (hh * HC + c) * 8 + e];
        } else if (j == nkeys - 1) {
#pragma unroll
            for (int c = 0; c < HC; ++c)
#pragma unroll
                for (int gi = 0; gi < GP; ++gi)
#pragma unroll
                    for (int e = 0; e < 8; ++e)
                        acc[gi] += sk[(hh * HC + c) * 8 + e] * sq[gi * HD + (hh * HC + c) * 8 + e];
        }
#pragma unroll
        for (int gi = 0; gi < GP; ++gi) {
            const float t = acc[gi] + __shfl_xor(acc[gi], 1, 64);
            if (hh == 0 && j < nkeys) sc[gi * ctx_cap + j] = t * scale;
        }
    }
    __syncthreads();

    // ---- softmax per head (fp32); the GP heads share each block-wide reduction (2 barriers instead of 2 GP) ----
    float linv[GP], mx[GP], sum[GP];
    __shared__ float redv[4][GP];
    auto reduce_all = [&](float (&v)[GP], bool is_max) {
#pragma unroll
        for (int gi = 0; gi < GP; ++gi) {
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) {
                const float t = __shfl_xor(v[gi], o, 64);
                v[gi] = is_max ? fmaxf(v[gi], t) : v[gi] + t;
            }
        }
        __syncthreads();
        if ((tid & 63) == 0) {
#pragma unroll
            for (int gi = 0; gi < GP; ++gi) redv[tid >> 6][gi] = v[gi];
        }
        __syncthreads();
#pragma unroll
        for (int gi = 0; gi < GP; ++gi) {
            float r = redv[0][gi];
#pragma unroll
            for (int w = 1; w < 4; ++w) r = is_max ? fmaxf(r, redv[w][gi]) : r + redv[w][gi];
            v[gi] = r;
        }
    };
#pragma unroll
    for (int gi = 0; gi < GP; ++gi) {
        mx[gi] = -INFINITY;
        for (int j = tid; j < nkeys; j += 256) mx[gi] = fmaxf(mx[gi], sc[gi * ctx_cap + j]);
    }
    reduce_all(mx, true);
#pragma unroll
    for (int gi = 0; gi < GP; ++gi) {
        sum[gi] = 0.f;
        for (int j = tid; j < nkeys; j += 256) {
            const float e = __expf(sc[gi * ctx_cap + j] - mx[gi]);
            // P is rounded to fp16 before the PV product, as the prefill kernel and HF (softmax .to(q.dtype))
            sc[gi * ctx_cap + j] = (float)(half_t)e;
            sum[gi] += e;
        }
    }
    reduce_all(sum, false);
#pragma unroll
    for (int gi = 0; gi < GP; ++gi) linv[gi] = 1.0f / sum[gi];
    __syncthreads();

    // ---- O = P V : thread = (8-wide column slice, key partition) ----
    float acc[GP][8];
#pragma unroll
    for (int gi = 0; gi < GP; ++gi)
#pragma unroll
        for (int e = 0; e < 8; ++e) acc[gi][e] = 0.f;
#pragma unroll
    for (int u = 0; u < VP; ++u) {
        const int j = part + u * PARTS;
        if (j < nkeys - 1) {
#pragma unroll
            for (int gi = 0; gi < GP; ++gi) {
                const float pj = sc[gi * ctx_cap + j];
#pragma unroll
                for (int e = 0; e < 8; ++e) acc[gi][e] += pj * (float)vpre[u][e];
            }
        }
    }
    for (int j = part + VP * PARTS; j < nkeys - 1; j += PARTS) {
        const h8 vv = *reinterpret_cast<const h8 *>(vcb + (int64_t)(kstart + j) * HD + dv * 8);
#pragma unroll
        for (int gi = 0; gi < GP; ++gi) {
            const float pj = sc[gi * ctx_cap + j];
#pragma unroll
            for (int e = 0; e < 8; ++e) acc[gi][e] += pj * (float)vv[e];
        }
    }
    if (part == (nkeys - 1) % PARTS) {
#pragma unroll
        for (int gi = 0; gi < GP; ++gi) {
            const float pj = sc[gi * ctx_cap + nkeys - 1];
#pragma unroll
            for (int e = 0; e < 8; ++e) acc[gi][e] += pj * sv[dv * 8 + e];
        }
    }
#pragma unroll
    for (int gi = 0; gi < GP; ++gi)
#pragma unroll
        for (int e = 0; e < 8; ++e) red[(part * GP + gi) * HD + dv * 8 + e] = acc[gi][e];
    __syncthreads();
    for (int i = tid; i < GP * HD; i += 256) {
        const int gi = i / HD, d = i % HD;
        float s = 0.f;
        for (int pp = 0; pp < PARTS; ++pp) s += red[(pp * GP + gi) * HD + d];
        float li = linv[0];
#pragma unroll
        for (int q = 1; q < GP; ++q) li = (q == gi) ? linv[q] : li;
        out[(int64_t)b * nh * HD + (int64_t)(h0 + gi) * HD + d] = (half_t)(s * li);
    }
}

template <int HD, int GP>
static hipError_t launch_t(const half_t *qkv, const float *cs, const int32_t *kstart, const int32_t *step, int T0, int B,
                           int nh, int nkv, half_t *kc, half_t *vc, int64_t cache_sb, int64_t cache_sh, int ctx_cap,
                           float scale, half_t *out, hipStream_t s) {
    const int parts = 256 / (HD / 8);
    const size_t lds = ((size_t)GP * HD + 2 * HD + (size_t)GP * ctx_cap + (size_t)parts * GP * HD) * sizeof(float);
    if (lds > 150 * 1024) return hipErrorInvalidValue;
    if (lds > 48 * 1024) {
        hipError_t ea = ensure_dyn_lds(reinterpret_cast<const void *>(&attn_decode_kernel<HD, GP>), lds);
        if (ea != hipSuccess) return ea;
    }
    OPUS_LAUNCH(KC_ATTN_DECODE, (attn_decode_kernel<HD, GP>), dim3(nh / GP, B), dim3(256), lds, s, qkv, cs, kstart, step, T0, nh,
                       nkv, kc, vc, cache_sb, cache_sh, ctx_cap, scale, out);
    return hipGetLastError();
}

template <int HD>
static hipError_t launch_hd(const half_t *qkv, const float *cs, const int32_t *kstart, const int32_t *step, int T0, int B,
                            int nh, int nkv, half_t *kc, half_t *vc, int64_t cache_sb, int64_t cache_sh, int ctx_cap,
                            float scale, half_t *out, hipStream_t s) {
    const int G = nh / nkv;
    // grouped form only when the per-head form would already fill the chip several times over
    static const int group_min = getenv("OPUS_ATTN_GROUP_MIN") ? atoi(getenv("OPUS_ATTN_GROUP_MIN")) : 256;   // tuning aid
    const bool grouped = G > 1 && (int64_t)B * nkv >= group_min;
#define OPUS_GO(GPV) return launch_t<HD, GPV>(qkv, cs, kstart, step, T0, B, nh, nkv, kc, vc, cache_sb, cache_sh, ctx_cap, scale, out, s)
    if (!grouped) OPUS_GO(1);
    switch (G) {
        case 2: OPUS_GO(2);
        case 4: OPUS_GO(4);
        case 8: OPUS_GO(8);
    }
    OPUS_GO(1);
#undef OPUS_GO
}

hipError_t launch_attn_decode(const half_t *qkv, const float *cs, const int32_t *kstart, const int32_t *step, int T0,
                              int B, int nh, int nkv, int hd, half_t *kc, half_t *vc, int64_t cache_sb,
                              int64_t cache_sh, int ctx_cap, float scale, half_t *out, hipStream_t s) {
    const int G = nh / nkv;
    if (G > MAXG || G * nkv != nh) return hipErrorInvalidValue;
    switch (hd) {
        case 16: return launch_hd<16>(qkv, cs, kstart, step, T0, B, nh, nkv, kc, vc, cache_sb, cache_sh, ctx_cap, scale, out, s);
        case 32: return launch_hd<32>(qkv, cs, kstart, step, T0, B, nh, nkv, kc, vc, cache_sb, cache_sh, ctx_cap, scale, out, s);
        case 64: return launch_hd<64>(qkv, cs, kstart, step, T0, B, nh, nkv, kc, vc, cache_sb, cache_sh, ctx_cap, scale, out, s);
        case 128: return launch_hd<128>(qkv, cs, kstart, step, T0, B, nh, nkv, kc, vc, cache_sb, cache_sh, ctx_cap, scale, out, s);
    }
    return hipErrorInvalidValue;
}

}  // namespace opus
