// Decode-step attention (row D3/D4): for one new token per batch row,
//   rotary(q, k) at position slot - kstart[b]  ->  append k, v to the cache at `slot`  ->
//   softmax(q K^T * scale over keys kstart[b]..slot) V     (GQA: one workgroup per (kv head, row)
//   serves its `group` query heads so K/V are read once).
// HBM-bound on the KV cache (2 * ctx * head_dim * 2 B per kv head); K/V go straight to registers
// (cdna_hip_programming.md Appendix B "Attention decode"): one key per lane for the scores, one
// 8-wide column slice per lane for PV.  slot = T0 + *step is read from device memory so the same
// launch can be replayed from a hipGraph.
#include "common.h"

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

template <int HD>
__global__ __launch_bounds__(256) void attn_decode_kernel(const half_t *__restrict__ qkv, const float *__restrict__ cs,
                                                          const int32_t *__restrict__ kstart_p,
                                                          const int32_t *__restrict__ step_p, int T0, int nh, int nkv,
                                                          half_t *__restrict__ kc, half_t *__restrict__ vc,
                                                          int64_t cache_sb, int64_t cache_sh, int ctx_cap, float scale,
                                                          half_t *__restrict__ out) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    constexpr int HALF = HD / 2, DV = HD / 8, PARTS = 256 / DV;
    const int G = nh / nkv;
    float *sq = sm;                       // [G][HD]
    float *sc = sq + MAXG * HD;           // [G][ctx_cap]
    float *red = sc + MAXG * ctx_cap;     // [PARTS][G][HD]
    __shared__ float scratch[4];

    const int kvh = blockIdx.x, b = blockIdx.y, tid = threadIdx.x;
    const int slot = T0 + *step_p;
    const int kstart = kstart_p[b];
    const int pos = slot - kstart;
    const int64_t ld = (int64_t)(nh + 2 * nkv) * HD;
    const half_t *row = qkv + (int64_t)b * ld;
    half_t *kcb = kc + b * cache_sb + kvh * cache_sh;
    half_t *vcb = vc + b * cache_sb + kvh * cache_sh;

    // ---- rotary on the G query heads and the new key; append k, v ----
    for (int i = tid; i < (G + 1) * HALF; i += 256) {
        const int j = i / HALF, d = i % HALF;
        const half_t *src = j < G ? row + (int64_t)(kvh * G + j) * HD : row + (int64_t)(nh + kvh) * HD;
        const float c = cs[((int64_t)pos * HALF + d) * 2], sn = cs[((int64_t)pos * HALF + d) * 2 + 1];
        const float a = (float)src[d], bb = (float)src[d + HALF];
        const half_t lo = (half_t)(a * c - bb * sn), hi = (half_t)(bb * c + a * sn);
        if (j < G) {
            sq[j * HD + d] = (float)lo;
            sq[j * HD + d + HALF] = (float)hi;
        } else {
            kcb[(int64_t)slot * HD + d] = lo;
            kcb[(int64_t)slot * HD + d + HALF] = hi;
        }
    }
    for (int d = tid; d < HD; d += 256) vcb[(int64_t)slot * HD + d] = row[(int64_t)(nh + nkv + kvh) * HD + d];
    __syncthreads();

    // ---- scores: one key per thread ----
    const int nkeys = slot - kstart + 1;
    for (int j = tid; j < nkeys; j += 256) {
        const h8 *kr = reinterpret_cast<const h8 *>(kcb + (int64_t)(kstart + j) * HD);
        float acc[MAXG];
#pragma unroll
        for (int gi = 0; gi < MAXG; ++gi) acc[gi] = 0.f;
#pragma unroll
        for (int c = 0; c < DV; ++c) {
            const h8 kv = kr[c];
#pragma unroll
            for (int gi = 0; gi < MAXG; ++gi) {
                if (gi < G) {
#pragma unroll
                    for (int e = 0; e < 8; ++e) acc[gi] += (float)kv[e] * sq[gi * HD + c * 8 + e];
                }
            }
        }
#pragma unroll
        for (int gi = 0; gi < MAXG; ++gi)
            if (gi < G) sc[gi * ctx_cap + j] = acc[gi] * scale;
    }
    __syncthreads();

    // ---- softmax per head (fp32) ----
    float linv[MAXG];
    for (int gi = 0; gi < G; ++gi) {
        float mx = -INFINITY;
        for (int j = tid; j < nkeys; j += 256) mx = fmaxf(mx, sc[gi * ctx_cap + j]);
        mx = block_reduce(mx, true, scratch);
        float sum = 0.f;
        for (int j = tid; j < nkeys; j += 256) {
            const float e = __expf(sc[gi * ctx_cap + j] - mx);
            // P is rounded to fp16 before the PV product, as the prefill kernel and HF (softmax .to(q.dtype))
            sc[gi * ctx_cap + j] = (float)(half_t)e;
            sum += e;
        }
        sum = block_reduce(sum, false, scratch);
        linv[gi] = 1.0f / sum;
    }
    __syncthreads();

    // ---- O = P V : thread = (8-wide column slice, key partition) ----
    const int dv = tid % DV, part = tid / DV;
    float acc[MAXG][8];
#pragma unroll
    for (int gi = 0; gi < MAXG; ++gi)
#pragma unroll
        for (int e = 0; e < 8; ++e) acc[gi][e] = 0.f;
    for (int j = part; j < nkeys; j += PARTS) {
        const h8 vv = *reinterpret_cast<const h8 *>(vcb + (int64_t)(kstart + j) * HD + dv * 8);
#pragma unroll
        for (int gi = 0; gi < MAXG; ++gi) {
            if (gi < G) {
                const float pj = sc[gi * ctx_cap + j];
#pragma unroll
                for (int e = 0; e < 8; ++e) acc[gi][e] += pj * (float)vv[e];
            }
        }
    }
#pragma unroll
    for (int gi = 0; gi < MAXG; ++gi)
        if (gi < G) {
#pragma unroll
            for (int e = 0; e < 8; ++e) red[(part * G + gi) * HD + dv * 8 + e] = acc[gi][e];
        }
    __syncthreads();
    for (int i = tid; i < G * HD; i += 256) {
        const int gi = i / HD, d = i % HD;
        float s = 0.f;
        for (int pp = 0; pp < PARTS; ++pp) s += red[(pp * G + gi) * HD + d];
        // linv is per-thread identical (block_reduce broadcasts), index by gi at runtime:
        float li = linv[0];
#pragma unroll
        for (int q = 1; q < MAXG; ++q) li = (q == gi) ? linv[q] : li;
        out[(int64_t)b * nh * HD + (int64_t)(kvh * G + gi) * HD + d] = (half_t)(s * li);
    }
}

hipError_t launch_attn_decode(const half_t *qkv, const float *cs, const int32_t *kstart, const int32_t *step, int T0,
                              int B, int nh, int nkv, int hd, half_t *kc, half_t *vc, int64_t cache_sb,
                              int64_t cache_sh, int ctx_cap, float scale, half_t *out, hipStream_t s) {
    const int G = nh / nkv;
    if (G > MAXG || G * nkv != nh) return hipErrorInvalidValue;
    const int parts = 256 / (hd / 8);
    const size_t lds = ((size_t)MAXG * hd + (size_t)MAXG * ctx_cap + (size_t)parts * G * hd) * sizeof(float);
    if (lds > 150 * 1024) return hipErrorInvalidValue;
#define OPUS_AD(HDV)                                                                                               \
    {                                                                                                              \
        if (lds > 48 * 1024) {                                                                                     \
            hipError_t ea = hipFuncSetAttribute(reinterpret_cast<const void *>(&attn_decode_kernel<HDV>),         \
                                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);             \
            if (ea != hipSuccess) return ea;                                                                       \
        }                                                                                                          \
        hipLaunchKernelGGL((attn_decode_kernel<HDV>), dim3(nkv, B), dim3(256), lds, s, qkv, cs, kstart, step, T0, \
                           nh, nkv, kc, vc, cache_sb, cache_sh, ctx_cap, scale, out);                              \
    }
    switch (hd) {
        case 16: OPUS_AD(16) break;
        case 32: OPUS_AD(32) break;
        case 64: OPUS_AD(64) break;
        case 128: OPUS_AD(128) break;
        default: return hipErrorInvalidValue;
    }
#undef OPUS_AD
    return hipGetLastError();
}

}  // namespace opus
