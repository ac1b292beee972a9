// Memory-bound glue kernels of the path (gather / rotary / cache append / splice / argmax / fill).
#include "common.h"

namespace opus {

// ------------------------------------------------------------------------------ ESM-2 embedding (E1)
// x[b,t,:] = emb[tok] * 0.88 / (1 - n_mask_b / n_nonpad_b), <mask> rows and <pad> rows zeroed
// (token-dropout rescale of fair_esm ESM2.forward; modeling_esm.py:252-268).
// Token-packed form (cu != nullptr): row b's tokens are tok[cu[b] .. cu[b + 1]) and its output rows x[cu[b] ..]; the kernel also
// writes the row -> position table pos[cu[b] + t] = t that the rotary of the packed encoder reads.
__global__ __launch_bounds__(256) void esm_embed_kernel(const int32_t *__restrict__ tok, const half_t *__restrict__ emb,
                                                        int T, int D, float *__restrict__ x, const int32_t *__restrict__ cu,
                                                        int32_t *__restrict__ pos) {
    __shared__ int s_cnt[2];
    const int b = blockIdx.y;
    int64_t r0 = (int64_t)b * T;
    if (cu) {
        r0 = cu[b];
        T = cu[b + 1] - cu[b];
        if ((int)blockIdx.x * 16 >= T) return;               // (uniform: before any barrier)
    }
    const int32_t *row = tok + r0;
    if (threadIdx.x < 2) s_cnt[threadIdx.x] = 0;
    __syncthreads();
    int nm = 0, nv = 0;
    for (int t = threadIdx.x; t < T; t += 256) {
        const int v = row[t];
        nm += (v == 32);
        nv += (v != 1);
    }
    atomicAdd(&s_cnt[0], nm);
    atomicAdd(&s_cnt[1], nv);
    __syncthreads();
    const float scale = (1.0f - 0.15f * 0.8f) / (1.0f - (float)s_cnt[0] / (float)s_cnt[1]);
    const int t0 = blockIdx.x * 16;
    const int nvec = D >> 3;
    for (int i = threadIdx.x; i < 16 * nvec; i += 256) {
        const int t = t0 + i / nvec, c = i % nvec;
        if (t >= T) break;
        const int v = row[t];
        float o[8];
        if (v == 1 || v == 32) {
#pragma unroll
            for (int j = 0; j < 8; ++j) o[j] = 0.f;
        } else {
            const h8 e = *reinterpret_cast<const h8 *>(emb + (int64_t)v * D + c * 8);
#pragma unroll
            for (int j = 0; j < 8; ++j) o[j] = (float)e[j] * scale;
        }
        float4 *dst = reinterpret_cast<float4 *>(x + (r0 + t) * D + c * 8);
        dst[0] = make_float4(o[0], o[1], o[2], o[3]);
        dst[1] = make_float4(o[4], o[5], o[6], o[7]);
        if (pos && c == 0) pos[r0 + t] = t;
    }
}

hipError_t launch_esm_embed(const int32_t *tok, const half_t *emb, int B, int T, int D, float *x, hipStream_t s) {
    hipLaunchKernelGGL(esm_embed_kernel, dim3(cdiv(T, 16), B), dim3(256), 0, s, tok, emb, T, D, x, nullptr, nullptr);
    return hipGetLastError();
}
hipError_t launch_esm_embed_packed(const int32_t *tok, const half_t *emb, const int32_t *cu, int B, int Tmax, int D, float *x,
                                   int32_t *pos, hipStream_t s) {
    if (!cu || !pos) return hipErrorInvalidValue;
    hipLaunchKernelGGL(esm_embed_kernel, dim3(cdiv(Tmax, 16), B), dim3(256), 0, s, tok, emb, Tmax, D, x, cu, pos);
    return hipGetLastError();
}

// ------------------------------------------------------------------------------ rotary helpers
// cs table: fp32 [pos][hd/2][2] = (cos, sin) of pos * theta^(-2i/hd).
// Half-rotation (rotate_half): out[d] = x[d] cos - x[d+half] sin ; out[d+half] = x[d+half] cos + x[d] sin.
__device__ __forceinline__ void rope8(half_t *base, int half, const float *cs, float scale) {
    h8 lo = *reinterpret_cast<h8 *>(base);
    h8 hi = *reinterpret_cast<h8 *>(base + half);
    h8 olo, ohi;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const float c = cs[2 * j], sn = cs[2 * j + 1];
        const float a = (float)lo[j] * scale, b = (float)hi[j] * scale;
        float rl, rh;
        rotate_pair(a, b, c, sn, rl, rh);
        olo[j] = (half_t)rl;
        ohi[j] = (half_t)rh;
    }
    *reinterpret_cast<h8 *>(base) = olo;
    *reinterpret_cast<h8 *>(base + half) = ohi;
}

// ESM-2 (E2): q <- rotary(q * hd^-0.5), k <- rotary(k), positions 0..T-1, in place on the fused
// [B*T, 3D] projection output (query scaled BEFORE the rotation, modeling_esm.py:374).
__global__ __launch_bounds__(256) void esm_rope_kernel(half_t *__restrict__ qkv, const float *__restrict__ cs, int T,
                                                       int heads, int hd, float qscale, int64_t total, const int32_t *__restrict__ pos) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= total) return;
    const int half = hd >> 1, vph = half >> 3;            // 8-wide vectors per half head
    const int v = (int)(i % vph);
    int64_t r = i / vph;
    const int h = (int)(r % heads); r /= heads;
    const int which = (int)(r & 1);                        // 0 = q, 1 = k
    const int64_t row = r >> 1;
    const int t = pos ? pos[row] : (int)(row % T);          // (token-packed batches: the row's position comes from the table)
    const int D = heads * hd;
    half_t *base = qkv + row * (3 * (int64_t)D) + which * D + h * hd + v * 8;
    rope8(base, half, cs + ((int64_t)t * half + v * 8) * 2, which == 0 ? qscale : 1.0f);
}

hipError_t launch_esm_rope(half_t *qkv, const float *cs, int B, int T, int heads, int hd, float qscale, hipStream_t s, const int32_t *pos) {
    if (hd & 15) return hipErrorInvalidValue;
    const int64_t total = (int64_t)B * T * 2 * heads * (hd >> 4);
    hipLaunchKernelGGL(esm_rope_kernel, dim3(cdiv(total, 256)), dim3(256), 0, s, qkv, cs, T, heads, hd, qscale, total, pos);
    return hipGetLastError();
}

// Llama prefill (D1, D4): rotary on q and k at position t - kstart[b] (first real token = 0), in place
// on the fused [B*T, (nh+2nkv)*hd] buffer, then K and V appended to the cache [b][kvh][slot t][hd].
__global__ __launch_bounds__(256) void dec_rope_cache_kernel(half_t *__restrict__ qkv, const float *__restrict__ cs,
                                                             const int32_t *__restrict__ kstart, int T, int nh, int nkv,
                                                             int hd, half_t *__restrict__ kc, half_t *__restrict__ vc,
                                                             int64_t cache_sb, int64_t cache_sh, int64_t total) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= total) return;
    const int half = hd >> 1, vph = half >> 3;
    const int v = (int)(i % vph);
    int64_t r = i / vph;
    const int hh = (int)(r % (nh + nkv)); r /= (nh + nkv);   // q heads then k heads
    const int64_t row = r;
    const int b = (int)(row / T), t = (int)(row % T);
    int pos = t - kstart[b];
    pos = pos < 0 ? 0 : pos;
    const int64_t ld = (int64_t)(nh + 2 * nkv) * hd;
    half_t *base = qkv + row * ld + (int64_t)hh * hd + v * 8;
    rope8(base, half, cs + ((int64_t)pos * half + v * 8) * 2, 1.0f);
    if (hh >= nh) {
        const int kh = hh - nh;
        half_t *kd = kc + b * cache_sb + kh * cache_sh + (int64_t)t * hd + v * 8;
        half_t *vd = vc + b * cache_sb + kh * cache_sh + (int64_t)t * hd + v * 8;
        const half_t *vs = base + (int64_t)nkv * hd;
        *reinterpret_cast<h8 *>(kd) = *reinterpret_cast<h8 *>(base);
        *reinterpret_cast<h8 *>(kd + half) = *reinterpret_cast<h8 *>(base + half);
        *reinterpret_cast<h8 *>(vd) = *reinterpret_cast<const h8 *>(vs);
        *reinterpret_cast<h8 *>(vd + half) = *reinterpret_cast<const h8 *>(vs + half);
    }
}

hipError_t launch_dec_rope_cache(half_t *qkv, const float *cs, const int32_t *kstart, int B, int T, int nh, int nkv,
                                 int hd, half_t *kc, half_t *vc, int64_t cache_sb, int64_t cache_sh, hipStream_t s) {
    if (hd & 15) return hipErrorInvalidValue;
    const int64_t total = (int64_t)B * T * (nh + nkv) * (hd >> 4);
    hipLaunchKernelGGL(dec_rope_cache_kernel, dim3(cdiv(total, 256)), dim3(256), 0, s, qkv, cs, kstart, T, nh, nkv, hd,
                       kc, vc, cache_sb, cache_sh, total);
    return hipGetLastError();
}

// ------------------------------------------------------------------------------ small copies
__global__ __launch_bounds__(256) void h2f_kernel(const half_t *__restrict__ in, float *__restrict__ out, int64_t n8) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n8) return;
    const h8 v = reinterpret_cast<const h8 *>(in)[i];
    float4 *o = reinterpret_cast<float4 *>(out) + 2 * i;
    o[0] = make_float4((float)v[0], (float)v[1], (float)v[2], (float)v[3]);
    o[1] = make_float4((float)v[4], (float)v[5], (float)v[6], (float)v[7]);
}
hipError_t launch_h2f(const half_t *in, float *out, int64_t n, hipStream_t s) {
    if (n & 7) return hipErrorInvalidValue;
    hipLaunchKernelGGL(h2f_kernel, dim3(cdiv(n >> 3, 256)), dim3(256), 0, s, in, out, n >> 3);
    return hipGetLastError();
}

// OPT / Galactica learned positions (transformers OPTLearnedPositionalEmbedding): x[b, tq, :] += pos[idx, :] with
// idx = cumsum(mask) * mask - 1 + 2, which for a left-padded row is (slot - kstart[b]) + 2 on real slots and 1 on pads.
// slot = t0 (+ *step when step != nullptr: the decode step reads it from the device so that a graph can replay) + tq.
__global__ __launch_bounds__(256) void add_pos_kernel(float *__restrict__ x, const half_t *__restrict__ pos,
                                                      const int32_t *__restrict__ kstart, const int32_t *__restrict__ step,
                                                      int t0, int Tq, int H, int max_idx) {
    const int row = blockIdx.x, b = row / Tq, tq = row % Tq;
    const int slot = (t0 >= 0 ? t0 : step[1]) + (step ? step[0] : 0) + tq;      // (t0 < 0: the prompt length from the device, d_step[1])
    int idx = slot >= kstart[b] ? slot - kstart[b] + 2 : 1;
    idx = idx > max_idx ? max_idx : idx;
    for (int c = threadIdx.x; c < (H >> 3); c += 256) {
        const h8 e = *reinterpret_cast<const h8 *>(pos + (int64_t)idx * H + c * 8);
        float4 *o = reinterpret_cast<float4 *>(x + (int64_t)row * H + c * 8);
        float4 a = o[0], d = o[1];
        a.x += (float)e[0]; a.y += (float)e[1]; a.z += (float)e[2]; a.w += (float)e[3];
        d.x += (float)e[4]; d.y += (float)e[5]; d.z += (float)e[6]; d.w += (float)e[7];
        o[0] = a;
        o[1] = d;
    }
}
hipError_t launch_add_pos(float *x, const half_t *pos, const int32_t *kstart, const int32_t *step, int t0, int B, int Tq,
                          int H, int max_idx, hipStream_t s) {
    if (H & 7) return hipErrorInvalidValue;
    hipLaunchKernelGGL(add_pos_kernel, dim3(B * Tq), dim3(256), 0, s, x, pos, kstart, step, t0, Tq, H, max_idx);
    return hipGetLastError();
}

// x[b,:] = fp32(emb[tok[b]]) : embedding of the token just generated (input of the next decode step).  With xh / ssq
// (H % 256 == 0) the row is also left as fp16 together with the sum of squares of each 256-column block: what the row-scale
// RMSNorm fusion of the first layer's QKV GEMM consumes (GemmParams::row_ssq).
__global__ __launch_bounds__(256) void embed_tokens_kernel(const int32_t *__restrict__ tok, const half_t *__restrict__ emb,
                                                           int H, int V, float *__restrict__ x, half_t *__restrict__ xh,
                                                           float *__restrict__ ssq, int xh_tiled, const float *__restrict__ cs,
                                                           const int32_t *__restrict__ kstart, const int32_t *__restrict__ step,
                                                           int T0, int half, float *__restrict__ cs_row, int *__restrict__ cnt, int ncnt) {
    const int b = blockIdx.x;
    // first kernel of every decode step: the ticket / flag words of the step's in-launch hand-offs start from zero whatever an
    // earlier launch left behind (cdna_hip_programming.md Guideline 16 "Re-initialise every call"; the last arriver's re-arm alone
    // never recovers from one bad launch).  Ordered before the step's GEMMs by the kernel boundary.
    if (cnt && b == 0)
        for (int i = threadIdx.x; i < ncnt; i += 256) cnt[i] = 0;
    if (cs) {   // rotary (cos, sin) row of this batch row's position in this step, for every attention launch of the step
        const int pos = (T0 >= 0 ? T0 : step[1]) + step[0] - kstart[b];     // (T0 < 0: the prompt length from the device, d_step[1])
        for (int d = threadIdx.x; d < half; d += 256)
            reinterpret_cast<float2 *>(cs_row)[(int64_t)b * half + d] = reinterpret_cast<const float2 *>(cs)[(int64_t)pos * half + d];
    }
    int id = tok[b];
    id = id < 0 ? 0 : (id >= V ? V - 1 : id);
    for (int c = threadIdx.x; c < (H >> 3); c += 256) {
        const h8 e = *reinterpret_cast<const h8 *>(emb + (int64_t)id * H + c * 8);
        float4 *o = reinterpret_cast<float4 *>(x + (int64_t)b * H + c * 8);
        o[0] = make_float4((float)e[0], (float)e[1], (float)e[2], (float)e[3]);
        o[1] = make_float4((float)e[4], (float)e[5], (float)e[6], (float)e[7]);
        if (xh) {
            *reinterpret_cast<h8 *>(xh + (xh_tiled ? tiled_off(b, c * 8, H) : (int64_t)b * H + c * 8)) = e;
            float q = 0.f;
#pragma unroll
            for (int j = 0; j < 8; ++j) q += (float)e[j] * (float)e[j];
            // 32 consecutive threads hold one 256-column block (c / 32 is uniform over each half wave)
#pragma unroll
            for (int o2 = 16; o2 > 0; o2 >>= 1) q += __shfl_xor(q, o2, 64);
            if ((threadIdx.x & 31) == 0) ssq[(int64_t)b * (H >> 8) + (c >> 5)] = q;
        }
    }
}
hipError_t launch_embed_tokens(const int32_t *tok, const half_t *emb, int B, int H, int V, float *x, half_t *xh, float *ssq,
                               int xh_tiled, const float *cs, const int32_t *kstart, const int32_t *step, int T0, int half,
                               float *cs_row, int *cnt, int ncnt, hipStream_t s) {
    if (xh && (H & 255)) return hipErrorInvalidValue;
    hipLaunchKernelGGL(embed_tokens_kernel, dim3(B), dim3(256), 0, s, tok, emb, H, V, x, xh, ssq, xh_tiled, cs, kstart, step, T0, half,
                       cs_row, cnt, ncnt);
    return hipGetLastError();
}

// x <- max(x, 0) in place on fp16 activations (row N4: the ReLU feed-forward of the OPT decoders other than Galactica,
// modeling_opt.py ACT2FN["relu"] between fc1 and fc2); 8 halves per thread
__global__ __launch_bounds__(256) void relu_h_kernel(half_t *__restrict__ x, int64_t n8) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n8) return;
    h8 v = reinterpret_cast<h8 *>(x)[i];
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] = v[e] > (half_t)0.f ? v[e] : (half_t)0.f;
    reinterpret_cast<h8 *>(x)[i] = v;
}
hipError_t launch_relu_h(half_t *x, int64_t n, hipStream_t s) {
    if (n & 7) return hipErrorInvalidValue;
    hipLaunchKernelGGL(relu_h_kernel, dim3(cdiv(n >> 3, 256)), dim3(256), 0, s, x, n >> 3);
    return hipGetLastError();
}

// out[b,:] = x[b, T-1, :]   (lm_head is applied to the last position only; row D2)
__global__ __launch_bounds__(256) void take_last_kernel(const float *__restrict__ x, int T, int H, float *__restrict__ out) {
    const int b = blockIdx.x;
    const float4 *src = reinterpret_cast<const float4 *>(x + ((int64_t)b * T + (T - 1)) * H);
    float4 *dst = reinterpret_cast<float4 *>(out + (int64_t)b * H);
    for (int c = threadIdx.x; c < (H >> 2); c += 256) dst[c] = src[c];
}
hipError_t launch_take_last(const float *x, int B, int T, int H, float *out, hipStream_t s) {
    hipLaunchKernelGGL(take_last_kernel, dim3(B), dim3(256), 0, s, x, T, H, out);
    return hipGetLastError();
}

// ------------------------------------------------------------------------------ synthetic fill
// Bit-identical twin of opus-pllm_amd/synth.py::hash_normal (splitmix64 -> Irwin-Hall(4x16 bit)).
// (splitmix64: common.h)
__device__ __forceinline__ float synth_value(uint64_t seed, uint64_t i, float scale, float mean, int has_mean) {
    const uint64_t z = splitmix64(i + seed);
    const int sum = (int)(z & 0xFFFF) + (int)((z >> 16) & 0xFFFF) + (int)((z >> 32) & 0xFFFF) + (int)(z >> 48);
    float v = __fmul_rn((float)(sum - 131070), scale);
    if (has_mean) v = __fadd_rn(v, mean);
    return (float)(half_t)v;                   // round-to-nearest-even to fp16
}
// tiled = 1: destination is the panel-tiled GEMM weight layout of gemm.hip (dst rows % 16 == 0,
// cols % 64 == 0).  fold: multiply column k by the synthetic vector element (fold_seed, k) - the
// RMSNorm weight folded into the projection that consumes the normalised activations.
__global__ __launch_bounds__(256) void fill_synth_kernel(void *__restrict__ dst, int dtype, int64_t rows, int64_t cols,
                                                         uint64_t seed, float scale, float mean, int has_mean,
                                                         int64_t rb, int64_t rs, int64_t ro, int tiled, int has_fold,
                                                         uint64_t fold_seed, float fold_scale, float fold_mean,
                                                         int fold_has_mean) {
    const int64_t n = rows * cols;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        float v = synth_value(seed, (uint64_t)i, scale, mean, has_mean);
        const int64_t r = i / cols, c = i % cols;
        if (has_fold) v = (float)(half_t)__fmul_rn(v, synth_value(fold_seed, (uint64_t)c, fold_scale, fold_mean, fold_has_mean));
        const int64_t dr = (r / rb) * rs + ro + (r % rb);
        int64_t off = dr * cols + c;
        if (tiled) {
            const int64_t kk = c & 63;
            off = ((dr >> 4) * (cols >> 6) + (c >> 6)) * 1024 + (kk >> 5) * 512 + ((((kk & 31) >> 3) << 4) + (dr & 15)) * 8 + (kk & 7);
        }
        if (dtype == 0) reinterpret_cast<half_t *>(dst)[off] = (half_t)v;
        else reinterpret_cast<float *>(dst)[off] = v;
    }
}
hipError_t launch_fill_synth(void *dst, int dtype, int64_t rows, int64_t cols, uint64_t seed, float std, float mean,
                             int64_t rb, int64_t rs, int64_t ro, int tiled, uint64_t fold_seed, float fold_std,
                             float fold_mean, hipStream_t s) {
    const float ih_std = 37837.2272372065f;    // sqrt(4 * (65536^2 - 1) / 12), rounded to fp32 as NumPy does
    const float scale = std / ih_std;          // one fp32 divide on the host, as synth.py
    const int has_fold = fold_std != 0.0f || fold_mean != 0.0f;
    const int64_t n = rows * cols;
    int grid = cdiv(n, 256);
    grid = grid > 8192 ? 8192 : grid;
    hipLaunchKernelGGL(fill_synth_kernel, dim3(grid), dim3(256), 0, s, dst, dtype, rows, cols, seed, scale, mean,
                       mean != 0.0f ? 1 : 0, rb, rs, ro, tiled, has_fold, fold_seed, fold_std / ih_std, fold_mean,
                       fold_mean != 0.0f ? 1 : 0);
    return hipGetLastError();
}

// row-major W[N][K] -> panel-tiled layout of gemm.hip (load-time, once per weight)
__global__ __launch_bounds__(256) void tile_weight_kernel(const half_t *__restrict__ src, half_t *__restrict__ dst,
                                                          int64_t N, int64_t K) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;       // one 16-B piece (8 consecutive k) per thread
    if (i >= N * (K >> 3)) return;
    const int64_t n = i / (K >> 3), k8 = i % (K >> 3);
    const int64_t k = k8 * 8, kk = k & 63;
    const int64_t off = ((n >> 4) * (K >> 6) + (k >> 6)) * 1024 + (kk >> 5) * 512 + ((((kk & 31) >> 3) << 4) + (n & 15)) * 8;
    *reinterpret_cast<h8 *>(dst + off) = *reinterpret_cast<const h8 *>(src + n * K + k);
}
hipError_t launch_tile_weight(const half_t *src, half_t *dst, int64_t N, int64_t K, hipStream_t s) {
    hipLaunchKernelGGL(tile_weight_kernel, dim3(cdiv(N * (K >> 3), 256)), dim3(256), 0, s, src, dst, N, K);
    return hipGetLastError();
}

// ------------------------------------------------------------------------------ LoRA merge (L1)
// W[o,i] <- fp16(W[o,i] + scale * sum_r B[o,r] A[r,i])   (fp32 accumulation, one rounding)
__global__ __launch_bounds__(256) void lora_merge_kernel(half_t *__restrict__ W, const half_t *__restrict__ A,
                                                         const half_t *__restrict__ Bm, float scale, int64_t out_f,
                                                         int64_t in_f, int r) {
    const int64_t vec = in_f >> 3;
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= out_f * vec) return;
    const int64_t o = i / vec, c = i % vec;
    float acc[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) acc[j] = 0.f;
    for (int k = 0; k < r; ++k) {
        const float bv = (float)Bm[o * r + k];
        const h8 a = *reinterpret_cast<const h8 *>(A + (int64_t)k * in_f + c * 8);
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[j] += bv * (float)a[j];
    }
    h8 *wp = reinterpret_cast<h8 *>(W + o * in_f + c * 8);
    h8 w = *wp;
#pragma unroll
    for (int j = 0; j < 8; ++j) w[j] = (half_t)((float)w[j] + scale * acc[j]);
    *wp = w;
}
hipError_t launch_lora_merge(half_t *W, const half_t *A, const half_t *B, float scale, int64_t out_f, int64_t in_f,
                             int r, hipStream_t s) {
    if (in_f & 7) return hipErrorInvalidValue;
    hipLaunchKernelGGL(lora_merge_kernel, dim3(cdiv(out_f * (in_f >> 3), 256)), dim3(256), 0, s, W, A, B, scale, out_f,
                       in_f, r);
    return hipGetLastError();
}

// ------------------------------------------------------------------------------ splice (S1-S3)
// plan[4b+0] = spliced length of row b (after optional truncation), [4b+1] = first protein block
// index, [4b+2] = #placeholders, [4b+3] = #valid ids.  plan[4B] = max length, plan[4B+1] = protein
// blocks consumed (a row without placeholder consumes one: opus_arch.py:196-203), plan[4B+2] = 1 if
// an id is outside [0,V) and != -200.
__global__ __launch_bounds__(1024) void splice_plan_kernel(const int64_t *__restrict__ ids, const uint8_t *__restrict__ mask, int B, int Tt,
                                                         int n_tok, int max_len, int V, int32_t *__restrict__ plan) {
    // 16 lanes per row count its valid ids / placeholders / out-of-range ids; the (short) scan over rows is serial
    __shared__ int s_bad;
    const int tid = threadIdx.x, sub = tid & 15;
    if (tid == 0) s_bad = 0;
    __syncthreads();
    for (int b0 = 0; b0 < B; b0 += 64) {
        const int b = b0 + (tid >> 4);
        int nv = 0, nph = 0, bad = 0;
        if (b < B) {
            for (int t = sub; t < Tt; t += 16) {
                if (mask && !mask[(int64_t)b * Tt + t]) continue;
                const int64_t id = ids[(int64_t)b * Tt + t];
                ++nv;
                if (id == -200) ++nph;
                else if (id < 0 || id >= V) bad = 1;
            }
        }
#pragma unroll
        for (int o = 8; o > 0; o >>= 1) {
            nv += __shfl_xor(nv, o, 64);
            nph += __shfl_xor(nph, o, 64);
            bad |= __shfl_xor(bad, o, 64);
        }
        if (b < B && sub == 0) {
            int len = nv - nph + nph * n_tok;
            if (max_len > 0 && len > max_len) len = max_len;
            plan[4 * b + 0] = len;
            plan[4 * b + 2] = nph;
            plan[4 * b + 3] = nv;
            if (bad) atomicOr(&s_bad, 1);
        }
    }
    __syncthreads();                      // plan[] rows written by this workgroup are visible to thread 0 (same CU, after the barrier)
    if (tid == 0) {
        int seq = 0, tmax = 0;
        for (int b = 0; b < B; ++b) {
            plan[4 * b + 1] = seq;
            const int nph = plan[4 * b + 2], len = plan[4 * b];
            seq += nph > 0 ? nph : 1;
            tmax = len > tmax ? len : tmax;
        }
        plan[4 * B] = tmax;
        plan[4 * B + 1] = seq;
        plan[4 * B + 2] = s_bad;
    }
}
__global__ __launch_bounds__(256) void splice_fill_kernel(const int64_t *__restrict__ ids, const uint8_t *__restrict__ mask,
                                                          int Tt, const half_t *__restrict__ prot, int n_tok, int H, int V,
                                                          const half_t *__restrict__ emb, const int32_t *__restrict__ plan,
                                                          int Tout, int left_pad, half_t *__restrict__ out,
                                                          uint8_t *__restrict__ mask_out, int32_t *__restrict__ pos_out) {
    __shared__ int64_t s_src;      // >= 0: embedding row ; < 0: -(1 + protein_row)
    const int b = blockIdx.y, to = blockIdx.x;
    const int len = plan[4 * b];
    const int j = left_pad ? to - (Tout - len) : to;
    const bool real = j >= 0 && j < len;
    if (threadIdx.x == 0) {
        int64_t src = 0;
        if (real) {
            int c = 0, seq = plan[4 * b + 1];
            for (int t = 0; t < Tt; ++t) {
                if (mask && !mask[(int64_t)b * Tt + t]) continue;
                const int64_t id = ids[(int64_t)b * Tt + t];
                if (id == -200) {
                    if (j < c + n_tok) { src = -(1 + ((int64_t)seq * n_tok + (j - c))); break; }
                    c += n_tok;
                    ++seq;
                } else {
                    if (j == c) { src = id < 0 ? 0 : (id >= V ? V - 1 : id); break; }
                    ++c;
                }
            }
        }
        s_src = src;
        mask_out[(int64_t)b * Tout + to] = real ? 1 : 0;
        pos_out[(int64_t)b * Tout + to] = real ? j : 0;
    }
    __syncthreads();
    const int64_t src = s_src;
    h8 *dst = reinterpret_cast<h8 *>(out + ((int64_t)b * Tout + to) * H);
    const h8 *sp = real ? reinterpret_cast<const h8 *>(src >= 0 ? emb + src * H : prot + (-(src + 1)) * H) : nullptr;
    const h8 zero = {0, 0, 0, 0, 0, 0, 0, 0};
    for (int c = threadIdx.x; c < (H >> 3); c += 256) dst[c] = real ? sp[c] : zero;
}

hipError_t launch_splice_plan(const int64_t *ids, const uint8_t *mask, int B, int Tt, int n_tok, int max_len, int V,
                                int32_t *plan, hipStream_t s) {
    hipLaunchKernelGGL(splice_plan_kernel, dim3(1), dim3(1024), 0, s, ids, mask, B, Tt, n_tok, max_len, V, plan);
    return hipGetLastError();
}
hipError_t launch_splice_fill(const int64_t *ids, const uint8_t *mask, int B, int Tt, const half_t *prot, int n_tok,
                              int H, int V, const half_t *emb, const int32_t *plan, int Tout, int left_pad,
                              half_t *out, uint8_t *mask_out, int32_t *pos_out, hipStream_t s) {
    hipLaunchKernelGGL(splice_fill_kernel, dim3(Tout, B), dim3(256), 0, s, ids, mask, Tt, prot, n_tok, H, V, emb, plan,
                       Tout, left_pad, out, mask_out, pos_out);
    return hipGetLastError();
}

// ------------------------------------------------------------------------------ greedy step (G1)
// Stage 1: APART partial (max, lowest index) per row, float4 loads, many workgroups per row.
// (APART = 64 parts per logits row: common.h)
__global__ __launch_bounds__(256) void argmax_partial_kernel(const float *__restrict__ logits, int V,
                                                             float *__restrict__ pval, int32_t *__restrict__ pidx) {
    __shared__ float s_v[256];
    __shared__ int s_i[256];
    const int b = blockIdx.y, part = blockIdx.x;
    const float *row = logits + (int64_t)b * V;
    const int per = ((V + APART - 1) / APART + 3) & ~3;
    const int lo = part * per, hi = (lo + per < V ? lo + per : V);
    float bv = -INFINITY;
    int bi = 0x7fffffff;
    if ((((uintptr_t)row) & 15) == 0) {
        for (int i = lo + threadIdx.x * 4; i + 3 < hi; i += 1024) {
            const float4 v = *reinterpret_cast<const float4 *>(row + i);
            if (v.x > bv) { bv = v.x; bi = i; }
            if (v.y > bv) { bv = v.y; bi = i + 1; }
            if (v.z > bv) { bv = v.z; bi = i + 2; }
            if (v.w > bv) { bv = v.w; bi = i + 3; }
        }
        for (int i = lo + ((hi - lo) & ~3) + threadIdx.x; i < hi; i += 256) {
            const float v = row[i];
            if (v > bv || (v == bv && i < bi)) { bv = v; bi = i; }
        }
    } else {
        for (int i = lo + threadIdx.x; i < hi; i += 256) {
            const float v = row[i];
            if (v > bv || (v == bv && i < bi)) { bv = v; bi = i; }
        }
    }
    s_v[threadIdx.x] = bv;
    s_i[threadIdx.x] = bi;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if (threadIdx.x < o) {
            const float v = s_v[threadIdx.x + o];
            const int i = s_i[threadIdx.x + o];
            if (v > s_v[threadIdx.x] || (v == s_v[threadIdx.x] && i < s_i[threadIdx.x])) {
                s_v[threadIdx.x] = v;
                s_i[threadIdx.x] = i;
            }
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        pval[b * APART + part] = s_v[0];
        pidx[b * APART + part] = s_i[0];
    }
}
hipError_t launch_argmax_partial(const float *logits, int B, int V, float *pval, int32_t *pidx, hipStream_t s) {
    hipLaunchKernelGGL(argmax_partial_kernel, dim3(APART, B), dim3(256), 0, s, logits, V, pval, pidx);
    return hipGetLastError();
}

// Stage 2: one wave per row combines the partials (lowest index wins ties, as torch.argmax), then the
// GenerationMixin bookkeeping: finished rows emit pad_id; a row finishes when it emits an EOS id.
__global__ __launch_bounds__(64) void argmax_step_kernel(const float *__restrict__ pval, const int32_t *__restrict__ pidx,
                                                         const int32_t *__restrict__ chosen,
                                                         const int32_t *__restrict__ eos, int n_eos, int pad_id,
                                                         int32_t *__restrict__ finished, int32_t *__restrict__ out_ids,
                                                         int max_new, const int32_t *__restrict__ step,
                                                         int32_t *__restrict__ next_tok, int32_t *__restrict__ n_unf,
                                                         const int32_t *__restrict__ stop, int n_stop) {
    const int b = blockIdx.x, lane = threadIdx.x;
    float bv = pval[b * APART + lane];
    int bi = pidx[b * APART + lane];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const float v = __shfl_xor(bv, o, 64);
        const int i = __shfl_xor(bi, o, 64);
        if (v > bv || (v == bv && i < bi)) { bv = v; bi = i; }
    }
    if (chosen) bi = chosen[b];               // sampling: the token was drawn by sample_select_kernel
    if (lane == 0) {
        const int st = *step;
        int fin = finished[b];
        int tok = fin ? pad_id : (bi == 0x7fffffff ? 0 : bi);
        if (st < max_new) out_ids[(int64_t)b * max_new + st] = tok;
        if (!fin)
            for (int e = 0; e < n_eos; ++e) fin |= (tok == eos[e]);
        // opt-in stop sequence (the ids of "###", which the reference cuts at after decoding: run_opus_ddp.py:19-27): a row
        // whose last n_stop new ids spell it is finished - what it would emit afterwards is cut from the text anyway
        if (!fin && n_stop > 0 && st + 1 >= n_stop && st < max_new) {
            bool hit = true;
            for (int k = 0; k < n_stop; ++k) hit = hit && out_ids[(int64_t)b * max_new + st - n_stop + 1 + k] == stop[k];
            fin |= hit ? 1 : 0;
        }
        finished[b] = fin;
        next_tok[b] = tok;
        if (!fin && st < max_new) atomicAdd(&n_unf[st], 1);
    }
}
hipError_t launch_argmax_step(const float *pval, const int32_t *pidx, const int32_t *chosen, int B, const int32_t *eos,
                              int n_eos, int pad_id, int32_t *finished, int32_t *out_ids, int max_new, const int32_t *step,
                              int32_t *next_tok, int32_t *n_unfinished, const int32_t *stop, int n_stop, hipStream_t s) {
    hipLaunchKernelGGL(argmax_step_kernel, dim3(B), dim3(64), 0, s, pval, pidx, chosen, eos, n_eos, pad_id, finished, out_ids,
                       max_new, step, next_tok, n_unfinished, stop, n_stop);
    return hipGetLastError();
}

// ------------------------------------------------------------------------------ sampling head (N1)
// HF sampling as the reference drives it (run_opus_ddp.py:126-128: temperature, top_p; transformers
// TemperatureLogitsWarper -> TopPLogitsWarper -> softmax -> multinomial), one workgroup per row:
//   p_i = exp(l_i / T - max);  keep token i iff  sum_{p_j <= p_i} p_j  >  (1 - top_p) * Z   (the ascending
//   cumulative sum of TopPLogitsWarper; tokens of exactly equal probability are kept or dropped together);
//   draw u from a counter-based generator (seed, row, step) and invert the CDF of the kept set in index order.
// The nucleus threshold is found by bisection on the probability value (40 halvings: narrower than fp32
// spacing), every pass a fixed-order block reduction, so a (seed, row, step) triple always gives the same token.
// top_k > 0 (transformers 4.46.3, the reference's pin, defaults GenerationConfig.top_k to 50; TopKLogitsWarper runs between
// the temperature and the nucleus): only the k most probable tokens - and everything tied with the k-th - enter the nucleus
// computation, whose mass Z is then theirs alone.  The k-th value is found by the same kind of bisection (a count instead of a sum).
__device__ __forceinline__ float block_sum256(float v, float *scratch) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) scratch[threadIdx.x >> 6] = v;
    __syncthreads();
    return (scratch[0] + scratch[1]) + (scratch[2] + scratch[3]);
}
__device__ __forceinline__ float block_max256(float v, float *scratch) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    __syncthreads();
    if ((threadIdx.x & 63) == 0) scratch[threadIdx.x >> 6] = v;
    __syncthreads();
    return fmaxf(fmaxf(scratch[0], scratch[1]), fmaxf(scratch[2], scratch[3]));
}

// inclusive prefix sums over the 256 threads of a workgroup: wave scans by shuffles + the four wave totals through LDS (two
// barriers instead of a 256-step serial walk by one thread).  Fixed order: reproducible; non-decreasing for non-negative inputs.
template <class T>
__device__ __forceinline__ T block_incl_scan256(T v, T *sw) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    T x = v;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const T y = __shfl_up(x, o, 64);
        if (lane >= o) x += y;
    }
    __syncthreads();
    if (lane == 63) sw[wave] = x;
    __syncthreads();
    T base = 0;
    for (int w = 0; w < wave; ++w) base += sw[w];
    return base + x;
}

// Stage 1 (APART workgroups per row, after argmax_partial has left the per-part maxima): p_i = exp(l_i/T - max),
// per-part sum, and a COMPACTED candidate list per part in index order.  A token with p_i <= (1 - top_p) / V can
// never be kept (Z >= 1 because the arg-max token has p = 1, so the ascending cumulative mass up to it is at most
// V * p_i <= (1 - top_p) * Z): only the others - a handful at the reference's temperature 0.1 - go to stage 2.
__global__ __launch_bounds__(256) void sample_stage1_kernel(const float *__restrict__ logits, int V, float inv_temp, float top_p,
                                                            const float *__restrict__ pmax, float *__restrict__ cand_p,
                                                            int32_t *__restrict__ cand_i, int32_t *__restrict__ cand_n,
                                                            float *__restrict__ zpart, float *__restrict__ spart) {
    __shared__ float scratch[4];
    __shared__ int s_cnt[8];
    const int b = blockIdx.y, part = blockIdx.x, tid = threadIdx.x;
    const float *row = logits + (int64_t)b * V;
    float gmax = pmax[b * APART];
    for (int k = 1; k < APART; ++k) gmax = fmaxf(gmax, pmax[b * APART + k]);
    gmax *= inv_temp;
    const int per = ((V + APART - 1) / APART + 3) & ~3;          // same partition as argmax_partial
    const int lo = part * per, hi = (lo + per < V ? lo + per : V);
    const int tper = (per + 255) / 256;                           // contiguous elements per thread
    const int i0 = lo + tid * tper, i1 = (i0 + tper < hi ? i0 + tper : hi);
    const float thr = (1.0f - top_p) / (float)V;
    float z = 0.f, small = 0.f;
    int cnt = 0;
    // a thread's share (8 values at V = 128 256, 10 at Qwen2's 152 064) stays in registers: ONE pass over the logits and one
    // exponential per value (round 5; rounds 1-4 read and exponentiated every value twice, once to count and once to compact)
    constexpr int TR = 16;
    const bool in_regs = tper <= TR;
    float pr[TR];
    if (in_regs) {
#pragma unroll
        for (int u = 0; u < TR; ++u) {
            const int i = i0 + u;
            pr[u] = (u < tper && i < i1) ? __expf(row[i] * inv_temp - gmax) : 0.f;     // (same expression as the two-pass form)
        }
        // (measured and not kept: two float4 loads per thread instead of eight dwords - 27.2 vs 27.1 us: the loads are not what
        //  this kernel's time is made of)
#pragma unroll
        for (int u = 0; u < TR; ++u) {              // (index order, as the loop below: the same sums)
            if (u < tper && i0 + u < i1) {
                z += pr[u];
                if (pr[u] > thr) ++cnt; else small += pr[u];
            }
        }
    } else {
        for (int i = i0; i < i1; ++i) {
            const float p = __expf(row[i] * inv_temp - gmax);
            z += p;
            if (p > thr) ++cnt; else small += p;
        }
    }
    const int incl = block_incl_scan256<int>(cnt, s_cnt);        // (round 5: wave scans instead of a 256-step serial walk by thread 0: 30.4 -> 27.1 us)
    if (tid == 255) s_cnt[4] = incl;
    int w = incl - cnt;
    float *cp = cand_p + ((int64_t)b * APART + part) * per;
    int32_t *ci = cand_i + ((int64_t)b * APART + part) * per;
    if (in_regs) {
#pragma unroll
        for (int u = 0; u < TR; ++u)
            if (u < tper && i0 + u < i1 && pr[u] > thr) { cp[w] = pr[u]; ci[w] = i0 + u; ++w; }
    } else {
        for (int i = i0; i < i1; ++i) {
            const float p = __expf(row[i] * inv_temp - gmax);
            if (p > thr) { cp[w] = p; ci[w] = i; ++w; }
        }
    }
    z = block_sum256(z, scratch);
    small = block_sum256(small, scratch);
    if (tid == 0) {                                               // (behind the barriers of the block sums above)
        cand_n[b * APART + part] = s_cnt[4];
        zpart[b * APART + part] = z;
        spart[b * APART + part] = small;
    }
}

// Stage 2 (one workgroup per row): nucleus threshold by bisection over the candidates, then the draw.
constexpr int SAMPLE_LDS_CAP = 6144;
// thr_out != nullptr: no draw - the row's keep threshold (a token is kept iff its p = exp(l / T - max / T) exceeds it) is written
// there instead (beam-sample: beam.hip draws M continuations per batch row from the K filtered rows jointly).
__global__ __launch_bounds__(256) void sample_stage2_kernel(int V, float top_p, int top_k, const uint64_t *__restrict__ seed_p,
                                                            const int32_t *__restrict__ step, const float *__restrict__ cand_p,
                                                            const int32_t *__restrict__ cand_i, const int32_t *__restrict__ cand_n,
                                                            const float *__restrict__ zpart, const float *__restrict__ spart,
                                                            int32_t *__restrict__ chosen, float *__restrict__ thr_out) {
    __shared__ float scratch[4];
    __shared__ float s_pref[257];
    __shared__ float s_wtot[4];
    __shared__ int s_off[APART + 1];
    __shared__ float s_p[SAMPLE_LDS_CAP];
    const int b = blockIdx.x, tid = threadIdx.x;
    const int per = ((V + APART - 1) / APART + 3) & ~3;
    if (tid < 64) {                                   // (APART = 64 parts: one wave scans their candidate counts)
        static_assert(APART == 64, "one wave scans the parts");
        int x = cand_n[b * APART + tid];
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const int y = __shfl_up(x, o, 64);
            if (tid >= o) x += y;
        }
        s_off[tid + 1] = x;
        if (tid == 0) s_off[0] = 0;
    }
    __syncthreads();
    const int nc = s_off[APART];
    float Z = 0.f, S0 = 0.f;
    for (int k = 0; k < APART; ++k) { Z += zpart[b * APART + k]; S0 += spart[b * APART + k]; }   // fixed order
    // candidate c (global order = index order) lives at part k, slot c - s_off[k]
    auto cand = [&](int c, int &idx) -> float {
        int k = 0;
        while (c >= s_off[k + 1]) ++k;
        const int64_t o = ((int64_t)b * APART + k) * per + (c - s_off[k]);
        idx = cand_i[o];
        return cand_p[o];
    };
    const bool in_lds = nc <= SAMPLE_LDS_CAP;
    int dummy_id;
    // kept = candidates with p > lo; CDF inversion in index order over contiguous chunks of the candidate list
    auto draw = [&](float lo) {
        const int cper = (nc + 255) / 256;
        const int c0 = tid * cper, c1 = (c0 + cper < nc ? c0 + cper : nc);
        float mine = 0.f;
        for (int c = c0; c < c1; ++c) {
            const float p = in_lds ? s_p[c] : cand(c, dummy_id);
            mine += p > lo ? p : 0.f;
        }
        const float incl = block_incl_scan256<float>(mine, s_wtot);
        s_pref[tid + 1] = incl;                      // (the intervals [s_pref[t], s_pref[t + 1]) partition [0, total) exactly)
        if (tid == 0) s_pref[0] = 0.f;
        __syncthreads();
        const float total = s_pref[256];
        const uint64_t h = splitmix64(*seed_p ^ (0x9E3779B97F4A7C15ull * (uint64_t)(b + 1)) ^ ((uint64_t)(*step + 1) << 32));
        const float target = (float)(h >> 40) * (1.0f / 16777216.0f) * total;          // u in [0,1)
        if (s_pref[tid] <= target && target < s_pref[tid + 1]) {
            float run = s_pref[tid];
            int pick = -1, last_kept = -1;
            for (int c = c0; c < c1; ++c) {
                int id;
                const float p = cand(c, id);
                if (p > lo) {
                    last_kept = id;
                    run += p;
                    if (target < run) { pick = id; break; }
                }
            }
            chosen[b] = pick >= 0 ? pick : last_kept;
        }
    };
    if (in_lds)
        for (int c = tid; c < nc; c += 256) { int id; s_p[c] = cand(c, id); }
    __syncthreads();
    auto pval = [&](int c) -> float { int id; return in_lds ? s_p[c] : cand(c, id); };
    // top-k: lo_k < (k-th largest p) <= hi_k, i.e. count(p > lo_k) >= k > count(p > hi_k); the tokens that stay are p > lo_k.
    // Every token that can survive the nucleus is a stage-1 candidate (see there), so with fewer than k candidates all of them
    // stay; the other members of the top-k set then carry < k (1 - top_p) / V of the mass and are left out of Z.
    // Short candidate lists (a handful of tokens at the reference's temperature 0.1; up to EXACT_CAP): both thresholds EXACTLY, by
    // rank - one O(nc^2 / 256) pass instead of two bisections of 40 block reductions each (round 5: 31.6 -> ~8 us per step at
    // batch 64).  The same kept sets: a token survives top-k iff fewer than k tokens are more probable (ties with the k-th stay),
    // and the nucleus iff the ascending cumulative mass up to and including its ties exceeds (1 - top_p) Z.
    constexpr int EXACT_CAP = 1024;
    if (in_lds && nc <= EXACT_CAP) {
        float lok = 0.f;
        const bool use_k = top_k > 0 && top_k < V && nc > top_k;
        float mine[EXACT_CAP / 256];
#pragma unroll
        for (int q = 0; q < EXACT_CAP / 256; ++q) mine[q] = tid + 256 * q < nc ? s_p[tid + 256 * q] : -1.f;
        if (use_k) {
            float dropped = 0.f;                        // largest probability that top-k drops
#pragma unroll
            for (int q = 0; q < EXACT_CAP / 256; ++q) {
                if (mine[q] < 0.f) continue;
                int above = 0;
                for (int j = 0; j < nc; ++j) above += s_p[j] > mine[q] ? 1 : 0;
                if (above >= top_k) dropped = fmaxf(dropped, mine[q]);
            }
            lok = block_max256(dropped, scratch);
        }
        if (top_k > 0 && top_k < V) {                   // (as below: with top-k the nucleus mass is the survivors' alone)
            float zz = 0.f;
            for (int c = tid; c < nc; c += 256) zz += s_p[c] > lok ? s_p[c] : 0.f;
            Z = block_sum256(zz, scratch);
            S0 = 0.f;
        }
        const float cut_e = (1.0f - top_p) * Z;
        float dropped = 0.f;                            // largest surviving-top-k probability that the nucleus drops
#pragma unroll
        for (int q = 0; q < EXACT_CAP / 256; ++q) {
            if (mine[q] < 0.f || !(mine[q] > lok)) continue;
            float sle = S0;
            for (int j = 0; j < nc; ++j) sle += (s_p[j] <= mine[q] && s_p[j] > lok) ? s_p[j] : 0.f;      // (fixed order)
            if (!(sle > cut_e)) dropped = fmaxf(dropped, mine[q]);
        }
        float lo_e = fmaxf(block_max256(dropped, scratch), lok);
        // every token that is NOT a stage-1 candidate (p <= (1 - top_p) / V) is dropped too: a consumer that tests all tokens
        // against this threshold (beam-sample) must not take them for kept when no candidate was dropped
        lo_e = fmaxf(lo_e, (1.0f - top_p) / (float)V);
        if (thr_out) {
            if (tid == 0) thr_out[b] = lo_e;
            return;
        }
        draw(lo_e);
        return;
    }
    float lo_k = 0.f;
    if (top_k > 0 && top_k < V) {
        if (nc > top_k) {
            float hi_k = 1.0f;
            for (int it = 0; it < 40; ++it) {
                const float mid = 0.5f * (lo_k + hi_k);
                float n = 0.f;
                for (int c = tid; c < nc; c += 256) n += pval(c) > mid ? 1.f : 0.f;
                n = block_sum256(n, scratch);                         // (counts <= 2^24: exact in fp32)
                if (n >= (float)top_k) lo_k = mid; else hi_k = mid;
            }
        }
        float z = 0.f;
        for (int c = tid; c < nc; c += 256) {
            const float p = pval(c);
            z += p > lo_k ? p : 0.f;
        }
        Z = block_sum256(z, scratch);
        S0 = 0.f;
    }
    const float cut = (1.0f - top_p) * Z;
    // bisection: S_le(x) = S0 + sum of the (top-k) candidates <= x ; invariant S_le(lo) <= cut < S_le(hi)
    float lo = 0.f, hi = 1.0f;
    for (int it = 0; it < 40; ++it) {
        const float mid = 0.5f * (lo + hi);
        float s = 0.f;
        for (int c = tid; c < nc; c += 256) {
            const float p = pval(c);
            s += (p <= mid && p > lo_k) ? p : 0.f;
        }
        s = S0 + block_sum256(s, scratch);
        if (s > cut) hi = mid; else lo = mid;
    }
    lo = fmaxf(lo, lo_k);
    if (thr_out) {
        if (tid == 0) thr_out[b] = lo;
        return;
    }
    draw(lo);
}

hipError_t launch_sample_select(const float *logits, int B, int V, float temperature, float top_p, int top_k, const uint64_t *seed,
                                const int32_t *step, float *pmax, int32_t *pidx, float *cand_p, int32_t *cand_i,
                                int32_t *cand_n, float *zpart, float *spart, int32_t *chosen, float *thr_out, hipStream_t s) {
    hipLaunchKernelGGL(argmax_partial_kernel, dim3(APART, B), dim3(256), 0, s, logits, V, pmax, pidx);
    hipLaunchKernelGGL(sample_stage1_kernel, dim3(APART, B), dim3(256), 0, s, logits, V, 1.0f / temperature, top_p, pmax, cand_p,
                       cand_i, cand_n, zpart, spart);
    hipLaunchKernelGGL(sample_stage2_kernel, dim3(B), dim3(256), 0, s, V, top_p, top_k, seed, step, cand_p, cand_i, cand_n, zpart,
                       spart, chosen, thr_out);
    return hipGetLastError();
}

__global__ void step_advance_kernel(int32_t *step) {
    if (threadIdx.x == 0 && blockIdx.x == 0) *step += 1;
}
// A few host integers to device memory THROUGH THE KERNEL ARGUMENTS (<= 256 per launch): no host buffer has to outlive the call,
// nothing synchronises, and the launch is capturable (the packed encoder's row offsets: B + 1 values per call).
struct I32Chunk { int32_t v[256]; };
__global__ __launch_bounds__(256) void upload_i32_kernel(I32Chunk c, int n, int32_t *__restrict__ dst) {
    if ((int)threadIdx.x < n) dst[threadIdx.x] = c.v[threadIdx.x];
}
hipError_t launch_upload_i32(const int32_t *h, int n, int32_t *dst, hipStream_t s) {
    for (int i0 = 0; i0 < n; i0 += 256) {
        I32Chunk c;
        const int m = n - i0 < 256 ? n - i0 : 256;
        for (int i = 0; i < 256; ++i) c.v[i] = i < m ? h[i0 + i] : 0;
        hipLaunchKernelGGL(upload_i32_kernel, dim3(1), dim3(256), 0, s, c, m, dst + i0);
    }
    return hipGetLastError();
}

hipError_t launch_step_advance(int32_t *step, hipStream_t s) {
    hipLaunchKernelGGL(step_advance_kernel, dim3(1), dim3(64), 0, s, step);
    return hipGetLastError();
}

// kstart[b] = index of the first 1 of a left-padded mask row (T if none)
__global__ void mask_to_kstart_kernel(const uint8_t *__restrict__ mask, int B, int T, int32_t *__restrict__ kstart) {
    const int b = blockIdx.x * 64 + threadIdx.x;
    if (b >= B) return;
    int k = T;
    for (int t = 0; t < T; ++t)
        if (mask[(int64_t)b * T + t]) { k = t; break; }
    kstart[b] = k;
}
hipError_t launch_mask_to_kstart(const uint8_t *mask, int B, int T, int32_t *kstart, hipStream_t s) {
    hipLaunchKernelGGL(mask_to_kstart_kernel, dim3(cdiv(B, 64)), dim3(64), 0, s, mask, B, T, kstart);
    return hipGetLastError();
}

}  // namespace opus
