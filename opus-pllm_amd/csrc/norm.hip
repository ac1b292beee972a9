// Row-wise normalisations (HBM-bound, one wave per row, float4 loads, wave64 shuffles).
//   layernorm : ESM-2 pre-LN / final LN (fp32 in -> fp16 and/or fp32 out)      rows E2, E3
//   rmsnorm   : Llama RMSNorm, fp32 variance (modeling_llama.py:62-67)           row D1
//   l2norm    : F.normalize(x, dim=-1) of CSTPBase.protein_forward               row P1
//   masked_mean: mean over residues 1..len-2 of the final representations        row E4
#include "common.h"

namespace opus {

constexpr int NV_MAX = 20;  // float4 per lane -> rows up to 64*4*20 = 5120 columns

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

// NV = float4 per lane: instantiated for 5 (D <= 1280: ESM2-650M), 10 (<= 2560: ESM2-3B), 16 (<= 4096) and 20 (<= 5120), so the
// row lives in exactly the registers it needs (occupancy) and no predicated-off iterations are issued
template <int MODE, int NV>  // MODE: 0 layernorm, 1 rmsnorm, 2 l2norm
__global__ __launch_bounds__(256) void rownorm_kernel(const float *__restrict__ x, const float *__restrict__ w,
                                                      const float *__restrict__ b, float eps, int64_t rows, int D,
                                                      half_t *__restrict__ out_h, float *__restrict__ out_f) {
    const int lane = threadIdx.x & 63;
    const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const float4 *xr = reinterpret_cast<const float4 *>(x + row * D);
    const int nvec = D >> 2;
    float4 v[NV];
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const int c = i * 64 + lane;
        if (c < nvec) {
            v[i] = xr[c];
            if (MODE == 0) s += (v[i].x + v[i].y) + (v[i].z + v[i].w);
            else s += (v[i].x * v[i].x + v[i].y * v[i].y) + (v[i].z * v[i].z + v[i].w * v[i].w);
        }
    }
    s = wave_sum(s);
    float mean = 0.f, rstd;
    if (MODE == 0) {
        mean = s / D;
        float q = 0.f;
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const int c = i * 64 + lane;
            if (c < nvec) {
                const float a = v[i].x - mean, bb = v[i].y - mean, cc = v[i].z - mean, d = v[i].w - mean;
                q += (a * a + bb * bb) + (cc * cc + d * d);
            }
        }
        q = wave_sum(q);
        rstd = rsqrtf(q / D + eps);
    } else if (MODE == 1) {
        rstd = rsqrtf(s / D + eps);
    } else {
        rstd = 1.0f / fmaxf(sqrtf(s), 1e-12f);   // F.normalize: x / max(||x||, eps)
    }
    const float4 *wr = reinterpret_cast<const float4 *>(w);
    const float4 *br = reinterpret_cast<const float4 *>(b);
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const int c = i * 64 + lane;
        if (c < nvec) {
            float4 y;
            y.x = (v[i].x - mean) * rstd; y.y = (v[i].y - mean) * rstd;
            y.z = (v[i].z - mean) * rstd; y.w = (v[i].w - mean) * rstd;
            if (MODE != 2 && w) { const float4 ww = wr[c]; y.x *= ww.x; y.y *= ww.y; y.z *= ww.z; y.w *= ww.w; }
            if (MODE == 0 && b) { const float4 bv = br[c]; y.x += bv.x; y.y += bv.y; y.z += bv.z; y.w += bv.w; }
            if (out_h) {
                h4 o = {(half_t)y.x, (half_t)y.y, (half_t)y.z, (half_t)y.w};
                reinterpret_cast<h4 *>(out_h + row * D)[c] = o;
            }
            if (out_f) reinterpret_cast<float4 *>(out_f + row * D)[c] = y;
        }
    }
}

template <int MODE>
static hipError_t launch_rownorm(const float *x, const float *w, const float *b, float eps, int64_t rows, int D, half_t *out_h,
                                 float *out_f, hipStream_t s) {
    if (D > NV_MAX * 256 || (D & 3)) return hipErrorInvalidValue;
    const dim3 grid(cdiv(rows, 4)), block(256);
    if (D <= 5 * 256) OPUS_LAUNCH(KC_NORM, (rownorm_kernel<MODE, 5>), grid, block, 0, s, x, w, b, eps, rows, D, out_h, out_f);
    else if (D <= 10 * 256) OPUS_LAUNCH(KC_NORM, (rownorm_kernel<MODE, 10>), grid, block, 0, s, x, w, b, eps, rows, D, out_h, out_f);
    else if (D <= 16 * 256) OPUS_LAUNCH(KC_NORM, (rownorm_kernel<MODE, 16>), grid, block, 0, s, x, w, b, eps, rows, D, out_h, out_f);
    else OPUS_LAUNCH(KC_NORM, (rownorm_kernel<MODE, 20>), grid, block, 0, s, x, w, b, eps, rows, D, out_h, out_f);
    return hipGetLastError();
}
hipError_t launch_layernorm(const float *x, const float *w, const float *b, float eps, int64_t rows, int D,
                            half_t *out_h, float *out_f, hipStream_t s) {
    return launch_rownorm<0>(x, w, b, eps, rows, D, out_h, out_f, s);
}
hipError_t launch_rmsnorm(const float *x, const float *w, float eps, int64_t rows, int D, half_t *out, hipStream_t s) {
    return launch_rownorm<1>(x, w, nullptr, eps, rows, D, out, nullptr, s);
}
hipError_t launch_l2norm(const float *x, int64_t rows, int D, half_t *out, hipStream_t s) {
    return launch_rownorm<2>(x, nullptr, nullptr, 0.f, rows, D, out, nullptr, s);
}

// (mu, rstd) of every row from the per-64-column (sum x, sum x^2) partials of the LayerNorm-producing GEMM epilogue
// (GemmParams::ln_part), summed in a fixed order.  var = E[x^2] - mu^2 in fp32 (clamped at 0): the residual streams this serves
// have |mu| of the order of the standard deviation or below, where the cancellation costs a few ulps of the variance.
__global__ __launch_bounds__(256) void ln_finalize_kernel(const float2 *__restrict__ part, int64_t rows, int nslab, float inv_d,
                                                          float eps, int rms, float2 *__restrict__ stat) {
    // 16 lanes per row: lane q takes slabs q, q + 16, ... (all requested at once), then a fixed-order 16-lane tree
    const int q = threadIdx.x & 15;
    int64_t m = (int64_t)blockIdx.x * 16 + (threadIdx.x >> 4);
    const bool live = m < rows;
    m = live ? m : rows - 1;
    const float2 *src = part + m * nslab;
    float s1 = 0.f, s2 = 0.f;
    for (int j0 = 0; j0 < nslab; j0 += 64) {
        float2 t[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int j = j0 + 16 * u + q;
            t[u] = src[j < nslab ? j : nslab - 1];
            if (j >= nslab) t[u] = make_float2(0.f, 0.f);
        }
        s1 += (t[0].x + t[1].x) + (t[2].x + t[3].x);
        s2 += (t[0].y + t[1].y) + (t[2].y + t[3].y);
    }
#pragma unroll
    for (int o = 1; o < 16; o <<= 1) { s1 += __shfl_xor(s1, o, 64); s2 += __shfl_xor(s2, o, 64); }
    if (!live || q) return;
    if (rms) {                                                        // RMSNorm: (0, rsqrt(mean x^2 + eps))
        stat[m] = make_float2(0.f, rsqrtf(s2 * inv_d + eps));
        return;
    }
    const float mu = s1 * inv_d;
    const float var = fmaxf(s2 * inv_d - mu * mu, 0.f);
    stat[m] = make_float2(mu, rsqrtf(var + eps));
}
hipError_t launch_ln_finalize(const float *part, int64_t rows, int nslab, int D, float eps, int rms, float *stat, hipStream_t s) {
    hipLaunchKernelGGL(ln_finalize_kernel, dim3(cdiv(rows, 16)), dim3(256), 0, s, reinterpret_cast<const float2 *>(part), rows, nslab,
                       1.0f / (float)D, eps, rms, reinterpret_cast<float2 *>(stat));
    return hipGetLastError();
}

// fp32 -> fp16 cast (the identity protein projector of opus_arch.py:70-80 hands the pooled fp32 embedding straight to the
// switch projector, whose autocast Linear rounds it to fp16)
__global__ __launch_bounds__(256) void f2h_kernel(const float *__restrict__ x, int64_t n4, half_t *__restrict__ out) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n4) return;
    const float4 v = reinterpret_cast<const float4 *>(x)[i];
    reinterpret_cast<h4 *>(out)[i] = h4{(half_t)v.x, (half_t)v.y, (half_t)v.z, (half_t)v.w};
}
hipError_t launch_f2h(const float *x, int64_t n, half_t *out, hipStream_t s) {
    if (n & 3) return hipErrorInvalidValue;
    hipLaunchKernelGGL(f2h_kernel, dim3(cdiv(n >> 2, 256)), dim3(256), 0, s, x, n >> 2, out);
    return hipGetLastError();
}

// out[b][d] = mean_{t=1}^{len_b-2} h[b][t][d]; fixed summation order (bitwise reproducible).
// (token-packed form, cu != nullptr: row b's tokens are the rows cu[b] .. cu[b + 1] - 1 of h)
__global__ __launch_bounds__(256) void masked_mean_kernel(const float *__restrict__ h, const int32_t *__restrict__ lens,
                                                          int T, int D, float *__restrict__ out, const int32_t *__restrict__ cu) {
    const int b = blockIdx.y;
    const int d = blockIdx.x * 256 + threadIdx.x;
    if (d >= D) return;
    const int n = (cu ? cu[b + 1] - cu[b] : lens[b]) - 2;
    const float *p = h + ((cu ? (int64_t)cu[b] : (int64_t)b * T) + 1) * D + d;
    // eight interleaved partial sums (t mod 8) keep eight loads in flight; combined in a fixed order
    float s8[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    int t = 0;
    for (; t + 8 <= n; t += 8) {
#pragma unroll
        for (int k = 0; k < 8; ++k) s8[k] += p[(int64_t)(t + k) * D];
    }
    for (int k = 0; t < n; ++t, ++k) s8[k] += p[(int64_t)t * D];
    const float s = ((s8[0] + s8[1]) + (s8[2] + s8[3])) + ((s8[4] + s8[5]) + (s8[6] + s8[7]));
    out[(int64_t)b * D + d] = s / (float)n;   // n == 0 -> NaN, as torch's mean over an empty slice
}

hipError_t launch_masked_mean(const float *h, const int32_t *lens, int B, int T, int D, float *out, hipStream_t s) {
    hipLaunchKernelGGL(masked_mean_kernel, dim3(cdiv(D, 256), B), dim3(256), 0, s, h, lens, T, D, out, nullptr);
    return hipGetLastError();
}
hipError_t launch_masked_mean_packed(const float *h, const int32_t *cu, int B, int D, float *out, hipStream_t s) {
    if (!cu) return hipErrorInvalidValue;
    hipLaunchKernelGGL(masked_mean_kernel, dim3(cdiv(D, 256), B), dim3(256), 0, s, h, nullptr, 0, D, out, cu);
    return hipGetLastError();
}

}  // namespace opus
