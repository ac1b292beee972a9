// GEMM kernels for gfx950: C = epi(A W^T + bias) (+ residual), fp16 operands, fp32 accumulation.
//
// Weight layout in HBM ("panel-tiled", built once at load time by weights.py / opus_fill_synth):
//   W[N][K] is cut into blocks of 16 rows x 64 k; block (p = n/16, c = k/64) is the 2 KB at
//   ((p * K/64) + c) * 1024 halfs and is stored in MFMA B-fragment order
//       [s = 0..1][lane = g*16 + li][8 halfs]   with   n = 16p + li,  k = 64c + 32s + 8g + e
//   so one wave-wide 16-B load fetches 1 KB of contiguous HBM and IS the B operand of one
//   mfma_f32_16x16x32_f16 (no LDS round trip, no shuffles), and a 16-row panel is one contiguous
//   K*32-byte stream.  Measured on MI355X (tools/bench_skinny.hip): 6.3-6.8 TB/s vs 4.5-5.3 TB/s for
//   the same kernel reading row-major weights.
//
//  * gemm_skinny (M <= 64): weight-streaming kernel for decode / projectors.  HBM-bound: every weight
//    byte is loaded exactly once, straight to registers (non-temporal), K is split over the waves of a
//    workgroup and combined through LDS.  Optional fused RMSNorm prologue: A is the fp32 residual
//    stream, sum(h^2) is accumulated from the very loads that feed the MFMA and 1/rms is applied in the
//    epilogue (the norm weight is folded into W at load time), which removes two launches per layer.
//  * gemm_tile (M > 64): 128x128x64 LDS-tiled MFMA kernel (4 waves x 64x64), register-staged double
//    buffering; A image XOR-swizzled, B image kept in fragment order (linear ds_read_b128).
#include "common.h"
#include <cstdlib>
#include <map>
#include <type_traits>
#include <mutex>
#include <utility>

namespace opus {

thread_local LaunchEvents *tl_launch_ev = nullptr;
// OPUS_KNOB_MISC<i>=<int> presets the scratch knobs (A/B runs of whole bench.py steps; opus_debug_knob changes them at run time)
static Knobs knobs_from_env() {
    Knobs k;
    for (int i = 0; i < 8; ++i) {
        char name[32];
        snprintf(name, sizeof(name), "OPUS_KNOB_MISC%d", i);
        if (const char *v = getenv(name)) k.misc[i] = atoi(v);
    }
    return k;
}
Knobs g_knobs = knobs_from_env();

hipError_t ensure_dyn_lds(const void *fn, size_t bytes) {
    static std::mutex mu;
    static std::map<std::pair<int, const void *>, size_t> done;
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return e;
    std::lock_guard<std::mutex> lock(mu);
    size_t &have = done[std::make_pair(dev, fn)];
    if (bytes <= have) return hipSuccess;
    e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
    if (e == hipSuccess) have = bytes;
    return e;
}

// erf-GELU (nn.GELU(), fair_esm gelu): 0.5 x (1 + erf(x / sqrt 2)).  erf by Abramowitz-Stegun 7.1.26
// (|error| <= 1.5e-7, far below the fp16 rounding of the result): one v_exp + one v_rcp + 6 FMAs instead
// of the ~40-instruction libm erff; on the 168 M activations of an ESM-2 fc1 GEMM that is 20 % of the kernel.
__device__ __forceinline__ float gelu_erf(float x) {
    const float z = fabsf(x) * 0.70710678118654752f;
    const float t = __builtin_amdgcn_rcpf(1.0f + 0.3275911f * z);   // v_rcp_f32 (1 ulp): an IEEE division is ~10 instructions
    const float poly = t * (0.254829592f + t * (-0.284496736f + t * (1.421413741f + t * (-1.453152027f + t * 1.061405429f))));
    const float erf_abs = 1.0f - poly * __expf(-z * z);
    return 0.5f * x * (1.0f + copysignf(erf_abs, x));
}
__device__ __forceinline__ float silu(float x) { return x * __builtin_amdgcn_rcpf(1.0f + __expf(-x)); }   // (v_rcp_f32, as gelu_erf)

// ------------------------------------------------------------------------------------------------
// skinny.  Workgroup = PB panels (PB = 2 for the gate/up pair, else 1) x (nwaves / PB) k-parts.
// MFMA 16x16x32 f16: A lane (m = li, g) holds x[m][32s + 8g + e]; B lane holds the 16 B it loaded;
// C lane holds C[m = 4g + r][n = li].
//
// ALDS (small M: M*K*2 <= 64 KB, i.e. decode): the workgroup first stages the activation rows in LDS
// as fp16 - with NORM it reads the fp32 residual stream, accumulates sum(h^2) per row on the way and
// converts - while its first batch of weight loads is already in flight; the main loop then feeds the
// MFMA A operand from LDS (broadcast reads) and the vector-memory path carries nothing but weights,
// double-buffered U chunks deep (2 * U KB in flight per wave).
// !ALDS (16 < M <= 64 or huge K): A fragments come through a buffer descriptor (rows >= M read as
// zeros without a branch), NORM accumulates sum(h^2) from those same loads.
template <int MT, int EPI, bool NORM, bool ALDS>
__global__ __launch_bounds__(1024) void gemm_skinny_kernel(GemmParams p) {
    extern __shared__ __attribute__((aligned(16))) float red[];
    OPUS_ARGS_ONE_BATCH(p);
    constexpr int PB = EPI == EPI_SILU_GU16 ? 2 : 1;
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);     // (uniform: the k-range arithmetic below stays scalar and 32-bit -
    const int nwaves = blockDim.x >> 6;                            //  as per-lane 64-bit divisions it was ~350 VALU instructions
    const int wpp = nwaves / PB;                       // waves per panel     in front of the first weight load)
    const int pl = PB == 1 ? 0 : wave / wpp, kpart = wave - pl * wpp;
    const int chunks = p.K >> 6;
    const int npanels = (p.N + 15) >> 4;   // W is stored with its rows padded to a multiple of 16
    int panel = blockIdx.x * PB + pl;
    panel = panel < npanels ? panel : npanels - 1;
    const int c0 = kpart_begin(chunks, kpart, wpp);
    const int c1 = kpart_begin(chunks, kpart + 1, wpp);
    const int g = lane >> 4, li = lane & 15;
    // dynamic LDS: [red nwaves*MT*256 f32][rss nwaves*MT*16 f32][wss 16*16 f32][xs M*K f16]
    float *rss = red + nwaves * MT * 256;
    float *wss = rss + nwaves * MT * 16;
    half_t *xs = reinterpret_cast<half_t *>(wss + 256);

    const half_t *wp = p.W + ((int64_t)panel * chunks) * 1024 + lane * 8;
    f4 acc[MT];
    float ssq[MT];
#pragma unroll
    for (int i = 0; i < MT; ++i) { acc[i] = f4{0.f, 0.f, 0.f, 0.f}; ssq[i] = 0.f; }

    if constexpr (ALDS) {
        static_assert(!ALDS || MT == 1, "LDS-staged activations are for M <= 16");
        constexpr int U = 4;
        // two NAMED register sets (a runtime-indexed [2][U] array would be placed in scratch)
        h8 wlA[U], whA[U], wlB[U], whB[U];
        // rotated start of every wave's k-range (see gemm_wide_kernel: panel stride = 128 KB at K = 4096 puts the whole chip on
        // the same memory channels).  Single-GEMM timings are within noise either way; the batch-1 step as a whole is 3.5 %
        // faster (109.8 -> 106.0 ms, two alternating runs each).  The start depends on the wave only: of the multipliers
        // tried on the step, (workgroup, wave) = (0, 3) was best (104.1 ms), (5, 3) 105.4-106.0, (1, 0) no better than none.
        const int nck = c1 - c0;
        const int rot = (!p.no_rot && nck >= 2) ? (int)((wave * 3u) % (unsigned)nck) : 0;
        auto phys = [&](int c) { const int q = c + rot; return q < c1 ? q : q - nck; };
        auto wload = [&](h8 (&wl)[U], h8 (&wh)[U], int c) {
#pragma unroll
            for (int u = 0; u < U; ++u) {
                if (c + u < c1) {
                    const h8 *ptr = reinterpret_cast<const h8 *>(wp + (int64_t)phys(c + u) * 1024);
                    wl[u] = __builtin_nontemporal_load(ptr);
                    wh[u] = __builtin_nontemporal_load(ptr + 64);
                } else {
                    wl[u] = h8{0, 0, 0, 0, 0, 0, 0, 0};
                    wh[u] = wl[u];
                }
            }
        };
        wload(wlA, whA, c0);                            // weights first: HBM latency covers the staging
        // ---- stage A rows (and sum of squares) ----
        const int nthr = blockDim.x;
        for (int m = 0; m < p.M; ++m) {
            if (NORM) {
                const float4 *src = reinterpret_cast<const float4 *>(p.Af + (int64_t)m * p.lda);
                float s = 0.f;
                for (int i = tid; i < (p.K >> 2); i += nthr) {
                    const float4 v = src[i];
                    s += (v.x * v.x + v.y * v.y) + (v.z * v.z + v.w * v.w);
                    *reinterpret_cast<h4 *>(xs + (int64_t)m * p.K + 4 * i) = h4{(half_t)v.x, (half_t)v.y, (half_t)v.z, (half_t)v.w};
                }
#pragma unroll
                for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
                if (lane == 0) wss[m * 16 + wave] = s;
            } else {
                const h8 *src = reinterpret_cast<const h8 *>(p.A + (int64_t)m * p.lda);
                for (int i = tid; i < (p.K >> 3); i += nthr)
                    *reinterpret_cast<h8 *>(xs + (int64_t)m * p.K + 8 * i) = src[i];
            }
        }
        __syncthreads();
        const int mr = li < p.M ? li : p.M - 1;          // rows >= M duplicate the last row: never stored
        const h8 *xr = reinterpret_cast<const h8 *>(xs + (int64_t)mr * p.K + g * 8);
        auto compute = [&](const h8 (&wl)[U], const h8 (&wh)[U], int c) {
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int cc = phys(c + u < c1 ? c + u : c1 - 1);
                const h8 al = xr[cc * 8], ah = xr[cc * 8 + 4];
                acc[0] = mfma16(al, wl[u], acc[0]);
                acc[0] = mfma16(ah, wh[u], acc[0]);
            }
        };
        for (int c = c0; c < c1; c += 2 * U) {
            if (c + U < c1) wload(wlB, whB, c + U);
            compute(wlA, whA, c);
            if (c + U < c1) {
                if (c + 2 * U < c1) wload(wlA, whA, c + 2 * U);
                compute(wlB, whB, c + U);
            }
        }
    } else {
        // A rows through a buffer descriptor: lanes whose row does not exist get an out-of-range
        // offset, for which the hardware returns zeros without touching memory and without a branch.
        constexpr int ESZ = NORM ? 4 : 2;
        const __amdgpu_buffer_rsrc_t arsrc = __builtin_amdgcn_make_buffer_rsrc(
            NORM ? (void *)p.Af : (void *)p.A, 0, (int)((int64_t)((p.M - 1) * p.lda + p.K) * ESZ), 0x00020000);
        int aoff[MT];
#pragma unroll
        for (int i = 0; i < MT; ++i) {
            const int m = 16 * i + li;
            aoff[i] = m < p.M ? (int)((m * p.lda + g * 8) * ESZ) : 0x40000000;
        }
        // chunks in flight per wave (bounded by the 128-VGPR budget of a 16-wave workgroup)
        constexpr int U = NORM ? (MT == 1 ? 4 : 1) : (MT == 1 ? 4 : (MT == 2 ? 2 : 1));
        typedef unsigned int u4 __attribute__((ext_vector_type(4)));
        struct ARaw { u4 v[NORM ? 4 : 2]; };
        auto load_a = [&](int i, int c, ARaw &r) {
            const int o = aoff[i] + c * 64 * ESZ;
            if (NORM) {
                r.v[0] = __builtin_amdgcn_raw_buffer_load_b128(arsrc, o, 0, 0);
                r.v[1] = __builtin_amdgcn_raw_buffer_load_b128(arsrc, o + 16, 0, 0);
                r.v[2] = __builtin_amdgcn_raw_buffer_load_b128(arsrc, o + 128, 0, 0);
                r.v[3] = __builtin_amdgcn_raw_buffer_load_b128(arsrc, o + 144, 0, 0);
            } else {
                r.v[0] = __builtin_amdgcn_raw_buffer_load_b128(arsrc, o, 0, 0);
                r.v[1] = __builtin_amdgcn_raw_buffer_load_b128(arsrc, o + 64, 0, 0);
            }
        };
        auto conv_a = [&](int i, const ARaw &r, h8 &lo, h8 &hi) {
            if (NORM) {
                const float4 a0 = __builtin_bit_cast(float4, r.v[0]), a1 = __builtin_bit_cast(float4, r.v[1]);
                const float4 b0 = __builtin_bit_cast(float4, r.v[2]), b1 = __builtin_bit_cast(float4, r.v[3]);
                ssq[i] += (a0.x * a0.x + a0.y * a0.y) + (a0.z * a0.z + a0.w * a0.w) + (a1.x * a1.x + a1.y * a1.y) +
                          (a1.z * a1.z + a1.w * a1.w) + (b0.x * b0.x + b0.y * b0.y) + (b0.z * b0.z + b0.w * b0.w) +
                          (b1.x * b1.x + b1.y * b1.y) + (b1.z * b1.z + b1.w * b1.w);
                lo = h8{(half_t)a0.x, (half_t)a0.y, (half_t)a0.z, (half_t)a0.w, (half_t)a1.x, (half_t)a1.y, (half_t)a1.z, (half_t)a1.w};
                hi = h8{(half_t)b0.x, (half_t)b0.y, (half_t)b0.z, (half_t)b0.w, (half_t)b1.x, (half_t)b1.y, (half_t)b1.z, (half_t)b1.w};
            } else {
                lo = __builtin_bit_cast(h8, r.v[0]);
                hi = __builtin_bit_cast(h8, r.v[1]);
            }
        };
        int c = c0;
        for (; c + U <= c1; c += U) {
            h8 wl[U], wh[U];
            ARaw ar[U][MT];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const h8 *ptr = reinterpret_cast<const h8 *>(wp + (int64_t)(c + u) * 1024);
                wl[u] = __builtin_nontemporal_load(ptr);
                wh[u] = __builtin_nontemporal_load(ptr + 64);
            }
#pragma unroll
            for (int u = 0; u < U; ++u)
#pragma unroll
                for (int i = 0; i < MT; ++i) load_a(i, c + u, ar[u][i]);
#pragma unroll
            for (int u = 0; u < U; ++u)
#pragma unroll
                for (int i = 0; i < MT; ++i) {
                    h8 al, ah;
                    conv_a(i, ar[u][i], al, ah);
                    acc[i] = mfma16(al, wl[u], acc[i]);
                    acc[i] = mfma16(ah, wh[u], acc[i]);
                }
        }
        for (; c < c1; ++c) {
            const h8 *ptr = reinterpret_cast<const h8 *>(wp + (int64_t)c * 1024);
            const h8 wl = __builtin_nontemporal_load(ptr), wh = __builtin_nontemporal_load(ptr + 64);
            ARaw ar[MT];
#pragma unroll
            for (int i = 0; i < MT; ++i) load_a(i, c, ar[i]);
#pragma unroll
            for (int i = 0; i < MT; ++i) {
                h8 al, ah;
                conv_a(i, ar[i], al, ah);
                acc[i] = mfma16(al, wl, acc[i]);
                acc[i] = mfma16(ah, wh, acc[i]);
            }
        }
        if (NORM) {   // row m's squares sit in the 4 lanes (li = m, g = 0..3)
#pragma unroll
            for (int i = 0; i < MT; ++i) {
                ssq[i] += __shfl_xor(ssq[i], 16, 64);
                ssq[i] += __shfl_xor(ssq[i], 32, 64);
            }
        }
    }

    // cross-wave reduction through LDS: red[wave][i][r][lane] (+ rss[wave][i][16] for !ALDS NORM)
    if (nwaves > 1 || (NORM && !ALDS)) {
#pragma unroll
        for (int i = 0; i < MT; ++i) {
#pragma unroll
            for (int r = 0; r < 4; ++r) red[((wave * MT + i) * 4 + r) * 64 + lane] = acc[i][r];
            if (NORM && !ALDS && g == 0) rss[(wave * MT + i) * 16 + li] = ssq[i];
        }
        __syncthreads();
        if (wave != 0) return;
    }
    f4 up[MT];
#pragma unroll
    for (int i = 0; i < MT; ++i) {
        up[i] = f4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            float s = acc[i][r];
            for (int w = 1; w < wpp; ++w) s += red[((w * MT + i) * 4 + r) * 64 + lane];
            acc[i][r] = s;
            if (PB == 2) {
                float u = 0.f;
                for (int w = wpp; w < nwaves; ++w) u += red[((w * MT + i) * 4 + r) * 64 + lane];
                up[i][r] = u;
            }
        }
    }

    // epilogue: lane holds C[m = 16i + 4g + r][n = 16 * panel0 + li]
    const int n0 = blockIdx.x * PB * 16;
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int ml = 4 * g + r;
            const int m = 16 * i + ml;
            if (m >= p.M) continue;
            float rstd = 1.0f;
            if (NORM) {
                float q = 0.f;
                if (ALDS) for (int w = 0; w < nwaves; ++w) q += wss[m * 16 + w];
                else for (int w = 0; w < wpp; ++w) q += rss[(w * MT + i) * 16 + ml];
                rstd = rsqrtf(q / (float)p.K + p.norm_eps);
            }
            const int n = n0 + li;
            if (n >= p.N) continue;
            float v;
            int no;
            if (EPI == EPI_SILU_GU16) {
                float gate = acc[i][r] * rstd, upv = up[i][r] * rstd;
                if (p.bias) { gate += p.bias[n]; upv += p.bias[n + 16]; }
                v = silu(gate) * upv;
                no = (n0 >> 1) + li;
            } else {
                v = acc[i][r] * rstd;
                if (p.bias) v += p.bias[n];
                if (EPI == EPI_GELU) v = gelu_erf(v);
                no = n;
            }
            if (p.residual) v += p.residual[(int64_t)m * p.ldr + no];
            if (p.out_f32) reinterpret_cast<float *>(p.C)[(int64_t)m * p.ldc + no] = v;
            else reinterpret_cast<half_t *>(p.C)[(int64_t)m * p.ldc + no] = (half_t)v;
        }
}

// Epilogue for accumulators computed as C^T tiles (weights as the MFMA A operand, activations as B):
// lane (li, g) holds C[m = li][n = 4g + r], r = 0..3, i.e. FOUR CONSECUTIVE COLUMNS of one row, so bias /
// residual / output move as one 16-B (fp32) or 8-B (fp16) access per lane instead of four scalar ones.
template <int EPI>
__device__ __forceinline__ void store4(const GemmParams &p, int m, int nb, f4 v, f4 up) {
    // nb = first of the 4 columns in the GEMM's N space (for GU16: of the gate tile); up = matching up tile
    const bool vec = ((p.ldc | p.ldr) & 3) == 0;
    if (EPI == EPI_SILU_GU16) {
        const int no = ((nb >> 5) << 4) + (nb & 15);          // column in the [M, N/2] output
        if (nb + 3 < p.N) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                float gate = v[r], u = up[r];
                if (p.bias) { gate += p.bias[nb + r]; u += p.bias[nb + 16 + r]; }
                v[r] = silu(gate) * u;
            }
        }
        nb = no;
    } else {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            if (p.bias && nb + r < p.N) v[r] += p.bias[nb + r];
            if (EPI == EPI_GELU) v[r] = gelu_erf(v[r]);
        }
    }
    const int nlim = EPI == EPI_SILU_GU16 ? p.N / 2 : p.N;
    if (vec && nb + 3 < nlim) {
        if (p.residual) {
            const float4 rr = *reinterpret_cast<const float4 *>(p.residual + (int64_t)m * p.ldr + nb);
            v[0] += rr.x; v[1] += rr.y; v[2] += rr.z; v[3] += rr.w;
        }
        if (p.out_f32) *reinterpret_cast<float4 *>(reinterpret_cast<float *>(p.C) + (int64_t)m * p.ldc + nb) = make_float4(v[0], v[1], v[2], v[3]);
        else *reinterpret_cast<h4 *>(reinterpret_cast<half_t *>(p.C) + (p.c_tiled ? tiled_off(m, nb, nlim) : (int64_t)m * p.ldc + nb)) = h4{(half_t)v[0], (half_t)v[1], (half_t)v[2], (half_t)v[3]};
    } else {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            if (nb + r >= nlim) continue;
            float x = v[r];
            if (p.residual) x += p.residual[(int64_t)m * p.ldr + nb + r];
            if (p.out_f32) reinterpret_cast<float *>(p.C)[(int64_t)m * p.ldc + nb + r] = x;
            else reinterpret_cast<half_t *>(p.C)[p.c_tiled ? tiled_off(m, nb + r, nlim) : (int64_t)m * p.ldc + nb + r] = (half_t)x;
        }
    }
}

// ------------------------------------------------------------------------------------------------
// tile: 128 x 128 x 64, 256 threads = 2x2 waves of 64x64 (4x4 MFMA 16x16x32 tiles each).
// A image: [128 rows][8 chunks of 16 B], chunk c of row r at c ^ ((r>>1)&7) (conflict-free
// ds_read_b128 for its 16-lane groups).  B image: the 8 weight panels' 2 KB blocks, copied linearly.
constexpr int TBM = 128, TBN = 128, TBK = 64;

__device__ __forceinline__ int swz(int row, int chunk) { return chunk ^ ((row >> 1) & 7); }

template <int EPI>
__global__ __launch_bounds__(256) void gemm_tile_kernel(GemmParams p, int tiles_m, int tiles_n, int ksplit) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    h8 *lds = reinterpret_cast<h8 *>(smem);   // [buf][A 1024 | B 1024] h8
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wr = wave >> 1, wc = wave & 1;
    // XCD-aware tile order: workgroup ids are dealt round-robin over the 8 XCDs, so give each XCD a
    // contiguous run of tiles (neighbours share the A row-panel in that XCD's private L2).
    int bid = blockIdx.x;
    {
        const int nwg = gridDim.x, q = nwg >> 3, rr = nwg & 7, xcd = bid & 7;
        bid = (xcd < rr ? xcd * (q + 1) : rr * (q + 1) + (xcd - rr) * q) + (bid >> 3);
    }
    // split-K (few output tiles, long K): slice `ks` of a tile accumulates k-tiles [kt0, kt1) and
    // stores its raw fp32 tile into slab ks of the workspace; splitk_reduce_kernel sums the slabs in a
    // fixed order (bitwise reproducible, no atomics) and applies the epilogue.
    const int ntile = tiles_m * tiles_n;
    const int ks = bid / ntile;
    bid -= ks * ntile;
    const int tm = bid / tiles_n, tn = bid % tiles_n;
    const int m0 = tm * TBM, n0 = tn * TBN;
    const int KT = p.K / TBK;
    const int kt0 = kpart_begin(KT, ks, ksplit), kt1 = kpart_begin(KT, ks + 1, ksplit);
    const int npanels = (p.N + 15) >> 4;   // W is stored with its rows padded to a multiple of 16

    const int srow = tid >> 3, schunk = tid & 7;
    const half_t *ag[4], *bg[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        int ra = m0 + srow + 32 * i;
        ra = ra < p.M ? ra : p.M - 1;
        ag[i] = p.A + (int64_t)ra * p.lda + schunk * 8;
        const int q = tid + 256 * i;                       // 16-B piece of the 8-panel B tile
        int pn = (n0 >> 4) + (q >> 7);
        pn = pn < npanels ? pn : npanels - 1;
        bg[i] = p.W + ((int64_t)pn * KT) * 1024 + (q & 127) * 8;
    }
    h8 sa[4], sb[4];
    auto gload = [&](int kt) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            sa[i] = *reinterpret_cast<const h8 *>(ag[i] + (int64_t)kt * TBK);
            sb[i] = *reinterpret_cast<const h8 *>(bg[i] + (int64_t)kt * 1024);
        }
    };
    auto lstore = [&](int buf) {
        h8 *A = lds + buf * 2048, *B = A + 1024;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int r = srow + 32 * i;
            A[r * 8 + swz(r, schunk)] = sa[i];
            B[tid + 256 * i] = sb[i];
        }
    };

    f4 acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = f4{0.f, 0.f, 0.f, 0.f};

    const int g = lane >> 4, li = lane & 15;
    gload(kt0);
    lstore(kt0 & 1);
    __syncthreads();
    for (int kt = kt0; kt < kt1; ++kt) {
        const int buf = kt & 1;
        if (kt + 1 < kt1) gload(kt + 1);
        const h8 *A = lds + buf * 2048, *B = A + 1024;
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            h8 af[4], bf[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int r = wr * 64 + i * 16 + li;
                af[i] = A[r * 8 + swz(r, 4 * s + g)];
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) bf[j] = B[(wc * 4 + j) * 128 + s * 64 + lane];
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    acc[i][j] = mfma16(bf[j], af[i], acc[i][j]);   // C^T tile
        }
        if (kt + 1 < kt1) lstore(buf ^ 1);
        __syncthreads();
    }

    if (ksplit > 1) {
        float *slab = p.ws + (int64_t)ks * p.M * p.N;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int m = m0 + wr * 64 + i * 16 + li;
            if (m >= p.M) continue;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int nb = n0 + wc * 64 + j * 16 + 4 * g;
                if ((p.N & 3) == 0 && nb + 3 < p.N) {
                    *reinterpret_cast<float4 *>(slab + (int64_t)m * p.N + nb) = make_float4(acc[i][j][0], acc[i][j][1], acc[i][j][2], acc[i][j][3]);
                } else {
#pragma unroll
                    for (int r = 0; r < 4; ++r)
                        if (nb + r < p.N) slab[(int64_t)m * p.N + nb + r] = acc[i][j][r];
                }
            }
        }
        return;
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int m = m0 + wr * 64 + i * 16 + li;
        if (m >= p.M) continue;
        if (EPI == EPI_SILU_GU16) {
#pragma unroll
            for (int jj = 0; jj < 2; ++jj) store4<EPI>(p, m, n0 + wc * 64 + jj * 32 + 4 * g, acc[i][2 * jj], acc[i][2 * jj + 1]);
        } else {
#pragma unroll
            for (int j = 0; j < 4; ++j) store4<EPI>(p, m, n0 + wc * 64 + j * 16 + 4 * g, acc[i][j], acc[i][j]);
        }
    }
}

// ------------------------------------------------------------------------------------------------
template <int MT, int EPI, bool NORM, bool ALDS>
static hipError_t launch_skinny_t(const GemmParams &p, hipStream_t s) {
    constexpr int PB = EPI == EPI_SILU_GU16 ? 2 : 1;
    const int groups = cdiv(p.N, 16 * PB);
    const int chunks = p.K / 64;
    // Waves per workgroup W in {4, 8, 16} (divisors of the 16 waves a CU holds at ~100 VGPRs, so whole
    // workgroups tile the CU), the smallest that puts >= ~3000 waves on the chip (tools/bench_skinny.hip:
    // the stream saturates from ~8 waves per CU), never more k-parts than 64-wide chunks.
    int W = (int64_t)groups * 4 >= 3000 ? 4 : ((int64_t)groups * 8 >= 3000 ? 8 : 16);
    static const int w_override = getenv("OPUS_SKINNY_W") ? atoi(getenv("OPUS_SKINNY_W")) : 0;   // tuning aid
    if (w_override >= PB) W = w_override;
    while (W > PB && (W / PB > chunks || (size_t)W * MT * 272 * sizeof(float) > 48 * 1024)) W >>= 1;
    int wpp = W / PB;
    if (wpp < 1) wpp = 1;
    W = wpp * PB;
    const size_t lds = (size_t)W * MT * (256 + 16) * sizeof(float) + 256 * sizeof(float) +
                       (ALDS ? (size_t)p.M * p.K * sizeof(half_t) : 0);
    if (lds > 48 * 1024) {
        hipError_t ea = ensure_dyn_lds(reinterpret_cast<const void *>(&gemm_skinny_kernel<MT, EPI, NORM, ALDS>), lds);
        if (ea != hipSuccess) return ea;
    }
    OPUS_LAUNCH(KC_SKINNY, (gemm_skinny_kernel<MT, EPI, NORM, ALDS>), dim3(groups), dim3(64 * W), lds, s, p);
    return hipGetLastError();
}

template <int EPI, bool NORM>
static hipError_t launch_skinny_e(const GemmParams &p, hipStream_t s) {
    if (p.M <= 16 && (int64_t)p.M * p.K * 2 <= 64 * 1024) return launch_skinny_t<1, EPI, NORM, true>(p, s);
    switch (cdiv(p.M, 16)) {
        case 1: return launch_skinny_t<1, EPI, NORM, false>(p, s);
        case 2: return launch_skinny_t<2, EPI, NORM, false>(p, s);
        case 3:
        case 4: return launch_skinny_t<4, EPI, NORM, false>(p, s);
    }
    return hipErrorInvalidValue;
}

// ------------------------------------------------------------------------------------------------
// mid (16 < M <= 128: batched decode, batched projectors, B = 1 prefill).  Still a weight-streaming
// problem (every weight byte once, HBM-bound) but with enough rows that the activations must be shared:
// workgroup = 4 waves = 4 weight panels (64 output columns) x all M rows x one k-part.  Per 64-k chunk
// the [M x 64] activation slice is staged once into LDS (double-buffered, XOR-swizzled; with NORM it is
// converted from the fp32 residual stream and sum(h^2) is accumulated on the way) and read by the four
// waves as MFMA fragments; each wave streams ITS panel straight from HBM into a 4-chunk register ring
// (non-temporal, 8 KB in flight per wave).  Accumulators are C^T tiles (16-B epilogue accesses).
// k-parts (gridDim.y > 1, chosen when N is too small to fill the chip) write fp32 slabs + partial
// sum(h^2); splitk_reduce_kernel combines them in a fixed order.
template <int MT, int EPI, bool NORM>
__global__ __launch_bounds__(256) void gemm_mid_kernel(GemmParams p, int ksplit) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int MP = 16 * MT;
    // 64-k chunks per LDS stage (one barrier each): deeper stages pay while the fp32/fp16 staging
    // registers still leave 2+ waves per SIMD (M <= 32); above that one chunk per stage measured faster
    constexpr int CH = MT <= 2 ? 4 : 1;
    constexpr int STAGE = CH * MP * 128;                             // bytes
    float *rss = reinterpret_cast<float *>(smem + 2 * STAGE);        // [MP] sum of squares per row
    float *xch = rss + MP;                                           // [2][MT][4][64] gate/up exchange
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int g = lane >> 4, li = lane & 15;
    const int chunks = p.K >> 6;
    const int npanels = (p.N + 15) >> 4;
    int panel = blockIdx.x * 4 + wave;
    const bool panel_ok = panel < npanels;
    panel = panel_ok ? panel : npanels - 1;
    const int ks = blockIdx.y;
    const int c0 = kpart_begin(chunks, ks, ksplit), c1 = kpart_begin(chunks, ks + 1, ksplit);
    const half_t *wp = p.W + ((int64_t)panel * chunks) * 1024 + lane * 8;

    // ---- A staging of one stage (CH chunks): NORM: float4 pieces (row = q/16, col4 = q%16 inside a chunk);
    //      else h8 pieces (row = q/8, 16-B chunk = q%8).  Pieces of chunk u are q = tid + 256 j.
    constexpr int NP = NORM ? MT : (MT + 1) / 2;                     // pieces per thread per chunk
    float4 af32[NORM ? CH * MT : 1];
    h8 af16[NORM ? 1 : CH * ((MT + 1) / 2)];
    float ssq[NORM ? MT : 1];
#pragma unroll
    for (int j = 0; j < (NORM ? MT : 1); ++j) ssq[j] = 0.f;
    auto a_load = [&](int c) {                                       // chunks c .. c+CH-1 (clamped)
#pragma unroll
        for (int u = 0; u < CH; ++u) {
            const int cc = c + u < c1 ? c + u : c1 - 1;
#pragma unroll
            for (int j = 0; j < NP; ++j) {
                const int q = tid + 256 * j;
                if (NORM) {
                    int row = q >> 4;
                    row = row < p.M ? row : p.M - 1;
                    af32[u * MT + j] = *reinterpret_cast<const float4 *>(p.Af + (int64_t)row * p.lda + (int64_t)cc * 64 + (q & 15) * 4);
                } else if (q < MP * 8) {
                    int row = q >> 3;
                    row = row < p.M ? row : p.M - 1;
                    af16[u * NP + j] = *reinterpret_cast<const h8 *>(p.A + (int64_t)row * p.lda + (int64_t)cc * 64 + (q & 7) * 8);
                }
            }
        }
    };
    auto a_store = [&](int buf, int c) {
        char *base = smem + buf * STAGE;
#pragma unroll
        for (int u = 0; u < CH; ++u) {
            const bool live = c + u < c1;                            // clamped duplicates must not be counted twice
#pragma unroll
            for (int j = 0; j < NP; ++j) {
                const int q = tid + 256 * j;
                if (NORM) {
                    const int row = q >> 4, c4 = q & 15;
                    const float4 v = af32[u * MT + j];
                    if (live) ssq[j] += (v.x * v.x + v.y * v.y) + (v.z * v.z + v.w * v.w);
                    const int chunk = (c4 >> 1) ^ ((row >> 1) & 7);
                    *reinterpret_cast<h4 *>(base + u * MP * 128 + row * 128 + chunk * 16 + (c4 & 1) * 8) =
                        h4{(half_t)v.x, (half_t)v.y, (half_t)v.z, (half_t)v.w};
                } else if (q < MP * 8) {
                    const int row = q >> 3;
                    *reinterpret_cast<h8 *>(base + u * MP * 128 + row * 128 + (((q & 7) ^ ((row >> 1) & 7)) << 4)) = af16[u * NP + j];
                }
            }
        }
    };

    f4 acc[MT];
#pragma unroll
    for (int i = 0; i < MT; ++i) acc[i] = f4{0.f, 0.f, 0.f, 0.f};
    // weights: two NAMED register sets of one stage each (runtime-indexed arrays would go to scratch)
    h8 wlA[CH], whA[CH], wlB[CH], whB[CH];
    auto w_load = [&](h8 (&wl)[CH], h8 (&wh)[CH], int c) {
#pragma unroll
        for (int u = 0; u < CH; ++u) {
            if (c + u < c1) {
                const h8 *ptr = reinterpret_cast<const h8 *>(wp + (int64_t)(c + u) * 1024);
                wl[u] = __builtin_nontemporal_load(ptr);
                wh[u] = __builtin_nontemporal_load(ptr + 64);
            } else {
                wl[u] = h8{0, 0, 0, 0, 0, 0, 0, 0};
                wh[u] = wl[u];
            }
        }
    };
    auto compute = [&](const h8 (&wl)[CH], const h8 (&wh)[CH], int buf) {
        const char *base = smem + buf * STAGE;
#pragma unroll
        for (int u = 0; u < CH; ++u)
#pragma unroll
            for (int i = 0; i < MT; ++i) {
                const int r = 16 * i + li;
                const h8 a0 = *reinterpret_cast<const h8 *>(base + u * MP * 128 + r * 128 + ((g ^ ((r >> 1) & 7)) << 4));
                const h8 a1 = *reinterpret_cast<const h8 *>(base + u * MP * 128 + r * 128 + (((4 + g) ^ ((r >> 1) & 7)) << 4));
                acc[i] = mfma16(wl[u], a0, acc[i]);   // C^T tile
                acc[i] = mfma16(wh[u], a1, acc[i]);
            }
    };
    w_load(wlA, whA, c0);
    a_load(c0);
    int buf = 0;
    for (int c = c0; c < c1; c += 2 * CH) {
        a_store(buf, c);
        __syncthreads();
        if (c + CH < c1) { a_load(c + CH); w_load(wlB, whB, c + CH); }
        compute(wlA, whA, buf);
        buf ^= 1;
        if (c + CH < c1) {
            a_store(buf, c + CH);
            __syncthreads();
            if (c + 2 * CH < c1) { a_load(c + 2 * CH); w_load(wlA, whA, c + 2 * CH); }
            compute(wlB, whB, buf);
            buf ^= 1;
        }
    }

    // ---- row statistics ----
    if (NORM) {
        __syncthreads();
#pragma unroll
        for (int j = 0; j < MT; ++j) {
            float s = ssq[j];
            s += __shfl_xor(s, 1, 64); s += __shfl_xor(s, 2, 64); s += __shfl_xor(s, 4, 64); s += __shfl_xor(s, 8, 64);
            if ((tid & 15) == 0) rss[(tid >> 4) + 16 * j] = s;
        }
        __syncthreads();
    }

    if (ksplit > 1) {
        float *slab = p.ws + (int64_t)ks * p.M * p.N;
        if (panel_ok) {
#pragma unroll
            for (int i = 0; i < MT; ++i) {
                const int m = 16 * i + li;
                const int nb = panel * 16 + 4 * g;
                if (m < p.M) {
#pragma unroll
                    for (int r = 0; r < 4; ++r)
                        if (nb + r < p.N) slab[(int64_t)m * p.N + nb + r] = acc[i][r];
                }
            }
        }
        if (NORM && blockIdx.x == 0 && tid < p.M) (p.ws + (int64_t)ksplit * p.M * p.N)[ks * p.M + tid] = rss[tid];
        return;
    }

    // ---- in-kernel epilogue ----
    float rstd[MT];
#pragma unroll
    for (int i = 0; i < MT; ++i) rstd[i] = NORM ? rsqrtf(rss[16 * i + li] / (float)p.K + p.norm_eps) : 1.0f;
    if (EPI == EPI_SILU_GU16) {
        // panels alternate gate / up: odd waves hand their tile to the even wave on their left
        __syncthreads();
        if (wave & 1) {
#pragma unroll
            for (int i = 0; i < MT; ++i)
#pragma unroll
                for (int r = 0; r < 4; ++r) xch[(((wave >> 1) * MT + i) * 4 + r) * 64 + lane] = acc[i][r] * rstd[i];
        }
        __syncthreads();
        if ((wave & 1) == 0 && panel_ok) {
#pragma unroll
            for (int i = 0; i < MT; ++i) {
                const int m = 16 * i + li;
                if (m >= p.M) continue;
                f4 up;
#pragma unroll
                for (int r = 0; r < 4; ++r) up[r] = xch[(((wave >> 1) * MT + i) * 4 + r) * 64 + lane];
                store4<EPI>(p, m, panel * 16 + 4 * g, acc[i] * rstd[i], up);
            }
        }
    } else if (panel_ok) {
#pragma unroll
        for (int i = 0; i < MT; ++i) {
            const int m = 16 * i + li;
            if (m < p.M) store4<EPI>(p, m, panel * 16 + 4 * g, acc[i] * rstd[i], acc[i]);
        }
    }
}

// ------------------------------------------------------------------------------------------------
// ring: the LDS-DMA pipelined MFMA kernel.  8 waves (2 x 4), wave tile (16 TM) x (16 TN), workgroup tile
// BM x BN = (32 TM) x (64 TN), K streamed in 32-deep stages through an NS-slot LDS ring filled by LDS-DMA
// (global_load_lds_dwordx4: no staging registers, no ds_write).  NS-1 stages stay in flight across the single
// raw s_barrier per stage behind a COUNTED s_waitcnt vmcnt (never 0 in the steady state):
// cdna_hip_programming.md "Pipelining across barriers" / T3+T4.  Instances:
//   <8,4,4>  256 x 256, 4 x 32 KB : encoder / prefill / batched-projector GEMMs (MFMA-bound, 1.0-1.1 PFLOP/s)
//   <4,2,6>  128 x 128, 6 x 16 KB : 64 < M <= 128 (weight-streaming, k-parts when N is small)
//   <2,2,8>   64 x 128, 8 x 12 KB : 16 < M <= 64 batched decode (weight-streaming, 7 stages in flight)
// A stage image [BM rows][4 x 16 B], the 16-B chunk c of row r stored in slot c ^ T[(r>>2)&3], T = {0,3,2,1}:
// conflict-free for the 16-lane groups of ds_read_b128 (rows 64 B apart).  glds writes LDS linearly, so the
// permutation is applied to the per-lane SOURCE address (rule 21).  B stage image = the weight panels' 1-KB half
// blocks, already in fragment order: linear.  Every wave issues the same number of glds per stage (when the A
// image has fewer than 8 pieces, waves repeat a piece: identical bytes to the same place) so one vmcnt count fits all.
// Hazards: RAW - a stage is read only after every wave's vmcnt for it and the barrier; WAR - the slot refilled in
// iteration ks was last read in iteration ks-1, those reads are retired (lgkmcnt(0)) before the barrier of iteration
// ks, and the refill is issued after it.
// k-parts (gridDim.y > 1): raw fp32 slabs, combined by splitk_reduce_kernel in a fixed order.
constexpr int GBK = 32;

__device__ __forceinline__ int aswz(int row) { return (-(row >> 2)) & 3; }

template <int N>
__device__ __forceinline__ void vmcnt_wait() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

template <int TM, int TN, int NS, int EPI>
__global__ __launch_bounds__(512) void gemm_ring_kernel(GemmParams p, int tiles_m, int tiles_n, int ksplit) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int BM = 32 * TM, BN = 64 * TN;
    constexpr int A_BYTES = BM * 64, STAGE = A_BYTES + BN * 64;
    constexpr int NA = BM / 16, CA = NA >= 8 ? NA / 8 : 1;           // 1-KB A pieces per stage / per wave
    constexpr int CB = BN / 128;                                     // 1-KB B pieces per wave (BN/16 panels over 8 waves)
    constexpr int P = CA + CB;                                       // glds per wave per stage
    static_assert(BN % 128 == 0 && (NS - 1) * P <= 63, "tile configuration");
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 2, wn = wave & 3;
    const int g = lane >> 4, li = lane & 15;
    int bid = blockIdx.x;
    {   // XCD-aware: consecutive workgroup ids are dealt round-robin over the 8 XCDs; give each XCD a contiguous run
        const int nwg = gridDim.x, q = nwg >> 3, rr = nwg & 7, xcd = bid & 7;
        bid = (xcd < rr ? xcd * (q + 1) : rr * (q + 1) + (xcd - rr) * q) + (bid >> 3);
    }
    // Grouped tile order inside each XCD's run: GM tile-rows are walked column by column, so the ~32 tiles an XCD
    // works on at one time form a compact 2-D patch sharing A rows and weight panels in that XCD's L2.
    constexpr int GM = 8;
    const int per_group = GM * tiles_n;
    const int grp = bid / per_group, rem_id = bid - grp * per_group;
    const int rows_here = (tiles_m - grp * GM) < GM ? (tiles_m - grp * GM) : GM;
    const int tm = grp * GM + rem_id % rows_here, tn = rem_id / rows_here;
    const int m0 = tm * BM, n0 = tn * BN;
    const int KS_all = p.K / GBK, KT64 = p.K >> 6;
    const int ky = blockIdx.y;
    const int s0 = kpart_begin(KS_all, ky, ksplit), s1 = kpart_begin(KS_all, ky + 1, ksplit);
    const int KS = s1 - s0;
    const int npanels = (p.N + 15) >> 4;

    typedef const __attribute__((address_space(1))) void *gptr_t;
    typedef __attribute__((address_space(3))) void *lptr_t;
    const half_t *srcA[CA], *srcB[CB];
    int ldsA[CA];
#pragma unroll
    for (int j = 0; j < CA; ++j) {
        const int piece = (wave * CA + j) % NA;                      // 16 rows x 64 B
        const int rl = 16 * piece + (lane >> 2);                     // row inside the tile
        int row = m0 + rl;
        row = row < p.M ? row : p.M - 1;
        srcA[j] = p.A + (int64_t)row * p.lda + (((lane & 3) ^ aswz(rl)) << 3) + (int64_t)s0 * GBK;
        ldsA[j] = piece * 1024;
    }
#pragma unroll
    for (int j = 0; j < CB; ++j) {
        int pn = (n0 >> 4) + wave * CB + j;
        pn = pn < npanels ? pn : npanels - 1;
        srcB[j] = p.W + ((int64_t)pn * KT64) * 1024 + lane * 8;
    }
    // rotated k-walk of the weight-streaming configuration (k-parts: the batched decode wo / down), as in gemm_wide_kernel;
    // a function of the COLUMN block and k-part only, so that a row's result does not depend on which rows share its launch
    const int krot = (ksplit > 1 && !p.no_rot && KS >= 2) ? (int)((unsigned)(tn * 3 + ky) % (unsigned)KS) : 0;
    auto stage_load = [&](int ks) {                                   // ks relative to s0
        char *base = smem + (ks % NS) * STAGE;
        const int kr = ks + krot < KS ? ks + krot : ks + krot - KS;   // k-step actually fetched into slot ks % NS
        const int ka = s0 + kr;
#pragma unroll
        for (int j = 0; j < CA; ++j)
            __builtin_amdgcn_global_load_lds((gptr_t)(srcA[j] + (int64_t)kr * GBK), (lptr_t)(base + ldsA[j]), 16, 0, 0);
#pragma unroll
        for (int j = 0; j < CB; ++j)
            __builtin_amdgcn_global_load_lds((gptr_t)(srcB[j] + (int64_t)(ka >> 1) * 1024 + (ka & 1) * 512),
                                             (lptr_t)(base + A_BYTES + (wave * CB + j) * 1024), 16, 0, 0);
    };

    f4 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) acc[i][j] = f4{0.f, 0.f, 0.f, 0.f};

    int aoff[TM];
#pragma unroll
    for (int i = 0; i < TM; ++i) {
        const int rl = wm * (16 * TM) + i * 16 + li;
        aoff[i] = rl * 64 + ((g ^ aswz(rl)) << 4);
    }
    const int boff = A_BYTES + (wn * TN) * 1024 + lane * 16;

    // Iteration ks: wait for this wave's part of stage ks+1 (the later ones stay in flight) and retire its fragment
    // reads of stage ks -> barrier (stage ks+1 complete, slot ks free) -> refill slot ks with stage ks+NS -> issue the
    // fragment reads of stage ks+1 into the second register set -> MFMAs of stage ks (they overlap those reads).
    auto wait_glds = [&](int rem) {   // rem = stages issued after the one being waited for
        switch (rem) {
            case 0: vmcnt_wait<0>(); break;
            case 1: vmcnt_wait<P>(); break;
            case 2: vmcnt_wait<(NS > 2 ? 2 : 1) * P>(); break;
            case 3: vmcnt_wait<(NS > 3 ? 3 : 1) * P>(); break;
            case 4: vmcnt_wait<(NS > 4 ? 4 : 1) * P>(); break;
            case 5: vmcnt_wait<(NS > 5 ? 5 : 1) * P>(); break;
            case 6: vmcnt_wait<(NS > 6 ? 6 : 1) * P>(); break;
            default: vmcnt_wait<(NS > 7 ? 7 : 1) * P>(); break;
        }
    };
    auto read_frags = [&](int ks, h8 (&af)[TM], h8 (&bf)[TN]) {
        const char *base = smem + (ks % NS) * STAGE;
#pragma unroll
        for (int i = 0; i < TM; ++i) af[i] = *reinterpret_cast<const h8 *>(base + aoff[i]);
#pragma unroll
        for (int j = 0; j < TN; ++j) bf[j] = *reinterpret_cast<const h8 *>(base + boff + j * 1024);
    };
    auto mfmas = [&](const h8 (&af)[TM], const h8 (&bf)[TN], int i0, int i1) {
#pragma unroll
        for (int i = 0; i < TM; ++i)
            if (i >= i0 && i < i1) {
#pragma unroll
                for (int j = 0; j < TN; ++j)
                    acc[i][j] = mfma16(bf[j], af[i], acc[i][j]);   // C^T tile
            }
    };
    const int last = KS - 1;
    auto issued_upto = [&](int x) { return x < last ? x : last; };
    // The fragment reads of stage ks+1 are placed AFTER the first row of MFMAs of stage ks: the compiler's waitcnt pass
    // cannot see through the hand-placed counters and puts a full LDS wait in front of the first MFMA that follows a
    // ds_read; there it only covers the (long finished) reads of the previous iteration, and the twelve new reads run
    // under the remaining 28 MFMAs instead of in front of all 32.
    auto step = [&](int ks, const h8 (&ca)[TM], const h8 (&cb)[TN], h8 (&na)[TM], h8 (&nb)[TN]) {
        if (ks + 1 < KS) {
            wait_glds(issued_upto(ks + NS - 1) - (ks + 1));
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // WAR: this wave's reads of slot ks are done
            __builtin_amdgcn_s_barrier();
            if (ks + NS < KS) stage_load(ks + NS);
        }
        __builtin_amdgcn_s_setprio(1);
        mfmas(ca, cb, 0, 1);
        __builtin_amdgcn_sched_barrier(0);
        if (ks + 1 < KS) read_frags(ks + 1, na, nb);
        __builtin_amdgcn_sched_barrier(0);
        mfmas(ca, cb, 1, TM);
        __builtin_amdgcn_s_setprio(0);
    };
    h8 a0[TM], b0[TN], a1[TM], b1[TN];
#pragma unroll
    for (int s = 0; s < NS; ++s)
        if (s < KS) stage_load(s);
    wait_glds(issued_upto(NS - 1));
    __builtin_amdgcn_s_barrier();
    read_frags(0, a0, b0);
    for (int ks = 0; ks < KS; ks += 2) {
        step(ks, a0, b0, a1, b1);
        if (ks + 1 < KS) step(ks + 1, a1, b1, a0, b0);
    }

    if (ksplit > 1) {
        float *slab = p.ws + (int64_t)ky * p.M * p.N;
#pragma unroll
        for (int i = 0; i < TM; ++i) {
            const int m = m0 + wm * (16 * TM) + i * 16 + li;
            if (m >= p.M) continue;
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                const int nb = n0 + wn * (16 * TN) + j * 16 + 4 * g;
                if ((p.N & 3) == 0 && nb + 3 < p.N) {
                    *reinterpret_cast<float4 *>(slab + (int64_t)m * p.N + nb) = make_float4(acc[i][j][0], acc[i][j][1], acc[i][j][2], acc[i][j][3]);
                } else {
#pragma unroll
                    for (int r = 0; r < 4; ++r)
                        if (nb + r < p.N) slab[(int64_t)m * p.N + nb + r] = acc[i][j][r];
                }
            }
        }
        return;
    }
#pragma unroll
    for (int i = 0; i < TM; ++i) {
        const int m = m0 + wm * (16 * TM) + i * 16 + li;
        if (m >= p.M) continue;
        if (EPI == EPI_SILU_GU16) {
#pragma unroll
            for (int jj = 0; jj < TN / 2; ++jj)
                store4<EPI>(p, m, n0 + wn * (16 * TN) + jj * 32 + 4 * g, acc[i][2 * jj], acc[i][2 * jj + 1]);
        } else {
#pragma unroll
            for (int j = 0; j < TN; ++j) store4<EPI>(p, m, n0 + wn * (16 * TN) + j * 16 + 4 * g, acc[i][j], acc[i][j]);
        }
    }
}

// ------------------------------------------------------------------------------------------------
// pp: 256 x 256 x 64 tile, 8 waves as 2 wave-rows x 4 wave-columns (128 x 64 outputs per wave), the two wave-rows run
// half a phase apart ("ping-pong"): every phase is  [fragment reads + 2 LDS-DMA requests + counted vmcnt] barrier
// [16 MFMAs at raised priority] barrier, and wave-row 1 starts one barrier late, so on every SIMD (waves w and w+4) one
// wave is in its MFMA cluster while the other reads LDS and issues DMA - neither waits behind the other's issue stalls.
// (cdna_hip_programming.md "The 256^2 8-phase template": same geometry, schedule re-derived here.)
//
// One K-tile of 64 = four phases = the four 64 x 32 quadrants of the wave's output in snake order
//   (rows 0-63, cols 0-31) -> (rows 0-63, cols 32-63) -> (rows 64-127, cols 32-63) -> (rows 64-127, cols 0-31)
// so a phase needs at most one new operand set: A rows-lo + B cols-lo, then B cols-hi, then A rows-hi, then nothing.
// The tile's operands therefore arrive as four 16-KB "half-tiles" in that order of need
//   H0 = A rows {0-63, 128-191}   H1 = B cols-lo of the 4 wave-columns   H2 = B cols-hi   H3 = A rows {64-127, 192-255}
// in 8 LDS slots (two K-tiles); half-tile h is requested in phase h-4 and first read in phase h - {0,1,1,1}, i.e. at
// least 3 phases later; its slot was last read >= 2 phases (4 barriers) before the request.  RAW: the reader passes a
// barrier after every wave's counted vmcnt (one phase later, both wave-rows); WAR: see above.
constexpr int PP_DIST = 6;   // half-tiles requested ahead (4..6)

// tile id -> (tile row, tile column): GM tile-rows are walked column by column, so the ~32 tiles an XCD works on at one time
// form a compact 2-D patch sharing A rows and weight panels in that XCD's L2
__host__ __device__ __forceinline__ void pp_tile_coords(int id, int tiles_m, int tiles_n, int &tm, int &tn, int GM = 8) {
    const int per_group = GM * tiles_n;
    const int grp = id / per_group, rem_id = id - grp * per_group;
    const int rows_here = (tiles_m - grp * GM) < GM ? (tiles_m - grp * GM) : GM;
    tm = grp * GM + rem_id % rows_here;
    tn = rem_id / rows_here;
}
template <int EPI, bool LNA>
__device__ __forceinline__ void pp_finish(const GemmParams &p, f4 (&acc)[8][4], char *smem, const int wave, const int m0, const int n0,
                                          const int bid, const int full_tiles, const int tail_split, const int kpart, const bool partial);
template <int EPI, bool LNA>
__device__ __forceinline__ void pp_epilogue(const GemmParams &p, f4 (&acc)[8][4], char *smem, const int wave, const int m0, const int n0,
                                            const char *pair_slab);
// LNA: consumer side of the fused LayerNorm (GemmParams::ln_stat / ln_colsum): y = rstd (acc - mu s) + c2 in the epilogue
template <int EPI, bool LNA = false>
__global__ __launch_bounds__(512) void gemm_pp_kernel(GemmParams p, int tiles_m, int tiles_n, int full_tiles, int tail_split) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int HT = 16384;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave >> 2, wc = wave & 3;
    const int g = lane >> 4, li = lane & 15;
    // Tail split.  The first `full_tiles` workgroups (a whole number of rounds of 256) each own one tile over the full K; the
    // tiles left over - a last, partly filled round that would cost a full tile time - are cut into `tail_split` k-parts each,
    // so that the round is as many workgroups but 1 / tail_split as long; the parts leave raw fp32 tiles in the workspace
    // for pp_tail_reduce_kernel (fixed summation order: bitwise reproducible).
    // Staggered start (GemmParams::stagger): every workgroup of a round reaches its epilogue together, and that burst - 164 MB
    // per round of 256 tiles with an fp32 + residual epilogue - is what the epilogue's time is made of.  Half of the first
    // round's workgroups (alternate CUs of every XCD under round-robin placement: a speed matter only) start `stagger` x ~4 us
    // late, so that their epilogues fall into the other half's main loops for the rest of the launch.
    if (p.stagger > 0 && blockIdx.x < 256 && ((blockIdx.x >> 3) & 1))
        for (int i = 0; i < p.stagger; ++i) __builtin_amdgcn_s_sleep(127);
    if (p.trace && tid == 0) p.trace[blockIdx.x * 4] = wall_clock64();
    int bid = blockIdx.x, kpart = 0;
    const bool partial = bid >= full_tiles;
    if (!partial) {   // XCD-aware order, as the ring kernel
        const int nwg = full_tiles, q = nwg >> 3, rr = nwg & 7, xcd = bid & 7;
        bid = (xcd < rr ? xcd * (q + 1) : rr * (q + 1) + (xcd - rr) * q) + (bid >> 3);
    } else {
        const int j = bid - full_tiles;
        bid = full_tiles + j / tail_split;
        kpart = j % tail_split;
    }
    int tm, tn;
    pp_tile_coords(bid, tiles_m, tiles_n, tm, tn, p.pp_gm);
    const int m0 = tm * 256, n0 = tn * 256;
    const int KT_all = p.K >> 6;
    const int kt0 = partial ? (int)((int64_t)KT_all * kpart / tail_split) : 0;
    const int kt1 = partial ? (int)((int64_t)KT_all * (kpart + 1) / tail_split) : KT_all;
    const int KT = kt1 - kt0, NH = 4 * KT;
    const int npanels = (p.N + 15) >> 4;

    typedef const __attribute__((address_space(1))) void *gptr_t;
    typedef __attribute__((address_space(3))) void *lptr_t;
    // DMA sources of this wave's two 1-KB pieces (q = 2 wave + j) of each kind of half-tile: 32-bit BYTE offsets from the
    // (scalar) matrix bases - the K-tile advance goes into the scalar base, so the main loop keeps 8 address registers instead
    // of 16 (it runs at the 256-register limit; the launcher refuses operands beyond 4 GB)
    unsigned offA[2][2], offB[2][2];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int q = 2 * wave + j;
        const int lr = 8 * q + (lane >> 3);                           // row inside the half-tile image
        const int chunk = (lane & 7) ^ ((lr >> 1) & 7);               // swizzle goes on the source address
#pragma unroll
        for (int rh = 0; rh < 2; ++rh) {
            int row = m0 + (lr >> 6) * 128 + rh * 64 + (lr & 63);
            row = row < p.M ? row : p.M - 1;
            offA[j][rh] = (unsigned)(((int64_t)row * p.lda + chunk * 8) * 2);
        }
        const int pi = q >> 1, s = q & 1;
#pragma unroll
        for (int ch = 0; ch < 2; ++ch) {
            int pn = (n0 >> 4) + (pi >> 1) * 4 + ch * 2 + (pi & 1);
            pn = pn < npanels ? pn : npanels - 1;
            offB[j][ch] = (unsigned)((((int64_t)pn * KT_all) * 1024 + s * 512 + lane * 8) * 2);
        }
    }
    const char *baseA = reinterpret_cast<const char *>(p.A) + (int64_t)kt0 * 128;
    const char *baseB = reinterpret_cast<const char *>(p.W) + (int64_t)kt0 * 2048;
    auto issue_half = [&](int hq) {
        const int kt = hq >> 2, i = hq & 3;
        char *dst = smem + (hq & 7) * HT + (2 * wave) * 1024;
        const char *sa = baseA + (int64_t)kt * 128, *sb = baseB + (int64_t)kt * 2048;     // scalar
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const char *src = (i == 0 || i == 3) ? sa + offA[j][i == 3] : sb + offB[j][i == 2];
            __builtin_amdgcn_global_load_lds((gptr_t)src, (lptr_t)(dst + j * 1024), 16, 0, 0);
        }
    };

    f4 acc[8][4];
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = f4{0.f, 0.f, 0.f, 0.f};
    h8 a0[4][2], a1[4][2], b0[2][2], b1[2][2];
    int aoff[4][2];
#pragma unroll
    for (int rt = 0; rt < 4; ++rt) {
        const int lr = wr * 64 + rt * 16 + li;
#pragma unroll
        for (int s = 0; s < 2; ++s) aoff[rt][s] = lr * 128 + (((4 * s + g) ^ ((lr >> 1) & 7)) << 4);
    }
    const int boff = (wc * 2) * 2048 + lane * 16;
    auto read_a = [&](int kt, int rh, h8 (&af)[4][2]) {
        const char *base = smem + (4 * (kt & 1) + (rh ? 3 : 0)) * HT;
#pragma unroll
        for (int rt = 0; rt < 4; ++rt)
#pragma unroll
            for (int s = 0; s < 2; ++s) af[rt][s] = *reinterpret_cast<const h8 *>(base + aoff[rt][s]);
    };
    auto read_b = [&](int kt, int ch, h8 (&bf)[2][2]) {
        const char *base = smem + (4 * (kt & 1) + 1 + ch) * HT + boff;
#pragma unroll
        for (int ct = 0; ct < 2; ++ct)
#pragma unroll
            for (int s = 0; s < 2; ++s) bf[ct][s] = *reinterpret_cast<const h8 *>(base + ct * 2048 + s * 1024);
    };
    auto mfma_quadrant = [&](int rh, int ch, const h8 (&af)[4][2], const h8 (&bf)[2][2]) {
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int rt = 0; rt < 4; ++rt)
#pragma unroll
            for (int ct = 0; ct < 2; ++ct)
#pragma unroll
                for (int s = 0; s < 2; ++s) {
#pragma unroll
                    for (int rhh = 0; rhh < 2; ++rhh)
#pragma unroll
                        for (int chh = 0; chh < 2; ++chh)
                            if (rhh == rh && chh == ch)
                                acc[4 * rhh + rt][2 * chh + ct] =
                                    mfma16(bf[ct][s], af[rt][s], acc[4 * rhh + rt][2 * chh + ct]);
                }
        __builtin_amdgcn_s_setprio(0);
    };
    // end of a phase's load section: request half-tile phi+D, then retire everything up to half-tile phi+2 (the reads of
    // phase phi+1).  D <= 6: the slot of phi+D last held phi+D-8, read no later than phase phi+D-8, i.e. >= 2 phases ago.
    constexpr int D = PP_DIST;
    auto feed = [&](int phi) {
        if (phi + D < NH) {
            issue_half(phi + D);
            vmcnt_wait<2 * (D - 2)>();
        } else {
            const int left = NH - 1 - (phi + 2);                      // half-tiles issued beyond phi+2 (tail)
            if (left >= 3) vmcnt_wait<6>();
            else if (left == 2) vmcnt_wait<4>();
            else if (left == 1) vmcnt_wait<2>();
            else vmcnt_wait<0>();
        }
    };

    // prologue: the first D half-tiles; K-tile 0 (half-tiles 0..3) must have landed
#pragma unroll
    for (int h = 0; h < D; ++h)
        if (h < NH) issue_half(h);
    if (NH >= D) vmcnt_wait<2 * (D - 4)>();
    else vmcnt_wait<0>();
    __builtin_amdgcn_s_barrier();
    if (p.trace && tid == 0) p.trace[blockIdx.x * 4 + 1] = wall_clock64();
    if (wr == 1) __builtin_amdgcn_s_barrier();                        // wave-row 1 runs one barrier behind
    read_a(0, 0, a0);

    for (int kt = 0; kt < KT; ++kt) {
        const int phi = 4 * kt;
        // phase 0: rows-lo x cols-lo   (rows-lo fragments were read in the previous phase 3 / the prologue)
        read_b(kt, 0, b0);
        feed(phi);
        __builtin_amdgcn_s_barrier();
        mfma_quadrant(0, 0, a0, b0);
        __builtin_amdgcn_s_barrier();
        // phase 1: rows-lo x cols-hi
        read_b(kt, 1, b1);
        feed(phi + 1);
        __builtin_amdgcn_s_barrier();
        mfma_quadrant(0, 1, a0, b1);
        __builtin_amdgcn_s_barrier();
        // phase 2: rows-hi x cols-hi
        read_a(kt, 1, a1);
        feed(phi + 2);
        __builtin_amdgcn_s_barrier();
        mfma_quadrant(1, 1, a1, b1);
        __builtin_amdgcn_s_barrier();
        // phase 3: rows-hi x cols-lo; its load section fetches the next K-tile's rows-lo fragments
        if (kt + 1 < KT) read_a(kt + 1, 0, a0);
        feed(phi + 3);
        __builtin_amdgcn_s_barrier();
        mfma_quadrant(1, 0, a1, b0);
        __builtin_amdgcn_s_barrier();
    }
    if (wr == 0) __builtin_amdgcn_s_barrier();                        // every wave executes the same number of barriers
    if (p.trace && tid == 0) p.trace[blockIdx.x * 4 + 2] = wall_clock64();
    pp_finish<EPI, LNA>(p, acc, smem, wave, m0, n0, bid, full_tiles, tail_split, kpart, partial);
}

// Everything behind the main loop of gemm_pp_kernel: the in-launch pair combine of a two-part tail tile, the raw slab of a
// k-part, the epilogue.  The main loop runs at the 256-register limit (128 accumulators + 96 operand fragments + DMA
// addresses), so nothing but the accumulators and wave-uniform values (scalar registers) crosses this boundary: the lane's
// coordinates are re-derived here from v_mbcnt (an asm the compiler cannot match with the thread id it was handed at entry, so
// that id is dead behind the prologue instead of being spilled around the loop).  Inlined: the accumulators stay where they are.
template <int EPI, bool LNA>
__device__ __forceinline__ void pp_finish(const GemmParams &p, f4 (&acc)[8][4], char *smem, const int wave, const int m0, const int n0,
                                          const int bid, const int full_tiles, const int tail_split, const int kpart, const bool partial) {
    int lane;
    asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(lane));
    const int tid = wave * 64 + lane;
    const int wr = wave >> 2, wc = wave & 3;
    const int g = lane >> 4, li = lane & 15;

    const char *pair_slab = nullptr;                                  // (uniform) the partner's published accumulators, if any
    if (partial && tail_split == 2 && p.combine_cnt) {
        // Two k-parts: combined INSIDE the launch, and the tile finished by this kernel's own epilogue.  The workgroup of the
        // pair that finishes its half first publishes its accumulators (write-through stores, drained, then a flag); the other
        // one waits for the flag - its partner is running, and waits for nobody - adds them to its own and runs the epilogue.
        // a + b in fp32 is the same number whichever half arrives first: bitwise reproducible.  One slab written and read
        // instead of two, and no pp_tail_reduce_kernel launch (29 us behind the prefill wo / down GEMMs).
        typedef unsigned int u4p __attribute__((ext_vector_type(4)));
        const int tt = bid - full_tiles;
        int *cnt = p.combine_cnt + 512 + 2 * tt;                      // (gemm_stream_kernel's tickets live below 512)
        int *word = reinterpret_cast<int *>(smem);
        if (tid == 0) *word = __hip_atomic_fetch_add(cnt, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __syncthreads();
        const int order = *word;                                      // uniform
        __syncthreads();
        if (order > 1 && tid == 0)                                    // a third ticket: the word was poisoned (an aborted launch)
            __hip_atomic_store(p.combine_cnt + HANDOFF_ERR, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const __amdgpu_buffer_rsrc_t srs = __builtin_amdgcn_make_buffer_rsrc((void *)(p.ws + ((int64_t)tt << 16)), 0, 1 << 18, 0x00020000);
        if (order == 0) {
#pragma unroll
            for (int i = 0; i < 8; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u4p, acc[i][j]), srs, ((i * 4 + j) * 512 + tid) * 16, 0, 16);   // aux 16 = sc1
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            if (tid == 0) __hip_atomic_store(cnt + 1, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            return;
        }
        if (tid == 0) {
            // bounded: the partner drew the first ticket, so it is running and needs nobody - unless the words were poisoned by an
            // aborted launch; then this waiter gives up, tells the host (error word -> OPUS_EHIP at its next synchronisation) and
            // finishes with whatever the slab holds instead of hanging the GPU
            int spins = 0;
            while (__hip_atomic_load(cnt + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0) {
                __builtin_amdgcn_s_sleep(4);
                if (++spins > HANDOFF_SPIN_LIMIT) {
                    __hip_atomic_store(p.combine_cnt + HANDOFF_ERR, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    break;
                }
            }
            __hip_atomic_store(cnt, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);       // re-armed for the next launch
            __hip_atomic_store(cnt + 1, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        __syncthreads();
        // The partner's accumulators are NOT added here: the epilogue adds them row tile by row tile (pair_slab below), eight
        // 16-byte write-through-coherent (sc1) loads per lane and pair of row tiles, requested one pair ahead.  Added into the
        // accumulators in one piece, the 128 sums were new values that had to be copied back into the registers the epilogue
        // expects where this path joins the plain one: 64 moves on EVERY tile and 9-12 spilled registers (round 3).
        pair_slab = reinterpret_cast<const char *>(p.ws + ((int64_t)tt << 16));
    } else if (partial) {   // raw fp32 tile [256][256] of this k-part (rows / columns beyond M / N hold clamped-row products: never read)
        float *slab = p.ws + (((int64_t)(bid - full_tiles) * tail_split + kpart) << 16);
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int r = wr * 128 + i * 16 + li;
#pragma unroll
            for (int j = 0; j < 4; ++j)
                *reinterpret_cast<float4 *>(slab + r * 256 + wc * 64 + j * 16 + 4 * g) = make_float4(acc[i][j][0], acc[i][j][1], acc[i][j][2], acc[i][j][3]);
        }
        return;
    }
    // ---- epilogue ----
    // (the lane's coordinates once more, from an asm placed HERE: every per-lane address of the epilogue then depends on a value
    //  that does not exist before this point, so the compiler cannot hoist that arithmetic - ~50 registers of it - above the
    //  pair combine, where the 128 accumulators and the partner's tile already fill the register file)
    pp_epilogue<EPI, LNA>(p, acc, smem, wave, m0, n0, pair_slab);
}

template <int EPI, bool LNA>
__device__ __forceinline__ void pp_epilogue(const GemmParams &p, f4 (&acc)[8][4], char *smem, const int wave, const int m0, const int n0,
                                            const char *pair_slab) {
    int lane;
    asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(lane));
    const int tid = wave * 64 + lane;
    const int wr = wave >> 2, wc = wave & 3;
    const int g = lane >> 4, li = lane & 15;
    // The accumulators hold 4 consecutive columns of 16 different rows per lane group: stored directly, a wave-wide
    // store touches 16 rows x 32-B pieces.  The LDS is idle now, so each wave turns its 16-row x 64-column slabs
    // through a private LDS patch and writes / reads-modifies-writes whole 16-B-per-lane row segments instead.
    constexpr int NO = EPI == EPI_SILU_GU16 ? 32 : 64;                 // output columns per wave
    const int nlim = EPI == EPI_SILU_GU16 ? p.N / 2 : p.N;
    const int nw0 = EPI == EPI_SILU_GU16 ? (n0 >> 1) + wc * 32 : n0 + wc * 64;   // first output column of this wave
    const bool rows16 = ((p.ldc & 7) == 0) && ((p.ldr & 3) == 0) && nw0 + NO <= nlim;
    // the partner's accumulators of a two-part tail tile, as published: element (i, j) of thread t at ((4 i + j) 512 + t) 16 bytes
    const __amdgpu_buffer_rsrc_t psrs = __builtin_amdgcn_make_buffer_rsrc((void *)pair_slab, 0, pair_slab ? 1 << 18 : 0, 0x00020000);
    auto pair_load = [&](int i, int j) {
        return __builtin_bit_cast(f4, __builtin_amdgcn_raw_buffer_load_b128(psrs, ((i * 4 + j) * 512 + tid) * 16, 0, 16));   // aux 16 = sc1
    };
    if (!rows16) {                                                    // ragged right edge / odd strides: element-wise path
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            f4 av[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) av[j] = acc[i][j];
            if (pair_slab) {
#pragma unroll
                for (int j = 0; j < 4; ++j) av[j] += pair_load(i, j);
            }
            const int m = m0 + wr * 128 + i * 16 + li;
            if (m >= p.M) continue;
            if (EPI == EPI_SILU_GU16) {
#pragma unroll
                for (int jj = 0; jj < 2; ++jj) store4<EPI>(p, m, n0 + wc * 64 + jj * 32 + 4 * g, av[2 * jj], av[2 * jj + 1]);
            } else {
#pragma unroll
                for (int j = 0; j < 4; ++j) store4<EPI>(p, m, n0 + wc * 64 + j * 16 + 4 * g, av[j], av[j]);
            }
        }
        return;
    }
    // Patch image.  64 output columns: rows of 64 floats, the 16-B slot q of row r kept at q ^ (r & 15) - conflict-free both
    // for the 8-lane groups of the ds_write_b128 (8 rows, one slot column) and for the 16-lane groups of the ds_read_b128
    // (lanes 0-3 / 12-15 of one row with lanes 4-11 of the next).  32 output columns (gate / up): rows padded by one slot;
    // the row-segment reads then collide 2-way on one lane of 16 (one extra LDS cycle per read: this padding, on both
    // widths, was the SQ_LDS_BANK_CONFLICT count of round 1's projector profile - 1.3 M cycles in 5 launches, 0.05 %).
    float *patch = reinterpret_cast<float *>(smem + wave * 16384);
    constexpr bool XSW = NO == 64;
    constexpr int PS = XSW ? NO : NO + 4;
    auto pidx = [&](int r, int c) { return XSW ? r * PS + ((((c >> 2) ^ (r & 15)) << 2) | (c & 3)) : r * PS + c; };
    // The bias of this lane's 16 GEMM columns is the same for all 8 row tiles, and the per-tile LDS round trip below is fenced
    // with asm barriers the compiler will not move loads across: fetched once here, not once per row tile (8 dependent L2
    // round trips, ~1 us each, were 3/4 of this epilogue).  The residual rows of tile i+1 are requested before tile i is
    // processed for the same reason.
    float bz[16];
#pragma unroll
    for (int q = 0; q < 16; ++q) bz[q] = 0.f;
    // (only the gate / up epilogue keeps its bias in registers: the others read it back from LDS per row tile - see `cb` below -,
    //  which is what lets the fp32 + residual modes hold the pair combine's 16 registers beside 128 + 64)
    if (p.bias && !LNA && EPI == EPI_SILU_GU16) {
        if (EPI == EPI_SILU_GU16) {
#pragma unroll
            for (int jj = 0; jj < 2; ++jj)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int nb = n0 + wc * 64 + jj * 32 + 4 * g + r;
                    bz[jj * 8 + r] = p.bias[nb];
                    bz[jj * 8 + 4 + r] = p.bias[nb + 16];
                }
        } else {
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int r = 0; r < 4; ++r) bz[j * 4 + r] = p.bias[n0 + wc * 64 + j * 16 + 4 * g + r];
        }
    }
    // fused LayerNorm, consumer side: the bias c2 and the column sums s of the folded weight for this wave's 64 columns are
    // kept in LDS behind the two patch images (32 persistent registers otherwise: the kernel is at the 256-register limit),
    // and - per pair of row tiles, requested one pair ahead like the residual rows - (mu, rstd) of this lane's row of each tile
    float *cb = patch + 2048;                                         // [bias 64 | colsum 64] floats
    float2 lst[2][2];
    auto load_stat = [&](int i, float2 &st) {
        int m = m0 + wr * 128 + i * 16 + li;
        m = m < p.M ? m : p.M - 1;
        st = reinterpret_cast<const float2 *>(p.ln_stat)[m];
    };
    if constexpr (EPI != EPI_SILU_GU16) {
        const int nc = n0 + wc * 64 + lane;
        cb[lane] = (p.bias && nc < p.N) ? p.bias[nc] : 0.f;
        if constexpr (LNA) cb[64 + lane] = p.ln_colsum ? p.ln_colsum[nc] : 0.f;   // (nullptr: RMSNorm, mu = 0)
    }
    if constexpr (LNA) {
        load_stat(0, lst[0][0]);
        load_stat(1, lst[0][1]);
    }
    // residual pieces of one row tile: fp32 output - 16 / RPP passes of one float4; fp16 output - 16 / RPP passes of two
    constexpr int LPR32 = NO / 4, RPP32 = 64 / LPR32, LPR16 = NO / 8, RPP16 = 64 / LPR16;
    constexpr int NR = (16 / RPP32) > 2 * (16 / RPP16) ? (16 / RPP32) : 2 * (16 / RPP16);
    auto load_res = [&](int i, float4 (&rr)[NR]) {
        const int mrow0 = m0 + wr * 128 + i * 16;
        if (p.out_f32) {
#pragma unroll
            for (int ps = 0; ps < 16 / RPP32; ++ps) {
                const int r = ps * RPP32 + lane / LPR32, c = (lane % LPR32) * 4;
                const int m = mrow0 + r < p.M ? mrow0 + r : p.M - 1;
                rr[ps] = *reinterpret_cast<const float4 *>(p.residual + (int64_t)m * p.ldr + nw0 + c);
            }
        } else {
#pragma unroll
            for (int ps = 0; ps < 16 / RPP16; ++ps) {
                const int r = ps * RPP16 + lane / LPR16, c = (lane % LPR16) * 8;
                const int m = mrow0 + r < p.M ? mrow0 + r : p.M - 1;
                rr[2 * ps] = *reinterpret_cast<const float4 *>(p.residual + (int64_t)m * p.ldr + nw0 + c);
                rr[2 * ps + 1] = *reinterpret_cast<const float4 *>(p.residual + (int64_t)m * p.ldr + nw0 + c + 4);
            }
        }
    };
    // Two row tiles per LDS round trip (the wave's 16-KB region holds four 4-KB patch images; two keep the residual
    // prefetch at 2 x 16 registers per stage): 4 serialized write -> wait -> read trips per tile instead of 8.
    constexpr int RB = 2, PIMG = 16 * PS;
    // Fused ESM rotary (GemmParams::rope_cs; the host side guarantees EPI_NONE, fp16 output, no residual, no ragged edge):
    // this wave's 64 columns are one head.  A lane then owns row lane / 4 and the 8 + 8 columns of four rotary pairs'
    // worth of dims (8 pc .. 8 pc + 7 and the same + 32) and needs 8 (cos, sin) pairs = 4 float4 of the table per row tile,
    // fetched a stage ahead through the (unused) residual slots.
    constexpr bool ROPE_OK = EPI == EPI_NONE && NO == 64 && NR >= 4;
    const bool rope_on = ROPE_OK && p.rope_cs != nullptr && !p.out_f32 && nw0 < p.rope_cols;   // wave-uniform
    // position of this lane's row (row lane / 4 of each of the 8 row tiles): row % rope_T, or - token-packed batches - from
    // the row -> position table, all 8 requested in one batch at the head of the rotary loop (inside load_cs each would be a
    // dependent round trip in front of the (cos, sin) fetch of its stage)
    auto rope_positions = [&](int (&rpos)[8]) {
        if constexpr (ROPE_OK) {
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const int mb = m0 + wr * 128 + i * 16;
                if (p.rope_pos) {
                    const int m = mb + (lane >> 2);
                    rpos[i] = p.rope_pos[m < p.M ? m : p.M - 1];
                } else {
                    int t = __builtin_amdgcn_readfirstlane(mb % p.rope_T) + (lane >> 2);
                    while (t >= p.rope_T) t -= p.rope_T;
                    rpos[i] = t;
                }
            }
        }
    };
    auto load_cs = [&](int i, float4 (&rr)[NR], const int (&rpos)[8]) {
        if constexpr (ROPE_OK) {
            const int t = rpos[i];
            const float4 *src = reinterpret_cast<const float4 *>(p.rope_cs + ((int64_t)t * 32 + (lane & 3) * 8) * 2);
#pragma unroll
            for (int q = 0; q < 4; ++q) rr[q] = src[q];
        }
    };
    // Stores go through buffer descriptors that cover exactly the rows of this tile that exist: a row >= M is dropped by the
    // hardware's range check, with no branch.  (With `if (m < M) store` the stores sit in exec-masked blocks; the compiler then
    // cannot count them and every wait for the NEXT pair's prefetched residual / rotary / LayerNorm rows became a drain of all
    // outstanding stores - 4 to 8 us per tile.)  For the same reason the whole loop is instantiated per epilogue mode - what
    // is loaded and stored is fixed at compile time inside each instance - and the mode is chosen once, outside.
    const int rows_left = p.M - m0 < 256 ? p.M - m0 : 256;
    const int esz = p.out_f32 ? 4 : 2;
    const __amdgpu_buffer_rsrc_t crs = __builtin_amdgcn_make_buffer_rsrc(
        reinterpret_cast<char *>(p.C) + (int64_t)m0 * p.ldc * esz, 0, (int)((int64_t)rows_left * p.ldc * esz), 0x00020000);
    const __amdgpu_buffer_rsrc_t xrs = __builtin_amdgcn_make_buffer_rsrc(
        p.ln_part ? reinterpret_cast<char *>(p.xh_out) + (int64_t)m0 * p.N * 2 : nullptr, 0, p.ln_part ? (int)((int64_t)rows_left * p.N * 2) : 0, 0x00020000);
    const __amdgpu_buffer_rsrc_t prs = __builtin_amdgcn_make_buffer_rsrc(
        p.ln_part ? reinterpret_cast<char *>(p.ln_part) + (int64_t)m0 * (p.N >> 6) * 8 : nullptr, 0, p.ln_part ? (int)((int64_t)rows_left * (p.N >> 6) * 8) : 0, 0x00020000);
    typedef unsigned int u4v __attribute__((ext_vector_type(4)));
    typedef unsigned int u2v __attribute__((ext_vector_type(2)));
    enum { M_F16 = 0, M_F32RES = 1, M_F32RES_LN = 2, M_ROPE = 3, M_GENERIC = 4 };
    auto run = [&](auto mode_tag, auto pair_tag) {
        constexpr int MODE = decltype(mode_tag)::value;
        // PAIR: the loop of a two-part tail tile that also adds its partner's half (pair_slab).  A separate instance, so that the
        // loop every other tile runs is exactly the one without it; its next-pair prefetch is issued BEHIND the current pair
        // (fp32 + residual epilogues hold 128 accumulators + 2 x 32 residual registers: the partner's 16 fit only while one of
        // the two residual stages is empty).
        constexpr bool PAIR = decltype(pair_tag)::value;
        constexpr bool RES = MODE == M_F32RES || MODE == M_F32RES_LN;          // fp32 output + fp32 residual
        constexpr bool CS = MODE == M_ROPE;
        float4 rbuf[2][RB][NR];
        if constexpr (RES || CS || MODE == M_GENERIC) {
#pragma unroll
            for (int q = 0; q < NR; ++q)
#pragma unroll
                for (int u = 0; u < RB; ++u) rbuf[0][u][q] = rbuf[1][u][q] = make_float4(0.f, 0.f, 0.f, 0.f);
        }
        int rpos[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        if constexpr (CS) rope_positions(rpos);
        auto prefetch = [&](int ib2) {                                    // rows of the pair ib2 (unconditional inside a mode)
#pragma unroll
            for (int u = 0; u < RB; ++u) {
                if constexpr (RES) load_res(ib2 * RB + u, rbuf[ib2 & 1][u]);
                else if constexpr (CS) load_cs(ib2 * RB + u, rbuf[ib2 & 1][u], rpos);
                else if constexpr (MODE == M_GENERIC) { if (p.residual) load_res(ib2 * RB + u, rbuf[ib2 & 1][u]); }
            }
            if constexpr (LNA) {
                load_stat(ib2 * RB, lst[ib2 & 1][0]);
                load_stat(ib2 * RB + 1, lst[ib2 & 1][1]);
            }
        };
        prefetch(0);
        f4 pb[4];                                                         // the partner's share of the NEXT row tile (one tile ahead)
        auto pair_fetch = [&](int i) {
#pragma unroll
            for (int j = 0; j < 4; ++j) pb[j] = pair_load(i, j);
        };
        // (the pair combine exists in the modes of the path - fp16 output, fp32 + residual (+ LayerNorm partials); the launcher sends
        //  two-part tiles of any other epilogue through the reduce kernel: launch_pp, `pair`)
        if constexpr (PAIR) pair_fetch(0);
#pragma unroll
        for (int ib = 0; ib < 8 / RB; ++ib) {
            if constexpr (!PAIR) {
                if (ib + 1 < 8 / RB) prefetch(ib + 1);
            }
            // 1. bias / activation in registers, 4 columns per lane -> patch image u, row li
#pragma unroll
            for (int u = 0; u < RB; ++u) {
                const int i = ib * RB + u;
                float *pu = patch + u * PIMG;
                // this row tile's sums: the accumulators as they stand, plus the partner's half on a two-part tail tile (a + b is the
                // same fp32 number whichever half waits).  A copy, so that only these 16 registers - not the accumulator file -
                // have two definitions where the two cases meet.
                f4 av[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) av[j] = acc[i][j];
                if constexpr (PAIR) {
#pragma unroll
                    for (int j = 0; j < 4; ++j) av[j] += pb[j];
                    if (i + 1 < 8) pair_fetch(i + 1);                     // (16 registers, re-used at once for the next row tile)
                }
                if (EPI == EPI_SILU_GU16) {
#pragma unroll
                    for (int jj = 0; jj < 2; ++jj) {
                        f4 v;
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            float gate = av[2 * jj][r], up = av[2 * jj + 1][r];
                            if constexpr (LNA) {                     // RMSNorm row scale (no bias in the gate / up projections)
                                gate *= lst[ib & 1][u].y;
                                up *= lst[ib & 1][u].y;
                            } else {
                                gate += bz[jj * 8 + r];
                                up += bz[jj * 8 + 4 + r];
                            }
                            v[r] = silu(gate) * up;
                        }
                        *reinterpret_cast<f4 *>(pu + pidx(li, jj * 16 + 4 * g)) = v;
                    }
                } else {
                    // (LNA: the 8 LDS reads of this row tile's bias / column-sum values are issued together, ahead of the patch
                    //  writes: read-wait-compute-write per 16 columns serialised the LDS queue, ~3 us per tile)
                    f4 b4[4], c4[LNA ? 4 : 1];
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        b4[j] = *reinterpret_cast<const f4 *>(cb + j * 16 + 4 * g);
                        if constexpr (LNA) c4[j] = *reinterpret_cast<const f4 *>(cb + 64 + j * 16 + 4 * g);
                    }
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        f4 v = av[j];
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            if constexpr (LNA) v[r] = __builtin_fmaf(lst[ib & 1][u].y, v[r] - lst[ib & 1][u].x * c4[j][r], b4[j][r]);
                            else v[r] += b4[j][r];
                            if (EPI == EPI_GELU) v[r] = gelu_erf(v[r]);
                        }
                        *reinterpret_cast<f4 *>(pu + pidx(li, j * 16 + 4 * g)) = v;
                    }
                }
            }
            // same-wave LDS round trip (in-order LDS queue); the fences keep the compiler from reordering across it
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            // 2. row segments: lane -> (row, 4- or 8-column piece)
#pragma unroll
            for (int u = 0; u < RB; ++u) {
                const int i = ib * RB + u;
                const float *pu = patch + u * PIMG;
                float4 (&rcur)[NR] = rbuf[ib & 1][u];
                const int tr0 = wr * 128 + i * 16;                          // first row of this row tile inside the 256-row tile
                const bool f32_out = MODE == M_GENERIC ? p.out_f32 != 0 : RES;
                if (f32_out) {
#pragma unroll
                    for (int ps = 0; ps < 16 / RPP32; ++ps) {
                        const int r = ps * RPP32 + lane / LPR32, c = (lane % LPR32) * 4;
                        f4 v = *reinterpret_cast<const f4 *>(pu + pidx(r, c));
                        if constexpr (RES || MODE == M_GENERIC) {
                            const float4 rr = rcur[ps];
                            v[0] += rr.x; v[1] += rr.y; v[2] += rr.z; v[3] += rr.w;
                        }
                        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u4v, v), crs, ((tr0 + r) * (int)p.ldc + nw0 + c) * 4, 0, 0);
                        if constexpr (MODE == M_F32RES_LN) {
                            // fused LayerNorm / RMSNorm, producer side: fp16(x) + this slab's (sum x, sum x^2) of the row
                            const h4 xh = h4{(half_t)v[0], (half_t)v[1], (half_t)v[2], (half_t)v[3]};
                            __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u2v, xh), xrs, ((tr0 + r) * p.N + nw0 + c) * 2, 0, 0);
                            float s1 = (v[0] + v[1]) + (v[2] + v[3]);
                            float s2 = (v[0] * v[0] + v[1] * v[1]) + (v[2] * v[2] + v[3] * v[3]);
#pragma unroll
                            for (int o = 1; o < LPR32; o <<= 1) { s1 += __shfl_xor(s1, o, 64); s2 += __shfl_xor(s2, o, 64); }
                            const float2 st = make_float2(s1, s2);
                            // (one lane per row writes; the others are sent out of range)
                            const int poff = (lane % LPR32) == 0 ? ((tr0 + r) * (p.N >> 6) + (nw0 >> 6)) * 8 : 0x7ffffff0;
                            __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u2v, st), prs, poff, 0, 0);
                        }
                    }
                } else if (CS || (MODE == M_GENERIC && rope_on)) {
                    if constexpr (ROPE_OK) {
                        const int r = lane >> 2, c = (lane & 3) * 8;
                        const f4 l0 = *reinterpret_cast<const f4 *>(pu + pidx(r, c)), l1 = *reinterpret_cast<const f4 *>(pu + pidx(r, c + 4));
                        const f4 h0 = *reinterpret_cast<const f4 *>(pu + pidx(r, c + 32)), h1 = *reinterpret_cast<const f4 *>(pu + pidx(r, c + 36));
                        const float qs = nw0 < p.rope_qcols ? p.rope_qscale : 1.0f;
                        const float csv[16] = {rcur[0].x, rcur[0].y, rcur[0].z, rcur[0].w, rcur[1].x, rcur[1].y, rcur[1].z, rcur[1].w,
                                               rcur[2].x, rcur[2].y, rcur[2].z, rcur[2].w, rcur[3].x, rcur[3].y, rcur[3].z, rcur[3].w};
                        h8 olo, ohi;
#pragma unroll
                        for (int e = 0; e < 8; ++e) {
                            // the projection is rounded to fp16 first, exactly as it is when stored and rotated by esm_rope_kernel
                            const float a = (float)(half_t)(e < 4 ? l0[e] : l1[e - 4]) * qs, b = (float)(half_t)(e < 4 ? h0[e] : h1[e - 4]) * qs;
                            const float cc = csv[2 * e], sn = csv[2 * e + 1];
                            float rl, rh;
                            rotate_pair(a, b, cc, sn, rl, rh);
                            olo[e] = (half_t)rl;
                            ohi[e] = (half_t)rh;
                        }
                        const int off = ((tr0 + r) * (int)p.ldc + nw0 + c) * 2;
                        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u4v, olo), crs, off, 0, 0);
                        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u4v, ohi), crs, off + 64, 0, 0);
                    }
                } else {
#pragma unroll
                    for (int ps = 0; ps < 16 / RPP16; ++ps) {
                        const int r = ps * RPP16 + lane / LPR16, c = (lane % LPR16) * 8;
                        f4 lo = *reinterpret_cast<const f4 *>(pu + pidx(r, c)), hi = *reinterpret_cast<const f4 *>(pu + pidx(r, c + 4));
                        if constexpr (MODE == M_GENERIC) {                // (fp16 output with an fp32 residual: not on the path)
                            const float4 r0 = rcur[2 * ps], r1 = rcur[2 * ps + 1];
                            lo[0] += r0.x; lo[1] += r0.y; lo[2] += r0.z; lo[3] += r0.w;
                            hi[0] += r1.x; hi[1] += r1.y; hi[2] += r1.z; hi[3] += r1.w;
                        }
                        const h8 o = h8{(half_t)lo[0], (half_t)lo[1], (half_t)lo[2], (half_t)lo[3], (half_t)hi[0], (half_t)hi[1], (half_t)hi[2], (half_t)hi[3]};
                        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u4v, o), crs, ((tr0 + r) * (int)p.ldc + nw0 + c) * 2, 0, 0);
                    }
                }
            }
            // the patch images are rewritten by the next pair of row tiles: the reads above must have retired (same wave, in order)
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            if constexpr (PAIR) {
                if (ib + 1 < 8 / RB) prefetch(ib + 1);
            }
        }
    };
    typedef std::false_type NP;
    typedef std::true_type WP;
    // (the pair combine exists in the modes of the path - fp16 output, fp32 + residual (+ LayerNorm partials); the launcher sends
    //  two-part tiles of any other epilogue through the reduce kernel: launch_pp, `pair`)
    if constexpr (LNA) {                                               // (the host side guarantees fp16 output, no residual)
        if (rope_on) run(std::integral_constant<int, ROPE_OK ? M_ROPE : M_F16>{}, NP{});
        else if (pair_slab) run(std::integral_constant<int, M_F16>{}, WP{});
        else run(std::integral_constant<int, M_F16>{}, NP{});
    } else if constexpr (EPI != EPI_NONE) {                            // GELU / gate-up epilogues: fp16 output on the path
        if (!p.out_f32 && !p.residual) {
            if (pair_slab) run(std::integral_constant<int, M_F16>{}, WP{});
            else run(std::integral_constant<int, M_F16>{}, NP{});
        } else run(std::integral_constant<int, M_GENERIC>{}, NP{});
    } else {
        if (rope_on) run(std::integral_constant<int, M_ROPE>{}, NP{});
        else if (!p.out_f32 && !p.residual) {
            if (pair_slab) run(std::integral_constant<int, M_F16>{}, WP{});
            else run(std::integral_constant<int, M_F16>{}, NP{});
        } else if (p.out_f32 && p.residual && p.ln_part) {
            if (pair_slab) run(std::integral_constant<int, M_F32RES_LN>{}, WP{});
            else run(std::integral_constant<int, M_F32RES_LN>{}, NP{});
        } else if (p.out_f32 && p.residual) {
            if (pair_slab) run(std::integral_constant<int, M_F32RES>{}, WP{});
            else run(std::integral_constant<int, M_F32RES>{}, NP{});
        } else run(std::integral_constant<int, M_GENERIC>{}, NP{});
    }
    if (p.trace && tid == 0) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        p.trace[blockIdx.x * 4 + 3] = wall_clock64();
    }
}

// Combines the k-parts of the tail tiles of gemm_pp_kernel.  Workgroup = 32 rows of one tile (8 per wave); a lane owns 4
// consecutive GEMM columns, so every load of a slab row is one fully coalesced 1-KB wave access and the epilogue moves 16-B
// (fp32) / 8-B (fp16) pieces.  Parts are summed in a fixed order, then the epilogue of the GEMM is applied.  With
// EPI_SILU_GU16 the gate columns (col % 32 < 16) fetch their up values from the lane 4 above.
template <int EPI>
__global__ __launch_bounds__(256) void pp_tail_reduce_kernel(GemmParams p, int tiles_m, int tiles_n, int full_tiles, int tail_split) {
    int tm, tn;
    pp_tile_coords(full_tiles + blockIdx.x, tiles_m, tiles_n, tm, tn, p.pp_gm);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int col = 4 * lane, n = tn * 256 + col;
    const float *base = p.ws + ((int64_t)blockIdx.x * tail_split << 16) + col;
    f4 bias4 = f4{0.f, 0.f, 0.f, 0.f}, bias_up = f4{0.f, 0.f, 0.f, 0.f};
    const bool gate_lane = (col & 31) < 16;
    if (p.bias) {
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            if (n + c < p.N) bias4[c] = p.bias[n + c];
            if (EPI == EPI_SILU_GU16 && gate_lane && n + 16 + c < p.N) bias_up[c] = p.bias[n + 16 + c];
        }
    }
    constexpr int RW = 8;                                            // rows per wave
    // fused LayerNorm, consumer side (GemmParams::ln_stat): column sums of this lane's 4 columns, (mu, rstd) per row below
    f4 csum = f4{0.f, 0.f, 0.f, 0.f};
    if (p.ln_stat && p.ln_colsum) {
#pragma unroll
        for (int c = 0; c < 4; ++c)
            if (n + c < p.N) csum[c] = p.ln_colsum[n + c];
    }
    f4 v[RW];
#pragma unroll
    for (int i = 0; i < RW; ++i) v[i] = f4{0.f, 0.f, 0.f, 0.f};
    const int r0 = blockIdx.y * 32 + wave * RW;
    for (int k0 = 0; k0 < tail_split; k0 += 4) {                      // 4 parts x 8 rows requested at a time, added in part order
        float4 t[4][RW];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int kk = k0 + u < tail_split ? k0 + u : tail_split - 1;
#pragma unroll
            for (int i = 0; i < RW; ++i) t[u][i] = *reinterpret_cast<const float4 *>(base + ((int64_t)kk << 16) + (r0 + i) * 256);
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            if (k0 + u >= tail_split) continue;
#pragma unroll
            for (int i = 0; i < RW; ++i) { v[i][0] += t[u][i].x; v[i][1] += t[u][i].y; v[i][2] += t[u][i].z; v[i][3] += t[u][i].w; }
        }
    }
    // residual rows of the 16-B path: requested together, ahead of the stores they would otherwise queue behind
    const int no_v = EPI == EPI_SILU_GU16 ? ((n >> 5) << 4) + (n & 15) : n;
    const bool vec_ok = no_v + 3 < (EPI == EPI_SILU_GU16 ? p.N / 2 : p.N) && ((p.ldc | p.ldr) & 3) == 0;
    float4 rres[RW];
#pragma unroll
    for (int i = 0; i < RW; ++i) {
        const int m = tm * 256 + r0 + i;
        rres[i] = make_float4(0.f, 0.f, 0.f, 0.f);
        if (p.residual && vec_ok && m < p.M) rres[i] = *reinterpret_cast<const float4 *>(p.residual + (int64_t)m * p.ldr + no_v);
    }
#pragma unroll
    for (int i = 0; i < RW; ++i) {
        const int m = tm * 256 + r0 + i;
        f4 x = v[i];
        int no = n;
        bool live = m < p.M;                                          // (the shuffle below needs every lane)
        if (EPI == EPI_SILU_GU16) {
            f4 up;
#pragma unroll
            for (int c = 0; c < 4; ++c) up[c] = __shfl_down(x[c], 4, 64);
            const float rs = p.ln_stat ? reinterpret_cast<const float2 *>(p.ln_stat)[m < p.M ? m : p.M - 1].y : 1.0f;   // RMSNorm row scale
#pragma unroll
            for (int c = 0; c < 4; ++c) x[c] = silu(x[c] * rs + bias4[c]) * (up[c] * rs + bias_up[c]);
            no = ((n >> 5) << 4) + (n & 15);
            live = live && gate_lane && n + 16 < p.N;
        } else {
            float2 st = make_float2(0.f, 1.f);
            if (p.ln_stat) st = reinterpret_cast<const float2 *>(p.ln_stat)[m < p.M ? m : p.M - 1];
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                x[c] = p.ln_stat ? __builtin_fmaf(st.y, x[c] - st.x * csum[c], bias4[c]) : x[c] + bias4[c];
                if (EPI == EPI_GELU) x[c] = gelu_erf(x[c]);
            }
            live = live && n < p.N;
        }
        if (!live) continue;
        const int nlim = EPI == EPI_SILU_GU16 ? p.N / 2 : p.N;
        if (vec_ok) {
            if (p.residual) {
                const float4 rr = rres[i];
                x[0] += rr.x; x[1] += rr.y; x[2] += rr.z; x[3] += rr.w;
            }
            if (p.out_f32) *reinterpret_cast<float4 *>(reinterpret_cast<float *>(p.C) + (int64_t)m * p.ldc + no) = make_float4(x[0], x[1], x[2], x[3]);
            else *reinterpret_cast<h4 *>(reinterpret_cast<half_t *>(p.C) + (int64_t)m * p.ldc + no) = h4{(half_t)x[0], (half_t)x[1], (half_t)x[2], (half_t)x[3]};
            if (EPI == EPI_NONE && p.ln_part) {
                // fused LayerNorm / RMSNorm, producer side, as gemm_pp_kernel's epilogue (N % 256 == 0: `live` is wave-uniform
                // here, and 16 consecutive lanes hold one 64-column slab of the row)
                *reinterpret_cast<h4 *>(p.xh_out + (int64_t)m * p.N + no) = h4{(half_t)x[0], (half_t)x[1], (half_t)x[2], (half_t)x[3]};
                float s1 = (x[0] + x[1]) + (x[2] + x[3]);
                float s2 = (x[0] * x[0] + x[1] * x[1]) + (x[2] * x[2] + x[3] * x[3]);
#pragma unroll
                for (int o = 1; o < 16; o <<= 1) { s1 += __shfl_xor(s1, o, 64); s2 += __shfl_xor(s2, o, 64); }
                if ((lane & 15) == 0) reinterpret_cast<float2 *>(p.ln_part)[(int64_t)m * (p.N >> 6) + (no >> 6)] = make_float2(s1, s2);
            }
        } else {
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                if (no + c >= nlim) continue;
                float y = x[c];
                if (p.residual) y += p.residual[(int64_t)m * p.ldr + no + c];
                if (p.out_f32) reinterpret_cast<float *>(p.C)[(int64_t)m * p.ldc + no + c] = y;
                else reinterpret_cast<half_t *>(p.C)[(int64_t)m * p.ldc + no + c] = (half_t)y;
            }
        }
    }
}

template <int EPI>
static hipError_t launch_pp(const GemmParams &p_in, hipStream_t s) {
    if (p_in.row_ssq) return hipErrorInvalidValue;                    // no row scale in this kernel's epilogue
    // gemm_pp_kernel addresses A and W with 32-bit unsigned byte offsets: operands of 4 GiB and more are refused (launch_gemm_
    // then falls back to the ring kernel's 64-bit addressing)
    if ((int64_t)p_in.M * p_in.lda * 2 >= (1ll << 32) || (int64_t)p_in.N * p_in.K * 2 >= (1ll << 32)) return hipErrorInvalidValue;
    GemmParams p = p_in;
    // fused LayerNorm (GemmParams::ln_*): the consumer form must be honoured (the caller handed over un-normalised rows),
    // the producer form is best effort (*ln_done reports it)
    const bool lna = p.ln_stat != nullptr;
    if (lna && (p.out_f32 || p.residual || (p.N & 255) || (p.ldc & 7) || (EPI == EPI_SILU_GU16 && (p.bias || p.ln_colsum)))) return hipErrorInvalidValue;
    const bool lnp = p.ln_part && p.xh_out && p.ln_done && EPI == EPI_NONE && p.out_f32 && (p.N & 255) == 0 && (p.ldc & 7) == 0 &&
                     (p.ldr & 3) == 0 && p.residual;
    if (lnp) *p.ln_done = 1;
    else p.ln_part = nullptr;
    p.stagger = g_knobs.misc[5];
    p.pp_gm = g_knobs.pp_gm;
    void (*kern)(GemmParams, int, int, int, int) = gemm_pp_kernel<EPI, false>;
    if (lna) kern = gemm_pp_kernel<EPI, true>;
    const int bm = cdiv(p.M, 256), bn = cdiv(p.N, 256);
    hipError_t ea = ensure_dyn_lds(reinterpret_cast<const void *>(kern), 8 * 16384);
    if (ea != hipSuccess) return ea;
    // tail split: when the last round of 256 workgroups would be at most half full, its tiles are cut into k-parts that fill it
    const int T = bm * bn, KT = p.K >> 6;
    int full = T, split = 1;
    static const bool no_tail = getenv("OPUS_NO_PP_TAIL") != nullptr;   // A/B aid
    // fused rotary: every wave must take the 16-B row-segment path and every tile this kernel's own epilogue
    static const bool no_rope_fuse = getenv("OPUS_NO_ROPE_FUSION") != nullptr;   // A/B aid
    const bool rope = p.rope_cs && p.rope_done && !no_rope_fuse && EPI == EPI_NONE && !p.out_f32 && !p.residual && (p.ldc & 7) == 0 &&
                      (p.N & 255) == 0 && (p.rope_cols & 63) == 0 && (p.rope_qcols & 63) == 0 && p.rope_T > 0;
    if (rope) *p.rope_done = 1;
    else p.rope_cs = nullptr;
    const int R = T % 256;
    if (!no_tail && !rope && p.ws && T > 256 && R > 0 && R <= 128) {   // (the fused rotary runs in gemm_pp_kernel's epilogue only)
        int sp = 256 / R;
        sp = sp > 8 ? 8 : sp;
        if (sp > KT / 4) sp = KT / 4;                                 // at least 4 k-tiles per part (pipeline prologue)
        while (sp > 1 && ((int64_t)R * sp << 18) > p.ws_bytes) --sp;
        // Worth it only when the part of a tile time it saves exceeds what the k-parts cost: a tile takes ~1.5 us per k-tile +
        // ~12 us of prologue / epilogue (OPUS_PP_TRACE), two parts exchanged inside the launch cost ~22 us, slabs + the reduce
        // launch ~30 us.  Measured at K = 1280, 2.5 rounds (tools/bench_gemm.py pair2): wo 135 us unsplit / 151 pair / 164 reduce,
        // QKV 291 / 295 / 300; at K = 5120: fc2 412 / 381 / 400.
        const double t_tile = 1.5 * KT + 12.0, gain = t_tile * (1.0 - 1.0 / sp), cost = (sp == 2 && p.combine_cnt && !g_knobs.misc[6]) ? 22.0 : 30.0;
        if (sp > 1 && gain > cost) { full = T - R; split = sp; }
    }
    // (Not kept, round 4: a stream-K cut of the last rounds - the last whole round and the partly filled one behind it dealt to
    //  256 workgroups as equal runs of K-tiles, a run = tail piece of one tile + whole tiles + head piece of another, the two
    //  pieces of a tile combined by the pair hand-off above; tools/experiments/r04_streamk_cut.patch, correct and bitwise
    //  reproducible on 12 shapes.  Slower everywhere (profiles/r04_streamk_ab.txt): ESM wo 159 vs 138 us, fc2 440 vs 414,
    //  21000-row fc2 314 vs 265, decoder down 601 vs 569.  The stream-K part runs at 2.47 us per k-tile instead of 1.6: runs start
    //  at different k offsets, so the 32 workgroups of an XCD no longer walk K in step, nothing they stream is shared in that
    //  XCD's L2 at the time it is needed, and 256 CUs x 64 KB per k-tile is 6.5 TB/s through the fabric - the kernel NEEDS
    //  the >= 2x L2 reuse that tiles in k-lockstep give it.  Equal k-parts (the split above) keep the lockstep; unequal ones cannot.)
    const int tail = T - full;
    static const bool trace = getenv("OPUS_PP_TRACE") != nullptr;      // tuning aid: per-workgroup section times on stderr
    if (trace) {
        static long long *tb = nullptr;
        const int nwg = full + tail * split;
        if (!tb) (void)hipMalloc((void **)&tb, (size_t)(1 << 16) * 4 * sizeof(long long));
        if (tb && nwg <= (1 << 16)) {
            GemmParams q = p;
            q.trace = tb;
            hipLaunchKernelGGL(kern, dim3(nwg), dim3(512), 8 * 16384, s, q, bm, bn, full, split);
            (void)hipStreamSynchronize(s);
            std::vector<long long> h((size_t)nwg * 4);
            (void)hipMemcpy(h.data(), tb, h.size() * sizeof(long long), hipMemcpyDeviceToHost);
            long long t0 = h[0], t1 = h[3];
            double pro = 0, mainl = 0, epi = 0;
            for (int i = 0; i < nwg; ++i) {
                t0 = std::min(t0, h[4 * i]); t1 = std::max(t1, h[4 * i + 3]);
                if (i < full) { pro += h[4 * i + 1] - h[4 * i]; mainl += h[4 * i + 2] - h[4 * i + 1]; epi += h[4 * i + 3] - h[4 * i + 2]; }
            }
            // start-time gaps: for every workgroup after the first 256, the time between its start and the latest earlier end on...
            // (unknown CU) - report instead the mean start time of each block of 256 workgroups relative to the kernel start
            fprintf(stderr, "[pp trace] M=%d N=%d K=%d epi=%d wg=%d (full %d): kernel %.1f us; per full tile: prologue %.2f us, main %.2f us, epilogue %.2f us\n",
                    p.M, p.N, p.K, EPI, nwg, full, (t1 - t0) * 0.01, pro / full * 0.01, mainl / full * 0.01, epi / full * 0.01);
            for (int r0 = 0; r0 < nwg; r0 += 256) {
                double st = 0, en = 0; int n = 0;
                for (int i = r0; i < nwg && i < r0 + 256; ++i, ++n) { st += h[4 * i] - t0; en += h[4 * i + 3] - t0; }
                fprintf(stderr, "   wg %5d..: mean start %.1f us, mean end %.1f us\n", r0, st / n * 0.01, en / n * 0.01);
            }
        }
    }
    // two k-parts: combined in the launch (gemm_pp_kernel); knob misc6 = 1: slabs + pp_tail_reduce_kernel as for more parts
    // the epilogue modes that carry the combine (pp_epilogue's dispatch): fp16 output without a residual in every epilogue;
    // fp32 output + fp32 residual in the plain epilogue only - with GELU / gate-up that combination runs the generic loop,
    // which never reads the partner's half: such tiles go through the reduce kernel
    const bool pair_mode = (!p.out_f32 && !p.residual) || (EPI == EPI_NONE && p.out_f32 && p.residual);
    const bool pair = split == 2 && p.combine_cnt && !g_knobs.misc[6] && pair_mode && (p.ldc & 7) == 0 && (p.ldr & 3) == 0 && (p.N & 255) == 0;
    if (!pair) p.combine_cnt = nullptr;
    OPUS_LAUNCH(KC_PP, kern, dim3(full + tail * split), dim3(512), 8 * 16384, s, p, bm, bn, full, split);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess || split == 1 || pair) return e;
    if (tl_launch_ev) tl_launch_ev->aux_bytes = ((double)tail * split * 65536 * 4) + (double)tail * 65536 * (p.out_f32 ? 4 : 2);
    OPUS_LAUNCH(KC_REDUCE, (pp_tail_reduce_kernel<EPI>), dim3(tail, 8), dim3(256), 0, s, p, bm, bn, full, split);
    return hipGetLastError();
}

// out[m][no] = epi(sum_ks slab[ks][m][n] + bias) (+ residual): one thread per output element.
template <int EPI>
__global__ __launch_bounds__(256) void splitk_reduce_kernel(GemmParams p, int ksplit) {
    const int nout = EPI == EPI_SILU_GU16 ? p.N / 2 : p.N;
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= (int64_t)p.M * nout) return;
    const int m = (int)(i / nout), no = (int)(i % nout);
    const int64_t slab = (int64_t)p.M * p.N;
    float rstd = 1.0f;
    if (p.Af) {   // fused RMSNorm: the k-parts left their partial sum(h^2) behind the slabs
        float q = 0.f;
        for (int k = 0; k < ksplit; ++k) q += p.ws[ksplit * slab + (int64_t)k * p.M + m];
        rstd = rsqrtf(q / (float)p.K + p.norm_eps);
    } else if (p.row_ssq) {   // row-scale fusion: the producer of A left per-block sums of squares of its rows
        const float *src = p.row_ssq + (int64_t)m * p.row_nblk;
        float q0 = 0.f, q1 = 0.f, q2 = 0.f, q3 = 0.f;
        int j = 0;
        for (; j + 4 <= p.row_nblk; j += 4) { q0 += src[j]; q1 += src[j + 1]; q2 += src[j + 2]; q3 += src[j + 3]; }
        for (; j < p.row_nblk; ++j) q0 += src[j];
        rstd = rsqrtf(((q0 + q1) + (q2 + q3)) / (float)p.K + p.norm_eps);
    }
    float v;
    if (EPI == EPI_SILU_GU16) {
        const int n = (no >> 4) * 32 + (no & 15);
        float gate = 0.f, up = 0.f;
        for (int k0 = 0; k0 < ksplit; k0 += 8) {                      // 8 slabs requested at a time, added in index order
            float tg[8], tu[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int kk = k0 + u < ksplit ? k0 + u : ksplit - 1;
                tg[u] = p.ws[kk * slab + (int64_t)m * p.N + n];
                tu[u] = p.ws[kk * slab + (int64_t)m * p.N + n + 16];
            }
#pragma unroll
            for (int u = 0; u < 8; ++u)
                if (k0 + u < ksplit) { gate += tg[u]; up += tu[u]; }
        }
        gate *= rstd;
        up *= rstd;
        if (p.bias) { gate += p.bias[n]; up += p.bias[n + 16]; }
        v = silu(gate) * up;
    } else {
        v = 0.f;
        for (int k0 = 0; k0 < ksplit; k0 += 8) {
            float t[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int kk = k0 + u < ksplit ? k0 + u : ksplit - 1;
                t[u] = p.ws[kk * slab + (int64_t)m * p.N + no];
            }
#pragma unroll
            for (int u = 0; u < 8; ++u)
                if (k0 + u < ksplit) v += t[u];
        }
        v *= rstd;
        if (p.bias) v += p.bias[no];
        if (EPI == EPI_GELU) v = gelu_erf(v);
    }
    if (p.residual) v += p.residual[(int64_t)m * p.ldr + no];
    if (p.out_f32) reinterpret_cast<float *>(p.C)[(int64_t)m * p.ldc + no] = v;
    else reinterpret_cast<half_t *>(p.C)[(int64_t)m * p.ldc + no] = (half_t)v;
    if (EPI == EPI_NONE && p.xh_out) {   // (host guarantees N % 256 == 0: the workgroup lies inside one row, no early return above)
        p.xh_out[p.xh_tiled ? tiled_off(m, no, p.N) : (int64_t)m * p.N + no] = (half_t)v;
        __shared__ float part[4];
        float q = v * v;
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) q += __shfl_xor(q, o, 64);
        if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = q;
        __syncthreads();
        if (threadIdx.x == 0) p.ssq_out[(int64_t)m * (p.N >> 8) + (no >> 8)] = (part[0] + part[1]) + (part[2] + part[3]);
    }
}

// The same for EPI_NONE with N % 256 == 0 (the wo / down / fc2 reduces of the decode step and of the few-tile GEMMs): a lane
// owns 4 consecutive columns - one 16-B load per slab, 16-B residual / output accesses - and a wave owns one 256-column block
// of a row, so the block's sum of squares for the row-scale fusion is one wave reduction (no barrier).
__global__ __launch_bounds__(256) void splitk_reduce4_kernel(GemmParams p, int ksplit) {
    const int64_t i = ((int64_t)blockIdx.x * 256 + threadIdx.x) * 4;
    if (i >= (int64_t)p.M * p.N) return;                             // (whole waves: N % 256 == 0)
    const int m = (int)(i / p.N), no = (int)(i % p.N);
    const int64_t slab = (int64_t)p.M * p.N;
    float rstd = 1.0f;
    if (p.Af) {
        float q = 0.f;
        for (int k = 0; k < ksplit; ++k) q += p.ws[ksplit * slab + (int64_t)k * p.M + m];
        rstd = rsqrtf(q / (float)p.K + p.norm_eps);
    } else if (p.row_ssq) {
        const float *src = p.row_ssq + (int64_t)m * p.row_nblk;
        float q0 = 0.f, q1 = 0.f, q2 = 0.f, q3 = 0.f;
        int j = 0;
        for (; j + 4 <= p.row_nblk; j += 4) { q0 += src[j]; q1 += src[j + 1]; q2 += src[j + 2]; q3 += src[j + 3]; }
        for (; j < p.row_nblk; ++j) q0 += src[j];
        rstd = rsqrtf(((q0 + q1) + (q2 + q3)) / (float)p.K + p.norm_eps);
    }
    // every load of the kernel is requested before the first sum (a `load, add` loop over the slabs is ksplit dependent round
    // trips: at 8 slabs that chain, not the 10 MB moved, was the kernel's 5 us); slabs are still added in index order
    float4 rr = make_float4(0.f, 0.f, 0.f, 0.f);
    f4 bz = f4{0.f, 0.f, 0.f, 0.f};
    if (p.residual) rr = *reinterpret_cast<const float4 *>(p.residual + (int64_t)m * p.ldr + no);
    if (p.bias) {
#pragma unroll
        for (int c = 0; c < 4; ++c) bz[c] = p.bias[no + c];
    }
    f4 v = f4{0.f, 0.f, 0.f, 0.f};
    for (int k0 = 0; k0 < ksplit; k0 += 8) {
        float4 t[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int kk = k0 + u < ksplit ? k0 + u : ksplit - 1;
            t[u] = *reinterpret_cast<const float4 *>(p.ws + kk * slab + i);
        }
#pragma unroll
        for (int u = 0; u < 8; ++u)
            if (k0 + u < ksplit) { v[0] += t[u].x; v[1] += t[u].y; v[2] += t[u].z; v[3] += t[u].w; }
    }
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        v[c] *= rstd;
        if (p.bias) v[c] += bz[c];
    }
    if (p.residual) { v[0] += rr.x; v[1] += rr.y; v[2] += rr.z; v[3] += rr.w; }
    if (p.out_f32) *reinterpret_cast<float4 *>(reinterpret_cast<float *>(p.C) + (int64_t)m * p.ldc + no) = make_float4(v[0], v[1], v[2], v[3]);
    else *reinterpret_cast<h4 *>(reinterpret_cast<half_t *>(p.C) + (int64_t)m * p.ldc + no) = h4{(half_t)v[0], (half_t)v[1], (half_t)v[2], (half_t)v[3]};
    if (p.xh_out) {
        *reinterpret_cast<h4 *>(p.xh_out + (p.xh_tiled ? tiled_off(m, no, p.N) : (int64_t)m * p.N + no)) = h4{(half_t)v[0], (half_t)v[1], (half_t)v[2], (half_t)v[3]};
        float q = (v[0] * v[0] + v[1] * v[1]) + (v[2] * v[2] + v[3] * v[3]);
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) q += __shfl_xor(q, o, 64);
        if ((threadIdx.x & 63) == 0) p.ssq_out[(int64_t)m * (p.N >> 8) + (no >> 8)] = q;
    }
}

template <int EPI>
static hipError_t launch_reduce(const GemmParams &p, int ks, hipStream_t s) {
    const int nout = EPI == EPI_SILU_GU16 ? p.N / 2 : p.N;
    GemmParams q = p;
    if (EPI == EPI_NONE && p.xh_out && p.ssq_out && p.fused_done && p.out_f32 && !p.Af && (p.N & 255) == 0 &&
        (int64_t)p.M * (p.N >> 8) <= p.ssq_cap) {
        *p.fused_done = 1;
        if (p.nblk_out) *p.nblk_out = p.N >> 8;
    } else {
        q.xh_out = nullptr;
    }
    if (tl_launch_ev) tl_launch_ev->aux_bytes = 4.0 * ks * p.M * p.N + (double)p.M * nout * ((p.out_f32 ? 4 : 2) + (p.residual ? 4 : 0));
    if (EPI == EPI_NONE && (p.N & 255) == 0 && ((p.ldc | p.ldr) & 3) == 0)
        OPUS_LAUNCH(KC_REDUCE, splitk_reduce4_kernel, dim3(cdiv((int64_t)p.M * p.N, 1024)), dim3(256), 0, s, q, ks);
    else
        OPUS_LAUNCH(KC_REDUCE, (splitk_reduce_kernel<EPI>), dim3(cdiv((int64_t)p.M * nout, 256)), dim3(256), 0, s, q, ks);
    return hipGetLastError();
}

hipError_t launch_splitk_reduce(const GemmParams &p, int ks, hipStream_t s) {
    switch (p.epi) {
        case EPI_NONE: return launch_reduce<EPI_NONE>(p, ks, s);
        case EPI_GELU: return launch_reduce<EPI_GELU>(p, ks, s);
        case EPI_SILU_GU16: return launch_reduce<EPI_SILU_GU16>(p, ks, s);
    }
    return hipErrorInvalidValue;
}

// ------------------------------------------------------------------------------------------------
// wide (16 < M <= 64, fp16 activations): batched-decode weight streaming with ONE barrier per 512 k.
// Workgroup = 8 waves = 8 weight panels (128 output columns) x all M rows x one k-part.  The [M x 512]
// activation slice of a stage is brought into LDS by LDS-DMA (8 x 1 KB pieces per wave, a full stage ahead,
// double-buffered: 2 x 64 KB at M = 64) and stays put while every wave streams ITS panel's weights for that
// slice straight from HBM through a 4-chunk register ring (non-temporal), so the vector-memory queue carries
// weights and nothing waits on a barrier for 64 MFMAs at a time.  vmcnt counts LDS-DMA and register loads
// together in issue order: the stage's DMA is issued before the 16 weight loads of the stage, of which at
// most 8 are still in flight at the end, so `s_waitcnt vmcnt(8)` retires the DMA without draining the ring.
template <int MT, int EPI>
__global__ __launch_bounds__(512) void gemm_wide_kernel(GemmParams p, int ksplit) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    OPUS_ARGS_ONE_BATCH(p);
    constexpr int MP = 16 * MT;
    constexpr int SC = MT <= 4 ? 8 : 4;                              // 64-k chunks per stage (2 stages <= 128 KB of LDS)
    constexpr int CIMG = MP * 128;                                   // bytes of one chunk image [MP][64] fp16
    constexpr int STAGE = SC * CIMG;
    constexpr int NPIECE = SC * MP / 8;                              // 1-KB DMA pieces per stage (8 rows x 128 B)
    constexpr int PW = NPIECE / 8;                                   // per wave (MT = 2: 4, MT = 4: 8)
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int g = lane >> 4, li = lane & 15;
    const int chunks = p.K >> 6;
    const int npanels = (p.N + 15) >> 4;
    int panel = blockIdx.x * 8 + wave;
    const bool panel_ok = panel < npanels;
    panel = panel_ok ? panel : npanels - 1;
    const int ky = blockIdx.y;
    const int c0 = kpart_begin(chunks, ky, ksplit), c1 = kpart_begin(chunks, ky + 1, ksplit);
    const half_t *wp = p.W + ((int64_t)panel * chunks) * 1024 + lane * 8;

    typedef const __attribute__((address_space(1))) void *gptr_t;
    typedef __attribute__((address_space(3))) void *lptr_t;
    // DMA piece q of a stage: chunk = q / (MP/8), rows 8*(q % (MP/8)) .. +7; lane -> row + lane/8, 16-B slot lane%8
    // holding source chunk16 = slot ^ ((row>>1)&7)  (the swizzle goes on the SOURCE address, rule 21)
    int pc[PW];
    const half_t *psrc[PW];
#pragma unroll
    for (int j = 0; j < PW; ++j) {
        const int q = wave * PW + j;
        const int ch = q / (MP / 8), r = 8 * (q % (MP / 8)) + (lane >> 3);
        int row = r < p.M ? r : p.M - 1;
        pc[j] = ch;
        psrc[j] = p.A + (int64_t)row * p.lda + (((lane & 7) ^ ((r >> 1) & 7)) << 3);
    }
    // Rotated start.  Every panel starts K * 32 bytes after the previous one - 128 KB at K = 4096 - and all waves of the chip
    // walk their panels at the same pace: at any moment every request has the same address modulo the panel stride, i.e.
    // falls on the same few memory channels (K = 4224 streams 7 % faster than 4096 for this reason alone).  Workgroup b
    // therefore walks its k-part from stage b % nstages and wraps around (+4 % at K = 4096; only the order of the fp32
    // sums changes).  Whole stages only.
    const int nck = c1 - c0;
    const int rot = (p.no_rot || nck % SC || nck < 2 * SC) ? 0 : (int)((blockIdx.x * 3u + blockIdx.y) % (unsigned)(nck / SC)) * SC;
    auto phys = [&](int c) { const int q = c + rot; return q < c1 ? q : q - nck; };
    auto dma_stage = [&](int buf, int c) {                            // chunks c .. c+SC-1 (clamped to the k-part)
#pragma unroll
        for (int j = 0; j < PW; ++j) {
            const int cc = phys(c + pc[j] < c1 ? c + pc[j] : c1 - 1);
            const int q = wave * PW + j;
            __builtin_amdgcn_global_load_lds((gptr_t)(psrc[j] + (int64_t)cc * 64),
                                             (lptr_t)(smem + buf * STAGE + q * 1024), 16, 0, 0);
        }
    };

    f4 acc[MT];
#pragma unroll
    for (int i = 0; i < MT; ++i) acc[i] = f4{0.f, 0.f, 0.f, 0.f};
    constexpr int U = 4;                                            // weight ring: 4 chunks = 8 KB per wave (8 measured slower)
    h8 wl[U], wh[U];
    auto w_load = [&](int u, int c) {
        if (c < c1) {
            const h8 *ptr = reinterpret_cast<const h8 *>(wp + (int64_t)phys(c) * 1024);
            wl[u] = __builtin_nontemporal_load(ptr);
            wh[u] = __builtin_nontemporal_load(ptr + 64);
        }
    };
    int aoff[MT];
#pragma unroll
    for (int i = 0; i < MT; ++i) {
        const int r = 16 * i + li;
        aoff[i] = r * 128 + ((g ^ ((r >> 1) & 7)) << 4);             // second k-step: ^ (4 << 4) on the chunk index
    }

    dma_stage(0, c0);
    // The weight ring is loaded with inline asm and waited for with hand-counted vmcnt: the compiler's waitcnt pass does not
    // count across a mix of LDS-DMA and register loads and would put vmcnt(0) - a full drain, DMA included - in front
    // of each stage's first MFMA.  The waits take the registers as in/out operands so that no MFMA can move above them.
    auto ld_nt = [&](h8 &lo, h8 &hi, const half_t *ptr) {
        asm volatile("global_load_dwordx4 %0, %2, off nt\n\tglobal_load_dwordx4 %1, %2, off offset:1024 nt"
                     : "=&v"(lo), "=&v"(hi) : "v"(ptr));            // (read-only weights: no memory clobber, so the
    };                                                                  //  LDS fragment reads can be scheduled around it)
#pragma unroll
    for (int u = 0; u < U; ++u) {
        const int cn = phys(c0 + u < c1 ? c0 + u : c1 - 1);
        ld_nt(wl[u], wh[u], wp + (int64_t)cn * 1024);
    }
    // 1 / rms of the activation rows for the fused RMSNorm (GemmParams::row_ssq), used in the epilogue: the workgroup sums the
    // producer's per-block sums of squares once, into LDS.  Requested BEHIND the first stage's DMA and weight loads (loads
    // return in order: the wait the compiler puts in front of the sums also retires those, which the first MFMA needs anyway).
    float *rstd_s = reinterpret_cast<float *>(smem + 2 * STAGE);
    if (p.row_ssq) block_row_rstd<8, MP>(p.row_ssq, p.M, p.row_nblk, 1.0f / (float)p.K, p.norm_eps, rstd_s, wave, lane);
    int buf = 0, cs = c0;
    // Full stages run a branch-free body (DMA of the next stage when there is one, 8 x [MFMAs, clamped refill]).  With
    // conditional loads or a conditional DMA the compiler cannot count what is in flight and drains the whole queue - the
    // just-issued DMA included - in front of the stage's first MFMA.  HAS_NEXT = false is the same body for a final FULL
    // stage (nothing to prefetch): short k-parts (the QKV / wo GEMMs of the batched decode step: 8-16 chunks) then never touch
    // the guarded path below, which is left to a genuinely partial last stage.
    auto full_stage = [&](auto has_next_tag) {
        constexpr bool HAS_NEXT = decltype(has_next_tag)::value;
        constexpr int PWN = HAS_NEXT ? PW : 0;
        // the DMA of this stage was issued before the (at most) 2 U weight loads still in flight
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * U) : "memory");
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");          // this wave's fragment reads of the other buffer
        __builtin_amdgcn_s_barrier();
        if (HAS_NEXT) dma_stage(buf ^ 1, cs + SC);
        const char *base = smem + buf * STAGE;
        // activation fragments of chunk u8+1 are read while chunk u8 is multiplied (two register sets, static parity)
        h8 fa[2][MT][2];
        auto read_chunk = [&](int u8, h8 (&f)[MT][2]) {
#pragma unroll
            for (int i = 0; i < MT; ++i) {
                f[i][0] = *reinterpret_cast<const h8 *>(base + u8 * CIMG + aoff[i]);
                f[i][1] = *reinterpret_cast<const h8 *>(base + u8 * CIMG + (aoff[i] ^ 64));
            }
        };
        read_chunk(0, fa[0]);
#pragma unroll
        for (int u8 = 0; u8 < SC; ++u8) {
            const int c = cs + u8;
            const int u = u8 & (U - 1);
            if (u8 + 1 < SC) read_chunk(u8 + 1, fa[(u8 + 1) & 1]);
            // younger than this chunk's pair (requested U chunks ago): the U-1 later refills, plus this stage's PW DMA
            // requests when the pair was requested before them (first U chunks of the stage)
            if (u8 < U) asm volatile("s_waitcnt vmcnt(%2)" : "+v"(wl[u]), "+v"(wh[u]) : "n"(2 * (U - 1) + PWN));
            else asm volatile("s_waitcnt vmcnt(%2)" : "+v"(wl[u]), "+v"(wh[u]) : "n"(2 * (U - 1)));
#pragma unroll
            for (int i = 0; i < MT; ++i) {
                acc[i] = mfma16(wl[u], fa[u8 & 1][i][0], acc[i]);   // C^T tile
                acc[i] = mfma16(wh[u], fa[u8 & 1][i][1], acc[i]);
            }
            const int cn = phys(c + U < c1 ? c + U : c1 - 1);        // (clamped: a few redundant loads at the very end)
            // the refill overwrites registers the MFMAs above read: in/out operands order it behind them
            asm volatile("" : "+v"(wl[u]), "+v"(wh[u]));
            ld_nt(wl[u], wh[u], wp + (int64_t)cn * 1024);
        }
    };
    for (; cs + SC < c1; cs += SC, buf ^= 1) full_stage(std::true_type{});
    if (cs + SC == c1) {
        full_stage(std::false_type{});
        cs += SC;
        buf ^= 1;
    }
    // everything the inline-asm loads have in flight is retired before the registers are handed back to the compiler
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
    for (int u = 0; u < U; ++u) asm volatile("" : "+v"(wl[u]), "+v"(wh[u]));
    if (cs < c1) {   // partial last stage: guarded loads (its DMA, issued a stage ago, has landed: vmcnt(0) above)
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        const char *base = smem + buf * STAGE;
#pragma unroll
        for (int u8 = 0; u8 < SC; ++u8) {
            const int c = cs + u8;
            if (c < c1) {                                            // wave-uniform
                const int u = u8 & (U - 1);
#pragma unroll
                for (int i = 0; i < MT; ++i) {
                    const h8 a0 = *reinterpret_cast<const h8 *>(base + u8 * CIMG + aoff[i]);
                    const h8 a1 = *reinterpret_cast<const h8 *>(base + u8 * CIMG + (aoff[i] ^ 64));
                    acc[i] = mfma16(wl[u], a0, acc[i]);   // C^T tile
                    acc[i] = mfma16(wh[u], a1, acc[i]);
                }
                w_load(u, c + U);
            }
        }
    }

    if (ksplit > 1) {
        float *slab = p.ws + (int64_t)ky * p.M * p.N;
        if (panel_ok) {
#pragma unroll
            for (int i = 0; i < MT; ++i) {
                const int m = 16 * i + li, nb = panel * 16 + 4 * g;
                if (m < p.M) {
#pragma unroll
                    for (int r = 0; r < 4; ++r)
                        if (nb + r < p.N) slab[(int64_t)m * p.N + nb + r] = acc[i][r];
                }
            }
        }
        return;
    }
    if (p.row_ssq) {   // RMSNorm of the activation rows, applied to the (linear) result: see GemmParams::row_ssq
#pragma unroll
        for (int i = 0; i < MT; ++i) {
            const float rs = rstd_s[16 * i + li];           // (written before the main loop's barriers)
#pragma unroll
            for (int r = 0; r < 4; ++r) acc[i][r] *= rs;
        }
    }
    if (EPI == EPI_SILU_GU16) {
        // panels alternate gate / up: odd waves hand their tile to the even wave on their left (LDS is free now)
        __syncthreads();
        float *xch = reinterpret_cast<float *>(smem);
        if (wave & 1) {
#pragma unroll
            for (int i = 0; i < MT; ++i)
#pragma unroll
                for (int r = 0; r < 4; ++r) xch[(((wave >> 1) * MT + i) * 4 + r) * 64 + lane] = acc[i][r];
        }
        __syncthreads();
        if ((wave & 1) == 0 && panel_ok) {
#pragma unroll
            for (int i = 0; i < MT; ++i) {
                const int m = 16 * i + li;
                if (m >= p.M) continue;
                f4 up;
#pragma unroll
                for (int r = 0; r < 4; ++r) up[r] = xch[(((wave >> 1) * MT + i) * 4 + r) * 64 + lane];
                store4<EPI>(p, m, panel * 16 + 4 * g, acc[i], up);
            }
        }
    } else if (panel_ok) {
#pragma unroll
        for (int i = 0; i < MT; ++i) {
            const int m = 16 * i + li;
            if (m < p.M) store4<EPI>(p, m, panel * 16 + 4 * g, acc[i], acc[i]);
        }
    }
}

// (measured and not kept, round 4: a 16-wave form of gemm_wide_kernel - two waves per panel, each taking half of every 8-chunk
//  stage, weights two stages ahead in an 8-chunk register ring: 256 KB of loads in flight per CU instead of 64, the depth at which
//  tools/stream_sweep.hip streams 235 MB at 7.5 TB/s.  Parity-green, 114 VGPRs, no scratch - and SLOWER: gate / up 49.7 vs 42.9 us,
//  lm_head 226 vs 189 us in isolation (tools/bench_gemm.py wide16 of that commit), the decode phase +6.5 ms.  The queue depth of a
//  pure stream is not what limits this kernel.)
// (measured and not kept, late round 4: 7 panels per workgroup - the eighth wave only stages activations - so that gate / up's 1792
//  panels make 256 workgroups instead of 224 (what an [8 gate | 8 up] panel layout would allow with the pair product done across
//  lanes): 41.8 vs 42.4 us on the 28672-column shape (-1.4 %; lm_head 207 vs 197 us), timing only.  0.6 us per layer is not
//  worth a second copy of every gate / up weight: the 224-workgroup grid costs less than the sweep's 256-vs-224 pure-stream gap.)
template <int MT, int EPI>
static hipError_t launch_wide(const GemmParams &p, hipStream_t s) {
    const int blocks = cdiv((p.N + 15) >> 4, 8), chunks = p.K / 64;
    int ks = 1;
    if (p.ws && blocks < 200) {                                      // few column groups: split K over workgroups
        // one workgroup per CU (128 KB of LDS each): blocks x ks must not spill into a second round of 256
        ks = 256 / blocks;
        ks = ks > 8 ? 8 : ks;
        if (ks > chunks / 8) ks = chunks / 8;
        while (ks > 1 && (int64_t)ks * p.M * p.N * 4 > p.ws_bytes) --ks;
        if (ks < 1) ks = 1;
    }
    if (p.ks_out) *p.ks_out = ks;
    const int lds = 2 * (MT <= 4 ? 8 : 4) * 16 * MT * 128 + 16 * MT * (int)sizeof(float);   // two stages + the rows' 1 / rms
    hipError_t ea = ensure_dyn_lds(reinterpret_cast<const void *>(&gemm_wide_kernel<MT, EPI>), lds);
    if (ea != hipSuccess) return ea;
    if (ks > 1 && p.row_ssq) {       // k-parts leave raw slabs: the row scale is applied by whoever combines them, not here
        GemmParams q = p;
        q.row_ssq = nullptr;
        OPUS_LAUNCH(KC_WIDE, (gemm_wide_kernel<MT, EPI>), dim3(blocks, ks), dim3(512), lds, s, q, ks);
    } else {
        OPUS_LAUNCH(KC_WIDE, (gemm_wide_kernel<MT, EPI>), dim3(blocks, ks), dim3(512), lds, s, p, ks);
    }
    hipError_t e = hipGetLastError();
    if (e != hipSuccess || ks == 1 || p.slab_only) return e;
    return launch_reduce<EPI>(p, ks, s);
}


template <int MT, int EPI, bool NORM>
static hipError_t launch_mid_t(const GemmParams &p, hipStream_t s) {
    const int blocks = cdiv((p.N + 15) >> 4, 4), chunks = p.K / 64;
    // k-parts: enough workgroups for ~8 waves per CU, at least 4 chunks each, slabs within the workspace
    int ks = 1;
    if (p.ws && blocks < 384) {
        ks = cdiv(512, blocks);
        ks = ks > 8 ? 8 : ks;
        if (ks > chunks / 4) ks = chunks / 4;
        while (ks > 1 && ((int64_t)ks * p.M * p.N + (int64_t)ks * p.M) * 4 > p.ws_bytes) --ks;
        if (ks < 1) ks = 1;
    }
    constexpr int CH = MT <= 2 ? 4 : 1;
    const size_t lds = 2 * CH * 16 * MT * 128 + 16 * MT * sizeof(float) + 2 * MT * 4 * 64 * sizeof(float);
    if (lds > 48 * 1024) {
        hipError_t ea = ensure_dyn_lds(reinterpret_cast<const void *>(&gemm_mid_kernel<MT, EPI, NORM>), lds);
        if (ea != hipSuccess) return ea;
    }
    if (p.row_ssq && ks == 1) return hipErrorInvalidValue;            // only the split-K reduce applies a row scale here
    OPUS_LAUNCH(KC_MID, (gemm_mid_kernel<MT, EPI, NORM>), dim3(blocks, ks), dim3(256), lds, s, p, ks);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess || ks == 1) return e;
    return launch_reduce<EPI>(p, ks, s);
}

template <int EPI, bool NORM>
static hipError_t launch_mid_e(const GemmParams &p, hipStream_t s) {
    switch (cdiv(p.M, 32)) {
        case 1: return launch_mid_t<2, EPI, NORM>(p, s);
        case 2: return launch_mid_t<4, EPI, NORM>(p, s);
        case 3: return launch_mid_t<6, EPI, NORM>(p, s);
        case 4: return launch_mid_t<8, EPI, NORM>(p, s);
    }
    return hipErrorInvalidValue;
}

template <int EPI>
static hipError_t launch_reduce(const GemmParams &p, int ks, hipStream_t s);

// ring kernel launcher; allow_split: choose k-parts so that ~256-512 workgroups stream the weights
template <int TM, int TN, int NS, int EPI>
static hipError_t launch_ring(const GemmParams &p, hipStream_t s, bool allow_split) {
    constexpr int BM = 32 * TM, BN = 64 * TN;
    constexpr int LDS = NS * (BM * 64 + BN * 64);
    const int bm = cdiv(p.M, BM), bn = cdiv(p.N, BN), KS = p.K / GBK;
    int ks = 1;
    if (allow_split && p.ws && bm * bn < 200) {
        // k-parts that fit ONE round of 256 workgroups (floor) rather than spill into a second one (ceil: 150 tiles x 2 = 300),
        // except where that would leave a long-K GEMM unsplit: measured (profiles/r05_ring_ks.txt, ceil -> floor) ESM QKV at 514 rows
        // 31.0 -> 24.7 us, fc2 32.9 -> 26.7, wo at 1 028 rows 26.0 -> 21.6 - no slabs / reduce launch, or one round instead of
        // 1.2 - but fc2 at 2 056 rows (170 tiles, K = 5120) 63.9 -> 73.1 and the batch-8 prefill (192 tiles, K >= 4096) +6 us per
        // GEMM: there the second half-round of k-parts is worth more than the reduce costs.
        const int ks_floor = 256 / (bm * bn);
        ks = (ks_floor >= 2 || p.K <= 2048) ? ks_floor : cdiv(256, bm * bn);
        if (ks < 1) ks = 1;
        ks = ks > 16 ? 16 : ks;
        if (ks > KS / 8) ks = KS / 8;
        while (ks > 1 && (int64_t)ks * p.M * p.N * 4 > p.ws_bytes) --ks;
        if (ks < 1) ks = 1;
    }
    hipError_t ea = ensure_dyn_lds(reinterpret_cast<const void *>(&gemm_ring_kernel<TM, TN, NS, EPI>), LDS);
    if (ea != hipSuccess) return ea;
    if (p.row_ssq && ks == 1) return hipErrorInvalidValue;            // only the split-K reduce applies a row scale here
    OPUS_LAUNCH(KC_RING, (gemm_ring_kernel<TM, TN, NS, EPI>), dim3(bm * bn, ks), dim3(512), LDS, s, p, bm, bn, ks);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess || ks == 1) return e;
    return launch_reduce<EPI>(p, ks, s);
}

template <int EPI>
static hipError_t launch_tile_e(const GemmParams &p, hipStream_t s) {
    const int tm = cdiv(p.M, TBM), tn = cdiv(p.N, TBN);
    hipError_t ea = ensure_dyn_lds(reinterpret_cast<const void *>(&gemm_tile_kernel<EPI>), 65536);
    if (ea != hipSuccess) return ea;
    static const bool no_big = getenv("OPUS_NO_BIG_GEMM") != nullptr;   // A/B aid
    static const int min_tiles = getenv("OPUS_PP_MIN_TILES") ? atoi(getenv("OPUS_PP_MIN_TILES")) : 128;   // tuning aid (pp wins from about half a chip of tiles)
    if (!no_big && (int64_t)cdiv(p.M, 256) * cdiv(p.N, 256) >= min_tiles) {   // enough 256 x 256 tiles to fill the chip
        static const bool no_pp = getenv("OPUS_NO_PP") != nullptr;          // A/B aid: single-phase ring kernels instead
        // gemm_pp_kernel addresses A and W with 32-bit byte offsets: operands of 4 GiB and more take the ring kernel (64-bit pointers)
        const bool over4g = (int64_t)p.M * p.lda * 2 >= (1ll << 32) || (int64_t)p.N * p.K * 2 >= (1ll << 32);
        if (no_pp || (over4g && !p.ln_stat)) {
            if ((p.residual || EPI == EPI_GELU) && p.K <= 4096) return launch_ring<4, 2, 4, EPI>(p, s, false);
            return launch_ring<8, 4, 4, EPI>(p, s, false);
        }
        return launch_pp<EPI>(p, s);
    }
    // Too few output tiles to fill 256 CUs (B = 1 prefill, single-protein encoder): split K so that
    // ~256-320 workgroups stream the weights, at least 4 k-tiles each, slabs within the workspace.
    const int ntile = tm * tn, KT = p.K / TBK;
    // single-protein encoder shapes (514..4112 rows): the LDS-DMA ring at the same 128 x 128 tile keeps 3 k-steps in flight
    // instead of one register prefetch: -0..20 % (tools/bench_gemm.py m96); <= 128 rows stay on the tile kernel (equal)
    static const bool no_small_ring = getenv("OPUS_NO_SMALL_RING") != nullptr;   // A/B aid
    if (p.M > 128 && !no_small_ring) return launch_ring<4, 2, 4, EPI>(p, s, true);
    int ks = 1;
    if (p.ws && ntile < 160) {
        ks = 320 / ntile;
        ks = ks > 8 ? 8 : ks;
        if (ks > KT / 4) ks = KT / 4;
        while (ks > 1 && (int64_t)ks * p.M * p.N * 4 > p.ws_bytes) --ks;
        if (ks < 1) ks = 1;
    }
    if (p.row_ssq && ks == 1) return hipErrorInvalidValue;            // only the split-K reduce applies a row scale here
    OPUS_LAUNCH(KC_TILE, (gemm_tile_kernel<EPI>), dim3(ntile * ks), dim3(256), 65536, s, p, tm, tn, ks);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess || ks == 1) return e;
    return launch_reduce<EPI>(p, ks, s);
}

// A/B aid: OPUS_NARROW_WIDE=1 routes every narrow GEMM at 5..64 rows (wo / down / qkv of the batched decode step) through
// gemm_wide_kernel + k-parts instead of the mid / ring kernels
static bool narrow_wide() {
    static const bool v = getenv("OPUS_NARROW_WIDE") != nullptr;
    return v;
}

int skinny_max_m() {
    static const int v = [] {
        const char *e = getenv("OPUS_SKINNY_MAX_M");
        const int x = e ? atoi(e) : 4;      // measured: batch 8 149 vs 166 ms, batch 16 168 vs 210 ms with 5..16 rows on the mid / wide kernels
        return x < 1 ? 1 : (x > SKINNY_MAX_M_CAP ? SKINNY_MAX_M_CAP : x);
    }();
    return v;
}

bool gemm_goes_wide(int M, int N) {
    static const bool no_mid = getenv("OPUS_NO_MID_GEMM") != nullptr, mid_v1 = getenv("OPUS_MID_V1") != nullptr;
    if (mid_v1 || N < 16384 || M <= SKINNY_MAX_M || M > 96) return false;
    return M > MID_MAX_M || !no_mid;
}

static hipError_t launch_gemm_(const GemmParams &p, hipStream_t s, int *klass);
hipError_t launch_gemm(const GemmParams &p, hipStream_t s, int *klass) {
    static const bool no_rot = getenv("OPUS_NO_KROT") != nullptr;       // A/B aid
    if (!no_rot) return launch_gemm_(p, s, klass);
    GemmParams q = p;
    q.no_rot = 1;
    return launch_gemm_(q, s, klass);
}
static bool pp_min_tiles_ok(int M, int N) {
    static const bool no_big = getenv("OPUS_NO_BIG_GEMM") != nullptr, no_pp = getenv("OPUS_NO_PP") != nullptr;   // A/B aids
    static const int min_tiles = getenv("OPUS_PP_MIN_TILES") ? atoi(getenv("OPUS_PP_MIN_TILES")) : 128;
    return !no_big && !no_pp && (int64_t)cdiv(M, 256) * cdiv(N, 256) >= min_tiles;
}
bool gemm_goes_pp(int M, int N) { return M > 96 && pp_min_tiles_ok(M, N); }

static hipError_t launch_gemm_(const GemmParams &p, hipStream_t s, int *klass) {
    if (p.ln_stat && !gemm_goes_pp(p.M, p.N)) return hipErrorInvalidValue;   // only gemm_pp_kernel applies the fused LayerNorm
    if (p.M <= 0 || p.N <= 0 || p.K <= 0 || (p.K & 63) || (p.lda & 7)) return hipErrorInvalidValue;
    if (p.epi == EPI_SILU_GU16 && (p.N & 31)) return hipErrorInvalidValue;
    const bool skinny = p.M <= SKINNY_MAX_M;
    static const bool no_mid = getenv("OPUS_NO_MID_GEMM") != nullptr;   // A/B aid
    const bool mid = !skinny && p.M <= MID_MAX_M && !no_mid;
    static const bool mid_v1 = getenv("OPUS_MID_V1") != nullptr;        // A/B aid: 4-panel kernel for every mid shape
    if (klass) *klass = skinny ? KC_SKINNY : KC_TILE;   // (refined by the launch itself in timing mode: LaunchEvents::main_class)
    if (p.Af && !skinny && !mid) return hipErrorInvalidValue;   // fused norm: skinny and mid kernels only
    if (p.Af && p.epi == EPI_GELU) return hipErrorInvalidValue;
    // the fragment-ordered activation layout: read by gemm_stream_kernel only; written (fp16 C) by it and by gemm_wide_kernel
    const bool to_stream = mid && !p.Af && p.N < 16384 && gemm_stream_ok(p);
    const bool to_wide = !to_stream && mid && !p.Af && !mid_v1 && (p.N >= 16384 || p.force_wide || narrow_wide());
    if (p.a_tiled && !to_stream) return hipErrorInvalidValue;
    if (p.c_tiled && (p.out_f32 || !(to_stream || to_wide) || ((p.epi == EPI_SILU_GU16 ? p.N / 2 : p.N) & 63))) return hipErrorInvalidValue;
    if (p.row_ssq && skinny) return hipErrorInvalidValue;         // no row scale in the skinny kernel (fused norm instead)
    if (skinny) {
        if (p.Af) {
            switch (p.epi) {
                case EPI_NONE: return launch_skinny_e<EPI_NONE, true>(p, s);
                case EPI_SILU_GU16: return launch_skinny_e<EPI_SILU_GU16, true>(p, s);
            }
            return hipErrorInvalidValue;
        }
        switch (p.epi) {
            case EPI_NONE: return launch_skinny_e<EPI_NONE, false>(p, s);
            case EPI_GELU: return launch_skinny_e<EPI_GELU, false>(p, s);
            case EPI_SILU_GU16: return launch_skinny_e<EPI_SILU_GU16, false>(p, s);
        }
    } else if (mid && !p.Af && !mid_v1 && p.N < 16384 && gemm_stream_ok(p)) {
        // narrow outputs at 5..64 rows whose panels map onto the CUs (wo, down, QKV with two k-parts): one launch, K cut over
        // the waves of a workgroup, no slabs / no reduce launch (gemm_stream.hip)
        return launch_gemm_stream(p, s);
    } else if (mid && !p.Af && !mid_v1 && (p.N >= 16384 || p.force_wide || narrow_wide())) {
        // wide outputs (wgu, lm_head): one barrier per 512 k, weights through per-wave register rings
        const bool m2 = p.M <= 32;
        switch (p.epi) {
            case EPI_NONE: return m2 ? launch_wide<2, EPI_NONE>(p, s) : launch_wide<4, EPI_NONE>(p, s);
            case EPI_GELU: return m2 ? launch_wide<2, EPI_GELU>(p, s) : launch_wide<4, EPI_GELU>(p, s);
            case EPI_SILU_GU16: return m2 ? launch_wide<2, EPI_SILU_GU16>(p, s) : launch_wide<4, EPI_SILU_GU16>(p, s);
        }
    } else if (mid && !p.Af && !mid_v1 && p.M > 32) {
        // narrow outputs at 33..64 rows (wo, wd): the LDS-DMA ring with a 128 x 128 tile and k-parts
        switch (p.epi) {
            case EPI_NONE: return launch_ring<2, 2, 8, EPI_NONE>(p, s, true);
            case EPI_GELU: return launch_ring<2, 2, 8, EPI_GELU>(p, s, true);
            case EPI_SILU_GU16: return launch_ring<2, 2, 8, EPI_SILU_GU16>(p, s, true);
        }
    } else if (mid) {
        if (p.Af) {
            switch (p.epi) {
                case EPI_NONE: return launch_mid_e<EPI_NONE, true>(p, s);
                case EPI_SILU_GU16: return launch_mid_e<EPI_SILU_GU16, true>(p, s);
            }
            return hipErrorInvalidValue;
        }
        switch (p.epi) {
            case EPI_NONE: return launch_mid_e<EPI_NONE, false>(p, s);
            case EPI_GELU: return launch_mid_e<EPI_GELU, false>(p, s);
            case EPI_SILU_GU16: return launch_mid_e<EPI_SILU_GU16, false>(p, s);
        }
    } else if (!p.Af && p.M <= 96 && p.N >= 16384 && !mid_v1) {
        // 65..96 rows and a wide output (gate/up of the B = 1 prefill): still a weight stream; six row tiles per wave
        // keep it one pass over the weights (47 vs 62 us for the split-K tile kernel; narrow outputs measured slower)
        switch (p.epi) {
            case EPI_NONE: return launch_wide<6, EPI_NONE>(p, s);
            case EPI_GELU: return launch_wide<6, EPI_GELU>(p, s);
            case EPI_SILU_GU16: return launch_wide<6, EPI_SILU_GU16>(p, s);
        }
    } else {
        switch (p.epi) {
            case EPI_NONE: return launch_tile_e<EPI_NONE>(p, s);
            case EPI_GELU: return launch_tile_e<EPI_GELU>(p, s);
            case EPI_SILU_GU16: return launch_tile_e<EPI_SILU_GU16>(p, s);
        }
    }
    return hipErrorInvalidValue;
}

}  // namespace opus
