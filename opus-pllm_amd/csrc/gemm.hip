// GEMM kernels for gfx950: C = epi(A W^T + bias) (+ residual), fp16 operands, fp32 accumulation.
//
//  * gemm_skinny (M <= 64): weight-streaming kernel for decode / projector / small prefill.  HBM-bound
//    by construction: every weight byte is loaded exactly once, straight from HBM to registers as
//    the MFMA B operand (16 B per lane, 128 contiguous bytes per weight row per load pair), K is
//    split over the waves of a workgroup and combined through LDS.  No LDS staging of weights: they
//    are not shared between waves (cdna_hip_programming.md "GEMV / M<=16 decode weights" row).
//  * gemm_tile (M > 64): 128x128x64 LDS-tiled MFMA kernel, register-staged double buffering,
//    XOR-swizzled LDS (conflict-free ds_read_b128), 4 waves x (64x64) accumulators.
#include "common.h"

namespace opus {

__device__ __forceinline__ float gelu_erf(float x) { return 0.5f * x * (1.0f + erff(x * 0.70710678118654752f)); }
__device__ __forceinline__ float silu(float x) { return x / (1.0f + __expf(-x)); }

// ------------------------------------------------------------------------------------------------
// skinny: one workgroup = W waves, NT*16 output columns, all M (<= 16*MT) rows; wave w streams the
// k-chunks [c0,c1) of 64 and the partial 16x16 tiles are summed through LDS.
// MFMA 16x16x32 f16: A lane (m = l&15, g = l>>4) holds x[m][kslot 8g..8g+7]; B lane holds
// W[n = l&15][same kslots]; C lane holds C[m = 4g + r][n = l&15].  The k order inside an MFMA is a
// free permutation as long as A and B agree: each lane takes 16 consecutive k (32 B) per 64-chunk
// and feeds halves to two MFMAs, so a row's four lane-groups cover one full 128-B line.
template <int MT, int NT, int EPI>
__global__ __launch_bounds__(1024) void gemm_skinny_kernel(GemmParams p) {
    extern __shared__ __attribute__((aligned(16))) float red[];
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const int nwaves = blockDim.x >> 6;
    const int n0 = blockIdx.x * (16 * NT);
    const int chunks = p.K >> 6;
    const int c0 = (int)((int64_t)chunks * wave / nwaves);
    const int c1 = (int)((int64_t)chunks * (wave + 1) / nwaves);
    const int g = lane >> 4, li = lane & 15;

    const half_t *wrow[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        int n = n0 + 16 * t + li;
        n = n < p.N ? n : p.N - 1;
        wrow[t] = p.W + (int64_t)n * p.K + g * 16;
    }
    const half_t *arow[MT];
#pragma unroll
    for (int i = 0; i < MT; ++i) {
        int m = 16 * i + li;
        m = m < p.M ? m : p.M - 1;
        arow[i] = p.A + (int64_t)m * p.lda + g * 16;
    }
    f4 acc[MT][NT];
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int t = 0; t < NT; ++t) acc[i][t] = f4{0.f, 0.f, 0.f, 0.f};

    // chunks in flight per wave: 2*U*NT weight loads of 16 B per lane (bounded by the 128-VGPR budget
    // of a 16-wave workgroup)
    constexpr int U = MT == 1 ? 4 : (MT == 2 ? 2 : 1);
    int c = c0;
    for (; c + U <= c1; c += U) {
        h8 wl[U][NT], wh[U][NT], al[U][MT], ah[U][MT];
#pragma unroll
        for (int u = 0; u < U; ++u) {
#pragma unroll
            for (int t = 0; t < NT; ++t) {
                const h8 *ptr = reinterpret_cast<const h8 *>(wrow[t] + (int64_t)(c + u) * 64);
                wl[u][t] = __builtin_nontemporal_load(ptr);
                wh[u][t] = __builtin_nontemporal_load(ptr + 1);
            }
#pragma unroll
            for (int i = 0; i < MT; ++i) {
                const h8 *ptr = reinterpret_cast<const h8 *>(arow[i] + (int64_t)(c + u) * 64);
                al[u][i] = ptr[0];
                ah[u][i] = ptr[1];
            }
        }
#pragma unroll
        for (int u = 0; u < U; ++u)
#pragma unroll
            for (int i = 0; i < MT; ++i)
#pragma unroll
                for (int t = 0; t < NT; ++t) {
                    acc[i][t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(al[u][i], wl[u][t], acc[i][t], 0, 0, 0);
                    acc[i][t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[u][i], wh[u][t], acc[i][t], 0, 0, 0);
                }
    }
    for (; c < c1; ++c) {
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            const h8 *wp = reinterpret_cast<const h8 *>(wrow[t] + (int64_t)c * 64);
            h8 wl = __builtin_nontemporal_load(wp), wh = __builtin_nontemporal_load(wp + 1);
#pragma unroll
            for (int i = 0; i < MT; ++i) {
                const h8 *ap = reinterpret_cast<const h8 *>(arow[i] + (int64_t)c * 64);
                acc[i][t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ap[0], wl, acc[i][t], 0, 0, 0);
                acc[i][t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ap[1], wh, acc[i][t], 0, 0, 0);
            }
        }
    }

    // cross-wave K reduction through LDS: red[wave][i][t][r][lane]
    if (nwaves > 1) {
#pragma unroll
        for (int i = 0; i < MT; ++i)
#pragma unroll
            for (int t = 0; t < NT; ++t)
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    red[(((wave * MT + i) * NT + t) * 4 + r) * 64 + lane] = acc[i][t][r];
        __syncthreads();
        if (wave != 0) return;
#pragma unroll
        for (int i = 0; i < MT; ++i)
#pragma unroll
            for (int t = 0; t < NT; ++t)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    float s = acc[i][t][r];
                    for (int w = 1; w < nwaves; ++w) s += red[(((w * MT + i) * NT + t) * 4 + r) * 64 + lane];
                    acc[i][t][r] = s;
                }
    }

    // epilogue: lane holds C[m = 16i + 4g + r][n = n0 + 16t + li]
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int m = 16 * i + 4 * g + r;
            if (m >= p.M) continue;
            if (EPI == EPI_SILU_GU16) {
                static_assert(EPI != EPI_SILU_GU16 || NT == 2, "gate/up pairing needs NT == 2");
                const int ng = n0 + li;
                if (ng >= p.N) continue;
                float gate = acc[i][0][r], up = acc[i][NT - 1][r];
                if (p.bias) { gate += p.bias[ng]; up += p.bias[ng + 16]; }
                float v = silu(gate) * up;
                const int no = (n0 >> 1) + li;
                if (p.residual) v += p.residual[(int64_t)m * p.ldr + no];
                if (p.out_f32) reinterpret_cast<float *>(p.C)[(int64_t)m * p.ldc + no] = v;
                else reinterpret_cast<half_t *>(p.C)[(int64_t)m * p.ldc + no] = (half_t)v;
            } else {
#pragma unroll
                for (int t = 0; t < NT; ++t) {
                    const int n = n0 + 16 * t + li;
                    if (n >= p.N) continue;
                    float v = acc[i][t][r];
                    if (p.bias) v += p.bias[n];
                    if (EPI == EPI_GELU) v = gelu_erf(v);
                    if (p.residual) v += p.residual[(int64_t)m * p.ldr + n];
                    if (p.out_f32) reinterpret_cast<float *>(p.C)[(int64_t)m * p.ldc + n] = v;
                    else reinterpret_cast<half_t *>(p.C)[(int64_t)m * p.ldc + n] = (half_t)v;
                }
            }
        }
}

// ------------------------------------------------------------------------------------------------
// tile: 128 x 128 x 64, 256 threads = 2x2 waves of 64x64 (4x4 MFMA 16x16x32 tiles each).
// LDS image per operand: [128 rows][8 chunks of 16 B], chunk c of row r stored at c ^ ((r>>1)&7)
// (conflict-free for the 16-lane groups of ds_read_b128, see MI355X_MICROARCH.md LDS table).
constexpr int TBM = 128, TBN = 128, TBK = 64;

__device__ __forceinline__ int swz(int row, int chunk) { return chunk ^ ((row >> 1) & 7); }

template <int EPI>
__global__ __launch_bounds__(256) void gemm_tile_kernel(GemmParams p, int tiles_m, int tiles_n) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    // [buf][A|B][128][64] halfs : 2 * 2 * 16 KB
    h8 *lds = reinterpret_cast<h8 *>(smem);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wr = wave >> 1, wc = wave & 1;
    // XCD-aware tile order: consecutive workgroup ids land on different XCDs (round-robin), so give
    // each XCD a contiguous run of tiles that share the same A row-panel in its private L2.
    int bid = blockIdx.x;
    {
        const int nwg = gridDim.x, q = nwg >> 3, rr = nwg & 7, xcd = bid & 7;
        bid = (xcd < rr ? xcd * (q + 1) : rr * (q + 1) + (xcd - rr) * q) + (bid >> 3);
    }
    const int tm = bid / tiles_n, tn = bid % tiles_n;
    const int m0 = tm * TBM, n0 = tn * TBN;

    // staging: thread handles rows (tid>>3) + 32*i, chunk tid&7
    const int srow = tid >> 3, schunk = tid & 7;
    const half_t *ag[4], *bg[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        int ra = m0 + srow + 32 * i;
        ra = ra < p.M ? ra : p.M - 1;
        int rb = n0 + srow + 32 * i;
        rb = rb < p.N ? rb : p.N - 1;
        ag[i] = p.A + (int64_t)ra * p.lda + schunk * 8;
        bg[i] = p.W + (int64_t)rb * p.K + schunk * 8;
    }
    h8 sa[4], sb[4];
    auto gload = [&](int kt) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            sa[i] = *reinterpret_cast<const h8 *>(ag[i] + (int64_t)kt * TBK);
            sb[i] = *reinterpret_cast<const h8 *>(bg[i] + (int64_t)kt * TBK);
        }
    };
    auto lstore = [&](int buf) {
        h8 *A = lds + buf * 2048, *B = A + 1024;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int r = srow + 32 * i;
            A[r * 8 + swz(r, schunk)] = sa[i];
            B[r * 8 + swz(r, schunk)] = sb[i];
        }
    };

    f4 acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = f4{0.f, 0.f, 0.f, 0.f};

    const int KT = p.K / TBK;
    const int g = lane >> 4, li = lane & 15;
    gload(0);
    lstore(0);
    __syncthreads();
    for (int kt = 0; kt < KT; ++kt) {
        const int buf = kt & 1;
        if (kt + 1 < KT) gload(kt + 1);
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
            for (int j = 0; j < 4; ++j) {
                const int r = wc * 64 + j * 16 + li;
                bf[j] = B[r * 8 + swz(r, 4 * s + g)];
            }
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(af[i], bf[j], acc[i][j], 0, 0, 0);
        }
        if (kt + 1 < KT) lstore(buf ^ 1);
        __syncthreads();
    }

    // epilogue
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int m = m0 + wr * 64 + i * 16 + 4 * g + r;
            if (m >= p.M) continue;
            if (EPI == EPI_SILU_GU16) {
#pragma unroll
                for (int jj = 0; jj < 2; ++jj) {
                    const int ng = n0 + wc * 64 + jj * 32 + li;
                    if (ng >= p.N) continue;
                    float gate = acc[i][2 * jj][r], up = acc[i][2 * jj + 1][r];
                    if (p.bias) { gate += p.bias[ng]; up += p.bias[ng + 16]; }
                    float v = silu(gate) * up;
                    const int no = ((n0 + wc * 64 + jj * 32) >> 1) + li;
                    if (p.residual) v += p.residual[(int64_t)m * p.ldr + no];
                    if (p.out_f32) reinterpret_cast<float *>(p.C)[(int64_t)m * p.ldc + no] = v;
                    else reinterpret_cast<half_t *>(p.C)[(int64_t)m * p.ldc + no] = (half_t)v;
                }
            } else {
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int n = n0 + wc * 64 + j * 16 + li;
                    if (n >= p.N) continue;
                    float v = acc[i][j][r];
                    if (p.bias) v += p.bias[n];
                    if (EPI == EPI_GELU) v = gelu_erf(v);
                    if (p.residual) v += p.residual[(int64_t)m * p.ldr + n];
                    if (p.out_f32) reinterpret_cast<float *>(p.C)[(int64_t)m * p.ldc + n] = v;
                    else reinterpret_cast<half_t *>(p.C)[(int64_t)m * p.ldc + n] = (half_t)v;
                }
            }
        }
}

// ------------------------------------------------------------------------------------------------
template <int MT, int NT, int EPI>
static hipError_t launch_skinny_t(const GemmParams &p, hipStream_t s) {
    const int groups = cdiv(p.N, 16 * NT);
    const int chunks = p.K / 64;
    // enough waves in flight to cover HBM latency: aim for >= 16 waves per CU (4096 on 256 CUs)
    int W = cdiv(4096, groups);
    W = W < 4 ? 4 : (W > 16 ? 16 : W);
    const int wcap = 64 / (MT * NT);   // keep the LDS reduction buffer <= 64 KB
    if (W > wcap) W = wcap;
    if (W > chunks) W = chunks;
    if (W < 1) W = 1;
    const size_t lds = W > 1 ? (size_t)W * MT * NT * 4 * 64 * sizeof(float) : 0;
    hipLaunchKernelGGL((gemm_skinny_kernel<MT, NT, EPI>), dim3(groups), dim3(64 * W), lds, s, p);
    return hipGetLastError();
}

template <int EPI>
static hipError_t launch_skinny_e(const GemmParams &p, hipStream_t s) {
    const int mt = cdiv(p.M, 16);
    constexpr int NT = (EPI == EPI_SILU_GU16) ? 2 : 1;
    switch (mt) {
        case 1: return launch_skinny_t<1, NT, EPI>(p, s);
        case 2: return launch_skinny_t<2, NT, EPI>(p, s);
        case 3:
        case 4: return launch_skinny_t<4, NT, EPI>(p, s);
    }
    return hipErrorInvalidValue;
}

template <int EPI>
static hipError_t launch_tile_e(const GemmParams &p, hipStream_t s) {
    const int tm = cdiv(p.M, TBM), tn = cdiv(p.N, TBN);
    static bool attr_set = false;
    if (!attr_set) {
        hipError_t ea = hipFuncSetAttribute(reinterpret_cast<const void *>(&gemm_tile_kernel<EPI>),
                                            hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
        if (ea != hipSuccess) return ea;
        attr_set = true;
    }
    hipLaunchKernelGGL((gemm_tile_kernel<EPI>), dim3(tm * tn), dim3(256), 65536, s, p, tm, tn);
    return hipGetLastError();
}

hipError_t launch_gemm(const GemmParams &p, hipStream_t s, int *klass) {
    if (p.M <= 0 || p.N <= 0 || p.K <= 0 || (p.K & 63) || (p.lda & 7)) return hipErrorInvalidValue;
    if (p.epi == EPI_SILU_GU16 && (p.N & 31)) return hipErrorInvalidValue;
    const bool skinny = p.M <= 64;
    if (klass) *klass = skinny ? KC_SKINNY : KC_TILE;
    if (skinny) {
        switch (p.epi) {
            case EPI_NONE: return launch_skinny_e<EPI_NONE>(p, s);
            case EPI_GELU: return launch_skinny_e<EPI_GELU>(p, s);
            case EPI_SILU_GU16: return launch_skinny_e<EPI_SILU_GU16>(p, s);
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
