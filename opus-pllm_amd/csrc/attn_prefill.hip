// Flash-style attention for the encoder (bidirectional, key-padding) and the decoder prefill
// (causal, left-padded rows, GQA).  O(T) memory: the reference stack materialises the [h,T,T] scores.
//
// Workgroup = 4 waves = 64 query rows (16 per wave) of one (batch, head); K/V tiles of 64 keys go
// through LDS (K row-major, swizzled for conflict-free ds_read_b128; V transposed so the PV product
// reads 8 consecutive keys per lane).  Per wave and tile:
//   S  = Q K^T        MFMA 16x16x32 f16 : A = Q rows (registers), B = K rows (LDS)
//   online softmax    rows live on (lane>>4, reg), keys on lane&15 -> 4 xor-shuffles per reduction
//   O += P V          P (C layout) -> per-wave LDS patch -> A layout; B = V^T rows (LDS)
#include "common.h"

namespace opus {

constexpr int QB = 64;   // queries per workgroup
constexpr int KB = 64;   // keys per tile
constexpr int VPAD = 8;  // halfs of padding on V^T / P rows (keeps 16-B alignment, spreads banks)

template <int HD>
__device__ __forceinline__ int kswz(int row, int chunk) {
    if (HD == 128) return chunk ^ (row & 15);
    if (HD == 64) return chunk ^ ((row >> 1) & 7);
    return chunk;
}

template <int HD, bool CAUSAL>
__global__ __launch_bounds__(256) void attn_prefill_kernel(AttnParams p) {
    constexpr int HDP = HD < 32 ? 32 : HD;      // QK^T k extent (zero-padded for head_dim 16)
    constexpr int KS = HDP / 32;                // MFMA k-steps for QK^T
    constexpr int NO = HD / 16;                 // output column tiles
    constexpr int CH = HDP / 8;                 // 16-B chunks per K row
    __shared__ __attribute__((aligned(16))) half_t sK[KB * HDP];
    __shared__ __attribute__((aligned(16))) half_t sVt[HD * (KB + VPAD)];
    __shared__ __attribute__((aligned(16))) half_t sP[4 * 16 * (KB + VPAD)];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int g = lane >> 4, li = lane & 15;
    const int b = blockIdx.z, h = blockIdx.y, hk = h / p.group;
    const int q0 = blockIdx.x * QB;
    const int kstart = p.kstart ? p.kstart[b] : 0;
    const int kend = p.kend ? p.kend[b] : p.T;

    const half_t *Qb = p.Q + (int64_t)b * p.q_sb + (int64_t)h * HD;
    const half_t *Kb = p.K + (int64_t)b * p.k_sb + (int64_t)hk * HD;
    const half_t *Vb = p.V + (int64_t)b * p.v_sb + (int64_t)hk * HD;

    // Q fragments: A operand rows = this wave's 16 queries
    h8 qf[KS];
    {
        int qr = q0 + wave * 16 + li;
        qr = qr < p.T ? qr : p.T - 1;
        const half_t *src = Qb + (int64_t)qr * p.q_st;
#pragma unroll
        for (int s = 0; s < KS; ++s) {
            const int d = 32 * s + 8 * g;
            if (d < HD) qf[s] = *reinterpret_cast<const h8 *>(src + d);
            else qf[s] = h8{0, 0, 0, 0, 0, 0, 0, 0};
        }
    }
    f4 o[NO];
#pragma unroll
    for (int n = 0; n < NO; ++n) o[n] = f4{0.f, 0.f, 0.f, 0.f};
    float mrow[4], lrow[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) { mrow[r] = -INFINITY; lrow[r] = 0.f; }

    int k_lo = kstart / KB * KB;
    int k_hi = kend;
    if (CAUSAL) {
        const int last_q = (q0 + QB - 1 < p.T - 1 ? q0 + QB - 1 : p.T - 1);
        k_hi = k_hi < last_q + 1 ? k_hi : last_q + 1;
    }
    half_t *myP = sP + wave * 16 * (KB + VPAD);
    const float sc = p.scale * 1.4426950408889634f;   // softmax in base 2

    // K/V tiles are double-buffered through registers: the global loads of tile t+1 are issued right after tile t has
    // been copied into LDS, so their latency runs under the MFMA / softmax work of tile t.
    constexpr int KL = KB * CH / 256;            // 16-B pieces of K per thread and tile
    constexpr int VN = KB * (HD / 8);            // 16-B pieces of V per tile (128 for head_dim 16: half the threads idle)
    constexpr int VL = VN >= 256 ? VN / 256 : 1;
    static_assert(KL >= 1, "tile too small for 256 threads");
    h8 kreg[KL], vreg[VL];
    auto load_tile = [&](int kt) {
#pragma unroll
        for (int u = 0; u < KL; ++u) {
            const int i = tid + 256 * u, r = i / CH, c = i % CH;
            int kr = kt + r;
            kr = kr < p.T ? kr : p.T - 1;
            kreg[u] = h8{0, 0, 0, 0, 0, 0, 0, 0};
            if (c * 8 < HD) kreg[u] = *reinterpret_cast<const h8 *>(Kb + (int64_t)kr * p.k_st + c * 8);
        }
#pragma unroll
        for (int u = 0; u < VL; ++u) {
            const int i = tid + 256 * u, r = i / (HD / 8), c = i % (HD / 8);
            int kr = kt + r;
            kr = kr < p.T ? kr : p.T - 1;
            vreg[u] = h8{0, 0, 0, 0, 0, 0, 0, 0};
            if (i < VN) vreg[u] = *reinterpret_cast<const h8 *>(Vb + (int64_t)kr * p.v_st + c * 8);
        }
    };
    auto store_tile = [&]() {
#pragma unroll
        for (int u = 0; u < KL; ++u) {
            const int i = tid + 256 * u, r = i / CH, c = i % CH;
            *reinterpret_cast<h8 *>(sK + r * HDP + kswz<HDP>(r, c) * 8) = kreg[u];
        }
        // V^T: lanes (r, c) and (r^1, c) sit HD/8 lanes apart; they swap halves of their 8 dims so that each writes
        // four {key r&~1, key r|1} pairs as 32-bit words instead of eight 16-bit ones
#pragma unroll
        for (int u = 0; u < VL; ++u) {
            const int i = tid + 256 * u, r = i / (HD / 8), c = i % (HD / 8);
            typedef unsigned int u4v __attribute__((ext_vector_type(4)));
            const u4v mine = __builtin_bit_cast(u4v, vreg[u]);
            const bool odd = r & 1;
            // send the half the partner writes, keep the half this lane writes (even rows: dims 0..3, odd rows: 4..7)
            const unsigned s0 = odd ? mine[0] : mine[2], s1 = odd ? mine[1] : mine[3];
            const unsigned k0 = odd ? mine[2] : mine[0], k1 = odd ? mine[3] : mine[1];
            const unsigned o0 = __shfl_xor(s0, HD / 8, 64), o1 = __shfl_xor(s1, HD / 8, 64);
            // k0/k1: this lane's key, two dims per word; o0/o1: the partner key, same dims
            const unsigned lo0 = odd ? o0 : k0, hi0 = odd ? k0 : o0;     // even key in the low half of each output word
            const unsigned lo1 = odd ? o1 : k1, hi1 = odd ? k1 : o1;
            const int d0 = c * 8 + (odd ? 4 : 0), re = r & ~1;
            unsigned *dst = reinterpret_cast<unsigned *>(sVt);
            constexpr int RW = (KB + VPAD) / 2;                           // words per V^T row
            if (i >= VN) continue;
            dst[(d0 + 0) * RW + (re >> 1)] = (lo0 & 0xffffu) | (hi0 << 16);
            dst[(d0 + 1) * RW + (re >> 1)] = (lo0 >> 16) | (hi0 & 0xffff0000u);
            dst[(d0 + 2) * RW + (re >> 1)] = (lo1 & 0xffffu) | (hi1 << 16);
            dst[(d0 + 3) * RW + (re >> 1)] = (lo1 >> 16) | (hi1 & 0xffff0000u);
        }
    };

    if (k_lo < k_hi) load_tile(k_lo);
    for (int kt = k_lo; kt < k_hi; kt += KB) {
        __syncthreads();   // previous tile fully consumed
        store_tile();
        if (kt + KB < k_hi) load_tile(kt + KB);
        __syncthreads();

        // ---- S = Q K^T : 4 column tiles of 16 keys ----
        f4 s[4];
#pragma unroll
        for (int n = 0; n < 4; ++n) {
            s[n] = f4{0.f, 0.f, 0.f, 0.f};
            const int r = 16 * n + li;
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
                const h8 kf = *reinterpret_cast<const h8 *>(sK + r * HDP + kswz<HDP>(r, 4 * ks + g) * 8);
                s[n] = __builtin_amdgcn_mfma_f32_16x16x32_f16(qf[ks], kf, s[n], 0, 0, 0);
            }
        }
        // ---- mask + online softmax (row = 4g + r, key = kt + 16n + li) ----
        // interior: every key of the tile is visible to every query row of this workgroup (block-uniform)
        const bool interior = kt >= kstart && kt + KB <= kend && (!CAUSAL || kt + KB - 1 <= q0);
        float alpha[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int qi = q0 + wave * 16 + 4 * g + r;
            float mx = -INFINITY;
#pragma unroll
            for (int n = 0; n < 4; ++n) {
                float v = s[n][r] * sc;
                if (!interior) {                                     // tiles that touch a padding / causal boundary
                    const int kj = kt + 16 * n + li;
                    bool vis = kj >= kstart && kj < kend;
                    if (CAUSAL) vis = vis && kj <= qi;
                    v = vis ? v : -INFINITY;
                }
                s[n][r] = v;
                mx = fmaxf(mx, v);
            }
#pragma unroll
            for (int off = 1; off < 16; off <<= 1) mx = fmaxf(mx, __shfl_xor(mx, off, 64));
            const float mnew = fmaxf(mrow[r], mx);
            const float msafe = mnew == -INFINITY ? 0.f : mnew;
            alpha[r] = exp2f(mrow[r] - msafe);          // 0 when mrow = -inf
            float rs = 0.f;
#pragma unroll
            for (int n = 0; n < 4; ++n) {
                const float e = exp2f(s[n][r] - msafe);
                s[n][r] = e;
                rs += e;
            }
#pragma unroll
            for (int off = 1; off < 16; off <<= 1) rs += __shfl_xor(rs, off, 64);
            lrow[r] = lrow[r] * alpha[r] + rs;
            mrow[r] = mnew;
        }
#pragma unroll
        for (int n = 0; n < NO; ++n)
#pragma unroll
            for (int r = 0; r < 4; ++r) o[n][r] *= alpha[r];
        // ---- P: C layout -> LDS patch [16 q][64 keys] -> A layout ----
#pragma unroll
        for (int n = 0; n < 4; ++n)
#pragma unroll
            for (int r = 0; r < 4; ++r) myP[(4 * g + r) * (KB + VPAD) + 16 * n + li] = (half_t)s[n][r];
        // same-wave LDS round trip: the wave's own ds_write -> ds_read ordering is kept by hardware
        // (in-order LDS queue); the compiler fence stops reordering of the accesses.
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_s_waitcnt(0xc07f);
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        h8 pf[2];
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
            pf[ks] = *reinterpret_cast<const h8 *>(myP + li * (KB + VPAD) + 32 * ks + 8 * g);
        // ---- O += P V ----
#pragma unroll
        for (int n = 0; n < NO; ++n)
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                const h8 vf = *reinterpret_cast<const h8 *>(sVt + (16 * n + li) * (KB + VPAD) + 32 * ks + 8 * g);
                o[n] = __builtin_amdgcn_mfma_f32_16x16x32_f16(pf[ks], vf, o[n], 0, 0, 0);
            }
    }

    // ---- epilogue: O / l -> fp16 [b, t, h*HD + d] ----
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int qi = q0 + wave * 16 + 4 * g + r;
        if (qi >= p.T) continue;
        const float inv = lrow[r] > 0.f ? 1.0f / lrow[r] : 0.f;
        half_t *dst = p.O + (int64_t)b * p.o_sb + (int64_t)qi * p.o_st + (int64_t)h * HD;
#pragma unroll
        for (int n = 0; n < NO; ++n) dst[16 * n + li] = (half_t)(o[n][r] * inv);
    }
}

template <int HD>
static hipError_t launch_hd(const AttnParams &p, hipStream_t s) {
    dim3 grid(cdiv(p.T, QB), p.heads, p.B);
    if (p.causal) OPUS_LAUNCH(KC_ATTN_PREFILL, (attn_prefill_kernel<HD, true>), grid, dim3(256), 0, s, p);
    else OPUS_LAUNCH(KC_ATTN_PREFILL, (attn_prefill_kernel<HD, false>), grid, dim3(256), 0, s, p);
    return hipGetLastError();
}

hipError_t launch_attn_prefill(const AttnParams &p, hipStream_t s) {
    if (p.T <= 0 || p.B <= 0) return hipErrorInvalidValue;
    switch (p.head_dim) {
        case 16: return launch_hd<16>(p, s);
        case 32: return launch_hd<32>(p, s);
        case 64: return launch_hd<64>(p, s);
        case 128: return launch_hd<128>(p, s);
    }
    return hipErrorInvalidValue;
}

}  // namespace opus
