// Flash-style attention for the encoder (bidirectional, key-padding) and the decoder prefill (causal, left-padded rows,
// GQA).  O(T) memory: the reference stack materialises the [h,T,T] scores.
//
// Workgroup = 4 waves = 64 QT query rows (QT 16-query tiles per wave) of one (batch, head); K/V tiles of 64 keys are
// double-buffered in LDS (global -> registers one tile ahead, written after the tile's compute: one barrier per tile).
// The products are "swapped" so that a QUERY lives on a lane and the softmax never crosses lanes inside a tile:
//   S^T = K Q^T      MFMA 16x16x32 f16 : A = 16 keys x 32 dims (LDS, XOR-swizzled rows), B = Q^T (registers).  The result
//                    lane (query = lane & 15, g = lane >> 4) holds the scores of ITS query for keys 16n + 4g + r.
//   online softmax   16 scores per lane and tile; the row maximum needs two xor-shuffles (over g), the row sum none until
//                    the very end; the running rescale factor is a per-lane scalar.
//   O^T += V^T P^T   the fp16-rounded probabilities ARE the B operand of the next MFMA as they stand (the contraction index
//                    of a 32-key step is the permutation key(g, e) = 16 (2j + e/4) + 4g + e%4, cdna_hip_programming.md
//                    "An accumulator tile as the next MFMA's operand"); the A operand V^T[dim][key(g, e)] is read from the
//                    row-major V tile with the transposing LDS read ds_read_b64_tr_b16 (two per fragment) - no transposing
//                    store, no probability patch in LDS.
// Per tile and wave: 32 MFMAs, 8 ds_read_b128 + 16 ds_read_b64_tr (head_dim 64), no LDS writes besides the tile staging.
#include "common.h"
#include <type_traits>

namespace opus {

constexpr int KB = 64;          // keys per tile

template <int HDP>
__device__ __forceinline__ int kswz(int row, int chunk) {
    if (HDP == 128) return chunk ^ (row & 15);
    if (HDP == 64) return chunk ^ ((row >> 1) & 7);
    return chunk;
}
// V tile: row-major [key][HD], 16-B chunk c of row r stored at c ^ vswz(r): the 8 consecutive rows that the two 16-lane
// groups of a half wave read with one ds_read_b64_tr_b16 then fall on 64 distinct banks
template <int HD>
__device__ __forceinline__ int vswz(int row) {
    if (HD == 128) return (row & 7) << 1;            // 256-B rows: every row starts on bank 0
    if (HD == 64) return ((row >> 1) & 3) << 1;      // 128-B rows: rows r, r + 2 collide
    if (HD == 32) return ((row >> 2) & 1) << 1;      //  64-B rows: rows r, r + 4 collide
    return 0;                                        //  32-B rows: 8 rows = one bank row
}

typedef float f2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ float max3(float a, float b, float c) {
    float r;
    asm("v_max3_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
    return r;
}
// maximum over the lane pairs (l, l ^ 16) and (l, l ^ 32) with gfx950's row swaps instead of ds_bpermute round trips through
// the LDS pipe: v_permlane16_swap exchanges the odd 16-lane rows of its first operand with the even rows of the second, so
// with both operands holding x the results are [x0 x0 x2 x2] and [x1 x1 x3 x3]; v_permlane32_swap does the same with halves.
// (s_nop 1: the operands come from a VALU instruction the assembler cannot see)
__device__ __forceinline__ float xor16_max(float x) {
    float a = x, b = x, r;
    asm volatile("s_nop 1\n\tv_permlane16_swap_b32 %0, %1" : "+v"(a), "+v"(b));
    asm("v_max_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
__device__ __forceinline__ void xor32_pair(float x, float &a, float &b) {
    a = x;
    b = x;
    asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1" : "+v"(a), "+v"(b));
}

typedef short s4v __attribute__((ext_vector_type(4)));
typedef short s8v __attribute__((ext_vector_type(8)));

// QT = 16-query tiles per wave (queries per wave QW = 16 QT, per workgroup QB = 64 QT).  Two tiles share every K / V fragment
// read; one tile halves the registers (two more waves per SIMD at head_dim 64) and the ragged last query block.
template <int HD, bool CAUSAL, int QT>
__global__ __launch_bounds__(256) void attn_prefill_kernel(AttnParams p) {
    constexpr int QW = 16 * QT, QB = 4 * QW;
    constexpr int HDP = HD < 32 ? 32 : HD;      // QK^T k extent (zero-padded for head_dim 16)
    constexpr int KS = HDP / 32;                // MFMA k-steps for QK^T
    constexpr int NO = HD / 16;                 // output dim tiles
    constexpr int CH = HDP / 8;                 // 16-B chunks per K row
    constexpr int VC = HD / 8;                  // 16-B chunks per V row
    __shared__ __attribute__((aligned(16))) half_t sK[2][KB * HDP];
    __shared__ __attribute__((aligned(16))) half_t sV[2][KB * HD];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int g = lane >> 4, li = lane & 15;
    // Workgroup ids are dealt round-robin over the 8 XCDs: ids that are equal mod 8 share an L2.  All query blocks of one
    // (batch, head) - which stream the same K / V - therefore take ids of one residue class: K / V come from HBM / the
    // Infinity Cache once per (batch, head) instead of once per query block.  (Placement is a speed matter only.)
    // AttnParams::q_trim (token-packed batches only): the first and the last token of every row are not queries - the last ESM-2
    // layer, whose <cls> / <eos> rows the mean-pool drops (cstp_v3/modelling.py:52-54): 512 residues are then four query blocks of
    // 128 instead of five, the fifth holding the last residue and <eos>.  Keys are untouched; the trimmed rows of O keep what they held.
    const int qtrim = (p.q_trim && p.cu) ? 1 : 0;
    const int nqb = (p.T - 2 * qtrim + QB - 1) / QB;
    const int id = blockIdx.x, slot = id >> 3;
    const int bh = (slot / nqb) * 8 + (id & 7);
    if (bh >= p.B * p.heads) return;
    const int b = bh / p.heads, h = bh % p.heads, hk = h / p.group;
    const int q0 = (slot % nqb) * QB + qtrim, qw = q0 + wave * QW;
    // token-packed batch (AttnParams::cu): this row's tokens are rows cu[b] .. cu[b + 1] - 1 of Q / K / V / O, all of them
    // visible keys; query blocks past the row's last token (the grid is sized by the longest row) leave at once
    int T = p.T, Tq = p.T;                                            // Tq: end of the query range
    int64_t qb_off = (int64_t)b * p.q_sb, kb_off = (int64_t)b * p.k_sb, vb_off = (int64_t)b * p.v_sb, ob_off = (int64_t)b * p.o_sb;
    if (p.cu) {
        const int r0 = p.cu[b];
        T = p.cu[b + 1] - r0;
        Tq = T - qtrim;
        if (q0 >= Tq) return;                                         // (uniform, before any barrier)
        qb_off = (int64_t)r0 * p.q_st; kb_off = (int64_t)r0 * p.k_st; vb_off = (int64_t)r0 * p.v_st; ob_off = (int64_t)r0 * p.o_st;
    }
    const int kstart = (p.kstart && !p.cu) ? p.kstart[b] : 0;
    const int kend = (p.kend && !p.cu) ? p.kend[b] : T;

    const half_t *Qb = p.Q + qb_off + (int64_t)h * HD;
    const half_t *Kb = p.K + kb_off + (int64_t)hk * HD;
    const half_t *Vb = p.V + vb_off + (int64_t)hk * HD;

    // Q^T fragments (B operand): column li = query, k = 32 s + 8 g + e
    h8 qf[QT][KS];
#pragma unroll
    for (int t = 0; t < QT; ++t) {
        int qr = qw + 16 * t + li;
        qr = qr < Tq ? qr : Tq - 1;
        const half_t *src = Qb + (int64_t)qr * p.q_st;
#pragma unroll
        for (int s = 0; s < KS; ++s) {
            const int d = 32 * s + 8 * g;
            qf[t][s] = d < HD ? *reinterpret_cast<const h8 *>(src + d) : h8{0, 0, 0, 0, 0, 0, 0, 0};
        }
    }
    f4 o[QT][NO];
    float mrow[QT], lrow[QT];
#pragma unroll
    for (int t = 0; t < QT; ++t) {
        mrow[t] = -INFINITY;
        lrow[t] = 0.f;
#pragma unroll
        for (int n = 0; n < NO; ++n) o[t][n] = f4{0.f, 0.f, 0.f, 0.f};
    }

    int k_lo = kstart / KB * KB;
    int k_hi = kend;
    if (CAUSAL) {
        const int last_q = (q0 + QB - 1 < T - 1 ? q0 + QB - 1 : T - 1);
        k_hi = k_hi < last_q + 1 ? k_hi : last_q + 1;
    }
    const float sc = p.scale * 1.4426950408889634f;   // softmax in base 2
    const bool wave_live = qw < Tq;                 // waves past the last query only help with the tile staging

    // K/V tiles: global -> registers one tile ahead of the LDS copy
    constexpr int KL = KB * CH / 256;            // 16-B pieces of K per thread and tile
    constexpr int VN = KB * VC;                  // 16-B pieces of V per tile (128 for head_dim 16: half the threads idle)
    constexpr int VL = VN >= 256 ? VN / 256 : 1;
    static_assert(KL >= 1, "tile too small for 256 threads");
    h8 kreg[KL], vreg[VL];
    // through buffer descriptors that end with the batch row's last key: rows >= T of the last tile read as zeros without a
    // clamp, and a tile's addresses are the thread's fixed byte offsets plus one wave-uniform term (no 64-bit address
    // arithmetic per tile: that was ~30 VALU instructions of every tile, 8 of them quarter-rate multiplies)
    const __amdgpu_buffer_rsrc_t krs = __builtin_amdgcn_make_buffer_rsrc((void *)Kb, 0, (int)(((int64_t)(T - 1) * p.k_st + HD) * 2), 0x00020000);
    const __amdgpu_buffer_rsrc_t vrs = __builtin_amdgcn_make_buffer_rsrc((void *)Vb, 0, (int)(((int64_t)(T - 1) * p.v_st + HD) * 2), 0x00020000);
    int koff[KL], voff[VL];
#pragma unroll
    for (int u = 0; u < KL; ++u) {
        const int i = tid + 256 * u, r = i / CH, c = i % CH;
        koff[u] = c * 8 < HD ? (int)((r * p.k_st + c * 8) * 2) : 0x7ffffff0;        // (zero padding of head_dim 16: out of range)
    }
#pragma unroll
    for (int u = 0; u < VL; ++u) {
        const int i = tid + 256 * u, r = i / VC, c = i % VC;
        voff[u] = i < VN ? (int)((r * p.v_st + c * 8) * 2) : 0x7ffffff0;
    }
    auto load_tile = [&](int kt) {
        const int kb = kt * (int)p.k_st * 2, vb = kt * (int)p.v_st * 2;               // (wave-uniform)
#pragma unroll
        for (int u = 0; u < KL; ++u) kreg[u] = __builtin_bit_cast(h8, __builtin_amdgcn_raw_buffer_load_b128(krs, koff[u] + kb, 0, 0));
#pragma unroll
        for (int u = 0; u < VL; ++u) vreg[u] = __builtin_bit_cast(h8, __builtin_amdgcn_raw_buffer_load_b128(vrs, voff[u] + vb, 0, 0));
    };
    auto store_tile = [&](int buf) {
#pragma unroll
        for (int u = 0; u < KL; ++u) {
            const int i = tid + 256 * u, r = i / CH, c = i % CH;
            *reinterpret_cast<h8 *>(&sK[buf][r * HDP + kswz<HDP>(r, c) * 8]) = kreg[u];
        }
#pragma unroll
        for (int u = 0; u < VL; ++u) {
            const int i = tid + 256 * u, r = i / VC, c = i % VC;
            if (i < VN) *reinterpret_cast<h8 *>(&sV[buf][r * HD + (c ^ vswz<HD>(r)) * 8]) = vreg[u];
        }
    };

    if (k_lo < k_hi) {
        load_tile(k_lo);
        store_tile(0);
        if (k_lo + KB < k_hi) load_tile(k_lo + KB);
    }
    __syncthreads();
    // the tile's compute for the first NSUB 16-key subtiles (4 = the whole tile; fewer for a short last tile: T = 514 ends
    // with a tile of 2 keys, which costs a quarter of a full one this way)
    auto tile_body = [&](auto nsub_tag, int kt, int buf) {
        {
            constexpr int NSUB = decltype(nsub_tag)::value, NJ = (NSUB + 1) / 2;
            const half_t *tK = sK[buf], *tV = sV[buf];
            // ---- S^T = K Q^T : NSUB key subtiles x QT query tiles ----
            f4 s[QT][4];
#pragma unroll
            for (int n = 0; n < NSUB; ++n) {
                const int r = 16 * n + li;
                h8 kf[KS];
#pragma unroll
                for (int ks = 0; ks < KS; ++ks) kf[ks] = *reinterpret_cast<const h8 *>(tK + r * HDP + kswz<HDP>(r, 4 * ks + g) * 8);
#pragma unroll
                for (int t = 0; t < QT; ++t) {
                    s[t][n] = f4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                    for (int ks = 0; ks < KS; ++ks) s[t][n] = mfma16(kf[ks], qf[t][ks], s[t][n]);
                }
            }
            // ---- mask + online softmax: lane (li, g) owns query qw + 16 t + li, keys kt + 16 n + 4 g + r ----
            // interior: every key of the tile is visible to every query row of this wave (wave-uniform): no masking code.
            // A boundary tile masks the scores in place and then runs the SAME softmax code (one copy of it: two copies
            // merged through ~40 register moves per tile)
            const bool interior = kt >= kstart && kt + KB <= kend && (!CAUSAL || kt + KB - 1 <= qw);
            if (__builtin_expect(!interior, 0)) {
#pragma unroll
                for (int t = 0; t < QT; ++t) {
                    const int qi = qw + 16 * t + li;
#pragma unroll
                    for (int n = 0; n < NSUB; ++n)
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            const int kj = kt + 16 * n + 4 * g + r;
                            bool vis = kj >= kstart && kj < kend;
                            if (CAUSAL) vis = vis && kj <= qi;
                            s[t][n][r] = vis ? s[t][n][r] : -INFINITY;
                        }
                }
            }
            h8 pf[QT][2];
#pragma unroll
            for (int t = 0; t < QT; ++t) {
                // row maximum of the raw scores (the scale is positive): 4 NSUB in-lane values (v_max3 chain), then the four
                // lane groups with two row swaps (v_permlane16_swap / v_permlane32_swap: no LDS round trip)
                // row maximum of the raw scores (the scale is positive): the in-lane values in C++ (the compiler knows the wait
                // states a VALU read of an MFMA result needs - an asm block reading the scores directly would not be covered by
                // its hazard recogniser; -fno-honor-nans keeps it at 13 max instructions per 16 scores), then the four lane groups
                // with two row swaps
                float mx = fmaxf(fmaxf(s[t][0][0], s[t][0][1]), fmaxf(s[t][0][2], s[t][0][3]));
#pragma unroll
                for (int n = 1; n < NSUB; ++n) mx = fmaxf(fmaxf(mx, fmaxf(s[t][n][0], s[t][n][1])), fmaxf(s[t][n][2], s[t][n][3]));
                float ma, mb;
                mx = xor16_max(mx);
                xor32_pair(mx, ma, mb);
                const float mnew = max3(mrow[t], ma, mb);
                const float msafe = mnew == -INFINITY ? 0.f : mnew;       // (rows with no visible key so far)
                const float msc = msafe * sc;
                if (!__all(mnew == mrow[t])) {                           // (wave-uniform) some row's maximum grew: rescale
                    const float alpha = __builtin_amdgcn_exp2f((mrow[t] - msafe) * sc);   // 0 when mrow = -inf
                    lrow[t] *= alpha;
#pragma unroll
                    for (int n = 0; n < NO; ++n) o[t][n] *= alpha;
                    mrow[t] = mnew;
                }
                // exponentials: the arguments and the row sum two at a time (v_pk_fma_f32 / v_pk_add_f32)
                const f2 sc2 = f2{sc, sc}, msc2 = f2{msc, msc};
                f2 rs2 = f2{0.f, 0.f};
#pragma unroll
                for (int n = 0; n < NSUB; ++n)
#pragma unroll
                    for (int r = 0; r < 4; r += 2) {
                        const f2 a = f2{s[t][n][r], s[t][n][r + 1]} * sc2 - msc2;
                        const f2 e = f2{__builtin_amdgcn_exp2f(a[0]), __builtin_amdgcn_exp2f(a[1])};
                        s[t][n][r] = e[0];
                        s[t][n][r + 1] = e[1];
                        rs2 += e;
                    }
                lrow[t] += rs2[0] + rs2[1];                              // this lane's keys only: summed over g at the end
                // P^T fragments: 32-key step j = subtiles 2j, 2j+1; element e -> key 16 (2j + e/4) + 4g + e%4
#pragma unroll
                for (int j = 0; j < NJ; ++j) {
                    const bool two = 2 * j + 1 < NSUB;                   // (odd NSUB: the second subtile of the last step is absent)
                    pf[t][j] = h8{(half_t)s[t][2 * j][0], (half_t)s[t][2 * j][1], (half_t)s[t][2 * j][2], (half_t)s[t][2 * j][3],
                                  two ? (half_t)s[t][two ? 2 * j + 1 : 0][0] : (half_t)0.f, two ? (half_t)s[t][two ? 2 * j + 1 : 0][1] : (half_t)0.f,
                                  two ? (half_t)s[t][two ? 2 * j + 1 : 0][2] : (half_t)0.f, two ? (half_t)s[t][two ? 2 * j + 1 : 0][3] : (half_t)0.f};
                }
            }
            // ---- O^T += V^T P^T : A = V^T[dim 16 n + li][key(g, e)] by two transposing reads per fragment ----
            // ds_read_b64_tr_b16: lane i of a 16-lane group supplies the address of row (i >> 2), columns 4 (i & 3) .. + 3 of
            // a 4-row x 16-column block and receives column i of the four rows
#pragma unroll
            for (int n = 0; n < NO; ++n)
#pragma unroll
                for (int j = 0; j < NJ; ++j) {
                    const int r0 = 16 * (2 * j) + 4 * g + (li >> 2), r1 = r0 + 16;
                    const int cc = 2 * n + ((li & 3) >> 1), off = (li & 1) * 4;
                    const s4v lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
                        (__attribute__((address_space(3))) s4v *)(tV + r0 * HD + (cc ^ vswz<HD>(r0)) * 8 + off));
                    const s4v hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
                        (__attribute__((address_space(3))) s4v *)(tV + r1 * HD + (cc ^ vswz<HD>(r1)) * 8 + off));
                    const s8v both = s8v{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
                    const h8 vf = __builtin_bit_cast(h8, both);
#pragma unroll
                    for (int t = 0; t < QT; ++t) o[t][n] = mfma16(vf, pf[t][j], o[t][n]);
                }
        }
    };
    // The loop over whole tiles has ONE body (the short last tile is peeled off behind it, and the waves past the last query,
    // which only help with the staging, run their own copy of the loop): with the three tile lengths and the idle case merging
    // inside one loop body the accumulators went through ~20 register moves per tile.
    auto run = [&](auto live_tag) {
        constexpr bool LIVE = decltype(live_tag)::value;
        int kt = k_lo, buf = 0;
        for (; k_hi - kt > 32; kt += KB, buf ^= 1) {
            if constexpr (LIVE) tile_body(std::integral_constant<int, 4>{}, kt, buf);
            // next tile: registers -> the other LDS buffer (its last readers passed the previous barrier), then fetch the one after
            if (kt + KB < k_hi) {
                store_tile(buf ^ 1);
                if (kt + 2 * KB < k_hi) load_tile(kt + 2 * KB);
            }
            __syncthreads();
        }
        if constexpr (LIVE) {
            if (kt < k_hi) {                                             // (the last tile: nothing is staged behind it)
                if (k_hi - kt > 16) tile_body(std::integral_constant<int, 2>{}, kt, buf);
                else tile_body(std::integral_constant<int, 1>{}, kt, buf);
            }
        }
    };
    if (wave_live) run(std::true_type{});
    else run(std::false_type{});

    // ---- epilogue: O / l -> fp16 [b, t, h*HD + d]; lane (li, g) holds dims 16 n + 4 g + r of query li ----
    if (!wave_live) return;
#pragma unroll
    for (int t = 0; t < QT; ++t) {
        float l = lrow[t];
        l += __shfl_xor(l, 16, 64);
        l += __shfl_xor(l, 32, 64);
        const int qi = qw + 16 * t + li;
        if (qi >= Tq) continue;
        const float inv = l > 0.f ? 1.0f / l : 0.f;
        half_t *dst = p.O + ob_off + (int64_t)qi * p.o_st + (int64_t)h * HD;
#pragma unroll
        for (int n = 0; n < NO; ++n)
            *reinterpret_cast<h4 *>(dst + 16 * n + 4 * g) =
                h4{(half_t)(o[t][n][0] * inv), (half_t)(o[t][n][1] * inv), (half_t)(o[t][n][2] * inv), (half_t)(o[t][n][3] * inv)};
    }
}

template <int HD, int QT>
static hipError_t launch_qt(const AttnParams &p, hipStream_t s) {
    const int nqb = cdiv(p.T - ((p.q_trim && p.cu) ? 2 : 0), 64 * QT);
    dim3 grid(((p.B * p.heads + 7) / 8) * 8 * nqb);
    if (p.causal) OPUS_LAUNCH(KC_ATTN_PREFILL, (attn_prefill_kernel<HD, true, QT>), grid, dim3(256), 0, s, p);
    else OPUS_LAUNCH(KC_ATTN_PREFILL, (attn_prefill_kernel<HD, false, QT>), grid, dim3(256), 0, s, p);
    return hipGetLastError();
}
template <int HD>
static hipError_t launch_hd(const AttnParams &p, hipStream_t s) {
    // Measured (tools/bench_attn.py ab, one MI355X, both forms in one process, round 3 after the softmax clean-up: one copy of
    // the softmax code, v_max3 chain + row swaps, packed fma / add, buffer-descriptor staging).  Two query tiles per wave
    // share every K / V fragment read, the tile staging and the barrier between 32 instead of 16 queries per wave (156 VGPRs:
    // 3 waves per SIMD): 64 x 514 x 20 heads x 64: 163 vs 222 us (530 vs 390 TFLOP/s; T = 512: 137 vs 188 us = 625 TFLOP/s);
    // 32 x 1026 x 40 x 64: 527 vs 635 us (655 TFLOP/s).  One tile per wave (104 VGPRs: 4 waves per SIMD, twice the
    // workgroups) stays ahead where the grid is small (batch 1: 12.6 vs 13.8 us) and at head_dim 128, where two tiles need
    // 240 VGPRs (batch-64 decoder prefill, 96 positions, causal: 47.8 vs 50.6 us; batch 1: 7.0 vs 8.6 us).
    // (Not kept: the ragged last query block - T = 514 leaves 2 of 128 queries - as a second launch of the one-tile form:
    //  167 vs 158-165 us on one box; the 19-30 % that block costs varies more between boxes than the split recovers.)
    // (Not kept, round 4: a fifth wave per workgroup that is alive only in a row's last block and takes the <= 32 queries
    //  behind it, so that T = 514 needs 4 blocks instead of 5.  320-thread workgroups at 157 VGPRs fit two to a CU (10 of 12
    //  wave slots) where 256-thread ones fit three, and the fifth waves that leave at once do not give the slots back in
    //  time: 64 x 514: 281 vs 167 us, 64 x 512: 158 vs 138 us, 32 x 1026 x 40: 685 vs 542 us.)
    bool one = !(HD <= 64 && (int64_t)p.B * p.heads * cdiv(p.T, 128) >= 512);
    if (g_knobs.misc[3]) one = !one;                                 // A/B aid
    return one ? launch_qt<HD, 1>(p, s) : launch_qt<HD, 2>(p, s);
}

hipError_t launch_attn_prefill(const AttnParams &p, hipStream_t s) {
    if (p.T <= 0 || p.B <= 0) return hipErrorInvalidValue;
    // the 8-byte output stores and 16-byte operand loads need these alignments (every caller of the path satisfies them)
    if ((p.o_st & 3) || (p.o_sb & 3) || (p.q_st & 7) || (p.k_st & 7) || (p.v_st & 7) || (p.q_sb & 7) || (p.k_sb & 7) || (p.v_sb & 7))
        return hipErrorInvalidValue;
    if ((int64_t)p.T * (p.k_st > p.v_st ? p.k_st : p.v_st) * 2 >= (1ll << 31)) return hipErrorInvalidValue;   // 32-bit buffer offsets
    switch (p.head_dim) {
        case 16: return launch_hd<16>(p, s);
        case 32: return launch_hd<32>(p, s);
        case 64: return launch_hd<64>(p, s);
        case 128: return launch_hd<128>(p, s);
    }
    return hipErrorInvalidValue;
}

}  // namespace opus
