// Decode-step attention (rows D3/D4): for one new token per batch row,
//   [sum of the QKV GEMM's k-part slabs * rstd + bias] -> rotary(q, k) at position slot - kstart[b]
//   -> append k, v to the cache at `slot` -> softmax(q K^T * scale over keys kstart[b]..slot) V.
//
// Workgroup = GP query heads of one kv group of one batch row (GP = 1 at small batch: 32 workgroups per row instead of 8 -
// the step is latency-bound, so spread it; GP = group at large batch: K/V of a kv head are read once per row), 4 waves.
// The keys are split over the WAVES in tiles of 32 and every wave runs its own online softmax ("flash decoding"): there is
// no block-wide reduction inside the key loop - two barriers in the whole kernel (after the rotary staging, before the final
// combine) instead of the eight of a block-wide max / sum / PV pipeline; that chain of barriers, not bytes, was the time.
//   scores   S^T[key][head] = K[key][:] q[head][:]  MFMA 16x16x32: A = 16 cached keys x 32 dims straight from the cache
//            (64 contiguous bytes per key and instruction), B = the rotated query heads from LDS (columns >= GP are zero)
//   softmax  per head = per MFMA column: 8 scores per lane, combined over the four 16-lane groups with two xor-shuffles
//   P V      lane = (8-wide slice of the head dim, group of 32 / KPN keys): V rows are read with fully coalesced 16-B loads,
//            P comes back from a wave-private LDS patch (consecutive probabilities per lane)
//   combine  (m, l, O) of the 4 waves x KPN key groups through LDS, one pass.
// The new key / value are used from LDS, so nothing depends on in-launch global visibility.
// slot = T0 + *step: both are read from device memory (d_step = {step, T0}) so the same launch can be replayed from a hipGraph
// for any step of any prompt length.
// Latency chain (round 3).  The kernel is one wave of workgroups whose time is a chain of dependent memory round trips, not
// bytes.  Two links are gone: (1) key tiles are indexed by ABSOLUTE cache slot (tile t = slots 32 t .. 32 t + 31, masked to
// kstart[b] <= slot < T0 + step afterwards), so the first tile's K / V addresses depend on nothing the kernel has to load and
// are requested at entry beside *step and kstart[b] (rows whose left padding is >= 32 slots re-request: speed only); (2) the
// rotary (cos, sin) row of each batch row's position comes from a per-step table cs_row[b] written once per decode step by
// the embedding kernel instead of from the position table behind (step, kstart).
#include "common.h"
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <type_traits>

namespace opus {

constexpr int MAXG = 8;

typedef short s4v __attribute__((ext_vector_type(4)));
typedef short s8v __attribute__((ext_vector_type(8)));
// V tile in LDS: row-major [key][HD], 16-B chunk c of row r stored at c ^ vswz_d(r) - the swizzle of attn_prefill_kernel's V
// tile: the 8 consecutive rows that the two 16-lane groups of a half wave read with one ds_read_b64_tr_b16 fall on 64 banks
template <int HD>
__device__ __forceinline__ int vswz_d(int row) {
    if (HD == 128) return (row & 7) << 1;
    if (HD == 64) return ((row >> 1) & 3) << 1;
    if (HD == 32) return ((row >> 2) & 1) << 1;
    return 0;
}

template <int HD, int GP>
__global__ __launch_bounds__(256) void attn_decode_kernel(AttnDecodeParams p) {
    extern __shared__ __attribute__((aligned(16))) char smraw[];
    constexpr int HALF = HD / 2;
    constexpr int HDP = HD < 32 ? 32 : HD;          // k extent of the score MFMAs (zero-padded for head_dim 16)
    constexpr int KS = HDP / 32;                    // MFMA k-steps per 16-key subtile
    constexpr int DV = HD / 8;                      // 16-B slices per K / V row
    constexpr int KPN = 64 / DV;                    // key groups of the PV lanes
    constexpr int KPK = 32 / KPN;                   // keys per group and tile
    constexpr int NO = HD / 16;                     // output dim tiles of the PV MFMAs
    // dynamic LDS: [sq fp16 GP x HDP][sk fp16 HDP][sv fp16 HDP][stats 4 x GP x 2 f32][red 4 x GP x HD f32][vt fp16 4 x 32 x HD]
    half_t *sq = reinterpret_cast<half_t *>(smraw);
    half_t *sk = sq + GP * HDP;
    half_t *sv = sk + HDP;
    float *stats = reinterpret_cast<float *>(sv + HDP);
    float *red = stats + 4 * GP * 2;
    half_t *vt = reinterpret_cast<half_t *>(red + 4 * GP * HD);

    // every kernel argument the staging phase needs in ONE scalar round trip: left to itself the compiler fetches them in two
    // batches (the second one just before the first vector load), i.e. two dependent misses before any data is requested
    asm volatile("" ::"s"(p.qkv), "s"(p.slabs), "s"(p.slab_stride), "s"(p.row_ssq), "s"(p.bias), "s"(p.cs_row), "s"(p.kstart), "s"(p.step),
                 "s"(p.kc), "s"(p.vc), "s"(p.cache_sb), "s"(p.cache_sh), "s"(p.out));
    asm volatile("" ::"s"(p.ks), "s"(p.row_nblk), "s"(p.eps), "s"(p.K), "s"(p.T0), "s"(p.nh), "s"(p.nkv), "s"(p.ctx_cap), "s"(p.scale), "s"(p.out_tiled));
    const int b = blockIdx.y, tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wgid = blockIdx.y * gridDim.x + blockIdx.x;
    if (p.trace && tid == 0) p.trace[wgid * 8] = wall_clock64();
    const int g = lane >> 4, li = lane & 15;
    const int G = p.nh / p.nkv;
    const int h0 = blockIdx.x * GP;                 // first query head of this workgroup
    const int kvh = h0 / G;
    const bool writer = (h0 % G) == 0;              // one workgroup per kv head appends to the cache
    half_t *kcb = p.kc + b * p.cache_sb + kvh * p.cache_sh;
    half_t *vcb = p.vc + b * p.cache_sb + kvh * p.cache_sh;
    const int64_t ld = (int64_t)(p.nh + 2 * p.nkv) * HD;

    // ---- this wave's first tile (K fragments + V rows of absolute slots 32 wave ..): requested inside stage(), right behind
    // the new token's operands ----
    h8 kf[2][KS], vr[KPK];
    const int dv = lane % DV, kp = lane / DV;
    const int last_slot = p.ctx_cap - 1;
    auto load_tile = [&](int t) {
        const int j0 = 32 * t;
#pragma unroll
        for (int a = 0; a < 2; ++a) {
            int j = j0 + 16 * a + li;
            j = j < last_slot ? j : last_slot;       // (slots outside [kstart, slot) are masked in tile())
            const half_t *kr = kcb + (int64_t)j * HD;
#pragma unroll
            for (int s = 0; s < KS; ++s) {
                const int d = 32 * s + 8 * g;
                kf[a][s] = d < HD ? *reinterpret_cast<const h8 *>(kr + d) : h8{0, 0, 0, 0, 0, 0, 0, 0};
            }
        }
#pragma unroll
        for (int i = 0; i < KPK; ++i) {
            int j = j0 + KPK * kp + i;
            j = j < last_slot ? j : last_slot;
            vr[i] = *reinterpret_cast<const h8 *>(vcb + (int64_t)j * HD + dv * 8);
        }
    };
    // *step and kstart[b] through the SCALAR cache (constant address space: s_load, its own counter): as vector loads their
    // first use drained every vector load issued before them.  They are requested inside stage(), BEHIND the vector loads
    // (scheduling barrier): placed here the compiler waited for them - a second scalar round trip behind the kernel
    // arguments' - before it issued the first vector load.
    typedef const __attribute__((address_space(4))) int32_t *cint_p;
    int slot = 0, kstart = 0;

    // ---- the new token's q / k / v (optionally: sum of the QKV GEMM's k-part slabs, RMSNorm row scale, bias), rotary on
    // the query heads and the key; stage them in LDS and append k, v to the cache ----
    // Every global value this phase needs is REQUESTED before the first one is used: the row's sums of squares one block per
    // lane (one round trip instead of row_nblk), then - per thread - cos / sin, bias and the ks slab terms of its (at most NE)
    // rotary pairs and of its v element.  Written as "load, add, load, add" (a runtime-length loop over slabs, stores to the
    // cache in between) this phase was a chain of ~30 dependent L2 round trips and two thirds of the kernel's time.
    float rstd = 1.0f;
    float tq[4] = {0.f, 0.f, 0.f, 0.f};              // the row's sums of squares: 4 blocks per lane requested here, summed in stage()
    if (p.row_ssq) {
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int j = 64 * u + lane;
            tq[u] = p.row_ssq[(int64_t)b * p.row_nblk + (j < p.row_nblk ? j : p.row_nblk - 1)];   // (masked where it is used)
        }
    }
    auto finish_rstd = [&]() {                       // called behind the requests of stage(): one round trip for everything
        if (!p.row_ssq) return;
#pragma unroll
        for (int u = 0; u < 4; ++u) tq[u] = 64 * u + lane < p.row_nblk ? tq[u] : 0.f;
        float q = (tq[0] + tq[1]) + (tq[2] + tq[3]);
        for (int j0 = 256; j0 < p.row_nblk; j0 += 256) {          // (more than 256 blocks: residual streams wider than 4096)
            float t[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int j = j0 + 64 * u + lane;
                t[u] = p.row_ssq[(int64_t)b * p.row_nblk + (j < p.row_nblk ? j : p.row_nblk - 1)];
                t[u] = j < p.row_nblk ? t[u] : 0.f;
            }
            q += (t[0] + t[1]) + (t[2] + t[3]);
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) q += __shfl_xor(q, o, 64);
        rstd = rsqrtf(q / (float)p.K + p.eps);
    };
    constexpr int NE = ((GP + 1) * HALF + 255) / 256;
    auto stage = [&](auto ksn_tag) {
        constexpr int KSN = decltype(ksn_tag)::value;          // number of slabs; 0: finished fp16 projections in p.qkv
        constexpr int NT = KSN ? KSN : 1;
        float ta[NE][NT], tb[NE][NT], tv[NT], cc[NE], sn[NE], ba[NE], bb2[NE], bv = 0.f;
        int jj[NE], dd[NE];
        bool ok[NE];
        const bool vok = tid < HD;
        const int64_t vcol = (int64_t)(p.nh + p.nkv + kvh) * HD + (vok ? tid : 0);
#pragma unroll
        for (int e = 0; e < NE; ++e) {
            const int i = tid + 256 * e;
            ok[e] = i < (GP + 1) * HALF;
            const int ii = ok[e] ? i : 0;
            jj[e] = ii / HALF;
            dd[e] = ii % HALF;
            const int64_t col = (jj[e] < GP ? (int64_t)(h0 + jj[e]) * HD : (int64_t)(p.nh + kvh) * HD) + dd[e];
            const float2 csv = *reinterpret_cast<const float2 *>(p.cs_row + ((int64_t)b * HALF + dd[e]) * 2);
            cc[e] = csv.x;
            sn[e] = csv.y;
            if (KSN) {
#pragma unroll
                for (int k = 0; k < NT; ++k) {
                    ta[e][k] = p.slabs[k * p.slab_stride + b * ld + col];
                    tb[e][k] = p.slabs[k * p.slab_stride + b * ld + col + HALF];
                }
                ba[e] = p.bias ? p.bias[col] : 0.f;
                bb2[e] = p.bias ? p.bias[col + HALF] : 0.f;
            } else {
                ta[e][0] = (float)p.qkv[b * ld + col];
                tb[e][0] = (float)p.qkv[b * ld + col + HALF];
                ba[e] = bb2[e] = 0.f;
            }
        }
        if (KSN) {
#pragma unroll
            for (int k = 0; k < NT; ++k) tv[k] = p.slabs[k * p.slab_stride + b * ld + vcol];
            bv = p.bias ? p.bias[vcol] : 0.f;
        } else {
            tv[0] = (float)p.qkv[b * ld + vcol];
        }
        // this wave's first key tile is requested BEHIND the new token's operands (vector-memory results return in issue
        // order): the ~29 MB of cached K / V of a batch-64 step stream in for ~6 us - HBM-bound - and the rotary / staging
        // work below runs underneath instead of behind them
        if (32 * wave < p.ctx_cap) load_tile(wave);
        __builtin_amdgcn_sched_barrier(0);
        slot = (p.T0 >= 0 ? p.T0 : ((cint_p)p.step)[1]) + ((cint_p)p.step)[0];   // (T0 < 0: prompt length from d_step[1] - a captured step serves any T)
        kstart = ((cint_p)p.kstart)[b];
        finish_rstd();
        // projection output rounded to fp16, as the unfused GEMM stores it
        auto fin = [&](const float (&t)[NT], float bias) -> float {
            if (!KSN) return t[0];
            float v = 0.f;
#pragma unroll
            for (int k = 0; k < NT; ++k) v += t[k];
            v *= rstd;
            if (p.bias) v += bias;
            return (float)(half_t)v;
        };
#pragma unroll
        for (int e = 0; e < NE; ++e) {
            if (!ok[e]) continue;
            const float a = fin(ta[e], ba[e]), bb = fin(tb[e], bb2[e]);
            float rl, rh;
            rotate_pair(a, bb, cc[e], sn[e], rl, rh);              // (same arithmetic as the prefill's rotary kernel)
            const half_t lo = (half_t)rl, hi = (half_t)rh;
            const int j = jj[e], d = dd[e];
            if (j < GP) {
                sq[j * HDP + d] = lo;
                sq[j * HDP + d + HALF] = hi;
            } else {
                sk[d] = lo;
                sk[d + HALF] = hi;
                if (writer) {
                    kcb[(int64_t)slot * HD + d] = lo;
                    kcb[(int64_t)slot * HD + d + HALF] = hi;
                }
            }
        }
        if (vok) {
            const half_t v = (half_t)fin(tv, bv);
            sv[tid] = v;
            if (writer) vcb[(int64_t)slot * HD + tid] = v;
        }
    };
    static_assert(HD <= 256, "one v element per thread");
    switch (p.slabs ? p.ks : 0) {
        case 0: stage(std::integral_constant<int, 0>{}); break;
        case 1: stage(std::integral_constant<int, 1>{}); break;
        case 2: stage(std::integral_constant<int, 2>{}); break;
        case 3: stage(std::integral_constant<int, 3>{}); break;
        case 4: stage(std::integral_constant<int, 4>{}); break;
        case 5: stage(std::integral_constant<int, 5>{}); break;
        case 6: stage(std::integral_constant<int, 6>{}); break;
        case 7: stage(std::integral_constant<int, 7>{}); break;
        default: stage(std::integral_constant<int, 8>{}); break;   // launch_attn_decode rejects ks > 8
    }
    const int t_first = kstart >> 5;                // first tile with a visible key
    const int t_new = slot >> 5;                    // cached keys are slots kstart .. slot-1; the new key (slot) comes from LDS
    if (HD < HDP) {                                  // zero padding of the 32-wide MFMA k extent
        for (int i = tid; i < (GP + 1) * (HDP - HD); i += 256) {
            const int j = i / (HDP - HD), d = HD + i % (HDP - HD);
            if (j < GP) sq[j * HDP + d] = (half_t)0.f;
            else sk[d] = (half_t)0.f;
        }
    }
    if (p.trace && tid == 0) p.trace[wgid * 8 + 1] = wall_clock64();       // (this wave's q / k / v staged)
    __syncthreads();
    if (p.trace && tid == 0) p.trace[wgid * 8 + 2] = wall_clock64();

    // query fragments: B operand, column li = head (zero beyond GP)
    h8 qf[KS];
#pragma unroll
    for (int s = 0; s < KS; ++s)
        qf[s] = li < GP ? *reinterpret_cast<const h8 *>(sq + li * HDP + 32 * s + 8 * g) : h8{0, 0, 0, 0, 0, 0, 0, 0};

    // per lane (li = head, g): running maximum, this lane's share of the row sum (summed over g at the end) and
    // O^T[dim 16 n + 4 g + r][head li] = o[n][r]
    float m_run = -INFINITY, l_run = 0.f;
    f4 o[NO];
#pragma unroll
    for (int n = 0; n < NO; ++n) o[n] = f4{0.f, 0.f, 0.f, 0.f};
    half_t *myv = vt + wave * 32 * HD;               // this wave's V tile: row-major [32 keys][HD], 16-B chunks XOR-swizzled

    // one tile of 32 keys of which [lo, hi) are visible (lo < hi); K fragments in kf, V rows in vr.
    //   S^T = K q^T      lane (li = head, g): s2[a][r] = score of key 16a + 4g + r
    //   O^T += V^T P^T   the fp16-rounded probabilities are the B operand as they stand (contraction index key(g, e) =
    //                    16 (e / 4) + 4 g + e % 4, as in attn_prefill_kernel); V^T fragments come from the wave's row-major
    //                    V tile in LDS through the transposing read ds_read_b64_tr_b16.  (Round 2 ran this product on the
    //                    VALU from a probability patch in LDS: 256 FMAs + 40 LDS reads per lane and tile at GP = 4.)
    auto tile = [&](int lo, int hi) {
        // V rows -> LDS; rows outside [lo, hi) as zeros (their probabilities are exactly 0, and 0 x anything must stay 0)
        const bool ragged = lo > 0 || hi < 32;       // (wave-uniform)
#pragma unroll
        for (int i = 0; i < KPK; ++i) {
            const int j = KPK * kp + i;
            h8 v = vr[i];
            if (ragged && (j < lo || j >= hi)) v = h8{0, 0, 0, 0, 0, 0, 0, 0};
            *reinterpret_cast<h8 *>(myv + j * HD + (dv ^ vswz_d<HD>(j)) * 8) = v;
        }
        f4 s2[2];
#pragma unroll
        for (int a = 0; a < 2; ++a) {
            s2[a] = f4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int s = 0; s < KS; ++s) s2[a] = mfma16(kf[a][s], qf[s], s2[a]);
        }
        float mx = -INFINITY;
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int kk = 16 * a + 4 * g + r;
                const float v = (kk >= lo && kk < hi) ? s2[a][r] * p.scale : -INFINITY;
                s2[a][r] = v;
                mx = fmaxf(mx, v);
            }
        mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
        mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
        const float m_new = fmaxf(m_run, mx);        // lo < hi: finite
        const float alpha = __expf(m_run - m_new);   // 0 on the first tile
        float ps = 0.f;
        h8 pf;
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                // P is rounded to fp16 before the PV product, as the prefill kernel and HF (softmax .to(q.dtype))
                const float e = __expf(s2[a][r] - m_new);   // (0 for masked keys)
                ps += e;
                pf[4 * a + r] = (half_t)e;
            }
        l_run = l_run * alpha + ps;
        m_run = m_new;
        // same-wave LDS round trip of the V tile (in-order LDS queue); the fences keep the compiler from reordering across it
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_s_waitcnt(0xc07f);
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
        for (int n = 0; n < NO; ++n) {
            const int r0 = 4 * g + (li >> 2), r1 = r0 + 16;
            const int cc = 2 * n + ((li & 3) >> 1), off = (li & 1) * 4;
            const s4v lo4 = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
                (__attribute__((address_space(3))) s4v *)(myv + r0 * HD + (cc ^ vswz_d<HD>(r0)) * 8 + off));
            const s4v hi4 = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
                (__attribute__((address_space(3))) s4v *)(myv + r1 * HD + (cc ^ vswz_d<HD>(r1)) * 8 + off));
            const s8v both = s8v{lo4[0], lo4[1], lo4[2], lo4[3], hi4[0], hi4[1], hi4[2], hi4[3]};
            o[n] *= alpha;
            o[n] = mfma16(__builtin_bit_cast(h8, both), pf, o[n]);
        }
        // the V tile is rewritten by the next tile: its reads above must have retired (same wave, in order)
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_s_waitcnt(0xc07f);
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    };

    // The new key / value (fed from LDS: nothing depends on in-launch global visibility) takes its place - cache slot `slot` -
    // in the tile that slot belongs to: no separate one-key tile on top of one wave's share (that wave was the critical path).
    // (measured and not kept, round 5: the wave's next tile (t + 4) requested into a second register set before the current one is
    //  multiplied - 256 instead of 196 VGPRs, still 2 waves per SIMD - for contexts of 128 - 352 slots: attn_decode 173.0 / 174.0 ms
    //  per 256-token step with the prefetch against 170.9 / 168.8 with the loads just before their use in the same build, and 133.3 ms
    //  for this single-set form (profiles/r05_decode_ab.txt).  With 512 workgroups x 4 waves every CU already has 8 tile loads in
    //  flight; the kernel moves its 59 MB of K / V per launch in ~11 us (5.5 TB/s) + ~5 us of fixed cost - queue depth is not its limit.)
    for (int t = t_first + wave; t <= t_new; t += 4) {
        if (t != wave) load_tile(t);                 // (the tile requested at entry is the right one unless the row is padded by >= 32)
        const int lo = kstart - 32 * t;
        int hi = slot - 32 * t;
        if (t == t_new) {                            // (wave-uniform) hi = the new key's index in this tile
#pragma unroll
            for (int a = 0; a < 2; ++a)
#pragma unroll
                for (int s = 0; s < KS; ++s) {
                    const h8 nk = *reinterpret_cast<const h8 *>(sk + 32 * s + 8 * g);
                    kf[a][s] = 16 * a + li == hi ? nk : kf[a][s];
                }
            const h8 nv = *reinterpret_cast<const h8 *>(sv + dv * 8);
#pragma unroll
            for (int i = 0; i < KPK; ++i) vr[i] = KPK * kp + i == hi ? nv : vr[i];
            ++hi;
        }
        tile(lo > 0 ? lo : 0, hi < 32 ? hi : 32);
    }
    if (p.trace && tid == 0) p.trace[wgid * 8 + 3] = wall_clock64();       // (wave 0's key tiles done)

    // ---- publish (m, l) per head and the partial outputs; combine ----
    l_run += __shfl_xor(l_run, 16, 64);
    l_run += __shfl_xor(l_run, 32, 64);
    if (li < GP) {
        if (g == 0) {
            stats[(wave * GP + li) * 2] = m_run;
            stats[(wave * GP + li) * 2 + 1] = l_run;
        }
#pragma unroll
        for (int n = 0; n < NO; ++n) *reinterpret_cast<f4 *>(red + (wave * GP + li) * HD + 16 * n + 4 * g) = o[n];
    }
    __syncthreads();
    if (p.trace && tid == 0) p.trace[wgid * 8 + 4] = wall_clock64();
    // 8 consecutive head dims per thread: one 16-B store into either output layout (a fragment-ordered row keeps 8 consecutive
    // k together; one half per thread was 128-512 two-byte stores per workgroup)
    for (int i = tid; i < GP * HD / 8; i += 256) {
        const int h = i / (HD / 8), d = (i % (HD / 8)) * 8;
        float M = -INFINITY;
#pragma unroll
        for (int w = 0; w < 4; ++w) M = fmaxf(M, stats[(w * GP + h) * 2]);
        float num[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f}, den = 0.f;
#pragma unroll
        for (int w = 0; w < 4; ++w) {
            const float mw = stats[(w * GP + h) * 2];
            const float f = mw == -INFINITY ? 0.f : __expf(mw - M);
            const f4 lo = *reinterpret_cast<const f4 *>(red + (w * GP + h) * HD + d), hi = *reinterpret_cast<const f4 *>(red + (w * GP + h) * HD + d + 4);
#pragma unroll
            for (int e = 0; e < 4; ++e) { num[e] += f * lo[e]; num[4 + e] += f * hi[e]; }
            den += f * stats[(w * GP + h) * 2 + 1];
        }
        h8 o;
#pragma unroll
        for (int e = 0; e < 8; ++e) o[e] = (half_t)(num[e] / den);
        const int kcol = (h0 + h) * HD + d;             // first column of the piece in the [B, nh HD] context matrix
        *reinterpret_cast<h8 *>(p.out + (p.out_tiled ? tiled_off(b, kcol, p.nh * HD) : (int64_t)b * p.nh * HD + kcol)) = o;
    }
    if (p.trace && tid == 0) p.trace[wgid * 8 + 5] = wall_clock64();
}

template <int HD, int GP>
static hipError_t launch_t(const AttnDecodeParams &p, int B, hipStream_t s) {
    constexpr int HDP = HD < 32 ? 32 : HD;
    const size_t lds = (size_t)(GP * HDP + 2 * HDP) * sizeof(half_t) + ((size_t)4 * GP * 2 + (size_t)4 * GP * HD) * sizeof(float) +
                       (size_t)4 * 32 * HD * sizeof(half_t);
    if (lds > 150 * 1024) return hipErrorInvalidValue;
    if (lds > 48 * 1024) {
        hipError_t ea = ensure_dyn_lds(reinterpret_cast<const void *>(&attn_decode_kernel<HD, GP>), lds);
        if (ea != hipSuccess) return ea;
    }
    static const bool trace = getenv("OPUS_ATTN_TRACE") != nullptr;     // tuning aid: per-workgroup section stamps on stderr
    if (trace) {
        static long long *tb = nullptr;
        static int calls = 0;
        const int nwg = p.nh / GP * B;
        if (!tb) (void)hipMalloc((void **)&tb, (size_t)8192 * 8 * sizeof(long long));
        ++calls;
        if (tb && nwg <= 8192 && calls >= 200 && calls < 204) {        // (a few launches of a warm decode loop; run with OPUS_NO_GRAPH=1)
            AttnDecodeParams q = p;
            q.trace = tb;
            hipLaunchKernelGGL((attn_decode_kernel<HD, GP>), dim3(p.nh / GP, B), dim3(256), lds, s, q);
            (void)hipStreamSynchronize(s);
            std::vector<long long> h((size_t)nwg * 8);
            (void)hipMemcpy(h.data(), tb, h.size() * sizeof(long long), hipMemcpyDeviceToHost);
            long long t0 = h[0];
            for (int i = 0; i < nwg; ++i) t0 = std::min(t0, h[8 * i]);
            const char *nm[6] = {"start", "q / k / v staged (wave 0)", "after the staging barrier", "wave 0's key tiles done", "after the combine barrier", "end"};
            fprintf(stderr, "[attn_decode trace] B=%d heads=%d GP=%d wgs=%d (times after the first start)\n", B, p.nh, GP, nwg);
            for (int k = 0; k < 6; ++k) {
                std::vector<double> v;
                for (int i = 0; i < nwg; ++i) v.push_back((h[8 * i + k] - t0) * 0.01);
                std::sort(v.begin(), v.end());
                fprintf(stderr, "   %-28s min %6.2f  p10 %6.2f  median %6.2f  p90 %6.2f  max %6.2f us\n", nm[k], v.front(), v[v.size() / 10], v[v.size() / 2],
                        v[v.size() * 9 / 10], v.back());
            }
        }
    }
    OPUS_LAUNCH(KC_ATTN_DECODE, (attn_decode_kernel<HD, GP>), dim3(p.nh / GP, B), dim3(256), lds, s, p);
    return hipGetLastError();
}

template <int HD>
static hipError_t launch_hd(const AttnDecodeParams &p, int B, hipStream_t s) {
    const int G = p.nh / p.nkv;
    // grouped form only when the per-head form would already fill the chip several times over
    static const int group_min = getenv("OPUS_ATTN_GROUP_MIN") ? atoi(getenv("OPUS_ATTN_GROUP_MIN")) : 256;   // tuning aid
    const bool grouped = G > 1 && (int64_t)B * p.nkv >= group_min;
    if (!grouped) return launch_t<HD, 1>(p, B, s);
    switch (G) {
        case 2: return launch_t<HD, 2>(p, B, s);
        case 4: return launch_t<HD, 4>(p, B, s);
        case 8: return launch_t<HD, 8>(p, B, s);
    }
    return launch_t<HD, 1>(p, B, s);
}

hipError_t launch_attn_decode(const AttnDecodeParams &p, int B, int hd, hipStream_t s) {
    const int G = p.nh / p.nkv;
    if (G > MAXG || G * p.nkv != p.nh) return hipErrorInvalidValue;
    if (p.slabs && (p.ks < 1 || p.ks > 8)) return hipErrorInvalidValue;
    switch (hd) {
        case 16: return launch_hd<16>(p, B, s);
        case 32: return launch_hd<32>(p, B, s);
        case 64: return launch_hd<64>(p, B, s);
        case 128: return launch_hd<128>(p, B, s);
    }
    return hipErrorInvalidValue;
}

}  // namespace opus
