// gemm_stream: ONE-launch weight-streaming GEMM for the narrow projections of the batched decode step
// (5 <= M <= 64 rows; QKV, wo, down of SURVEY 8a D3: N <= 6144 output columns, K = 4096 .. 14336).
//
// Why another kernel.  With 16 output columns per weight panel a 4096-column projection has 256 panels - one per CU - so
// the kernels that share an activation slice between the panels of a workgroup through LDS (gemm_wide / gemm_ring) must
// cut K over WORKGROUPS to fill the chip: fp32 slabs, a second launch to sum them, and 5-9 us of fixed cost around 5-13 us
// of streaming (round 2: 2.6-2.9 TB/s on these shapes).  Here a workgroup owns P whole panels over the whole K (or over
// one of `ksplit` k-parts) and cuts K over its WAVES instead:
//   * wave w multiplies chunks w, w + NW, w + 2 NW, ... (64 k each) of the workgroup's panels: per chunk and panel one
//     2-KB weight block straight from HBM to registers (non-temporal; the panel-tiled layout IS the MFMA A fragment) and,
//     per 16-row tile, two 16-B-per-lane activation fragments straight from the L2-resident activation matrix into the
//     MFMA B layout (buffer loads: rows >= M read as zeros) - no LDS staging, no barrier, no DMA in the main loop;
//   * the NW partial tiles are combined through LDS in wave order (fixed order: bitwise reproducible, and a row's result
//     does not depend on which rows share the launch), and the combining lanes own 4 consecutive columns of one row, so
//     bias / residual / output / the fp16 copy move as 16-B / 8-B accesses;
//   * with ksplit == 1 the epilogue is the whole GEMM epilogue, including the producer side of the row-scale RMSNorm
//     fusion (GemmParams::xh_out / ssq_out: fp16(x) and the sum of squares of every 16-column block of the row);
//     with ksplit > 1 (the QKV projection: 384 panels = 1.5 per CU, cut as 128 column groups x 2 k-parts) the k-parts
//     leave raw fp32 slabs for the consumer (attn_decode_kernel sums them), as gemm_wide_kernel did with 5.
// Cost model (measured, tools/bench_gemm.py narrow): a CU takes in at most ~67 GB/s through its L1 from L2, weights
// included, and ~24 GB/s of that from HBM (1/256 of the chip's rate).  A workgroup moves 32 P bytes of weights per k and
// 32 MT bytes of activations per k (MT = M / 16 row tiles): at 64 rows 160 K bytes per CU against 48 K for a k-split over 8
// workgroups, so the kernel is activation-bound where M * K / ksplit is large (down at > 16 rows: the round-2 kernels keep
// those) and HBM-bound + ~3.5 us of ramp elsewhere.
// Activation layout.  A 16-row x 32-k MFMA B fragment read from a row-major matrix touches 16 lines with 64 bytes each
// (42 GB/s per CU measured); read from a matrix stored in FRAGMENT order - the panel-tiled layout of the weights, with the
// row in the place of the output column: block (row / 16, k / 64) = 2 KB, [k-step][lane = 16 (k % 32) / 8 + row % 16][8] -
// it is one contiguous 1-KB wave access (66 GB/s).  GemmParams::a_tiled selects it; the producers of the decode step
// (attn_decode_kernel, the embedding kernel, the epilogues that write fp16(x)) write that layout when asked to.
#include "common.h"
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <type_traits>

namespace opus {

__device__ __forceinline__ float gelu_erf_s(float x) {   // same arithmetic as gemm.hip's gelu_erf
    const float z = fabsf(x) * 0.70710678118654752f;
    const float t = __builtin_amdgcn_rcpf(1.0f + 0.3275911f * z);   // v_rcp_f32 (1 ulp): an IEEE division is ~10 instructions
    const float poly = t * (0.254829592f + t * (-0.284496736f + t * (1.421413741f + t * (-1.453152027f + t * 1.061405429f))));
    const float erf_abs = 1.0f - poly * __expf(-z * z);
    return 0.5f * x * (1.0f + copysignf(erf_abs, x));
}

// MT row tiles of 16, P panels per workgroup, NT threads, U chunk sets in flight per wave
template <int MT, int P, int NT, int U, int EPI>
__global__ __launch_bounds__(NT) void gemm_stream_kernel(GemmParams p, int ksplit) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    // Every kernel argument the main path reads, fetched in ONE scalar batch: left to itself the compiler loads the 330-byte
    // argument block piecemeal - six dependent scalar round trips before the first weight load was issued (OPUS_STREAM_TRACE's
    // "start -> first sets" included them).
    asm volatile("" ::"s"(p.A), "s"(p.W), "s"(p.C), "s"(p.bias), "s"(p.residual), "s"(p.ws), "s"(p.xh_out), "s"(p.ssq_out), "s"(p.combine_cnt),
                 "s"(p.trace), "s"(p.lda), "s"(p.ldc), "s"(p.ldr), "s"(p.M), "s"(p.N), "s"(p.K), "s"(p.out_f32), "s"(p.a_tiled), "s"(p.c_tiled),
                 "s"(p.xh_tiled), "s"(p.no_rot), "s"(ksplit));
    constexpr int NW = NT / 64;
    constexpr int TILES = P * MT;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int g = lane >> 4, li = lane & 15;
    const int chunks = p.K >> 6;
    const int ky = blockIdx.y;
    // (the planner's k-part counts are powers of two: a shift, not the ~200 scalar instructions of two 64-bit divisions that
    //  stood between the kernel-argument fetch and the first weight load)
    const int ksh = __builtin_ctz((unsigned)ksplit);
    const int c0 = (chunks * ky) >> ksh, c1 = (chunks * (ky + 1)) >> ksh;
    const int nck = c1 - c0;
    const int nsteps = (nck + NW - 1) / NW;
    const int panel0 = blockIdx.x * P;
    const int wg = blockIdx.y * gridDim.x + blockIdx.x;
    if (p.trace && tid == 0) p.trace[wg * 8] = wall_clock64();

    const half_t *wp[P];
#pragma unroll
    for (int j = 0; j < P; ++j) wp[j] = p.W + ((int64_t)(panel0 + j) * chunks) * 1024 + lane * 8;
    // activations through a buffer descriptor: lanes whose row does not exist get an out-of-range offset, for which the
    // hardware returns zeros without touching memory and without a branch
    const __amdgpu_buffer_rsrc_t arsrc = __builtin_amdgcn_make_buffer_rsrc(
        (void *)p.A, 0, p.a_tiled ? MT * chunks * 2048 : (int)(((int64_t)(p.M - 1) * p.lda + p.K) * 2), 0x00020000);
    int aoff[MT];
    const bool atiled = p.a_tiled != 0;
    const int cstride = atiled ? 2048 : 128, sstride = atiled ? 1024 : 64;   // bytes between chunks / between the two k-steps
#pragma unroll
    for (int i = 0; i < MT; ++i) {
        const int m = 16 * i + li;
        if (atiled) aoff[i] = i * chunks * 2048 + lane * 16;
        else aoff[i] = m < p.M ? (int)((m * p.lda + g * 8) * 2) : 0x40000000;
    }
    // rotated k-walk (memory-channel camping of the panel stride, see gemm_wide_kernel): a function of the column group and
    // the k-part only
    const int rot = (p.no_rot || nsteps < 2) ? 0 : (int)((blockIdx.x * 3u + (unsigned)ky) % (unsigned)nsteps);

    typedef unsigned int u4 __attribute__((ext_vector_type(4)));
    h8 wr[U][P][2];
    u4 ar[U][MT][2];
    f4 acc[P][MT];
#pragma unroll
    for (int j = 0; j < P; ++j)
#pragma unroll
        for (int i = 0; i < MT; ++i) acc[j][i] = f4{0.f, 0.f, 0.f, 0.f};

    auto chunk_of = [&](int step) {          // chunk this wave multiplies in pipeline step `step` (may be >= c1: nothing)
        int ph = step + rot;
        ph = ph >= nsteps ? ph - nsteps : ph;
        return c0 + ph * NW + wave;
    };
    auto load = [&](auto u_tag, int step) {
        constexpr int u = decltype(u_tag)::value;
        int c = chunk_of(step);
        c = c < c1 ? c : c1 - 1;             // ragged last step: a harmless re-read, skipped by compute()
#pragma unroll
        for (int j = 0; j < P; ++j) {
            const h8 *ptr = reinterpret_cast<const h8 *>(wp[j] + (int64_t)c * 1024);
            wr[u][j][0] = __builtin_nontemporal_load(ptr);
            wr[u][j][1] = __builtin_nontemporal_load(ptr + 64);
        }
#pragma unroll
        for (int i = 0; i < MT; ++i) {
            ar[u][i][0] = __builtin_amdgcn_raw_buffer_load_b128(arsrc, aoff[i] + c * cstride, 0, 0);
            ar[u][i][1] = __builtin_amdgcn_raw_buffer_load_b128(arsrc, aoff[i] + c * cstride + sstride, 0, 0);
        }
    };
    auto compute = [&](auto u_tag, int step) {
        constexpr int u = decltype(u_tag)::value;
        if (chunk_of(step) >= c1) return;    // wave-uniform
#pragma unroll
        for (int j = 0; j < P; ++j)
#pragma unroll
            for (int i = 0; i < MT; ++i) {
                acc[j][i] = mfma16(wr[u][j][0], __builtin_bit_cast(h8, ar[u][i][0]), acc[j][i]);   // C^T tile
                acc[j][i] = mfma16(wr[u][j][1], __builtin_bit_cast(h8, ar[u][i][1]), acc[j][i]);
            }
    };
    // U named sets: set u holds step s with s % U == u (static register indices)
    auto for_sets = [&](auto &&f) {
        f(std::integral_constant<int, 0>{});
        if constexpr (U > 1) f(std::integral_constant<int, 1>{});
        if constexpr (U > 2) f(std::integral_constant<int, 2>{});
        if constexpr (U > 3) f(std::integral_constant<int, 3>{});
    };
    for_sets([&](auto u_tag) {
        constexpr int u = decltype(u_tag)::value;
        if (u < nsteps) load(u_tag, u);
    });
    for (int s0 = 0; s0 < nsteps; s0 += U) {
        for_sets([&](auto u_tag) {
            constexpr int u = decltype(u_tag)::value;
            const int s = s0 + u;
            if (s < nsteps) {
                compute(u_tag, s);
                if (s + U < nsteps) load(u_tag, s + U);
            }
        });
        if (p.trace && tid == 0 && s0 == 0) p.trace[wg * 8 + 1] = wall_clock64();     // (the first sets have landed)
    }
    if (p.trace && tid == 0) p.trace[wg * 8 + 2] = wall_clock64();

    // ---- combine the NW partial tiles through LDS, in wave order ----
    f4 *red = reinterpret_cast<f4 *>(smem);          // [NW][TILES][64] f4
#pragma unroll
    for (int j = 0; j < P; ++j)
#pragma unroll
        for (int i = 0; i < MT; ++i) red[(wave * TILES + j * MT + i) * 64 + lane] = acc[j][i];
    __syncthreads();
    if (p.trace && tid == 0) p.trace[wg * 8 + 4] = wall_clock64();
    const int npanels = p.N >> 4;
    // the GEMM epilogue on the finished sums of one tile: this lane holds columns n .. n + 3 of row m
    auto finish = [&](int tile, f4 v, const float4 *rpre = nullptr) {    // rpre: the residual values, loaded earlier
        const int j = tile / MT, i = tile - j * MT;
        const int m = 16 * i + li;
        const int panel = panel0 + j;
        const int n = panel * 16 + 4 * g;
        const bool live = m < p.M;
        if (p.bias) {
            const float4 bz = *reinterpret_cast<const float4 *>(p.bias + n);
            v[0] += bz.x; v[1] += bz.y; v[2] += bz.z; v[3] += bz.w;
        }
        if (EPI == EPI_GELU) {
#pragma unroll
            for (int r = 0; r < 4; ++r) v[r] = gelu_erf_s(v[r]);
        }
        if (p.residual && live) {
            const float4 rr = rpre ? *rpre : *reinterpret_cast<const float4 *>(p.residual + (int64_t)m * p.ldr + n);
            v[0] += rr.x; v[1] += rr.y; v[2] += rr.z; v[3] += rr.w;
        }
        if (live) {
            if (p.out_f32) *reinterpret_cast<float4 *>(reinterpret_cast<float *>(p.C) + (int64_t)m * p.ldc + n) = make_float4(v[0], v[1], v[2], v[3]);
            else *reinterpret_cast<h4 *>(reinterpret_cast<half_t *>(p.C) + (p.c_tiled ? tiled_off(m, n, p.N) : (int64_t)m * p.ldc + n)) = h4{(half_t)v[0], (half_t)v[1], (half_t)v[2], (half_t)v[3]};
        }
        if (p.xh_out) {                                  // producer side of the row-scale RMSNorm fusion (wave-uniform)
            if (live) *reinterpret_cast<h4 *>(p.xh_out + (p.xh_tiled ? tiled_off(m, n, p.N) : (int64_t)m * p.N + n)) = h4{(half_t)v[0], (half_t)v[1], (half_t)v[2], (half_t)v[3]};
            float q = (v[0] * v[0] + v[1] * v[1]) + (v[2] * v[2] + v[3] * v[3]);
            q += __shfl_xor(q, 16, 64);                  // the four lanes (g = 0..3) of a row hold its 16 columns
            q += __shfl_xor(q, 32, 64);
            if (live && g == 0) p.ssq_out[(int64_t)m * npanels + panel] = q;
        }
    };
    typedef unsigned int u4s __attribute__((ext_vector_type(4)));
    // k-part slabs through a descriptor: written through to memory (sc1) when another workgroup of this launch will read them
    const __amdgpu_buffer_rsrc_t wrs = __builtin_amdgcn_make_buffer_rsrc(
        (void *)p.ws, 0, ksplit > 1 ? (int)((int64_t)ksplit * p.M * p.N * 4) : 0, 0x00020000);
    const bool combine = ksplit > 1 && p.combine_cnt != nullptr;      // (kernel argument: uniform)
    for (int tile = wave; tile < TILES; tile += NW) {    // wave-uniform
        f4 v = red[tile * 64 + lane];
#pragma unroll
        for (int w = 1; w < NW; ++w) v += red[(w * TILES + tile) * 64 + lane];
        if (p.trace && tid == 0 && tile == 0) p.trace[wg * 8 + 5] = wall_clock64();
        if (ksplit == 1) { finish(tile, v); continue; }
        const int j = tile / MT, i = tile - j * MT;
        const int m = 16 * i + li, n = (panel0 + j) * 16 + 4 * g;
        const int off = m < p.M ? (int)((((int64_t)ky * p.M + m) * p.N + n) * 4) : 0x7ffffff0;   // (rows >= M: dropped by the range check)
        if (combine) __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u4s, v), wrs, off, 0, 16);   // aux 16 = sc1
        else __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u4s, v), wrs, off, 0, 0);
    }
    if (p.trace && tid == 0) p.trace[wg * 8 + 3] = wall_clock64();
    if (!combine) return;
    // ---- in-launch combine of the k-parts (cdna_hip_programming.md "In-launch split-K reduction", sc1 form) ----
    // every wave has drained its write-through slab stores; one lane draws a ticket; the workgroup that draws the last one
    // reads all slabs of its column group with sc1 loads (L1 bypassed), sums them in k-part order - the same sums whichever
    // workgroup arrives last - and runs the epilogue; it also re-arms the counter for the next launch.
    // (the residual values of the tiles this wave would finish are requested before the drain: only the workgroup that arrives
    //  last uses them, and nobody writes them before it does - its own epilogue is the only writer of these columns)
    constexpr int TPW = (TILES + NW - 1) / NW;
    float4 rpre[TPW];
#pragma unroll
    for (int t = 0; t < TPW; ++t) {
        const int tile = wave + t * NW;
        const int j = tile / MT, i = tile - j * MT;
        const int m = 16 * i + li, n = (panel0 + j) * 16 + 4 * g;
        rpre[t] = make_float4(0.f, 0.f, 0.f, 0.f);
        if (p.residual && tile < TILES && m < p.M) rpre[t] = *reinterpret_cast<const float4 *>(p.residual + (int64_t)m * p.ldr + n);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    int *flag = reinterpret_cast<int *>(smem + (size_t)NW * TILES * 1024);
    if (p.trace && tid == 0) p.trace[wg * 8 + 6] = wall_clock64();       // (slab stores drained)
    if (tid == 0) *flag = __hip_atomic_fetch_add(p.combine_cnt + blockIdx.x, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __syncthreads();
    if (p.trace && tid == 0) p.trace[wg * 8 + 7] = wall_clock64();       // (ticket drawn)
    if (*flag >= ksplit && tid == 0)                     // a ticket no clean launch can draw: the word was poisoned (an aborted launch)
        __hip_atomic_store(p.combine_cnt + HANDOFF_ERR, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (*flag != ksplit - 1) return;                     // uniform
    if (tid == 0) __hip_atomic_store(p.combine_cnt + blockIdx.x, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    // all slab loads of this wave's tiles are requested together (k-parts in groups of 4), then added in k-part order
    f4 vsum[TPW];
#pragma unroll
    for (int t = 0; t < TPW; ++t) vsum[t] = f4{0.f, 0.f, 0.f, 0.f};
    for (int k0 = 0; k0 < ksplit; k0 += 4) {
        u4s tl[TPW][4];
#pragma unroll
        for (int t = 0; t < TPW; ++t) {
            const int tile = wave + t * NW < TILES ? wave + t * NW : wave;
            const int j = tile / MT, i = tile - j * MT;
            int m = 16 * i + li;
            m = m < p.M ? m : p.M - 1;
            const int n = (panel0 + j) * 16 + 4 * g;
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int kk = k0 + u < ksplit ? k0 + u : ksplit - 1;
                tl[t][u] = __builtin_amdgcn_raw_buffer_load_b128(wrs, (int)((((int64_t)kk * p.M + m) * p.N + n) * 4), 0, 16);
            }
        }
#pragma unroll
        for (int t = 0; t < TPW; ++t)
#pragma unroll
            for (int u = 0; u < 4; ++u)
                if (k0 + u < ksplit) vsum[t] += __builtin_bit_cast(f4, tl[t][u]);
    }
#pragma unroll
    for (int t = 0; t < TPW; ++t)
        if (wave + t * NW < TILES) finish(wave + t * NW, vsum[t], &rpre[t]);
    if (p.trace && tid == 0) p.trace[wg * 8 + 3] = wall_clock64();       // (the combining workgroup: its real end)
}

// ---- launcher ----
struct StreamPlan { int P, ks, nt; };

// P panels per workgroup x ks k-parts with (N / 16 / P) * ks workgroups on at most 256 CUs and at least 3/4 of them.
// ks == 1 writes the finished output (no slabs, no reduce launch); ks > 1 leaves slabs - for a consumer that sums them
// (slab_only: the QKV projection, 3 panels x 2 k-parts) or for splitk_reduce behind the launch (the down projection at
// > 16 rows: 4 panels x 4 k-parts cut the activation bytes per CU by 4, which an in-workgroup k-split cannot).  The activation bytes a
// CU re-reads from L2, 2 M K / ks, must stay within what its L1 takes in beside the weight stream (header): measured
// break-even against the round-2 kernels at ~600 KB for fragment-ordered activations and ~280 KB for row-major ones.
static bool stream_plan(int M, int N, int K, bool slab_only, bool a_tiled, int64_t ws_bytes, StreamPlan &pl) {
    if ((N & 15) || (K & 63) || M < 1 || M > 64) return false;
    const int npanels = N >> 4, chunks = K >> 6;
    int best = 0;
    // (measured and not kept for the QKV shape: 6 panels x 4 k-parts on 4 waves - half the activation bytes per CU, twice the
    //  slabs - 14.4 vs 13.7 us at 64 rows)
    // (5 panels x 4 k-parts: 320-panel outputs - the 13B decoders' hidden size 5120 - on exactly 256 workgroups; up to 3 row tiles:
    //  the partial tiles of 8 waves x 5 panels x 4 row tiles would not fit the LDS)
    const int cand[][2] = {{1, 1}, {3, 2}, {3, 4}, {4, 4}, {5, 4}};    // (ties: the earlier candidate, i.e. the smaller ks)
    // (measured and not kept, round 4, OPUS_STREAM_ROUNDS: twice the k-parts = two rounds of 256 workgroups, after tools/stream_sweep.hip
    //  showed a pure stream running 10-15 % faster on multi-round grids - the batched decode phase 120.3 -> 137.7 ms: twice the
    //  slabs and twice the combine tails cost far more than the better balance returns)
    // (measured and not kept, round 4: the gate / up projection through this kernel - 4 panels (two gate / up pairs) x the whole K per
    //  workgroup, 448 workgroups in 1.75 rounds, silu(g) u + the row scale in the combine - parity-green and 58 us per launch against
    //  gemm_wide_kernel's 45.8: every workgroup re-reads the whole activation matrix from L2, 229 MB beside 235 MB of weights, and a
    //  CU's L1 takes in ~67 GB/s)
    // (measured and not kept, round 5, profiles/r05_decode_ab.txt - the 2-D plans the round-4 review asked about for the wo projection
    //  at 64 rows, isolated, fragment-ordered A: this planner's 1 panel x whole K 13.7 us; 4 panels x 4 k-parts (as down) 15.7 us - a
    //  quarter of the activation bytes per CU, but 16 chunks over 8 waves leave two chunks per wave and the slabs + ticket cost more than
    //  the L1 intake they relieve; 2 panels x 2 k-parts 13.05 us (-5 %: 0.65 us per layer, 0.3 % of the step) - not taken: it would
    //  either break "a row's result does not depend on the rows that share its launch" between <= 16 and > 16 rows or need the pair
    //  plan at every row count.  down as 2 x 2 instead of 4 x 4: 31.2 vs 28.3 us - its 917 KB of activations per CU are back at the
    //  L1 limit.  In the step: 249.8 / 250.7 ms (planner) vs 252.7 / 252.9 (wo 4 x 4) vs 251.8 / 251.7 (wo and down 2 x 2).)
    for (auto &c : cand) {
        const int P = c[0], ks = c[1];
        if (npanels % P) continue;
        if (P == 3 && !slab_only) continue;
        if (P >= 4 && (slab_only || !a_tiled)) continue;
        if (P == 5 && M > 48) continue;
        const int nt = P == 1 ? 1024 : 512;
        if (chunks / ks < nt / 64) continue;              // at least one chunk per wave
        const int wgs = npanels / P * ks;
        if (wgs > 256 || wgs < 192) continue;
        if (ks > 1 && (int64_t)ks * M * N * 4 > ws_bytes) continue;
        if (2ll * M * K / ks > (a_tiled ? 600 : 280) * 1024) continue;
        if (wgs > best) { best = wgs; pl = StreamPlan{P, ks, nt}; }
    }
    return best > 0;
}

static bool stream_off() {
    static const bool off = getenv("OPUS_NO_STREAM") != nullptr;      // A/B aid
    return off || g_knobs.no_stream;
}

// would launch_gemm route this fp16-A GEMM (EPI_NONE / GELU, 16-B aligned strides) to gemm_stream_kernel?
bool gemm_stream_would(int M, int N, int K, int slab_only, int a_tiled, int row_scale, int64_t ws_bytes) {
    StreamPlan pl;
    if (stream_off() || !stream_plan(M, N, K, slab_only != 0, a_tiled != 0, ws_bytes, pl)) return false;
    return !(row_scale && pl.ks == 1);                                // (no row scale in this kernel's own epilogue)
}

bool gemm_stream_ok(const GemmParams &p) {
    if (p.Af || (p.epi != EPI_NONE && p.epi != EPI_GELU)) return false;
    if ((p.ldc & 3) || (p.residual && (p.ldr & 3)) || (p.lda & 7)) return false;
    if (!p.a_tiled && (int64_t)(p.M - 1) * p.lda + p.K >= (1ll << 29)) return false;   // 32-bit buffer offsets
    if (p.a_tiled && p.lda != p.K) return false;
    if ((p.c_tiled && (p.out_f32 || p.ldc != p.N))) return false;
    // a row scale (GemmParams::row_ssq) is applied by whoever sums the slabs, never by this kernel's own epilogue
    return gemm_stream_would(p.M, p.N, p.K, p.slab_only, p.a_tiled, p.row_ssq != nullptr, p.ws_bytes);
}

template <int MT, int P, int NT, int U, int EPI>
static hipError_t launch_stream_t(const GemmParams &p_in, const StreamPlan &pl, hipStream_t s) {
    GemmParams p = p_in;
    constexpr int NW = NT / 64;
    const size_t lds = (size_t)NW * P * MT * 1024 + 16;      // partial tiles + the "last arriver" word of the in-launch combine
    hipError_t ea = ensure_dyn_lds(reinterpret_cast<const void *>(&gemm_stream_kernel<MT, P, NT, U, EPI>), lds);
    if (ea != hipSuccess) return ea;
    const int npanels = p.N >> 4;
    // k-parts with a finished output: combined inside the launch by the workgroup that arrives last (no reduce launch)
    static const bool no_combine = getenv("OPUS_NO_COMBINE") != nullptr;   // A/B aid: slabs + splitk_reduce instead
    const bool combine = pl.ks > 1 && !p.slab_only && p.combine_cnt && !p.row_ssq && !no_combine && !g_knobs.misc[4];
    if (!combine) p.combine_cnt = nullptr;
    // producer side of the row-scale fusion (this kernel's own epilogue: one k-part, or the in-launch combine): fp32 output
    // of the plain epilogue, room for M x N/16 partial sums.  With a split-K reduce behind the launch, the reduce does it.
    const bool own_epilogue = pl.ks == 1 || combine;
    const bool fuse = own_epilogue && EPI == EPI_NONE && p.xh_out && p.ssq_out && p.fused_done && p.out_f32 &&
                      (int64_t)p.M * npanels <= p.ssq_cap;
    if (fuse) { *p.fused_done = 1; if (p.nblk_out) *p.nblk_out = npanels; }
    else p.xh_out = nullptr;
    if (pl.ks > 1) p.row_ssq = nullptr;
    if (p.ks_out) *p.ks_out = pl.ks;
    static const bool trace = getenv("OPUS_STREAM_TRACE") != nullptr;   // tuning aid: per-workgroup section stamps on stderr
    if (trace) {
        static long long *tb = nullptr;
        const int nwg = npanels / P * pl.ks;
        if (!tb) (void)hipMalloc((void **)&tb, (size_t)1024 * 8 * sizeof(long long));
        if (tb && nwg <= 1024) {
            GemmParams q = p;
            q.trace = tb;
            for (int rep = 0; rep < 3; ++rep)      // (the third launch is reported: code and activations warm, weights from HBM if > caches)
                hipLaunchKernelGGL((gemm_stream_kernel<MT, P, NT, U, EPI>), dim3(npanels / P, pl.ks), dim3(NT), lds, s, q, pl.ks);
            (void)hipStreamSynchronize(s);
            std::vector<long long> h((size_t)nwg * 8);
            (void)hipMemcpy(h.data(), tb, h.size() * sizeof(long long), hipMemcpyDeviceToHost);
            long long t0 = h[0];
            for (int i = 0; i < nwg; ++i) t0 = std::min(t0, h[8 * i]);
            std::vector<double> st, fl, ml, en, ba, su, dr, tk;
            for (int i = 0; i < nwg; ++i) {
                st.push_back((h[8 * i] - t0) * 0.01); fl.push_back((h[8 * i + 1] - h[8 * i]) * 0.01);
                ml.push_back((h[8 * i + 2] - t0) * 0.01); en.push_back((h[8 * i + 3] - t0) * 0.01);
                ba.push_back((h[8 * i + 4] - t0) * 0.01); su.push_back((h[8 * i + 5] - t0) * 0.01);
                if (combine) { dr.push_back((h[8 * i + 6] - t0) * 0.01); tk.push_back((h[8 * i + 7] - t0) * 0.01); }
            }
            auto q3 = [](std::vector<double> v, const char *nm) {
                std::sort(v.begin(), v.end());
                fprintf(stderr, "   %-34s min %6.2f  p10 %6.2f  median %6.2f  p90 %6.2f  max %6.2f us\n", nm, v.front(), v[v.size() / 10],
                        v[v.size() / 2], v[v.size() * 9 / 10], v.back());
            };
            fprintf(stderr, "[stream trace] M=%d N=%d K=%d P=%d ks=%d nt=%d wgs=%d tiled=%d combine=%d\n", p.M, p.N, p.K, P, pl.ks, NT, nwg, p.a_tiled, (int)combine);
            q3(st, "start after the first start");
            q3(fl, "start -> first sets multiplied");
            q3(ml, "main loop end, wave 0 (after first start)");
            q3(ba, "all waves' partial tiles in LDS");
            q3(su, "wave 0's first tile summed");
            if (combine) { q3(dr, "slab stores drained"); q3(tk, "ticket drawn"); }
            q3(en, "end (after first start)");
        }
    }
    if (pl.ks & (pl.ks - 1)) return hipErrorInvalidValue;               // (the kernel cuts K with a shift)
    OPUS_LAUNCH(KC_STREAM, (gemm_stream_kernel<MT, P, NT, U, EPI>), dim3(npanels / P, pl.ks), dim3(NT), lds, s, p, pl.ks);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess || pl.ks == 1 || p_in.slab_only || combine) return e;
    return launch_splitk_reduce(p_in, pl.ks, s);                      // (applies bias / residual / row scale, writes xh_out + sums of squares)
}

template <int EPI>
static hipError_t launch_stream_e(const GemmParams &p, const StreamPlan &pl, hipStream_t s) {
    const int mt = cdiv(p.M, 16);
    if (pl.P == 4) {
        switch (mt) {
            case 1: return launch_stream_t<1, 4, 512, 2, EPI>(p, pl, s);
            case 2: return launch_stream_t<2, 4, 512, 2, EPI>(p, pl, s);
            case 3: return launch_stream_t<3, 4, 512, 2, EPI>(p, pl, s);
            case 4: return launch_stream_t<4, 4, 512, 2, EPI>(p, pl, s);
        }
    }
    if (pl.P == 5) {
        switch (mt) {
            case 1: return launch_stream_t<1, 5, 512, 2, EPI>(p, pl, s);
            case 2: return launch_stream_t<2, 5, 512, 2, EPI>(p, pl, s);
            case 3: return launch_stream_t<3, 5, 512, 2, EPI>(p, pl, s);
        }
        return hipErrorInvalidValue;
    }
    if (pl.P == 1) {
        switch (mt) {
            case 1: return launch_stream_t<1, 1, 1024, 2, EPI>(p, pl, s);
            case 2: return launch_stream_t<2, 1, 1024, 2, EPI>(p, pl, s);
            case 3: return launch_stream_t<3, 1, 1024, 2, EPI>(p, pl, s);
            case 4: return launch_stream_t<4, 1, 1024, 2, EPI>(p, pl, s);
        }
    } else {
        switch (mt) {
            case 1: return launch_stream_t<1, 3, 512, 2, EPI>(p, pl, s);
            case 2: return launch_stream_t<2, 3, 512, 2, EPI>(p, pl, s);
            case 3: return launch_stream_t<3, 3, 512, 2, EPI>(p, pl, s);
            case 4: return launch_stream_t<4, 3, 512, 2, EPI>(p, pl, s);
        }
    }
    return hipErrorInvalidValue;
}

hipError_t launch_gemm_stream(const GemmParams &p, hipStream_t s) {
    StreamPlan pl;
    if (!gemm_stream_ok(p) || !stream_plan(p.M, p.N, p.K, p.slab_only != 0, p.a_tiled != 0, p.ws_bytes, pl)) return hipErrorInvalidValue;
    if (p.epi == EPI_GELU) return launch_stream_e<EPI_GELU>(p, pl, s);
    return launch_stream_e<EPI_NONE>(p, pl, s);
}

}  // namespace opus
