// Shared declarations for libopus_pllm.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <stdint.h>

namespace opus {

// The 16-bit operand type of the build: IEEE fp16 (default: the reference's unquantised dtype, model/builder.py:57) or - with
// -DOPUS_BF16, build.py's second library libopus_pllm_bf16.so - bfloat16 (SURVEY 8(d) "bf16 switchable": Llama-3 checkpoints are
// bf16-native).  Every kernel is written against half_t / h8 and mfma16(); accumulation, residual stream, norms and softmax are
// fp32 in both builds.
#ifdef OPUS_BF16
typedef __bf16 half_t;
typedef __bf16 h8 __attribute__((ext_vector_type(8)));
typedef __bf16 h4 __attribute__((ext_vector_type(4)));
typedef __bf16 h2 __attribute__((ext_vector_type(2)));
#else
typedef _Float16 half_t;
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef _Float16 h4 __attribute__((ext_vector_type(4)));
typedef _Float16 h2 __attribute__((ext_vector_type(2)));
#endif
typedef float f4 __attribute__((ext_vector_type(4)));
// D = A (16 x 32) B (32 x 16) + C on the matrix cores, operands in the build's 16-bit type
__device__ __forceinline__ f4 mfma16(h8 a, h8 b, f4 c) {
#ifdef OPUS_BF16
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
#else
    return __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0);
#endif
}

enum Epi { EPI_NONE = 0, EPI_GELU = 1, EPI_SILU_GU16 = 2 };
// first unit of k-part `part` of `parts` over `n` units (n * parts < 2^31): 32-bit, and free when there is one part - the
// 64-bit form of this was ~100 scalar instructions in front of the first load of every weight-streaming kernel
__device__ __forceinline__ int kpart_begin(int n, int part, int parts) {
    return parts == 1 ? (part ? n : 0) : (int)((unsigned)(n * part) / (unsigned)parts);
}
// Every GemmParams field a weight-streaming kernel reads before its first load, fetched in ONE scalar batch (the compiler
// otherwise loads the ~330-byte argument block piecemeal: up to six dependent scalar round trips before the first weight load)
#define OPUS_ARGS_ONE_BATCH(p)                                                                                                        \
    asm volatile("" ::"s"((p).A), "s"((p).Af), "s"((p).W), "s"((p).C), "s"((p).bias), "s"((p).residual), "s"((p).ws), "s"((p).xh_out),  \
                 "s"((p).ssq_out), "s"((p).row_ssq), "s"((p).lda), "s"((p).ldc), "s"((p).ldr), "s"((p).M), "s"((p).N), "s"((p).K),     \
                 "s"((p).out_f32), "s"((p).norm_eps), "s"((p).row_nblk), "s"((p).no_rot), "s"((p).c_tiled))
// M up to which the weight-streaming skinny kernel CAN be used (LDS-staged activations, fused RMSNorm); the mid / wide
// kernels take over above skinny_max_m().
constexpr int SKINNY_MAX_M_CAP = 16;
int skinny_max_m();   // rows up to which the skinny kernel is used (default 4, at most 16; OPUS_SKINNY_MAX_M tunes it)
#define SKINNY_MAX_M skinny_max_m()
// M up to which the mid kernel (LDS-shared activations, per-wave weight stream, fused RMSNorm) is used
constexpr int MID_MAX_M = 64;   // (65..128 rows measured faster on the split-K tile kernel)
// kernel classes of the per-launch timing (opus_timing_get): one per kernel family
enum KClass { KC_SKINNY = 0, KC_MID, KC_WIDE, KC_RING, KC_PP, KC_TILE, KC_REDUCE, KC_ATTN_PREFILL, KC_ATTN_DECODE, KC_NORM,
              KC_OTHER, KC_STREAM, KC_COUNT };
// phases of the path a launch belongs to (set by the entry points of api.cpp)
enum Phase { PH_ENCODE = 0, PH_PROJECT, PH_SPLICE, PH_PREFILL, PH_DECODE, PH_OTHER, PH_COUNT };

// Measurement hook.  While a LaunchEvents record is armed (thread-local, set by api.cpp in timing mode only), the next
// principal kernel launch is dispatched with hipExtLaunchKernelGGL so that (main0, main1) carry the dispatch's own start /
// end timestamps - the interval rocprofv3 --kernel-trace reports - and a split-K reduce launched behind it uses (aux0, aux1).
struct LaunchEvents {
    hipEvent_t main0 = nullptr, main1 = nullptr, aux0 = nullptr, aux1 = nullptr;
    int main_class = -1;
    bool main_used = false, aux_used = false;
    double aux_bytes = 0.0;
};
extern thread_local LaunchEvents *tl_launch_ev;
#define OPUS_LAUNCH(klass, kernel, grid, block, lds, stream, ...)                                                        \
    do {                                                                                                                  \
        ::opus::LaunchEvents *ev_ = ::opus::tl_launch_ev;                                                                 \
        if (ev_ && (klass) != ::opus::KC_REDUCE && !ev_->main_used) {                                                     \
            ev_->main_used = true;                                                                                        \
            ev_->main_class = (klass);                                                                                    \
            hipExtLaunchKernelGGL(kernel, grid, block, lds, stream, ev_->main0, ev_->main1, 0, __VA_ARGS__);              \
        } else if (ev_ && (klass) == ::opus::KC_REDUCE && !ev_->aux_used) {                                               \
            ev_->aux_used = true;                                                                                         \
            hipExtLaunchKernelGGL(kernel, grid, block, lds, stream, ev_->aux0, ev_->aux1, 0, __VA_ARGS__);                \
        } else {                                                                                                          \
            hipLaunchKernelGGL(kernel, grid, block, lds, stream, __VA_ARGS__);                                            \
        }                                                                                                                 \
    } while (0)
// hipFuncAttributeMaxDynamicSharedMemorySize, remembered per (device, kernel): a second device in the same process gets its own call
hipError_t ensure_dyn_lds(const void *fn, size_t bytes);

// C[M,Nout] = epi(A[M,K] * W[N,K]^T + bias) (+ residual).  fp16 operands, fp32 accumulate.
// EPI_SILU_GU16: W rows come in 32-row groups [16 gate | 16 up]; Nout = N/2, out = silu(g) * u.
// One rotary pair, with the contraction spelled out (one rounded product, one fused multiply-add) so that every kernel
// that rotates - the stand-alone kernels and the fused GEMM epilogue - produces the same bits from the same inputs.
// The results are pinned in fp32 registers: left to itself the compiler folds a following conversion to fp16 into the
// multiply-add (v_fma_mixlo_f16: ONE rounding, straight to fp16) in some kernels and not in others - a 1-ulp difference in
// 6e-5 of the elements; fp32 first, then fp16, is also what the reference's fp32 rotary followed by .to(fp16) does.
__device__ __forceinline__ void rotate_pair(float a, float b, float c, float sn, float &lo, float &hi) {
    lo = __builtin_fmaf(a, c, -__fmul_rn(b, sn));
    hi = __builtin_fmaf(b, c, __fmul_rn(a, sn));
    asm volatile("" : "+v"(lo), "+v"(hi));
}

// offset (in elements) of element (row, k) of a fragment-ordered [rows, K] fp16 matrix: 16-row x 64-k blocks of 2 KB in MFMA
// operand order [k-step][lane = 16 ((k % 32) / 8) + row % 16][8] - the panel-tiled weight layout with rows for output columns
__host__ __device__ __forceinline__ int64_t tiled_off(int row, int k, int K) {
    return ((int64_t)(row >> 4) * (K >> 6) + (k >> 6)) * 1024 + ((k & 63) >> 5) * 512 + ((((k & 31) >> 3) << 4) + (row & 15)) * 8 + (k & 7);
}

constexpr int HANDOFF_WORDS = 1024, HANDOFF_ERR = 1023;
constexpr int HANDOFF_SPIN_LIMIT = 1 << 19;   // polls (s_sleep(4) + one L2 round trip each: >= 0.3 s) before a waiter gives up

struct GemmParams {
    const half_t *A;        // fp16 activations [M,K] (row-major, lda), or
    const float *Af;        // fp32 residual stream [M,K]: fused RMSNorm (skinny kernel only), else nullptr
    float norm_eps;
    int64_t lda;
    const half_t *W;        // panel-tiled [N,K] (see gemm.hip)
    int M, N, K;
    const float *bias;      // [N] or nullptr
    const float *residual;  // fp32 [M,Nout] or nullptr (may alias C when C is fp32)
    int64_t ldr;
    void *C;
    int64_t ldc;
    int out_f32;            // 1: C is fp32, 0: fp16
    int epi;
    float *ws;              // split-K workspace (fp32 partial slabs) or nullptr
    int64_t ws_bytes;
    // Row-scale fusion (RMSNorm between a split-K GEMM and the wide kernel that consumes its output):
    //  producer side - the flat split-K reduce (EPI_NONE, fp32 output, N % 256 == 0) also writes the row as fp16 to xh_out
    //  and the sum of squares of each 256-column block to ssq_out[m * N/256 + j], then sets *fused_done;
    //  consumer side - gemm_wide_kernel multiplies row m of its result by rsqrt(sum_j row_ssq[m * row_nblk + j] / K + norm_eps).
    //  The blocks are 256 columns wide when a split-K reduce (or the embedding kernel) produced them and 16 columns wide when
    //  gemm_stream_kernel did: *nblk_out (HOST) receives the number of blocks per row the producer wrote (at most ssq_cap floats
    //  in all), the consumer gets it back as row_nblk.
    half_t *xh_out;
    float *ssq_out;
    int *fused_done;
    const float *row_ssq;
    int row_nblk;
    int64_t ssq_cap = 0;
    int *nblk_out = nullptr;
    // slab_only: when the launch splits K over workgroups, leave the raw fp32 k-part slabs in ws ([ks][M][N], no epilogue) and
    // report their number in *ks_out instead of launching the reduce - the consumer (attn_decode_kernel) sums them, applies the
    // row scale and the bias itself; *ks_out = 1 means the kernel wrote the finished output as usual.
    int slab_only = 0;
    int *ks_out = nullptr;
    int force_wide = 0;     // route a narrow output (N < 16384) through gemm_wide_kernel + k-parts
    // ESM rotary fused into the epilogue of the QKV projection (gemm_pp_kernel, head_dim 64, fp16 output): columns
    // < rope_cols are rotated in heads of 64 - pairs (d, d + 32), angle of position (row % rope_T) from the [T][32][2]
    // (cos, sin) table - after the usual rounding to fp16, columns < rope_qcols being scaled by rope_qscale first: the same
    // arithmetic as esm_rope_kernel on the stored projection.  *rope_done = 1 when the launch applied it.
    const float *rope_cs = nullptr;
    const int32_t *rope_pos = nullptr;      // token-packed batches: position of every row (instead of row % rope_T)
    int rope_T = 0, rope_cols = 0, rope_qcols = 0;
    float rope_qscale = 1.0f;
    int *rope_done = nullptr;
    // Fragment-ordered ("tiled") activation matrices of the batched decode step (gemm_stream.hip, "Activation layout"):
    // element (row, k) of an [rows, K] matrix at tiled_off(row, k, K).  a_tiled: A is stored that way (gemm_stream_kernel only);
    // xh_tiled / c_tiled: write xh_out / the fp16 output C that way (for a consumer that will read it with a_tiled).
    int a_tiled = 0, xh_tiled = 0, c_tiled = 0;
    // gemm_stream_kernel with k-parts and a finished output: ticket counters (one int per column group, all zero between
    // launches) of the in-launch combine; nullptr: slabs + a split-K reduce launch.  HANDOFF_WORDS ints: [0, 512) tickets of
    // gemm_stream_kernel, [512, HANDOFF_ERR) (ticket, flag) pairs of gemm_pp_kernel's two-part tail tiles, [HANDOFF_ERR] the
    // error word a hand-off sets when its bounded wait runs out (the host reads it at its next synchronisation: OPUS_EHIP).
    // Everything below HANDOFF_ERR is zeroed at the head of every encode / prefill / decode step.
    int *combine_cnt = nullptr;
    // LayerNorm fused around the big tiled GEMM (gemm_pp_kernel; the encoder's pre-LN blocks).  With the norm
    // weight folded into the consumer's weight W' = W diag(gamma) and beta into its bias c2 = W beta + b,
    //   LN(x) W^T + b  =  rstd (x W'^T - mu s) + c2,   s[n] = sum_k W'[n][k]:
    //  producer side (fp32 output + residual, EPI_NONE, N % 256 == 0): the epilogue also writes fp16(x) to xh_out (row-major)
    //   and, per row and 64-column slab, (sum x, sum x^2) to ln_part[(m * N/64 + slab) * 2]; sets *ln_done;
    //  consumer side (fp16 output, EPI_NONE / GELU): A is fp16(x) as it stands, ln_stat[m] = (mu, rstd), ln_colsum = s,
    //   bias = c2; the epilogue applies the affine form above before the activation / the fused rotary.
    //  RMSNorm (the decoder prefill) is the same with mu = 0: ln_colsum = nullptr, any epilogue incl. the gate / up pair.
    float *ln_part = nullptr;
    int *ln_done = nullptr;
    const float *ln_stat = nullptr;
    const float *ln_colsum = nullptr;
    int pp_gm = 8;                // gemm_pp_kernel: tile rows per rasterisation group (8 = an 8 x 4 tile patch per XCD)
    int stagger = 0;              // gemm_pp_kernel: late start of half of the first round, in units of ~4 us (see the kernel)
    int no_rot = 0;               // A/B aid (OPUS_NO_KROT): weight-streaming kernels walk k from chunk 0 in every workgroup
    long long *trace = nullptr;   // tuning aid (OPUS_PP_TRACE): gemm_pp_kernel writes 4 wall-clock stamps per workgroup
};

// Flash-style attention over strided Q/K/V (fp16).  Q(b,h,t,:) = Q + b*q_sb + t*q_st + h*HD etc.;
// kv head of q head h is h / group.  Key j of batch row b is visible to query i iff
// kstart[b] <= j < kend[b] and (!causal || j <= i).  O is [B, T, heads*HD] row-major fp16.
struct AttnParams {
    const half_t *Q, *K, *V;
    int64_t q_sb, q_st, k_sb, k_st, v_sb, v_st;
    half_t *O;
    int64_t o_sb, o_st;
    const int32_t *kstart;  // [B] or nullptr (= 0)
    const int32_t *kend;    // [B] or nullptr (= T)
    int B, T, heads, group, head_dim, causal;
    float scale;
    // Token-packed ("varlen") batches: cu[B + 1] row offsets into Q / K / V / O, whose rows are the batch rows' tokens back to
    // back with no padding (q_sb / k_sb / v_sb / o_sb unused).  Row b has cu[b + 1] - cu[b] tokens, all of them visible keys
    // (kstart / kend unused); T = the longest row (sizes the grid).
    const int32_t *cu = nullptr;
    int q_trim = 0;         // token-packed only: 1 = the first and the last token of every row are keys but not queries (last ESM-2 layer)
};

hipError_t launch_gemm(const GemmParams &p, hipStream_t s, int *klass);
// gemm_stream.hip: one-launch weight streaming for narrow outputs at 5..64 rows (K cut over the waves of a workgroup)
hipError_t launch_splitk_reduce(const GemmParams &p, int ks, hipStream_t s);   // gemm.hip: sums the k-part slabs in p.ws, applies the epilogue
bool gemm_stream_ok(const GemmParams &p);
bool gemm_stream_would(int M, int N, int K, int slab_only, int a_tiled, int row_scale, int64_t ws_bytes);
hipError_t launch_gemm_stream(const GemmParams &p, hipStream_t s);
// Run-time tuning knobs (A/B aids; opus_debug_knob sets them, environment variables give the defaults)
struct Knobs {
    int no_stream = 0;      // 1: narrow decode GEMMs take the round-2 kernels (gemm_wide / gemm_ring / gemm_mid + reduce)
    int no_ln_fusion = 0;   // 1: stand-alone normalisation kernels instead of the norms fused around gemm_pp_kernel
    int debug_a_tiled = 0;  // 1: opus_debug_gemm takes A in fragment order (GemmParams::a_tiled)
    int pp_gm = 8;          // tile-rows per group of the gemm_pp / gemm_ring rasterisation
    int enc_full_last_layer = 0;   // 1: the packed encoder's last layer computes the <cls> / <eos> rows too (AttnParams::q_trim off)
    int misc[8] = {0, 0, 0, 0, 0, 0, 0, 0};   // scratch knobs for experiments
};
extern Knobs g_knobs;
// rstd[r] = rsqrt(sum_j ssq[r * nblk + j] / K + eps) for rows r < rows, computed by the whole workgroup (NWAVES waves) into
// LDS `out`: every load of a wave's rows is requested before the first sum (one round trip, not nblk / 4)
template <int NWAVES, int ROWS>
__device__ __forceinline__ void block_row_rstd(const float *__restrict__ ssq, int M, int nblk, float inv_k, float eps, float *out,
                                               int wave, int lane) {
    constexpr int RW = (ROWS + NWAVES - 1) / NWAVES;
    float part[RW];
#pragma unroll
    for (int r = 0; r < RW; ++r) part[r] = 0.f;
    for (int j0 = 0; j0 < nblk; j0 += 256) {
        float t[RW][4];
#pragma unroll
        for (int r = 0; r < RW; ++r) {
            int row = wave + NWAVES * r;
            row = row < M ? row : M - 1;
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int j = j0 + 64 * u + lane;
                t[r][u] = ssq[(int64_t)row * nblk + (j < nblk ? j : nblk - 1)];
                t[r][u] = j < nblk ? t[r][u] : 0.f;
            }
        }
#pragma unroll
        for (int r = 0; r < RW; ++r) part[r] += (t[r][0] + t[r][1]) + (t[r][2] + t[r][3]);
    }
#pragma unroll
    for (int r = 0; r < RW; ++r) {
        float q = part[r];
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) q += __shfl_xor(q, o, 64);
        const int row = wave + NWAVES * r;
        if (lane == 0 && row < ROWS) out[row] = rsqrtf(q * inv_k + eps);
    }
}
bool gemm_goes_wide(int M, int N);
bool gemm_goes_pp(int M, int N);     // would launch_gemm route an fp16-A GEMM of this shape to gemm_pp_kernel (the big tiled kernel)?   // would launch_gemm route an fp16-A GEMM of this shape to gemm_wide_kernel?
hipError_t launch_attn_prefill(const AttnParams &p, hipStream_t s);

// norm.hip
hipError_t launch_layernorm(const float *x, const float *w, const float *b, float eps, int64_t rows, int D,
                            half_t *out_h, float *out_f, hipStream_t s);   // (w == b == nullptr: (x - mu) rstd, no affine part)
// (mu, rstd) per row from the per-slab (sum x, sum x^2) partials the LayerNorm-producing GEMM epilogue left (GemmParams::ln_part)
// (rms != 0: RMSNorm - (0, rsqrt(mean x^2 + eps)))
hipError_t launch_ln_finalize(const float *part, int64_t rows, int nslab, int D, float eps, int rms, float *stat, hipStream_t s);
hipError_t launch_rmsnorm(const float *x, const float *w, float eps, int64_t rows, int D, half_t *out,
                          hipStream_t s);
hipError_t launch_l2norm(const float *x, int64_t rows, int D, half_t *out, hipStream_t s);
hipError_t launch_f2h(const float *x, int64_t n, half_t *out, hipStream_t s);
hipError_t launch_masked_mean(const float *h, const int32_t *lens, int B, int T, int D, float *out,
                              hipStream_t s);
// token-packed forms (cu[B + 1] row offsets; Tmax = the longest row): embedding, pooling, and the row -> position table
hipError_t launch_masked_mean_packed(const float *h, const int32_t *cu, int B, int D, float *out, hipStream_t s);
hipError_t launch_esm_embed_packed(const int32_t *tok, const half_t *emb, const int32_t *cu, int B, int Tmax, int D, float *x,
                                   int32_t *pos, hipStream_t s);

// elementwise.hip
hipError_t launch_esm_embed(const int32_t *tok, const half_t *emb, int B, int T, int D, float *x, hipStream_t s);
hipError_t launch_esm_rope(half_t *qkv, const float *cs, int B, int T, int heads, int hd, float qscale,
                           hipStream_t s, const int32_t *pos = nullptr);   // pos: per-row positions (packed batches; B * T rows)
hipError_t launch_dec_rope_cache(half_t *qkv, const float *cs, const int32_t *kstart, int B, int T, int nh,
                                 int nkv, int hd, half_t *kc, half_t *vc, int64_t cache_sb, int64_t cache_sh,
                                 hipStream_t s);
hipError_t launch_h2f(const half_t *in, float *out, int64_t n, hipStream_t s);
// x[b,:] = fp32(emb[tok[b]]); optionally also the fp16 copy xh and the per-256-column sums of squares ssq[b][H/256] (the
// producer side of the row-scale RMSNorm fusion, GemmParams::row_ssq)
// ... and (cs != nullptr) the rotary rows of this step: cs_row[b][d] = cs[T0 + *step - kstart[b]][d], d < half (float2 each)
hipError_t launch_embed_tokens(const int32_t *tok, const half_t *emb, int B, int H, int V, float *x, half_t *xh, float *ssq,
                               int xh_tiled, const float *cs, const int32_t *kstart, const int32_t *step, int T0, int half,
                               float *cs_row, int *cnt, int ncnt, hipStream_t s);
hipError_t launch_add_pos(float *x, const half_t *pos, const int32_t *kstart, const int32_t *step, int t0, int B, int Tq,
                          int H, int max_idx, hipStream_t s);
hipError_t launch_take_last(const float *x, int B, int T, int H, float *out, hipStream_t s);
hipError_t launch_relu_h(half_t *x, int64_t n, hipStream_t s);
hipError_t launch_fill_synth(void *dst, int dtype, int64_t rows, int64_t cols, uint64_t seed, float std,
                             float mean, int64_t rb, int64_t rs, int64_t ro, int tiled, uint64_t fold_seed,
                             float fold_std, float fold_mean, hipStream_t s);
hipError_t launch_argmax_partial(const float *logits, int B, int V, float *pval, int32_t *pidx, hipStream_t s);
hipError_t launch_tile_weight(const half_t *src, half_t *dst, int64_t N, int64_t K, hipStream_t s);
hipError_t launch_lora_merge(half_t *W, const half_t *A, const half_t *B, float scale, int64_t out_f,
                             int64_t in_f, int r, hipStream_t s);
hipError_t launch_splice_plan(const int64_t *ids, const uint8_t *mask, int B, int Tt, int n_tok, int max_len,
                              int V, int32_t *plan, hipStream_t s);
hipError_t launch_splice_fill(const int64_t *ids, const uint8_t *mask, int B, int Tt, const half_t *prot,
                              int n_tok, int H, int V, const half_t *emb, const int32_t *plan, int Tout,
                              int left_pad, half_t *out, uint8_t *mask_out, int32_t *pos_out, hipStream_t s);
// (thr_out != nullptr: no draw; the rows' keep thresholds on p = exp(l / T - max / T) are written there - beam-sample)
hipError_t launch_sample_select(const float *logits, int B, int V, float temperature, float top_p, int top_k, const uint64_t *seed,
                                const int32_t *step, float *pmax, int32_t *pidx, float *cand_p, int32_t *cand_i,
                                int32_t *cand_n, float *zpart, float *spart, int32_t *chosen, float *thr_out, hipStream_t s);
constexpr int APART = 64;     // parts a logits row is cut into by argmax_partial / sample_stage1 (pmax[row * APART + part])
__device__ __forceinline__ uint64_t splitmix64(uint64_t x) {
    x += 0x9E3779B97F4A7C15ull;
    x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
    x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
    return x ^ (x >> 31);
}
hipError_t launch_argmax_step(const float *pval, const int32_t *pidx, const int32_t *chosen, int B, const int32_t *eos, int n_eos, int pad_id,
                              int32_t *finished, int32_t *out_ids, int max_new, const int32_t *step,
                              int32_t *next_tok, int32_t *n_unfinished, const int32_t *stop, int n_stop, hipStream_t s);
hipError_t launch_step_advance(int32_t *step, hipStream_t s);
// beam.hip: best M of the K V continuations per batch row (log_softmax + running scores), cache rows of the surviving beams
hipError_t launch_beam_topk(const float *logits, const float *run, int B, int K, int V, int M, float *lse, float *out_s,
                            int32_t *out_i, hipStream_t s);
// beam-sample: M continuations per batch row drawn without replacement from softmax over the K filtered rows' accumulated
// log-probabilities (thr: launch_sample_select's thresholds, pmax: its per-part maxima), in the order drawn
hipError_t launch_beam_sample(const float *logits, const float *run, int B, int K, int V, int M, float temperature, const float *pmax,
                              const float *thr, int min_keep, float *kth_s, int32_t *kth_i, uint64_t seed, int step, float *lse,
                              float *out_s, int32_t *out_i, hipStream_t s);
hipError_t launch_kv_gather_rows(half_t *cache, half_t *tmp, const int32_t *idx, int R, int64_t row_halfs, int nsub, int64_t sub_halfs,
                                 const int32_t *step, int T0, int hd, int back, hipStream_t s);
hipError_t launch_upload_i32(const int32_t *h, int n, int32_t *dst, hipStream_t s);   // host ints -> device through kernel arguments
hipError_t launch_mask_to_kstart(const uint8_t *mask, int B, int T, int32_t *kstart, hipStream_t s);

// attn_decode.hip : rope(q,k at the new slot) + cache append + single-query attention
struct AttnDecodeParams {
    const half_t *qkv;          // [B][(nh + 2 nkv) hd] fp16 projection output of the new tokens, or nullptr when slabs != nullptr
    // fused split-K reduce of the QKV GEMM: element (b, col) = (sum_k slabs[k * slab_stride + b * ld + col]) * rstd(b) + bias[col],
    // rstd(b) = rsqrt(sum_j row_ssq[b * row_nblk + j] / K + eps) when row_ssq != nullptr (the fused RMSNorm row scale), else 1
    const float *slabs;
    int ks;
    int64_t slab_stride;
    const float *row_ssq;
    int row_nblk;
    float eps;
    int K;
    const float *bias;
    const float *cs_row;        // [B][hd / 2][2] (cos, sin) of each row's position in THIS step (written by the embedding kernel)
    const int32_t *kstart, *step;   // step[0] = the step counter, step[1] = the prompt length T0 (api.cpp reset_step)
    int T0, nh, nkv;                // T0 < 0: read step[1] (a captured decode step then serves any prompt length)
    half_t *kc, *vc;            // this layer's cache
    int64_t cache_sb, cache_sh;
    int ctx_cap;
    float scale;
    half_t *out;                // [B][nh hd]
    int out_tiled = 0;          // write `out` in fragment order (tiled_off) for a GemmParams::a_tiled consumer
    long long *trace = nullptr; // tuning aid (OPUS_ATTN_TRACE): 8 wall-clock stamps per workgroup
};
hipError_t launch_attn_decode(const AttnDecodeParams &p, int B, int hd, hipStream_t s);

static inline int cdiv(int64_t a, int64_t b) { return (int)((a + b - 1) / b); }

}  // namespace opus
