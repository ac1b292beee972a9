/*
 * opus_pllm.h - C ABI of the MI355X-native multi_modality_v1 inference path (libopus_pllm.so).
 *
 * The reference (Fanchuana/OPUS-PLLM) has no plugin / FFI interface: its boundary is the Python call
 * surface used by the eval scripts.  This header is the boundary a binding for that surface
 * uses; every entry point cites the reference code it replaces.  The Python shim in
 * opus-pllm_amd/_cabi.py binds exactly these symbols with ctypes (INTEGRATION.md shows the stub).
 *
 * Conventions
 *   - extern "C", plain C types, opaque opus_ctx*; no C++ exception crosses the boundary.
 *   - Every function returns 0 (OPUS_OK) or a negative error class; the message is available from
 *     opus_last_error() (thread-local).
 *   - All pointers named d_* are DEVICE pointers owned by the caller (PyTorch keeps ownership and
 *     lifetime of weights and I/O buffers); the library owns only the context, its workspace and
 *     its KV cache, all sized at opus_ctx_create from the config's capacity fields.
 *   - All work is ordered on the caller's hipStream_t (passed as void*; NULL = default stream).
 *     No hidden device synchronisation, except where a function returns a HOST scalar that depends
 *     on device data (documented per function: it synchronises the given stream once).
 *   - A context is bound to one device and is not thread-safe: one host thread drives a context at a time.  A process may
 *     hold several contexts on the same device that BIND THE SAME weight pointers (read-only; each context owns its workspace,
 *     KV cache, decode graph and hand-off words): two batches in flight per GPU (`model.new_context()`, one host thread and one
 *     stream per context).  One process per GPU as in the reference (model/builder.py:41).
 *   - Kernels that combine split-K parts INSIDE one launch wait only for workgroups that are already running, with a bounded
 *     wait; a wait that runs out (ticket words poisoned by an aborted launch) sets a device error word instead of hanging, which
 *     the next call that synchronises (opus_generate_*, opus_check_error) returns as OPUS_EHIP.  The words are re-zeroed at
 *     the head of every encode / projector / prefill / decode step.
 */
#ifndef OPUS_PLLM_H
#define OPUS_PLLM_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define OPUS_ABI_VERSION 10

enum opus_status {
    OPUS_OK = 0,
    OPUS_EBADARG = -1,      /* null pointer, bad enum, negative size */
    OPUS_ESHAPE = -2,       /* shape exceeds the context capacity or violates a kernel granule */
    OPUS_EHIP = -3,         /* a HIP runtime call failed */
    OPUS_ERCCL = -4,        /* reserved: collective failure */
    OPUS_EUNSUPPORTED = -5, /* valid request the build does not implement */
    OPUS_ESTATE = -6        /* missing weights / call out of order */
};

enum opus_dtype { OPUS_F16 = 0, OPUS_F32 = 1, OPUS_I32 = 2, OPUS_I64 = 3, OPUS_U8 = 4 };

/* Shapes of the path.  The reference hard-codes or threads these through a global class
 * (model/builder.py:24-28, protein_projector/builder.py:7-13, protein_mlp/builder.py:11-15,
 * cstp_v3/modelling.py:21); here they are explicit.  Field order == opus-pllm_amd/config.py. */
typedef struct opus_config {
    int32_t enc_layers, enc_dim, enc_heads, enc_ffn, enc_vocab;
    float enc_ln_eps, enc_rope_theta;
    int32_t has_protein_projector, proj_dim, n_prot_tokens, switch_depth;
    int32_t dec_layers, dec_dim, dec_heads, dec_kv_heads, dec_head_dim, dec_ffn, dec_vocab;
    float dec_rms_eps, dec_rope_theta;
    int32_t max_batch, max_enc_tokens, max_prompt, max_new_tokens;
    /* decoder family (model/builder.py:60-92): dec_arch 0 = Llama / Qwen2 (dec_qkv_bias: q/k/v biases),
     * 1 = OPT / Galactica with do_layer_norm_before (learned positions [dec_max_pos + 2, dim], LayerNorm, biased
     * projections, fc1 - activation - fc2; dec_act 0 = GELU (Galactica), 1 = ReLU (facebook/opt-*); dec_rms_eps is then the LayerNorm epsilon) */
    int32_t dec_arch, dec_qkv_bias, dec_act, dec_max_pos;
} opus_config;

typedef struct opus_ctx opus_ctx;

int opus_abi_version(void);
/* The 16-bit operand type of THIS build of the library: 0 = IEEE fp16 (libopus_pllm.so: the reference's unquantised dtype,
 * model/builder.py:57 `torch_dtype=torch.float16`), 1 = bfloat16 (libopus_pllm_bf16.so, built from the same sources with
 * -DOPUS_BF16: SURVEY 8(d) "bf16 switchable").  Wherever this header says "fp16" for a matrix, an activation or the KV cache it
 * means this type (dtype tag OPUS_F16 = "the build's 16-bit type"); accumulation, residual stream, norms, softmax and logits
 * are fp32 in both builds. */
int opus_operand_dtype(void);
const char *opus_last_error(void);

/* Bytes of device memory opus_ctx_create will allocate for this config (workspace + KV cache). */
int64_t opus_workspace_bytes(const opus_config *cfg);

/* Replaces the module construction of load_pretrained_model / initialize_protein_modules
 * (model/builder.py:29-131, model/opus_arch.py:46-90): creates the per-process context on `device`. */
int opus_ctx_create(const opus_config *cfg, int device, opus_ctx **out);
int opus_ctx_destroy(opus_ctx *ctx);

/* Bind one weight tensor (borrowed device pointer).  Names and layouts: DESIGN.md "Weights in HBM"
 * (fused [q;k;v] rows, gate/up interleaved in 16-row groups, RMSNorm weights folded, panel-tiled).  Replaces the state-dict loads of
 * model/builder.py:60-65,107-111 and opus_arch.py:81-90.  fp16 for matrices, fp32 for vectors. */
int opus_bind_weight(opus_ctx *ctx, const char *name, const void *d_ptr, int dtype, int ndim,
                     const int64_t *shape);
/* 0 when every tensor the config requires is bound; otherwise OPUS_ESTATE and the first missing
 * name in opus_last_error(). */
int opus_weights_ready(opus_ctx *ctx);

/* Row L1: PeftModel.merge_and_unload (model/builder.py:107-109): W[out,in] += scale * B[out,r] A[r,in],
 * fp16 weights, fp32 accumulation, in place. */
int opus_lora_merge(void *d_W, const void *d_A, const void *d_B, float scale, int64_t out_f, int64_t in_f,
                    int32_t r, void *stream);

/* Deterministic synthetic tensor fill (no checkpoints exist offline; opus-pllm_amd/synth.py is the
 * NumPy twin, bit-identical).  Element (row, col) of the LOGICAL [rows, cols] tensor goes to dst row
 * (row / row_block) * row_stride + row_off + row % row_block; tiled != 0 writes the panel-tiled GEMM
 * weight layout (DESIGN.md "Weights in HBM"); fold_std/fold_mean != 0 multiplies column k by element k
 * of the synthetic vector (fold_seed, fold_std, fold_mean): an RMSNorm weight folded into the
 * projection that consumes the normalised activations. */
int opus_fill_synth(void *d_dst, int dtype, int64_t rows, int64_t cols, uint64_t tensor_seed, float std,
                    float mean, int64_t row_block, int64_t row_stride, int64_t row_off, int32_t tiled,
                    uint64_t fold_seed, float fold_std, float fold_mean, void *stream);

/* Load-time re-layout of one GEMM weight: row-major fp16 W[N,K] (nn.Linear layout) -> the panel-tiled
 * layout the kernels stream (16-row x 64-k blocks in MFMA B-fragment order).  N % 16 == 0, K % 64 == 0. */
int opus_tile_weight(const void *d_src, void *d_dst, int64_t N, int64_t K, void *stream);

/* Rows E1-E4: ProteinSeqEmbeddingExtractor.get_protein_seq_embeddings (cstp_v3/modelling.py:37-57):
 * tokens int32 [B,T] (<cls> seq <eos>, pad = 1), lens int32 [B] (incl. <cls>,<eos>) ->
 * pooled fp32 [B, enc_dim] = mean over residues of representations[enc_layers]. */
int opus_esm2_encode(opus_ctx *ctx, const int32_t *d_tokens, const int32_t *d_lens, int32_t B, int32_t T,
                     float *d_pooled, void *stream);
/* The same rows on a TOKEN-PACKED batch (no padding: the reference pads to the longest protein of the batch,
 * cstp_v3/modelling.py:44-46, and multiplies the padding): d_tokens int32 [cu[B]] = the proteins' tokens (<cls> seq <eos>) back
 * to back, h_cu (HOST) int32 [B + 1] their row offsets (cu[0] = 0; 2 <= cu[b+1] - cu[b] <= max_enc_tokens;
 * cu[B] <= max_batch * max_enc_tokens) -> pooled fp32 [B, enc_dim], the same values as opus_esm2_encode gives each protein
 * (different GEMM tile boundaries: equal to the tolerance of DESIGN.md section 3, not bitwise).  Mixed lengths need no buckets.
 * opus_esm2_last_hidden(ctx, out, 1, cu[B]) then returns the packed [cu[B], enc_dim] representations of the RESIDUE rows; the
 * <cls> / <eos> rows, which the mean-pool drops, are not computed by the last layer (knob "enc_full_last_layer" = 1: they are). */
int opus_esm2_encode_packed(opus_ctx *ctx, const int32_t *d_tokens, const int32_t *h_cu, int32_t B, float *d_pooled, void *stream);
/* Debug/parity tap: copy of representations[enc_layers] fp32 [B,T,enc_dim] of the last encode. */
int opus_esm2_last_hidden(opus_ctx *ctx, float *d_out, int32_t B, int32_t T, void *stream);

/* Rows P1+P2: encode_projector_embedding + switch_projector_embedding (opus_arch.py:115-131,
 * modelling.py:396-400, protein_mlp/builder.py:11-25): pooled fp32 [B,enc_dim] ->
 * fp16 [B, n_prot_tokens, dec_dim].  d_proj_out (optional, may be NULL) receives the P1 output
 * fp16 [B, proj_dim].  B is NOT limited by max_batch: the batched stage of the two-stage pipeline (SURVEY 8f N3,
 * opus_arch.py:151-161 + scripts/generate_esm_embedding.py) projects whole dataset shards at M >= 512, in chunks of
 * max(max_batch, 4096) rows.  has_protein_projector = 0 is the identity module of opus_arch.py:70-80: P1 is a cast. */
int opus_projector_forward(opus_ctx *ctx, const float *d_pooled, int32_t B, void *d_out, void *d_proj_out,
                           void *stream);
/* Row P1 alone: encode_projector_embedding (opus_arch.py:115-121): fp32 [B,enc_dim] -> fp16 [B,proj_dim]. */
int opus_protein_projector(opus_ctx *ctx, const float *d_pooled, int32_t B, void *d_out, void *stream);
/* Row P2 alone: switch_projector_embedding (opus_arch.py:122-131): fp16 [B,switch_in] -> fp16 [B,n,H]. */
int opus_switch_projector(opus_ctx *ctx, const void *d_in, int32_t B, void *d_out, void *stream);

/* Rows S1-S3: the splice of prepare_inputs_labels_for_multimodal (opus_arch.py:166-270).
 * ids int64 [B,T_text] (-200 = <seq>), mask u8 [B,T_text] (NULL = all ones), prot fp16
 * [n_prot, n_prot_tokens, dec_dim] consumed in order (a row without placeholder consumes one).
 * Outputs sized for max_prompt: embeds fp16 [B,T_out,H] (zeros in pad slots), mask u8 [B,T_out],
 * pos int32 [B,T_out].  T_out is a HOST int: the call synchronises `stream` once to return it.
 * max_length > 0 truncates rows (config.tokenizer_model_max_length, :234-237).
 * Errors: OPUS_ESHAPE when fewer protein blocks are supplied than the rows consume, or T_out
 * exceeds max_prompt. */
int opus_splice_pad(opus_ctx *ctx, const int64_t *d_ids, const uint8_t *d_mask, int32_t B, int32_t T_text,
                    const void *d_prot, int32_t n_prot, int32_t inference_mode, int32_t max_length,
                    void *d_embeds, uint8_t *d_mask_out, int32_t *d_pos_out, int32_t *T_out, void *stream);

/* Rows D1,D2,D4: LlamaForCausalLM prefill as called by generate (language_model/opus_llama.py:127-132):
 * embeds fp16 [B,T,H], mask u8 [B,T] (left-padded rows) -> logits of the LAST position fp32 [B,V];
 * fills the context's KV cache and resets its step counter. */
int opus_llama_prefill(opus_ctx *ctx, const void *d_embeds, const uint8_t *d_mask, int32_t B, int32_t T,
                       float *d_last_logits, void *stream);
/* Row D3: one decode step for the B rows of the last prefill: tok int32 [B] -> logits fp32 [B,V]. */
int opus_llama_decode_step(opus_ctx *ctx, const int32_t *d_tok, float *d_logits, void *stream);

/* Rows G1 (+D1-D4): greedy search of GenerationMixin as driven by opus_llama.py:95-132.
 * next = argmax(last logits); finished rows emit pad_id; a row finishes on any of eos_ids (host
 * array, may be empty); stops when all rows are finished or after max_new steps.
 * out ids int32 [B,max_new] (device; unused tail = pad_id); n_out (HOST) = steps produced, the
 * second dimension HF would return.  Synchronises `stream` (it returns a host scalar). */
int opus_generate_greedy(opus_ctx *ctx, const void *d_embeds, const uint8_t *d_mask, int32_t B, int32_t T,
                         int32_t max_new, const int32_t *eos_ids, int32_t n_eos, int32_t pad_id,
                         int32_t *d_out_ids, int32_t *n_out, void *stream);

/* Row N2, "### early-stop as an opt-in" (the reference decodes to max_new_tokens and cuts the text at the first "###"
 * afterwards, eval/run_opus_ddp.py:19-27): with the token ids of "###" set here (HOST array, at most 8; n = 0 clears), a row
 * is finished once its new ids end with that sequence - later positions hold pad_id, as after an EOS - so the text after the
 * cut is unchanged and a batch can stop early.  Applies to opus_generate_greedy / opus_generate_sample of this context. */
int opus_set_stop_sequence(opus_ctx *ctx, const int32_t *ids, int32_t n);

/* Row N1 (sampling head, the reference's default decode mode: run_opus_ddp.py:126-128,156-157 temperature 0.1,
 * top_p 0.7): same loop with next = multinomial(softmax(top_p_filter(logits / temperature))) per transformers'
 * TemperatureLogitsWarper + TopPLogitsWarper; draws come from a counter-based generator keyed by
 * (seed, row, step), so a given seed reproduces its tokens.  Parity with the reference is distributional. */
int opus_generate_sample(opus_ctx *ctx, const void *d_embeds, const uint8_t *d_mask, int32_t B, int32_t T,
                         int32_t max_new, const int32_t *eos_ids, int32_t n_eos, int32_t pad_id, float temperature,
                         float top_p, uint64_t seed, int32_t *d_out_ids, int32_t *n_out, void *stream);
/* Diagnostic: one draw per row of fp32 logits [B, dec_vocab] (synchronises the stream). */
int opus_debug_sample(opus_ctx *ctx, const float *d_logits, int32_t B, float temperature, float top_p, uint64_t seed,
                      int32_t step, int32_t *d_tokens, void *stream);

/* Diagnostic entry points (kernel-level parity tests and micro-benchmarks; not part of the path's
 * drop-in surface).  opus_debug_gemm: C[M,Nout] = epi(A[M,K] W[N,K]^T + bias) (+ residual fp32);
 * epi 0 none, 1 erf-GELU, 2 silu(gate)*up with W rows in [16 gate | 16 up] groups (Nout = N/2).
 * opus_debug_attention: softmax(scale * Q K^T + mask) V over [B,T,heads*hd] fp16 tensors
 * (K,V have heads/group heads); keys visible iff kstart[b] <= j < kend[b] (NULL = 0 / T) and
 * (!causal || j <= i). */
int opus_debug_gemm(opus_ctx *ctx, const void *d_A, const void *d_W, const float *d_bias, const float *d_residual,
                    void *d_C, int32_t M, int32_t N, int32_t K, int32_t epi, int32_t out_f32, void *stream);
/* Same with the fused RMSNorm prologue: A is fp32 [M,K], C = epi(rmsnorm(A) W^T) (norm weight folded in W). */
int opus_debug_gemm_norm(opus_ctx *ctx, const float *d_A, const void *d_W, void *d_C, int32_t M, int32_t N, int32_t K,
                         int32_t epi, int32_t out_f32, float eps, void *stream);
/* The QKV projection of the batched decode step as decode_step issues it (5..64 rows, narrow output, k-parts leave raw fp32
 * slabs that the attention kernel sums): d_slabs fp32 [*ks][M][N] receives them, *ks (HOST) their number; *ks = 1: the launch
 * wrote a finished output instead and nothing is copied. */
int opus_debug_gemm_slabs(opus_ctx *ctx, const void *d_A, const void *d_W, float *d_slabs, int32_t M, int32_t N, int32_t K,
                          int32_t *ks, void *stream);
/* Process-wide tuning knob of the benchmarks / parity tests (no reference counterpart): "no_stream" = 1 routes the narrow
 * GEMMs of the batched decode step through the round-2 split-K kernels instead of gemm_stream_kernel; "pp_gm" = tile rows
 * per rasterisation group of the big tiled GEMM; "debug_a_tiled" = 1: opus_debug_gemm takes A in fragment order; "no_ln_fusion" = 1: stand-alone normalisation kernels
 * instead of the norms fused around the big tiled GEMM; "enc_full_last_layer" = 1: the token-packed encoder's last layer computes
 * the <cls> / <eos> rows as well; "poison_handoff" = 1 (needs ctx): leaves the hand-off words as an
 * aborted launch would (test aid); "misc0".."misc7" scratch.  ctx (may be NULL) drops its captured decode graph. */
int opus_debug_knob(opus_ctx *ctx, const char *name, int32_t value);
/* The row-scale RMSNorm fusion as the decoder issues it (api.cpp prefill / decode_step): X <- X + A W1^T through a GEMM
 * (gemm_stream_kernel at 5..64 rows, else a split-K GEMM) whose epilogue / reduce also writes fp16(X) and per-block sums of squares, then C = epi(rmsnorm(X) W2^T) with
 * the rows scaled inside the consumer GEMM.  A fp16 [M,K1], W1 [N1,K1] and W2 [N2,N1] panel-tiled, X fp32 [M,N1] in/out,
 * C fp16 [M, N2 or N2/2]; *fused (HOST) = 1 when the fused kernels ran. */
int opus_debug_gemm_rowscale(opus_ctx *ctx, const void *d_A, const void *d_W1, float *d_X, const void *d_W2, void *d_C,
                             int32_t M, int32_t N1, int32_t K1, int32_t N2, int32_t epi, float eps, int32_t *fused,
                             void *stream);
/* The ESM-2 QKV projection + rotary as opus_esm2_encode issues it: out[M, 3 D] fp16 = rotary(A[M,K] W[3 D,K]^T + bias), query
 * third scaled by head_dim^-0.5 before the rotation, position of row m = m % T (cstp_v3 / modeling_esm.py:374, rotary
 * embedding).  The context's encoder head_dim must equal D / heads.  allow_fuse = 0 forces the stand-alone rotary kernel on
 * the stored projection; *fused (HOST) = 1 when the rotation ran in the GEMM's epilogue (large M, head_dim 64). */
int opus_debug_gemm_rope(opus_ctx *ctx, const void *d_A, const void *d_W, const float *d_bias, void *d_out, int32_t M,
                         int32_t D, int32_t K, int32_t T, int32_t heads, int32_t allow_fuse, int32_t *fused, void *stream);
int opus_debug_attention(opus_ctx *ctx, const void *d_Q, const void *d_K, const void *d_V, void *d_O,
                         const int32_t *d_kstart, const int32_t *d_kend, int32_t B, int32_t T, int32_t heads,
                         int32_t group, int32_t head_dim, int32_t causal, float scale, void *stream);

/* attn_decode_kernel as opus_llama_decode_step launches it, alone, on layer 0 of this context's KV cache (kernel-level parity
 * of rows D3 / D4 at any cache length: transformers' eager attention over a cache, modeling_llama.py:191-213, reached from
 * language_model/opus_llama.py:127).  d_qkv fp16 [B, (heads + 2 kv) hd]: the new token's q | k | v projections, not yet rotated;
 * d_k_hist / d_v_hist fp16 [B, kv, L, hd], L = T0 + step: the cache contents of slots 0 .. L-1 (keys rotated); d_kstart int32
 * [B]: first visible slot of each (left-padded) row.  The new token sits at slot L, position L - kstart[b].
 * d_out fp16 [B, heads hd]; d_k_new / d_v_new (optional) fp16 [B, kv, hd]: slot L of the cache afterwards.
 * Overwrites the cache of the last prefill (opus_llama_decode_step then fails with OPUS_ESTATE until the next prefill). */
int opus_debug_attn_decode(opus_ctx *ctx, const void *d_qkv, const void *d_k_hist, const void *d_v_hist, const int32_t *d_kstart,
                           int32_t B, int32_t T0, int32_t step, void *d_out, void *d_k_new, void *d_v_new, void *stream);

/* Synchronises `stream` and returns OPUS_EHIP if an in-launch split-K hand-off of this context gave up waiting since the last
 * check (see Conventions; the results of the calls in between are invalid), OPUS_OK otherwise.  opus_generate_* make the same
 * check before they return.  No reference counterpart (torch raises asynchronously on device-side faults). */
int opus_check_error(opus_ctx *ctx, void *stream);

/* Row N1, `num_beams` (eval/run_opus_ddp.py:129,158 -> transformers GenerationMixin._beam_search).  The decoder runs B x K rows
 * (row = b K + k) through opus_llama_prefill / opus_llama_decode_step; these two do the per-step work that touches O(K V) values or
 * the KV cache, the caller keeps the O(K) bookkeeping of running / finished beams (opus-pllm_amd/beam.py, as the reference's
 * Python does).
 * opus_beam_topk: over the fp32 logits of this context's last step, scores fp32 [B, M] (descending; ties: lower index first)
 *   and flat indices k * dec_vocab + token int32 [B, M] of the M best values of log_softmax(logits[b K + k]) + run_scores[b K + k]
 *   (generation/utils.py _get_top_k_continuations; M = max(2, 1 + #eos) K <= 16).
 * opus_kv_reorder: KV cache rows r <- rows d_src_rows[r] in every layer (Cache.reorder_cache(beam_idx)); R = rows of the last prefill. */
int opus_beam_topk(opus_ctx *ctx, const float *d_run_scores, int32_t B, int32_t K, int32_t M, float *d_scores, int32_t *d_idx,
                   void *stream);
int opus_kv_reorder(opus_ctx *ctx, const int32_t *d_src_rows, int32_t R, void *stream);
/* Beam-sample (`num_beams` > 1 with temperature > 0: run_opus_ddp.py:126-129 passes both; _get_top_k_continuations' do_sample
 * branch = torch.multinomial(softmax(accumulated), M), without replacement).  Per decoder row: log_softmax, then the warpers on
 * the log-probabilities (temperature, top_k of opus_set_sampling_top_k, top_p - both with min_tokens_to_keep = M / K, i.e. #eos + 1
 * and at least 2, as GenerationMixin._get_logits_processor builds them when num_beams > 1); per batch row: M continuations drawn without
 * replacement from softmax over the K rows' kept values + run_scores.  d_scores fp32 [B, M] = the accumulated log-probabilities of
 * the draws, d_idx int32 [B, M] = k * dec_vocab + token, in the order drawn; entries beyond the continuations of non-zero
 * probability are -inf / 0x7fffffff (torch.multinomial raises there: the caller should).  Draws come from a counter-based generator
 * keyed by (seed, step, row, token): distributional parity, reproducible per seed.  d_logits NULL = this context's last step. */
int opus_beam_sample_topk(opus_ctx *ctx, const float *d_logits, const float *d_run_scores, int32_t B, int32_t K, int32_t M,
                          float temperature, float top_p, uint64_t seed, int32_t step, float *d_scores, int32_t *d_idx, void *stream);
/* TopKLogitsWarper of this context's sampling paths (opus_generate_sample, opus_debug_sample, opus_beam_sample_topk), between the
 * temperature and the nucleus: k > 0 keeps the k most probable tokens and whatever ties with the k-th; 0 (the state of a new
 * context) = off.  transformers 4.46.3 - requirements.txt:20 of the reference - defaults GenerationConfig.top_k to 50 when it
 * samples (transformers >= 5: None); the Python mirror's generate(top_k=...) defaults to 50 accordingly. */
int opus_set_sampling_top_k(opus_ctx *ctx, int32_t k);

/* fp32 logits [B, dec_vocab] of the most recent prefill / decode step (device copy on `stream`): the payload of the
 * optional logits all-gather of SURVEY 8e (ids are what eval/run_opus_ddp.py:138 gathers). */
int opus_last_logits(opus_ctx *ctx, float *d_out, int32_t B, void *stream);

/* Counters of a context (no reference counterpart; -1 for an unknown name).  "graph_instantiations": decode-step hipGraphs
 * instantiated since the context was created - the captured step reads the prompt length from device memory, so a dataset's
 * batches share one graph whatever their T (the reference's loop, eval/run_opus_ddp.py:88-135, brings a new T with every batch);
 * a different number of rows / token budget / sampling setting is another graph (a few are kept).  "graph_replays": decode steps
 * launched from a graph.  "graphs_cached".  "decode_steps": decode steps the generate loops enqueued - with an EOS id or a stop
 * sequence the loop polls the rows' state with a bounded run-ahead and stops at most 2 steps after the last row finished (HF stops
 * at once; the extra steps emit pad ids only, the returned ids and n_out are exact). */
int64_t opus_stat(opus_ctx *ctx, const char *name);

/* Measurement support (bench.py).  With timing enabled (off by default; decode runs eagerly instead of from the
 * hipGraph) every kernel launch of the path is recorded with its own dispatch start / end events on the launch stream
 * (hipExtLaunchKernelGGL - the interval rocprofv3 --kernel-trace reports), its kernel class, the phase of the path it
 * belongs to, and its ALGORITHMIC bytes and FLOPs.  opus_timing_get sums the records since the last reset that match
 * kernel_class and phase ("*" = any); opus_timing_names returns "class,class,...;phase,phase,..." . */
int opus_timing_enable(opus_ctx *ctx, int32_t on);
int opus_timing_reset(opus_ctx *ctx);
int opus_timing_get(opus_ctx *ctx, const char *kernel_class, const char *phase, double *ms, int64_t *launches, double *bytes,
                    double *flops);
int opus_timing_names(char *buf, int32_t cap);

#ifdef __cplusplus
}
#endif
#endif /* OPUS_PLLM_H */
