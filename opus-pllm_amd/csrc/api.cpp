// C ABI + runtime of libopus_pllm.so: context, weight registry, workspace / KV cache, the kernel
// sequences of the path (encoder, projectors, splice, prefill, decode), the on-device greedy loop
// with hipGraph replay of the decode step, and per-kernel-class event timing for bench.py.
#include "../../include/opus_pllm.h"
#include "common.h"

#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <string>
#include <vector>

using namespace opus;

// ------------------------------------------------------------------------------------------------
static thread_local char g_err[512] = "";
static int fail(int code, const char *fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
    return code;
}
#define HIPC(expr)                                                                                   \
    do {                                                                                             \
        hipError_t e_ = (expr);                                                                      \
        if (e_ != hipSuccess) return fail(OPUS_EHIP, "%s failed: %s (%s:%d)", #expr,                 \
                                          hipGetErrorString(e_), __FILE__, __LINE__);                \
    } while (0)
#define OPC(expr)                  \
    do {                           \
        int r_ = (expr);           \
        if (r_ != OPUS_OK) return r_; \
    } while (0)

struct Tensor {
    const void *p = nullptr;
    int dtype = -1;
    std::vector<int64_t> shape;
};

struct EncLayer {   // the LayerNorm weights are folded into wqkv / w1 and the biases bqkv / b1 at load time (weights.py);
    const float *bqkv, *bo, *b1, *b2, *sqkv, *s1;   // sqkv / s1: column sums of the folded weights (fused LayerNorm, GemmParams::ln_*)
    const half_t *wqkv, *wo, *w1, *w2;
};
struct DecLayer {   // arch 0: RMSNorm weights are folded into wqkv / wgu (and dec.lnf into lm_head) at load time
    const half_t *wqkv = nullptr, *wo = nullptr, *wgu = nullptr, *wd = nullptr;
    const float *bqkv = nullptr;                      // Qwen2 q/k/v bias, OPT bias
    // arch 1 (OPT / Galactica): LayerNorm affine parameters, fc1 / fc2 and the remaining biases
    const half_t *w1 = nullptr, *w2 = nullptr;
    const float *bo = nullptr, *b1 = nullptr, *b2 = nullptr, *ln1w = nullptr, *ln1b = nullptr, *ln2w = nullptr, *ln2b = nullptr;
};

struct TimeRec {
    int klass, phase;
    hipEvent_t e0, e1;
    double bytes, flops;
};

struct opus_ctx {
    opus_config cfg;
    int device = 0;
    std::map<std::string, Tensor> w;
    bool resolved = false;
    // resolved weights
    const half_t *enc_emb = nullptr, *dec_emb = nullptr, *lm_head = nullptr, *proj_w = nullptr, *dec_pos = nullptr;
    const float *enc_lnfw = nullptr, *enc_lnfb = nullptr, *proj_b = nullptr, *dec_lnfw = nullptr, *dec_lnfb = nullptr;
    std::vector<EncLayer> enc;
    std::vector<DecLayer> dec;
    std::vector<const half_t *> sw_w;
    std::vector<const float *> sw_b;
    // workspace
    char *ws = nullptr;
    size_t ws_bytes = 0;
    float *e_x, *e_hid, *p_pool_dummy, *e_part, *e_stat;
    int32_t *e_cu, *e_pos;
    std::vector<int32_t> h_cu;           // host copy of the packed encoder's row offsets (source of the async upload)
    half_t *e_xn, *e_qkv, *e_ctx, *e_h1;
    half_t *p_xn, *p_y, *p_z[2];
    float *d_x, *d_xl, *d_logits, *d_pval, *gemm_ws, *d_probs, *d_zpart, *d_spart, *d_part, *d_stat;
    int32_t *d_cand_i, *d_cand_n;
    uint64_t *d_seed;
    int32_t *d_chosen;
    // sampling head (0 = greedy)
    float samp_temp = 0.f, samp_top_p = 1.f;
    int samp_top_k = 0;                       // opus_set_sampling_top_k (0 = no TopKLogitsWarper)
    float *d_bthr = nullptr, *d_blse = nullptr;   // beam-sample: keep threshold and log-sum-exp per decoder row
    int64_t gemm_ws_bytes = 0;
    half_t *d_xn, *d_qkv, *d_ctx, *d_act, *d_xln, *kc, *vc;
    float *cs_enc, *cs_dec, *cs_row;
    int32_t *d_kstart, *d_step, *d_next, *d_fin, *d_nunf, *d_eos, *d_plan, *d_pidx, *d_stop, *d_cnt;
    int n_stop = 0;                      // opt-in stop sequence (opus_set_stop_sequence)
    // "has every row finished?" is polled with a bounded run-ahead (generate_impl): the step's unfinished count lands in pinned host
    // memory behind an event per step
    static constexpr int POLL_RING = 4, POLL_LAG = 2;
    int32_t *h_nunf = nullptr;
    hipEvent_t poll_ev[POLL_RING] = {nullptr, nullptr, nullptr, nullptr};
    int64_t cache_sl, cache_sb, cache_sh;   // strides (halfs): layer, batch row, kv head
    // decode state
    int cur_B = 0, cur_T = 0;
    bool prefilled = false;
    // Graph cache for the decode iteration.  A captured step does NOT depend on the prompt length: T0 reaches its kernels through
    // the device word d_step[1] (d_step[0] = the step counter), so one instantiated graph serves every batch of a dataset whatever
    // its T.  It does depend on the number of rows (grid sizes, kernel routing), the token budget (stride of the id matrix), the
    // pad / EOS / sampling settings and the id buffer: a few entries are kept (a dataset's full batches + its short last batch),
    // least recently used first out.
    struct GraphEntry {
        hipGraphExec_t exec = nullptr;
        int B = -1, maxnew = -1, pad = 0, neos = -1;
        const int32_t *out = nullptr;
        float temp = 0.f, top_p = 1.f;
        uint64_t used = 0;
    };
    static constexpr int MAX_GRAPHS = 4;
    std::vector<GraphEntry> graphs;
    uint64_t graph_clock = 0;
    int64_t graph_instantiations = 0;    // opus_stat("graph_instantiations")
    int64_t graph_replays = 0;
    int64_t decode_steps = 0;            // decode steps enqueued by opus_generate_* (eager or from a graph)
    // row-scale fusion (GemmParams::xh_out / row_ssq): one-shot request for the next gemm() and its outcome
    half_t *rq_xh = nullptr;
    int rq_done = 0;
    // one-shot request: fuse the ESM rotary into the next GEMM's epilogue (GemmParams::rope_*); rq_rope_done reports back
    const float *rq_rope_cs = nullptr;
    const int32_t *rq_rope_pos = nullptr;
    int rq_rope_T = 0, rq_rope_cols = 0, rq_rope_qcols = 0, rq_rope_done = 0;
    float rq_rope_qscale = 1.0f;
    // one-shot requests for the next gemm(): route a narrow output through the wide kernel / leave raw k-part slabs
    int rq_force_wide = 0, rq_slab_only = 0, rq_ks = 1;
    // one-shot: the next gemm() reads A / writes the fp16 copy of its output in fragment order (GemmParams::a_tiled / xh_tiled)
    int rq_a_tiled = 0, rq_xh_tiled = 0, rq_c_tiled = 0;
    // one-shot requests, fused LayerNorm around gemm_pp_kernel (GemmParams::ln_*): producer (fp16(x) into e_xn + partials into
    // e_part; rq_ln_done reports back) and consumer (rows scaled / shifted with e_stat and this column-sum vector)
    float *rq_ln_part = nullptr;         // producer: partials here, fp16(x) into rq_ln_xh
    half_t *rq_ln_xh = nullptr;
    int rq_ln_done = 0;
    const float *rq_ln_stat = nullptr, *rq_ln_colsum = nullptr;   // consumer
    bool xln_tiled = false;              // d_xln currently holds fp16(x) in fragment order
    const float *xh_src = nullptr;       // fp32 buffer whose fp16 copy + sum-of-squares partials are valid
    bool use_row_scale = false;          // one-shot: the next gemm() multiplies its rows by the rstd from d_ssq
    float *d_ssq = nullptr;
    int64_t ssq_cap = 0;                 // floats in d_ssq
    int ssq_nblk = 0;                    // blocks per row the last producer wrote (N / 256: split-K reduce, embedding; N / 16: gemm_stream)
    float row_eps = 0.f;
    // timing
    bool timing = false;
    int phase = PH_OTHER;
    std::vector<TimeRec> recs;
    // rows the projector workspace (p_xn, p_y, p_z) holds: larger batches are projected in chunks of this many rows.  The
    // context's own workspace holds max_batch rows; the first call with more rows (the batched stage of the two-stage
    // pipeline) allocates a separate BIG_PROJ_ROWS-row workspace (proj_big) and switches to it.
    int proj_rows = 0;
    char *proj_big = nullptr;
    // beam search (opus_beam_topk / opus_kv_reorder): one layer's K and V cache rows while they are permuted (first use allocates)
    half_t *kv_tmp = nullptr;
};

// ------------------------------------------------------------------------------------------------
static size_t align_up(size_t v) { return (v + 255) & ~(size_t)255; }

struct Carver {
    char *base;
    size_t off = 0;
    template <class T>
    T *take(size_t n) {
        T *p = base ? reinterpret_cast<T *>(base + off) : nullptr;
        off += align_up(n * sizeof(T));
        return p;
    }
};

static void carve(opus_ctx *c, char *base, size_t *total) {
    const opus_config &g = c->cfg;
    Carver k{base};
    const size_t Me = (size_t)g.max_batch * g.max_enc_tokens;
    const size_t De = g.enc_dim, Fe = g.enc_ffn;
    c->e_x = k.take<float>(Me * De);
    c->e_hid = k.take<float>(Me * De);
    c->e_xn = k.take<half_t>(Me * De);
    c->e_qkv = k.take<half_t>(Me * 3 * De);
    c->e_ctx = k.take<half_t>(Me * De);
    c->e_h1 = k.take<half_t>(Me * Fe);
    c->e_part = k.take<float>(Me * (De / 64 + 1) * 2);              // (sum x, sum x^2) per row and 64-column slab (fused LayerNorm)
    c->e_stat = k.take<float>(Me * 2);                              // (mu, rstd) per row
    c->e_cu = k.take<int32_t>((size_t)g.max_batch + 4);            // token-packed encoder: row offsets, row -> position table
    c->e_pos = k.take<int32_t>(Me);
    const size_t B = g.max_batch, H = g.dec_dim, SW = (size_t)g.dec_dim * g.n_prot_tokens;
    const size_t PR = B;                         // (the two-stage pipeline's big workspace is allocated on first use: ensure_proj_rows)
    c->proj_rows = (int)PR;
    c->p_xn = k.take<half_t>(PR * De);
    c->p_y = k.take<half_t>(PR * (size_t)(g.has_protein_projector ? g.proj_dim : g.enc_dim));
    c->p_z[0] = k.take<half_t>(PR * SW);
    c->p_z[1] = k.take<half_t>(PR * SW);
    // rows of the decoder activation buffers: whole 16-row tiles, so that the fragment-ordered layout of the batched decode
    // step (tiled_off) fits for every batch size
    const size_t B16 = (B + 15) & ~(size_t)15;
    const size_t Md = B * g.max_prompt > B16 ? B * g.max_prompt : B16;
    const size_t QKV = (size_t)(g.dec_heads + 2 * g.dec_kv_heads) * g.dec_head_dim;
    const size_t QD = (size_t)g.dec_heads * g.dec_head_dim;
    c->d_x = k.take<float>(Md * H);
    c->d_xn = k.take<half_t>(Md * H);
    c->d_qkv = k.take<half_t>(Md * QKV);
    c->d_ctx = k.take<half_t>(Md * QD);
    c->d_act = k.take<half_t>(Md * g.dec_ffn);
    c->d_part = k.take<float>(Md * (H / 64 + 1) * 2);               // prefill RMSNorm fused around the big GEMM: partials, (0, rstd)
    c->d_stat = k.take<float>(Md * 2);
    c->d_xl = k.take<float>(B * H);
    c->d_xln = k.take<half_t>(B16 * H);
    c->d_logits = k.take<float>(B * (size_t)g.dec_vocab);
    const size_t ctx = (size_t)g.max_prompt + g.max_new_tokens;
    c->cache_sh = (int64_t)ctx * g.dec_head_dim;
    c->cache_sb = c->cache_sh * g.dec_kv_heads;
    c->cache_sl = c->cache_sb * g.max_batch;
    c->kc = k.take<half_t>((size_t)c->cache_sl * g.dec_layers);
    c->vc = k.take<half_t>((size_t)c->cache_sl * g.dec_layers);
    const size_t maxpos_e = g.max_enc_tokens, maxpos_d = ctx;
    c->cs_enc = k.take<float>(maxpos_e * (g.enc_dim / g.enc_heads));
    c->cs_dec = k.take<float>(maxpos_d * g.dec_head_dim);
    c->cs_row = k.take<float>(B * (size_t)g.dec_head_dim);          // (cos, sin) rows of the current decode step, one per batch row
    c->d_kstart = k.take<int32_t>(B + 4);
    c->d_step = k.take<int32_t>(4);
    c->d_next = k.take<int32_t>(B);
    c->d_fin = k.take<int32_t>(B);
    c->d_nunf = k.take<int32_t>((size_t)g.max_new_tokens + 4);
    c->d_eos = k.take<int32_t>(64);
    c->d_stop = k.take<int32_t>(16);
    c->d_cnt = k.take<int32_t>(HANDOFF_WORDS);                      // ticket / flag words of the in-launch hand-offs + the error word (GemmParams::combine_cnt)
    c->d_plan = k.take<int32_t>(4 * B + 8);
    c->d_pval = k.take<float>(64 * B);
    c->d_probs = k.take<float>(B * ((size_t)g.dec_vocab + 64 * 4));     // candidate probabilities (per-part slots)
    c->d_cand_i = k.take<int32_t>(B * ((size_t)g.dec_vocab + 64 * 4));
    c->d_cand_n = k.take<int32_t>(64 * B);
    c->d_zpart = k.take<float>(64 * B);
    c->d_bthr = k.take<float>(B);
    c->d_blse = k.take<float>(B);
    c->d_spart = k.take<float>(64 * B);
    c->d_seed = k.take<uint64_t>(2);
    c->d_chosen = k.take<int32_t>(B);
    c->gemm_ws_bytes = 64ll << 20;   // split-K slabs of the tile GEMM
    c->gemm_ws = k.take<float>((size_t)c->gemm_ws_bytes / sizeof(float));
    c->d_pidx = k.take<int32_t>(64 * B);
    // per-row sums of squares of the row-scale RMSNorm fusion: up to 128 rows x one float per 16 columns of the residual stream
    c->ssq_cap = 128 * (int64_t)(g.dec_dim / 16 > 32 ? g.dec_dim / 16 : 32);
    c->d_ssq = k.take<float>((size_t)c->ssq_cap);
    *total = k.off;
}

static int check_cfg(const opus_config *g) {
    if (!g) return fail(OPUS_EBADARG, "config is null");
    if (g->enc_layers < 1 || g->dec_layers < 1 || g->enc_heads < 1 || g->dec_heads < 1 || g->dec_kv_heads < 1)
        return fail(OPUS_EBADARG, "layer/head counts must be positive");
    if (g->enc_dim % g->enc_heads) return fail(OPUS_ESHAPE, "enc_dim %% enc_heads != 0");
    const int ehd = g->enc_dim / g->enc_heads;
    auto okhd = [](int h) { return h == 16 || h == 32 || h == 64 || h == 128; };
    if (!okhd(ehd) || !okhd(g->dec_head_dim)) return fail(OPUS_ESHAPE, "head_dim must be 16/32/64/128");
    if (g->dec_heads % g->dec_kv_heads || g->dec_heads / g->dec_kv_heads > 8)
        return fail(OPUS_ESHAPE, "dec_heads / dec_kv_heads must be an integer <= 8");
    const int ks[] = {g->enc_dim, g->enc_ffn, g->dec_dim, g->dec_ffn, g->dec_heads * g->dec_head_dim,
                      g->has_protein_projector ? g->proj_dim : 64, g->dec_dim * g->n_prot_tokens};
    for (int k : ks)
        if (k % 64) return fail(OPUS_ESHAPE, "every GEMM reduction dim must be a multiple of 64 (got %d)", k);
    if (g->enc_dim > 5120 || g->dec_dim > 5120) return fail(OPUS_ESHAPE, "norm width > 5120 unsupported");
    if (g->switch_depth < 1 || g->n_prot_tokens < 1) return fail(OPUS_EBADARG, "switch_depth/n_prot_tokens");
    if (g->max_batch < 1 || g->max_enc_tokens < 3 || g->max_prompt < 1 || g->max_new_tokens < 1)
        return fail(OPUS_EBADARG, "capacity fields must be positive");
    if (g->dec_arch != 0 && g->dec_arch != 1) return fail(OPUS_EBADARG, "dec_arch must be 0 (Llama/Qwen2) or 1 (OPT/Galactica)");
    if (g->dec_arch == 1) {
        if (g->dec_heads != g->dec_kv_heads) return fail(OPUS_ESHAPE, "OPT attention is multi-head: dec_kv_heads == dec_heads");
        if (g->dec_act != 0 && g->dec_act != 1) return fail(OPUS_EUNSUPPORTED, "OPT activation: 0 = GELU (Galactica) or 1 = ReLU (facebook/opt-*)");
        if (g->dec_ffn & 7) return fail(OPUS_ESHAPE, "dec_ffn must be a multiple of 8");
        if (g->max_prompt + g->max_new_tokens > g->dec_max_pos)
            return fail(OPUS_ESHAPE, "max_prompt + max_new_tokens exceeds the learned position table (dec_max_pos=%d)", g->dec_max_pos);
    }
    return OPUS_OK;
}

// ------------------------------------------------------------------------------------------------
extern "C" int opus_abi_version(void) { return OPUS_ABI_VERSION; }
extern "C" int opus_operand_dtype(void) {
#ifdef OPUS_BF16
    return 1;
#else
    return 0;
#endif
}
extern "C" const char *opus_last_error(void) { return g_err; }

extern "C" int64_t opus_workspace_bytes(const opus_config *cfg) {
    if (check_cfg(cfg) != OPUS_OK) return -1;
    opus_ctx tmp;
    tmp.cfg = *cfg;
    size_t total = 0;
    carve(&tmp, nullptr, &total);
    return (int64_t)total;
}

static void fill_cs(std::vector<float> &t, int maxpos, int hd, float theta) {
    const int half = hd / 2;
    t.resize((size_t)maxpos * half * 2);
    for (int i = 0; i < half; ++i) {
        // fp32 inv_freq as torch computes it, angle evaluated in double
        const float inv = 1.0f / powf(theta, (float)(2 * i) / (float)hd);
        for (int p = 0; p < maxpos; ++p) {
            const float ang = (float)p * inv;
            t[((size_t)p * half + i) * 2] = (float)cos((double)ang);
            t[((size_t)p * half + i) * 2 + 1] = (float)sin((double)ang);
        }
    }
}

extern "C" int opus_ctx_create(const opus_config *cfg, int device, opus_ctx **out) {
    if (!out) return fail(OPUS_EBADARG, "out is null");
    OPC(check_cfg(cfg));
    HIPC(hipSetDevice(device));
    opus_ctx *c = new opus_ctx();
    c->cfg = *cfg;
    c->device = device;
    size_t total = 0;
    carve(c, nullptr, &total);
    hipError_t e = hipMalloc((void **)&c->ws, total);
    if (e != hipSuccess) {
        delete c;
        return fail(OPUS_EHIP, "hipMalloc(%zu bytes of workspace) failed: %s", total, hipGetErrorString(e));
    }
    c->ws_bytes = total;
    carve(c, c->ws, &total);
    std::vector<float> t;
    fill_cs(t, cfg->max_enc_tokens, cfg->enc_dim / cfg->enc_heads, cfg->enc_rope_theta);
    HIPC(hipMemcpy(c->cs_enc, t.data(), t.size() * sizeof(float), hipMemcpyHostToDevice));
    fill_cs(t, cfg->max_prompt + cfg->max_new_tokens, cfg->dec_head_dim, cfg->dec_rope_theta);
    if (cfg->dec_arch == 1)      // no rotary in OPT: (cos, sin) = (1, 0) makes the fused rotate-and-cache kernels plain copies
        for (size_t i = 0; i < t.size(); i += 2) { t[i] = 1.0f; t[i + 1] = 0.0f; }
    HIPC(hipMemcpy(c->cs_dec, t.data(), t.size() * sizeof(float), hipMemcpyHostToDevice));
    HIPC(hipMemset(c->d_step, 0, 16));
    HIPC(hipMemset(c->d_cnt, 0, HANDOFF_WORDS * sizeof(int32_t)));
    // the decode attention reads whole 32-slot tiles and masks afterwards: every cache slot must hold finite values
    HIPC(hipMemset(c->kc, 0, (size_t)c->cache_sl * cfg->dec_layers * sizeof(half_t)));
    HIPC(hipMemset(c->vc, 0, (size_t)c->cache_sl * cfg->dec_layers * sizeof(half_t)));
    HIPC(hipHostMalloc((void **)&c->h_nunf, ((size_t)cfg->max_new_tokens + 4) * sizeof(int32_t), hipHostMallocDefault));
    for (auto &e : c->poll_ev) HIPC(hipEventCreateWithFlags(&e, hipEventDisableTiming));
    *out = c;
    return OPUS_OK;
}

static void timing_clear(opus_ctx *c) {
    for (auto &r : c->recs) {
        (void)hipEventDestroy(r.e0);
        (void)hipEventDestroy(r.e1);
    }
    c->recs.clear();
}

static void drop_graphs(opus_ctx *c) {   // a captured step holds weight pointers, the stop sequence, top-k, knobs ...
    for (auto &g : c->graphs)
        if (g.exec) (void)hipGraphExecDestroy(g.exec);
    c->graphs.clear();
}

extern "C" int opus_ctx_destroy(opus_ctx *c) {
    if (!c) return OPUS_OK;
    (void)hipSetDevice(c->device);
    timing_clear(c);
    drop_graphs(c);
    if (c->ws) (void)hipFree(c->ws);
    if (c->proj_big) (void)hipFree(c->proj_big);
    if (c->kv_tmp) (void)hipFree(c->kv_tmp);
    if (c->h_nunf) (void)hipHostFree(c->h_nunf);
    for (auto &e : c->poll_ev) if (e) (void)hipEventDestroy(e);
    delete c;
    return OPUS_OK;
}

// ------------------------------------------------------------------------------------------------ weights
extern "C" int opus_bind_weight(opus_ctx *c, const char *name, const void *d_ptr, int dtype, int ndim,
                                const int64_t *shape) {
    if (!c || !name || !d_ptr || !shape) return fail(OPUS_EBADARG, "null argument");
    if (ndim < 1 || ndim > 4) return fail(OPUS_EBADARG, "ndim %d", ndim);
    if (dtype != OPUS_F16 && dtype != OPUS_F32) return fail(OPUS_EBADARG, "weights are fp16 or fp32");
    if (((uintptr_t)d_ptr) & 15) return fail(OPUS_EBADARG, "%s: pointer must be 16-byte aligned", name);
    Tensor t;
    t.p = d_ptr;
    t.dtype = dtype;
    t.shape.assign(shape, shape + ndim);
    c->w[name] = t;
    c->resolved = false;
    // a captured decode graph holds the device pointers of the weights it was recorded with
    drop_graphs(c);
    return OPUS_OK;
}

static int getw(opus_ctx *c, const std::string &name, int dtype, std::vector<int64_t> shape, const void **out) {
    auto it = c->w.find(name);
    if (it == c->w.end()) return fail(OPUS_ESTATE, "weight '%s' is not bound", name.c_str());
    const Tensor &t = it->second;
    if (t.dtype != dtype) return fail(OPUS_ESHAPE, "weight '%s' has dtype %d, expected %d", name.c_str(), t.dtype, dtype);
    if (t.shape != shape) {
        std::string got, exp;
        for (auto v : t.shape) got += std::to_string(v) + ",";
        for (auto v : shape) exp += std::to_string(v) + ",";
        return fail(OPUS_ESHAPE, "weight '%s' has shape [%s] expected [%s]", name.c_str(), got.c_str(), exp.c_str());
    }
    *out = t.p;
    return OPUS_OK;
}
#define GW(name, dt, shape, dst) OPC(getw(c, name, dt, shape, reinterpret_cast<const void **>(&(dst))))

extern "C" int opus_weights_ready(opus_ctx *c) {
    if (!c) return fail(OPUS_EBADARG, "ctx is null");
    if (c->resolved) return OPUS_OK;
    const opus_config &g = c->cfg;
    const int64_t De = g.enc_dim, Fe = g.enc_ffn, H = g.dec_dim, F = g.dec_ffn, V = g.dec_vocab;
    const int64_t QKV = (int64_t)(g.dec_heads + 2 * g.dec_kv_heads) * g.dec_head_dim, QD = (int64_t)g.dec_heads * g.dec_head_dim;
    GW("enc.emb", OPUS_F16, (std::vector<int64_t>{g.enc_vocab, De}), c->enc_emb);
    c->enc.resize(g.enc_layers);
    for (int l = 0; l < g.enc_layers; ++l) {
        const std::string p = "enc." + std::to_string(l) + ".";
        EncLayer &L = c->enc[l];
        GW(p + "wqkv", OPUS_F16, (std::vector<int64_t>{3 * De, De}), L.wqkv);
        GW(p + "bqkv", OPUS_F32, (std::vector<int64_t>{3 * De}), L.bqkv);
        GW(p + "sqkv", OPUS_F32, (std::vector<int64_t>{3 * De}), L.sqkv);
        GW(p + "wo", OPUS_F16, (std::vector<int64_t>{De, De}), L.wo);
        GW(p + "bo", OPUS_F32, (std::vector<int64_t>{De}), L.bo);
        GW(p + "w1", OPUS_F16, (std::vector<int64_t>{Fe, De}), L.w1);
        GW(p + "b1", OPUS_F32, (std::vector<int64_t>{Fe}), L.b1);
        GW(p + "s1", OPUS_F32, (std::vector<int64_t>{Fe}), L.s1);
        GW(p + "w2", OPUS_F16, (std::vector<int64_t>{De, Fe}), L.w2);
        GW(p + "b2", OPUS_F32, (std::vector<int64_t>{De}), L.b2);
    }
    GW("enc.lnf.w", OPUS_F32, (std::vector<int64_t>{De}), c->enc_lnfw);
    GW("enc.lnf.b", OPUS_F32, (std::vector<int64_t>{De}), c->enc_lnfb);
    int64_t din = De;
    if (g.has_protein_projector) {
        GW("proj.w", OPUS_F16, (std::vector<int64_t>{g.proj_dim, De}), c->proj_w);
        GW("proj.b", OPUS_F32, (std::vector<int64_t>{g.proj_dim}), c->proj_b);
        din = g.proj_dim;
    }
    const int64_t SW = H * g.n_prot_tokens;
    c->sw_w.resize(g.switch_depth);
    c->sw_b.resize(g.switch_depth);
    for (int i = 0; i < g.switch_depth; ++i) {
        const std::string p = "sw." + std::to_string(i) + ".";
        GW(p + "w", OPUS_F16, (std::vector<int64_t>{SW, din}), c->sw_w[i]);
        GW(p + "b", OPUS_F32, (std::vector<int64_t>{SW}), c->sw_b[i]);
        din = SW;
    }
    GW("dec.emb", OPUS_F16, (std::vector<int64_t>{V, H}), c->dec_emb);
    c->dec.resize(g.dec_layers);
    for (int l = 0; l < g.dec_layers; ++l) {
        const std::string p = "dec." + std::to_string(l) + ".";
        DecLayer &L = c->dec[l];
        GW(p + "wqkv", OPUS_F16, (std::vector<int64_t>{QKV, H}), L.wqkv);
        GW(p + "wo", OPUS_F16, (std::vector<int64_t>{H, QD}), L.wo);
        if (g.dec_arch == 1) {
            GW(p + "ln1.w", OPUS_F32, (std::vector<int64_t>{H}), L.ln1w);
            GW(p + "ln1.b", OPUS_F32, (std::vector<int64_t>{H}), L.ln1b);
            GW(p + "bqkv", OPUS_F32, (std::vector<int64_t>{QKV}), L.bqkv);
            GW(p + "bo", OPUS_F32, (std::vector<int64_t>{H}), L.bo);
            GW(p + "ln2.w", OPUS_F32, (std::vector<int64_t>{H}), L.ln2w);
            GW(p + "ln2.b", OPUS_F32, (std::vector<int64_t>{H}), L.ln2b);
            GW(p + "w1", OPUS_F16, (std::vector<int64_t>{F, H}), L.w1);
            GW(p + "b1", OPUS_F32, (std::vector<int64_t>{F}), L.b1);
            GW(p + "w2", OPUS_F16, (std::vector<int64_t>{H, F}), L.w2);
            GW(p + "b2", OPUS_F32, (std::vector<int64_t>{H}), L.b2);
            continue;
        }
        if (g.dec_qkv_bias) GW(p + "bqkv", OPUS_F32, (std::vector<int64_t>{QKV}), L.bqkv);
        GW(p + "wgu", OPUS_F16, (std::vector<int64_t>{2 * F, H}), L.wgu);
        GW(p + "wd", OPUS_F16, (std::vector<int64_t>{H, F}), L.wd);
    }
    if (g.dec_arch == 1) {
        GW("dec.pos", OPUS_F16, (std::vector<int64_t>{g.dec_max_pos + 2, H}), c->dec_pos);
        GW("dec.lnf.w", OPUS_F32, (std::vector<int64_t>{H}), c->dec_lnfw);
        GW("dec.lnf.b", OPUS_F32, (std::vector<int64_t>{H}), c->dec_lnfb);
    }
    GW("dec.lm_head", OPUS_F16, (std::vector<int64_t>{V, H}), c->lm_head);
    c->resolved = true;
    return OPUS_OK;
}

// ------------------------------------------------------------------------------------------------ launch helpers
// Per-launch timing (timing mode only): the principal kernel of the bracketed call is dispatched with its own start /
// end events (OPUS_LAUNCH), a split-K reduce behind it with a second pair; a call that launches through plain
// hipLaunchKernelGGL falls back to events recorded around it on the stream.
struct Timed {
    opus_ctx *c;
    hipStream_t s;
    int klass;
    double bytes, flops;
    LaunchEvents ev;
    hipEvent_t b0 = nullptr, b1 = nullptr;
    Timed(opus_ctx *c_, hipStream_t s_, int k, double b, double f = 0.0) : c(c_), s(s_), klass(k), bytes(b), flops(f) {
        if (c->timing) {
            (void)hipEventCreate(&ev.main0); (void)hipEventCreate(&ev.main1);
            (void)hipEventCreate(&ev.aux0); (void)hipEventCreate(&ev.aux1);
            (void)hipEventCreate(&b0); (void)hipEventCreate(&b1);
            (void)hipEventRecord(b0, s);
            tl_launch_ev = &ev;
        }
    }
    ~Timed() {
        if (!c->timing) return;
        tl_launch_ev = nullptr;
        if (ev.main_used) {
            c->recs.push_back(TimeRec{ev.main_class >= 0 ? ev.main_class : klass, c->phase, ev.main0, ev.main1, bytes, flops});
            (void)hipEventDestroy(b0); (void)hipEventDestroy(b1);
        } else {
            (void)hipEventRecord(b1, s);
            c->recs.push_back(TimeRec{klass, c->phase, b0, b1, bytes, flops});
            (void)hipEventDestroy(ev.main0); (void)hipEventDestroy(ev.main1);
        }
        if (ev.aux_used) c->recs.push_back(TimeRec{KC_REDUCE, c->phase, ev.aux0, ev.aux1, ev.aux_bytes, 0.0});
        else { (void)hipEventDestroy(ev.aux0); (void)hipEventDestroy(ev.aux1); }
    }
};

static int gemm_any(opus_ctx *c, hipStream_t s, const half_t *A, const float *Af, float eps, int64_t lda, const half_t *W,
                    int M, int N, int K, const float *bias, int epi, const float *residual, void *C, int64_t ldc,
                    int out_f32) {
    GemmParams p;
    p.A = A; p.Af = Af; p.norm_eps = eps; p.lda = lda; p.W = W; p.M = M; p.N = N; p.K = K; p.bias = bias;
    p.residual = residual; p.ldr = ldc; p.C = C; p.ldc = ldc; p.out_f32 = out_f32; p.epi = epi;
    p.ws = c->gemm_ws; p.ws_bytes = c->gemm_ws_bytes;
    p.xh_out = c->rq_xh; p.ssq_out = c->d_ssq; p.fused_done = &c->rq_done; p.ssq_cap = c->ssq_cap; p.nblk_out = &c->ssq_nblk;
    c->rq_done = 0;
    c->rq_xh = nullptr;                  // one-shot
    p.rope_cs = c->rq_rope_cs; p.rope_T = c->rq_rope_T; p.rope_cols = c->rq_rope_cols; p.rope_qcols = c->rq_rope_qcols;
    p.rope_qscale = c->rq_rope_qscale; p.rope_done = &c->rq_rope_done; p.rope_pos = c->rq_rope_pos;
    c->rq_rope_done = 0;
    c->rq_rope_cs = nullptr;             // one-shot
    c->rq_rope_pos = nullptr;
    p.row_ssq = nullptr; p.row_nblk = 0;
    if (c->use_row_scale) { p.row_ssq = c->d_ssq; p.row_nblk = c->ssq_nblk; p.norm_eps = c->row_eps; c->use_row_scale = false; }
    c->rq_ks = 1;
    p.force_wide = c->rq_force_wide; p.slab_only = c->rq_slab_only; p.ks_out = &c->rq_ks;
    c->rq_force_wide = c->rq_slab_only = 0;
    c->rq_ln_done = 0;
    if (c->rq_ln_part) { p.xh_out = c->rq_ln_xh; p.ssq_out = nullptr; p.ln_part = c->rq_ln_part; p.ln_done = &c->rq_ln_done; c->rq_ln_part = nullptr; }
    if (c->rq_ln_stat) { p.ln_stat = c->rq_ln_stat; p.ln_colsum = c->rq_ln_colsum; c->rq_ln_stat = c->rq_ln_colsum = nullptr; }
    p.combine_cnt = c->d_cnt;
    p.a_tiled = c->rq_a_tiled; p.xh_tiled = c->rq_xh_tiled; p.c_tiled = c->rq_c_tiled;
    c->rq_a_tiled = c->rq_xh_tiled = c->rq_c_tiled = 0;
    const int nout = epi == EPI_SILU_GU16 ? N / 2 : N;
    // algorithmic bytes: the weights once + activations in + result out (+ the residual read)
    const double bytes = 2.0 * N * K + (Af ? 4.0 : 2.0) * M * K + (double)M * nout * (out_f32 ? 4 : 2) +
                         (residual ? 4.0 * M * nout : 0.0);
    int klass = M <= SKINNY_MAX_M ? KC_SKINNY : KC_TILE;
    hipError_t e;
    {
        Timed t(c, s, klass, bytes, 2.0 * M * N * (double)K);
        e = launch_gemm(p, s, &klass);
    }
    if (e != hipSuccess) return fail(OPUS_EHIP, "gemm M=%d N=%d K=%d failed: %s", M, N, K, hipGetErrorString(e));
    return OPUS_OK;
}
static int gemm(opus_ctx *c, hipStream_t s, const half_t *A, int64_t lda, const half_t *W, int M, int N, int K,
                const float *bias, int epi, const float *residual, void *C, int64_t ldc, int out_f32) {
    return gemm_any(c, s, A, nullptr, 0.f, lda, W, M, N, K, bias, epi, residual, C, ldc, out_f32);
}
static bool fuse_rows() {
    static const bool off = getenv("OPUS_NO_ROW_FUSION") != nullptr;   // A/B aid
    return !off;
}
#define KLF(klass, bytes, flops, call)                                                                \
    do {                                                                                              \
        hipError_t e_;                                                                                \
        {                                                                                             \
            Timed t_(c, s, klass, bytes, flops);                                                      \
            e_ = (call);                                                                              \
        }                                                                                             \
        if (e_ != hipSuccess) return fail(OPUS_EHIP, "%s failed: %s", #call, hipGetErrorString(e_)); \
    } while (0)
#define KL(klass, bytes, call) KLF(klass, bytes, 0.0, call)

// C = epi(rmsnorm(X) W'^T) with the norm weight pre-folded into W': fused in the skinny kernel (M <= 64),
// otherwise a weight-less rmsnorm kernel into `scratch` followed by the tile kernel.
static int gemm_norm(opus_ctx *c, hipStream_t s, const float *X, float eps, half_t *scratch, const half_t *W, int M, int N,
                     int K, int epi, void *C, int64_t ldc, int out_f32, const float *bias = nullptr) {
    static const bool no_mid = getenv("OPUS_NO_MID_GEMM") != nullptr;
    static const bool mid_v1 = getenv("OPUS_MID_V1") != nullptr;
    if (c->xh_src == X && bias == nullptr && (K & 255) == 0 && gemm_goes_wide(M, N)) {
        // the producer's split-K reduce left fp16(X) in `scratch` and per-block sums of squares: no norm launch, the
        // wide kernel scales its rows instead
        c->xh_src = nullptr;
        c->use_row_scale = true;
        c->row_eps = eps;
        return gemm(c, s, scratch, K, W, M, N, K, nullptr, epi, nullptr, C, ldc, out_f32);
    }
    c->xh_src = nullptr;
    // 17..64 rows with a wide output (wgu, lm_head): the wide kernel wants fp16 activations, so norm separately
    const bool wide = M > SKINNY_MAX_M && N >= 16384 && !mid_v1;
    if (M <= SKINNY_MAX_M || (M <= MID_MAX_M && !no_mid && !wide))
        return gemm_any(c, s, nullptr, X, eps, K, W, M, N, K, bias, epi, nullptr, C, ldc, out_f32);
    KL(KC_NORM, 6.0 * M * K, launch_rmsnorm(X, nullptr, eps, M, K, scratch, s));
    return gemm(c, s, scratch, K, W, M, N, K, bias, epi, nullptr, C, ldc, out_f32);
}

static int need_ready(opus_ctx *c) {
    if (!c) return fail(OPUS_EBADARG, "ctx is null");
    OPC(opus_weights_ready(c));
    HIPC(hipSetDevice(c->device));
    return OPUS_OK;
}

// ------------------------------------------------------------------------------------------------ load-time ops
extern "C" int opus_lora_merge(void *W, const void *A, const void *B, float scale, int64_t out_f, int64_t in_f,
                               int32_t r, void *stream) {
    if (!W || !A || !B || out_f < 1 || in_f < 1 || r < 1) return fail(OPUS_EBADARG, "lora_merge: bad argument");
    if (in_f % 8) return fail(OPUS_ESHAPE, "lora_merge: in_features must be a multiple of 8");
    HIPC(launch_lora_merge((half_t *)W, (const half_t *)A, (const half_t *)B, scale, out_f, in_f, r, (hipStream_t)stream));
    return OPUS_OK;
}

extern "C" int opus_fill_synth(void *dst, int dtype, int64_t rows, int64_t cols, uint64_t seed, float std, float mean,
                               int64_t row_block, int64_t row_stride, int64_t row_off, int32_t tiled, uint64_t fold_seed,
                               float fold_std, float fold_mean, void *stream) {
    if (!dst || rows < 1 || cols < 1 || (dtype != OPUS_F16 && dtype != OPUS_F32) || row_block < 1 || row_stride < row_block)
        return fail(OPUS_EBADARG, "fill_synth: bad argument");
    if (tiled && (cols % 64 || row_block % 16 || row_stride % 16 || row_off % 16))
        return fail(OPUS_ESHAPE, "fill_synth: tiled layout needs cols %% 64 == 0 and 16-row aligned blocks");
    HIPC(launch_fill_synth(dst, dtype, rows, cols, seed, std, mean, row_block, row_stride, row_off, tiled ? 1 : 0,
                           fold_seed, fold_std, fold_mean, (hipStream_t)stream));
    return OPUS_OK;
}

extern "C" int opus_tile_weight(const void *d_src, void *d_dst, int64_t N, int64_t K, void *stream) {
    if (!d_src || !d_dst || d_src == d_dst) return fail(OPUS_EBADARG, "tile_weight: null or aliased pointers");
    if (N < 16 || K < 64 || N % 16 || K % 64) return fail(OPUS_ESHAPE, "tile_weight: N %% 16 == 0 and K %% 64 == 0 required");
    HIPC(launch_tile_weight((const half_t *)d_src, (half_t *)d_dst, N, K, (hipStream_t)stream));
    return OPUS_OK;
}

// ------------------------------------------------------------------------------------------------ encoder
// The encoder over M token rows.  Padded form (d_cu == nullptr): M = B T rows, row b t at b T + t, keys >= lens[b] masked.
// Token-packed form (SURVEY 5 "length-bucketed / varlen batches"): the rows are the proteins' tokens back to back, d_cu[B + 1] their
// offsets, T the longest protein - the GEMMs multiply no padding and a batch of mixed lengths is ONE set of large launches
// instead of one per length bucket; positions come from a row -> position table (written by the embedding kernel), the attention
// takes its rows from cu.
static int esm2_encode_rows(opus_ctx *c, hipStream_t s, const int32_t *d_tokens, const int32_t *d_lens, const int32_t *d_cu, int B, int T,
                            int M, float *d_pooled) {
    const opus_config &g = c->cfg;
    c->phase = PH_ENCODE;
    const int D = g.enc_dim, F = g.enc_ffn, nh = g.enc_heads, hd = D / nh;
    const bool packed = d_cu != nullptr;
    HIPC(hipMemsetAsync(c->d_cnt, 0, HANDOFF_ERR * sizeof(int32_t), s));       // hand-off words start from zero on every call
    if (packed) KL(KC_OTHER, 4.0 * M * D, launch_esm_embed_packed(d_tokens, c->enc_emb, d_cu, B, T, D, c->e_x, c->e_pos, s));
    else KL(KC_OTHER, 4.0 * M * D, launch_esm_embed(d_tokens, c->enc_emb, B, T, D, c->e_x, s));
    // Pre-LN blocks with the LayerNorm fused around the big tiled GEMM (GemmParams::ln_*): the epilogue that writes
    // the residual stream (wo, fc2) also leaves fp16(x) and per-row partial sums, a small kernel turns those into (mu, rstd),
    // and the consuming projection (fc1, the next layer's QKV) runs on fp16(x) as it stands and applies
    // rstd (acc - mu s) + c2 in its epilogue - no pass over the fp32 stream in between.  Shapes the big kernel does not take
    // (few rows), the first layer, and OPUS_NO_LN_FUSION=1 use the stand-alone normalisation (x - mu) rstd (the affine part
    // lives in the folded weights either way).
    static const bool no_ln_env = getenv("OPUS_NO_LN_FUSION") != nullptr;      // A/B aid (also the knob of the same name)
    const bool no_ln_fusion = no_ln_env || g_knobs.no_ln_fusion;
    const bool qkv_pp = !no_ln_fusion && (D & 255) == 0 && gemm_goes_pp(M, 3 * D);
    const bool fc1_pp = !no_ln_fusion && (D & 255) == 0 && gemm_goes_pp(M, F);
    bool have_stat = false;                        // e_xn = fp16(x), e_stat = (mu, rstd) of the current residual stream
    auto finalize = [&]() -> int {
        KL(KC_NORM, 8.0 * M * (D / 64) + 8.0 * M, launch_ln_finalize(c->e_part, M, D / 64, D, g.enc_ln_eps, 0, c->e_stat, s));
        return OPUS_OK;
    };
    double attn_flops = 0.0;                       // 4 T_b^2 D per protein
    if (packed) for (int b = 0; b < B; ++b) { const double t = c->h_cu[b + 1] - c->h_cu[b]; attn_flops += 4.0 * t * t * D; }
    else attn_flops = 4.0 * B * (double)T * T * D;
    for (int l = 0; l < g.enc_layers; ++l) {
        const EncLayer &L = c->enc[l];
        if (have_stat && qkv_pp) { c->rq_ln_stat = c->e_stat; c->rq_ln_colsum = L.sqkv; }
        else KL(KC_NORM, 6.0 * M * D, launch_layernorm(c->e_x, nullptr, nullptr, g.enc_ln_eps, M, D, c->e_xn, nullptr, s));
        // q <- rotary(q * hd^-0.5), k <- rotary(k): in the projection's epilogue when the big tiled kernel takes it (head_dim 64),
        // else by the stand-alone kernel on the stored projection (same arithmetic)
        if (hd == 64) {
            c->rq_rope_cs = c->cs_enc; c->rq_rope_T = packed ? g.max_enc_tokens : T; c->rq_rope_cols = 2 * D; c->rq_rope_qcols = D;
            c->rq_rope_qscale = 1.0f / sqrtf((float)hd);
            c->rq_rope_pos = packed ? c->e_pos : nullptr;
        }
        OPC(gemm(c, s, c->e_xn, D, L.wqkv, M, 3 * D, D, L.bqkv, EPI_NONE, nullptr, c->e_qkv, 3 * D, 0));
        if (!c->rq_rope_done) {
            if (packed) KL(KC_OTHER, 8.0 * M * D, launch_esm_rope(c->e_qkv, c->cs_enc, 1, M, nh, hd, 1.0f / sqrtf((float)hd), s, c->e_pos));
            else KL(KC_OTHER, 8.0 * M * D, launch_esm_rope(c->e_qkv, c->cs_enc, B, T, nh, hd, 1.0f / sqrtf((float)hd), s));
        }
        AttnParams a;
        a.Q = c->e_qkv; a.K = c->e_qkv + D; a.V = c->e_qkv + 2 * D;
        a.q_sb = a.k_sb = a.v_sb = packed ? 0 : (int64_t)T * 3 * D;
        a.q_st = a.k_st = a.v_st = 3 * D;
        a.O = c->e_ctx; a.o_sb = packed ? 0 : (int64_t)T * D; a.o_st = D;
        a.kstart = nullptr; a.kend = packed ? nullptr : d_lens; a.cu = d_cu;
        a.B = B; a.T = T; a.heads = nh; a.group = 1; a.head_dim = hd; a.causal = 0; a.scale = 1.0f;
        // last layer: the <cls> / <eos> rows of representations[L] are dropped by the mean-pool (cstp_v3/modelling.py:52-54): they
        // are keys but not queries (512 residues: four query blocks instead of five).  The debug tap opus_esm2_last_hidden then
        // holds no representation in those rows; knob "enc_full_last_layer" = 1 computes them (parity tests of every token's state).
        a.q_trim = packed && l + 1 == g.enc_layers && T > 2 && !g_knobs.enc_full_last_layer ? 1 : 0;
        KLF(KC_ATTN_PREFILL, 8.0 * M * D, attn_flops, launch_attn_prefill(a, s));
        if (fc1_pp) { c->rq_ln_part = c->e_part; c->rq_ln_xh = c->e_xn; }
        OPC(gemm(c, s, c->e_ctx, D, L.wo, M, D, D, L.bo, EPI_NONE, c->e_x, c->e_x, D, 1));
        have_stat = c->rq_ln_done != 0;
        if (have_stat) { OPC(finalize()); c->rq_ln_stat = c->e_stat; c->rq_ln_colsum = L.s1; }
        else KL(KC_NORM, 6.0 * M * D, launch_layernorm(c->e_x, nullptr, nullptr, g.enc_ln_eps, M, D, c->e_xn, nullptr, s));
        OPC(gemm(c, s, c->e_xn, D, L.w1, M, F, D, L.b1, EPI_GELU, nullptr, c->e_h1, F, 0));
        if (qkv_pp && l + 1 < g.enc_layers) { c->rq_ln_part = c->e_part; c->rq_ln_xh = c->e_xn; }
        OPC(gemm(c, s, c->e_h1, F, L.w2, M, D, F, L.b2, EPI_NONE, c->e_x, c->e_x, D, 1));
        have_stat = c->rq_ln_done != 0;
        if (have_stat) OPC(finalize());
    }
    KL(KC_NORM, 8.0 * M * D, launch_layernorm(c->e_x, c->enc_lnfw, c->enc_lnfb, g.enc_ln_eps, M, D, nullptr, c->e_hid, s));
    if (packed) KL(KC_OTHER, 4.0 * M * D, launch_masked_mean_packed(c->e_hid, d_cu, B, D, d_pooled, s));
    else KL(KC_OTHER, 4.0 * M * D, launch_masked_mean(c->e_hid, d_lens, B, T, D, d_pooled, s));
    return OPUS_OK;
}

extern "C" int opus_esm2_encode(opus_ctx *c, const int32_t *d_tokens, const int32_t *d_lens, int32_t B, int32_t T,
                                float *d_pooled, void *stream) {
    OPC(need_ready(c));
    if (!d_tokens || !d_lens || !d_pooled) return fail(OPUS_EBADARG, "esm2_encode: null pointer");
    const opus_config &g = c->cfg;
    if (B < 1 || B > g.max_batch || T < 3 || T > g.max_enc_tokens)
        return fail(OPUS_ESHAPE, "esm2_encode: B=%d T=%d exceed capacity (%d, %d)", B, T, g.max_batch, g.max_enc_tokens);
    return esm2_encode_rows(c, (hipStream_t)stream, d_tokens, d_lens, nullptr, B, T, B * T, d_pooled);
}

extern "C" int opus_esm2_encode_packed(opus_ctx *c, const int32_t *d_tokens, const int32_t *h_cu, int32_t B, float *d_pooled,
                                       void *stream) {
    OPC(need_ready(c));
    if (!d_tokens || !h_cu || !d_pooled) return fail(OPUS_EBADARG, "esm2_encode_packed: null pointer");
    const opus_config &g = c->cfg;
    if (B < 1 || B > g.max_batch) return fail(OPUS_ESHAPE, "esm2_encode_packed: B=%d exceeds max_batch=%d", B, g.max_batch);
    if (h_cu[0] != 0) return fail(OPUS_EBADARG, "esm2_encode_packed: cu[0] must be 0");
    int Tmax = 0;
    for (int b = 0; b < B; ++b) {
        const int t = h_cu[b + 1] - h_cu[b];
        if (t < 2 || t > g.max_enc_tokens)     // (<cls> <eos> at least)
            return fail(OPUS_ESHAPE, "esm2_encode_packed: protein %d has %d tokens (2 .. max_enc_tokens=%d)", b, t, g.max_enc_tokens);
        Tmax = t > Tmax ? t : Tmax;
    }
    const int64_t M = h_cu[B];
    if (M > (int64_t)g.max_batch * g.max_enc_tokens) return fail(OPUS_ESHAPE, "esm2_encode_packed: %lld tokens exceed the workspace", (long long)M);
    hipStream_t s = (hipStream_t)stream;
    c->h_cu.assign(h_cu, h_cu + B + 1);                              // (host copy: per-protein FLOP accounting of the timing records)
    HIPC(launch_upload_i32(h_cu, B + 1, c->e_cu, s));                 // through the kernel arguments: no host buffer outlives the call
    return esm2_encode_rows(c, s, d_tokens, nullptr, c->e_cu, B, Tmax, (int)M, d_pooled);
}

extern "C" int opus_esm2_last_hidden(opus_ctx *c, float *d_out, int32_t B, int32_t T, void *stream) {
    if (!c || !d_out) return fail(OPUS_EBADARG, "null pointer");
    if (B < 1 || T < 1 || (int64_t)B * T > (int64_t)c->cfg.max_batch * c->cfg.max_enc_tokens) return fail(OPUS_ESHAPE, "B/T out of range");   // (B = 1, T = all tokens: the packed form)
    HIPC(hipMemcpyAsync(d_out, c->e_hid, (size_t)B * T * c->cfg.enc_dim * sizeof(float), hipMemcpyDeviceToDevice,
                        (hipStream_t)stream));
    return OPUS_OK;
}

// ------------------------------------------------------------------------------------------------ projectors
// The projectors also serve the batched stage of the two-stage pipeline (whole shards at M >= 512; the 8H x 8H
// GEMM runs 1.18 / 1.30 / 1.34 PFLOP/s at 512 / 1024 / 4096 rows).  Its workspace (0.55 GB at the Llama-3-8B shape) is not
// part of every context: the first call that brings more rows than max_batch allocates it.
constexpr int BIG_PROJ_ROWS = 4096;
static int ensure_proj_rows(opus_ctx *c, int B) {
    if (B <= c->proj_rows || c->proj_big) return OPUS_OK;
    const opus_config &g = c->cfg;
    const size_t PR = BIG_PROJ_ROWS > g.max_batch ? BIG_PROJ_ROWS : g.max_batch;
    const size_t De = g.enc_dim, Dm = g.has_protein_projector ? g.proj_dim : g.enc_dim, SW = (size_t)g.dec_dim * g.n_prot_tokens;
    const size_t bytes = (align_up(PR * De * 2) + align_up(PR * Dm * 2) + 2 * align_up(PR * SW * 2));
    HIPC(hipMalloc((void **)&c->proj_big, bytes));
    Carver k{c->proj_big};
    c->p_xn = k.take<half_t>(PR * De);
    c->p_y = k.take<half_t>(PR * Dm);
    c->p_z[0] = k.take<half_t>(PR * SW);
    c->p_z[1] = k.take<half_t>(PR * SW);
    c->proj_rows = (int)PR;
    return OPUS_OK;
}

// The projector workspace holds c->proj_rows rows; larger inputs (the batched stage of the two-stage pipeline, SURVEY 8f N3:
// whole dataset shards at M >= 512, where the switch-projector GEMMs are MFMA-bound) are processed in chunks of that size.
static int protein_projector_rows(opus_ctx *c, hipStream_t s, const float *d_pooled, int B, half_t *d_out) {
    const opus_config &g = c->cfg;
    const int De = g.enc_dim;
    HIPC(hipMemsetAsync(c->d_cnt, 0, HANDOFF_ERR * sizeof(int32_t), s));
    if (!g.has_protein_projector) {
        // IdentityModule.protein_forward (opus_arch.py:70-80): the pooled embedding itself feeds the switch projector,
        // whose autocast Linear rounds it to fp16
        KL(KC_OTHER, 6.0 * B * De, launch_f2h(d_pooled, (int64_t)B * De, d_out, s));
        return OPUS_OK;
    }
    KL(KC_NORM, 6.0 * B * De, launch_l2norm(d_pooled, B, De, c->p_xn, s));
    return gemm(c, s, c->p_xn, De, c->proj_w, B, g.proj_dim, De, c->proj_b, EPI_NONE, nullptr, d_out, g.proj_dim, 0);
}

static int switch_projector_rows(opus_ctx *c, hipStream_t s, const half_t *in, int B, half_t *d_out) {
    const opus_config &g = c->cfg;
    const int SW = g.dec_dim * g.n_prot_tokens;
    int din = g.has_protein_projector ? g.proj_dim : g.enc_dim;
    for (int i = 0; i < g.switch_depth; ++i) {
        const bool last = i + 1 == g.switch_depth;
        half_t *o = last ? d_out : c->p_z[i & 1];
        // nn.GELU() sits between the Linear layers (protein_mlp/builder.py:21-24): fused as the epilogue
        OPC(gemm(c, s, in, din, c->sw_w[i], B, SW, din, c->sw_b[i], last ? EPI_NONE : EPI_GELU, nullptr, o, SW, 0));
        in = o;
        din = SW;
    }
    return OPUS_OK;
}

extern "C" int opus_protein_projector(opus_ctx *c, const float *d_pooled, int32_t B, void *d_out, void *stream) {
    OPC(need_ready(c));
    if (!d_pooled || !d_out) return fail(OPUS_EBADARG, "protein_projector: null pointer");
    if (B < 1) return fail(OPUS_ESHAPE, "protein_projector: B=%d", B);
    const opus_config &g = c->cfg;
    const int64_t din = g.enc_dim, dout = g.has_protein_projector ? g.proj_dim : g.enc_dim;
    c->phase = PH_PROJECT;
    OPC(ensure_proj_rows(c, B));
    for (int r0 = 0; r0 < B; r0 += c->proj_rows) {
        const int n = B - r0 < c->proj_rows ? B - r0 : c->proj_rows;
        OPC(protein_projector_rows(c, (hipStream_t)stream, d_pooled + r0 * din, n, (half_t *)d_out + r0 * dout));
    }
    return OPUS_OK;
}

extern "C" int opus_switch_projector(opus_ctx *c, const void *d_in, int32_t B, void *d_out, void *stream) {
    OPC(need_ready(c));
    if (!d_in || !d_out) return fail(OPUS_EBADARG, "switch_projector: null pointer");
    if (B < 1) return fail(OPUS_ESHAPE, "switch_projector: B=%d", B);
    const opus_config &g = c->cfg;
    const int64_t din = g.has_protein_projector ? g.proj_dim : g.enc_dim, SW = (int64_t)g.dec_dim * g.n_prot_tokens;
    c->phase = PH_PROJECT;
    OPC(ensure_proj_rows(c, B));
    for (int r0 = 0; r0 < B; r0 += c->proj_rows) {
        const int n = B - r0 < c->proj_rows ? B - r0 : c->proj_rows;
        OPC(switch_projector_rows(c, (hipStream_t)stream, (const half_t *)d_in + r0 * din, n, (half_t *)d_out + r0 * SW));
    }
    return OPUS_OK;
}

extern "C" int opus_projector_forward(opus_ctx *c, const float *d_pooled, int32_t B, void *d_out, void *d_proj_out,
                                      void *stream) {
    OPC(need_ready(c));
    if (!d_pooled || !d_out) return fail(OPUS_EBADARG, "projector_forward: null pointer");
    if (B < 1) return fail(OPUS_ESHAPE, "projector_forward: B=%d", B);
    const opus_config &g = c->cfg;
    hipStream_t s = (hipStream_t)stream;
    const int64_t din = g.enc_dim, dmid = g.has_protein_projector ? g.proj_dim : g.enc_dim, SW = (int64_t)g.dec_dim * g.n_prot_tokens;
    c->phase = PH_PROJECT;
    OPC(ensure_proj_rows(c, B));
    for (int r0 = 0; r0 < B; r0 += c->proj_rows) {
        const int n = B - r0 < c->proj_rows ? B - r0 : c->proj_rows;
        OPC(protein_projector_rows(c, s, d_pooled + r0 * din, n, c->p_y));
        if (d_proj_out)
            HIPC(hipMemcpyAsync((half_t *)d_proj_out + r0 * dmid, c->p_y, (size_t)n * dmid * sizeof(half_t), hipMemcpyDeviceToDevice, s));
        OPC(switch_projector_rows(c, s, c->p_y, n, (half_t *)d_out + r0 * SW));
    }
    return OPUS_OK;
}

// ------------------------------------------------------------------------------------------------ splice
extern "C" int opus_splice_pad(opus_ctx *c, const int64_t *d_ids, const uint8_t *d_mask, int32_t B, int32_t Tt,
                               const void *d_prot, int32_t n_prot, int32_t inference_mode, int32_t max_length,
                               void *d_embeds, uint8_t *d_mask_out, int32_t *d_pos_out, int32_t *T_out, void *stream) {
    OPC(need_ready(c));
    if (!d_ids || !d_prot || !d_embeds || !d_mask_out || !d_pos_out || !T_out) return fail(OPUS_EBADARG, "splice_pad: null pointer");
    const opus_config &g = c->cfg;
    if (B < 1 || B > g.max_batch || Tt < 1) return fail(OPUS_ESHAPE, "splice_pad: B=%d T_text=%d out of range", B, Tt);
    hipStream_t s = (hipStream_t)stream;
    c->phase = PH_SPLICE;
    KL(KC_OTHER, 9.0 * B * Tt, launch_splice_plan(d_ids, d_mask, B, Tt, g.n_prot_tokens, max_length, g.dec_vocab, c->d_plan, s));
    int32_t tail[3];
    HIPC(hipMemcpyAsync(tail, c->d_plan + 4 * B, sizeof(tail), hipMemcpyDeviceToHost, s));
    HIPC(hipStreamSynchronize(s));
    if (tail[2]) return fail(OPUS_EBADARG, "splice_pad: token id outside [0, vocab) (and not -200)");
    if (tail[1] > n_prot) return fail(OPUS_ESHAPE, "splice_pad: rows consume %d protein blocks, %d supplied", tail[1], n_prot);
    if (tail[0] > g.max_prompt) return fail(OPUS_ESHAPE, "splice_pad: spliced length %d exceeds max_prompt=%d", tail[0], g.max_prompt);
    if (tail[0] < 1) return fail(OPUS_ESHAPE, "splice_pad: every row is empty");
    *T_out = tail[0];
    KL(KC_OTHER, 2.0 * B * tail[0] * g.dec_dim * 2,
       launch_splice_fill(d_ids, d_mask, B, Tt, (const half_t *)d_prot, g.n_prot_tokens, g.dec_dim, g.dec_vocab, c->dec_emb,
                          c->d_plan, tail[0], inference_mode ? 1 : 0, (half_t *)d_embeds, d_mask_out, d_pos_out, s));
    return OPUS_OK;
}

// ------------------------------------------------------------------------------------------------ decoder
static int lm_head(opus_ctx *c, hipStream_t s, int B) {
    const opus_config &g = c->cfg;   // final RMSNorm (folded weight) fused into the lm_head GEMM
    return gemm_norm(c, s, c->d_xl, g.dec_rms_eps, c->d_xln, c->lm_head, B, g.dec_vocab, g.dec_dim, EPI_NONE, c->d_logits,
                     g.dec_vocab, 1);
}


// d_step = {step counter = 0, T0 = prompt length}: the decode step's kernels read BOTH from the device (a captured step is
// then independent of the prompt length; through the kernel arguments, so the launch is stream-ordered and needs no host buffer)
static int reset_step(opus_ctx *c, hipStream_t s, int T0, int step = 0) {
    const int32_t v[2] = {step, T0};
    HIPC(launch_upload_i32(v, 2, c->d_step, s));
    return OPUS_OK;
}

// decode-step attention of layer l over the projection output in d_qkv (rows of the last prefill, T prompt positions)
// (slab_ks > 0: the QKV GEMM left slab_ks raw k-part slabs in the GEMM workspace and the sums of squares of its input rows in
// d_ssq: summed, scaled and biased by the attention kernel itself)
static int attn_decode(opus_ctx *c, hipStream_t s, int l, int B, int T, int slab_ks = 0, const float *bias = nullptr, int out_tiled = 0) {
    const opus_config &g = c->cfg;
    AttnDecodeParams a;
    a.qkv = c->d_qkv; a.slabs = nullptr; a.ks = 0; a.slab_stride = 0; a.row_ssq = nullptr; a.row_nblk = 0; a.eps = 0.f; a.K = 0;
    a.bias = nullptr;
    if (slab_ks > 0) {
        const int64_t QKVd = (int64_t)(g.dec_heads + 2 * g.dec_kv_heads) * g.dec_head_dim;
        a.qkv = nullptr; a.slabs = c->gemm_ws; a.ks = slab_ks; a.slab_stride = (int64_t)B * QKVd;
        a.row_ssq = c->d_ssq; a.row_nblk = c->ssq_nblk; a.eps = g.dec_rms_eps; a.K = g.dec_dim; a.bias = bias;
    }
    a.cs_row = c->cs_row; a.kstart = c->d_kstart; a.step = c->d_step; a.T0 = -1; a.nh = g.dec_heads; a.nkv = g.dec_kv_heads;   // (T0 < 0: d_step[1])
    a.kc = c->kc + l * c->cache_sl; a.vc = c->vc + l * c->cache_sl; a.cache_sb = c->cache_sb; a.cache_sh = c->cache_sh;
    a.ctx_cap = g.max_prompt + g.max_new_tokens; a.scale = 1.0f / sqrtf((float)g.dec_head_dim); a.out = c->d_ctx;
    a.out_tiled = out_tiled;
    // algorithmic bytes: the rows' K / V history once (+ the new token's q, k, v and the output)
    const double bytes = 4.0 * B * g.dec_kv_heads * g.dec_head_dim * (T + 1) + 2.0 * B * (2.0 * g.dec_heads + 2.0 * g.dec_kv_heads) * g.dec_head_dim;
    KL(KC_ATTN_DECODE, bytes, launch_attn_decode(a, B, g.dec_head_dim, s));
    return OPUS_OK;
}

// ------------------------------------------------------------------------------------------------ OPT / Galactica decoder
// transformers OPTDecoder with do_layer_norm_before (reference wrapper language_model/opus_opt.py:18-40):
//   x = inputs_embeds + pos;  per layer  x += Wo attn(LN1 x) + bo;  x += W2 gelu(W1 LN2 x + b1) + b2;  logits = lm_head LNf x.
// No rotary: the (cos, sin) table holds (1, 0), so the fused rotate-and-cache kernels only append to the KV cache; the
// query scale head_dim^-0.5 is applied to the scores in fp32.
static int opt_layer(opus_ctx *c, hipStream_t s, const DecLayer &L, int l, float *x, half_t *xn, int M, int B, int T, bool decode) {
    const opus_config &g = c->cfg;
    const int H = g.dec_dim, F = g.dec_ffn, nh = g.dec_heads, nkv = g.dec_kv_heads, hd = g.dec_head_dim;
    const int QKV = (nh + 2 * nkv) * hd, QD = nh * hd;
    KL(KC_NORM, 6.0 * M * H, launch_layernorm(x, L.ln1w, L.ln1b, g.dec_rms_eps, M, H, xn, nullptr, s));
    OPC(gemm(c, s, xn, H, L.wqkv, M, QKV, H, L.bqkv, EPI_NONE, nullptr, c->d_qkv, QKV, 0));
    if (decode) {
        OPC(attn_decode(c, s, l, B, T));
    } else {
        KL(KC_OTHER, 4.0 * M * QKV,
           launch_dec_rope_cache(c->d_qkv, c->cs_dec, c->d_kstart, B, T, nh, nkv, hd, c->kc + l * c->cache_sl,
                                 c->vc + l * c->cache_sl, c->cache_sb, c->cache_sh, s));
        AttnParams a;
        a.Q = c->d_qkv; a.K = c->d_qkv + QD; a.V = c->d_qkv + QD + nkv * hd;
        a.q_sb = a.k_sb = a.v_sb = (int64_t)T * QKV;
        a.q_st = a.k_st = a.v_st = QKV;
        a.O = c->d_ctx; a.o_sb = (int64_t)T * QD; a.o_st = QD;
        a.kstart = c->d_kstart; a.kend = nullptr;
        a.B = B; a.T = T; a.heads = nh; a.group = nh / nkv; a.head_dim = hd; a.causal = 1;
        a.scale = 1.0f / sqrtf((float)hd);
        KLF(KC_ATTN_PREFILL, 2.0 * M * (QKV + QD), 2.0 * B * (double)T * T * QD, launch_attn_prefill(a, s));
    }
    OPC(gemm(c, s, c->d_ctx, QD, L.wo, M, H, QD, L.bo, EPI_NONE, x, x, H, 1));
    KL(KC_NORM, 6.0 * M * H, launch_layernorm(x, L.ln2w, L.ln2b, g.dec_rms_eps, M, H, xn, nullptr, s));
    if (g.dec_act == 0) {
        OPC(gemm(c, s, xn, H, L.w1, M, F, H, L.b1, EPI_GELU, nullptr, c->d_act, F, 0));
    } else {      // ReLU (facebook/opt-*): on the stored fp16 projection, as HF applies it to fc1's fp16 output
        OPC(gemm(c, s, xn, H, L.w1, M, F, H, L.b1, EPI_NONE, nullptr, c->d_act, F, 0));
        KL(KC_OTHER, 4.0 * M * F, launch_relu_h(c->d_act, (int64_t)M * F, s));
    }
    OPC(gemm(c, s, c->d_act, F, L.w2, M, H, F, L.b2, EPI_NONE, x, x, H, 1));
    return OPUS_OK;
}

static int lm_head_opt(opus_ctx *c, hipStream_t s, int B) {
    const opus_config &g = c->cfg;
    KL(KC_NORM, 6.0 * B * g.dec_dim, launch_layernorm(c->d_xl, c->dec_lnfw, c->dec_lnfb, g.dec_rms_eps, B, g.dec_dim, c->d_xln, nullptr, s));
    return gemm(c, s, c->d_xln, g.dec_dim, c->lm_head, B, g.dec_vocab, g.dec_dim, nullptr, EPI_NONE, nullptr, c->d_logits,
                g.dec_vocab, 1);
}

static int prefill_opt(opus_ctx *c, hipStream_t s, const half_t *embeds, const uint8_t *mask, int B, int T) {
    const opus_config &g = c->cfg;
    const int H = g.dec_dim, M = B * T;
    HIPC(hipMemsetAsync(c->d_cnt, 0, HANDOFF_ERR * sizeof(int32_t), s));
    KL(KC_OTHER, 1.0 * M, launch_mask_to_kstart(mask, B, T, c->d_kstart, s));
    KL(KC_OTHER, 6.0 * M * H, launch_h2f(embeds, c->d_x, (int64_t)M * H, s));
    KL(KC_OTHER, 10.0 * M * H, launch_add_pos(c->d_x, c->dec_pos, c->d_kstart, nullptr, 0, B, T, H, g.dec_max_pos + 1, s));
    for (int l = 0; l < g.dec_layers; ++l) OPC(opt_layer(c, s, c->dec[l], l, c->d_x, c->d_xn, M, B, T, false));
    KL(KC_OTHER, 8.0 * B * H, launch_take_last(c->d_x, B, T, H, c->d_xl, s));
    OPC(lm_head_opt(c, s, B));
    OPC(reset_step(c, s, T));
    c->cur_B = B;
    c->cur_T = T;
    c->prefilled = true;
    return OPUS_OK;
}

// the embedded new tokens are already in d_xl (decode_step)
static int decode_step_opt(opus_ctx *c, hipStream_t s) {
    const opus_config &g = c->cfg;
    const int B = c->cur_B, T = c->cur_T, H = g.dec_dim;
    KL(KC_OTHER, 10.0 * B * H, launch_add_pos(c->d_xl, c->dec_pos, c->d_kstart, c->d_step, -1, B, 1, H, g.dec_max_pos + 1, s));   // (t0 < 0: d_step[1])
    for (int l = 0; l < g.dec_layers; ++l) OPC(opt_layer(c, s, c->dec[l], l, c->d_xl, c->d_xln, B, B, T, true));
    OPC(lm_head_opt(c, s, B));
    KL(KC_OTHER, 8.0, launch_step_advance(c->d_step, s));
    return OPUS_OK;
}

static int prefill(opus_ctx *c, hipStream_t s, const half_t *embeds, const uint8_t *mask, int B, int T) {
    c->phase = PH_PREFILL;
    if (c->cfg.dec_arch == 1) return prefill_opt(c, s, embeds, mask, B, T);
    const opus_config &g = c->cfg;
    const int H = g.dec_dim, F = g.dec_ffn, nh = g.dec_heads, nkv = g.dec_kv_heads, hd = g.dec_head_dim;
    const int QKV = (nh + 2 * nkv) * hd, QD = nh * hd;
    const int M = B * T;
    HIPC(hipMemsetAsync(c->d_cnt, 0, HANDOFF_ERR * sizeof(int32_t), s));       // hand-off words start from zero on every call
    KL(KC_OTHER, 1.0 * M, launch_mask_to_kstart(mask, B, T, c->d_kstart, s));
    KL(KC_OTHER, 6.0 * M * H, launch_h2f(embeds, c->d_x, (int64_t)M * H, s));
    // RMSNorm fused around the big tiled GEMM, as the encoder's LayerNorm (GemmParams::ln_* with mu = 0): the wo / down epilogue
    // leaves fp16(x) + per-slab sums of squares, the consuming projection scales its rows by rstd in its epilogue
    static const bool no_ln_env = getenv("OPUS_NO_LN_FUSION") != nullptr;      // A/B aid (also the knob of the same name)
    const bool no_ln_fusion = no_ln_env || g_knobs.no_ln_fusion;
    const bool qkv_pp = !no_ln_fusion && (H & 255) == 0 && (QKV & 255) == 0 && gemm_goes_pp(M, QKV);
    const bool gu_pp = !no_ln_fusion && (H & 255) == 0 && (F & 127) == 0 && gemm_goes_pp(M, 2 * F);
    bool have_stat = false, last_rows_done = false;
    auto finalize = [&]() -> int {
        KL(KC_NORM, 8.0 * M * (H / 64) + 8.0 * M, launch_ln_finalize(c->d_part, M, H / 64, H, g.dec_rms_eps, 1, c->d_stat, s));
        return OPUS_OK;
    };
    for (int l = 0; l < g.dec_layers; ++l) {
        const DecLayer &L = c->dec[l];
        if (have_stat && qkv_pp) {
            c->rq_ln_stat = c->d_stat;
            OPC(gemm(c, s, c->d_xn, H, L.wqkv, M, QKV, H, L.bqkv, EPI_NONE, nullptr, c->d_qkv, QKV, 0));
        } else {
            OPC(gemm_norm(c, s, c->d_x, g.dec_rms_eps, c->d_xn, L.wqkv, M, QKV, H, EPI_NONE, c->d_qkv, QKV, 0, L.bqkv));
        }
        KL(KC_OTHER, 4.0 * M * QKV,
           launch_dec_rope_cache(c->d_qkv, c->cs_dec, c->d_kstart, B, T, nh, nkv, hd, c->kc + l * c->cache_sl,
                                 c->vc + l * c->cache_sl, c->cache_sb, c->cache_sh, s));
        AttnParams a;
        a.Q = c->d_qkv; a.K = c->d_qkv + QD; a.V = c->d_qkv + QD + nkv * hd;
        a.q_sb = a.k_sb = a.v_sb = (int64_t)T * QKV;
        a.q_st = a.k_st = a.v_st = QKV;
        a.O = c->d_ctx; a.o_sb = (int64_t)T * QD; a.o_st = QD;
        a.kstart = c->d_kstart; a.kend = nullptr;
        a.B = B; a.T = T; a.heads = nh; a.group = nh / nkv; a.head_dim = hd; a.causal = 1;
        a.scale = 1.0f / sqrtf((float)hd);
        KLF(KC_ATTN_PREFILL, 2.0 * M * (QKV + QD), 2.0 * B * (double)T * T * QD, launch_attn_prefill(a, s));
        if (l + 1 == g.dec_layers && T > 1 && !g_knobs.misc[7]) {
            // Last layer: its K / V are in the cache for every position, but behind the attention only the LAST position of a row
            // is ever used (lm_head reads that row alone, opus_arch.py -> HF generate keeps the last logits): wo, gate/up and
            // down run on B rows with the decode step's kernels instead of on B T rows (Llama-3-8B, 64 x 96: ~1.85 ms of tiled
            // GEMMs -> ~0.1 ms; same values for the rows that are kept).  Knob misc7 = 1: all rows, as every other layer.
            half_t *ctx_last = c->d_qkv;                              // (the projections are consumed: the buffer is free)
            KL(KC_OTHER, 4.0 * B * QD, launch_take_last(reinterpret_cast<const float *>(c->d_ctx), B, T, QD / 2, reinterpret_cast<float *>(ctx_last), s));
            KL(KC_OTHER, 8.0 * B * H, launch_take_last(c->d_x, B, T, H, c->d_xl, s));
            if (fuse_rows() && B <= 96) c->rq_xh = c->d_xln;
            OPC(gemm(c, s, ctx_last, QD, L.wo, B, H, QD, nullptr, EPI_NONE, c->d_xl, c->d_xl, H, 1));
            c->xh_src = c->rq_done ? c->d_xl : nullptr;
            OPC(gemm_norm(c, s, c->d_xl, g.dec_rms_eps, c->d_xln, L.wgu, B, 2 * F, H, EPI_SILU_GU16, c->d_act, F, 0));
            if (fuse_rows() && B <= 96) c->rq_xh = c->d_xln;
            OPC(gemm(c, s, c->d_act, F, L.wd, B, H, F, nullptr, EPI_NONE, c->d_xl, c->d_xl, H, 1));
            c->xh_src = c->rq_done ? c->d_xl : nullptr;
            c->xln_tiled = false;
            last_rows_done = true;
            break;
        }
        if (gu_pp) { c->rq_ln_part = c->d_part; c->rq_ln_xh = c->d_xn; }
        else if (fuse_rows() && M <= 96) c->rq_xh = c->d_xn;      // (d_ssq holds 128 rows)
        OPC(gemm(c, s, c->d_ctx, QD, L.wo, M, H, QD, nullptr, EPI_NONE, c->d_x, c->d_x, H, 1));
        c->xh_src = c->rq_done ? c->d_x : nullptr;
        have_stat = c->rq_ln_done != 0;
        if (have_stat) {
            OPC(finalize());
            c->rq_ln_stat = c->d_stat;
            OPC(gemm(c, s, c->d_xn, H, L.wgu, M, 2 * F, H, nullptr, EPI_SILU_GU16, nullptr, c->d_act, F, 0));
        } else {
            OPC(gemm_norm(c, s, c->d_x, g.dec_rms_eps, c->d_xn, L.wgu, M, 2 * F, H, EPI_SILU_GU16, c->d_act, F, 0));
        }
        if (qkv_pp && l + 1 < g.dec_layers) { c->rq_ln_part = c->d_part; c->rq_ln_xh = c->d_xn; }
        OPC(gemm(c, s, c->d_act, F, L.wd, M, H, F, nullptr, EPI_NONE, c->d_x, c->d_x, H, 1));
        have_stat = c->rq_ln_done != 0;
        if (have_stat) OPC(finalize());
    }
    if (!last_rows_done) {
        KL(KC_OTHER, 8.0 * B * H, launch_take_last(c->d_x, B, T, H, c->d_xl, s));
        c->xh_src = nullptr;
    }
    OPC(lm_head(c, s, B));
    OPC(reset_step(c, s, T));
    c->cur_B = B;
    c->cur_T = T;
    c->prefilled = true;
    return OPUS_OK;
}

// One decode step for the token ids in d_tok (device): embeds them, runs the stack at slot T + *step,
// leaves logits in c->d_logits and advances *step.
static int decode_step(opus_ctx *c, hipStream_t s, const int32_t *d_tok) {
    c->phase = PH_DECODE;
    const opus_config &g = c->cfg;
    const int B = c->cur_B, T = c->cur_T;
    const int H = g.dec_dim, F = g.dec_ffn, nh = g.dec_heads, nkv = g.dec_kv_heads, hd = g.dec_head_dim;
    const int QKV = (nh + 2 * nkv) * hd, QD = nh * hd;
    const int ctx_cap = g.max_prompt + g.max_new_tokens;
    // batched Llama / Qwen2 step: the embedding kernel also leaves fp16(x) and its per-block sums of squares, so that the first
    // layer's QKV GEMM can take the row-scale RMSNorm form like every later one (whose producer is the previous down GEMM)
    const bool rowscale = g.dec_arch == 0 && fuse_rows() && B > SKINNY_MAX_M && B <= MID_MAX_M && (H & 255) == 0;
    // Fragment-ordered fp16 activations for the GEMMs that gemm_stream_kernel will take (gemm_stream.hip "Activation layout"):
    // the producer of each such matrix is told to write that layout - fp16(x) for the QKV projection (embedding kernel for
    // layer 0, the down projection's epilogue / reduce afterwards) and the attention output for the wo projection.
    const bool qkv_tiled = rowscale && gemm_stream_would(B, QKV, H, 1, 1, 1, c->gemm_ws_bytes);
    const bool wo_tiled = rowscale && gemm_stream_would(B, H, QD, 0, 1, 0, c->gemm_ws_bytes);
    const bool down_tiled = rowscale && (F & 63) == 0 && gemm_stream_would(B, H, F, 0, 1, 0, c->gemm_ws_bytes);
    KL(KC_OTHER, 6.0 * B * H, launch_embed_tokens(d_tok, c->dec_emb, B, H, g.dec_vocab, c->d_xl, rowscale ? c->d_xln : nullptr,
                                                  rowscale ? c->d_ssq : nullptr, qkv_tiled ? 1 : 0, c->cs_dec, c->d_kstart, c->d_step, -1,
                                                  hd / 2, c->cs_row, c->d_cnt, HANDOFF_ERR, s));     // (+ zeroes the hand-off words; T0 < 0: d_step[1])
    c->xln_tiled = qkv_tiled;
    c->xh_src = rowscale ? c->d_xl : nullptr;
    if (rowscale) c->ssq_nblk = H >> 8;
    if (g.dec_arch == 1) return decode_step_opt(c, s);
    for (int l = 0; l < g.dec_layers; ++l) {
        const DecLayer &L = c->dec[l];
        if (rowscale && c->xh_src == c->d_xl) {
            // x is available as fp16 + sums of squares (from the embedding or the previous layer's down GEMM): the QKV
            // projection streams its weights through the wide kernel with k-parts and leaves the raw slabs; the attention
            // kernel sums them, applies the RMSNorm row scale and the bias on the fly - no norm launch, no reduce launch
            c->xh_src = nullptr;
            c->rq_force_wide = 1;
            c->rq_slab_only = 1;
            c->use_row_scale = true;
            c->row_eps = g.dec_rms_eps;
            c->rq_a_tiled = c->xln_tiled ? 1 : 0;
            OPC(gemm(c, s, c->d_xln, H, L.wqkv, B, QKV, H, L.bqkv, EPI_NONE, nullptr, c->d_qkv, QKV, 0));
            OPC(attn_decode(c, s, l, B, T, c->rq_ks > 1 ? c->rq_ks : 0, L.bqkv, wo_tiled ? 1 : 0));
        } else {
            OPC(gemm_norm(c, s, c->d_xl, g.dec_rms_eps, c->d_xln, L.wqkv, B, QKV, H, EPI_NONE, c->d_qkv, QKV, 0, L.bqkv));
            OPC(attn_decode(c, s, l, B, T, 0, nullptr, wo_tiled ? 1 : 0));
        }
        c->xln_tiled = false;
        if (fuse_rows() && B <= 96) c->rq_xh = c->d_xln;                 // (row-major: the gate/up kernel stages it by LDS-DMA)
        c->rq_a_tiled = wo_tiled ? 1 : 0;
        OPC(gemm(c, s, c->d_ctx, QD, L.wo, B, H, QD, nullptr, EPI_NONE, c->d_xl, c->d_xl, H, 1));
        c->xh_src = c->rq_done ? c->d_xl : nullptr;
        // the gate / up kernel writes silu(g) u in fragment order when the down projection will read it that way (only the wide
        // kernel - the row-scale form of gemm_norm - writes that layout)
        const bool act_tiled = down_tiled && c->xh_src == c->d_xl && gemm_goes_wide(B, 2 * F);
        c->rq_c_tiled = act_tiled ? 1 : 0;
        OPC(gemm_norm(c, s, c->d_xl, g.dec_rms_eps, c->d_xln, L.wgu, B, 2 * F, H, EPI_SILU_GU16, c->d_act, F, 0));
        const bool next_tiled = qkv_tiled && l + 1 < g.dec_layers;      // (the last layer's fp16(x) feeds lm_head: row-major)
        if (fuse_rows() && B <= 96) { c->rq_xh = c->d_xln; c->rq_xh_tiled = next_tiled ? 1 : 0; }
        c->rq_a_tiled = act_tiled ? 1 : 0;
        OPC(gemm(c, s, c->d_act, F, L.wd, B, H, F, nullptr, EPI_NONE, c->d_xl, c->d_xl, H, 1));
        c->xh_src = c->rq_done ? c->d_xl : nullptr;
        c->xln_tiled = c->rq_done && next_tiled;
    }
    OPC(lm_head(c, s, B));
    KL(KC_OTHER, 8.0, launch_step_advance(c->d_step, s));
    return OPUS_OK;
}

// Synchronises `s` and reports an in-launch hand-off that gave up waiting (GemmParams::combine_cnt[HANDOFF_ERR], set by a kernel
// whose bounded wait ran out: poisoned ticket words, a partner that never arrived).  The results of that call are not to be used.
static int handoff_check(opus_ctx *c, hipStream_t s) {
    int32_t err = 0;
    HIPC(hipMemcpyAsync(&err, c->d_cnt + HANDOFF_ERR, sizeof(err), hipMemcpyDeviceToHost, s));
    HIPC(hipStreamSynchronize(s));
    if (!err) return OPUS_OK;
    HIPC(hipMemsetAsync(c->d_cnt, 0, HANDOFF_WORDS * sizeof(int32_t), s));
    return fail(OPUS_EHIP, "an in-launch hand-off (split-K combine) gave up waiting for its partner: results of this call are invalid");
}

extern "C" int opus_check_error(opus_ctx *c, void *stream) {
    if (!c) return fail(OPUS_EBADARG, "ctx is null");
    HIPC(hipSetDevice(c->device));
    return handoff_check(c, (hipStream_t)stream);
}

static int check_prefill_args(opus_ctx *c, const void *e, const uint8_t *m, int B, int T) {
    if (!e || !m) return fail(OPUS_EBADARG, "prefill: null pointer");
    if (B < 1 || B > c->cfg.max_batch || T < 1 || T > c->cfg.max_prompt)
        return fail(OPUS_ESHAPE, "prefill: B=%d T=%d exceed capacity (%d, %d)", B, T, c->cfg.max_batch, c->cfg.max_prompt);
    return OPUS_OK;
}

extern "C" int opus_llama_prefill(opus_ctx *c, const void *d_embeds, const uint8_t *d_mask, int32_t B, int32_t T,
                                  float *d_last_logits, void *stream) {
    OPC(need_ready(c));
    OPC(check_prefill_args(c, d_embeds, d_mask, B, T));
    hipStream_t s = (hipStream_t)stream;
    OPC(prefill(c, s, (const half_t *)d_embeds, d_mask, B, T));
    if (d_last_logits)
        HIPC(hipMemcpyAsync(d_last_logits, c->d_logits, (size_t)B * c->cfg.dec_vocab * sizeof(float), hipMemcpyDeviceToDevice, s));
    return OPUS_OK;
}

extern "C" int opus_llama_decode_step(opus_ctx *c, const int32_t *d_tok, float *d_logits, void *stream) {
    OPC(need_ready(c));
    if (!c->prefilled) return fail(OPUS_ESTATE, "decode_step before prefill");
    if (!d_tok) return fail(OPUS_EBADARG, "decode_step: null pointer");
    hipStream_t s = (hipStream_t)stream;
    int32_t st = 0;
    HIPC(hipMemcpyAsync(&st, c->d_step, sizeof(st), hipMemcpyDeviceToHost, s));
    HIPC(hipStreamSynchronize(s));
    if (st >= c->cfg.max_new_tokens) return fail(OPUS_ESHAPE, "decode_step: KV cache is full (%d steps)", st);
    OPC(decode_step(c, s, d_tok));
    if (d_logits)
        HIPC(hipMemcpyAsync(d_logits, c->d_logits, (size_t)c->cur_B * c->cfg.dec_vocab * sizeof(float), hipMemcpyDeviceToDevice, s));
    return OPUS_OK;
}

// next token per row: argmax (greedy) or temperature / top-p sampling, then the GenerationMixin bookkeeping
static int argmax(opus_ctx *c, hipStream_t s, int max_new, int n_eos, int pad_id, int32_t *d_out) {
    c->phase = PH_DECODE;
    const opus_config &g = c->cfg;
    const int32_t *chosen = nullptr;
    if (c->samp_temp > 0.f) {
        KL(KC_OTHER, 4.0 * 4 * c->cur_B * g.dec_vocab,
           launch_sample_select(c->d_logits, c->cur_B, g.dec_vocab, c->samp_temp, c->samp_top_p, c->samp_top_k, c->d_seed, c->d_step,
                                c->d_pval, c->d_pidx, c->d_probs, c->d_cand_i, c->d_cand_n, c->d_zpart, c->d_spart, c->d_chosen, nullptr, s));
        chosen = c->d_chosen;
    } else {
        KL(KC_OTHER, 4.0 * c->cur_B * g.dec_vocab, launch_argmax_partial(c->d_logits, c->cur_B, g.dec_vocab, c->d_pval, c->d_pidx, s));
    }
    KL(KC_OTHER, 512.0 * c->cur_B,
       launch_argmax_step(c->d_pval, c->d_pidx, chosen, c->cur_B, c->d_eos, n_eos, pad_id, c->d_fin, d_out, max_new, c->d_step,
                          c->d_next, c->d_nunf, c->d_stop, c->n_stop, s));
    return OPUS_OK;
}

// iteration body of the greedy loop: pick the next token from the current logits, then decode it.
static int greedy_body(opus_ctx *c, hipStream_t s, int max_new, int n_eos, int pad_id, int32_t *d_out) {
    const opus_config &g = c->cfg;
    OPC(argmax(c, s, max_new, n_eos, pad_id, d_out));
    return decode_step(c, s, c->d_next);
}

static int generate_impl(opus_ctx *c, const void *d_embeds, const uint8_t *d_mask, int32_t B, int32_t T, int32_t max_new,
                         const int32_t *eos_ids, int32_t n_eos, int32_t pad_id, float temperature, float top_p, uint64_t seed,
                         int32_t *d_out_ids, int32_t *n_out, void *stream) {
    OPC(need_ready(c));
    if (temperature < 0.f || top_p <= 0.f || top_p > 1.f) return fail(OPUS_EBADARG, "generate: temperature >= 0 and 0 < top_p <= 1");
    c->samp_temp = temperature;
    c->samp_top_p = top_p;
    OPC(check_prefill_args(c, d_embeds, d_mask, B, T));
    if (!d_out_ids || !n_out) return fail(OPUS_EBADARG, "generate_greedy: null pointer");
    if (max_new < 1 || max_new > c->cfg.max_new_tokens)
        return fail(OPUS_ESHAPE, "generate_greedy: max_new=%d exceeds max_new_tokens=%d", max_new, c->cfg.max_new_tokens);
    if (n_eos < 0 || n_eos > 64 || (n_eos > 0 && !eos_ids)) return fail(OPUS_EBADARG, "generate_greedy: eos list");
    hipStream_t s = (hipStream_t)stream;
    const opus_config &g = c->cfg;
    if (n_eos) HIPC(hipMemcpyAsync(c->d_eos, eos_ids, n_eos * sizeof(int32_t), hipMemcpyHostToDevice, s));
    HIPC(hipMemcpyAsync(c->d_seed, &seed, sizeof(seed), hipMemcpyHostToDevice, s));
    HIPC(hipMemsetAsync(c->d_fin, 0, B * sizeof(int32_t), s));
    HIPC(hipMemsetAsync(c->d_nunf, 0, (size_t)max_new * sizeof(int32_t), s));
    OPC(prefill(c, s, (const half_t *)d_embeds, d_mask, B, T));

    // OPUS_NO_GRAPH=1: eager launches (rocprofv3 --pmc cannot collect counters through graph replays)
    const bool use_graph = s != nullptr && !c->timing && !getenv("OPUS_NO_GRAPH");
    auto find_graph = [&]() -> opus_ctx::GraphEntry * {
        for (auto &e : c->graphs)
            if (e.exec && e.B == B && e.maxnew == max_new && e.pad == pad_id && e.neos == n_eos && e.out == d_out_ids &&
                e.temp == temperature && e.top_p == top_p) return &e;
        return nullptr;
    };
    opus_ctx::GraphEntry *ge = use_graph ? find_graph() : nullptr;
    const bool same_graph = ge != nullptr;
    std::vector<int32_t> nunf(max_new, 1);
    int produced = 0;
    for (int i = 0; i < max_new; ++i) {
        if (i + 1 == max_new) {   // last token: no decode behind it
            OPC(argmax(c, s, max_new, n_eos, pad_id, d_out_ids));
            produced = i + 1;
            break;
        }
        ++c->decode_steps;
        if (!use_graph || (i == 0 && !same_graph)) {
            OPC(greedy_body(c, s, max_new, n_eos, pad_id, d_out_ids));   // eager (also warms lazily-set attributes)
        } else {
            if (!ge) {
                hipGraph_t graph = nullptr;
                HIPC(hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal));
                int rc = greedy_body(c, s, max_new, n_eos, pad_id, d_out_ids);
                hipError_t ee = hipStreamEndCapture(s, &graph);
                if (rc != OPUS_OK) { if (graph) (void)hipGraphDestroy(graph); return rc; }
                if (ee != hipSuccess) return fail(OPUS_EHIP, "hipStreamEndCapture: %s", hipGetErrorString(ee));
                opus_ctx::GraphEntry e;
                ee = hipGraphInstantiate(&e.exec, graph, nullptr, nullptr, 0);
                (void)hipGraphDestroy(graph);
                if (ee != hipSuccess) return fail(OPUS_EHIP, "hipGraphInstantiate: %s", hipGetErrorString(ee));
                ++c->graph_instantiations;
                e.B = B; e.maxnew = max_new; e.pad = pad_id; e.neos = n_eos; e.temp = temperature; e.top_p = top_p; e.out = d_out_ids;
                if ((int)c->graphs.size() >= opus_ctx::MAX_GRAPHS) {           // least recently used out
                    size_t v = 0;
                    for (size_t k = 1; k < c->graphs.size(); ++k) if (c->graphs[k].used < c->graphs[v].used) v = k;
                    (void)hipGraphExecDestroy(c->graphs[v].exec);
                    c->graphs.erase(c->graphs.begin() + v);
                }
                c->graphs.push_back(e);
                ge = &c->graphs.back();
            }
            ge->used = ++c->graph_clock;
            ++c->graph_replays;
            HIPC(hipGraphLaunch(ge->exec, s));
        }
        produced = i + 1;
        // HF stops as soon as every row has finished.  The host runs ahead of the GPU, so it polls with a BOUNDED run-ahead instead
        // of a synchronisation per step: step i's unfinished count is copied to pinned memory behind an event, and before step
        // i + 1 is enqueued the host waits for step i - 2's event (the GPU still has a step queued: no bubble) and stops when that
        // count is zero - at most 2 steps are decoded past the last row's EOS (rounds 1-4: a stream synchronisation every 8 steps,
        // up to 7 wasted steps).  Finished rows emit pad, so the ids are identical and n_out is computed exactly below.
        if (n_eos > 0 || c->n_stop > 0) {
            HIPC(hipMemcpyAsync(c->h_nunf + i, c->d_nunf + i, sizeof(int32_t), hipMemcpyDeviceToHost, s));
            HIPC(hipEventRecord(c->poll_ev[i % opus_ctx::POLL_RING], s));
            if (i >= opus_ctx::POLL_LAG) {
                HIPC(hipEventSynchronize(c->poll_ev[(i - opus_ctx::POLL_LAG) % opus_ctx::POLL_RING]));
                if (c->h_nunf[i - opus_ctx::POLL_LAG] == 0) break;
            }
        }
    }
    HIPC(hipMemcpyAsync(nunf.data(), c->d_nunf, (size_t)produced * sizeof(int32_t), hipMemcpyDeviceToHost, s));
    OPC(handoff_check(c, s));                                         // (synchronises the stream)
    int n = produced;
    for (int k = 0; k < produced; ++k) if (nunf[k] == 0) { n = k + 1; break; }
    *n_out = n;
    return OPUS_OK;
}

// Opt-in early stop on a token sequence ("### early-stop as an opt-in").  The reference decodes to
// max_new_tokens and cuts the TEXT at the first "###" afterwards (eval/run_opus_ddp.py:19-27); with the ids of "###" set
// here a row is finished as soon as its new ids end with them (later positions hold pad_id, as after an EOS), which leaves
// the cut text unchanged and lets a batch stop early.  n = 0 clears it.  Host ids, at most 8.
extern "C" int opus_set_stop_sequence(opus_ctx *c, const int32_t *ids, int32_t n) {
    if (!c || n < 0 || n > 8 || (n > 0 && !ids)) return fail(OPUS_EBADARG, "set_stop_sequence: 0 <= n <= 8 ids");
    HIPC(hipSetDevice(c->device));
    if (n) HIPC(hipMemcpy(c->d_stop, ids, n * sizeof(int32_t), hipMemcpyHostToDevice));
    c->n_stop = n;
    drop_graphs(c);                                                               // the captured step holds n_stop
    return OPUS_OK;
}

extern "C" int opus_generate_greedy(opus_ctx *c, const void *d_embeds, const uint8_t *d_mask, int32_t B, int32_t T,
                                    int32_t max_new, const int32_t *eos_ids, int32_t n_eos, int32_t pad_id,
                                    int32_t *d_out_ids, int32_t *n_out, void *stream) {
    return generate_impl(c, d_embeds, d_mask, B, T, max_new, eos_ids, n_eos, pad_id, 0.f, 1.f, 0, d_out_ids, n_out, stream);
}

extern "C" int opus_generate_sample(opus_ctx *c, const void *d_embeds, const uint8_t *d_mask, int32_t B, int32_t T,
                                    int32_t max_new, const int32_t *eos_ids, int32_t n_eos, int32_t pad_id, float temperature,
                                    float top_p, uint64_t seed, int32_t *d_out_ids, int32_t *n_out, void *stream) {
    if (!(temperature > 0.f)) return fail(OPUS_EBADARG, "generate_sample: temperature must be > 0 (use opus_generate_greedy)");
    return generate_impl(c, d_embeds, d_mask, B, T, max_new, eos_ids, n_eos, pad_id, temperature, top_p, seed, d_out_ids, n_out,
                         stream);
}

/* Diagnostic: one draw per row from fp32 logits [B,V] with the sampling head (step counter = `step`). */
extern "C" int opus_debug_sample(opus_ctx *c, const float *d_logits, int32_t B, float temperature, float top_p, uint64_t seed,
                                 int32_t step, int32_t *d_tokens, void *stream) {
    if (!c || !d_logits || !d_tokens) return fail(OPUS_EBADARG, "debug_sample: null pointer");
    if (B < 1 || B > c->cfg.max_batch || !(temperature > 0.f) || top_p <= 0.f || top_p > 1.f) return fail(OPUS_EBADARG, "debug_sample: argument");
    HIPC(hipSetDevice(c->device));
    hipStream_t s = (hipStream_t)stream;
    HIPC(hipMemcpyAsync(c->d_seed, &seed, sizeof(seed), hipMemcpyHostToDevice, s));
    HIPC(hipMemcpyAsync(c->d_plan, &step, sizeof(step), hipMemcpyHostToDevice, s));
    HIPC(launch_sample_select(d_logits, B, c->cfg.dec_vocab, temperature, top_p, c->samp_top_k, c->d_seed, c->d_plan, c->d_pval, c->d_pidx,
                              c->d_probs, c->d_cand_i, c->d_cand_n, c->d_zpart, c->d_spart, d_tokens, nullptr, s));
    HIPC(hipStreamSynchronize(s));   // seed / step are host temporaries
    return OPUS_OK;
}

// ------------------------------------------------------------------------------------------------ diagnostics
extern "C" int opus_debug_gemm(opus_ctx *c, const void *A, const void *W, const float *bias, const float *residual,
                               void *Cp, int32_t M, int32_t N, int32_t K, int32_t epi, int32_t out_f32, void *stream) {
    if (!c || !A || !W || !Cp) return fail(OPUS_EBADARG, "debug_gemm: null pointer");
    if (M < 1 || N < 1 || K < 64 || K % 64) return fail(OPUS_ESHAPE, "debug_gemm: K must be a positive multiple of 64");
    if (epi < 0 || epi > 2 || (epi == 2 && N % 32)) return fail(OPUS_EBADARG, "debug_gemm: epilogue");
    HIPC(hipSetDevice(c->device));
    const int nout = epi == EPI_SILU_GU16 ? N / 2 : N;
    c->rq_a_tiled = g_knobs.debug_a_tiled;     // (tests: A handed over in fragment order, ceil(M / 16) * 16 rows)
    return gemm(c, (hipStream_t)stream, (const half_t *)A, K, (const half_t *)W, M, N, K, bias, epi, residual, Cp, nout,
                out_f32);
}

extern "C" int opus_debug_gemm_norm(opus_ctx *c, const float *A, const void *W, void *Cp, int32_t M, int32_t N, int32_t K,
                                    int32_t epi, int32_t out_f32, float eps, void *stream) {
    if (!c || !A || !W || !Cp) return fail(OPUS_EBADARG, "debug_gemm_norm: null pointer");
    if (M < 1 || M > MID_MAX_M || N < 1 || K < 64 || K % 64) return fail(OPUS_ESHAPE, "debug_gemm_norm: M <= 64, K %% 64 == 0");
    if (epi != 0 && epi != 2) return fail(OPUS_EBADARG, "debug_gemm_norm: epilogue 0 or 2");
    HIPC(hipSetDevice(c->device));
    const int nout = epi == EPI_SILU_GU16 ? N / 2 : N;
    return gemm_any(c, (hipStream_t)stream, nullptr, A, eps, K, (const half_t *)W, M, N, K, nullptr, epi, nullptr, Cp, nout,
                    out_f32);
}

// The ESM QKV projection + rotary exactly as opus_esm2_encode issues it: out[M, 3 D] fp16 = rotary(A W^T + bias) with the
// query third scaled by head_dim^-0.5, positions row % T.  allow_fuse = 0 forces the stand-alone rotary kernel on the stored
// projection; *fused (HOST) = 1 when the GEMM's epilogue did the rotation.
extern "C" int opus_debug_gemm_rope(opus_ctx *c, const void *A, const void *W, const float *bias, void *out, int32_t M,
                                    int32_t D, int32_t K, int32_t T, int32_t heads, int32_t allow_fuse, int32_t *fused,
                                    void *stream) {
    if (!c || !A || !W || !out || !fused) return fail(OPUS_EBADARG, "debug_gemm_rope: null pointer");
    if (M < 1 || D < 16 || K < 64 || K % 64 || T < 1 || heads < 1 || D % heads || M % T)
        return fail(OPUS_ESHAPE, "debug_gemm_rope: M %% T == 0, D %% heads == 0, K %% 64 == 0");
    const int hd = D / heads;
    if (hd != c->cfg.enc_dim / c->cfg.enc_heads || T > c->cfg.max_enc_tokens)
        return fail(OPUS_ESHAPE, "debug_gemm_rope: head_dim / T must match the context's encoder rotary table");
    HIPC(hipSetDevice(c->device));
    hipStream_t s = (hipStream_t)stream;
    const float qs = 1.0f / sqrtf((float)hd);
    if (allow_fuse && hd == 64) {
        c->rq_rope_cs = c->cs_enc; c->rq_rope_T = T; c->rq_rope_cols = 2 * D; c->rq_rope_qcols = D; c->rq_rope_qscale = qs;
    }
    OPC(gemm(c, s, (const half_t *)A, K, (const half_t *)W, M, 3 * D, K, bias, EPI_NONE, nullptr, out, 3 * D, 0));
    *fused = c->rq_rope_done;
    if (!c->rq_rope_done) HIPC(launch_esm_rope((half_t *)out, c->cs_enc, M / T, T, heads, hd, qs, s));
    return OPUS_OK;
}

// The producer / consumer pair of the row-scale RMSNorm fusion exactly as prefill() and decode_step() issue it:
//   X <- X + A W1^T            (wo: a split-K GEMM whose reduce also writes fp16(X) and per-256-column sums of squares)
//   C  = epi(rmsnorm(X) W2^T)  (gate/up or lm_head: gemm_wide_kernel scales its rows by the rstd from those sums)
// *fused = 1 when the fused form ran (0: the separate-norm fallback ran; the result is the same function either way).
extern "C" int opus_debug_gemm_rowscale(opus_ctx *c, const void *A, const void *W1, float *X, const void *W2, void *Cp, int32_t M,
                                        int32_t N1, int32_t K1, int32_t N2, int32_t epi, float eps, int32_t *fused, void *stream) {
    if (!c || !A || !W1 || !X || !W2 || !Cp || !fused) return fail(OPUS_EBADARG, "debug_gemm_rowscale: null pointer");
    if (M < 1 || N1 < 64 || N1 % 64 || K1 < 64 || K1 % 64 || N2 < 32) return fail(OPUS_ESHAPE, "debug_gemm_rowscale: shape");
    if (epi != EPI_NONE && epi != EPI_SILU_GU16) return fail(OPUS_EBADARG, "debug_gemm_rowscale: epilogue 0 or 2");
    const opus_config &g = c->cfg;
    if ((int64_t)M * N1 > (int64_t)g.max_batch * g.max_prompt * g.dec_dim) return fail(OPUS_ESHAPE, "debug_gemm_rowscale: M * N1 exceeds the scratch");
    HIPC(hipSetDevice(c->device));
    hipStream_t s = (hipStream_t)stream;
    if (fuse_rows() && M <= 96) c->rq_xh = c->d_xn;
    OPC(gemm(c, s, (const half_t *)A, K1, (const half_t *)W1, M, N1, K1, nullptr, EPI_NONE, X, X, N1, 1));
    c->xh_src = c->rq_done ? X : nullptr;
    *fused = c->xh_src != nullptr && (N1 & 255) == 0 && gemm_goes_wide(M, N2) ? 1 : 0;
    const int nout = epi == EPI_SILU_GU16 ? N2 / 2 : N2;
    return gemm_norm(c, s, X, eps, c->d_xn, (const half_t *)W2, M, N2, N1, epi, Cp, nout, 0);
}

// The QKV projection of the batched decode step exactly as decode_step() issues it (narrow output routed through the
// k-part kernels, raw slabs left for the attention kernel): d_slabs fp32 [*ks][M][N] receives the slabs, *ks (HOST) their
// number; *ks = 1 means the launch wrote a finished fp16 output instead (nothing is copied).
extern "C" int opus_debug_gemm_slabs(opus_ctx *c, const void *A, const void *W, float *d_slabs, int32_t M, int32_t N, int32_t K,
                                     int32_t *ks, void *stream) {
    if (!c || !A || !W || !ks) return fail(OPUS_EBADARG, "debug_gemm_slabs: null pointer");   // (d_slabs may be NULL: timing runs)
    if (M < 1 || N < 16 || K < 64 || K % 64) return fail(OPUS_ESHAPE, "debug_gemm_slabs: shape");
    const opus_config &g = c->cfg;
    const int64_t QKVd = (int64_t)(g.dec_heads + 2 * g.dec_kv_heads) * g.dec_head_dim;
    if ((int64_t)M * N > (int64_t)g.max_batch * g.max_prompt * QKVd) return fail(OPUS_ESHAPE, "debug_gemm_slabs: M * N exceeds the scratch");
    HIPC(hipSetDevice(c->device));
    hipStream_t s = (hipStream_t)stream;
    c->rq_force_wide = 1;
    c->rq_slab_only = 1;
    c->rq_a_tiled = g_knobs.debug_a_tiled;
    OPC(gemm(c, s, (const half_t *)A, K, (const half_t *)W, M, N, K, nullptr, EPI_NONE, nullptr, c->d_qkv, N, 0));
    *ks = c->rq_ks;
    if (c->rq_ks > 1 && d_slabs)
        HIPC(hipMemcpyAsync(d_slabs, c->gemm_ws, (size_t)c->rq_ks * M * N * sizeof(float), hipMemcpyDeviceToDevice, s));
    return OPUS_OK;
}

// ------------------------------------------------------------------------------------------------ beam search support
// The best M continuations of every batch row over its K beams' last logits (this context's most recent prefill / decode step
// on B K rows, row = b K + k): transformers generation/utils.py _beam_search step b-c - log_softmax in fp32, + running beam
// scores, torch.topk over the flattened [K V] - scores fp32 [B, M] descending and flat indices k V + token int32 [B, M].
extern "C" int opus_beam_topk(opus_ctx *c, const float *d_run_scores, int32_t B, int32_t K, int32_t M, float *d_scores,
                              int32_t *d_idx, void *stream) {
    if (!c || !d_run_scores || !d_scores || !d_idx) return fail(OPUS_EBADARG, "beam_topk: null pointer");
    if (!c->prefilled) return fail(OPUS_ESTATE, "beam_topk before prefill");
    if (B < 1 || K < 1 || B * K != c->cur_B) return fail(OPUS_ESHAPE, "beam_topk: B=%d x K=%d rows, the last step had %d", B, K, c->cur_B);
    if (M < 1 || M > 16) return fail(OPUS_ESHAPE, "beam_topk: 1 <= M <= 16 candidates per row (got %d)", M);
    HIPC(hipSetDevice(c->device));
    hipStream_t s = (hipStream_t)stream;
    c->phase = PH_DECODE;
    KL(KC_OTHER, 12.0 * c->cur_B * c->cfg.dec_vocab,
       launch_beam_topk(c->d_logits, d_run_scores, B, K, c->cfg.dec_vocab, M, c->d_zpart, d_scores, d_idx, s));
    return OPUS_OK;
}

// Beam-sample step (do_sample with num_beams > 1; transformers _get_top_k_continuations, do_sample branch): the M
// continuations of every batch row drawn WITHOUT replacement from softmax over the K V accumulated log-probabilities, after
// log_softmax and the warpers (temperature, top_k of opus_set_sampling_top_k, top_p) on each of the K rows - scores fp32
// [B, M] (the accumulated log-probabilities of the drawn continuations) and flat indices k V + token int32 [B, M] in the order
// drawn; entries beyond the continuations of non-zero probability are -inf / 0x7fffffff.  d_logits NULL: the last step's.
extern "C" int opus_beam_sample_topk(opus_ctx *c, const float *d_logits, const float *d_run_scores, int32_t B, int32_t K, int32_t M,
                                     float temperature, float top_p, uint64_t seed, int32_t step, float *d_scores, int32_t *d_idx,
                                     void *stream) {
    if (!c || !d_run_scores || !d_scores || !d_idx) return fail(OPUS_EBADARG, "beam_sample_topk: null pointer");
    if (!d_logits && !c->prefilled) return fail(OPUS_ESTATE, "beam_sample_topk before prefill");
    if (B < 1 || K < 1 || B * K > c->cfg.max_batch || (!d_logits && B * K != c->cur_B))
        return fail(OPUS_ESHAPE, "beam_sample_topk: B=%d x K=%d rows (the last step had %d, the context holds %d)", B, K, c->cur_B, c->cfg.max_batch);
    if (M < 1 || M > 16) return fail(OPUS_ESHAPE, "beam_sample_topk: 1 <= M <= 16 candidates per row (got %d)", M);
    if (!(temperature > 0.f) || top_p <= 0.f || top_p > 1.f) return fail(OPUS_EBADARG, "beam_sample_topk: temperature > 0, 0 < top_p <= 1");
    HIPC(hipSetDevice(c->device));
    hipStream_t s = (hipStream_t)stream;
    const float *lg = d_logits ? d_logits : c->d_logits;
    const int V = c->cfg.dec_vocab, R = B * K;
    c->phase = PH_DECODE;
    // min_tokens_to_keep of both warpers under beam-sample: #eos + 1, at least 2 (generation/utils.py _get_logits_processor,
    // "keep at least one non-eos token") = M / K, since M = max(2, 1 + #eos) K candidates are drawn per batch row
    const int min_keep = M / K > 1 ? M / K : 1;
    const int top_k = c->samp_top_k > 0 && c->samp_top_k < min_keep ? min_keep : c->samp_top_k;    // TopKLogitsWarper: max(top_k, min_tokens_to_keep)
    KL(KC_OTHER, 4.0 * 4 * R * V,
       launch_sample_select(lg, R, V, temperature, top_p, top_k, nullptr, nullptr, c->d_pval, c->d_pidx, c->d_probs, c->d_cand_i,
                            c->d_cand_n, c->d_zpart, c->d_spart, nullptr, c->d_bthr, s));
    // (the candidate lists of the nucleus search are consumed: their buffers hold the rows' min_keep best logits next)
    KL(KC_OTHER, 12.0 * R * V,
       launch_beam_sample(lg, d_run_scores, B, K, V, M, temperature, c->d_pval, c->d_bthr, min_keep, c->d_probs, c->d_cand_i, seed, step,
                          c->d_blse, d_scores, d_idx, s));
    return OPUS_OK;
}

// TopKLogitsWarper of the sampling paths of this context (opus_generate_sample, opus_debug_sample, opus_beam_sample_topk):
// k > 0 keeps the k most probable tokens (and ties with the k-th) before the nucleus; 0 = off.  transformers 4.46.3 - the
// reference's pin - defaults GenerationConfig.top_k to 50 whenever it samples; the Python mirror sets that default.
extern "C" int opus_set_sampling_top_k(opus_ctx *c, int32_t k) {
    if (!c || k < 0) return fail(OPUS_EBADARG, "set_sampling_top_k: k >= 0");
    if (k != c->samp_top_k) drop_graphs(c);                                      // the captured step holds k
    c->samp_top_k = k;
    return OPUS_OK;
}

// KV cache rows r <- rows src[r] for every layer (Cache.reorder_cache(beam_idx) of the reference's stack: the beams that
// survive a step continue from their parents' caches).  R = the rows of the last prefill.  Two passes per layer through a
// one-layer scratch (a permutation cannot be applied in place row by row).
extern "C" int opus_kv_reorder(opus_ctx *c, const int32_t *d_src_rows, int32_t R, void *stream) {
    if (!c || !d_src_rows) return fail(OPUS_EBADARG, "kv_reorder: null pointer");
    if (!c->prefilled || R != c->cur_B) return fail(OPUS_ESHAPE, "kv_reorder: R=%d but the last prefill had %d rows", R, c->prefilled ? c->cur_B : 0);
    HIPC(hipSetDevice(c->device));
    hipStream_t s = (hipStream_t)stream;
    const opus_config &g = c->cfg;
    const size_t row = (size_t)c->cache_sb, layer = row * g.max_batch;
    if (!c->kv_tmp) HIPC(hipMalloc((void **)&c->kv_tmp, 2 * layer * sizeof(half_t)));
    c->phase = PH_DECODE;
    // only the filled slots (0 .. T + *step - 1, read from the device step word) of the rows that change place move
    const int nkv = g.dec_kv_heads, hd = g.dec_head_dim, T0 = c->cur_T;
    for (int l = 0; l < g.dec_layers; ++l) {
        half_t *kc = c->kc + l * c->cache_sl, *vc = c->vc + l * c->cache_sl;
        KL(KC_OTHER, 8.0 * R * nkv * hd * T0, launch_kv_gather_rows(kc, c->kv_tmp, d_src_rows, R, (int64_t)row, nkv, c->cache_sh, c->d_step, T0, hd, 0, s));
        HIPC(launch_kv_gather_rows(vc, c->kv_tmp + layer, d_src_rows, R, (int64_t)row, nkv, c->cache_sh, c->d_step, T0, hd, 0, s));
        HIPC(launch_kv_gather_rows(kc, c->kv_tmp, d_src_rows, R, (int64_t)row, nkv, c->cache_sh, c->d_step, T0, hd, 1, s));
        HIPC(launch_kv_gather_rows(vc, c->kv_tmp + layer, d_src_rows, R, (int64_t)row, nkv, c->cache_sh, c->d_step, T0, hd, 1, s));
    }
    return OPUS_OK;
}

// fp32 logits [B, dec_vocab] of the most recent prefill / decode step of this context (the optional logits gather of
// SURVEY 8e; also what a caller needs to apply its own logits processors).
extern "C" int opus_last_logits(opus_ctx *c, float *d_out, int32_t B, void *stream) {
    if (!c || !d_out) return fail(OPUS_EBADARG, "last_logits: null pointer");
    if (!c->prefilled) return fail(OPUS_ESTATE, "last_logits before prefill");
    if (B < 1 || B > c->cur_B) return fail(OPUS_ESHAPE, "last_logits: B=%d but the last prefill had %d rows", B, c->cur_B);
    HIPC(hipSetDevice(c->device));
    HIPC(hipMemcpyAsync(d_out, c->d_logits, (size_t)B * c->cfg.dec_vocab * sizeof(float), hipMemcpyDeviceToDevice, (hipStream_t)stream));
    return OPUS_OK;
}

extern "C" int opus_debug_attention(opus_ctx *c, const void *Q, const void *K, const void *V, void *O,
                                    const int32_t *kstart, const int32_t *kend, int32_t B, int32_t T, int32_t heads,
                                    int32_t group, int32_t hd, int32_t causal, float scale, void *stream) {
    if (!c || !Q || !K || !V || !O) return fail(OPUS_EBADARG, "debug_attention: null pointer");
    if (B < 1 || T < 1 || heads < 1 || group < 1 || heads % group) return fail(OPUS_ESHAPE, "debug_attention: shape");
    HIPC(hipSetDevice(c->device));
    hipStream_t s = (hipStream_t)stream;
    AttnParams a;
    const int kvh = heads / group;
    a.Q = (const half_t *)Q; a.K = (const half_t *)K; a.V = (const half_t *)V; a.O = (half_t *)O;
    a.q_st = (int64_t)heads * hd; a.q_sb = a.q_st * T;
    a.k_st = a.v_st = (int64_t)kvh * hd; a.k_sb = a.v_sb = a.k_st * T;
    a.o_st = a.q_st; a.o_sb = a.q_sb;
    a.kstart = kstart; a.kend = kend; a.B = B; a.T = T; a.heads = heads; a.group = group; a.head_dim = hd;
    a.causal = causal; a.scale = scale;
    KLF(KC_ATTN_PREFILL, 2.0 * B * T * hd * (2.0 * heads + 2.0 * kvh), (causal ? 2.0 : 4.0) * B * (double)T * T * heads * hd,
        launch_attn_prefill(a, s));
    return OPUS_OK;
}

// attn_decode_kernel exactly as decode_step() launches it (finished fp16 projections in, no k-part slabs), on layer 0 of this
// context's KV cache: the history d_k_hist / d_v_hist fp16 [B, kv heads, L, hd] (keys already rotated, as the cache holds them)
// is copied into slots 0 .. L-1 (L = T0 + step), the step word, kstart[] and the rotary rows of the step are set as the
// embedding kernel sets them, then ONE launch: rotary(q, k) at position L - kstart[b], append at slot L, softmax over the
// slots kstart[b] .. L.  d_out fp16 [B, heads hd]; d_k_new / d_v_new (optional) fp16 [B, kv heads, hd] = what slot L holds
// afterwards.  The kernel-level parity test of the key-tile loop (tests/test_gpu_longctx.py): rows D3 / D4.
extern "C" int opus_debug_attn_decode(opus_ctx *c, const void *d_qkv, const void *d_k_hist, const void *d_v_hist,
                                      const int32_t *d_kstart, int32_t B, int32_t T0, int32_t step, void *d_out, void *d_k_new,
                                      void *d_v_new, void *stream) {
    if (!c || !d_qkv || !d_kstart || !d_out) return fail(OPUS_EBADARG, "debug_attn_decode: null pointer");
    const opus_config &g = c->cfg;
    const int L = T0 + step, nkv = g.dec_kv_heads, hd = g.dec_head_dim;
    if (B < 1 || B > g.max_batch || T0 < 1 || T0 > g.max_prompt || step < 0 || step >= g.max_new_tokens)
        return fail(OPUS_ESHAPE, "debug_attn_decode: B=%d T0=%d step=%d exceed the context (%d, %d, %d)", B, T0, step, g.max_batch,
                    g.max_prompt, g.max_new_tokens);
    if (L > 0 && (!d_k_hist || !d_v_hist)) return fail(OPUS_EBADARG, "debug_attn_decode: history is null");
    HIPC(hipSetDevice(c->device));
    hipStream_t s = (hipStream_t)stream;
    const size_t pitch = (size_t)c->cache_sh * sizeof(half_t), roww = (size_t)L * hd * sizeof(half_t);
    HIPC(hipMemcpy2DAsync(c->kc, pitch, d_k_hist, roww, roww, (size_t)B * nkv, hipMemcpyDeviceToDevice, s));
    HIPC(hipMemcpy2DAsync(c->vc, pitch, d_v_hist, roww, roww, (size_t)B * nkv, hipMemcpyDeviceToDevice, s));
    HIPC(hipMemcpyAsync(c->d_kstart, d_kstart, (size_t)B * sizeof(int32_t), hipMemcpyDeviceToDevice, s));
    OPC(reset_step(c, s, T0, step));
    // (the embedding kernel with a zero-width row: only its rotary-row part runs)
    HIPC(launch_embed_tokens(c->d_next, nullptr, B, 0, 1, c->d_xl, nullptr, nullptr, 0, c->cs_dec, c->d_kstart, c->d_step, -1, hd / 2,
                             c->cs_row, nullptr, 0, s));
    AttnDecodeParams a;
    a.qkv = (const half_t *)d_qkv; a.slabs = nullptr; a.ks = 0; a.slab_stride = 0; a.row_ssq = nullptr; a.row_nblk = 0; a.eps = 0.f; a.K = 0;
    a.bias = nullptr;
    a.cs_row = c->cs_row; a.kstart = c->d_kstart; a.step = c->d_step; a.T0 = -1; a.nh = g.dec_heads; a.nkv = nkv;
    a.kc = c->kc; a.vc = c->vc; a.cache_sb = c->cache_sb; a.cache_sh = c->cache_sh;
    a.ctx_cap = g.max_prompt + g.max_new_tokens; a.scale = 1.0f / sqrtf((float)hd); a.out = (half_t *)d_out; a.out_tiled = 0;
    c->phase = PH_DECODE;
    KL(KC_ATTN_DECODE, 4.0 * B * nkv * hd * (L + 1), launch_attn_decode(a, B, hd, s));
    const size_t one = (size_t)hd * sizeof(half_t);
    if (d_k_new) HIPC(hipMemcpy2DAsync(d_k_new, one, c->kc + (size_t)L * hd, pitch, one, (size_t)B * nkv, hipMemcpyDeviceToDevice, s));
    if (d_v_new) HIPC(hipMemcpy2DAsync(d_v_new, one, c->vc + (size_t)L * hd, pitch, one, (size_t)B * nkv, hipMemcpyDeviceToDevice, s));
    c->prefilled = false;                                     // (the cache no longer belongs to a prefill)
    return OPUS_OK;
}

// Run-time tuning knobs (A/B aids of the benchmarks and tests; process-wide): "no_stream", "pp_gm", "misc0" .. "misc7".
extern "C" int opus_debug_knob(opus_ctx *c, const char *name, int32_t value) {
    if (!name) return fail(OPUS_EBADARG, "debug_knob: null name");
    if (c) drop_graphs(c);   // a captured decode graph replays the kernels it was recorded with
    if (!strcmp(name, "no_stream")) g_knobs.no_stream = value;
    else if (!strcmp(name, "debug_a_tiled")) g_knobs.debug_a_tiled = value;
    else if (!strcmp(name, "no_ln_fusion")) g_knobs.no_ln_fusion = value;
    else if (!strcmp(name, "enc_full_last_layer")) g_knobs.enc_full_last_layer = value;
    else if (!strcmp(name, "poison_handoff")) {   // test aid: what an aborted launch leaves behind - every ticket drawn once, no flag set
        if (!c) return fail(OPUS_EBADARG, "poison_handoff needs a context");
        std::vector<int32_t> h(HANDOFF_ERR, 0);
        // (a ticket count alone cannot tell "one stale ticket + three arrivals" from four arrivals: detection is certain only for
        //  values no clean launch can reach - value > 1 writes that into the tickets -; the cure is the per-call zeroing)
        if (value) for (int i = 0; i < HANDOFF_ERR; ++i) h[i] = i < 512 ? value : ((i & 1) ? 0 : 1);
        HIPC(hipSetDevice(c->device));
        HIPC(hipDeviceSynchronize());
        HIPC(hipMemcpy(c->d_cnt, h.data(), h.size() * sizeof(int32_t), hipMemcpyHostToDevice));
    }
    else if (!strcmp(name, "pp_gm")) { if (value < 1 || value > 64) return fail(OPUS_EBADARG, "pp_gm out of range"); g_knobs.pp_gm = value; }
    else if (!strncmp(name, "misc", 4) && name[4] >= '0' && name[4] <= '7' && !name[5]) g_knobs.misc[name[4] - '0'] = value;
    else return fail(OPUS_EBADARG, "debug_knob: unknown knob '%s'", name);
    return OPUS_OK;
}

// Counters of this context (no reference counterpart): "graph_instantiations" = decode-step hipGraphs instantiated since the
// context was created (one per distinct batch size / token budget / sampling setting, none per prompt length);
// "graph_replays" = decode steps launched from a graph; "graphs_cached"; "decode_steps" = decode steps opus_generate_* enqueued
// (with an EOS id or a stop sequence: at most 2 past the step at which the last row finished).  Unknown name / null: -1.
extern "C" int64_t opus_stat(opus_ctx *c, const char *name) {
    if (!c || !name) return -1;
    if (!strcmp(name, "graph_instantiations")) return c->graph_instantiations;
    if (!strcmp(name, "graph_replays")) return c->graph_replays;
    if (!strcmp(name, "graphs_cached")) return (int64_t)c->graphs.size();
    if (!strcmp(name, "decode_steps")) return c->decode_steps;
    return -1;
}

// ------------------------------------------------------------------------------------------------ timing
extern "C" int opus_timing_enable(opus_ctx *c, int32_t on) {
    if (!c) return fail(OPUS_EBADARG, "ctx is null");
    c->timing = on != 0;
    return OPUS_OK;
}
extern "C" int opus_timing_reset(opus_ctx *c) {
    if (!c) return fail(OPUS_EBADARG, "ctx is null");
    HIPC(hipDeviceSynchronize());
    timing_clear(c);
    return OPUS_OK;
}
static const char *kclass_names[KC_COUNT] = {"gemm_skinny", "gemm_mid", "gemm_wide", "gemm_ring", "gemm_pp", "gemm_tile", "splitk_reduce",
                                            "attn_prefill", "attn_decode", "norm", "other", "gemm_stream"};
static const char *phase_names[PH_COUNT] = {"encode", "project", "splice", "prefill", "decode", "other"};

extern "C" int opus_timing_get(opus_ctx *c, const char *kernel_class, const char *phase, double *ms, int64_t *launches,
                               double *bytes, double *flops) {
    if (!c || !kernel_class || !phase || !ms || !launches || !bytes || !flops) return fail(OPUS_EBADARG, "null argument");
    int k = -1, ph = -1;
    const bool any_k = !strcmp(kernel_class, "*"), any_p = !strcmp(phase, "*");
    for (int i = 0; i < KC_COUNT; ++i) if (!strcmp(kclass_names[i], kernel_class)) k = i;
    for (int i = 0; i < PH_COUNT; ++i) if (!strcmp(phase_names[i], phase)) ph = i;
    if (k < 0 && !any_k) return fail(OPUS_EBADARG, "unknown kernel class '%s'", kernel_class);
    if (ph < 0 && !any_p) return fail(OPUS_EBADARG, "unknown phase '%s'", phase);
    HIPC(hipDeviceSynchronize());
    double t = 0, b = 0, f = 0;
    int64_t n = 0;
    for (auto &r : c->recs) {
        if ((!any_k && r.klass != k) || (!any_p && r.phase != ph)) continue;
        float e = 0;
        HIPC(hipEventElapsedTime(&e, r.e0, r.e1));
        t += e;
        b += r.bytes;
        f += r.flops;
        ++n;
    }
    *ms = t;
    *launches = n;
    *bytes = b;
    *flops = f;
    return OPUS_OK;
}

extern "C" int opus_timing_names(char *buf, int32_t cap) {
    if (!buf || cap < 1) return fail(OPUS_EBADARG, "null buffer");
    std::string out;
    for (int i = 0; i < KC_COUNT; ++i) { out += kclass_names[i]; out += i + 1 < KC_COUNT ? "," : ";"; }
    for (int i = 0; i < PH_COUNT; ++i) { out += phase_names[i]; if (i + 1 < PH_COUNT) out += ","; }
    if ((int)out.size() + 1 > cap) return fail(OPUS_ESHAPE, "buffer too small (%zu needed)", out.size() + 1);
    memcpy(buf, out.c_str(), out.size() + 1);
    return OPUS_OK;
}
