// Beam search support (row N1 of SURVEY 8f: the `num_beams` pass-through of eval/run_opus_ddp.py:129,158 -> transformers
// GenerationMixin._beam_search).  The decoder runs B x K rows (row = b K + k) through the ordinary prefill / decode step; per
// step the device does the two things that touch O(K V) or the KV cache,
//   beam_lse / beam_topk   log_softmax of the K beams' logits in fp32, + the beams' running scores, and the best M of the K V
//                          continuations of each batch row (generation/utils.py _get_top_k_continuations: torch.topk over
//                          the flattened [K V] accumulated log-probabilities; M = max(2, 1 + #eos) K),
//   kv_gather_rows         the cache rows of the surviving beams (Cache.reorder_cache / index_select(0, beam_idx)),
// and the host (opus-pllm_amd/beam.py) keeps the O(K) bookkeeping of the finished / running beams, as the reference's Python does.
#include "common.h"

namespace opus {

// lse[row] = log sum_v exp(logits[row][v]) (max-shifted, fp32), one workgroup per decoder row
__global__ __launch_bounds__(256) void beam_lse_kernel(const float *__restrict__ logits, int V, float *__restrict__ lse) {
    __shared__ float red[256];
    const float *row = logits + (int64_t)blockIdx.x * V;
    float m = -INFINITY;
    for (int i = threadIdx.x; i < V; i += 256) m = fmaxf(m, row[i]);
    red[threadIdx.x] = m;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if ((int)threadIdx.x < o) red[threadIdx.x] = fmaxf(red[threadIdx.x], red[threadIdx.x + o]);
        __syncthreads();
    }
    m = red[0];
    __syncthreads();
    float s = 0.f;
    for (int i = threadIdx.x; i < V; i += 256) s += expf(row[i] - m);
    red[threadIdx.x] = s;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {            // fixed tree: bitwise reproducible
        if ((int)threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
        __syncthreads();
    }
    if (threadIdx.x == 0) lse[blockIdx.x] = m + logf(red[0]);
}

constexpr int BEAM_MAXM = 16;     // candidates kept per batch row and step

// The best M of the K V continuations of batch row b: score(k, v) = (logits[b K + k][v] - lse[b K + k]) + run[b K + k], sorted
// by score descending, ties by the lower flat index k V + v.  One workgroup per batch row: every thread keeps the best M of its
// strided share in registers, then M rounds of a block-wide arg-max over the heads of the threads' sorted lists.
__global__ __launch_bounds__(256) void beam_topk_kernel(const float *__restrict__ logits, const float *__restrict__ lse,
                                                        const float *__restrict__ run, int K, int V, int M,
                                                        float *__restrict__ out_s, int32_t *__restrict__ out_i) {
    __shared__ float s_v[256];
    __shared__ int s_i[256];
    __shared__ int s_who[256];
    const int b = blockIdx.x, tid = threadIdx.x;
    float bs[BEAM_MAXM];
    int bi[BEAM_MAXM];
#pragma unroll
    for (int j = 0; j < BEAM_MAXM; ++j) { bs[j] = -INFINITY; bi[j] = 0x7fffffff; }
    for (int k = 0; k < K; ++k) {
        const float *row = logits + (int64_t)(b * K + k) * V;
        const float off = lse ? lse[b * K + k] : 0.f, add = run ? run[b * K + k] : 0.f;   // (both null: the plain top-M of a logits row)
        for (int v = tid; v < V; v += 256) {
            const float sc = (row[v] - off) + add;
            const int id = k * V + v;
            if (sc > bs[BEAM_MAXM - 1] || (sc == bs[BEAM_MAXM - 1] && id < bi[BEAM_MAXM - 1])) {
                // insertion into the sorted list (static indices: the list stays in registers)
                float cs = sc;
                int ci = id;
#pragma unroll
                for (int j = 0; j < BEAM_MAXM; ++j) {
                    const bool better = cs > bs[j] || (cs == bs[j] && ci < bi[j]);
                    const float ts = better ? bs[j] : cs;
                    const int ti = better ? bi[j] : ci;
                    bs[j] = better ? cs : bs[j];
                    bi[j] = better ? ci : bi[j];
                    cs = ts;
                    ci = ti;
                }
            }
        }
    }
    int head = 0;                                    // this thread's next unused entry
    for (int r = 0; r < M; ++r) {
        float hv = -INFINITY;
        int hi = 0x7fffffff;
#pragma unroll
        for (int j = 0; j < BEAM_MAXM; ++j)
            if (j == head) { hv = bs[j]; hi = bi[j]; }
        s_v[tid] = hv; s_i[tid] = hi; s_who[tid] = tid;
        __syncthreads();
        for (int o = 128; o > 0; o >>= 1) {
            if (tid < o) {
                const float v = s_v[tid + o];
                const int i = s_i[tid + o];
                if (v > s_v[tid] || (v == s_v[tid] && i < s_i[tid])) { s_v[tid] = v; s_i[tid] = i; s_who[tid] = s_who[tid + o]; }
            }
            __syncthreads();
        }
        if (tid == 0) { out_s[b * M + r] = s_v[0]; out_i[b * M + r] = s_i[0]; }
        if (s_who[0] == tid) ++head;
        __syncthreads();
    }
}

// Beam-sample (do_sample with num_beams > 1): _get_top_k_continuations draws the M continuations of a batch row with
// torch.multinomial(softmax(accumulated), M) - without replacement - from the K V accumulated log-probabilities
//   a(k, v) = (logits[b K + k][v] - lse[b K + k]) / T + run[b K + k]     for the tokens the warpers keep, -inf for the others
// (log_softmax first, then TemperatureLogitsWarper / TopKLogitsWarper / TopPLogitsWarper on the log-probabilities: the kept set
// of a row is the sampling head's, p = exp(l / T - max / T) > thr[row]).  Drawing without replacement from softmax(a) is taking
// the M largest of a + g with g i.i.d. standard Gumbel (the order of the keys is the order of the draws): the same per-thread
// sorted lists and block arg-max rounds as beam_topk_kernel, on the keys.  g comes from a counter-based generator keyed by
// (seed, step, row, token): a given seed reproduces its draws; parity with the reference is distributional.
// out_s = a of the drawn continuations (NOT the keys), out_i = k V + v, in the order drawn.
__global__ __launch_bounds__(256) void beam_sample_kernel(const float *__restrict__ logits, const float *__restrict__ lse,
                                                          const float *__restrict__ run, const float *__restrict__ pmax,
                                                          const float *__restrict__ thr, const float *__restrict__ kth, int min_keep,
                                                          float inv_temp, uint64_t seed, int step,
                                                          int K, int V, int M, float *__restrict__ out_s, int32_t *__restrict__ out_i) {
    __shared__ float s_v[256];
    __shared__ int s_i[256];
    __shared__ int s_who[256];
    __shared__ float s_a[256];
    const int b = blockIdx.x, tid = threadIdx.x;
    float bk[BEAM_MAXM], ba[BEAM_MAXM];
    int bi[BEAM_MAXM];
#pragma unroll
    for (int j = 0; j < BEAM_MAXM; ++j) { bk[j] = -INFINITY; ba[j] = -INFINITY; bi[j] = 0x7fffffff; }
    for (int k = 0; k < K; ++k) {
        const int r = b * K + k;
        const float *row = logits + (int64_t)r * V;
        float gmax = pmax[r * APART];
        for (int q = 1; q < APART; ++q) gmax = fmaxf(gmax, pmax[r * APART + q]);
        gmax *= inv_temp;                                            // (the expression of sample_stage1_kernel: the same p)
        const float off = lse[r], add = run[r], th = thr[r];
        // min_tokens_to_keep of the warpers under beam-sample (#eos + 1, at least 2: GenerationMixin._get_logits_processor): the
        // min_keep best tokens of a row stay whatever the nucleus says - kth[r][min_keep - 1] is the row's min_keep-th largest
        // logit (ties with it stay too; the sorted-order cut of TopPLogitsWarper keeps exactly min_keep: a measure-zero difference)
        const float lm = kth ? kth[r * min_keep + min_keep - 1] : INFINITY;
        const uint64_t h0 = splitmix64(seed ^ (0x9E3779B97F4A7C15ull * (uint64_t)(r + 1)) ^ ((uint64_t)(step + 1) << 32));
        for (int v = tid; v < V; v += 256) {
            const float l = row[v];
            const float p = __expf(l * inv_temp - gmax);
            if (!(p > th) && !(l >= lm)) continue;
            const float a = (l - off) * inv_temp + add;
            const uint64_t h = splitmix64(h0 + 0xD1B54A32D192ED03ull * (uint64_t)(v + 1));
            const float u = ((float)(h >> 40) + 0.5f) * (1.0f / 16777216.0f);      // (0, 1)
            const float key = a - __logf(-__logf(u));
            const int id = k * V + v;
            if (key > bk[BEAM_MAXM - 1] || (key == bk[BEAM_MAXM - 1] && id < bi[BEAM_MAXM - 1])) {
                float ck = key, ca = a;
                int ci = id;
#pragma unroll
                for (int j = 0; j < BEAM_MAXM; ++j) {
                    const bool better = ck > bk[j] || (ck == bk[j] && ci < bi[j]);
                    const float tk = better ? bk[j] : ck, ta = better ? ba[j] : ca;
                    const int ti = better ? bi[j] : ci;
                    bk[j] = better ? ck : bk[j];
                    ba[j] = better ? ca : ba[j];
                    bi[j] = better ? ci : bi[j];
                    ck = tk; ca = ta; ci = ti;
                }
            }
        }
    }
    int head = 0;
    for (int r = 0; r < M; ++r) {
        float hv = -INFINITY, ha = -INFINITY;
        int hi = 0x7fffffff;
#pragma unroll
        for (int j = 0; j < BEAM_MAXM; ++j)
            if (j == head) { hv = bk[j]; ha = ba[j]; hi = bi[j]; }
        s_v[tid] = hv; s_i[tid] = hi; s_who[tid] = tid; s_a[tid] = ha;
        __syncthreads();
        for (int o = 128; o > 0; o >>= 1) {
            if (tid < o) {
                const float v = s_v[tid + o];
                const int i = s_i[tid + o];
                if (v > s_v[tid] || (v == s_v[tid] && i < s_i[tid])) { s_v[tid] = v; s_i[tid] = i; s_who[tid] = s_who[tid + o]; s_a[tid] = s_a[tid + o]; }
            }
            __syncthreads();
        }
        // (fewer than M kept continuations with a non-zero probability: the tail is -inf / index 0x7fffffff; the host refuses the step
        //  as torch.multinomial refuses to draw more samples than there are non-zero categories)
        if (tid == 0) { out_s[b * M + r] = s_a[0]; out_i[b * M + r] = s_i[0]; }
        if (s_who[0] == tid) ++head;
        __syncthreads();
    }
}

// min_keep > 1: kth_s / kth_i = scratch for the min_keep largest logits of each of the B K decoder rows (fp32 / int32 [B K min_keep])
hipError_t launch_beam_sample(const float *logits, const float *run, int B, int K, int V, int M, float temperature, const float *pmax,
                              const float *thr, int min_keep, float *kth_s, int32_t *kth_i, uint64_t seed, int step, float *lse,
                              float *out_s, int32_t *out_i, hipStream_t s) {
    if (M < 1 || M > BEAM_MAXM || K < 1 || (int64_t)K * V >= 0x7fffffff || !(temperature > 0.f)) return hipErrorInvalidValue;
    if (min_keep < 1 || min_keep > BEAM_MAXM || (min_keep > 1 && (!kth_s || !kth_i))) return hipErrorInvalidValue;
    hipLaunchKernelGGL(beam_lse_kernel, dim3(B * K), dim3(256), 0, s, logits, V, lse);
    if (min_keep > 1)      // the min_keep best logits of every decoder row (the top-M kernel on single rows, no offsets)
        hipLaunchKernelGGL(beam_topk_kernel, dim3(B * K), dim3(256), 0, s, logits, (const float *)nullptr, (const float *)nullptr, 1, V, min_keep,
                           kth_s, kth_i);
    hipLaunchKernelGGL(beam_sample_kernel, dim3(B), dim3(256), 0, s, logits, lse, run, pmax, thr, min_keep > 1 ? kth_s : (const float *)nullptr,
                       min_keep, 1.0f / temperature, seed, step, K, V, M, out_s, out_i);
    return hipGetLastError();
}

hipError_t launch_beam_topk(const float *logits, const float *run, int B, int K, int V, int M, float *lse, float *out_s,
                            int32_t *out_i, hipStream_t s) {
    if (M < 1 || M > BEAM_MAXM || K < 1 || (int64_t)K * V >= 0x7fffffff) return hipErrorInvalidValue;
    hipLaunchKernelGGL(beam_lse_kernel, dim3(B * K), dim3(256), 0, s, logits, V, lse);
    hipLaunchKernelGGL(beam_topk_kernel, dim3(B), dim3(256), 0, s, logits, lse, run, K, V, M, out_s, out_i);
    return hipGetLastError();
}

// One pass of the cache-row permutation of a beam step, on the FILLED slots only: a cache row is [kv head][slot][hd] with
// `sub_halfs` between kv heads, of which slots 0 .. T0 + *step - 1 hold keys / values (the rest is never read before it is
// written).  back = 0: tmp[r] = cache[idx[r]]; back = 1: cache[r] = tmp[r].  Rows that keep their place (idx[r] == r) are skipped
// in both passes.  (Round 4 moved whole capacity-sized rows, 4 x the cache per step: advisor finding.)
__global__ __launch_bounds__(256) void kv_gather_rows_kernel(half_t *__restrict__ cache, half_t *__restrict__ tmp,
                                                             const int32_t *__restrict__ idx, int64_t row_halfs, int64_t sub_halfs,
                                                             const int32_t *__restrict__ step, int T0, int hd, int back) {
    const int r = blockIdx.y, sub = blockIdx.z;
    const int from = idx[r];
    if (from == r) return;
    const int64_t n8 = ((int64_t)(T0 + *step) * hd) >> 3;
    const h8 *s = reinterpret_cast<const h8 *>(back ? tmp + (int64_t)r * row_halfs + sub * sub_halfs : cache + (int64_t)from * row_halfs + sub * sub_halfs);
    h8 *d = reinterpret_cast<h8 *>(back ? cache + (int64_t)r * row_halfs + sub * sub_halfs : tmp + (int64_t)r * row_halfs + sub * sub_halfs);
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n8; i += (int64_t)gridDim.x * 256) d[i] = s[i];
}
hipError_t launch_kv_gather_rows(half_t *cache, half_t *tmp, const int32_t *idx, int R, int64_t row_halfs, int nsub, int64_t sub_halfs,
                                 const int32_t *step, int T0, int hd, int back, hipStream_t s) {
    if ((row_halfs & 7) || (sub_halfs & 7) || (hd & 7) || nsub < 1) return hipErrorInvalidValue;
    const int gx = (int)((sub_halfs >> 3) + 255) / 256;
    hipLaunchKernelGGL(kv_gather_rows_kernel, dim3(gx > 16 ? 16 : gx, R, nsub), dim3(256), 0, s, cache, tmp, idx, row_halfs, sub_halfs, step,
                       T0, hd, back);
    return hipGetLastError();
}

}  // namespace opus
