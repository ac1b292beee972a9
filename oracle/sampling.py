"""Sampling head (row N1).  TEST INFRASTRUCTURE.

The reference samples by default (eval/run_opus_ddp.py:126-128,156-157: do_sample when temperature > 0,
temperature 0.1, top_p 0.7) through transformers' GenerationMixin: TemperatureLogitsWarper, TopPLogitsWarper,
softmax, torch.multinomial (local copy: transformers/generation/logits_process.py:515-540).  This restates the
distribution the next token is drawn from; the draw itself depends on the RNG, so parity is distributional.
"""
import torch


def _warp(scores: torch.Tensor, top_p: float, top_k: int, min_keep: int = 1) -> torch.Tensor:
    """TopKLogitsWarper then TopPLogitsWarper on already temperature-scaled scores [B,V]: filtered entries -> -inf.
    min_keep = the warpers' min_tokens_to_keep: 1 when sampling one sequence; with num_beams > 1 GenerationMixin builds both with
    #eos + 1 (2 without an EOS id) - generation/utils.py _get_logits_processor, "keep at least one non-eos token"."""
    if top_k and top_k > 0:                                       # logits_process.py TopKLogitsWarper: ties with the k-th stay
        k = min(max(int(top_k), int(min_keep)), scores.shape[-1])
        scores = scores.masked_fill(scores < torch.topk(scores, k)[0][..., -1, None], float("-inf"))
    sorted_logits, sorted_idx = torch.sort(scores, descending=False)
    cum = sorted_logits.softmax(-1).cumsum(-1)
    remove = cum <= (1 - top_p)
    remove[..., -int(min_keep):] = False                       # min_tokens_to_keep
    mask = remove.scatter(1, sorted_idx, remove)
    return scores.masked_fill(mask, float("-inf"))


def beam_sample_distribution(logits: torch.Tensor, run_scores: torch.Tensor, temperature: float, top_p: float,
                             top_k: int = 0, min_keep: int = 2) -> torch.Tensor:
    """Beam-sample (GenerationMixin._beam_search with do_sample; generation/utils.py: log_softmax, the warpers on the
    log-probabilities, + running beam scores, softmax over the flattened [K V]): fp32 logits [K,V] of one batch row's beams and
    their running scores [K] -> the probabilities [K V] the FIRST of the M continuations is drawn from (torch.multinomial
    without replacement draws the next ones from the same weights with the drawn entries removed).  min_keep = #eos + 1, at
    least 2: the min_tokens_to_keep GenerationMixin gives both warpers under beam-sample."""
    lp = torch.log_softmax(logits.float(), dim=-1) / temperature
    lp = _warp(lp, top_p, top_k, min_keep)
    return (lp + run_scores.float()[:, None]).reshape(-1).softmax(-1)


def sampling_distribution(logits: torch.Tensor, temperature: float, top_p: float, top_k: int = 0) -> torch.Tensor:
    """fp32 logits [B,V] -> probabilities [B,V] after temperature, top-k (0 = off: transformers >= 5's default; 4.46.3, the
    reference's pin, defaults to 50) and nucleus filtering."""
    scores = logits / temperature
    if top_k and top_k > 0:
        return _warp(scores, top_p, top_k).softmax(-1)
    sorted_logits, sorted_idx = torch.sort(scores, descending=False)
    cum = sorted_logits.softmax(-1).cumsum(-1)
    remove = cum <= (1 - top_p)
    remove[..., -1:] = False                                   # min_tokens_to_keep = 1
    mask = remove.scatter(1, sorted_idx, remove)
    return scores.masked_fill(mask, float("-inf")).softmax(-1)
