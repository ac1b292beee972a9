"""Sampling head (row N1).  TEST INFRASTRUCTURE.

The reference samples by default (eval/run_opus_ddp.py:126-128,156-157: do_sample when temperature > 0,
temperature 0.1, top_p 0.7) through transformers' GenerationMixin: TemperatureLogitsWarper, TopPLogitsWarper,
softmax, torch.multinomial (local copy: transformers/generation/logits_process.py:515-540).  This restates the
distribution the next token is drawn from; the draw itself depends on the RNG, so parity is distributional.
"""
import torch


def sampling_distribution(logits: torch.Tensor, temperature: float, top_p: float) -> torch.Tensor:
    """fp32 logits [B,V] -> probabilities [B,V] after temperature and nucleus filtering."""
    scores = logits / temperature
    sorted_logits, sorted_idx = torch.sort(scores, descending=False)
    cum = sorted_logits.softmax(-1).cumsum(-1)
    remove = cum <= (1 - top_p)
    remove[..., -1:] = False                                   # min_tokens_to_keep = 1
    mask = remove.scatter(1, sorted_idx, remove)
    return scores.masked_fill(mask, float("-inf")).softmax(-1)
