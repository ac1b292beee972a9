"""Whitespace tokenizer used by the tokenizer_seq_token fixtures (same rule as tools/gen_golden.py)."""
import types


class FakeTokenizer:
    def __init__(self, add_bos=True, bos_token_id=1):
        self.add_bos, self.bos_token_id = add_bos, bos_token_id
        self.eos_token_id = self.pad_token_id = self.unk_token_id = 2
        self.eos_token = self.pad_token = self.unk_token = "</s>"

    def __call__(self, text):
        ids = [3 + (sum(ord(c) * (i + 1) for i, c in enumerate(w)) % 90) for w in text.split()]
        return types.SimpleNamespace(input_ids=([self.bos_token_id] if self.add_bos else []) + ids)
