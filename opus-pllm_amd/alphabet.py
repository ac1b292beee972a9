"""ESM-2 token alphabet and batch converter (row E0 of SURVEY 8a).

fair_esm 2.0.0 is a third-party dependency absent from /root/reference (requirements.txt:6); its
`Alphabet.from_architecture("ESM-1b")` + `BatchConverter` are restated here from their published
behaviour, anchored on the reference call sites cstp_v3/modelling.py:34,44-45:
  * 33 symbols: <cls>=0 <pad>=1 <eos>=2 <unk>=3, "LAGVSERTIDPKQNFYMHWCXBUZO.-" = 4..30,
    <null_1>=31, <mask>=32;
  * a row is <cls> residues <eos>, right-padded with <pad> to the longest row + 2;
  * whitespace is dropped, multi-character symbols such as "<mask>" are recognised, and a symbol
    outside the alphabet raises KeyError (fair_esm indexes tok_to_idx without a default).
"""
from __future__ import annotations

from typing import List, Sequence, Tuple

import numpy as np

_STANDARD = "LAGVSERTIDPKQNFYMHWCXBUZO.-"
ALL_TOKS: List[str] = ["<cls>", "<pad>", "<eos>", "<unk>"] + list(_STANDARD) + ["<null_1>", "<mask>"]
TOK_TO_IDX = {t: i for i, t in enumerate(ALL_TOKS)}
CLS_IDX, PAD_IDX, EOS_IDX, UNK_IDX, MASK_IDX = 0, 1, 2, 3, 32
_SPECIALS = [t for t in ALL_TOKS if len(t) > 1]


# Byte table of the fast path: every single-character symbol -> its id, ASCII whitespace -> -1 (dropped), anything else -> -2
# (KeyError, as fair_esm).  The ASCII whitespace set is what str.isspace() accepts below 128.
_LUT = np.full(256, -2, dtype=np.int32)
for _t, _i in TOK_TO_IDX.items():
    if len(_t) == 1:
        _LUT[ord(_t)] = _i
for _c in "\t\n\v\f\r \x1c\x1d\x1e\x1f":
    _LUT[ord(_c)] = -1


def encode_array(seq: str) -> np.ndarray:
    """Token ids of one sequence as int32.  Plain residue strings (ASCII, no "<...>" symbol: every protein of a dataset) go
    through one table look-up over the bytes; anything else takes the per-character walk of `encode` (same rules)."""
    if seq.isascii() and "<" not in seq:
        ids = _LUT[np.frombuffer(seq.encode("ascii"), dtype=np.uint8)]
        if ids.size and ids.min() < 0:
            bad = np.flatnonzero(ids == -2)
            if bad.size:
                raise KeyError(seq[int(bad[0])])
            ids = ids[ids >= 0]
        return ids
    return np.asarray(encode(seq), dtype=np.int32)


def encode(seq: str) -> List[int]:
    ids: List[int] = []
    i, n = 0, len(seq)
    while i < n:
        ch = seq[i]
        if ch.isspace():
            i += 1
            continue
        if ch == "<":
            for sp in _SPECIALS:
                if seq.startswith(sp, i):
                    ids.append(TOK_TO_IDX[sp])
                    i += len(sp)
                    break
            else:
                raise KeyError(ch)
            continue
        ids.append(TOK_TO_IDX[ch])   # KeyError for symbols outside the alphabet, as fair_esm
        i += 1
    return ids


def batch_convert(seqs: Sequence[str]) -> Tuple[np.ndarray, np.ndarray]:
    """-> (tokens int32 [B, Lmax+2], lens int32 [B]) ; lens counts <cls> and <eos> (modelling.py:45)."""
    enc = [encode_array(s) for s in seqs]
    width = max((len(e) for e in enc), default=0) + 2
    toks = np.full((len(enc), width), PAD_IDX, dtype=np.int32)
    lens = np.zeros((len(enc),), dtype=np.int32)
    for b, e in enumerate(enc):
        toks[b, 0] = CLS_IDX
        toks[b, 1:1 + len(e)] = e
        toks[b, 1 + len(e)] = EOS_IDX
        lens[b] = len(e) + 2
    return toks, lens


def batch_convert_packed(seqs: Sequence[str]) -> Tuple[np.ndarray, np.ndarray]:
    """Token-packed form of `batch_convert` -> (tokens int32 [sum(len_b)], cu int32 [B + 1]): the rows <cls> residues <eos> back
    to back WITHOUT padding and their offsets (cu[0] = 0); what opus_esm2_encode_packed consumes."""
    enc = [encode_array(s) for s in seqs]
    cu = np.zeros((len(enc) + 1,), dtype=np.int32)
    for b, e in enumerate(enc):
        cu[b + 1] = cu[b] + len(e) + 2
    toks = np.empty((int(cu[-1]),), dtype=np.int32)
    for b, e in enumerate(enc):
        toks[cu[b]] = CLS_IDX
        toks[cu[b] + 1:cu[b + 1] - 1] = e
        toks[cu[b + 1] - 1] = EOS_IDX
    return toks, cu
