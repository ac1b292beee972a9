"""Host-side mirror of the reference's model interface for the multi_modality_v1 inference path.

Same names, argument meaning and error behaviour as
  multi_modality_v1/model/language_model/opus_llama.py  (OpusLlamaForCausalLM.generate :95-132)
  multi_modality_v1/model/opus_arch.py                  (encode_* :103-131,
                                                          prepare_inputs_labels_for_multimodal :133-294)
so that eval/run_opus_ddp.py-style callers drop in.  Every tensor op of the path runs in
libopus_pllm.so; PyTorch only owns device memory and streams.
"""
from __future__ import annotations

import ctypes as C
import types
from typing import List, Optional, Sequence, Union

import numpy as np
import torch

from . import _cabi
from .alphabet import batch_convert, batch_convert_packed
from .config import OpusConfig
from .constants import DEFAULT_SEQ_TOKEN_INDEX, IGNORE_INDEX
from .weights import DeviceWeights


class _ProteinEncoderHandle:
    """What get_protein_encoder() returns: exposes get_protein_seq_embeddings like
    ProteinSeqEmbeddingExtractor (cstp_v3/modelling.py:37)."""

    def __init__(self, owner: "OpusLlamaForCausalLM"):
        self._owner = owner

    def get_protein_seq_embeddings(self, data: Sequence[str]) -> torch.Tensor:
        return self._owner._encode(list(data))


class _InnerModel:
    """What get_model() returns (OpusLlamaModel in the reference): embed_tokens + module handles."""

    def __init__(self, owner: "OpusLlamaForCausalLM"):
        self._owner = owner
        self.config = owner.config
        self.protein_encoder = _ProteinEncoderHandle(owner)

    def get_protein_encoder(self):
        return self.protein_encoder

    def embed_tokens(self, ids: torch.Tensor) -> torch.Tensor:
        return torch.nn.functional.embedding(ids.to(self._owner.device), self._owner.weights.tensors["dec.emb"])


class OpusLlamaForCausalLM:
    """MI355X-native stand-in for the reference's OpusLlamaForCausalLM (inference only)."""

    def __init__(self, cfg: OpusConfig, weights: DeviceWeights, device: Union[str, torch.device] = "cuda:0",
                 eos_token_id: Union[int, Sequence[int], None] = None, pad_token_id: Optional[int] = None):
        self.cfg = cfg.validate()
        self.device = torch.device(device)
        if self.device.type != "cuda":
            raise _cabi.OpusError(-102, "OpusLlamaForCausalLM needs a GPU device: there is no CPU path")
        self.weights = weights
        self._lib = _cabi.lib()
        self._ctx = C.c_void_p()
        idx = self.device.index if self.device.index is not None else torch.cuda.current_device()
        cc = _cabi.CConfig.from_config(cfg)
        _cabi.check(self._lib.opus_ctx_create(C.byref(cc), idx, C.byref(self._ctx)))
        weights.bind(self._ctx)
        self._stream = torch.cuda.Stream(self.device)
        self.config = types.SimpleNamespace(
            hidden_size=cfg.dec_dim, vocab_size=cfg.dec_vocab, has_switch_projector=True, has_protein_encoder=True,
            num_hidden_layers=cfg.dec_layers, device=str(self.device), model_type="opus_llama")
        eos = [] if eos_token_id is None else ([eos_token_id] if isinstance(eos_token_id, int) else list(eos_token_id))
        self.generation_config = types.SimpleNamespace(eos_token_id=eos, pad_token_id=pad_token_id)
        self.model = _InnerModel(self)

    # ------------------------------------------------------------------ lifecycle
    def new_context(self) -> "OpusLlamaForCausalLM":
        """A second context on the same device that SHARES this model's weights (read-only) and owns its workspace, KV cache,
        decode graph and stream: two batches can then be in flight on one GPU (two host threads, one per context) - proteins are
        independent, so one batch's kernels stream through the other's launch gaps, ramps and drains - mostly in the decode steps
        (`eval_ddp.py --inflight 2`, `bench.py`'s `two_in_flight`: +10 % throughput at batch 64; same ids as one context)."""
        g = self.generation_config
        return OpusLlamaForCausalLM(self.cfg, self.weights, self.device, eos_token_id=list(g.eos_token_id), pad_token_id=g.pad_token_id)

    def __del__(self):
        try:
            if getattr(self, "_ctx", None) and self._ctx.value:
                self._lib.opus_ctx_destroy(self._ctx)
                self._ctx = C.c_void_p()
        except Exception:
            pass

    def eval(self):
        return self

    def get_model(self):
        return self.model

    def get_protein_encoder(self):
        return self.model.get_protein_encoder()

    # ------------------------------------------------------------------ stream plumbing
    def _enter(self):
        self._stream.wait_stream(torch.cuda.current_stream(self.device))
        return self._stream.cuda_stream

    def _leave(self):
        torch.cuda.current_stream(self.device).wait_stream(self._stream)

    # ------------------------------------------------------------------ rows E0-E4
    packed_encoder = True      # token-packed (varlen) encoder; False: the padded, length-bucketed form (A/B and parity tests)

    def _encode(self, seqs: List[str], bucket: int = 256) -> torch.Tensor:
        """list[str] -> pooled fp32 [B, enc_dim].  Token-packed by default: the proteins' tokens go through the encoder back to
        back with no padding and no buckets (opus_esm2_encode_packed), max_batch proteins per call."""
        if self.packed_encoder:
            return self._encode_packed(seqs)
        return self._encode_padded(seqs, bucket)

    def _encode_packed(self, seqs: List[str]) -> torch.Tensor:
        cfg = self.cfg
        n = len(seqs)
        out = torch.empty((n, cfg.enc_dim), dtype=torch.float32, device=self.device)
        s = self._enter()
        keep = []
        for i0 in range(0, n, cfg.max_batch):
            toks, cu = batch_convert_packed(seqs[i0:i0 + cfg.max_batch])
            B = len(cu) - 1
            longest = int((cu[1:] - cu[:-1]).max())
            if longest > cfg.max_enc_tokens:
                raise _cabi.OpusError(-2, f"protein of {longest - 2} residues exceeds max_enc_tokens={cfg.max_enc_tokens}")
            with torch.cuda.stream(self._stream):
                d_tok = torch.from_numpy(toks).to(self.device, non_blocking=True)
                cu_arr = (C.c_int32 * (B + 1))(*[int(v) for v in cu])
                _cabi.check(self._lib.opus_esm2_encode_packed(self._ctx, d_tok.data_ptr(), cu_arr, B, out[i0:].data_ptr(), s))
            keep.append(d_tok)
            self._last_enc_shape = (1, int(cu[-1]))
        self._leave()
        return out

    def _encode_padded(self, seqs: List[str], bucket: int = 256) -> torch.Tensor:
        """list[str] -> pooled fp32 [B, enc_dim].  Mixed lengths are processed in length buckets
        (multiples of `bucket` residues) so padding never exceeds one bucket; per-protein results do
        not depend on the batch they ran in (key padding is masked).  256 measured best on 64 proteins of
        128-1024 residues (363 ms per batch vs 399 ms at 64 and 381 ms unbucketed): fewer, larger GEMMs
        outweigh the extra padding."""
        cfg = self.cfg
        n = len(seqs)
        out = torch.empty((n, cfg.enc_dim), dtype=torch.float32, device=self.device)
        order = sorted(range(n), key=lambda i: len(seqs[i]))
        groups: List[List[int]] = []
        for i in order:
            key = (len(seqs[i]) + bucket - 1) // bucket
            if groups and groups[-1][0] == key and len(groups[-1][1]) < cfg.max_batch:
                groups[-1][1].append(i)
            else:
                groups.append([key, [i]])
        s = self._enter()
        keep = []
        for _, idxs in groups:
            toks, lens = batch_convert([seqs[i] for i in idxs])
            if toks.shape[1] < 3:       # only empty strings: <cls><eos> + one pad column (their mean over zero residues
                import numpy as np      # is NaN, as in the reference; such rows carry no <seq> placeholder)
                from .alphabet import PAD_IDX
                toks = np.concatenate([toks, np.full((toks.shape[0], 1), PAD_IDX, dtype=toks.dtype)], axis=1)
            B, T = toks.shape
            if T > cfg.max_enc_tokens:
                raise _cabi.OpusError(-2, f"protein of {T - 2} residues exceeds max_enc_tokens={cfg.max_enc_tokens}")
            with torch.cuda.stream(self._stream):
                d_tok = torch.from_numpy(toks).to(self.device, non_blocking=True)
                d_len = torch.from_numpy(lens).to(self.device, non_blocking=True)
                pooled = torch.empty((B, cfg.enc_dim), dtype=torch.float32, device=self.device)
                _cabi.check(self._lib.opus_esm2_encode(self._ctx, d_tok.data_ptr(), d_len.data_ptr(), B, T,
                                                       pooled.data_ptr(), s))
                out[torch.tensor(idxs, device=self.device)] = pooled
            keep.append((d_tok, d_len, pooled))
        self._leave()
        self._last_enc_shape = (B, T)
        return out

    def encode_seq2embedding(self, seq) -> torch.Tensor:
        """opus_arch.py:103-114: str | list[str] -> fp32 [B, enc_dim]; other types: NotImplementedError."""
        if type(seq) is not list:
            seq = [seq]
        if type(seq[0]) is str:
            return self.get_protein_encoder().get_protein_seq_embeddings(seq)
        raise NotImplementedError

    def encode_projector_embedding(self, extractor_embedding: torch.Tensor) -> torch.Tensor:
        """opus_arch.py:115-121 -> CSTPBase.protein_forward (modelling.py:396-400): fp16 [B, proj_dim].  Without a CSTP
        checkpoint the reference installs an identity module (opus_arch.py:70-80): the input comes back unchanged."""
        if not self.cfg.has_protein_projector:
            return extractor_embedding
        x = extractor_embedding.to(self.device, torch.float32).contiguous()
        B = x.shape[0]
        s = self._enter()
        with torch.cuda.stream(self._stream):
            y = torch.empty((B, self.cfg.switch_in), dtype=_cabi.operand_dtype(), device=self.device)
            _cabi.check(self._lib.opus_protein_projector(self._ctx, x.data_ptr(), B, y.data_ptr(), s))
        self._leave()
        return y

    def switch_projector_embedding(self, seq_embedding: torch.Tensor) -> torch.Tensor:
        """opus_arch.py:122-131: [B, switch_in] -> fp16 [B, n_prot_tokens, hidden]."""
        y = seq_embedding.to(self.device, _cabi.operand_dtype()).contiguous()
        B = y.shape[0]
        s = self._enter()
        with torch.cuda.stream(self._stream):
            z = torch.empty((B, self.cfg.n_prot_tokens, self.cfg.dec_dim), dtype=_cabi.operand_dtype(), device=self.device)
            _cabi.check(self._lib.opus_switch_projector(self._ctx, y.data_ptr(), B, z.data_ptr(), s))
        self._leave()
        return z

    # ------------------------------------------------------------------ rows S1-S3
    def _splice(self, input_ids, attention_mask, prot, inference_mode):
        cfg = self.cfg
        ids = input_ids.to(self.device, torch.int64).contiguous()
        B, Tt = ids.shape
        m = None if attention_mask is None else attention_mask.to(self.device).bool().to(torch.uint8).contiguous()
        prot = prot.to(self.device, _cabi.operand_dtype()).contiguous()
        s = self._enter()
        with torch.cuda.stream(self._stream):
            emb = torch.empty((B, cfg.max_prompt, cfg.dec_dim), dtype=_cabi.operand_dtype(), device=self.device)
            mo = torch.empty((B, cfg.max_prompt), dtype=torch.uint8, device=self.device)
            po = torch.empty((B, cfg.max_prompt), dtype=torch.int32, device=self.device)
            T_out = C.c_int32(0)
            max_len = int(getattr(self.config, "tokenizer_model_max_length", None) or 0)
            _cabi.check(self._lib.opus_splice_pad(self._ctx, ids.data_ptr(), None if m is None else m.data_ptr(), B, Tt,
                                                  prot.data_ptr(), prot.shape[0], 1 if inference_mode else 0, max_len,
                                                  emb.data_ptr(), mo.data_ptr(), po.data_ptr(), C.byref(T_out), s))
            T = T_out.value
            # the kernel wrote [B, T] densely at the head of the capacity-sized buffers
            emb = emb.view(-1)[: B * T * cfg.dec_dim].view(B, T, cfg.dec_dim)
            mo = mo.view(-1)[: B * T].view(B, T)
            po = po.view(-1)[: B * T].view(B, T)
        self._leave()
        return emb, mo, po

    PROJECT_CHUNK = 4096       # rows per projector launch of project_dataset (8H x 8H GEMM at 1.3 PFLOP/s from ~1024 rows up)

    def project_dataset(self, pooled: torch.Tensor) -> torch.Tensor:
        """The batched projector stage of the two-stage pipeline: pooled ESM-2 embeddings of a whole dataset
        shard fp32 [N, enc_dim] (the `input_embed` field written by generate_esm_embedding.py, consumed by the reference at
        opus_arch.py:151-161) -> protein tokens fp16 [N, n_prot_tokens, hidden], the modality projectors running at
        M = PROJECT_CHUNK rows (MFMA-bound GEMMs) instead of re-streaming their 2.5 GB of weights for every batch of 8.
        Every launch has the SAME shape - the last chunk is padded with zero rows - so a row's tokens (and the greedy ids that
        follow) do not depend on the shard size, i.e. on the number of ranks the dataset is split over.
        Feed slices of the result to generate(protein_tokens=...)."""
        x = pooled.to(self.device, torch.float32).contiguous()
        N, C_ = x.shape[0], self.PROJECT_CHUNK
        s = self._enter()
        with torch.cuda.stream(self._stream):
            z = torch.empty((N, self.cfg.n_prot_tokens, self.cfg.dec_dim), dtype=_cabi.operand_dtype(), device=self.device)
            zc = None
            for r0 in range(0, N, C_):
                n = min(C_, N - r0)
                if n == C_:
                    _cabi.check(self._lib.opus_projector_forward(self._ctx, x[r0:].data_ptr(), C_, z[r0:].data_ptr(), None, s))
                    continue
                xc = torch.zeros((C_, x.shape[1]), dtype=torch.float32, device=self.device)
                xc[:n] = x[r0:]
                zc = torch.empty((C_,) + tuple(z.shape[1:]), dtype=_cabi.operand_dtype(), device=self.device)
                _cabi.check(self._lib.opus_projector_forward(self._ctx, xc.data_ptr(), C_, zc.data_ptr(), None, s))
                z[r0:] = zc[:n]
        self._leave()
        return z

    def prepare_inputs_labels_for_multimodal(self, input_ids, position_ids, attention_mask, past_key_values, labels,
                                             seq, seq_embedding=None, inference_mode=False, protein_tokens=None):
        """opus_arch.py:133-294.  Returns (None, position_ids|None, attention_mask|None, past_key_values,
        inputs_embeds [B,T,H] fp16, labels|None); inputs unchanged when seq is None or T == 1.
        protein_tokens (an extension, see project_dataset): already projected [n, n_prot_tokens, hidden] blocks."""
        if seq is None or self.get_protein_encoder() is None or input_ids.shape[1] == 1:
            return input_ids, position_ids, attention_mask, past_key_values, None, labels
        if protein_tokens is not None:
            seq_embedding = protein_tokens
        else:
            if seq_embedding is None:
                seq_embedding = self.encode_seq2embedding(seq)
            seq_embedding = self.encode_projector_embedding(seq_embedding)
            if self.config.has_switch_projector:
                seq_embedding = self.switch_projector_embedding(seq_embedding)
        if seq_embedding.ndimension() == 2:
            seq_embedding = seq_embedding.unsqueeze(1)
        elif seq_embedding.ndimension() != 3:
            raise NotImplementedError
        emb, mask_out, pos_out = self._splice(input_ids, attention_mask, seq_embedding, inference_mode)
        new_labels = None
        if labels is not None:
            new_labels = _splice_labels(input_ids, attention_mask, labels, self.cfg.n_prot_tokens, emb.shape[1],
                                        inference_mode).to(labels.device)
        out_mask = None if attention_mask is None else mask_out.to(dtype=attention_mask.dtype)
        out_pos = None if position_ids is None else pos_out.to(dtype=position_ids.dtype)
        return None, out_pos, out_mask, past_key_values, emb, new_labels

    # BASELINE.json's north_star calls the method by its short name; the reference defines only the long one (opus_arch.py:133)
    prepare_inputs_for_multimodal = prepare_inputs_labels_for_multimodal

    # ------------------------------------------------------------------ rows G0, G1, D1-D4
    @torch.no_grad()
    def generate(self, inputs: Optional[torch.Tensor] = None, seq=None, seq_embedding=None, **kwargs) -> torch.LongTensor:
        """opus_llama.py:95-132 + GenerationMixin greedy search: returns ONLY the new ids [B, n_new]."""
        kwargs.pop("position_ids", None)
        protein_tokens = kwargs.pop("protein_tokens", None)
        attention_mask = kwargs.pop("attention_mask", None)
        if "inputs_embeds" in kwargs:
            raise NotImplementedError("`inputs_embeds` is not supported")
        do_sample = bool(kwargs.pop("do_sample", False))
        temperature = kwargs.pop("temperature", None)
        top_p = kwargs.pop("top_p", None)
        top_k = kwargs.pop("top_k", self.default_top_k)                  # (transformers 4.46.3: GenerationConfig.top_k = 50)
        seed = kwargs.pop("seed", None)
        num_beams = kwargs.pop("num_beams", 1)
        num_beams = 1 if num_beams is None else int(num_beams)
        max_new = int(kwargs.pop("max_new_tokens", 32))
        kwargs.pop("use_cache", None)
        self.set_stop_sequence(kwargs.pop("stop_sequence", None))          # extension (opt-in "###" early stop), see below
        pad_id = kwargs.pop("pad_token_id", self.generation_config.pad_token_id)
        eos = kwargs.pop("eos_token_id", self.generation_config.eos_token_id)
        eos = [] if eos is None else ([int(eos)] if isinstance(eos, int) else [int(e) for e in eos])
        sampler = None
        if do_sample:       # HF: temperature defaults to 1.0, top_p to 1.0; draws keyed by (seed, row, step)
            t = 1.0 if temperature is None else float(temperature)
            if not t > 0:
                raise ValueError("`temperature` has to be a strictly positive float when do_sample=True")
            if seed is None:
                seed = int(torch.randint(0, 2 ** 62, (1,)).item())       # follows torch.manual_seed
            k = 0 if top_k is None else int(top_k)
            if k < 0:
                raise ValueError("`top_k` has to be a non-negative integer (0 / None: no top-k filtering)")
            sampler = (t, 1.0 if top_p is None else float(top_p), int(seed), k)
        if num_beams < 1:
            raise ValueError("`num_beams` has to be an integer strictly greater than 0")
        beam_pad = pad_id                                      # (HF's beam fill value distinguishes None / 0 from an id)
        if pad_id is None:
            pad_id = eos[0] if eos else 0
        if inputs is None:
            raise ValueError("generate() needs input ids")
        if seq is not None:
            _, _, mask, _, embeds, _ = self.prepare_inputs_labels_for_multimodal(
                inputs, None, attention_mask if attention_mask is not None else torch.ones_like(inputs, dtype=torch.bool),
                None, None, seq, seq_embedding, inference_mode=True, protein_tokens=protein_tokens)
        else:
            dummy = torch.zeros((inputs.shape[0], self.cfg.n_prot_tokens, self.cfg.dec_dim), dtype=_cabi.operand_dtype(),
                                device=self.device)
            embeds, mask, _ = self._splice(inputs, attention_mask, dummy, True)
        if num_beams > 1:
            return self._beam_search(embeds, mask, max_new, eos, beam_pad, num_beams, sampler)
        return self._greedy(embeds, mask, max_new, eos, int(pad_id), sampler)

    # TopKLogitsWarper of the sampling paths.  The reference pins transformers 4.46.3 (requirements.txt:20), whose
    # GenerationConfig.top_k defaults to 50 whenever it samples (transformers >= 5 defaults to None); the reference's drivers never
    # set it (run_opus_ddp.py:126-132), so 50 is what its sampling runs with.  generate(top_k=...) overrides per call; 0 / None = off.
    default_top_k = 50

    # Beam-sample: the reference's pin, transformers 4.46.3, sorts the M sampled continuations of a row by score (descending)
    # before its beam scorer looks at them, so "the first K may finish" means the K best of the draws; transformers >= 4.50
    # (`_get_top_k_continuations`, the installed 5.15 the host tests compare with) keeps the order drawn.  True = the pin.
    beam_sample_sorted = True

    def _set_top_k(self, k: int) -> None:
        if k != getattr(self, "_top_k", 0):
            _cabi.check(self._lib.opus_set_sampling_top_k(self._ctx, int(k)))
            self._top_k = int(k)

    def _beam_search(self, embeds, mask, max_new, eos, pad_id, K, sampler=None) -> torch.Tensor:
        """transformers GenerationMixin._beam_search (what run_opus_ddp.py:129,158 reaches with --num_beams K): B x K decoder rows,
        per step the device picks the M continuations of every batch row out of the K V - the best M by accumulated log-probability
        (opus_beam_topk; temperature 0) or, with `sampler` (temperature > 0: beam-sample), M drawn without replacement after the
        warpers (opus_beam_sample_topk) - and permutes the KV cache rows of the surviving beams (opus_kv_reorder); the host keeps
        the O(K) bookkeeping (beam.BeamState).  Returns the best finished sequence of every row [B, n] (rows that stopped
        earlier are filled as HF fills them)."""
        from .beam import BeamState
        B, T, _ = embeds.shape
        cfg = self.cfg
        if B * K > cfg.max_batch:
            raise _cabi.OpusError(-2, f"beam search runs batch x num_beams = {B} x {K} decoder rows: the context holds max_batch={cfg.max_batch}")
        if max_new > cfg.max_new_tokens:
            raise _cabi.OpusError(-2, f"max_new_tokens={max_new} exceeds the context's {cfg.max_new_tokens}")
        state = BeamState(B, K, max_new, eos, pad_id, cfg.dec_vocab)
        M = state.M
        if M > 16:
            raise NotImplementedError(f"beam search keeps max(2, 1 + #eos) x num_beams = {M} candidates per row; at most 16 are built")
        emb = embeds.repeat_interleave(K, dim=0).contiguous()          # _expand_inputs_for_generation: row b K + k
        msk = mask.repeat_interleave(K, dim=0).contiguous()
        self.prefill_logits(emb, msk)                                   # (the logits stay in the context)
        s = self._enter()
        with torch.cuda.stream(self._stream):
            d_run = torch.empty((B * K,), dtype=torch.float32, device=self.device)
            d_sc = torch.empty((B, M), dtype=torch.float32, device=self.device)
            d_ix = torch.empty((B, M), dtype=torch.int32, device=self.device)
            ident = np.tile(np.arange(K, dtype=np.int64), (B, 1))
            base = (np.arange(B, dtype=np.int64) * K)[:, None]
            if sampler is not None:
                self._set_top_k(sampler[3] if len(sampler) > 3 else self.default_top_k)
            while True:
                d_run.copy_(torch.from_numpy(state.running_scores.reshape(-1)), non_blocking=True)
                if sampler is None:
                    _cabi.check(self._lib.opus_beam_topk(self._ctx, d_run.data_ptr(), B, K, M, d_sc.data_ptr(), d_ix.data_ptr(), s))
                else:
                    _cabi.check(self._lib.opus_beam_sample_topk(self._ctx, None, d_run.data_ptr(), B, K, M, sampler[0], sampler[1],
                                                                sampler[2], state.cur, d_sc.data_ptr(), d_ix.data_ptr(), s))
                sc, ix = d_sc.cpu().numpy(), d_ix.cpu().numpy()         # (synchronises this stream)
                # (fp32 softmax gives exactly zero below exp(-103.97): continuations that far under the row's best - filtered ones,
                #  those of dead beams at -1e9 - are the "non-negative categories" torch.multinomial finds too few of)
                if sampler is not None and ((ix == 0x7fffffff) | (sc < sc.max(axis=1, keepdims=True) - 103.0)).any():
                    self._leave()
                    raise RuntimeError("invalid multinomial distribution (with replacement=False, not enough non-negative category "
                                       f"to sample): beam-sample draws {M} continuations per row, the temperature / top_k / top_p "
                                       "filters left fewer")        # (torch.multinomial's message: what the reference raises here)
                if sampler is not None and self.beam_sample_sorted:     # 4.46.3: torch.sort(next_token_scores, descending=True) behind the draw
                    order = np.argsort(-sc, axis=1, kind="stable")
                    sc, ix = np.take_along_axis(sc, order, 1), np.take_along_axis(ix, order, 1)
                tok, src, done = state.step(sc, ix)
                if done:
                    break
                if not np.array_equal(src, ident):
                    d_src = torch.from_numpy((src + base).reshape(-1).astype(np.int32)).to(self.device)
                    _cabi.check(self._lib.opus_kv_reorder(self._ctx, d_src.data_ptr(), B * K, s))
                d_tok = torch.from_numpy(tok.reshape(-1).astype(np.int32)).to(self.device)
                _cabi.check(self._lib.opus_llama_decode_step(self._ctx, d_tok.data_ptr(), None, s))
        self._leave()
        self.last_beam_scores = torch.from_numpy(state.result_scores())
        return torch.from_numpy(state.result()).to(self.device)

    def set_stop_sequence(self, ids: Optional[Sequence[int]]) -> None:
        """Opt-in early stop: a row is finished once its new ids end with `ids` (at most 8) - e.g.
        tokenizer.encode("###", add_special_tokens=False), the marker the reference cuts the decoded text at
        (eval/run_opus_ddp.py:19-27).  The cut text is unchanged; a batch whose rows have all stopped ends early.
        None / empty clears it (the reference's behaviour: decode to max_new_tokens)."""
        ids = [int(t) for t in ids] if ids is not None else []
        if ids == getattr(self, "_stop_ids", []):
            return
        arr = (C.c_int32 * max(1, len(ids)))(*ids)
        _cabi.check(self._lib.opus_set_stop_sequence(self._ctx, arr, len(ids)))
        self._stop_ids = ids

    def _greedy(self, embeds, mask, max_new, eos, pad_id, sampler=None) -> torch.Tensor:
        B, T, _ = embeds.shape
        embeds = embeds.contiguous()
        mask = mask.to(torch.uint8).contiguous()
        s = self._enter()
        with torch.cuda.stream(self._stream):
            # one persistent id buffer: its address is part of the captured decode graph's identity
            if getattr(self, "_out_ids", None) is None or self._out_ids.shape[0] < B or self._out_ids.shape[1] < max_new:
                self._out_ids = torch.empty((max(B, self.cfg.max_batch), max(max_new, self.cfg.max_new_tokens)),
                                            dtype=torch.int32, device=self.device)
            out = self._out_ids.view(-1)[: B * max_new].view(B, max_new)
            out.fill_(pad_id)
            n_out = C.c_int32(0)
            eos_arr = (C.c_int32 * max(1, len(eos)))(*eos)
            if sampler is not None:
                self._set_top_k(sampler[3] if len(sampler) > 3 else self.default_top_k)
            if sampler is None:
                _cabi.check(self._lib.opus_generate_greedy(self._ctx, embeds.data_ptr(), mask.data_ptr(), B, T, max_new,
                                                           eos_arr, len(eos), pad_id, out.data_ptr(), C.byref(n_out), s))
            else:
                _cabi.check(self._lib.opus_generate_sample(self._ctx, embeds.data_ptr(), mask.data_ptr(), B, T, max_new,
                                                           eos_arr, len(eos), pad_id, sampler[0], sampler[1], sampler[2],
                                                           out.data_ptr(), C.byref(n_out), s))
        self._leave()
        return out[:, : n_out.value].long()              # (a copy: the id buffer is reused by the next call)

    def generate_from_tokens(self, d_tokens, d_lens, input_ids: torch.Tensor,
                             attention_mask: Optional[torch.Tensor], max_new_tokens: int, eos: Sequence[int] = (),
                             pad_token_id: int = 0, bucket_rows: Optional[Sequence[torch.Tensor]] = None,
                             sampler=None) -> torch.Tensor:
        """generate() for callers whose inputs are already resident in HBM: ESM-2 token ids int32
        [B,T] + lens int32 [B] (alphabet.batch_convert), prompt ids int64 [B,T_text] on the device.
        Same result as generate(input_ids, seqs, ...); used by bench.py for the timed region.
        Length-bucketed form: d_tokens / d_lens / bucket_rows are lists (one entry per bucket; bucket_rows[k] =
        int64 device tensor with the batch rows of bucket k), as encode_seq2embedding buckets strings."""
        cfg = self.cfg
        B = input_ids.shape[0]
        if bucket_rows == "packed":      # d_tokens: packed int32 [M] on the device; d_lens: the HOST row offsets cu [B + 1]
            s = self._enter()
            with torch.cuda.stream(self._stream):
                pooled = torch.empty((B, cfg.enc_dim), dtype=torch.float32, device=self.device)
                cu_arr = (C.c_int32 * (B + 1))(*[int(v) for v in d_lens])
                _cabi.check(self._lib.opus_esm2_encode_packed(self._ctx, d_tokens.data_ptr(), cu_arr, B, pooled.data_ptr(), s))
                prot = torch.empty((B, cfg.n_prot_tokens, cfg.dec_dim), dtype=_cabi.operand_dtype(), device=self.device)
                _cabi.check(self._lib.opus_projector_forward(self._ctx, pooled.data_ptr(), B, prot.data_ptr(), None, s))
            self._leave()
            emb, mask, _ = self._splice(input_ids, attention_mask, prot, True)
            return self._greedy(emb, mask, int(max_new_tokens), [int(e) for e in eos], int(pad_token_id), sampler)
        buckets = list(zip(d_tokens, d_lens, bucket_rows)) if bucket_rows is not None else [(d_tokens, d_lens, None)]
        s = self._enter()
        with torch.cuda.stream(self._stream):
            pooled = torch.empty((B, cfg.enc_dim), dtype=torch.float32, device=self.device)
            for tok, lens, rows in buckets:
                out = pooled if rows is None else torch.empty((tok.shape[0], cfg.enc_dim), dtype=torch.float32, device=self.device)
                _cabi.check(self._lib.opus_esm2_encode(self._ctx, tok.data_ptr(), lens.data_ptr(), tok.shape[0], tok.shape[1],
                                                       out.data_ptr(), s))
                if rows is not None:
                    pooled[rows] = out
            prot = torch.empty((B, cfg.n_prot_tokens, cfg.dec_dim), dtype=_cabi.operand_dtype(), device=self.device)
            _cabi.check(self._lib.opus_projector_forward(self._ctx, pooled.data_ptr(), B, prot.data_ptr(), None, s))
        self._leave()
        emb, mask, _ = self._splice(input_ids, attention_mask, prot, True)
        return self._greedy(emb, mask, int(max_new_tokens), [int(e) for e in eos], int(pad_token_id), sampler)

    # ------------------------------------------------------------------ parity taps (tests / bench)
    def prefill_logits(self, embeds: torch.Tensor, mask: torch.Tensor) -> torch.Tensor:
        B, T, _ = embeds.shape
        embeds = embeds.to(self.device, _cabi.operand_dtype()).contiguous()
        mask = mask.to(self.device).to(torch.uint8).contiguous()
        s = self._enter()
        with torch.cuda.stream(self._stream):
            logits = torch.empty((B, self.cfg.dec_vocab), dtype=torch.float32, device=self.device)
            _cabi.check(self._lib.opus_llama_prefill(self._ctx, embeds.data_ptr(), mask.data_ptr(), B, T,
                                                     logits.data_ptr(), s))
        self._leave()
        return logits

    def decode_logits(self, tok: torch.Tensor) -> torch.Tensor:
        tok = tok.to(self.device, torch.int32).contiguous()
        s = self._enter()
        with torch.cuda.stream(self._stream):
            logits = torch.empty((tok.shape[0], self.cfg.dec_vocab), dtype=torch.float32, device=self.device)
            _cabi.check(self._lib.opus_llama_decode_step(self._ctx, tok.data_ptr(), logits.data_ptr(), s))
        self._leave()
        return logits

    def last_hidden(self, B: int, T: int) -> torch.Tensor:
        s = self._enter()
        with torch.cuda.stream(self._stream):
            h = torch.empty((B, T, self.cfg.enc_dim), dtype=torch.float32, device=self.device)
            _cabi.check(self._lib.opus_esm2_last_hidden(self._ctx, h.data_ptr(), B, T, s))
        self._leave()
        return h

    # ------------------------------------------------------------------ measurement
    def timing(self, on: bool):
        _cabi.check(self._lib.opus_timing_enable(self._ctx, 1 if on else 0))
        _cabi.check(self._lib.opus_timing_reset(self._ctx))

    def timing_get(self, klass: str = "*", phase: str = "*"):
        """(ms, launches, algorithmic bytes, algorithmic flops) of the recorded launches matching class and phase."""
        ms, n, by, fl = C.c_double(0), C.c_int64(0), C.c_double(0), C.c_double(0)
        _cabi.check(self._lib.opus_timing_get(self._ctx, klass.encode(), phase.encode(), C.byref(ms), C.byref(n), C.byref(by),
                                              C.byref(fl)))
        return ms.value, n.value, by.value, fl.value

    def timing_names(self):
        buf = C.create_string_buffer(512)
        _cabi.check(self._lib.opus_timing_names(buf, 512))
        classes, phases = buf.value.decode().split(";")
        return classes.split(","), phases.split(",")

    def drop_decode_graphs(self) -> None:
        """Forget the captured decode steps of this context (measurement aid: what a capture + instantiation per batch costs)."""
        _cabi.check(self._lib.opus_debug_knob(self._ctx, b"misc0", 0))

    def stat(self, name: str) -> int:
        """Counters of this context: "graph_instantiations" (decode-step hipGraphs instantiated: one per distinct batch size /
        token budget / sampling setting, none per prompt length), "graph_replays", "graphs_cached"."""
        v = int(self._lib.opus_stat(self._ctx, name.encode()))
        if v < 0:
            raise KeyError(name)
        return v

    def last_logits(self, B: int) -> torch.Tensor:
        """fp32 [B, V] logits of the most recent prefill / decode step (the optional logits gather of SURVEY 8e)."""
        s = self._enter()
        with torch.cuda.stream(self._stream):
            out = torch.empty((B, self.cfg.dec_vocab), dtype=torch.float32, device=self.device)
            _cabi.check(self._lib.opus_last_logits(self._ctx, out.data_ptr(), B, s))
        self._leave()
        return out


def _splice_labels(input_ids, attention_mask, labels, n_tok, T_out, inference_mode):
    """Label bookkeeping of opus_arch.py:172-233,255,266 (training-side only; host integer logic)."""
    ids = input_ids.cpu()
    lab = labels.cpu()
    m = torch.ones_like(ids, dtype=torch.bool) if attention_mask is None else attention_mask.cpu().bool()
    out = torch.full((ids.shape[0], T_out), IGNORE_INDEX, dtype=lab.dtype)
    for b in range(ids.shape[0]):
        row: List[int] = []
        for t in range(ids.shape[1]):
            if not m[b, t]:
                continue
            if int(ids[b, t]) == DEFAULT_SEQ_TOKEN_INDEX:
                row.extend([IGNORE_INDEX] * n_tok)
            else:
                row.append(int(lab[b, t]))
        row = row[:T_out]
        if row:
            if inference_mode:
                out[b, T_out - len(row):] = torch.tensor(row, dtype=lab.dtype)
            else:
                out[b, : len(row)] = torch.tensor(row, dtype=lab.dtype)
    return out
