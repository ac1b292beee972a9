#!/usr/bin/env python3
"""Interactive annotation loop: the MI355X-native counterpart of eval/run_opus_online.py (one process, one GPU).

  python opus-pllm_amd/eval_online.py --model-base-path <hf dir | synthetic:c1_tiny> --opus-pllm-weights-path <adapter dir>

Per turn (run_opus_online.py:29-92): read an instruction and an optional protein sequence (re-asked until it is empty or
made of the 20 standard residues), build the v0 "Student / Professor" prompt, generate (text-only when no sequence is
given), cut the reply at the first '###', print.  `answer_once` is the loop body, importable for tests and services.
"""
from __future__ import annotations

import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import opus_pllm_amd as opa                                                    # noqa: E402
from opus_pllm_amd.builder import load_pretrained_model, return_cstp_path      # noqa: E402
from opus_pllm_amd.conversation import conv_vicuna_v0                          # noqa: E402
from opus_pllm_amd.prompt import is_protein_sequence, online_cut, online_prompt   # noqa: E402


def answer_once(model, tokenizer, instruction: str, seq: str, args, conv=conv_vicuna_v0):
    """-> (instruction as shown, sequence or None, reply text)."""
    dev = model.device
    prompt, shown = online_prompt(instruction, bool(seq), conv)
    if not seq:
        seq = None
        input_ids = torch.as_tensor(tokenizer([prompt]).input_ids).to(dev)
    else:
        input_ids = opa.tokenizer_seq_token(prompt, tokenizer, opa.DEFAULT_SEQ_TOKEN_INDEX, return_tensors="pt").unsqueeze(0).to(dev)
    with torch.inference_mode():
        out = model.generate(input_ids, seq, attention_mask=None, pad_token_id=tokenizer.eos_token_id, seq_embedding=None,
                             do_sample=args.temperature > 0, temperature=args.temperature, top_p=args.top_p,
                             num_beams=args.num_beams, max_new_tokens=args.max_new_tokens, use_cache=True)
    text = tokenizer.batch_decode(out, skip_special_tokens=True)[0]
    return shown, seq, online_cut(text, conv.sep)


def eval_model(args):
    model_name = opa.get_model_name_from_path(args.model_base_path)
    cstp_path = return_cstp_path(args.opus_pllm_weights_path, "modality_encoder/modality_encoding_adapter.ckpt")
    tokenizer, model, _ = load_pretrained_model(args.model_base_path, args.opus_pllm_weights_path, model_name,
                                                args.load_8bit, args.load_4bit, switch_projector_type=args.switch_projector_type,
                                                cstp_path=cstp_path, device="cuda:0", max_batch=max(1, args.num_beams),
                                                max_enc_tokens=args.max_residues + 2, max_prompt=args.max_prompt,
                                                max_new_tokens=max(args.max_new_tokens, 1))
    while True:
        try:
            instruction = input("Enter your instruction: ")
            while True:
                seq = input("Enter the protein sequence (or leave empty to skip): ").strip()
                if not seq or is_protein_sequence(seq):
                    print("Valid protein sequence:", seq)
                    break
                print("Invalid sequence!")
        except EOFError:
            return
        shown, seq, reply = answer_once(model, tokenizer, instruction, seq, args)
        print("----------------------------")
        print(f"Instruction: {shown}")
        print(f"Sequence: {seq}")
        print(f"Output: {reply}")
        print("----------------------------")


if __name__ == "__main__":
    p = argparse.ArgumentParser()
    p.add_argument("--model-base-path", type=str, default="synthetic:c1_tiny")
    p.add_argument("--opus-pllm-weights-path", type=str, default="synthetic")
    p.add_argument("--temperature", type=float, default=0.1)
    p.add_argument("--top_p", type=float, default=0.7)
    p.add_argument("--num_beams", type=int, default=1)
    p.add_argument("--max_new_tokens", type=int, default=32)         # run_opus_online.py:101
    p.add_argument("--switch_projector_type", type=str, default="mlp2x_gelu")
    p.add_argument("--load-4bit", action="store_true")
    p.add_argument("--load-8bit", action="store_true")
    p.add_argument("--max_residues", type=int, default=1024)
    p.add_argument("--max_prompt", type=int, default=256)
    eval_model(p.parse_args())
