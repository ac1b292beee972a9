"""Prompt assembly of the batch annotation driver (eval/run_opus_ddp.py:90-108).

Only what that driver reads from the reference's conversation presets is kept here: the v0 system text, the two
role names and the separator (multi_modality_v1/conversation.py:159-167); the full conversation class lives in
`conversation.py`.  Also the host-side pieces of the other two eval scripts: the multiple-choice
question template and answer scoring (eval_run_multichoice.py:76-83,171-206) and the interactive script's input check
and prompt (run_opus_online.py:12-14,40-58).
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import Tuple

from .constants import DEFAULT_SEQ_TOKEN


@dataclass(frozen=True)
class PromptPreset:
    system: str
    roles: Tuple[str, str]
    sep: str


conv_vicuna_v0 = PromptPreset(
    system="A chat between a curious student and a biological professor who is familiar with protein properties. "
           "The biological professor gives helpful, detailed, and professional answers to student's questions.",
    roles=("Student", "Professor"), sep="###")


def max_new_tokens_for(input_path: str) -> int:
    """Generation budget chosen by substring of the dataset path (run_opus_ddp.py:93-101)."""
    if "localization" in input_path:
        return 32
    if "keywords" in input_path:
        return 128
    return 256


def build_prompt(instruction: str, input_path: str = "", conv: PromptPreset = conv_vicuna_v0) -> str:
    """header + '### Student: <seq>\\n{instruction}\\n### Professor:' (run_opus_ddp.py:90-108)."""
    if DEFAULT_SEQ_TOKEN not in instruction:
        if "localization" in input_path:
            instruction = DEFAULT_SEQ_TOKEN + "\n" + instruction + "Kindly reply with only one word."
        else:
            instruction = DEFAULT_SEQ_TOKEN + "\n" + instruction
    return f"{conv.system}\n\n### {conv.roles[0]}: {instruction}\n### Professor:"


def after_process_output(outputs: str, conv: PromptPreset = conv_vicuna_v0) -> str:
    """Cut the decoded text at the first separator (run_opus_ddp.py:19-27)."""
    outputs = outputs.strip()
    idx = outputs.find(conv.sep)
    return (outputs if idx < 0 else outputs[:idx]).strip()


# ------------------------------------------------------------------------------------------------ multiple choice
import re as _re

_OPTION = _re.compile(r"\b([A-Da-d])[\s]*[).\.）\]】]|answer\s*:\s*([A-Da-d])|答案是\s*([A-Da-d])", _re.IGNORECASE)


def multichoice_prompt(question: str, options) -> str:
    """Question text of eval_run_multichoice.py:76-83 (options joined by newlines, :93-94); the indentation of the
    reference's triple-quoted literal is part of the prompt."""
    opts = options if isinstance(options, str) else "\n".join(options)
    return (f"Question: {question}\n\n        Options:\n        {opts}\n\n"
            "        Please carefully read the question and select the single correct answer from A-D.\n"
            "        You can only output one option from A), B), C), D) with format 'The correct answer is' without explanation.")


def extract_option_letter(text: str):
    """First `A)` / `b.` / `answer: c` / `答案是 D` style option in `text`, upper-cased; the text itself when there is
    none (eval_run_multichoice.py:176-187)."""
    m = _OPTION.search(text)
    if not m:
        return text
    letter = next((g for g in m.groups() if g is not None), None)
    return letter.upper() if letter else None


def score_multichoice(records):
    """records: [{'ground_truth', 'generated'}] -> (n_correct, per-option histogram) (eval_run_multichoice.py:189-210)."""
    hist = {"A": 0, "B": 0, "C": 0, "D": 0, "None": 0}
    correct = 0
    for r in records:
        got, want = extract_option_letter(r["generated"]), extract_option_letter(r["ground_truth"])
        correct += int(got == want)
        if got is not None and got in hist:
            hist[got] += 1
        else:
            hist["None"] += 1
    return correct, hist


# ------------------------------------------------------------------------------------------------ interactive script
_AMINO = frozenset("ACDEFGHIKLMNPQRSTVWY")


def is_protein_sequence(seq: str) -> bool:
    """Only the 20 standard residues, case-insensitive; the empty string passes (run_opus_online.py:12-14)."""
    return all(ch in _AMINO for ch in seq.upper())


def online_prompt(instruction: str, has_sequence: bool, conv=conv_vicuna_v0):
    """(prompt, instruction as shown) of run_opus_online.py:40-58: the <seq> placeholder is prepended only when a
    sequence is given and the instruction does not already carry it; the answer cue is always 'Professor:'."""
    if has_sequence and DEFAULT_SEQ_TOKEN not in instruction:
        instruction = DEFAULT_SEQ_TOKEN + "\n" + instruction
    return f"{conv.system}\n\n### {conv.roles[0]}: {instruction}\n### Professor:", instruction


def online_cut(outputs: str, sep: str = "###") -> str:
    """Text up to the first separator found from offset 2 (run_opus_online.py:86-92)."""
    outputs = outputs.strip()
    idx = outputs.find(sep, 2)
    return (outputs if idx < 0 else outputs[:idx]).strip()
