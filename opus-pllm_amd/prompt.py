"""Prompt assembly of the batch annotation driver (eval/run_opus_ddp.py:90-108).

Only what that driver reads from the reference's conversation presets is kept: the v0 system text, the two
role names and the separator (multi_modality_v1/conversation.py:159-167).  Chat-template front-ends are row N2
of SURVEY 8f.
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
