"""Prompt / chat front-end of the eval scripts.

Mirrors the call surface of the reference's `multi_modality_v1/conversation.py` (`Conversation`, `SeparatorStyle`,
the `conv_vicuna_v0..v3` presets, `get_prompt`, `get_prompt_eval`, `append_message`, `copy`, `dict`) so that the eval
drivers read the same way:

* a conversation is a system text, two role names and a list of `{'role', 'content'}` messages;
* `get_prompt()` renders with the tokenizer's chat template when the attached tokenizer has one
  (`conversation.py:99-104`), otherwise by separator style (`:35-98`): SINGLE (`system + sep`, then `role: content sep`,
  an empty message leaves a bare `role:` cue), TWO (alternating `sep` / `sep2`), MPT (`role content sep`), LLAMA_2
  (`[INST] <<SYS>>..` wrapping; the first turn must come from `roles[0]`), PLAIN (contents only); LLAMA_3 / Qwen_2
  have no separator renderer in the reference either and raise;
* `get_prompt_eval()` is the chat template with the generation cue appended (`:105-112`), and needs a tokenizer;
* `default_chat_template` is the ChatML fallback the multi-choice driver installs on a tokenizer that ships without
  one (`eval_run_multichoice.py:61-74`).

Golden renderings produced by the reference module are in `tests/golden/conversation.json` (`tools/gen_golden.py`).
"""
from __future__ import annotations

import dataclasses
from enum import Enum, auto
from typing import Any, Callable, Dict, List, Optional


class SeparatorStyle(Enum):
    SINGLE = auto()
    TWO = auto()
    MPT = auto()
    PLAIN = auto()
    LLAMA_2 = auto()
    LLAMA_3 = auto()
    Qwen_2 = auto()


def _render_single(c: "Conversation") -> str:
    out = [c.system, c.sep]
    for m in c.messages:
        out.append(f"{m['role']}: {m['content']}{c.sep}" if m["content"] else f"{m['role']}:")
    return "".join(out)


def _render_two(c: "Conversation") -> str:
    seps = (c.sep, c.sep2)
    out = [c.system, seps[0]]
    for i, m in enumerate(c.messages):
        out.append(f"{m['role']}: {m['content']}{seps[i % 2]}" if m["content"] else f"{m['role']}:")
    return "".join(out)


def _render_mpt(c: "Conversation") -> str:
    out = [c.system, c.sep]
    for m in c.messages:
        out.append(f"{m['role']}{m['content']}{c.sep}" if m["content"] else m["role"])
    return "".join(out)


def _render_llama2(c: "Conversation") -> str:
    out = ""
    for i, m in enumerate(c.messages):
        text = m["content"]
        if i == 0:
            if not text:
                raise AssertionError("first message should not be none")
            if m["role"] != c.roles[0]:
                raise AssertionError("first message should come from user")
        if not text:
            continue
        if i == 0 and c.system:
            text = f"<<SYS>>\n{c.system}\n<</SYS>>\n\n{text}"
        out += f"{c.sep}[INST] {text} [/INST]" if i % 2 == 0 else f" {text} {c.sep2}"
    return out.lstrip(c.sep)


def _render_plain(c: "Conversation") -> str:
    seps = (c.sep, c.sep2)
    return c.system + "".join(m["content"] + seps[i % 2] for i, m in enumerate(c.messages) if m["content"])


def _unrendered(c: "Conversation") -> str:
    raise NotImplementedError(f"{c.sep_style.name} prompts are rendered by the tokenizer's chat template only")


_RENDERERS: Dict[SeparatorStyle, Callable[["Conversation"], str]] = {
    SeparatorStyle.SINGLE: _render_single, SeparatorStyle.TWO: _render_two, SeparatorStyle.MPT: _render_mpt,
    SeparatorStyle.LLAMA_2: _render_llama2, SeparatorStyle.PLAIN: _render_plain,
    SeparatorStyle.LLAMA_3: _unrendered, SeparatorStyle.Qwen_2: _unrendered,
}


@dataclasses.dataclass
class Conversation:
    system: str
    roles: List[str]
    messages: List[Dict[str, str]]
    offset: int
    sep_style: SeparatorStyle = SeparatorStyle.SINGLE
    sep: str = "### "
    sep2: Optional[str] = None
    version: str = "Unknown"
    skip_next: bool = False
    tokenizer: Any = None

    def _templated(self, generation_cue: bool) -> str:
        return self.tokenizer.apply_chat_template(self.messages, tokenize=False, add_generation_prompt=generation_cue)

    def get_prompt(self) -> str:
        if self.tokenizer is not None and hasattr(self.tokenizer, "apply_chat_template"):
            return self._templated(False)
        try:
            render = _RENDERERS[self.sep_style]
        except KeyError:
            raise ValueError(f"Invalid style: {self.sep_style}") from None
        return render(self)

    def get_prompt_eval(self) -> str:
        if self.tokenizer is None:
            raise NotImplementedError("get_prompt_eval needs a tokenizer with a chat template")
        return self._templated(True)

    def append_message(self, role: str, message: Optional[str]) -> None:
        self.messages.append({"role": role, "content": message})

    def copy(self) -> "Conversation":
        return Conversation(system=self.system, roles=self.roles, messages=[dict(m) for m in self.messages],
                            offset=self.offset, sep_style=self.sep_style, sep=self.sep, sep2=self.sep2,
                            version=self.version, tokenizer=self.tokenizer)

    def dict(self) -> Dict[str, Any]:
        return {"system": self.system, "roles": self.roles, "messages": self.messages, "offset": self.offset,
                "sep": self.sep, "sep2": self.sep2}


# ChatML: one `<|im_start|>{role}\n{content}<|im_end|>\n` block per system / user / assistant message, then the cue.
# The whitespace is part of the reference's template text and is kept as is.
default_chat_template = """
{% for message in messages %}
    {% if message['role'] == 'system' %}
        <|im_start|>system\n{{ message['content'] }}<|im_end|>\n
    {% elif message['role'] == 'user' %}
        <|im_start|>user\n{{ message['content'] }}<|im_end|>\n
    {% elif message['role'] == 'assistant' %}
        <|im_start|>assistant\n{{ message['content'] }}<|im_end|>\n
    {% endif %}
{% endfor %}
{% if add_generation_prompt %}<|im_start|>assistant\n{% endif %}
"""

_PROFESSOR = ("A chat between a curious student and a biological professor who is familiar with protein properties. "
              "The biological professor gives helpful, detailed, and professional answers to student's questions.")

conv_vicuna_v0 = Conversation(system=_PROFESSOR, roles=["Student", "Professor"], messages=[], offset=2,
                              sep_style=SeparatorStyle.SINGLE, sep="###")
conv_vicuna_v1 = Conversation(
    system="You are an automated protein annotation system that provides precise, database-validated identifiers in "
           "required formats. Responses are strictly concise and correct.",
    roles=["Student", "Professor"], messages=[], offset=2, sep_style=SeparatorStyle.SINGLE, sep="###")
conv_vicuna_v2 = Conversation(
    system="A chat between a curious user and an artificial intelligence assistant. "
           "The assistant gives helpful, detailed, and polite answers to the user's questions.",
    roles=["USER", "ASSISTANT"], version="v1", messages=[], offset=0, sep_style=SeparatorStyle.TWO, sep=" ", sep2="</s>")
conv_vicuna_v3 = Conversation(
    system="A chat between a curious user and a biological assistant who is familiar with protein properties. "
           "The biological assistant gives helpful, detailed, and professional answers to user's questions.",
    roles=["user", "assistant"], messages=[], offset=2, sep_style=SeparatorStyle.SINGLE, sep="###")
