"""Sequence constants of the drop-in API (reference: multi_modality_v1/constants.py:7-9)."""
IGNORE_INDEX = -100
DEFAULT_SEQ_TOKEN_INDEX = -200
DEFAULT_SEQ_TOKEN = "<seq>"
