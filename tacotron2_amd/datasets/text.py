"""Text -> ids exactly as the reference (datasets/tts_dataset.py:19-47,137-163,218-229; run/say.py:46-60):
ASCII-fold -> lower -> delete everything outside `allowed_chars` -> (dataset only) 18 abbreviation expansions ->
append the end token -> sklearn OrdinalEncoder ids (categories sorted by code point) + 1, id 0 = padding.
`unidecode` is not available here, so the fold is NFKD + dropping non-ASCII marks (identical for Latin text)."""
from __future__ import annotations

import re
import unicodedata
from typing import List, Optional

ALLOWED_CHARS = "!'(),.:;? \\-ABCDEFGHIJKLMNOPQRSTUVWXYZabcdefghijklmnopqrstuvwxyz"

_ABBREVIATIONS = [(re.compile("\\b%s\\." % a, re.IGNORECASE), b) for a, b in [
    ("mrs", "misess"), ("mr", "mister"), ("dr", "doctor"), ("st", "saint"), ("co", "company"), ("jr", "junior"),
    ("maj", "major"), ("gen", "general"), ("drs", "doctors"), ("rev", "reverend"), ("lt", "lieutenant"),
    ("hon", "honorable"), ("sgt", "sergeant"), ("capt", "captain"), ("esq", "esquire"), ("ltd", "limited"),
    ("col", "colonel"), ("ft", "fort")]]


def ascii_fold(text: str) -> str:
    return unicodedata.normalize("NFKD", text).encode("ascii", "ignore").decode("ascii")


def expand_abbreviations(text: str) -> str:
    for rx, rep in _ABBREVIATIONS:
        text = re.sub(rx, rep, text)
    return text


class TextEncoder:
    def __init__(self, allowed_chars: str = ALLOWED_CHARS, end_token: Optional[str] = "^", expand_abbrev: bool = False):
        if end_token is not None and end_token in allowed_chars:
            raise Exception("end_token cannot be in allowed_chars!")
        self.allowed_re = re.compile(f"[^{allowed_chars}]+")
        self.end_token, self.expand_abbrev = end_token, expand_abbrev
        cats = sorted(set(list(allowed_chars) + ([end_token] if end_token is not None else [])))
        self.table = {c: i + 1 for i, c in enumerate(cats)}
        self.num_chars = len(allowed_chars) + (end_token is not None)    # run/train.py:218-219 (counts duplicates as given)

    def clean(self, text: str) -> str:
        t = self.allowed_re.sub("", ascii_fold(text).lower())
        if self.expand_abbrev:
            t = expand_abbreviations(t)
        return t + (self.end_token or "")

    def encode(self, text: str) -> List[int]:
        return [self.table[c] for c in self.clean(text)]
