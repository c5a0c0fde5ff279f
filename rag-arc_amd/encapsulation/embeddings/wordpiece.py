"""WordPiece tokenisation over a supplied `vocab.txt` (host side, pure python).

The reference hands raw TEXT to its provider (core/file_management/embeddings/huggingface.py:116-126:
`SentenceTransformer.encode(texts, ...)`), and what sits behind that call for the BERT family is the
BERT tokeniser: clean-up -> (lower-casing + accent stripping for uncased vocabularies) -> whitespace and
punctuation split, CJK characters on their own -> greedy longest-match-first WordPiece with "##"
continuations -> [CLS] ... [SEP], truncated to the model's maximum length.  This module restates that
algorithm so that `HipBertEmbeddings` takes texts like the reference's provider does; it is pinned token
for token against `transformers.BertTokenizer` on a synthetic vocabulary (tests/test_wordpiece.py).
"""
from __future__ import annotations

import unicodedata
from typing import Dict, Iterable, List, Optional


def _is_whitespace(ch: str) -> bool:
    return ch in " \t\n\r" or unicodedata.category(ch) == "Zs"


def _is_control(ch: str) -> bool:
    if ch in "\t\n\r":
        return False
    return unicodedata.category(ch).startswith("C")


def _is_punctuation(ch: str) -> bool:
    cp = ord(ch)
    if 33 <= cp <= 47 or 58 <= cp <= 64 or 91 <= cp <= 96 or 123 <= cp <= 126:
        return True
    return unicodedata.category(ch).startswith("P")


def _is_cjk(cp: int) -> bool:
    return (0x4E00 <= cp <= 0x9FFF or 0x3400 <= cp <= 0x4DBF or 0x20000 <= cp <= 0x2A6DF or 0x2A700 <= cp <= 0x2B73F
            or 0x2B740 <= cp <= 0x2B81F or 0x2B820 <= cp <= 0x2CEAF or 0xF900 <= cp <= 0xFAFF or 0x2F800 <= cp <= 0x2FA1F)


class WordPieceTokenizer:
    """`tokenizer(text) -> List[int]` including [CLS] / [SEP], at most `max_length` ids."""

    def __init__(self, vocab: Dict[str, int] | Iterable[str], do_lower_case: bool = True, max_length: int = 512,
                 unk_token: str = "[UNK]", cls_token: str = "[CLS]", sep_token: str = "[SEP]", pad_token: str = "[PAD]",
                 max_input_chars_per_word: int = 100, never_split: Optional[Iterable[str]] = None):
        if not isinstance(vocab, dict):
            vocab = {tok: i for i, tok in enumerate(vocab)}
        self.vocab = vocab
        for tok in (unk_token, cls_token, sep_token):
            if tok not in vocab:
                raise ValueError(f"vocabulary lacks the special token {tok}")
        self.do_lower_case = bool(do_lower_case)
        self.max_length = int(max_length)
        self.unk, self.cls, self.sep = vocab[unk_token], vocab[cls_token], vocab[sep_token]
        self.pad = vocab.get(pad_token, 0)
        self.unk_token = unk_token
        self.max_chars = int(max_input_chars_per_word)
        self.never_split = set(never_split or ()) | {unk_token, cls_token, sep_token, pad_token, "[MASK]"}

    @classmethod
    def from_file(cls, path: str, **kwargs) -> "WordPieceTokenizer":
        """vocab.txt: one token per line, id = line number (the format BERT checkpoints ship)."""
        vocab: Dict[str, int] = {}
        with open(path, "r", encoding="utf-8") as fh:
            for i, line in enumerate(fh):
                vocab[line.rstrip("\n")] = i
        return cls(vocab, **kwargs)

    # -- basic tokenisation -----------------------------------------------------------------------
    def _clean(self, text: str) -> str:
        out = []
        for ch in text:
            cp = ord(ch)
            if cp == 0 or cp == 0xFFFD or _is_control(ch):
                continue
            out.append(" " if _is_whitespace(ch) else ch)
        return "".join(out)

    @staticmethod
    def _space_cjk(text: str) -> str:
        out = []
        for ch in text:
            if _is_cjk(ord(ch)):
                out.extend((" ", ch, " "))
            else:
                out.append(ch)
        return "".join(out)

    @staticmethod
    def _strip_accents(text: str) -> str:
        return "".join(ch for ch in unicodedata.normalize("NFD", text) if unicodedata.category(ch) != "Mn")

    def _split_punct(self, tok: str) -> List[str]:
        if tok in self.never_split:
            return [tok]
        out: List[List[str]] = []
        start = True
        for ch in tok:
            if _is_punctuation(ch):
                out.append([ch])
                start = True
            else:
                if start:
                    out.append([])
                start = False
                out[-1].append(ch)
        return ["".join(x) for x in out]

    def basic_tokens(self, text: str) -> List[str]:
        text = self._space_cjk(self._clean(text))
        text = unicodedata.normalize("NFC", text)
        out: List[str] = []
        for tok in text.strip().split():
            if tok not in self.never_split and self.do_lower_case:
                tok = self._strip_accents(tok.lower())
            out.extend(self._split_punct(tok))
        return " ".join(out).strip().split()

    # -- wordpiece --------------------------------------------------------------------------------
    def wordpieces(self, token: str) -> List[str]:
        if len(token) > self.max_chars:
            return [self.unk_token]
        pieces: List[str] = []
        start = 0
        while start < len(token):
            end = len(token)
            cur = None
            while start < end:
                sub = token[start:end]
                if start > 0:
                    sub = "##" + sub
                if sub in self.vocab:
                    cur = sub
                    break
                end -= 1
            if cur is None:
                return [self.unk_token]
            pieces.append(cur)
            start = end
        return pieces

    def tokenize(self, text: str) -> List[str]:
        out: List[str] = []
        for tok in self.basic_tokens(text):
            out.extend([tok] if tok in self.never_split else self.wordpieces(tok))
        return out

    def __call__(self, text: str) -> List[int]:
        ids = [self.vocab.get(t, self.unk) for t in self.tokenize(text)]
        ids = ids[: max(0, self.max_length - 2)]
        return [self.cls] + ids + [self.sep]
