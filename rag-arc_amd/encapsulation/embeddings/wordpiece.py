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
                 max_input_chars_per_word: int = 100, never_split: Optional[Iterable[str]] = None,
                 mask_token: str = "[MASK]"):
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
        # SPECIAL tokens are recognised anywhere in the RAW text, glued to other characters or not ("[CLS][SEP]", "[MASK]."),
        # case-sensitively and before any normalisation — the `tokenizers` library behind sentence-transformers cuts the
        # text at its added special tokens first and runs the BERT normaliser / pre-tokeniser on what lies between
        # ("[mask]" and MPNet's "<MASK>" are ordinary text).  `never_split`: further tokens to treat the same way.
        self.never_split = {t for t in set(never_split or ()) | {unk_token, cls_token, sep_token, pad_token, mask_token}
                            if t in vocab}
        self.specials = sorted(self.never_split, key=lambda t: (-len(t), t))
        import re

        self._special_re = re.compile("|".join(re.escape(t) for t in self.specials))

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

    @staticmethod
    def _split_punct(tok: str) -> List[str]:
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
            if self.do_lower_case:
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

    def _segments(self, text: str):
        """(is_special, piece) runs of the raw text: special tokens (longest match at the earliest position) and what lies
        between them."""
        out, pos = [], 0
        for m in self._special_re.finditer(text):
            if m.start() > pos:
                out.append((False, text[pos:m.start()]))
            out.append((True, m.group()))
            pos = m.end()
        if pos < len(text):
            out.append((False, text[pos:]))
        return out

    def tokenize(self, text: str) -> List[str]:
        out: List[str] = []
        for special, piece in self._segments(text):
            if special:
                out.append(piece)
                continue
            for tok in self.basic_tokens(piece):
                out.extend(self.wordpieces(tok))
        return out

    def __call__(self, text: str) -> List[int]:
        ids = [self.vocab.get(t, self.unk) for t in self.tokenize(text)]
        ids = ids[: max(0, self.max_length - 2)]
        return [self.cls] + ids + [self.sep]

    # -- batches: the native tokeniser of librarc_hip.so (csrc/tokenizer.hip) ---------------------------------------
    def _native_handle(self):
        """The library-side image of this vocabulary (created once; None if the vocabulary cannot be expressed as a
        vocab.txt image — a token with a line break in it)."""
        import ctypes

        from ...hip import binding as B

        if getattr(self, "_native", None) is None:
            by_id = sorted(self.vocab.items(), key=lambda kv: kv[1])
            if any("\n" in t for t, _ in by_id) or (by_id and (by_id[0][1] < 0 or by_id[-1][1] > 4 * len(by_id) + 1024)):
                self._native = False
                return None
            toks = ["\x00gap"] * (by_id[-1][1] + 1 if by_id else 0)     # (a NUL never survives clean-up: a gap matches nothing)
            for t, i in by_id:
                toks[i] = t
            blob = "\n".join(toks).encode("utf-8")
            ns = "\n".join(self.specials).encode("utf-8")
            lib = B.load_library()
            h = ctypes.c_void_p()
            pad = next((t for t, i in by_id if i == self.pad), "[PAD]")
            cls = next(t for t, i in by_id if i == self.cls)
            sep = next(t for t, i in by_id if i == self.sep)
            B.check(lib.rarc_wordpiece_create(blob, len(blob), 1 if self.do_lower_case else 0, self.unk_token.encode(),
                                              cls.encode(), sep.encode(), pad.encode(), ns, len(ns), self.max_chars,
                                              ctypes.byref(h)), "rarc_wordpiece_create")
            self._native = (lib, h)
        return self._native or None

    def __del__(self):
        nat = getattr(self, "_native", None)
        if nat:
            try:
                nat[0].rarc_wordpiece_destroy(nat[1])
            except Exception:  # noqa: BLE001 - interpreter shutdown
                pass

    def encode_batch(self, texts, max_length: Optional[int] = None, n_threads: Optional[int] = None):
        """`self(text)` for every text of a list, as arrays: (ids int32 [n][L] padded with the pad id, lens int32 [n]),
        L = the longest row.  ASCII texts — an English corpus — are tokenised by the library on `n_threads` cores
        (rarc_wordpiece_encode: the same algorithm; for ASCII its Unicode steps are identities); a text with any
        non-ASCII character goes through the python code above, so every row equals `self(text)` exactly."""
        import os

        import numpy as np

        from ...hip import binding as B

        texts = list(texts)
        n = len(texts)
        cap = int(min(max_length or self.max_length, self.max_length))
        ids = np.full((n, max(cap, 2)), self.pad, dtype=np.int32)
        lens = np.zeros(n, dtype=np.int32)
        nat = self._native_handle() if n else None
        todo = range(n)
        if nat is not None:
            ascii_idx = [i for i, t in enumerate(texts) if t.isascii()]
            if ascii_idx:
                sel = [texts[i] for i in ascii_idx]
                blob = "".join(sel).encode("ascii")
                offs = np.zeros(len(sel) + 1, dtype=np.int64)
                np.cumsum([len(t) for t in sel], out=offs[1:])
                sub_ids = ids if len(sel) == n else np.empty((len(sel), ids.shape[1]), dtype=np.int32)
                sub_lens = lens if len(sel) == n else np.empty(len(sel), dtype=np.int32)
                threads = int(n_threads or min(32, os.cpu_count() or 1))
                B.check(nat[0].rarc_wordpiece_encode(nat[1], blob, offs.ctypes.data, len(sel), ids.shape[1],
                                                     sub_ids.ctypes.data, ids.shape[1], sub_lens.ctypes.data, threads),
                        "rarc_wordpiece_encode")
                if len(sel) != n:
                    ids[ascii_idx], lens[ascii_idx] = sub_ids, sub_lens
                done = set(ascii_idx) if len(sel) != n else None
                todo = [i for i in range(n) if i not in done] if done is not None else []
        for i in todo:                                   # non-ASCII texts (and vocabularies the library cannot take)
            row = self(texts[i])[:cap] if cap >= 2 else []
            if len(row) == cap and cap >= 2:
                row[-1] = self.sep
            ids[i, : len(row)] = row
            lens[i] = len(row)
        longest = int(lens.max()) if n else 0
        return ids[:, : max(longest, 1)], lens
