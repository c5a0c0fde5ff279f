"""Byte-level BPE tokeniser for the reranker's prompts (host side).

The reference loads `AutoTokenizer.from_pretrained(model, padding_side='left')` (core/rerank/Reranker_Qwen3.py:11), which
for the Qwen3 rerankers is transformers' Qwen2 tokeniser: NFC normalisation, added special tokens (`<|im_start|>`,
`<|im_end|>`, `<think>` ...) cut out first, a regular-expression pre-tokeniser, every piece mapped byte by byte to
printable characters (the GPT-2 byte alphabet) and merged by rank with the pairs of `merges.txt`, ids from
`vocab.json`.  This module restates that published algorithm over the model's own files — `vocab.json` +
`merges.txt`, or the single `tokenizer.json` — so that `HipQwen3Reranker` takes texts like the reference does; the
vocabulary itself does not ship here (no network).  Pinned in tests/test_bpe.py against transformers' Qwen2Tokenizer and
the `tokenizers` library on a vocabulary trained offline.
"""
from __future__ import annotations

import json
import unicodedata
from functools import lru_cache
from typing import Dict, Iterable, List, Optional, Sequence, Tuple, Union

import regex

# the pre-tokenisation pattern Qwen2 / Qwen3 tokenisers declare (tokenizer.json: pre_tokenizer.pretokenizers[0].pattern)
QWEN_PATTERN = (r"(?i:'s|'t|'re|'ve|'m|'ll|'d)|[^\r\n\p{L}\p{N}]?\p{L}+|\p{N}| ?[^\s\p{L}\p{N}]+[\r\n]*|\s*[\r\n]+|\s+(?!\S)|\s+")


@lru_cache(maxsize=1)
def byte_alphabet() -> Dict[int, str]:
    """byte -> printable character: bytes that are printable in Latin-1 keep their code point, the other 68 take the
    code points 256, 257, ... in byte order (the GPT-2 alphabet every byte-level BPE vocabulary is written in)."""
    keep = list(range(ord("!"), ord("~") + 1)) + list(range(0xA1, 0xAD)) + list(range(0xAE, 0x100))
    table, nxt = {}, 0
    for b in range(256):
        if b in keep:
            table[b] = chr(b)
        else:
            table[b] = chr(256 + nxt)
            nxt += 1
    return table


class ByteLevelBPETokenizer:
    """encode(text) -> ids, decode(ids) -> text; calling the object encodes (the `tokenize` callable of the reranker)."""

    def __init__(self, vocab: Dict[str, int], merges: Iterable[Union[str, Sequence[str]]],
                 special_tokens: Optional[Dict[str, int]] = None, pattern: str = QWEN_PATTERN, normalize: Optional[str] = "NFC",
                 unk_token: Optional[str] = None):
        self.vocab = dict(vocab)
        self.ranks: Dict[Tuple[str, str], int] = {}
        for line in merges:
            if isinstance(line, str) and (not line.strip() or line.startswith("#version")):
                continue                                       # header ("#version: 0.2" splits into two fields!) / blank line
            pair = tuple(line.split(" ")) if isinstance(line, str) else tuple(line)
            if len(pair) != 2:
                raise ValueError(f"merge rule {line!r} is not a pair")
            self.ranks.setdefault(pair, len(self.ranks))
        self.special = dict(special_tokens or {})
        self.pattern = regex.compile(pattern)
        self.normalize = normalize
        self.unk_id = self.vocab.get(unk_token) if unk_token is not None else None
        self._bytes = byte_alphabet()
        self._unbytes = {c: b for b, c in self._bytes.items()}
        self._id_to_token = {i: t for t, i in self.vocab.items()}
        self._id_to_token.update({i: t for t, i in self.special.items()})
        # longest first, so that a special token that is a prefix of another never shadows it
        self._special_re = (regex.compile("|".join(regex.escape(t) for t in sorted(self.special, key=len, reverse=True)))
                            if self.special else None)
        self._cache: Dict[str, List[int]] = {}

    # ---- construction from the model's files ----
    @classmethod
    def from_files(cls, vocab_json: str, merges_txt: str, special_tokens: Optional[Dict[str, int]] = None, **kw):
        with open(vocab_json, encoding="utf-8") as fh:
            vocab = json.load(fh)
        with open(merges_txt, encoding="utf-8") as fh:
            merges = [ln.rstrip("\n") for ln in fh]
        return cls(vocab, merges, special_tokens, **kw)

    @classmethod
    def from_tokenizer_json(cls, path: str, **kw):
        """The single-file form (`tokenizer.json`): model.vocab, model.merges, added_tokens."""
        with open(path, encoding="utf-8") as fh:
            spec = json.load(fh)
        model = spec["model"]
        if model.get("type") != "BPE":
            raise ValueError(f"{path}: model type {model.get('type')!r}, expected BPE")
        special = {t["content"]: int(t["id"]) for t in spec.get("added_tokens", [])}
        # the file's own split pattern and normaliser, where it declares them in the Qwen2 form (Sequence[Split(Regex), ByteLevel])
        pre = spec.get("pre_tokenizer") or {}
        for step in pre.get("pretokenizers", [pre]):
            rx = (step.get("pattern") or {}).get("Regex") if step.get("type") == "Split" else None
            if rx:
                kw.setdefault("pattern", rx)
        norm = spec.get("normalizer")
        kw.setdefault("normalize", norm.get("type") if norm and norm.get("type") in ("NFC", "NFD", "NFKC", "NFKD") else None)
        return cls(model["vocab"], model["merges"], special, **kw)

    # ---- BPE ----
    def _merge_piece(self, piece: str) -> List[int]:
        cached = self._cache.get(piece)
        if cached is not None:
            return cached
        word = [self._bytes[b] for b in piece.encode("utf-8")]
        while len(word) > 1:
            best, at = None, -1
            for i in range(len(word) - 1):
                r = self.ranks.get((word[i], word[i + 1]))
                if r is not None and (best is None or r < best):
                    best, at = r, i
            if best is None:
                break
            first, second = word[at], word[at + 1]
            merged, i = [], 0
            while i < len(word):                       # every occurrence of the best pair, left to right
                if i + 1 < len(word) and word[i] == first and word[i + 1] == second:
                    merged.append(first + second)
                    i += 2
                else:
                    merged.append(word[i])
                    i += 1
            word = merged
        ids = []
        for tok in word:
            tid = self.vocab.get(tok, self.unk_id)
            if tid is None:
                raise KeyError(f"token {tok!r} is not in the vocabulary (and no unk token is set)")
            ids.append(tid)
        if len(self._cache) < 200_000:
            self._cache[piece] = ids
        return ids

    def _encode_plain(self, text: str) -> List[int]:
        if self.normalize:
            text = unicodedata.normalize(self.normalize, text)
        out: List[int] = []
        for piece in self.pattern.findall(text):
            out.extend(self._merge_piece(piece))
        return out

    def encode(self, text: str) -> List[int]:
        """ids of `text`; no BOS / EOS is added (the reference encodes with add_special_tokens=False, and Qwen's
        tokeniser adds none anyway)."""
        if self._special_re is None:
            return self._encode_plain(text)
        out: List[int] = []
        pos = 0
        for m in self._special_re.finditer(text):
            if m.start() > pos:
                out.extend(self._encode_plain(text[pos:m.start()]))
            out.append(self.special[m.group()])
            pos = m.end()
        if pos < len(text):
            out.extend(self._encode_plain(text[pos:]))
        return out

    __call__ = encode

    def convert_tokens_to_ids(self, token: str) -> Optional[int]:
        return self.special.get(token, self.vocab.get(token, self.unk_id))

    def decode(self, ids: Iterable[int]) -> str:
        data = bytearray()
        for i in ids:
            tok = self._id_to_token[int(i)]
            if tok in self.special:
                data.extend(tok.encode("utf-8"))
            else:
                data.extend(self._unbytes[c] for c in tok)
        return data.decode("utf-8", errors="replace")
