"""Byte-level BPE tokeniser of the reranker (rag_arc_amd.core.rerank.bpe) against the implementations the reference
loads through AutoTokenizer (core/rerank/Reranker_Qwen3.py:11): transformers' Qwen2Tokenizer and the `tokenizers`
library, configured the way Qwen's tokenizer.json declares (NFC, regex split, byte-level alphabet, added special
tokens), on a vocabulary trained offline inside the test (Qwen's own vocabulary does not ship here)."""
import json
import unicodedata

import pytest

from rag_arc_amd.core.rerank.bpe import QWEN_PATTERN, ByteLevelBPETokenizer, byte_alphabet

CORPUS = [
    "Judge whether the Document meets the requirements based on the Query and the Instruct provided.",
    "Note that the answer can only be \"yes\" or \"no\".",
    "<Instruct>: Given the user query, retrieval the relevant passages",
    "<Query>: what's a vector index? <Document>: an index over 1234 vectors\n\nnew line  two spaces",
    "naïve café 北京 مرحبا 🙂 don't I'll we've they'd it's I'm you're",
    "def f(x):\n    return x ** 2  # comment\n\n\nclass A:\tpass\r\n",
] * 3
SPECIALS = ["<|endoftext|>", "<|im_start|>", "<|im_end|>", "<think>", "</think>"]
TEXTS = [
    "hello world", "<|im_start|>system\nJudge whether the Document<|im_end|>\n<|im_start|>user\n", "  leading", "a\n\n\nb",
    "x  y   z ", "don't DON'T I'LL we'Ve", "1234567 3.14 1e-5", "naïve café 北京 🙂 مرحبا", "", " ", "\n", "   ",
    "<think>\n\n</think>\n\n", "tabs\tand\r\nCRLF\r\n\r\n", "yes", "no", "e\u0301 vs \u00e9",             # NFC: both spell é
    "<|im_end|><|im_end|>text<|endoftext|>", "<|im_start", "unknown bytes: \x00\x7f\u200b\ufeff", "ﬁ ligature Ⅻ",
    "<Instruct>: Given the user query, retrieval the relevant passages\n<Query>: q?\n<Document>: d!",
    "A" * 300 + " " + "ab" * 200, "punct!!! ... ---\n\n  \n x",
]


@pytest.fixture(scope="module")
def trained():
    from tokenizers import AddedToken, Regex, Tokenizer, decoders, models, normalizers, pre_tokenizers, trainers

    tok = Tokenizer(models.BPE())
    tok.normalizer = normalizers.NFC()
    tok.pre_tokenizer = pre_tokenizers.Sequence([
        pre_tokenizers.Split(Regex(QWEN_PATTERN), behavior="isolated", invert=False),
        pre_tokenizers.ByteLevel(add_prefix_space=False, use_regex=False)])
    tok.decoder = decoders.ByteLevel()
    tok.train_from_iterator(CORPUS, trainers.BpeTrainer(vocab_size=700, initial_alphabet=pre_tokenizers.ByteLevel.alphabet(),
                                                        special_tokens=[], show_progress=False))
    tok.add_special_tokens([AddedToken(t, special=True, normalized=False) for t in SPECIALS])
    spec = json.loads(tok.to_str())
    special = {t: tok.token_to_id(t) for t in SPECIALS}
    return tok, spec["model"]["vocab"], [tuple(m) for m in spec["model"]["merges"]], special, spec


def test_byte_alphabet_is_the_byte_level_one():
    from tokenizers import pre_tokenizers

    table = byte_alphabet()
    assert len(table) == 256 and len(set(table.values())) == 256
    assert sorted(table.values()) == sorted(pre_tokenizers.ByteLevel.alphabet())
    assert table[ord("A")] == "A" and table[ord(" ")] == "\u0120" and table[ord("\n")] == "\u010a"


def test_ids_equal_the_tokenizers_library(trained):
    tok, vocab, merges, special, _ = trained
    mine = ByteLevelBPETokenizer(vocab, merges, special)
    for t in TEXTS:
        want = tok.encode(t, add_special_tokens=False).ids
        assert mine.encode(t) == want and mine(t) == want, t
        assert mine.decode(want) == tok.decode(want, skip_special_tokens=False) == unicodedata.normalize("NFC", t)
    assert mine.convert_tokens_to_ids("yes") == tok.token_to_id("yes") is not None
    assert mine.convert_tokens_to_ids("<|im_end|>") == special["<|im_end|>"]


def test_ids_equal_transformers_qwen2_tokenizer(trained):
    """The class AutoTokenizer resolves to for the Qwen3 rerankers, built over the same vocabulary and merges."""
    from transformers import Qwen2Tokenizer

    _, vocab, merges, special, _ = trained
    hf = Qwen2Tokenizer(vocab=dict(vocab), merges=[tuple(m) for m in merges])
    hf.add_special_tokens({"additional_special_tokens": [t for t in SPECIALS if t != "<|endoftext|>"]})
    hf_special = {t: hf.convert_tokens_to_ids(t) for t in SPECIALS}
    mine = ByteLevelBPETokenizer(vocab, merges, hf_special)
    for t in TEXTS:
        assert mine.encode(t) == hf.encode(t, add_special_tokens=False), t
    # the reference's own use: prefix / suffix token lists and the yes / no ids (Reranker_Qwen3.py:14-19)
    prefix = ("<|im_start|>system\nJudge whether the Document meets the requirements based on the Query and the Instruct "
              "provided. Note that the answer can only be \"yes\" or \"no\".<|im_end|>\n<|im_start|>user\n")
    suffix = "<|im_end|>\n<|im_start|>assistant\n<think>\n\n</think>\n\n"
    assert mine.encode(prefix) == hf.encode(prefix, add_special_tokens=False)
    assert mine.encode(suffix) == hf.encode(suffix, add_special_tokens=False)
    assert mine.convert_tokens_to_ids("yes") == hf.convert_tokens_to_ids("yes")
    assert mine.convert_tokens_to_ids("no") == hf.convert_tokens_to_ids("no")


def test_file_forms(trained, tmp_path):
    tok, vocab, merges, special, spec = trained
    (tmp_path / "vocab.json").write_text(json.dumps(vocab), encoding="utf-8")
    (tmp_path / "merges.txt").write_text("#version: 0.2\n" + "\n".join(" ".join(m) for m in merges) + "\n", encoding="utf-8")
    (tmp_path / "tokenizer.json").write_text(json.dumps(spec), encoding="utf-8")
    a = ByteLevelBPETokenizer.from_files(str(tmp_path / "vocab.json"), str(tmp_path / "merges.txt"), special)
    b = ByteLevelBPETokenizer.from_tokenizer_json(str(tmp_path / "tokenizer.json"))
    for t in TEXTS:
        want = tok.encode(t, add_special_tokens=False).ids
        assert a.encode(t) == want and b.encode(t) == want
    with pytest.raises(ValueError):
        ByteLevelBPETokenizer(vocab, ["a b c"])
    with pytest.raises(KeyError):
        ByteLevelBPETokenizer({"a": 0}, []).encode("b")
    assert ByteLevelBPETokenizer({"a": 0, "?": 1}, [], unk_token="?").encode("ab") == [0, 1]
