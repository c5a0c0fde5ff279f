"""The host WordPiece tokeniser against transformers.BertTokenizer (backed by the `tokenizers` library's BERT
normaliser / pre-tokeniser / WordPiece model — what sentence-transformers runs behind
core/file_management/embeddings/huggingface.py:122-126),
on a synthetic vocabulary: same token strings, same ids, same truncation."""
import os

import pytest

from rag_arc_amd.encapsulation.embeddings.wordpiece import WordPieceTokenizer

VOCAB = ["[PAD]", "[UNK]", "[CLS]", "[SEP]", "[MASK]", "the", "quick", "brown", "fox", "jump", "##s", "##ed", "##ing", "over",
         "lazy", "dog", "un", "##aff", "##able", ",", ".", "!", "?", "-", "'", "a", "b", "c", "##c", "##b", "re", "##rank", "##er",
         "embed", "##ding", "中", "文", "naive", "cafe", "resume", "hello", "world", "2", "0", "##2", "##5", "##0", "$", "#", "@",
         "x", "##x", "e", "##e", "rag", "arc", "mi", "##35", "##5x", "(", ")", "\"", "é"]

TEXTS = [
    "The quick brown fox jumps over the lazy dog.",
    "  unaffable,   reranker!  embedding-embedding ",
    "Hello, WORLD?! (hello)  \"world\"",
    "naïve café résumé",
    "中文 and 中a文",
    "tabs\tand\nnewlines\r\n and \x00 control \x7f chars",
    "2025 $20 #rag @arc mi355x",
    "unknownword xx xxxx e" + "e" * 120,
    "",
    "   ",
    "don't re-rank",
    "[CLS] the [SEP] [MASK] [UNK]",
]


@pytest.fixture(scope="module")
def pair(tmp_path_factory):
    path = tmp_path_factory.mktemp("vocab") / "vocab.txt"
    path.write_text("\n".join(VOCAB) + "\n", encoding="utf-8")
    transformers = pytest.importorskip("transformers")
    ref = transformers.BertTokenizer(vocab={t: i for i, t in enumerate(VOCAB)}, do_lower_case=True)
    return WordPieceTokenizer.from_file(str(path), do_lower_case=True, max_length=512), ref, str(path)


def test_same_tokens_and_ids_as_transformers(pair):
    mine, ref, _ = pair
    for text in TEXTS:
        assert mine.tokenize(text) == ref.tokenize(text), text
        assert mine(text) == ref.encode(text, add_special_tokens=True), text


def test_truncation_and_cased_mode(pair):
    mine, ref, path = pair
    long = "the quick brown fox " * 300
    short = WordPieceTokenizer.from_file(path, max_length=16)
    assert short(long) == ref.encode(long, add_special_tokens=True, truncation=True, max_length=16)
    transformers = pytest.importorskip("transformers")
    cased_ref = transformers.BertTokenizer(vocab={t: i for i, t in enumerate(VOCAB)}, do_lower_case=False)
    cased = WordPieceTokenizer.from_file(path, do_lower_case=False)
    for text in ("The quick é café", "Hello hello"):
        assert cased(text) == cased_ref.encode(text, add_special_tokens=True)


def test_missing_special_token_is_an_error():
    with pytest.raises(ValueError):
        WordPieceTokenizer(["a", "b"])
