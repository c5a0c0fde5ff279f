"""The library's batch WordPiece tokeniser (rarc_wordpiece_encode, csrc/tokenizer.hip: host code, runs without a GPU)
against the python restatement it mirrors (wordpiece.WordPieceTokenizer, itself pinned to transformers.BertTokenizer in
tests/test_wordpiece.py) AND against transformers.BertTokenizer directly: same ids, same lengths, same truncation — on
fixed awkward texts, on thousands of random ASCII strings (every printable and control character), with the MPNet
family's specials, cased and uncased; texts that are not ASCII are left to the python code and still come back right."""
import numpy as np
import pytest
from hypothesis import given, settings
from hypothesis import strategies as st

from rag_arc_amd.encapsulation.embeddings.wordpiece import WordPieceTokenizer
from tests.test_wordpiece import TEXTS, VOCAB

EXTRA = ["foo", "##bar", "i", "##i", "t", "##t", "s", "##s", "[", "]", "<", ">", "/", "_",
         "##_", "1", "##1", "co", "##op", "z" * 30, "##" + "y" * 40]
V2 = VOCAB + [t for t in EXTRA if t not in VOCAB]


def _rows(tok, texts, **kw):
    ids, lens = tok.encode_batch(texts, **kw)
    assert ids.dtype == np.int32 and lens.dtype == np.int32 and ids.shape[0] == len(texts)
    out = []
    for r, n in zip(ids, lens):
        assert (r[n:] == tok.pad).all()
        out.append(r[:n].tolist())
    return out


@pytest.fixture(scope="module")
def toks():
    transformers = pytest.importorskip("transformers")
    vocab = {t: i for i, t in enumerate(V2)}
    return (WordPieceTokenizer(vocab, do_lower_case=True, max_length=512),
            transformers.BertTokenizer(vocab=vocab, do_lower_case=True),
            WordPieceTokenizer(vocab, do_lower_case=False, max_length=512),
            transformers.BertTokenizer(vocab=vocab, do_lower_case=False))


def test_fixed_texts_equal_python_and_transformers(toks):
    mine, ref, cased, cased_ref = toks
    texts = TEXTS + ["a\x01b \x00 [MASK] [mask] FOO\tBar\r\nbaz\x7f!", "x" * 101 + " " + "x" * 100, "co-op coop co_op __init__",
                     "[CLS][SEP] [ MASK ] [MASK]. [UNK]x", "e" * 99 + "!" + "e" * 101, "\x0b\x0c\x1c\x1f a\x0bb", "  \t\n ", "#" * 300,
                     "z" * 30 + "y" * 40 + " " + "z" * 31, "'s it's", "A.B.C. U.S.A", "1" * 50 + "11"]
    got = _rows(mine, texts)
    for text, row in zip(texts, got):
        assert row == mine(text), text
        assert row == ref.encode(text, add_special_tokens=True), text
    for text, row in zip(texts, _rows(cased, texts)):
        assert row == cased(text) == cased_ref.encode(text, add_special_tokens=True), text


def test_truncation_and_batch_shape(toks):
    mine, ref, _, _ = toks
    long = "the quick brown fox jumps! " * 200
    texts = [long, "the fox", "", long[:40]]
    for cap in (2, 3, 16, 511, 512):
        rows = _rows(mine, texts, max_length=cap)
        want = [ref.encode(t, add_special_tokens=True, truncation=True, max_length=cap) for t in texts]
        assert rows == want, cap
        ids, lens = mine.encode_batch(texts, max_length=cap)
        assert ids.shape[1] == max(len(w) for w in want)
    ids, lens = mine.encode_batch([])
    assert ids.shape[0] == 0 and lens.shape == (0,)
    # thread counts do not matter
    many = [f"{long[: (7 * i) % 900]} {i}" for i in range(700)]
    a = _rows(mine, many, n_threads=1)
    assert a == _rows(mine, many, n_threads=7) == [mine(t) for t in many]


def test_non_ascii_texts_take_the_python_path_inside_a_batch(toks):
    mine, ref, _, _ = toks
    texts = ["the quick brown fox", "naïve café résumé", "中文 and 中a文", "hello world", "Ünaffable İstanbul ǅ", "dog."]
    rows = _rows(mine, texts)
    assert rows == [mine(t) for t in texts] == [ref.encode(t, add_special_tokens=True) for t in texts]


def test_mpnet_specials_equal_transformers(tmp_path):
    transformers = pytest.importorskip("transformers")
    words = ["<s>", "<pad>", "</s>", "<unk>", "[UNK]", "the", "quick", "brown", "fox", "jump", "##s", "##ed", "over", "lazy", "dog",
             ",", ".", "!", "a", "##b", "##c", "un", "##believ", "##able", "caf", "##e", "2024", "<mask>", "<", ">", "/", "mask", "unk",
             "s", "pad", "x"]
    vp = tmp_path / "vocab.txt"
    vp.write_text("\n".join(words) + "\n")
    ref = transformers.MPNetTokenizer(str(vp))
    mine = WordPieceTokenizer.from_file(str(vp), cls_token="<s>", sep_token="</s>", pad_token="<pad>", unk_token="[UNK]",
                                        mask_token="<mask>")
    texts = ["the <mask> fox", "the<mask>fox <MASK> x<s>x</s> <unk> a<unk>b [UNK]a <pad>x <PAD>", "<s>the</s>", "  <mask>  .<mask>",
             "The quick brown fox jumps over the lazy dog.", "Unbelievable, café! abc 2024 xyz", "</S> <S>"]
    rows = _rows(mine, texts)
    assert rows == [mine(t) for t in texts] == [ref(t)["input_ids"] for t in texts]


@settings(max_examples=300, deadline=None)
@given(st.lists(st.text(alphabet=st.characters(min_codepoint=0, max_codepoint=127), max_size=80), min_size=1, max_size=8))
def test_random_ascii_equals_python_and_transformers(texts):
    transformers = pytest.importorskip("transformers")
    vocab = {t: i for i, t in enumerate(V2)}
    mine = _cached("u", lambda: WordPieceTokenizer(vocab, do_lower_case=True, max_length=48))
    ref = _cached("r", lambda: transformers.BertTokenizer(vocab=vocab, do_lower_case=True))
    rows = _rows(mine, texts)
    for t, row in zip(texts, rows):
        assert row == mine(t), repr(t)
        assert row == ref.encode(t, add_special_tokens=True, truncation=True, max_length=48), repr(t)


_CACHE: dict = {}


def _cached(key, make):
    if key not in _CACHE:
        _CACHE[key] = make()
    return _CACHE[key]


def test_words_of_english_shape_at_volume(toks):
    """20,000 word-salad texts through 8 threads: every row equals the python tokeniser's."""
    mine = toks[0]
    rng = np.random.default_rng(5)
    words = np.array(["the", "quick", "brown", "fox", "jumps", "jumping", "jumped", "over", "lazy", "dog", "unaffable", "reranker",
                      "embedding", "Hello", "WORLD", "rag-arc", "mi355x", "2025", "$20", "don't", "(hello)", "xx", "qqq", "e" * 120])
    texts = [" ".join(rng.choice(words, size=int(rng.integers(0, 90)))) for _ in range(20_000)]
    rows = _rows(mine, texts, n_threads=8)
    step = 37
    assert rows[::step] == [mine(t) for t in texts[::step]]
    assert all(r[0] == mine.cls and r[-1] == mine.sep for r in rows)
