"""Semantic chunker: the oracle's float64 cosines and the host-side mirror of
core/file_management/chunker/spliter.py:307-534 against numbers and chunks the reference itself produced
(tests/golden/chunker.json, written by tests/golden/make_golden.py).  CPU only: the device entry point is replaced by
the oracle here; tests/test_gpu_chunker.py runs the kernels."""
import json
import os
import re
import struct

import numpy as np
import pytest

from tests.helpers import CHUNKER_CASES, CHUNKER_SHORT_TEXTS, CHUNKER_TEXT, ChunkerFakeEmbeddings

GOLD = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "chunker.json")))


def unhex(h):
    return struct.unpack(">d", bytes.fromhex(h))[0]


def test_oracle_distances_match_the_reference(oracle):
    for case in GOLD["cases"]:
        buf = case["params"].get("buffer_size", 1)
        pieces = re.split(r"(?<=[.?!])\s+", CHUNKER_TEXT)
        windows = [" ".join(pieces[max(0, i - buf): i + buf + 1]) for i in range(len(pieces))]
        emb = np.asarray(ChunkerFakeEmbeddings().embed_documents(windows), np.float32)
        want = np.array([unhex(h) for h in case["distances_hex"]])
        got = oracle.adjacent_cosine_distances(emb)
        assert got.shape == want.shape and np.max(np.abs(got - want)) < 1e-14     # summation order only
        full = oracle.cosine_matrix_f64(emb, emb)
        assert np.max(np.abs((1.0 - np.diagonal(full, 1)) - want)) < 1e-14


def test_oracle_zero_rows_give_zero_similarity(oracle):
    e = ChunkerFakeEmbeddings().embed_documents(["Rivers run.", "Bread bakes."])
    z = [0.0] * len(e[0])
    got = oracle.cosine_matrix_f64(np.array([e[0], z], np.float32), np.array([e[1], z, e[0]], np.float32))
    want = np.array([[unhex(h) for h in row] for row in GOLD["zero_row_matrix_hex"]])
    assert np.max(np.abs(got - want)) < 1e-15 and got[1].tolist() == [0.0, 0.0, 0.0] and got[0, 1] == 0.0


@pytest.fixture()
def chunker_on_oracle(oracle, monkeypatch):
    from rag_arc_amd.core.file_management.chunker import spliter

    monkeypatch.setattr(spliter, "device_cosine_distances",
                        lambda emb: oracle.adjacent_cosine_distances(np.asarray(emb, np.float32)).tolist())
    return spliter


def test_host_logic_reproduces_the_reference_chunks(chunker_on_oracle):
    assert len(CHUNKER_CASES) == len(GOLD["cases"])
    for params, case in zip(CHUNKER_CASES, GOLD["cases"]):
        assert case["params"] == params
        ch = chunker_on_oracle.SemanticChunker(ChunkerFakeEmbeddings(), **params)
        assert ch.split_text(CHUNKER_TEXT) == case["chunks"], params
    for text, case in zip(CHUNKER_SHORT_TEXTS, GOLD["short"]):
        assert chunker_on_oracle.SemanticChunker(ChunkerFakeEmbeddings()).split_text(text) == case["chunks"]
        got = chunker_on_oracle.SemanticChunker(ChunkerFakeEmbeddings(), breakpoint_threshold_type="gradient").split_text(text)
        assert got == case["chunks_gradient"]


def test_combine_sentences_and_errors(chunker_on_oracle):
    s = [{"sentence": w} for w in ("a", "b", "c", "d")]
    assert [e["combined_sentence"] for e in chunker_on_oracle.combine_sentences(s, 1)] == ["a b", "a b c", "b c d", "c d"]
    assert [e["combined_sentence"] for e in chunker_on_oracle.combine_sentences(s, 0)] == ["a", "b", "c", "d"]
    assert [e["combined_sentence"] for e in chunker_on_oracle.combine_sentences(s, 5)] == ["a b c d"] * 4
    with pytest.raises(ValueError):
        chunker_on_oracle.SemanticChunker(ChunkerFakeEmbeddings(), breakpoint_threshold_type="percentile")._threshold_from_clusters([0.1])
    bad = chunker_on_oracle.SemanticChunker(ChunkerFakeEmbeddings())
    bad.breakpoint_threshold_type = "nope"
    with pytest.raises(ValueError):
        bad._calculate_breakpoint_threshold([0.1, 0.2])
    with pytest.raises(KeyError):
        chunker_on_oracle.SemanticChunker(ChunkerFakeEmbeddings(), breakpoint_threshold_type="nope")
    assert chunker_on_oracle.cosine_similarity([], [[1.0]]).size == 0          # empty input: no device needed
