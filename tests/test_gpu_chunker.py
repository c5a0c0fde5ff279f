"""The chunker's cosine kernels (rarc_cosine_matrix_f32, rarc_adjacent_cosine_distance_f32) on the MI355X: bit-equal to
the oracle (same summation order), within 1e-14 of the numbers the reference produced, and the mirrored
SemanticChunker end to end — with a host embedding provider and with the HIP encoder handing its embeddings over in
HBM."""
import json
import os
import struct

import numpy as np
import pytest

from tests.helpers import CHUNKER_CASES, CHUNKER_SHORT_TEXTS, CHUNKER_TEXT, ChunkerFakeEmbeddings, fp16_grid_matrix

pytestmark = pytest.mark.gpu
GOLD = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "chunker.json")))


def unhex(h):
    return struct.unpack(">d", bytes.fromhex(h))[0]


@pytest.mark.parametrize("nx,ny,d", [(1, 1, 1), (3, 5, 63), (7, 4, 64), (33, 65, 384), (64, 130, 1000), (2, 2, 4097)])
def test_cosine_matrix_bit_equal_to_oracle(oracle, nx, ny, d):
    from rag_arc_amd.core.file_management.chunker import cosine_similarity, device_cosine_distances

    x, y = fp16_grid_matrix(nx, d, 5), fp16_grid_matrix(ny, d, 6)
    if nx > 2:
        x[1] = 0.0                                                   # a zero row: similarities 0, distances 1
    got = cosine_similarity(x, y)
    want = oracle.cosine_matrix_f64(x, y)
    assert got.dtype == np.float64 and got.shape == (nx, ny)
    assert np.array_equal(got.view(np.uint64), want.view(np.uint64))
    ref = np.dot(x.astype(np.float64), y.astype(np.float64).T)      # the reference's formula, numpy's own order
    with np.errstate(divide="ignore", invalid="ignore"):
        ref = ref / np.outer(np.linalg.norm(x.astype(np.float64), axis=1), np.linalg.norm(y.astype(np.float64), axis=1))
    ref[~np.isfinite(ref)] = 0.0
    assert np.max(np.abs(got - ref)) < 1e-13
    dist = np.asarray(device_cosine_distances(x))
    assert np.array_equal(dist.view(np.uint64), oracle.adjacent_cosine_distances(x).view(np.uint64))
    if nx > 2:
        assert dist[0] == 1.0 and dist[1] == 1.0
    with pytest.raises(ValueError):
        cosine_similarity(x, fp16_grid_matrix(ny, d + 1, 7))


def test_zero_rows_against_the_reference():
    from rag_arc_amd.core.file_management.chunker import cosine_similarity

    e = ChunkerFakeEmbeddings().embed_documents(["Rivers run.", "Bread bakes."])
    z = [0.0] * len(e[0])
    got = cosine_similarity([e[0], z], [e[1], z, e[0]])
    want = np.array([[unhex(h) for h in row] for row in GOLD["zero_row_matrix_hex"]])
    assert np.max(np.abs(got - want)) < 1e-15


def test_semantic_chunker_matches_the_reference_chunks():
    from rag_arc_amd.core.file_management.chunker import SemanticChunker

    for params, case in zip(CHUNKER_CASES, GOLD["cases"]):
        ch = SemanticChunker(ChunkerFakeEmbeddings(), **params)
        import re

        dist, sentences = ch._calculate_sentence_distances(re.split(ch.sentence_split_regex, CHUNKER_TEXT))
        want = np.array([unhex(h) for h in case["distances_hex"]])
        assert np.max(np.abs(np.asarray(dist) - want)) < 1e-14
        assert [s["distance_to_next"] for s in sentences[:-1]] == dist and "distance_to_next" not in sentences[-1]
        assert ch.split_text(CHUNKER_TEXT) == case["chunks"], params
    for text, case in zip(CHUNKER_SHORT_TEXTS, GOLD["short"]):
        assert SemanticChunker(ChunkerFakeEmbeddings()).split_text(text) == case["chunks"]


def test_chunker_with_the_hip_encoder_keeps_embeddings_on_the_device(oracle):
    """HipBertEmbeddings hands its embeddings over in HBM (embed_documents_device): distances from the device tensor
    equal the ones computed from the same embeddings taken through host lists."""
    from rag_arc_amd.core.file_management.chunker import SemanticChunker, device_cosine_distances
    from rag_arc_amd.encapsulation.embeddings.hip_bert import HipBertEmbeddings, HipBertEncoder

    sd = oracle.random_bert_state_dict(128, 2, 2, 256, vocab=200, max_pos=64, seed=4)

    def tokenize(text):
        import zlib

        return [101] + [3 + zlib.crc32(w.encode()) % 190 for w in text.split()][:40] + [102]

    enc = HipBertEmbeddings(HipBertEncoder(sd, num_heads=2), tokenize, max_length=64, batch_size=8)
    ch = SemanticChunker(enc, breakpoint_threshold_type="percentile", breakpoint_threshold_amount=60)
    import re

    pieces = re.split(ch.sentence_split_regex, CHUNKER_TEXT)
    dist, sentences = ch._calculate_sentence_distances(pieces)
    host = np.asarray([s["combined_sentence_embedding"] for s in sentences], np.float32)
    assert np.array_equal(np.asarray(dist).view(np.uint64), np.asarray(device_cosine_distances(host)).view(np.uint64))
    assert np.array_equal(np.asarray(dist).view(np.uint64), oracle.adjacent_cosine_distances(host).view(np.uint64))
    chunks = ch.split_text(CHUNKER_TEXT)
    assert " ".join(chunks) == " ".join(pieces) and len(chunks) >= 2
