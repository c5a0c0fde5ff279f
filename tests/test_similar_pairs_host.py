"""The oracle of the all-pairs cosine (oracle/cpu_ref.similar_pairs_f64: Base_Neo4j.py:559-583) against scikit-learn's
cosine_similarity — the function the reference calls — and against the committed vectors.  No GPU needed."""
import json
import os

import numpy as np
import pytest

GOLD = os.path.join(os.path.dirname(__file__), "golden", "similar_pairs.json")


def _reference_loop(embeddings, threshold):
    """Base_Neo4j.py:559-566 verbatim in behaviour: sklearn's matrix, then i < j with >= threshold."""
    from sklearn.metrics.pairwise import cosine_similarity

    similarity_matrix = cosine_similarity(np.array(embeddings))
    out = []
    for i in range(len(embeddings)):
        for j in range(i + 1, len(embeddings)):
            if similarity_matrix[i][j] >= threshold:
                out.append((i, j, float(similarity_matrix[i][j])))
    return out


def _cases():
    rng = np.random.default_rng(11)
    base = rng.standard_normal((40, 24))
    x = np.concatenate([base, base[:12] + 0.03 * rng.standard_normal((12, 24)), 3.0 * base[5:9], np.zeros((2, 24))])
    yield "near-duplicates, scaled copies, zero rows", x.tolist(), 0.95
    yield "a low threshold", rng.standard_normal((30, 8)).tolist(), 0.3
    yield "nothing similar", rng.standard_normal((25, 64)).tolist(), 0.95
    yield "two entities", [[1.0, 2.0, 3.0], [1.0, 2.0, 3.1]], 0.95
    yield "one entity", [[1.0, 2.0]], 0.95


def test_oracle_equals_sklearn_loop(oracle):
    pytest.importorskip("sklearn")
    for name, emb, thr in _cases():
        want = _reference_loop(emb, thr) if len(emb) >= 2 else []
        got = oracle.similar_pairs_f64(emb, thr)
        assert [(i, j) for i, j, _ in got] == [(i, j) for i, j, _ in want], name
        assert np.allclose([s for *_, s in got], [s for *_, s in want], rtol=0, atol=1e-14), name


def test_oracle_equals_committed_vectors(oracle):
    gold = json.load(open(GOLD))
    assert gold["cases"]
    for case in gold["cases"]:
        emb = [[float.fromhex(v) for v in row] for row in case["embeddings_hex"]]
        got = oracle.similar_pairs_f64(emb, case["threshold"])
        assert [[i, j] for i, j, _ in got] == case["pairs"], case["name"]
        assert np.allclose([s for *_, s in got], [float.fromhex(v) for v in case["scores_hex"]], rtol=0, atol=1e-14), case["name"]


def test_no_cpu_fallback():
    """Without a ROCm device the product call fails loudly (the oracle above is test infrastructure, never a fallback)."""
    import torch

    if torch.cuda.is_available():
        pytest.skip("a GPU is visible: tests/test_gpu_similar_pairs.py covers the call")
    from rag_arc_amd.encapsulation.database.graph_db import similar_pairs
    from rag_arc_amd.hip.binding import RarcError

    with pytest.raises(RarcError):
        similar_pairs([[1.0, 2.0], [2.0, 1.0]], 0.95)
