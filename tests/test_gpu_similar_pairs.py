"""All pairs with cosine >= threshold on the GPU (csrc/pairs.hip through rag_arc_amd...graph_db.similar_pairs) against the
oracle (numpy float64 restatement of Base_Neo4j.py:559-583, itself pinned to scikit-learn) and the committed vectors.
Tolerance: the pair SET is the oracle's except where a pair's float64 cosine lies within 1e-12 of the threshold (two
summation orders of a float64 dot product may fall on either side); scores within 1e-12."""
import json
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(__file__), "golden", "similar_pairs.json")
BAND = 1e-12


def _same(got, want, thr, what):
    g = {(i, j): s for i, j, s in got}
    w = {(i, j): s for i, j, s in want}
    for key in set(g) ^ set(w):
        s = g.get(key, w.get(key))
        assert abs(s - thr) <= BAND, (what, key, s, "in only one of the two answers")
    common = sorted(set(g) & set(w))
    assert len(common) >= len(w) - 2, what
    if common:
        assert max(abs(g[k] - w[k]) for k in common) <= 1e-12, what
    assert [(i, j) for i, j, _ in got] == sorted((i, j) for i, j, _ in got), what      # ordered by (i, j), as the reference's loop


def test_committed_vectors():
    from rag_arc_amd.encapsulation.database.graph_db import similar_pairs

    for case in json.load(open(GOLD))["cases"]:
        emb = [[float.fromhex(v) for v in row] for row in case["embeddings_hex"]]
        got = similar_pairs(emb, case["threshold"])
        want = [(i, j, float.fromhex(s)) for (i, j), s in zip(case["pairs"], case["scores_hex"])]
        _same(got, want, case["threshold"], case["name"])
        assert all(type(i) is int and type(j) is int and type(s) is float for i, j, s in got)


def _planted(rng, n, d, n_dup, noise):
    x = rng.standard_normal((n, d)).astype(np.float32)
    src = rng.integers(0, n, n_dup)
    dst = rng.integers(0, n, n_dup)
    x[dst] = x[src] * rng.uniform(0.2, 5.0, (n_dup, 1)).astype(np.float32) + noise * rng.standard_normal((n_dup, d)).astype(np.float32)
    return x


@pytest.mark.parametrize("n,d,thr,noise", [(700, 1, 0.95, 0.0), (1000, 37, 0.95, 0.08), (5000, 200, 0.95, 0.1), (3000, 1000, 0.9, 0.3),
                                           (9000, 384, 0.95, 0.15), (2500, 4096, 0.97, 0.1), (4097, 64, 0.8, 0.2), (255, 768, 0.95, 0.1)])
def test_random_corpora_with_planted_near_duplicates(oracle, n, d, thr, noise):
    from rag_arc_amd.encapsulation.database.graph_db import similar_pairs

    rng = np.random.default_rng(n + d)
    x = _planted(rng, n, d, max(5, n // 20), noise)
    if d == 1:
        x = x[:60]                       # (one dimension: every pair is +-1 — keep the answer small)
    got = similar_pairs(x, thr)
    want = oracle.similar_pairs_f64(x.astype(np.float64), thr)
    assert len(want) > 0
    _same(got, want, thr, (n, d, thr))


def test_lists_and_the_output_grow_when_everything_is_similar(oracle):
    """600 copies of three entities: 60,000+ pairs per group, every nomination list and the first output buffer overflow —
    the wrapper repeats with larger ones; a device tensor is taken where it is."""
    import torch

    from rag_arc_amd.encapsulation.database.graph_db import similar_pairs

    rng = np.random.default_rng(5)
    base = rng.standard_normal((3, 96)).astype(np.float32)
    x = base[rng.integers(0, 3, 600)] * rng.uniform(0.5, 2.0, (600, 1)).astype(np.float32)
    got = similar_pairs(torch.from_numpy(x).cuda(), 0.95)
    want = oracle.similar_pairs_f64(x.astype(np.float64), 0.95)
    assert len(want) > 50_000
    _same(got, want, 0.95, "all similar")


def test_edges():
    from rag_arc_amd.encapsulation.database.graph_db import similar_pairs
    from rag_arc_amd.hip.binding import RarcError

    assert similar_pairs([], 0.95) == [] and similar_pairs([[1.0, 2.0]], 0.95) == []
    assert similar_pairs([[1.0, 0.0], [2.0, 0.0], [0.0, 0.0], [0.0, 1.0]], 0.95) == [(0, 1, 1.0)]      # a zero row matches nothing
    with pytest.raises(ValueError):
        similar_pairs([[1.0, 2.0], [2.0, 1.0]], 0.0)
    with pytest.raises(RarcError):
        similar_pairs(np.zeros((3, 5000), dtype=np.float32), 0.95)


_FIRST, _LAST = (int(v) for v in os.environ.get("RARC_FUZZ_SEEDS", "0:10").split(":"))


@pytest.mark.parametrize("seed", range(_FIRST, _LAST))
def test_random_shapes(oracle, seed):
    """Seeded sweep (RARC_FUZZ_SEEDS widens it): sizes around the 256-row blocks and the 4096-column super-blocks, dimensions
    off every grid, thresholds from loose to 1, duplicates at distances that straddle the threshold."""
    from rag_arc_amd.encapsulation.database.graph_db import similar_pairs

    rng = np.random.default_rng(9000 + seed)
    n = int(rng.choice([2, 3, 255, 256, 257, 1000, 4095, 4096, 4097, 6000, 8193]))
    d = int(rng.choice([1, 2, 7, 64, 100, 257, 768, 1024, 1500, 3072]))
    thr = float(rng.choice([0.5, 0.8, 0.9, 0.95, 0.99, 0.9999]))
    if d <= 7:
        n = min(n, 300)                      # (few dimensions: most pairs are similar — keep the answer small)
    x = rng.standard_normal((n, d)).astype(np.float32)
    n_dup = max(1, n // 10)
    src, dst = rng.integers(0, n, n_dup), rng.integers(0, n, n_dup)
    # distances chosen so that the planted pairs' cosines spread on both sides of the threshold
    noise = np.sqrt(max(1.0 / thr ** 2 - 1.0, 1e-6)) * rng.uniform(0.3, 2.0, (n_dup, 1))
    x[dst] = (x[src] + noise * rng.standard_normal((n_dup, d)) * np.linalg.norm(x[src], axis=1, keepdims=True) / np.sqrt(d)).astype(np.float32)
    if seed % 3 == 0:
        x[rng.integers(0, n, max(1, n // 50))] = 0.0          # zero rows
    got = similar_pairs(x, thr)
    want = oracle.similar_pairs_f64(x.astype(np.float64), thr)
    _same(got, want, thr, (seed, n, d, thr))


def test_arrays_form(oracle):
    from rag_arc_amd.encapsulation.database.graph_db import similar_pairs

    rng = np.random.default_rng(3)
    x = _planted(rng, 2000, 96, 200, 0.05)
    as_list = similar_pairs(x, 0.95)
    pairs, cos = similar_pairs(x, 0.95, as_arrays=True)
    assert pairs.dtype == np.int64 and cos.dtype == np.float64 and pairs.shape == (len(as_list), 2)
    assert as_list == list(zip(pairs[:, 0].tolist(), pairs[:, 1].tolist(), cos.tolist()))
    p0, c0 = similar_pairs(x[:1], 0.95, as_arrays=True)
    assert p0.shape == (0, 2) and c0.shape == (0,)
