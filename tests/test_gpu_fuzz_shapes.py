"""Seeded sweep over shapes the hand-picked tests do not name: odd row counts around tile and launch boundaries,
dimensions off the padding grid, every storage format, both metrics, k from 1 to several hundred, data with outlier
tiles (what the per-tile error bound exists for) — ids and scores bit-identical to the oracle every time."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
# RARC_FUZZ_SEEDS="first:last" widens the sweep (the suite runs seeds 0-13)
_FIRST, _LAST = (int(v) for v in os.environ.get("RARC_FUZZ_SEEDS", "0:14").split(":"))


def _oracle_search(oracle, storage, X, Q, k, metric):
    norm = metric == "cosine"
    qn = oracle.normalize_L2(Q) if norm else Q
    if storage == "f16":
        rows, _ = oracle.ingest_f16(X, normalize=norm)
        return oracle.flat_search_f16(rows, qn, k)[:2]
    if storage == "f8":
        b8, s8, _ = oracle.ingest_f8(X, normalize=norm)
        return oracle.flat_search_f8(b8, s8, qn, k)[:2]
    rows, _ = oracle.ingest_f32(X, normalize=norm)
    return oracle.flat_search_f32(rows, qn, k)[:2]


@pytest.mark.parametrize("seed", range(_FIRST, _LAST))
def test_random_shape(oracle, seed):
    from rag_arc_amd.hip.engine import FlatIndexF16

    rng = np.random.default_rng(1000 + seed)
    storage = ("f16", "f8", "f32")[seed % 3]
    metric = "cosine" if seed % 4 else "ip"
    d = int(rng.choice([17, 100, 128, 300, 384, 500, 768, 1000, 1024]))
    n = int(rng.choice([1, 31, 33, 1000, 8191, 8193, 70_001, 140_000, 262_145]))
    nq = int(rng.choice([1, 7, 64, 256, 300]))
    k = int(rng.choice([1, 10, 100, 257, 600]))
    X = rng.standard_normal((n, d)).astype(np.float32)
    if seed % 2:                                     # outlier tiles / rows: one huge component here and there
        hot = rng.integers(0, n, max(1, n // 500))
        X[hot, rng.integers(0, d, hot.size)] *= 40.0
    if metric == "ip":
        X *= np.exp(rng.uniform(-2, 2, (n, 1))).astype(np.float32)
    Q = rng.standard_normal((nq, d)).astype(np.float32)
    scan = "auto" if storage != "f16" else ("q8", "mfma16", "auto")[seed % 3]
    if scan == "mfma16" and d > 768:
        scan = "q8"
    idx = FlatIndexF16(d, metric=metric, storage=storage, scan=scan)
    half = n // 2
    if half:
        idx.add(X[:half])
    idx.add(X[half:])                                # two appends: metadata of the boundary tile is recomputed
    D, I = idx.search(Q, k)
    rI, rD = _oracle_search(oracle, storage, X, Q, k, metric)
    assert np.array_equal(I, rI), (storage, metric, d, n, nq, k, scan)
    assert np.array_equal(D.view(np.uint32), rD.view(np.uint32)), (storage, metric, d, n, nq, k, scan)
    if k <= n:
        # the exhaustive batched verification agrees: no stored row beats any returned k-th entry
        import torch

        ids_t, sc_t = torch.from_numpy(I).cuda(), torch.from_numpy(D).cuda()
        assert idx.verify_batch(torch.from_numpy(Q).cuda(), ids_t, sc_t) == 0, (storage, metric, d, n, nq, k, scan)


@pytest.mark.parametrize("seed", range(_FIRST, max(_FIRST + 1, _FIRST + (_LAST - _FIRST) // 2)))
def test_random_wide_shape(oracle, seed):
    """The same sweep over what takes the wide path (csrc/wide.hip): rows of more than 1024 padded dimensions, and k beyond
    1024 at any width — fp16 and fp32 rows, both metrics, rows that tie (exact duplicates: the id order decides)."""
    from rag_arc_amd.hip.engine import FlatIndexF16

    rng = np.random.default_rng(5000 + seed)
    storage = ("f16", "f32")[seed % 2]
    metric = "cosine" if seed % 3 else "ip"
    if seed % 4 == 3:                                # a narrow row with a k only the wide path takes
        d, k = int(rng.choice([100, 768, 1024])), int(rng.choice([1025, 2500, 8192]))
    else:
        d, k = int(rng.choice([1025, 1100, 1536, 2000, 2048, 3072, 4000, 4096])), int(rng.choice([1, 10, 100, 1025, 3000]))
    n = int(rng.choice([1, 33, 1000, 2047, 2049, 8193, 40_001, 70_001, 131_073]))
    nq = int(rng.choice([1, 7, 256, 300]))
    while n * d * nq > 2.5e10 and nq > 1:            # (the oracle is a CPU: keep a case to seconds)
        nq = max(1, nq // 4)
    X = rng.standard_normal((n, d)).astype(np.float32)
    if seed % 2:
        hot = rng.integers(0, n, max(1, n // 500))
        X[hot, rng.integers(0, d, hot.size)] *= 40.0
    if seed % 5 == 0 and n > 100:                    # exact duplicates of a few rows, far apart: ties on the score
        src = rng.integers(0, n, 20)
        X[rng.integers(0, n, 20)] = X[src]
    if metric == "ip":
        X *= np.exp(rng.uniform(-2, 2, (n, 1))).astype(np.float32)
    Q = rng.standard_normal((nq, d)).astype(np.float32)
    idx = FlatIndexF16(d, metric=metric, storage=storage)
    assert idx._takes_wide_path(k)
    half = n // 2
    if half:
        idx.add(X[:half])
    idx.add(X[half:])
    D, I = idx.search(Q, k)
    rI, rD = _oracle_search(oracle, storage, X, Q, k, metric)
    assert np.array_equal(I, rI), (storage, metric, d, n, nq, k)
    assert np.array_equal(D.view(np.uint32), rD.view(np.uint32)), (storage, metric, d, n, nq, k)
