"""Seeded sweep over an index's LIFE, not one search: ragged adds into a growable arena, deletes by compaction (random ids,
runs, the ends), more adds, searches at every stage — through the register-resident scans AND the wide path (d beyond 1024,
k beyond 1024) — against the oracle on the rows that should be there.  ids and scores bit-identical every time."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
_FIRST, _LAST = (int(v) for v in os.environ.get("RARC_FUZZ_SEEDS", "0:12").split(":"))


def _oracle_search(oracle, storage, X, Q, k, metric):
    norm = metric == "cosine"
    qn = oracle.normalize_L2(Q) if norm else Q
    if storage == "f16":
        rows, _ = oracle.ingest_f16(X, normalize=norm)
        return oracle.flat_search_f16(rows, qn, k)[:2]
    if storage == "f8":
        b8, s8, _ = oracle.ingest_f8(X, normalize=norm)
        return oracle.flat_search_f8(b8, s8, qn, k)[:2]
    rows = oracle.ingest_f32(X, normalize=norm)[0]
    return oracle.flat_search_f32(rows, qn, k)[:2]


@pytest.mark.parametrize("seed", range(_FIRST, _LAST))
def test_random_lifecycle(oracle, seed):
    from rag_arc_amd.hip.engine import FlatIndexF16

    rng = np.random.default_rng(5000 + seed)
    wide = seed % 3 == 2
    d = int(rng.choice([1100, 1536, 2100, 3072]) if wide else rng.choice([24, 128, 384, 700, 1024]))
    storage = ("f16", "f32")[seed % 2] if wide else ("f16", "f8", "f32")[seed % 3]
    metric = "cosine" if seed % 5 else "ip"
    growable = bool(seed % 2)
    idx = FlatIndexF16(d, metric=metric, storage=storage, growable=growable)
    kept = np.zeros((0, d), dtype=np.float32)
    Q = rng.standard_normal((int(rng.choice([1, 9, 64])), d)).astype(np.float32)
    for step in range(5):
        n_add = int(rng.choice([1, 33, 500, 4097, 30_000]))
        X = rng.standard_normal((n_add, d)).astype(np.float32)
        if metric == "ip":
            X *= np.exp(rng.uniform(-1, 1, (n_add, 1))).astype(np.float32)
        idx.add(X)
        kept = np.concatenate([kept, X])
        if step % 2 and len(kept) > 3:
            n_del = int(rng.integers(1, max(2, len(kept) // 3)))
            holes = np.unique(np.concatenate([rng.integers(0, len(kept), n_del), np.array([0, len(kept) - 1][: int(rng.integers(0, 3))], dtype=np.int64)]).astype(np.int64))
            if rng.random() < 0.5 and len(kept) > 200:
                start = int(rng.integers(0, len(kept) - 100))
                holes = np.unique(np.concatenate([holes, np.arange(start, start + 100)]))
            assert idx.remove_rows(holes) == holes.size
            kept = np.delete(kept, holes, axis=0)
        assert idx.ntotal == len(kept)
        if len(kept) == 0:
            continue
        ks = [int(rng.choice([1, 10, 100]))] + ([int(rng.choice([1030, 2000]))] if (storage != "f8" and len(kept) > 1030 and step == 4) else [])
        for k in ks:
            kk = min(k, len(kept))
            D, I = idx.search(Q, kk)
            rI, rD = _oracle_search(oracle, storage, kept, Q, kk, metric)
            assert np.array_equal(I, rI), (seed, step, storage, metric, d, len(kept), kk)
            assert np.array_equal(D.view(np.uint32), rD.view(np.uint32)), (seed, step, storage, metric, d, len(kept), kk)
