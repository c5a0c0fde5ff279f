"""fp8 corpus (BASELINE.json config 5's storage: e4m3fn bytes + one fp32 scale per row) through the
C-ABI vs the CPU oracle: ingest bytes and scales bit-identical, search ids and scores bit-identical.

Reference call sites replaced: FaissVectorStore.add_texts / similarity_search_by_vector_with_score
(encapsulation/database/vector_db/VectorStore_Faiss.py:170-202, :258-272) with 8-bit row storage."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _data(n, d, nq, seed, spread=False):
    rng = np.random.default_rng(seed)
    X = rng.standard_normal((n, d)).astype(np.float32) * 3.0
    if spread:
        X *= np.exp(rng.uniform(-4, 4, (n, 1))).astype(np.float32)
    return X, rng.standard_normal((nq, d)).astype(np.float32) * 0.5


def _check(oracle, X, Q, k, metric="cosine"):
    from rag_arc_amd.hip.engine import FlatIndexF16

    n, d = X.shape
    idx = FlatIndexF16(d, metric=metric, storage="f8")
    idx.add(X[: n // 2])
    idx.add(X[n // 2:])                                   # two appends: the straddled tile is re-quantised
    ref_b, ref_s, ref_n2 = oracle.ingest_f8(X, normalize=(metric == "cosine"))
    assert idx.d_pad == ref_b.shape[1]
    assert np.array_equal(idx.rows.cpu().numpy(), ref_b), "fp8 bytes differ from the oracle"
    assert np.array_equal(idx.row_scales.cpu().numpy().view(np.uint32), ref_s.view(np.uint32)), "row scales differ"
    kk = min(k, n)
    D, I = idx.search(Q, kk)
    qn = oracle.normalize_L2(Q) if metric == "cosine" else Q
    rI, rD, _ = oracle.flat_search_f8(ref_b, ref_s, qn, kk)
    assert np.array_equal(I, rI), f"ids differ (n={n} d={d} k={kk})"
    assert np.array_equal(D.view(np.uint32), rD.view(np.uint32)), "scores not bit-identical"
    # and close to the float64 truth over the stored values
    dec = oracle.f8_decode(ref_b).astype(np.float64) * ref_s[:, None].astype(np.float64)
    qp = np.zeros((Q.shape[0], ref_b.shape[1])); qp[:, :d] = qn
    s64 = np.take_along_axis(qp @ dec.T, I, axis=1)
    assert np.max(np.abs(s64 - D) / np.maximum(1.0, np.abs(s64))) < 1e-5
    return idx


@pytest.mark.parametrize("n,d,nq,k", [(1, 384, 1, 1), (33, 768, 5, 10), (1000, 100, 9, 10), (5000, 768, 256, 100),
                                      (20000, 1024, 256, 100), (4097, 300, 64, 50)])
def test_fp8_search_matches_oracle(oracle, n, d, nq, k):
    X, Q = _data(n, d, nq, seed=n + d)
    _check(oracle, X, Q, k)


def test_fp8_inner_product_with_spread_norms(oracle):
    X, Q = _data(30_000, 512, 40, seed=9, spread=True)
    _check(oracle, X, Q, 64, metric="ip")


def test_fp8_repair_and_verify(oracle):
    """A candidate buffer far too small: every query overflows, is flagged and repaired exactly."""
    from rag_arc_amd.hip.engine import FlatIndexF16

    X, Q = _data(50_000, 768, 16, seed=77)
    idx = FlatIndexF16(768, cand_cap=512, storage="f8")      # x8 for k = 900: 16 slots per (workgroup, query)
    idx.add(X)
    D, I = idx.search(Q, 900)
    assert len(idx.last_repaired) == 16
    b, s, _ = oracle.ingest_f8(X)
    rI, rD, _ = oracle.flat_search_f8(b, s, oracle.normalize_L2(Q), 900)
    assert np.array_equal(I, rI) and np.array_equal(D.view(np.uint32), rD.view(np.uint32))
    ids, sc = idx.search_device(Q, 10)
    assert idx.verify_query(Q, 3, ids, sc) == 0


def test_fp8_split_scan_large_shard(oracle):
    """2.2M fp8 rows: the scan runs as two launches around the exact mid-scan pass (fp8 rescoring there too)."""
    from rag_arc_amd.hip.engine import FlatIndexF16

    n, d, nq, k = 2_200_000, 200, 24, 40
    rng = np.random.default_rng(77)
    X = rng.standard_normal((n, d)).astype(np.float32)
    Q = rng.standard_normal((nq, d)).astype(np.float32)
    idx = FlatIndexF16(d, metric="cosine", storage="f8")
    for s in range(0, n, 550_000):
        idx.add(X[s:s + 550_000])
    ref_b, ref_s, _ = oracle.ingest_f8(X, normalize=True)
    D, I = idx.search(Q, k)
    rI, rD, _ = oracle.flat_search_f8(ref_b, ref_s, oracle.normalize_L2(Q), k)
    assert np.array_equal(I, rI) and np.array_equal(D.view(np.uint32), rD.view(np.uint32))
    assert len(idx.last_repaired) == 0
