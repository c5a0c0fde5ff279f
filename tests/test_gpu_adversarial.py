"""Data the pruning heuristics hate: results must stay bit-identical to the oracle, whatever path
(certificate, overflow flag, exact repair) gets there."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _run(oracle, X, Q, k, metric="cosine"):
    from rag_arc_amd.hip.engine import FlatIndexF16

    rows, _ = oracle.ingest_f16(X, normalize=(metric == "cosine"))
    qn = oracle.normalize_L2(Q) if metric == "cosine" else Q
    rI, rD, _ = oracle.flat_search_f16(rows, qn, k)
    for scan in ("q8", "mfma16"):        # both scan kernels must survive the same abuse
        idx = FlatIndexF16(X.shape[1], metric=metric, scan=scan)
        idx.add(X)
        D, I = idx.search(Q, k)
        assert np.array_equal(I, rI), f"ids differ ({scan})"
        assert np.array_equal(D.view(np.uint32), rD.view(np.uint32)), f"scores differ ({scan})"
    return idx


def test_corpus_sorted_by_similarity_to_a_query(oracle):
    """Every later tile beats every earlier one for query 0: thresholds chase the data the whole scan."""
    rng = np.random.default_rng(0)
    n, d = 60_000, 768
    X = rng.standard_normal((n, d)).astype(np.float32)
    Q = rng.standard_normal((8, d)).astype(np.float32)
    order = np.argsort(X @ Q[0])          # ascending similarity to query 0
    idx = _run(oracle, X[order], Q, 100)
    print("repaired:", getattr(idx, "last_repaired", None))


def test_tight_cluster_of_near_duplicates(oracle):
    rng = np.random.default_rng(1)
    n, d = 40_000, 384
    X = rng.standard_normal((n, d)).astype(np.float32)
    c = rng.standard_normal(d).astype(np.float32)
    X[5000:7000] = c + 0.01 * rng.standard_normal((2000, d)).astype(np.float32)   # 2000 rows within ~1e-4 in cosine
    Q = np.stack([c, c + 0.02 * rng.standard_normal(d).astype(np.float32), rng.standard_normal(d).astype(np.float32)])
    _run(oracle, X, Q, 100)


def test_all_rows_identical(oracle):
    rng = np.random.default_rng(2)
    X = np.tile(rng.standard_normal((1, 256)).astype(np.float32), (6000, 1))
    Q = rng.standard_normal((3, 256)).astype(np.float32)
    _run(oracle, X, Q, 50)


def test_scores_spanning_many_magnitudes_ip(oracle):
    rng = np.random.default_rng(3)
    X = rng.standard_normal((20_000, 512)).astype(np.float32) * np.exp(rng.uniform(-6, 3, (20_000, 1))).astype(np.float32)
    Q = rng.standard_normal((5, 512)).astype(np.float32) * 3
    _run(oracle, X, Q, 64, metric="ip")


def test_k_equals_one_and_large_k(oracle):
    rng = np.random.default_rng(4)
    X = rng.standard_normal((30_000, 768)).astype(np.float32)
    Q = rng.standard_normal((4, 768)).astype(np.float32)
    _run(oracle, X, Q, 1)
    _run(oracle, X, Q, 900)


def test_flags_are_raised_and_repair_fixes_them(oracle):
    """Force the certificate to fail (k' == k leaves no margin) and check repair restores exactness."""
    import torch

    from rag_arc_amd.hip import binding as B
    from rag_arc_amd.hip.engine import FlatIndexF16

    rng = np.random.default_rng(5)
    X = rng.standard_normal((50_000, 768)).astype(np.float32)
    Q = rng.standard_normal((16, 768)).astype(np.float32)
    idx = FlatIndexF16(768, scan="mfma16")  # the fp16 scan is the one with a certificate to fail
    idx.add(X)
    idx.kprime_for = lambda k: k            # no margin: every query is UNCERTAIN by construction
    D, I = idx.search(Q, 100)
    assert len(idx.last_repaired) == 16
    rows, _ = oracle.ingest_f16(X)
    rI, rD, _ = oracle.flat_search_f16(rows, oracle.normalize_L2(Q), 100)
    assert np.array_equal(I, rI) and np.array_equal(D.view(np.uint32), rD.view(np.uint32))


def test_dense_score_band_large_k(oracle):
    """20 000 rows whose scores to one query lie within two error bounds of each other, k = 996: the rows that
    must be rescored (everything within eps8 of the k-th best) outnumber the finalize's 6144-row buffer, so it
    takes them in bands of approximate score — and must still return the oracle's answer without a repair."""
    from rag_arc_amd.hip.engine import FlatIndexF16

    rng = np.random.default_rng(31)
    n, d, k = 300_000, 128, 996
    X = rng.standard_normal((n, d)).astype(np.float32)
    X /= np.linalg.norm(X, axis=1, keepdims=True)
    q = np.zeros((3, d), np.float32)
    q[0, 0] = 1.0
    q[1:] = rng.standard_normal((2, d)).astype(np.float32)
    c = rng.uniform(0.80, 0.83, 20_000).astype(np.float32)          # cosines to q[0], 0.03 wide
    u = X[:20_000].copy()
    u[:, 0] = 0.0
    u /= np.linalg.norm(u, axis=1, keepdims=True)
    X[:20_000] = c[:, None] * q[0] + np.sqrt(1.0 - c * c)[:, None] * u
    X = X[rng.permutation(n)]
    idx = FlatIndexF16(d, scan="q8")
    idx.add(X)
    D, I = idx.search(q, k)
    rows, _ = oracle.ingest_f16(X)
    rI, rD, _ = oracle.flat_search_f16(rows, oracle.normalize_L2(q), k)
    assert np.array_equal(I, rI) and np.array_equal(D.view(np.uint32), rD.view(np.uint32))
    assert len(idx.last_repaired) == 0


def test_dense_score_band_100k_rows(oracle):
    """90 000 rows whose cosines to one query lie within 0.008 of each other (less than one int8 error bound),
    k = 100: every one of them has to be rescored canonically, fifteen buffers' worth.  The banded pass must get
    through all of them (one iteration per band, the width handed on) and return the oracle's answer; if it ever
    runs out of iterations the query is flagged and repaired — a wrong top-k is never returned silently."""
    from rag_arc_amd.hip.engine import FlatIndexF16

    rng = np.random.default_rng(32)
    n, d, k, m = 400_000, 128, 100, 90_000
    X = rng.standard_normal((n, d)).astype(np.float32)
    X /= np.linalg.norm(X, axis=1, keepdims=True)
    q = np.zeros((2, d), np.float32)
    q[0, 0] = 1.0
    q[1] = rng.standard_normal(d).astype(np.float32)
    c = rng.uniform(0.800, 0.808, m).astype(np.float32)
    u = X[:m].copy()
    u[:, 0] = 0.0
    u /= np.linalg.norm(u, axis=1, keepdims=True)
    X[:m] = c[:, None] * q[0] + np.sqrt(1.0 - c * c)[:, None] * u
    X = X[rng.permutation(n)]
    idx = FlatIndexF16(d, scan="q8")
    idx.add(X)
    D, I = idx.search(q, k)
    rows, _ = oracle.ingest_f16(X)
    rI, rD, _ = oracle.flat_search_f16(rows, oracle.normalize_L2(q), k)
    assert np.array_equal(I, rI) and np.array_equal(D.view(np.uint32), rD.view(np.uint32))
    print("repaired:", idx.last_repaired)


@pytest.mark.parametrize("storage", ["f16", "f8"])
def test_overflowing_candidate_segments_are_rerun_from_a_score_floor(oracle, storage):
    """A candidate capacity far too small for the shard: every query overflows its segments on the first attempt.
    The flagged queries are searched again TOGETHER, starting from the k-th score the first attempt did establish
    (rarc_qblock_set_floor), and come back exact — no per-query full scan."""
    from rag_arc_amd.hip.engine import FlatIndexF16

    rng = np.random.default_rng(41)
    n, d, nq, k = 400_000, 256, 48, 100
    X = rng.standard_normal((n, d)).astype(np.float32)
    Q = rng.standard_normal((nq, d)).astype(np.float32)
    idx = FlatIndexF16(d, scan="q8" if storage == "f16" else "auto", storage=storage, cand_cap=4096)
    idx.add(X)
    D, I = idx.search(Q, k)
    if storage == "f16":
        rows, _ = oracle.ingest_f16(X)
        rI, rD, _ = oracle.flat_search_f16(rows, oracle.normalize_L2(Q), k)
    else:
        b8, s8, _ = oracle.ingest_f8(X)
        rI, rD, _ = oracle.flat_search_f8(b8, s8, oracle.normalize_L2(Q), k)
    assert np.array_equal(I, rI) and np.array_equal(D.view(np.uint32), rD.view(np.uint32))
    assert len(idx.last_repaired) >= nq // 2, "the test is meant to overflow"
    assert idx.last_rerun == len(idx.last_repaired), "the re-run should have settled every flagged query"
    # the overflow taught the index something about this corpus: the capacity of LATER searches has doubled (sticky), and
    # after a few batches nothing overflows any more — no batch keeps paying a second scan of the shard
    assert idx.cand_cap == 8192 and idx.cand_cap_grown == 1
    for _ in range(6):
        D2, I2 = idx.search(Q, k)
        assert np.array_equal(I2, rI) and np.array_equal(D2.view(np.uint32), rD.view(np.uint32))
        if not idx.last_repaired:
            break
    assert not idx.last_repaired and idx.cand_cap <= 4096 * 64
    grown = idx.cand_cap_grown
    idx.search(Q, k)
    assert idx.cand_cap_grown == grown and not idx.last_repaired          # steady state


def test_clustered_corpus_whole_cluster_inside_the_int8_margin(oracle):
    """What real embedding corpora look like: anisotropic clusters, queries from the same mixture.  A
    query's WHOLE cluster (12,500 rows here) sits inside the int8 error margin of its k-th best score: more rows at or above
    the histogram's k-th edge than the 6144-key rescore buffer holds, most of them ABOVE a histogram window that was sized
    from a sample the cluster was barely in.  The round-3 tree flagged such a query on every batch (a second scan of the
    shard) and ranked thousands of keys by counting (5.5 of 8.8 ms per batch at 10M rows).  Now: nothing flagged, nothing
    repaired, ids and scores bit-identical to the oracle."""
    import torch

    from rag_arc_amd.hip.engine import FlatIndexF16

    rng = np.random.default_rng(2024)
    n, d, nc, nq, k = 500_000, 256, 40, 64, 100
    spec = (np.arange(1, d + 1, dtype=np.float32) ** -0.5)
    spec = spec / np.linalg.norm(spec) * np.sqrt(d)
    centers = rng.standard_normal((nc, d)).astype(np.float32) * spec
    centers /= np.linalg.norm(centers, axis=1, keepdims=True)

    def make(m):
        c = rng.integers(0, nc, m)
        return (centers[c] + 0.3 / np.sqrt(d) * rng.standard_normal((m, d)).astype(np.float32) * spec).astype(np.float32)

    X, Q = make(n), make(nq)
    rows, _ = oracle.ingest_f16(X)
    rI, rD, _ = oracle.flat_search_f16(rows, oracle.normalize_L2(Q), k)
    for kk in (k, 10):
        idx = FlatIndexF16(d, metric="cosine", scan="q8")
        idx.add(X)
        ids, sc = idx.search_device(torch.from_numpy(Q).cuda(), kk, repair=False)       # first attempt only
        st = idx.last_status.cpu().numpy()
        assert not st.any(), f"flagged on the first attempt: {sorted(set(hex(int(v)) for v in st[st != 0]))}"
        assert np.array_equal(ids.cpu().numpy(), rI[:, :kk])
        assert np.array_equal(sc.cpu().numpy().view(np.uint32), rD[:, :kk].view(np.uint32))
    D, I = idx.search(Q, 1000)                     # k = 1000: the banded rescore with thousands of rows per band
    rI2, rD2, _ = oracle.flat_search_f16(rows, oracle.normalize_L2(Q), 1000)
    assert np.array_equal(I, rI2) and np.array_equal(D.view(np.uint32), rD2.view(np.uint32))


def test_warm_up_pays_the_first_batch_cost_and_a_saved_index_keeps_what_it_learnt(oracle, tmp_path):
    """Clustered rows and a capacity too small for them: warm_up() (stored rows searched as queries) grows the candidate
    capacity before any user query; the first user batch is then exact on its first attempt.  The capacity an index has grown
    to travels with save_local / load_local."""
    import torch

    from rag_arc_amd.encapsulation.database.vector_db import HipFlatVectorStore
    from rag_arc_amd.hip.engine import FlatIndexF16

    rng = np.random.default_rng(77)
    n, d, nc, nq, k = 300_000, 128, 12, 32, 50
    centers = rng.standard_normal((nc, d)).astype(np.float32)
    centers /= np.linalg.norm(centers, axis=1, keepdims=True)

    def make(m):
        return (centers[rng.integers(0, nc, m)] + 0.25 / np.sqrt(d) * rng.standard_normal((m, d))).astype(np.float32)

    X, Q = make(n), make(nq)

    class _Emb:          # texts are row numbers
        def embed_documents(self, texts):
            return [X[int(t)].tolist() for t in texts]

        def embed_query(self, text):
            return Q[int(text)].tolist()

    small = lambda dim, metric, dev: FlatIndexF16(dim, metric=metric, device=dev, scan="q8", cand_cap=4096)   # noqa: E731
    store = HipFlatVectorStore(_Emb(), engine_factory=small)
    store.index = store._make_engine(d)
    store.index.add(X)
    cold = FlatIndexF16(d, scan="q8", cand_cap=4096)
    cold.add(X)
    cold.search_device(torch.from_numpy(Q).cuda(), k, repair=False)
    assert (cold.last_status.cpu().numpy() & 0x100).any(), "the test is meant to overflow a cold index"
    grown = store.warm_up(k)
    eng = store.index
    assert grown >= 1 and eng.cand_cap > 4096
    ids, sc = eng.search_device(torch.from_numpy(Q).cuda(), k, repair=False)          # first USER batch, first attempt only
    assert not (eng.last_status.cpu().numpy() & 0x100).any()
    rows, _ = oracle.ingest_f16(X)
    rI, rD, _ = oracle.flat_search_f16(rows, oracle.normalize_L2(Q), k)
    D, I = eng.search(Q, k)
    assert np.array_equal(I, rI) and np.array_equal(D.view(np.uint32), rD.view(np.uint32))
    assert store.warm_up(k) == 0                                                        # settled
    # the learnt capacity is part of what save_local writes
    store.docstore, store.index_to_docstore_id = {}, {}
    store.save_local(str(tmp_path), "idx")
    back = HipFlatVectorStore.load_local(str(tmp_path), _Emb(), "idx", engine_factory=small)
    assert back.index.cand_cap == eng.cand_cap > 4096
    D2, I2 = back.index.search(Q, k)
    assert np.array_equal(I2, rI) and np.array_equal(D2.view(np.uint32), rD.view(np.uint32))


@pytest.mark.parametrize("storage", ["f16", "f8", "f32"])
def test_warm_up_is_a_no_op_on_a_corpus_the_default_capacity_fits(storage):
    """Isotropic rows: warm_up() searches a sample of the stored rows (decoded from whatever the storage is), finds nothing
    to learn, and every sampled row finds itself first."""
    import torch

    from rag_arc_amd.hip.engine import FlatIndexF16

    rng = np.random.default_rng(3)
    X = rng.standard_normal((50_000, 256)).astype(np.float32)
    idx = FlatIndexF16(256, storage=storage)
    idx.add(X)
    cap = idx.cand_cap
    assert idx.warm_up(k=10, n_queries=16) == 0 and idx.cand_cap == cap
    pick = torch.linspace(0, idx.ntotal - 1, 16, device="cuda").long().cpu().numpy()
    D, I = idx.search(X[pick], 1)
    assert np.array_equal(I[:, 0], pick)
    assert FlatIndexF16(256, storage=storage).warm_up() == 0          # empty index
