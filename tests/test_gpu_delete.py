"""delete(ids) by device-side compaction: the reference clears the index and embeds every surviving
text again (VectorStore_Faiss.py:374-415); here the surviving rows move down over the holes in HBM (rarc_compact_rows) and
the docstore follows.

* engine: after remove_rows the stored rows (and fp8 scales, the fp32 index's image, the int8 shadow) equal the original
  rows minus the holes bit for bit, for every storage format, holes at the ends / in runs / across chunk boundaries;
  searches equal the ORACLE on the kept set; the encoder is not called;
* store: delete semantics of the reference (None clears, unknown id -> False and nothing changes, [] -> True), results
  after a delete equal a store built from the kept documents; add after delete; the async twin; save / load after delete;
* an id that names several rows takes the reference's rebuild (one row per id survives);
* 1M rows x 768, 1k ids: under 50 ms;
* the sharded store as two gloo ranks on one device equals the single store after the same deletes.
"""
import asyncio
import os
import subprocess
import sys
import time

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def hip():
    import torch

    assert torch.cuda.is_available(), "gpu tests need a ROCm device"
    from rag_arc_amd.hip import engine

    return engine


def _bits(t):
    import torch

    return t.view(torch.int32) if t.dtype == torch.float32 else t


@pytest.mark.parametrize("storage,dim,growable", [("f16", 384, False), ("f16", 200, True), ("f8", 520, False), ("f32", 130, True)])
def test_remove_rows_is_the_original_minus_the_holes_bit_for_bit(hip, oracle, storage, dim, growable):
    import torch

    rng = np.random.default_rng(5)
    n = 50_017
    X = rng.standard_normal((n, dim)).astype(np.float32)
    Q = rng.standard_normal((10, dim)).astype(np.float32)
    idx = hip.FlatIndexF16(dim, metric="cosine", storage=storage, growable=growable)
    idx.COMPACT_TMP_BYTES = 3 << 20          # many chunks: the chunk-by-chunk ordering is what keeps this correct
    idx.add(X)
    before = idx.rows.clone()
    scales = idx.row_scales.clone() if storage == "f8" else None
    image = idx._image16[:n].clone() if storage == "f32" else None
    holes = np.unique(np.concatenate([[0, 1, 2, n - 1, n - 2], np.arange(7000, 7300), rng.integers(0, n, 900)]))
    keep = np.setdiff1d(np.arange(n), holes)
    assert idx.remove_rows(holes[::-1]) == holes.size and idx.ntotal == keep.size
    kt = torch.from_numpy(keep).cuda()
    assert torch.equal(_bits(idx.rows), _bits(before[kt]))
    if scales is not None:
        assert torch.equal(_bits(idx.row_scales), _bits(scales[kt]))
    if image is not None:
        assert torch.equal(idx._image16[: keep.size], image[kt])
    assert int(idx._rows[keep.size: min(n, idx._rows.shape[0])].view(torch.uint8).max().item()) == 0      # the vacated tail is zero
    # the oracle on the kept vectors (ids = positions among the survivors)
    qn = oracle.normalize_L2(Q)
    if storage == "f16":
        ref_I, ref_D, _ = oracle.flat_search_f16(oracle.ingest_f16(X[keep])[0], qn, 30)
    elif storage == "f8":
        r8, s8, _ = oracle.ingest_f8(X[keep])
        ref_I, ref_D = oracle.flat_search_f8(r8, s8, qn, 30)[:2]
    else:
        ref_I, ref_D = oracle.flat_search_f32(oracle.ingest_f32(X[keep])[0], qn, 30)[:2]
    for scan in (("q8", "mfma16") if storage == "f16" and idx.d_pad <= 768 else ("auto",)):
        idx.scan = scan
        D, I = idx.search(Q, 30)
        assert np.array_equal(I, ref_I) and np.array_equal(D.view(np.uint32), ref_D.view(np.uint32)), (storage, scan)
    idx.scan = "auto"
    idx.add(X[:100])                          # an index that lost rows keeps growing
    assert idx.ntotal == keep.size + 100
    probe = int(keep[3])                      # a kept row that was just added a second time: the earlier copy wins the tie
    assert probe < 100 and idx.search(X[probe:probe + 1], 2)[1][0].tolist() == [3, keep.size + probe]
    assert idx.remove_rows(np.arange(idx.ntotal)) == keep.size + 100 and idx.ntotal == 0
    assert idx.search(Q, 3)[1].tolist() == [[-1] * 3] * 10


def test_shadow_image_is_compacted_too(hip):
    rng = np.random.default_rng(6)
    X = rng.standard_normal((20_000, 256)).astype(np.float32)
    a = hip.FlatIndexF16(256, metric="cosine", shadow=True, scan="q8")
    a.add(X)
    holes = rng.integers(0, 20_000, 500)
    a.remove_rows(holes)
    b = hip.FlatIndexF16(256, metric="cosine", shadow=True, scan="q8")
    b.add(X[np.setdiff1d(np.arange(20_000), holes)])
    import torch

    assert torch.equal(a._shadow[: a.ntotal], b._shadow[: b.ntotal])
    Da, Ia = a.search(X[:8], 20)
    Db, Ib = b.search(X[:8], 20)
    assert np.array_equal(Ia, Ib) and np.array_equal(Da.view(np.uint32), Db.view(np.uint32))


class _CountingEmbeddings:
    def __init__(self, dim):
        from tests.helpers import HashEmbeddings

        self.inner, self.calls = HashEmbeddings(dim), 0

    def embed_documents(self, texts):
        self.calls += 1
        return self.inner.embed_documents(texts)

    def embed_query(self, text):
        return self.inner.embed_query(text)


def test_store_delete_follows_the_reference_without_the_encoder(hip, tmp_path):
    from rag_arc_amd.encapsulation.database.vector_db import HipFlatVectorStore

    emb = _CountingEmbeddings(160)
    texts = [f"passage {i} about topic {i % 17}" for i in range(6000)]
    ids = [f"p{i}" for i in range(6000)]
    store = HipFlatVectorStore.from_texts(texts, emb, ids=ids, metadatas=[{"n": i} for i in range(6000)])
    calls = emb.calls
    assert store.delete(["p3", "nope"]) is False and store.ntotal == 6000           # an unknown id: nothing changes
    assert store.delete([]) is True and store.ntotal == 6000
    gone = ["p0", "p5999", "p77", "p77"] + [f"p{i}" for i in range(1000, 1200)]
    assert store.delete(gone) is True
    kept = [i for i in range(6000) if f"p{i}" not in set(gone)]
    assert store.ntotal == len(kept) == 6000 - 203 and emb.calls == calls, "delete called the encoder"
    assert list(store.docstore) == [f"p{i}" for i in kept]
    assert store.index_to_docstore_id == {r: f"p{i}" for r, i in enumerate(kept)}
    assert store.get_by_ids(["p77", "p78"]) == [store.docstore["p78"]]
    # same answers as a store built from the kept documents (the reference's rebuild)
    fresh = HipFlatVectorStore.from_texts([texts[i] for i in kept], emb, ids=[f"p{i}" for i in kept],
                                          metadatas=[{"n": i} for i in kept])
    for q in ("passage 1100 about topic 12", "passage 42 about topic 8", texts[5998]):
        a = [(d.id, d.metadata, s) for d, s in store.similarity_search_with_score(q, k=12)]
        assert a == [(d.id, d.metadata, s) for d, s in fresh.similarity_search_with_score(q, k=12)]
        assert not any(i in set(gone) for i, _, _ in a)
    assert [[d.id for d in one] for one in store.batch_similarity_search(texts[2000:2300], k=5)] == \
        [[d.id for d in one] for one in fresh.batch_similarity_search(texts[2000:2300], k=5)]
    # keeps growing, deletes again (slots: rows renumbered twice), the async twin, persistence
    store.add_texts(["a late passage"], ids=["late"])
    assert store.similarity_search("a late passage", k=1)[0].id == "late"
    assert asyncio.run(store.adelete(["late", "p3"])) is True and store.ntotal == len(kept) - 1
    assert store.similarity_search(texts[4], k=1)[0].id == "p4" and "p3" not in store.docstore
    store.save_local(str(tmp_path / "after"))
    again = HipFlatVectorStore.load_local(str(tmp_path / "after"), emb)
    assert [(d.id, s) for d, s in again.similarity_search_with_score(texts[4321], k=9)] == \
        [(d.id, s) for d, s in store.similarity_search_with_score(texts[4321], k=9)]
    assert again.delete(["p4321"]) is True and again.similarity_search(texts[4321], k=1)[0].id != "p4321"   # slots rebuilt after a load
    assert store.delete(None) is True and store.ntotal == 0 and store.docstore == {} and store.index_to_docstore_id == {}
    assert store.similarity_search("anything", k=3) == []


def test_an_id_that_names_several_rows_takes_the_reference_rebuild(hip):
    """The reference rebuilds from its docstore — one document per ID (the latest), at the place of the id's first use —
    so rows that share an id collapse into one (VectorStore_Faiss.py:390-413).  Compaction cannot reproduce that."""
    from rag_arc_amd.encapsulation.database.vector_db import HipFlatVectorStore

    emb = _CountingEmbeddings(64)
    store = HipFlatVectorStore.from_texts(["one", "two", "three"], emb, ids=["a", "b", "c"])
    store.add_texts(["one again"], ids=["a"])
    assert store.ntotal == 4
    calls = emb.calls
    assert store.delete(["b"]) is True
    assert emb.calls == calls + 1 and store.ntotal == 2 and list(store.docstore) == ["a", "c"]
    assert [d.content for d in store.similarity_search("one again", k=2)] == ["one again", "three"]


def test_delete_1k_of_1m_rows_under_50_ms(hip):
    import torch

    from rag_arc_amd.core.utils.data_model import Document
    from rag_arc_amd.encapsulation.database.vector_db import HipFlatVectorStore
    from rag_arc_amd.hip import binding as B

    n, d = 1_000_000, 768
    lib = B.load_library()
    rows = torch.empty((n, d), dtype=torch.float16, device="cuda")
    B.check(lib.rarc_synth_rows_f16(rows.data_ptr(), d, d, 0, n, 1234, 0))
    idx = hip.FlatIndexF16(d, metric="cosine")
    idx.add_rows_f16(rows, 1.001)
    store = HipFlatVectorStore(embedding=None).adopt(idx, [Document(content=str(i), metadata={}, id=str(i)) for i in range(n)])
    q = torch.empty((4, d), dtype=torch.float32, device="cuda")
    B.check(lib.rarc_synth_rows_f32(q.data_ptr(), d, d, 0, 4, 4321, 0))
    want = idx.search(q, 50)[1]
    store.delete(["999999"])                  # the first delete of an adopted store builds its id -> slot table
    rng = np.random.default_rng(3)
    ids = [str(i) for i in rng.choice(n - 1, 1000, replace=False)]
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    assert store.delete(ids) is True
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) * 1e3
    print(f"delete of 1000 ids from 1M x 768 rows: {ms:.1f} ms")
    assert store.ntotal == n - 1001 and ms < 50.0
    gone = np.array(sorted(int(i) for i in ids))
    got = idx.search(q, 50)[1]
    got_ids = np.array([[int(store._row_docs[r].id) for r in row] for row in got])
    for qi in range(4):      # the old answer without the deleted ids is a prefix-compatible subsequence of the new one
        old_kept = [i for i in want[qi].tolist() if i not in set(gone.tolist()) and i != 999999]
        assert got_ids[qi, : len(old_kept)].tolist() == old_kept


def test_two_rank_sharded_delete_equals_single_store(tmp_path):
    worker = os.path.join(ROOT, "tests", "_delete_rank_worker.py")
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29671", PYTHONPATH=ROOT)
    procs = [subprocess.Popen([sys.executable, worker, str(r), "2", str(tmp_path)], env=env, stdout=subprocess.PIPE,
                              stderr=subprocess.PIPE, text=True) for r in range(2)]
    outs = [p.communicate(timeout=600) for p in procs]
    for p, (so, se) in zip(procs, outs):
        assert p.returncode == 0, se[-3000:]
    assert "DELETE_OK" in outs[0][0]


def test_searches_stay_consistent_while_another_thread_deletes_and_adds(hip):
    """A delete renumbers every later row; an answer whose row numbers were taken before it must not be turned into Documents
    after it.  Four threads search for texts that are never deleted (top-1 must be that very text, score ~ 1) while the main
    thread deletes and adds around them — the store's readers-writer guard is what keeps each answer whole."""
    import threading

    from rag_arc_amd.encapsulation.database.vector_db import HipFlatVectorStore
    from tests.helpers import HashEmbeddings

    emb = HashEmbeddings(128)
    texts = [f"steady text {i}" for i in range(4000)]
    store = HipFlatVectorStore.from_texts(texts, emb, ids=[f"s{i}" for i in range(4000)])
    kept = list(range(3000, 4000))                 # rows behind everything that gets deleted: renumbered by every delete
    stop, errors, done = threading.Event(), [], [0]

    def searcher(seed):
        rng = np.random.default_rng(seed)
        while not stop.is_set():
            j = int(rng.choice(kept))
            got = store.similarity_search_with_score(texts[j], k=1)
            if not got or got[0][0].id != f"s{j}" or got[0][0].content != texts[j] or abs(got[0][1] - 1.0) > 1e-3:
                errors.append((j, got[0][0].id if got else None))
                return
            done[0] += 1

    threads = [threading.Thread(target=searcher, args=(s,)) for s in range(4)]
    for t in threads:
        t.start()
    for step in range(30):
        assert store.delete([f"s{step * 50 + i}" for i in range(50)]) is True
        store.add_texts([f"churn {step} {i}" for i in range(20)], ids=[f"c{step}_{i}" for i in range(20)])
        time.sleep(0.02)                           # (writers have priority: leave the searchers a window per step)
    stop.set()
    for t in threads:
        t.join()
    assert not errors, errors[:3]
    assert done[0] > 30 and store.ntotal == 4000 - 1500 + 600
