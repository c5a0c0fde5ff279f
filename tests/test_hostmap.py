"""The native host side of an answer (csrc/hostmap.c) against the python loops it replaces, and the row-indexed docstores.

The loops restated here are the reference's: the hit loop of similarity_search_by_vector_with_score
(VectorStore_Faiss.py:265-272) and the bookkeeping of RRFusion.fuse (core/utils/Fusion.py:51-64).  No GPU needed."""
import pickle

import numpy as np
import pytest

from rag_arc_amd.core.utils.data_model import Document
from rag_arc_amd.encapsulation.database.vector_db.docstore import ColumnarDocstore, rows_from_dicts
from rag_arc_amd.encapsulation.database.vector_db.hip_flat import HipFlatVectorStore
from rag_arc_amd.hip import hostmap


def _hit_loop(docstore, index_to_docstore_id, scores, rows):
    out = []
    for score, row in zip(scores, rows):          # VectorStore_Faiss.py:266-271
        if row == -1:
            continue
        out.append((docstore[index_to_docstore_id[int(row)]], float(score)))
    return out


def _corpus(n):
    docs = [Document(content=f"text {i % 7 if i % 5 == 0 else i}", metadata={"i": i}, id=f"id-{i}") for i in range(n)]
    return docs, {d.id: d for d in docs}, {i: d.id for i, d in enumerate(docs)}


def test_rows_to_pairs_and_docs_equal_the_reference_hit_loop():
    H = hostmap.load()
    docs, docstore, i2d = _corpus(1000)
    rng = np.random.default_rng(0)
    rows = rng.integers(0, 1000, (37, 25)).astype(np.int64)
    rows[3, 20:] = -1                     # a short answer
    rows[9, :] = -1                       # an empty one
    scores = rng.standard_normal((37, 25)).astype(np.float32)
    want = [_hit_loop(docstore, i2d, s, r) for s, r in zip(scores, rows)]
    got = H.rows_to_pairs(docs, rows, scores, 37, 25)
    assert got == want
    assert all(a[0] is b[0] for ga, wa in zip(got, want) for a, b in zip(ga, wa))      # the very objects
    assert all(type(p[1]) is float for one in got for p in one)
    assert H.rows_to_docs(docs, rows, 37, 25) == [[d for d, _ in one] for one in want]
    # a plain tuple of Documents is a sequence like any other (only a 6-tuple that starts with a class is read as columns)
    assert H.rows_to_docs(tuple(docs), rows, 37, 25) == H.rows_to_docs(docs, rows, 37, 25)
    assert H.rows_to_docs(tuple(docs[:6]), np.array([[5, 0]], dtype=np.int64), 1, 2) == [[docs[5], docs[0]]]
    # any sequence will do (the columnar docstore is one), and rows may carry a base
    col = ColumnarDocstore.from_texts([d.content for d in docs], [d.id for d in docs], [d.metadata for d in docs])
    assert H.rows_to_docs(col, rows, 37, 25) == [[d for d, _ in one] for one in want]
    # ... and the columnar docstore's native form: Documents built in C equal Document(content=, metadata=, id=)
    got_c = H.rows_to_pairs(col.columns(), rows, scores, 37, 25)
    assert got_c == want and all(type(d) is Document and d.__dict__ == w.__dict__ and list(d.__dict__) == list(w.__dict__)
                                 for gc_, wc in zip(got_c, want) for (d, _), (w, _) in zip(gc_, wc))
    plain = ColumnarDocstore.decimal(1000)
    docs_p = H.rows_to_docs(plain.columns(), rows, 37, 25)
    assert docs_p == [[plain[int(r)] for r in row if r != -1] for row in rows]
    assert docs_p[0][0].metadata == {} and docs_p[0][0].metadata is not docs_p[0][1].metadata and docs_p[0][0].id is docs_p[0][0].content
    with pytest.raises(IndexError):
        H.rows_to_docs(plain.columns(), np.array([[1000]], dtype=np.int64), 1, 1)
    assert H.rows_to_docs(docs, rows + np.where(rows >= 0, 500, 0), 37, 25, 500) == [[d for d, _ in one] for one in want]
    with pytest.raises(IndexError):
        H.rows_to_docs(docs, np.array([[1000]], dtype=np.int64), 1, 1)
    with pytest.raises(ValueError):
        H.rows_to_docs(docs, rows, 38, 25)            # buffer too small for the shape


def _rrf_bookkeeping(results):
    """first-seen keys + last Document per content, as RRFusion.fuse keeps them (Fusion.py:51-64)."""
    key_of, doc_of, table = {}, {}, []
    for one in results:
        row = []
        for doc in one:
            key = key_of.setdefault(doc.content, len(key_of))
            doc_of[key] = doc
            row.append(key)
        table.append(row)
    return table, [doc_of[k] for k in range(len(key_of))]


def test_rrf_tables_and_pick_docs_equal_the_python_bookkeeping():
    H = hostmap.load()
    docs, _, _ = _corpus(400)
    twins = [Document(content=d.content, metadata={"twin": True}, id="t" + d.id) for d in docs]   # same content, other object
    rng = np.random.default_rng(1)
    batch = []
    for q in range(23):
        a = [docs[i] for i in rng.integers(0, 400, rng.integers(0, 30))]
        b = [twins[i] for i in rng.integers(0, 400, rng.integers(0, 30))]
        batch.append([a, b] if q % 4 else [a])
    keys_b, lens_b, by_key = H.rrf_tables(batch, 2, 30)
    keys = np.frombuffer(keys_b, dtype=np.int64).reshape(23, 2, 30)
    lens = np.frombuffer(lens_b, dtype=np.int32).reshape(23, 2)
    for q, results in enumerate(batch):
        table, last = _rrf_bookkeeping(results)
        assert [len(r) for r in results] == lens[q, : len(results)].tolist()
        for l, row in enumerate(table):
            assert keys[q, l, : len(row)].tolist() == row
        assert len(by_key[q]) == len(last) and all(a is b for a, b in zip(by_key[q], last))
    fused = np.zeros((23, 5), dtype=np.int64)
    counts = np.array([min(5, len(b)) for b in by_key], dtype=np.int32)
    for q in range(23):
        fused[q, : counts[q]] = np.arange(counts[q])[::-1]
    picked = H.pick_docs(by_key, fused, counts, 5)
    assert [[d is by_key[q][k] for d, k in zip(picked[q], fused[q])] for q in range(23)] == [[True] * c for c in counts]
    with pytest.raises(IndexError):
        H.pick_docs(by_key, np.full((23, 5), 10 ** 6, dtype=np.int64), np.full(23, 1, dtype=np.int32), 5)


def test_columnar_docstore():
    col = ColumnarDocstore.decimal(1200)
    assert len(col) == 1200 and col[7] == Document(content="0007", metadata={}, id="0007") and col[-1].id == "1199"
    assert col.row_of("0420") == 420 and col.row_of("nope") is None
    back = pickle.loads(pickle.dumps(col))
    assert [back[i] for i in (0, 999, 1199)] == [col[i] for i in (0, 999, 1199)]
    texts = ["alpha", "βeta γ", "", "delta"]
    col2 = ColumnarDocstore.from_texts(texts, ["a", "b", "c", "d"], [{"n": i} for i in range(4)])
    assert [col2[i].content for i in range(4)] == texts and col2[1].id == "b" and col2[3].metadata == {"n": 3}
    with pytest.raises(IndexError):
        col2[4]


class _NoIndex:
    ntotal = 0


def test_row_list_follows_the_dicts():
    """The row list is the two dicts of the reference in another form: a re-used id answers with the NEW document on its
    earlier rows too (docstore[id] is overwritten, VectorStore_Faiss.py:206), and dicts assigned from outside are picked up."""
    store = HipFlatVectorStore(embedding=None)
    store._remember(0, ["a", "b"], [{}, {}], ["x", "y"])
    store._remember(2, ["c"], [{}], ["x"])                       # id x again
    rows = store._docs_by_row()
    assert [d.content for d in rows] == ["c", "b", "c"] and rows == rows_from_dicts(store.docstore, store.index_to_docstore_id)
    store._remember(3, ["d"], [{}], ["z"])
    assert [d.content for d in store._docs_by_row()] == ["c", "b", "c", "d"]
    store.docstore = {"k": Document(content="only", metadata={}, id="k")}
    store.index_to_docstore_id = {0: "k"}
    assert [d.content for d in store._docs_by_row()] == ["only"]


def test_hostmap_is_reference_count_neutral():
    """A C extension that leaks or over-releases a reference shows up as a slow memory leak or a crash in somebody's server:
    after the results are dropped every Document, list and buffer is back at the count it started with — on the error paths
    too."""
    import gc
    import sys

    H = hostmap.load()
    docs, _, _ = _corpus(300)
    rng = np.random.default_rng(3)
    rows = rng.integers(0, 300, (16, 20)).astype(np.int64)
    rows[2, 15:] = -1
    scores = rng.standard_normal((16, 20)).astype(np.float32)
    col = ColumnarDocstore.from_texts([d.content for d in docs], [d.id for d in docs], [d.metadata for d in docs])
    cols = col.columns()
    batch = [[[docs[i] for i in rng.integers(0, 300, 12)], [docs[i] for i in rng.integers(0, 300, 9)]] for _ in range(7)]
    fused = np.zeros((7, 4), dtype=np.int64)
    counts = np.full(7, 3, dtype=np.int32)
    watched = [docs[int(r)] for r in rows[0, :5]] + [docs, rows, scores, col.text_blob, col.text_off, cols, batch, batch[0][0]]

    def snapshot():
        gc.collect()
        return [sys.getrefcount(o) for o in watched]

    before = snapshot()
    for _ in range(50):
        a = H.rows_to_pairs(docs, rows, scores, 16, 20)
        b = H.rows_to_docs(docs, rows, 16, 20)
        c = H.rows_to_pairs(cols, rows, scores, 16, 20)
        d = H.rows_to_docs(col, rows, 16, 20)
        keys_b, lens_b, by_key = H.rrf_tables(batch, 2, 12)
        e = H.pick_docs(by_key, fused, counts, 4)
        for bad in (lambda: H.rows_to_docs(docs, np.array([[999]], dtype=np.int64), 1, 1),
                    lambda: H.rows_to_pairs(cols, np.array([[999]], dtype=np.int64), np.zeros((1, 1), np.float32), 1, 1),
                    lambda: H.rows_to_docs(docs, rows, 17, 20),
                    lambda: H.rrf_tables(batch, 1, 12),
                    lambda: H.pick_docs(by_key, np.full((7, 4), 99, dtype=np.int64), counts, 4)):
            with pytest.raises((IndexError, ValueError)):
                bad()
        del a, b, c, d, keys_b, lens_b, by_key, e
    assert snapshot() == before
