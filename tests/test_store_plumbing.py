"""BASELINE.json config 1 (plumbing): 384-d, 10k docs, cosine top-10 through the registered-backend
surface (Embeddings -> VectorStore -> VectorStoreRetriever -> MultiPathRetriever), with the index
arithmetic supplied by the CPU oracle through the store's engine hook — no GPU involved."""
import numpy as np
import pytest

from rag_arc_amd.core.retrieval import MultiPathRetriever, VectorStoreRetriever
from rag_arc_amd.encapsulation.database.vector_db import HipFlatVectorStore
from tests.helpers import HashEmbeddings, OracleFusion, OracleIndex, ScriptedRetriever


def _engine(dim, metric, device):
    return OracleIndex(dim, metric)


@pytest.fixture(scope="module")
def store():
    emb = HashEmbeddings(384)
    texts = [f"document number {i}" for i in range(10_000)]
    return HipFlatVectorStore.from_texts(texts, emb, ids=[str(i) for i in range(10_000)], engine_factory=_engine), texts


def test_config1_cosine_top10(store, oracle):
    st, texts = store
    emb = st.embedding
    q = "document number 4242"
    docs = VectorStoreRetriever(st, search_kwargs={"k": 10}).invoke(q)
    assert len(docs) == 10 and docs[0].content == q       # a text retrieves itself first
    # same answer as the oracle asked directly
    X = np.array(emb.embed_documents(texts), dtype=np.float32)
    rows, _ = oracle.ingest_f16(X)
    ids, sc, _ = oracle.flat_search_f16(rows, oracle.normalize_L2(np.array([emb.embed_query(q)], np.float32)), 10)
    assert [d.id for d in docs] == [str(i) for i in ids[0]]
    pairs = st.similarity_search_with_score(q, k=10)
    assert [s for _, s in pairs] == [float(s) for s in sc[0]]
    assert abs(pairs[0][1] - 1.0) < 2e-3


def test_store_semantics_follow_the_reference(store):
    st, _ = store
    assert st.similarity_search("document number 1", k=10 ** 9)[:1][0].content == "document number 1"  # k=min(k,ntotal)
    assert st.get_by_ids(["5", "nope"])[0].content == "document number 5"
    rel = st.similarity_search_with_relevance_scores("document number 7", k=3)
    assert rel[0][1] == 1.0 - st.similarity_search_with_score("document number 7", k=3)[0][1]  # the 1 - cosine quirk
    assert len(st.max_marginal_relevance_search("document number 9", k=3, fetch_k=8)) == 3
    empty = HipFlatVectorStore(HashEmbeddings(16), engine_factory=_engine)
    assert empty.similarity_search("x") == [] and empty.add_texts([]) == []
    with pytest.raises(ValueError):
        empty.add_texts(["a", "b"], ids=["only-one"])
    with pytest.raises(ValueError):
        HipFlatVectorStore(HashEmbeddings(16), index_type="ivf")


def test_delete_rebuilds_and_persistence_round_trips(tmp_path):
    emb = HashEmbeddings(64)
    st = HipFlatVectorStore.from_texts([f"t{i}" for i in range(50)], emb, ids=[f"id{i}" for i in range(50)],
                                       engine_factory=_engine)
    assert st.delete(["id3", "missing"]) is False and st.ntotal == 50
    assert st.delete(["id3", "id4"]) is True and st.ntotal == 48
    assert "t3" not in [d.content for d in st.similarity_search("t3", k=48)]
    before = st.similarity_search_with_score("t10", k=5)
    st.save_local(str(tmp_path))
    st2 = HipFlatVectorStore.load_local(str(tmp_path), emb, engine_factory=_engine)
    after = st2.similarity_search_with_score("t10", k=5)
    assert [(d.id, s) for d, s in before] == [(d.id, s) for d, s in after]
    assert st.delete() is True and st.ntotal == 0 and st.similarity_search("t1") == []
    # an emptied store saved into the same folder must not leave the old shard file behind (the reference rewrites
    # its index file on every save, VectorStore_Faiss.py:438): load_local then yields an empty, searchable store
    st.save_local(str(tmp_path))
    st3 = HipFlatVectorStore.load_local(str(tmp_path), emb, engine_factory=_engine)
    assert st3.ntotal == 0 and st3.similarity_search("t10") == []


def test_multipath_over_dense_and_lexical_lists(store):
    st, _ = store
    dense = VectorStoreRetriever(st)
    lexical = ScriptedRetriever(st.get_by_ids(["17", "4242", "99"]))       # the "supplied BM25 rank list"
    out = MultiPathRetriever([dense, lexical], fusion_method=OracleFusion(), top_k_per_retriever=20).invoke(
        "document number 4242", top_k=5)
    assert out[0].content == "document number 4242" and len(out) == 5
    assert lexical.seen[0]["kwargs"] == {"top_k": 5, "k": 20}


def test_async_entry_points(store):
    import asyncio

    st, _ = store
    docs = asyncio.run(VectorStoreRetriever(st, search_kwargs={"k": 3}).ainvoke("document number 11"))
    assert [d.content for d in docs][:1] == ["document number 11"] and len(docs) == 3
