"""Host-side mirror of the reference's retriever / store control flow vs goldens recorded from the
reference itself (core/retrieval/dense.py, mutipath.py, VectorStoreBase.py)."""
import asyncio
import contextlib
import io
import warnings

import pytest

from rag_arc_amd.core.retrieval import MultiPathRetriever, VectorStoreRetriever
from rag_arc_amd.core.utils.data_model import Document
from tests.helpers import OracleFusion, ScriptedRetriever, ScriptedStore, golden, scripted_scored, unhex


def _docs(names):
    return [Document(content=n, metadata={"n": n}, id=f"id-{n}") for n in names]


@pytest.mark.parametrize("case", golden("dense.json"), ids=lambda c: f"{c['search_type']}-{c['search_kwargs']}-{c['async']}")
def test_vectorstore_retriever_matches_reference(case):
    store = ScriptedStore(scripted_scored())
    if "init_error" in case:
        with pytest.raises(ValueError):
            VectorStoreRetriever(store, search_type=case["search_type"], search_kwargs=case["search_kwargs"])
        return
    r = VectorStoreRetriever(store, search_type=case["search_type"], search_kwargs=case["search_kwargs"])
    with warnings.catch_warnings(record=True) as w:
        warnings.simplefilter("always")
        if case["async"]:
            out = asyncio.run(r.ainvoke("q", **case["invoke_kwargs"]))
        else:
            out = r.invoke("q", **case["invoke_kwargs"])
    assert [d.content for d in out] == case["out"]
    assert store.calls == case["calls"]          # same store method, same k, same leaked kwargs
    assert len(w) == case["n_warnings"]


@pytest.mark.parametrize("case", golden("multipath.json"), ids=lambda c: str(c["kw"]))
def test_multipath_retriever_matches_reference(case):
    rs = [ScriptedRetriever(_docs(l), f) for l, f in zip(case["lists"], case["fail"])]
    mp = MultiPathRetriever(rs, fusion_method=OracleFusion(), top_k_per_retriever=case["per"])
    buf = io.StringIO()
    with contextlib.redirect_stdout(buf):
        out = mp.invoke("the query", **case["kw"])
    assert [d.content for d in out] == case["out"]
    assert [r.seen for r in rs] == case["seen"]
    assert len([l for l in buf.getvalue().splitlines() if l.strip()]) == case["printed_lines"]
    assert hasattr(mp, "search_kwargs") == case["has_search_kwargs"]


def test_relevance_threshold_quirk_matches_reference():
    for case in golden("relevance.json")["threshold"]:
        store = ScriptedStore(scripted_scored())
        with warnings.catch_warnings(record=True) as w:
            warnings.simplefilter("always")
            kw = {} if case["thr"] is None else {"score_threshold": case["thr"]}
            out = store.similarity_search_with_relevance_scores("q", k=8, **kw)
        assert [[d.content, s] for d, s in out] == [[c, unhex(h)] for c, h in case["out"]]
        assert len(w) == case["n_warnings"]


def test_multipath_management_helpers():
    a, b = ScriptedRetriever(_docs(["x"])), ScriptedRetriever(_docs(["y"]))
    mp = MultiPathRetriever([a], fusion_method=OracleFusion())
    mp.add_retriever(b)
    assert len(mp.retrievers) == 2
    mp.remove_retriever("ScriptedRetriever")
    assert mp.retrievers == [b]
    f2 = OracleFusion(k=1.0)
    mp.set_fusion_method(f2)
    assert mp.fusion_method is f2 and mp.get_name() == "MultiPathRetriever"


def test_encoder_and_reranker_configs_are_tagged_members_of_the_registry_unions():
    """The HIP encoder and the reranker are selectable from JSON like any other backend: tagged configs in the
    discriminated unions (pattern: framework/module_test.py:47-101 in the reference); wrong tags are rejected."""
    import pydantic
    import pytest

    from rag_arc_amd.config.modules import (HipBertEmbeddingsConfig, HipFlatVectorStoreConfig, HipLogitRerankerConfig,
                                            TableEmbeddingsConfig)

    vs = HipFlatVectorStoreConfig(**{"type": "hip_flat_vectorstore", "storage": "f32",
                                     "embedding": {"type": "hip_bert_embeddings", "weights_path": "w.safetensors",
                                                   "vocab_path": "vocab.txt", "num_heads": 12}})
    assert isinstance(vs.embedding, HipBertEmbeddingsConfig) and vs.embedding.normalize_embeddings is True
    vs2 = HipFlatVectorStoreConfig(**{"type": "hip_flat_vectorstore", "embedding": {"type": "table_embeddings", "path": "e.npz"}})
    assert isinstance(vs2.embedding, TableEmbeddingsConfig)
    with pytest.raises(pydantic.ValidationError):
        HipFlatVectorStoreConfig(**{"type": "hip_flat_vectorstore", "embedding": {"type": "no_such_provider", "path": "x"}})
    with pytest.raises(pydantic.ValidationError):
        HipBertEmbeddingsConfig(**{"type": "hip_bert_embeddings", "weights_path": "w.npz"})      # vocab_path, num_heads missing
    rr = HipLogitRerankerConfig(**{"type": "hip_logit_reranker", "logits_path": "l.npz"})
    assert rr.device == 0
    with pytest.raises(pydantic.ValidationError):
        HipLogitRerankerConfig(**{"type": "rrf", "logits_path": "l.npz"})


def test_table_logits_lookup(tmp_path):
    import numpy as np
    import pytest

    from rag_arc_amd.core.rerank.hip_reranker import TableLogits

    np.savez(tmp_path / "l.npz", queries=np.array(["q1", "q2"]), docs=np.array(["a", "b", "c"]),
             z_no=np.arange(6, dtype=np.float16).reshape(2, 3), z_yes=-np.arange(6, dtype=np.float16).reshape(2, 3))
    t = TableLogits.from_npz(str(tmp_path / "l.npz"))
    zn, zy = t("q2", ["c", "a"])
    assert zn.tolist() == [5.0, 3.0] and zy.tolist() == [-5.0, -3.0]
    with pytest.raises(KeyError):
        t("q3", ["a"])


def test_mmr_selection_matches_reference_recorded_picks():
    """_mmr_select (host mirror) against picks recorded from the reference's own function (VectorStore_Faiss.py:16-62):
    duplicates of the first pick, exact ties, k >= n, lambda 0 and 1."""
    import numpy as np

    from rag_arc_amd.core.utils.data_model import Document
    from rag_arc_amd.encapsulation.database.vector_db.hip_flat import _mmr_select

    for c in golden("mmr.json")["cases"]:
        E = [[unhex(v) for v in row] for row in c["emb_hex"]]
        q = [unhex(v) for v in c["query_hex"]]
        docs = [(Document(content=f"c{i}", metadata={}, id=str(i)), 0.0) for i in range(c["n"])]
        got = _mmr_select(docs, E, q, c["k"], c["lambda"])
        assert [int(d.id) for d in got] == c["picked"], (c["n"], c["k"], c["lambda"])
