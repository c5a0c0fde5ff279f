"""The rest of the VectorStore / VectorStoreRetriever surface against what the REFERENCE's own classes did with the same
script of calls (tests/golden/surface.json, produced by tests/golden/make_golden.py from
encapsulation/database/vector_db/VectorStoreBase.py:92-232, :444-627 and core/retrieval/dense.py:219-380): the
`search` / `asearch` dispatchers, adelete, aget_by_ids, asimilarity_search_by_vector, amax_marginal_relevance_search(_by_vector),
add_documents / aadd_documents, from_documents / afrom_documents / afrom_texts, and the retriever's passthroughs, info, name,
repr and update_search_params.  Same script, same recording store (tests/helpers.py) — only the classes under it differ.

Where the reference's async plumbing kills the call with a TypeError of its own (keyword arguments handed to
run_in_executor; add_texts' keyword-only `ids` passed positionally) the mirror answers what the SYNC twin answers; those
cases are listed here by name, so a new divergence cannot hide among them."""
import pytest

from rag_arc_amd.core.retrieval import VectorStoreRetriever
from rag_arc_amd.core.utils.data_model import Document
from rag_arc_amd.encapsulation.database.vector_db.base import VectorStore
from tests.helpers import golden, make_recording_store, run_surface_ops

# the reference raises TypeError from its own executor plumbing; value = what the mirror returns instead
REFERENCE_PLUMBING_ERRORS = {
    "asearch:similarity_extra_kw": ["doc0", "doc1"],
    "adelete:kw": True,
    "asimilarity_search_by_vector:extra_kw": ["doc0", "doc1"],
    "aadd_documents:kw_ids": ["x", "y"],
    "aadd_texts": ["1", "2"],
    "afrom_documents:some_ids": [["from_texts", ["t0", "t1"], "EMB", [{"a": 1}, {}], ["i0", None]], ["init_kw", {}]],
    "afrom_texts:ids": [["from_texts", ["p", "q"], "EMB", [{"m": 1}, {}], ["1", "2"]], ["init_kw", {}]],
    "afrom_texts:extra_kw": [["from_texts", ["p"], "EMB", None, None], ["init_kw", {}]],
}


@pytest.fixture(scope="module")
def replay():
    scored = [(Document(content=f"doc{i}", metadata={}, id=str(i)), s)
              for i, s in enumerate([0.9, 0.75, 0.5, 0.25, 0.1, 0.05, -0.2, 0.0])]
    got = run_surface_ops(Document, make_recording_store(VectorStore), VectorStoreRetriever, scored)
    return {c["op"]: c for c in got}


def test_the_script_has_not_drifted(replay):
    assert [c["op"] for c in golden("surface.json")] == list(replay)


@pytest.mark.parametrize("case", golden("surface.json"), ids=lambda c: c["op"])
def test_same_outcome_as_the_reference(replay, case):
    mine = replay[case["op"]]
    if case["op"] in REFERENCE_PLUMBING_ERRORS:
        assert case["error"] == "TypeError" and case["calls"] == []          # the reference never reached the store
        assert mine["error"] is None and mine["result"] == REFERENCE_PLUMBING_ERRORS[case["op"]]
        return
    assert mine["error"] == case["error"]
    assert mine["result"] == case["result"]
    assert mine["calls"] == case["calls"]                                    # same calls into the store, same arguments


def test_every_plumbing_exception_is_a_reference_typeerror():
    errs = {c["op"] for c in golden("surface.json") if c["error"] == "TypeError"}
    assert errs == set(REFERENCE_PLUMBING_ERRORS)


def test_as_retriever_works_here():
    store = make_recording_store(VectorStore)([])
    r = store.as_retriever(search_type="mmr", search_kwargs={"k": 2})
    assert isinstance(r, VectorStoreRetriever) and r.search_type == "mmr" and r.vectorstore is store
