"""Generate golden vectors by RUNNING the reference's own code (build container only).

    python tests/golden/make_golden.py        # needs /root/reference; writes tests/golden/*.json

The reference's `core/__init__.py` imports a package that does not exist (`rag_arc`), so `core`
is pre-seeded as a namespace whose submodules load unmodified from /root/reference.
Only DATA (inputs and the reference's outputs) is written; no reference source travels.

Pinned here:
  rrf.json        RRFusion.fuse                               core/utils/Fusion.py:45-76
  multipath.json  MultiPathRetriever._get_relevant_documents  core/retrieval/mutipath.py:37-93
  dense.json      VectorStoreRetriever (sync + async)         core/retrieval/dense.py:122-218
  surface.json    VectorStore.search / asearch, the async twins, from_documents / afrom_*; the retriever's passthroughs
                                                              VectorStoreBase.py:92-232, :444-627; core/retrieval/dense.py:219-380
  relevance.json  VectorStore.similarity_search_with_relevance_scores + score fns
                                                              VectorStoreBase.py:263-273, :347-392
  cosine.json     spliter.cosine_similarity (numpy branch)    core/file_management/chunker/spliter.py:307-332
  mmr.json        _mmr_select (the greedy MMR loop)           VectorStore_Faiss.py:16-62  (the module imports faiss,
                  absent here: an EMPTY module object named `faiss` is put in sys.modules so that the import statement
                  passes — the function under test never touches it)
  chunker.json    SemanticChunker.split_text / calculate_cosine_distances / cosine_similarity with a zero row
                                                              core/file_management/chunker/spliter.py:307-534
  cosine_pin_d384.npz, cosine_pin_d768.npz
                  the same function on 64 x 4096 fp16-representable vectors (inputs: tests/helpers.py:pin_inputs):
                  the reference's float64 cosine matrix — the tight pin of the flat-search arithmetic
"""
import asyncio
import json
import os
import random
import struct
import sys
import types
import warnings

REF = "/root/reference"
OUT = os.path.dirname(os.path.abspath(__file__))


def _seed_reference():
    sys.path.insert(0, REF)
    core = types.ModuleType("core")
    core.__path__ = [os.path.join(REF, "core")]
    sys.modules["core"] = core


def hexf(x: float) -> str:
    return struct.pack(">d", float(x)).hex()


def main():
    _seed_reference()
    import numpy as np
    from core.utils.data_model import Document
    from core.utils.Fusion import RRFusion, RetrievalResult
    from core.retrieval.base import BaseRetriever
    from core.retrieval.dense import VectorStoreRetriever
    from core.retrieval.mutipath import MultiPathRetriever
    from encapsulation.database.vector_db.VectorStoreBase import VectorStore

    rnd = random.Random(20250905)

    # ------------------------------------------------------------------ RRF
    def run_rrf(lists, k, top_k):
        res = [[RetrievalResult(document=Document(content=c, metadata={"src": li, "pos": pi}), score=1.0)
                for pi, c in enumerate(one)] for li, one in enumerate(lists)]
        fused = RRFusion(k=k).fuse(res, top_k)
        return {
            "lists": lists, "k": k, "top_k": top_k,
            "fused": [{"content": r.document.content, "score_hex": hexf(r.score), "rank": r.rank,
                       "doc_src": r.document.metadata["src"], "doc_pos": r.document.metadata["pos"]} for r in fused],
            "input_ranks_after": [[r.rank for r in one] for one in res],
        }

    cases = []
    cases.append(run_rrf([["a", "b", "c"], ["b", "a", "d"]], 60.0, 10))          # exact tie a/b
    cases.append(run_rrf([["a", "b"], []], 60.0, 5))                              # empty list
    cases.append(run_rrf([[], []], 60.0, 5))                                      # all empty
    cases.append(run_rrf([["x", "x", "y", "x"]], 60.0, 5))                        # in-list duplicates
    cases.append(run_rrf([["p", "q", "r", "s"]], 60.0, 2))                        # single list, truncation
    cases.append(run_rrf([["a", "b", "c"], ["c", "b", "a"], ["b", "c", "a"]], 1.0, 3))
    cases.append(run_rrf([["a"], ["b"], ["c"], ["d"]], 60.0, 10))                 # 4-way tie, insertion order
    cases.append(run_rrf([["a", "b", "c"], ["d", "e", "f"]], 0.5, 0))             # top_k = 0
    for trial in range(12):  # C3-shaped: 2 lists x 100 ids, ~30 % overlap, decimal-id contents
        n = rnd.choice([5, 20, 100, 100, 100])
        dense = rnd.sample(range(1000), n)
        overlap = rnd.sample(dense, max(1, int(0.3 * n)))
        rest = [x for x in rnd.sample(range(1000, 3000), n) if x not in dense][: n - len(overlap)]
        bm25 = overlap + rest
        rnd.shuffle(bm25)
        cases.append(run_rrf([[str(x) for x in dense], [str(x) for x in bm25]], 60.0, rnd.choice([10, 50, 100])))
    for trial in range(4):  # 3-5 lists with repeats across and inside lists
        L = rnd.randint(3, 5)
        lists = [[str(rnd.randint(0, 40)) for _ in range(rnd.randint(0, 60))] for _ in range(L)]
        cases.append(run_rrf(lists, rnd.choice([60.0, 10.0, 0.0 + 1e-3]), rnd.choice([5, 25, 200])))
    json.dump(cases, open(os.path.join(OUT, "rrf.json"), "w"), indent=0)

    # ------------------------------------------------------------------ MultiPathRetriever
    class Scripted(BaseRetriever):
        def __init__(self, docs, fail=False):
            super().__init__()
            self.docs, self.fail, self.seen = docs, fail, []

        def _get_relevant_documents(self, query, **kwargs):
            self.seen.append({"query": query, "kwargs": dict(kwargs)})
            if self.fail:
                raise RuntimeError("boom")
            return self.docs[: kwargs.get("k", len(self.docs))]

    def docs(names):
        return [Document(content=n, metadata={"n": n}, id=f"id-{n}") for n in names]

    mp_cases = []
    for spec in [
        dict(lists=[["a", "b", "c", "d"], ["c", "a", "e"]], fail=[False, False], kw={}, per=50),
        dict(lists=[["a", "b", "c", "d"], ["c", "a", "e"]], fail=[False, True], kw={"top_k": 2}, per=3),
        dict(lists=[["a"], ["b"]], fail=[True, True], kw={}, per=50),
        dict(lists=[[], []], fail=[False, False], kw={"top_k": 4}, per=50),
        dict(lists=[[str(i) for i in range(80)], [str(i) for i in range(79, -1, -3)]], fail=[False, False],
             kw={"top_k": 25, "extra": 1}, per=60),
    ]:
        rs = [Scripted(docs(l), f) for l, f in zip(spec["lists"], spec["fail"])]
        mp = MultiPathRetriever(rs, top_k_per_retriever=spec["per"])
        import io, contextlib
        buf = io.StringIO()
        with contextlib.redirect_stdout(buf):
            out = mp.invoke("the query", **spec["kw"])
        mp_cases.append({**spec, "out": [d.content for d in out], "seen": [r.seen for r in rs],
                         "printed_lines": len([l for l in buf.getvalue().splitlines() if l.strip()]),
                         "has_search_kwargs": hasattr(mp, "search_kwargs")})
    json.dump(mp_cases, open(os.path.join(OUT, "multipath.json"), "w"), indent=0)

    # ------------------------------------------------------------------ VectorStoreRetriever
    class FakeStore(VectorStore):
        def __init__(self, scored, relevance="cosine"):
            super().__init__()
            self.scored, self.calls, self.relevance = scored, [], relevance

        def similarity_search(self, query, k=4, **kwargs):
            self.calls.append(["similarity_search", query, k, dict(kwargs)])
            return [d for d, _ in self.scored][:k]

        def similarity_search_with_score(self, query, k=4, **kwargs):
            self.calls.append(["similarity_search_with_score", query, k, dict(kwargs)])
            return self.scored[:k]

        def max_marginal_relevance_search(self, query, k=4, fetch_k=20, lambda_mult=0.5, **kwargs):
            self.calls.append(["max_marginal_relevance_search", query, k, fetch_k, lambda_mult, dict(kwargs)])
            return [d for d, _ in self.scored][::-1][:k]

        def _select_relevance_score_fn(self):
            return {"cosine": self._cosine_relevance_score_fn, "ip": self._max_inner_product_relevance_score_fn,
                    "l2": self._euclidean_relevance_score_fn}[self.relevance]

        @classmethod
        def from_texts(cls, texts, embedding, metadatas=None, *, ids=None, **kwargs):
            raise NotImplementedError

    scored = [(Document(content=f"doc{i}", metadata={}, id=str(i)), s)
              for i, s in enumerate([0.9, 0.75, 0.5, 0.25, 0.1, 0.05, -0.2, 0.0])]
    dense_cases = []

    def run_dense(search_type, search_kwargs, invoke_kwargs, use_async=False):
        st = FakeStore(scored)
        try:
            r = VectorStoreRetriever(st, search_type=search_type, search_kwargs=search_kwargs)
        except Exception as e:
            dense_cases.append({"search_type": search_type, "search_kwargs": search_kwargs,
                                "invoke_kwargs": invoke_kwargs, "async": use_async,
                                "init_error": type(e).__name__})
            return
        with warnings.catch_warnings(record=True) as w:
            warnings.simplefilter("always")
            try:
                if use_async:
                    out = asyncio.run(r.ainvoke("q", **invoke_kwargs))
                else:
                    out = r.invoke("q", **invoke_kwargs)
                err = None
            except Exception as e:
                out, err = [], type(e).__name__
        dense_cases.append({"search_type": search_type, "search_kwargs": search_kwargs,
                            "invoke_kwargs": invoke_kwargs, "async": use_async, "error": err,
                            "out": [d.content for d in out], "calls": st.calls, "n_warnings": len(w)})

    run_dense("similarity", {}, {})                       # default k = 5
    run_dense("similarity", {"k": 3}, {})
    run_dense("similarity", {"k": 3}, {"k": 2, "top_k": 9})   # leaked top_k
    run_dense("similarity", {}, {}, use_async=True)       # async: no default k, no truncation
    run_dense("similarity", {"k": 2}, {}, use_async=True)
    run_dense("similarity_score_threshold", {"score_threshold": 0.4}, {})
    run_dense("similarity_score_threshold", {"score_threshold": 0.4, "k": 8}, {})
    run_dense("similarity_score_threshold", {"score_threshold": 0.95, "k": 8}, {})
    run_dense("similarity_score_threshold", {}, {})       # init error
    run_dense("similarity_score_threshold", {"score_threshold": 1.5}, {})
    run_dense("mmr", {"k": 3, "fetch_k": 6}, {})
    run_dense("bogus", {}, {})
    json.dump(dense_cases, open(os.path.join(OUT, "dense.json"), "w"), indent=0)

    # ------------------------------------------------------------------ the rest of the store / retriever surface
    # VectorStore.search / asearch dispatch, the async twins (adelete, aget_by_ids, asimilarity_search_by_vector,
    # amax_marginal_relevance_search(_by_vector), aadd_documents), from_documents / afrom_documents / afrom_texts, and the
    # retriever's adelete_documents / aget_by_ids / get_vectorstore_info / get_name / update_search_params / repr
    # (VectorStoreBase.py:92-232, :444-627; core/retrieval/dense.py:219-380), over a store that records its calls.
    # Where the reference's own plumbing fails (run_in_executor takes no keyword arguments; add_texts' keyword-only `ids`
    # passed positionally) the recorded outcome is that error.
    sys.path.insert(0, os.path.dirname(os.path.dirname(OUT)))      # repo root (tests.helpers imports the oracle package)
    from tests.helpers import make_recording_store, run_surface_ops
    Recording = make_recording_store(VectorStore)
    surface = run_surface_ops(Document, Recording, VectorStoreRetriever, scored)
    json.dump(surface, open(os.path.join(OUT, "surface.json"), "w"), indent=0, default=str)

    # ------------------------------------------------------------------ relevance score quirk
    rel = {"fns": [], "threshold": []}
    for s in [0.9, 0.5, 0.1, 0.0, -0.3, 1.0, 1.0000001192092896, 0.13347]:
        rel["fns"].append({"score_hex": hexf(s),
                           "cosine_hex": hexf(VectorStore._cosine_relevance_score_fn(s)),
                           "ip_hex": hexf(VectorStore._max_inner_product_relevance_score_fn(s)),
                           "l2_hex": hexf(VectorStore._euclidean_relevance_score_fn(s))})
    for thr in [None, 0.0, 0.4, 0.55, 0.96]:
        st = FakeStore(scored)
        with warnings.catch_warnings(record=True) as w:
            warnings.simplefilter("always")
            kw = {} if thr is None else {"score_threshold": thr}
            out = st.similarity_search_with_relevance_scores("q", k=8, **kw)
        rel["threshold"].append({"thr": thr, "out": [[d.content, hexf(s)] for d, s in out], "n_warnings": len(w)})
    json.dump(rel, open(os.path.join(OUT, "relevance.json"), "w"), indent=0)

    # ------------------------------------------------------------------ cosine helper (float64)
    stubs = {}
    for name in ["simsimd"]:
        pass  # absent on purpose: forces the numpy branch
    from core.file_management.chunker import spliter
    rng = np.random.default_rng(7)
    X = rng.standard_normal((6, 48)).astype(np.float32)
    Y = rng.standard_normal((9, 48)).astype(np.float32)
    X16 = X.astype(np.float16).astype(np.float32)
    Y16 = Y.astype(np.float16).astype(np.float32)
    S = spliter.cosine_similarity(X16.tolist(), Y16.tolist())
    json.dump({"X_f16_bits": X.astype(np.float16).view(np.uint16).tolist(),
               "Y_f16_bits": Y.astype(np.float16).view(np.uint16).tolist(),
               "cos_hex": [[hexf(v) for v in row] for row in np.asarray(S, dtype=np.float64)]},
              open(os.path.join(OUT, "cosine.json"), "w"), indent=0)
    # ------------------------------------------------------------------ MMR selection (float64, python lists)
    class _Absent(types.ModuleType):          # import-time placeholder only (annotations like faiss.Index resolve
        def __getattr__(self, name):          # to `object`); nothing in it is ever called by _mmr_select
            return object
    sys.modules.setdefault("faiss", _Absent("faiss"))
    from encapsulation.database.vector_db.VectorStore_Faiss import _mmr_select
    rng = np.random.default_rng(62)
    mmr_cases = []
    for n, d, k, lam in ((20, 16, 5, 0.5), (20, 16, 20, 0.5), (20, 16, 25, 0.5), (12, 8, 4, 0.0), (12, 8, 4, 1.0),
                         (30, 24, 10, 0.7), (8, 4, 3, 0.5), (1, 4, 1, 0.5), (40, 32, 12, 0.3)):
        E = rng.standard_normal((n, d))
        E[3 % n] = E[0] * 1.0                                     # an exact duplicate of the first pick (max redundancy)
        if n > 6:
            E[5] = E[4]                                           # two identical candidates: an exact tie
        E /= np.linalg.norm(E, axis=1, keepdims=True)
        qv = rng.standard_normal(d)
        qv /= np.linalg.norm(qv)
        docs = [(Document(content=f"c{i}", metadata={}, id=str(i)), 0.0) for i in range(n)]
        picked = _mmr_select(docs, E.tolist(), qv.tolist(), k, lam)
        mmr_cases.append({"n": n, "d": d, "k": k, "lambda": lam, "emb_hex": [[hexf(v) for v in row] for row in E],
                          "query_hex": [hexf(v) for v in qv], "picked": [int(x.id) for x in picked]})
    json.dump({"cases": mmr_cases}, open(os.path.join(OUT, "mmr.json"), "w"), indent=0)

    # ------------------------------------------------------------------ flat-search pin (float64, 64 x 4096)
    # fp16-representable inputs, so every storage format of the build holds them exactly (raw inner product) and
    # the reference's float64 numbers are the one true answer for all of them.  Stored: the full cosine matrix
    # as the reference returned it (float64), nothing derived.
    sys.path.insert(0, os.path.dirname(OUT))                       # tests/  (helpers)
    sys.path.insert(0, os.path.dirname(os.path.dirname(OUT)))      # repo root (helpers imports the oracle package)
    from tests.helpers import PIN_SHAPES, pin_inputs
    for d, nq, n, _seed in PIN_SHAPES:
        Xp, Yp = pin_inputs(d)
        Sp = np.asarray(spliter.cosine_similarity(Xp.astype(np.float64), Yp.astype(np.float64)), dtype=np.float64)
        assert Sp.shape == (nq, n)
        np.savez_compressed(os.path.join(OUT, f"cosine_pin_d{d}.npz"), cos=Sp)
    # ------------------------------------------------------------------ semantic chunker
    # the reference's SemanticChunker over a deterministic fake provider (tests/helpers.py): distances as float64
    # bit patterns + the chunks, per parameter set; short texts exercise the early returns
    from tests.helpers import CHUNKER_CASES, CHUNKER_SHORT_TEXTS, CHUNKER_TEXT, ChunkerFakeEmbeddings
    ch_cases = []
    for params in CHUNKER_CASES:
        ch = spliter.SemanticChunker(ChunkerFakeEmbeddings(), **params)
        pieces = __import__("re").split(ch.sentence_split_regex, CHUNKER_TEXT)
        dist, _ = ch._calculate_sentence_distances(pieces)
        ch_cases.append({"params": params, "distances_hex": [hexf(v) for v in dist], "chunks": ch.split_text(CHUNKER_TEXT)})
    short = [{"text": t, "chunks": spliter.SemanticChunker(ChunkerFakeEmbeddings()).split_text(t),
              "chunks_gradient": spliter.SemanticChunker(ChunkerFakeEmbeddings(), breakpoint_threshold_type="gradient").split_text(t)}
             for t in CHUNKER_SHORT_TEXTS]
    zero = ChunkerFakeEmbeddings().embed_documents(["Rivers run.", "Bread bakes."])
    zmat = spliter.cosine_similarity([zero[0], [0.0] * len(zero[0])], [zero[1], [0.0] * len(zero[0]), zero[0]])
    json.dump({"cases": ch_cases, "short": short, "zero_row_matrix_hex": [[hexf(v) for v in row] for row in zmat]},
              open(os.path.join(OUT, "chunker.json"), "w"), indent=0)
    print("golden vectors written to", OUT)


if __name__ == "__main__":
    main()
