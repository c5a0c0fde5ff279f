"""GPU parity for the rank-level kernels and the registered-backend surface, through the C-ABI.
  rarc_rrf_fuse      vs goldens recorded from the reference's RRFusion.fuse   (core/utils/Fusion.py:45-76)
  rarc_rerank_order  vs the oracle restatement                                (core/rerank/Reranker_Qwen3.py:41-49,:70-74)
  rarc_topk_merge    vs the oracle merge
  HipFlatVectorStore / registry end to end vs the oracle-backed store
"""
import json

import numpy as np
import pytest

from tests.helpers import HashEmbeddings, OracleIndex, golden, unhex

pytestmark = pytest.mark.gpu


def _rr(contents):
    from rag_arc_amd.core.utils import Document, RetrievalResult

    return [RetrievalResult(document=Document(content=c, metadata={"pos": i}), score=1.0) for i, c in enumerate(contents)]


def test_rrf_kernel_matches_reference_goldens_bit_for_bit():
    from rag_arc_amd.core.utils import HipRRFusion

    for case in golden("rrf.json"):
        res = [_rr(one) for one in case["lists"]]
        fused = HipRRFusion(k=case["k"]).fuse(res, case["top_k"])
        assert [f.document.content for f in fused] == [g["content"] for g in case["fused"]]
        assert [f.score for f in fused] == [unhex(g["score_hex"]) for g in case["fused"]]      # exact fp64
        assert [f.rank for f in fused] == [g["rank"] for g in case["fused"]]
        assert [[r.rank for r in one] for one in res] == case["input_ranks_after"]             # inputs re-ranked
        # "last occurrence wins" for the returned Document object
        for f, g in zip(fused, case["fused"]):
            assert f.document.metadata["pos"] == g["doc_pos"]


def test_rrf_batched_ids_c3_shape(oracle):
    """256 queries x (dense top-100, supplied lexical top-100 with ~30 % overlap): ids and fp64 scores."""
    import torch

    from rag_arc_amd.core.utils import HipRRFusion

    rng = np.random.default_rng(777)
    B, L = 256, 100
    keys = np.zeros((B, 2, L), np.int64)
    for b in range(B):
        dense = rng.choice(1_000_000, L, replace=False)
        over = rng.choice(dense, 30, replace=False)
        rest = rng.choice(np.arange(1_000_000, 2_000_000), L - 30, replace=False)
        lex = np.concatenate([over, rest])
        rng.shuffle(lex)
        keys[b, 0], keys[b, 1] = dense, lex
    lens = np.full((B, 2), L, np.int32)
    lens[3, 1] = 0
    lens[4] = (17, 55)
    fk, fs, fn = HipRRFusion().fuse_ids(torch.from_numpy(keys).cuda(), torch.from_numpy(lens).cuda(), 100)
    fk, fs, fn = fk.cpu().numpy(), fs.cpu().numpy(), fn.cpu().numpy()
    for b in range(B):
        want = oracle.rrf_fuse([keys[b, r, : lens[b, r]].tolist() for r in range(2)], 60.0, 100)
        assert fn[b] == len(want)
        assert fk[b, : fn[b]].tolist() == [k for k, _ in want]
        assert fs[b, : fn[b]].tolist() == [s for _, s in want]


def test_rerank_order_kernel(oracle):
    from rag_arc_amd.core.rerank import HipLogitReranker
    from rag_arc_amd.core.utils import Document

    rng = np.random.default_rng(9)
    zn = (rng.standard_normal((32, 100)) * 4).astype(np.float16)
    zy = (rng.standard_normal((32, 100)) * 4).astype(np.float16)
    zy[0, 10:20] = zy[0, 10]
    zn[0, 10:20] = zn[0, 10]                                         # exact ties keep retrieval order
    rr = HipLogitReranker(lambda q, texts: (None, None))
    scores, perm = rr.score_order(zn, zy)
    scores, perm = scores.cpu().numpy(), perm.cpu().numpy()
    want = oracle.rerank_scores_f16(zn, zy)
    tol = np.maximum(np.abs(want.astype(np.float64)) * 2.0 ** -6, 2.0 ** -24)   # exp amplifies a last-place ls flip
    assert np.all(np.abs(scores.astype(np.float64) - want.astype(np.float64)) <= tol)
    assert (scores == want).mean() > 0.98
    for b in range(32):
        assert perm[b].tolist() == oracle.stable_desc_order(scores[b]).tolist()   # stable sort of ITS scores
    # the Reranker contract: same Document objects, reordered, optional k, batches of 8
    docs = [Document(content=f"d{i}") for i in range(100)]
    table = {f"d{i}": (zn[1, i], zy[1, i]) for i in range(100)}
    calls = []

    def logit_fn(query, texts):
        calls.append(len(texts))
        return [table[t][0] for t in texts], [table[t][1] for t in texts]

    out = HipLogitReranker(logit_fn).rerank("q", docs, k=7)
    assert [d.content for d in out] == [f"d{i}" for i in perm[1][:7]] and all(o is docs[int(o.content[1:])] for o in out)
    assert calls == [8] * 12 + [4]


def test_topk_merge_kernel(oracle):
    import torch

    from rag_arc_amd.hip.sharded import ShardedFlatSearch

    rng = np.random.default_rng(4)
    G, nq, k = 8, 64, 100
    ids = rng.integers(0, 10 ** 9, (G, nq, k)).astype(np.int64)
    sc = np.sort(rng.standard_normal((G, nq, k)).astype(np.float32), axis=2)[:, :, ::-1].copy()
    sc[:, 0, :] = 0.25                                   # a query where everything ties: id ascending
    ids[2, 1, 50:] = -1
    sc[2, 1, 50:] = -np.inf                              # a short shard
    s = ShardedFlatSearch.__new__(ShardedFlatSearch)
    s.torch = torch
    mi, ms = s._hip_merge(torch.from_numpy(ids).cuda(), torch.from_numpy(sc).cuda(), k)
    wi, ws = oracle.topk_merge(ids, sc, k)
    assert np.array_equal(mi.cpu().numpy(), wi) and np.array_equal(ms.cpu().numpy().view(np.uint32), ws.view(np.uint32))
    # the packed form that crosses the all-gather: pack each shard's block, merge the packed lists
    from rag_arc_amd.hip import binding as B

    lib = B.load_library()
    d_ids, d_sc = torch.from_numpy(ids).cuda(), torch.from_numpy(sc).cuda()
    packed = torch.empty((G, nq, k, 3), dtype=torch.int32, device="cuda")
    for g in range(G):
        B.check(lib.rarc_pack_results(d_ids[g].data_ptr(), d_sc[g].data_ptr(), nq, k, packed[g].data_ptr(), 0))
    pi = torch.empty((nq, k), dtype=torch.int64, device="cuda")
    ps = torch.empty((nq, k), dtype=torch.float32, device="cuda")
    B.check(lib.rarc_topk_merge_packed(packed.data_ptr(), G, nq, k, pi.data_ptr(), ps.data_ptr(), 0))
    assert np.array_equal(pi.cpu().numpy(), wi) and np.array_equal(ps.cpu().numpy().view(np.uint32), ws.view(np.uint32))


def test_store_end_to_end_and_registry(tmp_path, oracle):
    from rag_arc_amd.config.app_registration import register_multipath_retriever, registrator
    from rag_arc_amd.core.retrieval import VectorStoreRetriever
    from rag_arc_amd.encapsulation.database.vector_db import HipFlatVectorStore

    emb = HashEmbeddings(384)
    texts = [f"passage {i}" for i in range(10_000)]
    ids = [str(i) for i in range(10_000)]
    hip = HipFlatVectorStore.from_texts(texts, emb, ids=ids)                       # real HIP engine
    ref = HipFlatVectorStore.from_texts(texts, emb, ids=ids, engine_factory=lambda d, m, dev: OracleIndex(d, m))
    for q in ("passage 77", "passage 9999", "something else entirely"):
        a, b = hip.similarity_search_with_score(q, k=10), ref.similarity_search_with_score(q, k=10)
        assert [(d.id, s) for d, s in a] == [(d.id, s) for d, s in b]
    sc, rows = hip.batch_search_by_vector(emb.embed_documents(texts[:300]), k=10)   # nq > 256: two scan passes
    sc2, rows2 = ref.batch_search_by_vector(emb.embed_documents(texts[:300]), k=10)
    assert np.array_equal(rows, rows2) and np.array_equal(sc.view(np.uint32), sc2.view(np.uint32))
    assert [d.id for d in VectorStoreRetriever(hip).invoke("passage 5")] == [d.id for d in VectorStoreRetriever(ref).invoke("passage 5")]
    hip.save_local(str(tmp_path / "idx"))
    again = HipFlatVectorStore.load_local(str(tmp_path / "idx"), emb)
    assert [(d.id, s) for d, s in again.similarity_search_with_score("passage 77", k=10)] == \
           [(d.id, s) for d, s in hip.similarity_search_with_score("passage 77", k=10)]

    # MMR from the resident rows (no re-embedding): same candidates, same first pick; later picks may
    # differ from the re-embedding form only where MMR values tie within fp16 rounding (hash embeddings
    # are mutually near-orthogonal, so ties are the rule here) — both are valid MMR orders of one pool
    for q in ("passage 77", "passage 4242"):
        a = hip.max_marginal_relevance_search(q, k=5, fetch_k=20)
        b = hip.max_marginal_relevance_search(q, k=5, fetch_k=20, reembed=False)   # rarc_mmr_select on the resident rows
        pool = {d.id for d in hip.similarity_search(q, k=20)}
        assert len(b) == 5 and a[0].id == b[0].id and {d.id for d in b} <= pool and len({d.id for d in b}) == 5
        # the device selection equals the host restatement of the reference's loop run on the same stored vectors
        qv = np.asarray(emb.embed_query(q), np.float64)
        sc_, rows_ = hip.index.search(np.array([emb.embed_query(q)], np.float32), 20)
        cand = hip._stored_vectors([int(r) for r in rows_[0]])
        docs_ = [(hip.docstore[hip.index_to_docstore_id[int(r)]], float(s_)) for s_, r in zip(sc_[0], rows_[0])]
        from rag_arc_amd.encapsulation.database.vector_db.hip_flat import _mmr_select
        want = _mmr_select(docs_, (cand / np.linalg.norm(cand, axis=1, keepdims=True)).tolist(), (qv / np.linalg.norm(qv)).tolist(), 5, 0.5)
        assert [d.id for d in b][0] == want[0].id and {d.id for d in b} <= pool

    # the same store over fp8 rows (half the HBM): answers equal the fp8 oracle's, and survive save / load
    hip8 = HipFlatVectorStore.from_texts(texts, emb, ids=ids, storage="f8")
    X = np.asarray(emb.embed_documents(texts), np.float32)
    b8, s8, _ = oracle.ingest_f8(X)
    for q in ("passage 77", "something else entirely"):
        got = hip8.similarity_search_with_score(q, k=10)
        oi, osc, _ = oracle.flat_search_f8(b8, s8, oracle.normalize_L2(np.asarray([emb.embed_query(q)], np.float32)), 10)
        assert [d.id for d, _ in got] == [str(i) for i in oi[0]]
        assert [s for _, s in got] == [float(v) for v in osc[0]]
    hip8.save_local(str(tmp_path / "idx8"))
    again8 = HipFlatVectorStore.load_local(str(tmp_path / "idx8"), emb)
    assert again8.storage == "f8"
    assert [(d.id, s) for d, s in again8.similarity_search_with_score("passage 77", k=10)] == \
           [(d.id, s) for d, s in hip8.similarity_search_with_score("passage 77", k=10)]

    # JSON -> Register -> module graph -> invoke (the framework surface, with HIP fusion)
    qs = ["passage 1", "passage 2"]
    np.savez(tmp_path / "emb.npz", texts=np.array(texts + ["query a"]), vectors=np.array(emb.embed_documents(texts + ["query a"]), np.float32))
    np.savez(tmp_path / "corpus.npz", texts=np.array(texts[:2000]), ids=np.array(ids[:2000]))
    vs = {"type": "hip_flat_vectorstore", "metric": "cosine", "embedding": {"type": "table_embeddings", "path": str(tmp_path / "emb.npz")},
          "corpus_path": str(tmp_path / "corpus.npz")}
    cfg = {"type": "multipath_retriever", "top_k_per_retriever": 20, "fusion": {"type": "rrf", "k": 60.0},
           "retrievers": [{"type": "vectorstore_retriever", "vectorstore": vs, "search_kwargs": {}},
                          {"type": "vectorstore_retriever", "vectorstore": vs, "search_type": "mmr", "search_kwargs": {"fetch_k": 30}}]}
    (tmp_path / "app.json").write_text(json.dumps(cfg))
    register_multipath_retriever(str(tmp_path / "app.json"), "t_app")
    app = registrator.get_object("t_app")
    out = app.invoke("passage 1", top_k=5)
    assert out[0].content == "passage 1" and len(out) == 5


def test_mmr_kernel_reproduces_reference_picks():
    """rarc_mmr_select on the inputs recorded with the reference's _mmr_select (VectorStore_Faiss.py:16-62): the picks
    are equal wherever the inputs survive the cast to the kernel's fp32 candidates without creating new ties (the
    recorded cases include exact duplicates and exact ties, which stay exact in fp32)."""
    import torch

    from rag_arc_amd.hip import binding as B

    lib = B.load_library()
    for c in golden("mmr.json")["cases"]:
        E = np.array([[unhex(v) for v in row] for row in c["emb_hex"]], np.float64)
        q = np.array([unhex(v) for v in c["query_hex"]], np.float64)
        E32 = E.astype(np.float32)
        # reference picks for the fp32-rounded candidates (host mirror, itself pinned to the reference on the fp64 inputs)
        from rag_arc_amd.core.utils.data_model import Document
        from rag_arc_amd.encapsulation.database.vector_db.hip_flat import _mmr_select
        docs = [(Document(content=f"c{i}", metadata={}, id=str(i)), 0.0) for i in range(c["n"])]
        want = [int(d.id) for d in _mmr_select(docs, E32.astype(np.float64).tolist(), q.tolist(), c["k"], c["lambda"])]
        n, d = E32.shape
        if c["k"] >= n:
            continue                                    # (the store returns all candidates without calling the kernel)
        cand = torch.from_numpy(E32).cuda()
        qd = torch.from_numpy(q).cuda()
        work = torch.empty(int(lib.rarc_mmr_workspace_doubles(n, d)), dtype=torch.float64, device="cuda")
        out = torch.empty(c["k"], dtype=torch.int32, device="cuda")
        B.check(lib.rarc_mmr_select(cand.data_ptr(), d, qd.data_ptr(), n, d, 0, c["k"], float(c["lambda"]), work.data_ptr(),
                                    out.data_ptr(), 0), "rarc_mmr_select")
        assert out.cpu().tolist() == want, (c["n"], c["k"], c["lambda"])
