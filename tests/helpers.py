"""Test-only helpers: an oracle-backed stand-in for the HIP index (so host logic can run without a
GPU), scripted retrievers / stores, and a FusionMethod that runs the oracle's RRF."""
import hashlib
import json
import os
import struct

import numpy as np

from oracle import cpu_ref
from rag_arc_amd.core.retrieval.base import BaseRetriever
from rag_arc_amd.core.utils.data_model import Document
from rag_arc_amd.core.utils.fusion import FusionMethod, RetrievalResult
from rag_arc_amd.encapsulation.database.vector_db.base import VectorStore
from rag_arc_amd.encapsulation.embeddings.base import Embeddings

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def fp16_grid_matrix(n: int, d: int, seed: int) -> np.ndarray:
    """Deterministic [n][d] float32 matrix whose every entry is exactly representable in fp16 (and so in fp32 and
    float64): a bell-shaped sum of four hashed bytes on a grid of 1/128, |v| < 4.  Pure numpy integer arithmetic
    (splitmix64), so the golden generator (tests/golden/make_golden.py, which feeds these very numbers to the
    reference) and the tests build identical inputs without storing them."""
    with np.errstate(over="ignore"):
        base = np.array([seed], dtype=np.uint64) * np.uint64(0x9E3779B97F4A7C15)   # wraps mod 2^64
        z = np.arange(n * d, dtype=np.uint64) + base + np.uint64(0x9E3779B97F4A7C15)
        z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        z = z ^ (z >> np.uint64(31))
    v = ((z & np.uint64(0xFF)) + ((z >> np.uint64(8)) & np.uint64(0xFF)) + ((z >> np.uint64(16)) & np.uint64(0xFF))
         + ((z >> np.uint64(24)) & np.uint64(0xFF))).astype(np.int64) - 510
    out = (v.astype(np.float32) / np.float32(128.0)).reshape(n, d)
    assert np.array_equal(out.astype(np.float16).astype(np.float32), out)
    return out


PIN_SHAPES = ((384, 64, 4096, 11), (768, 64, 4096, 12))   # (d, queries, rows, seed) of tests/golden/cosine_pin_d*.npz


def pin_inputs(d: int):
    """(X queries, Y rows) of the flat-search pin for dimension d."""
    dd, nq, n, seed = next(sh for sh in PIN_SHAPES if sh[0] == d)
    return fp16_grid_matrix(nq, dd, seed), fp16_grid_matrix(n, dd, seed + 100)


def pin_reference(d: int):
    """The reference's float64 cosine matrix for pin_inputs(d) plus what follows from it: float64 norms, the raw
    inner products cos * |x| * |y|, and per query the ranking by either."""
    X, Y = pin_inputs(d)
    cos = np.load(os.path.join(GOLDEN, f"cosine_pin_d{d}.npz"))["cos"]
    xn, yn = np.linalg.norm(X.astype(np.float64), axis=1), np.linalg.norm(Y.astype(np.float64), axis=1)
    return X, Y, cos, xn, yn


def check_against_pin(ref_scores: np.ndarray, got_ids: np.ndarray, got_scores: np.ndarray, tol: float = 1e-5):
    """ref_scores: float64 [nq][n] (the reference's numbers); got: top-k of some implementation, scores already in
    the same unit.  Every returned score within `tol` of the reference's number for that row; the returned SET
    equals the reference's top-k wherever its k/k+1 gap exceeds `tol`; the ORDER too wherever all the gaps do.
    Returns (max |delta|, queries set-checked, queries order-checked)."""
    nq, k = got_ids.shape
    worst, n_set, n_ord = 0.0, 0, 0
    for b in range(nq):
        order = np.lexsort((np.arange(ref_scores.shape[1]), -ref_scores[b]))
        top = ref_scores[b][order[: k + 1]]
        worst = max(worst, float(np.max(np.abs(ref_scores[b][got_ids[b]] - got_scores[b].astype(np.float64)))))
        if k == ref_scores.shape[1] or top[k - 1] - top[k] > tol:
            n_set += 1
            assert set(got_ids[b].tolist()) == set(order[:k].tolist()), f"query {b}: top-{k} set differs"
        if np.min(top[:-1] - top[1:]) > tol / 5:
            n_ord += 1
            assert got_ids[b].tolist() == order[:k].tolist(), f"query {b}: top-{k} order differs"
    assert worst < tol, f"max |score - reference| = {worst}"
    return worst, n_set, n_ord


def golden(name):
    with open(os.path.join(GOLDEN, name)) as fh:
        return json.load(fh)


def unhex(h: str) -> float:
    return struct.unpack(">d", bytes.fromhex(h))[0]


class OracleIndex:
    """Same surface as rag_arc_amd.hip.engine.FlatIndexF16, computed by the CPU oracle."""

    def __init__(self, dim, metric="cosine", device=0):
        self.dim, self.metric, self.ntotal, self.max_norm = dim, metric, 0, 0.0
        self.d_pad = cpu_ref.padded_dim(dim)
        self._rows = np.zeros((0, self.d_pad), np.uint16)

    @property
    def rows(self):
        return self._rows

    def add(self, vectors):
        r, n2 = cpu_ref.ingest_f16(np.asarray(vectors, np.float32), normalize=(self.metric == "cosine"))
        self._rows = np.concatenate([self._rows, r])
        self.ntotal = self._rows.shape[0]
        self.max_norm = max(self.max_norm, float(np.sqrt(n2.max()))) if len(n2) else self.max_norm

    def load_rows(self, rows, max_norm, row_scales=None):
        self._rows = np.concatenate([self._rows, np.asarray(rows).view(np.uint16)])
        self.ntotal = self._rows.shape[0]
        self.max_norm = max(self.max_norm, max_norm)

    # the store's persistence interface towards its engine (hip/engine.py: save_shard / load_shard), served from host
    # arrays here — this double lives in tests/ and never ships
    id_base = 0

    def save_shard(self, path, blocks=None, rank=0, world=1, global_ntotal=None, **_):
        from rag_arc_amd.hip import shardfile as SF

        n = self.ntotal
        blocks = [(0, n)] if blocks is None else list(blocks)
        hdr = SF.ShardHeader(n, self.dim, self.d_pad, SF.CODES["f16"], float(self.max_norm), rank, world,
                             n if global_ntotal is None else global_ntotal, blocks)
        with open(path + ".tmp", "wb") as fh:
            fh.write(hdr.pack())
            fh.truncate(hdr.file_bytes)
            fh.seek(hdr.rows_offset)
            fh.write(np.ascontiguousarray(self._rows).tobytes())
            fh.seek(hdr.idmap_offset)
            fh.write(hdr.idmap_bytes())
        os.replace(path + ".tmp", path)
        return dict(bytes=n * hdr.row_bytes, seconds=0.0, gb_per_s=0.0)

    def load_shard(self, path, row_ranges=None, header=None, **_):
        from rag_arc_amd.hip import shardfile as SF

        hdr = header or SF.read_header(path)
        assert hdr.storage == "f16" and hdr.d_pad == self.d_pad and hdr.dim == self.dim
        rows = np.memmap(path, dtype=np.uint16, mode="r", offset=hdr.rows_offset, shape=(hdr.n_rows, hdr.d_pad))
        for a, c in ([(0, hdr.n_rows)] if row_ranges is None else row_ranges):
            self._rows = np.concatenate([self._rows, np.array(rows[a:a + c])])
        self.ntotal = self._rows.shape[0]
        self.max_norm = max(self.max_norm, hdr.max_norm)
        return dict(bytes=0, seconds=0.0, gb_per_s=0.0)

    def reset(self):
        self._rows = np.zeros((0, self.d_pad), np.uint16)
        self.ntotal = 0

    def search(self, queries, k):
        q = np.asarray(queries, np.float32)
        if self.metric == "cosine":
            q = cpu_ref.normalize_L2(q)
        ids, scores, _ = cpu_ref.flat_search_f16(self._rows, q, k)
        return scores, ids


class HashEmbeddings(Embeddings):
    """Deterministic text -> vector map (no model weights exist offline)."""

    def __init__(self, dim=384):
        super().__init__()
        self.dim = dim

    def _one(self, text):
        seed = int.from_bytes(hashlib.sha256(text.encode()).digest()[:8], "little")
        return np.random.default_rng(seed).standard_normal(self.dim).astype(np.float32)

    def embed_documents(self, texts):
        return [self._one(t.replace("\n", " ")).tolist() for t in texts]

    def embed_query(self, text):
        return self.embed_documents([text])[0]


class OracleFusion(FusionMethod):
    """RRF through oracle.cpu_ref.rrf_fuse with the reference's side effects (ranks, last doc wins)."""

    def __init__(self, k=60.0):
        self.k = k

    def fuse(self, results, top_k):
        docs = {}
        for one in results:
            for i, r in enumerate(one):
                r.rank = i + 1
                docs[r.document.content] = r.document
        fused = cpu_ref.rrf_fuse([[r.document.content for r in one] for one in results], self.k, top_k)
        return [RetrievalResult(document=docs[c], score=s, rank=i + 1) for i, (c, s) in enumerate(fused)]


class ScriptedRetriever(BaseRetriever):
    def __init__(self, docs, fail=False):
        super().__init__()
        self.docs, self.fail, self.seen = docs, fail, []

    def _get_relevant_documents(self, query, **kwargs):
        self.seen.append({"query": query, "kwargs": dict(kwargs)})
        if self.fail:
            raise RuntimeError("boom")
        return self.docs[: kwargs.get("k", len(self.docs))]


class ScriptedStore(VectorStore):
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


def scripted_scored():
    return [(Document(content=f"doc{i}", metadata={}, id=str(i)), s)
            for i, s in enumerate([0.9, 0.75, 0.5, 0.25, 0.1, 0.05, -0.2, 0.0])]


# ---- semantic chunker fixture (tests/golden/chunker.json): inputs shared by the golden generator and the tests ----
class ChunkerFakeEmbeddings:
    """Deterministic stand-in for an embedding provider: a topic vector chosen by the sentence window's LAST topic word
    plus hashed per-text noise, all on the fp16 grid (exact in fp32) — so the reference (float64 on these floats) and
    the device kernels (fp32 rows, float64 sums) see identical numbers."""

    def __init__(self, d: int = 96):
        self.d = d

    def embed_documents(self, texts):
        import zlib

        out = []
        for t in texts:
            words = [w.strip(".?!,").lower() for w in t.split()]
            topic = next((w for w in reversed(words) if w in CHUNKER_TOPICS), "none")
            base = fp16_grid_matrix(1, self.d, 1000 + sorted(CHUNKER_TOPICS + ("none",)).index(topic))[0]
            noise = fp16_grid_matrix(1, self.d, zlib.crc32(t.encode()) & 0xFFFF)[0]
            out.append((base * np.float32(2.0) + noise * np.float32(0.25)).astype(np.float32).tolist())
        return out

    def embed_query(self, text):
        return self.embed_documents([text])[0]


CHUNKER_TOPICS = ("rivers", "engines", "bread", "stars")
CHUNKER_TEXT = (
    "Rivers carry silt to the sea. Many rivers flood in spring! Do rivers freeze in the north? Old rivers meander widely. "
    "Engines burn fuel to turn shafts. Small engines idle roughly. Are engines quieter now? Diesel engines last long. "
    "Modern engines use sensors. Bread needs flour and water. Good bread takes time! Why does bread rise? "
    "Stale bread makes fine crumbs. Stars fuse hydrogen for ages. Some stars collapse. Can stars be counted? "
    "Bright stars guide sailors. Distant stars look red.")
CHUNKER_CASES = (
    {"breakpoint_threshold_type": "percentile"},
    {"breakpoint_threshold_type": "percentile", "breakpoint_threshold_amount": 70, "buffer_size": 0},
    {"breakpoint_threshold_type": "standard_deviation", "breakpoint_threshold_amount": 1.0},
    {"breakpoint_threshold_type": "interquartile", "breakpoint_threshold_amount": 0.5, "buffer_size": 2},
    {"breakpoint_threshold_type": "gradient", "breakpoint_threshold_amount": 80},
    {"number_of_chunks": 4},
    {"number_of_chunks": 40},
    {"number_of_chunks": 1},
    {"breakpoint_threshold_type": "percentile", "breakpoint_threshold_amount": 50, "min_chunk_size": 120},
)
CHUNKER_SHORT_TEXTS = ("One sentence only.", "Two sentences here. That is all.", "")


# ---- the store / retriever surface: one script of operations, run by tests/golden/make_golden.py over the REFERENCE's
# classes (the outcomes are tests/golden/surface.json) and by tests/test_surface_golden.py over this repo's mirror ----------
def make_recording_store(VectorStoreBase):
    """A VectorStore over `VectorStoreBase` (the reference's ABC, or the mirror's) that records every call it receives."""

    class Recording(VectorStoreBase):
        def __init__(self, scored=None, relevance="cosine", **kw):
            super().__init__()
            self.scored, self.calls, self.relevance, self.init_kw = (scored if scored is not None else []), [], relevance, kw

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

        def add_texts(self, texts, metadatas=None, *, ids=None, **kwargs):
            self.calls.append(["add_texts", list(texts), metadatas, ids, dict(kwargs)])
            return list(ids) if ids else [f"gen{i}" for i, _ in enumerate(texts)]

        def delete(self, ids=None, **kwargs):
            self.calls.append(["delete", ids, dict(kwargs)])
            return True if ids != ["nope"] else False

        def get_by_ids(self, ids):
            self.calls.append(["get_by_ids", list(ids)])
            return [d for d, _ in self.scored if d.id in ids]

        def similarity_search_by_vector(self, embedding, k=4, **kwargs):
            self.calls.append(["similarity_search_by_vector", list(embedding), k, dict(kwargs)])
            return [d for d, _ in self.scored][:k]

        def max_marginal_relevance_search_by_vector(self, embedding, k=4, fetch_k=20, lambda_mult=0.5, **kwargs):
            self.calls.append(["max_marginal_relevance_search_by_vector", list(embedding), k, fetch_k, lambda_mult, dict(kwargs)])
            return [d for d, _ in self.scored][::-1][:k]

        @classmethod
        def from_texts(cls, texts, embedding, metadatas=None, *, ids=None, **kwargs):
            st = cls(**kwargs)
            st.calls.append(["from_texts", list(texts), embedding, metadatas, ids])
            return st

    return Recording


def run_surface_ops(Document, Recording, Retriever, scored):
    """Run the script; returns [{op, result, error, calls}] (JSON-able)."""
    import asyncio
    import warnings

    surface = []

    def record(name, fn, is_async=False, show=lambda r: r):
        st = Recording(scored)
        with warnings.catch_warnings(record=True):
            warnings.simplefilter("always")
            try:
                res = asyncio.run(fn(st)) if is_async else fn(st)
                out, err = show(res), None
            except Exception as e:   # noqa: BLE001
                out, err = None, type(e).__name__
        surface.append({"op": name, "result": out, "error": err, "calls": st.calls})

    names = lambda docs_: [d.content for d in docs_]   # noqa: E731
    for stype in ("similarity", "similarity_score_threshold", "mmr", "bogus"):
        record(f"search:{stype}", lambda st, t=stype: st.search("q", t, k=3), show=names)
        record(f"asearch:{stype}", lambda st, t=stype: st.asearch("q", t, k=3), is_async=True, show=names)
    record("search:threshold_kw", lambda st: st.search("q", "similarity_score_threshold", k=8, score_threshold=0.5), show=names)
    record("asearch:threshold_kw", lambda st: st.asearch("q", "similarity_score_threshold", k=8, score_threshold=0.5),
           is_async=True, show=names)
    record("asearch:similarity_extra_kw", lambda st: st.asearch("q", "similarity", k=2, top_k=9), is_async=True, show=names)
    record("adelete:ids", lambda st: st.adelete(["1", "2"]), is_async=True)
    record("adelete:none", lambda st: st.adelete(), is_async=True)
    record("adelete:missing", lambda st: st.adelete(["nope"]), is_async=True)
    record("adelete:kw", lambda st: st.adelete(["1"], soft=True), is_async=True)
    record("aget_by_ids", lambda st: st.aget_by_ids(["1", "3", "zzz"]), is_async=True, show=names)
    record("asimilarity_search_by_vector", lambda st: st.asimilarity_search_by_vector([0.5, 0.25], 2), is_async=True, show=names)
    record("asimilarity_search_by_vector:kw", lambda st: st.asimilarity_search_by_vector([0.5, 0.25], k=2), is_async=True, show=names)
    record("asimilarity_search_by_vector:extra_kw", lambda st: st.asimilarity_search_by_vector([0.5], 2, flag=1), is_async=True, show=names)
    record("amax_marginal_relevance_search", lambda st: st.amax_marginal_relevance_search("q", 2, 6, 0.25), is_async=True, show=names)
    record("amax_marginal_relevance_search:kw", lambda st: st.amax_marginal_relevance_search("q", k=2, fetch_k=6), is_async=True, show=names)
    record("amax_marginal_relevance_search_by_vector", lambda st: st.amax_marginal_relevance_search_by_vector([1.0, 0.0], 3, 7, 0.75),
           is_async=True, show=names)
    record("amax_marginal_relevance_search_by_vector:kw", lambda st: st.amax_marginal_relevance_search_by_vector([1.0], k=3),
           is_async=True, show=names)
    some = [Document(content="t0", metadata={"a": 1}, id="i0"), Document(content="t1", metadata={}, id=None)]
    none = [Document(content="u0", metadata={}, id=None), Document(content="u1", metadata={"b": 2}, id=None)]
    record("add_documents:some_ids", lambda st: st.add_documents(some))
    record("add_documents:no_ids", lambda st: st.add_documents(none))
    record("add_documents:kw_ids", lambda st: st.add_documents(some, ids=["x", "y"]))
    record("aadd_documents:some_ids", lambda st: st.aadd_documents(some), is_async=True)
    record("aadd_documents:kw_ids", lambda st: st.aadd_documents(some, ids=["x", "y"]), is_async=True)
    record("aadd_texts", lambda st: st.aadd_texts(["p", "q"], [{}, {}], ids=["1", "2"]), is_async=True)
    made = lambda st: st.calls + [["init_kw", st.init_kw]]   # noqa: E731
    record("from_documents:some_ids", lambda _: Recording.from_documents(some, "EMB"), show=made)
    record("from_documents:no_ids", lambda _: Recording.from_documents(none, "EMB"), show=made)
    record("from_documents:kw_ids", lambda _: Recording.from_documents(some, "EMB", ids=["x", "y"]), show=made)
    record("afrom_documents:some_ids", lambda _: Recording.afrom_documents(some, "EMB"), is_async=True, show=made)
    record("afrom_documents:no_ids", lambda _: Recording.afrom_documents(none, "EMB"), is_async=True, show=made)
    record("afrom_texts:ids", lambda _: Recording.afrom_texts(["p", "q"], "EMB", [{"m": 1}, {}], ids=["1", "2"]), is_async=True, show=made)
    record("afrom_texts:no_ids", lambda _: Recording.afrom_texts(["p", "q"], "EMB"), is_async=True, show=made)
    record("afrom_texts:extra_kw", lambda _: Recording.afrom_texts(["p"], "EMB", relevance="ip"), is_async=True, show=made)

    def with_retriever(fn, **rkw):
        return lambda st: fn(Retriever(st, **rkw))

    record("retriever.delete_documents", with_retriever(lambda r: r.delete_documents(["1"])))
    record("retriever.delete_documents:none", with_retriever(lambda r: r.delete_documents()))
    record("retriever.adelete_documents", with_retriever(lambda r: r.adelete_documents(["1", "2"])), is_async=True)
    record("retriever.adelete_documents:none", with_retriever(lambda r: r.adelete_documents()), is_async=True)
    record("retriever.get_by_ids", with_retriever(lambda r: r.get_by_ids(["2", "zzz"])), show=names)
    record("retriever.aget_by_ids", with_retriever(lambda r: r.aget_by_ids(["2", "5"])), is_async=True, show=names)
    record("retriever.add_documents", with_retriever(lambda r: r.add_documents(some)))
    record("retriever.aadd_documents", with_retriever(lambda r: r.aadd_documents(some)), is_async=True)
    record("retriever.get_vectorstore_info", with_retriever(lambda r: r.get_vectorstore_info(), search_kwargs={"k": 3}))
    record("retriever.get_name", with_retriever(lambda r: r.get_name()))
    record("retriever.repr", with_retriever(lambda r: repr(r), search_type="mmr", search_kwargs={"k": 2}))

    def upd(r, **kw):
        r.update_search_params(**kw)
        return {"search_type": r.search_type, "search_kwargs": r.search_kwargs}

    record("retriever.update_search_params:k", with_retriever(lambda r: upd(r, k=7), search_kwargs={"k": 3}))
    record("retriever.update_search_params:type", with_retriever(lambda r: upd(r, search_type="mmr", fetch_k=9)))
    record("retriever.update_search_params:bad_type", with_retriever(lambda r: upd(r, search_type="bogus")))
    record("retriever.update_search_params:threshold_missing", with_retriever(lambda r: upd(r, search_type="similarity_score_threshold")))
    return surface
