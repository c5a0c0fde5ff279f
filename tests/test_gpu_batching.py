"""The benched throughput reached THROUGH the registered backend, on the GPU:
  * 256 threads calling `retriever.invoke` on a 1M-row store finish in <= 3 scan launches (counted by the library's own
    launch brackets, rarc_profile_begin / rarc_profile_end) with the answers of 256 one-query searches;
  * `batch_invoke` of the dense and the multi-path retriever (fuse_many: one RRF launch) equals `invoke` query by query;
  * a row-sharded store built from JSON on two ranks (gloo, one device) equals the single-shard store."""
import ctypes
import json
import os
import sys
import threading

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


class _SynthEmbeddings:
    """Documents "doc<i>" -> row i of the synthetic corpus stream (seed 1234), queries "q<j>" -> row j of the query
    stream (seed 4321); a contiguous run of documents is generated on the device in one call."""

    def __init__(self, dim):
        import torch

        from rag_arc_amd.hip import binding as B

        self.dim, self.torch, self.lib, self.B = dim, torch, B.load_library(), B
        self.query_batches = []

    def _rows(self, first, n, seed):
        out = self.torch.empty((n, self.dim), dtype=self.torch.float32, device="cuda")
        self.B.check(self.lib.rarc_synth_rows_f32(out.data_ptr(), self.dim, self.dim, first, n, seed, 0))
        return out

    def embed_documents_device(self, texts):
        first = int(texts[0][3:])
        assert all(t == f"doc{first + i}" for i, t in ((0, texts[0]), (len(texts) - 1, texts[-1])))
        return self._rows(first, len(texts), 1234)

    def embed_documents(self, texts):
        return self.embed_documents_device(texts).cpu().numpy().tolist()

    def embed_query(self, text):
        return self._rows(int(text[1:]), 1, 4321)[0].cpu().numpy().tolist()

    def embed_queries_device(self, texts):            # embed_query applied to each text, in one go
        self.query_batches.append(len(texts))
        js = [int(t[1:]) for t in texts]
        if js == list(range(js[0], js[0] + len(js))):
            return self._rows(js[0], len(js), 4321)
        return self.torch.cat([self._rows(j, 1, 4321) for j in js])


def _scan_launches(fn):
    from rag_arc_amd.hip import binding as B

    lib = B.load_library()
    B.check(lib.rarc_profile_begin(4096))
    out = fn()
    ms, n = ctypes.c_double(0), ctypes.c_int(0)
    B.check(lib.rarc_profile_end(ctypes.byref(ms), ctypes.byref(n)))
    return out, n.value


def test_256_concurrent_invokes_share_three_scans():
    from rag_arc_amd.core.retrieval.dense import VectorStoreRetriever
    from rag_arc_amd.encapsulation.database.vector_db import HipFlatVectorStore

    n, d, k = 1_000_000, 128, 10
    emb = _SynthEmbeddings(d)
    store = HipFlatVectorStore(emb)
    for s0 in range(0, n, 250_000):
        store.add_texts([f"doc{i}" for i in range(s0, s0 + 250_000)], ids=[str(i) for i in range(s0, s0 + 250_000)])
    assert store.ntotal == n
    r = VectorStoreRetriever(store)
    got = [None] * 256
    go = threading.Barrier(256)

    def call(i):
        go.wait()
        got[i] = [d_.id for d_ in r.invoke(f"q{i}", k=k)]

    def storm():
        threads = [threading.Thread(target=call, args=(i,)) for i in range(256)]
        for t in threads:
            t.start()
        for t in threads:
            t.join()

    _, launches = _scan_launches(storm)
    scans_per_search = 2          # 1M x 128 rows, k = 10: the int8 path below the cascade's first cut = the hybrid's two stages
                                  # (fp16 scan of the first eighth, int8 scan of the rest: DESIGN 4.2), no cascade
    assert store.coalesced_launches[1] == 256 and store.coalesced_launches[0] <= 3, store.coalesced_launches
    assert launches <= 3 * scans_per_search, launches
    assert len(emb.query_batches) == store.coalesced_launches[0]            # the queries of a launch were embedded together
    # the answers are those of 256 one-query searches
    plain = HipFlatVectorStore(emb, coalesce=False)
    plain.index = store.index
    plain.docstore, plain.index_to_docstore_id = store.docstore, store.index_to_docstore_id
    for i in range(0, 256, 5):
        assert got[i] == [d_.id for d_ in plain.similarity_search(f"q{i}", k=k)], i
    # and batch_invoke gives them in one pass (256 queries: one search = the two scan stages)
    many, launches = _scan_launches(lambda: r.batch_invoke([f"q{i}" for i in range(256)], k=k))
    assert launches == scans_per_search and [[d_.id for d_ in docs] for docs in many] == got


def test_batch_invoke_multipath_fuses_all_queries_in_one_launch():
    from rag_arc_amd.core.retrieval.dense import VectorStoreRetriever
    from rag_arc_amd.core.retrieval.multipath import MultiPathRetriever
    from rag_arc_amd.core.utils.fusion import HipRRFusion
    from rag_arc_amd.encapsulation.database.vector_db import HipFlatVectorStore

    emb_a, emb_b = _SynthEmbeddings(64), _SynthEmbeddings(96)
    stores = []
    for emb, n in ((emb_a, 5000), (emb_b, 3000)):
        st = HipFlatVectorStore(emb)
        st.add_texts([f"doc{i}" for i in range(n)], ids=[str(i) for i in range(n)])
        stores.append(st)
    mp_r = MultiPathRetriever([VectorStoreRetriever(s) for s in stores], fusion_method=HipRRFusion(), top_k_per_retriever=30)
    queries = [f"q{i}" for i in range(300)]                       # more than one 256-query launch
    want = [[d.id for d in mp_r.invoke(q, top_k=12)] for q in queries[:40]]
    got = mp_r.batch_invoke(queries, top_k=12)
    assert [[d.id for d in docs] for docs in got[:40]] == want and all(len(docs) == 12 for docs in got)


def _sharded_worker(rank, world, port, cfg_path, out_dir):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), WORLD_SIZE=str(world), RANK=str(rank), LOCAL_RANK=str(rank))
    import torch.distributed as dist

    from rag_arc_amd.config.app_registration import register_sharded_vectorstore, registrator
    from rag_arc_amd.core.retrieval.dense import VectorStoreRetriever

    register_sharded_vectorstore(cfg_path, "sharded")
    store = registrator.get_object("sharded").impl
    assert store.shard[:2] == (rank, world) and store.shard[3] == 4001 and store.shard[2] in (2000, 2001)
    r = VectorStoreRetriever(store)
    queries = [f"query {i}" for i in range(300)]        # two chunks of 256 / 44: the pipelined path (_PendingShard) on every rank
    one = [[(d.id, s) for d, s in store.similarity_search_with_score(q, k=25)] for q in queries[:6]]
    many = [[d.id for d in docs] for docs in r.batch_invoke(queries, k=25)]
    with open(os.path.join(out_dir, f"r{rank}.json"), "w") as fh:
        json.dump({"one": one, "many": many}, fh)
    dist.barrier()
    dist.destroy_process_group()


def test_sharded_store_from_json_on_two_ranks_equals_single_store(tmp_path):
    import torch.multiprocessing as mp

    from rag_arc_amd.core.retrieval.dense import VectorStoreRetriever
    from rag_arc_amd.encapsulation.database.vector_db import HipFlatVectorStore
    from rag_arc_amd.encapsulation.embeddings.table import TableEmbeddings

    rng = np.random.default_rng(31)
    n, d = 4001, 96
    texts = [f"chunk {i}" for i in range(n)] + [f"query {i}" for i in range(300)]
    vecs = rng.standard_normal((len(texts), d)).astype(np.float32)
    np.savez(tmp_path / "table.npz", texts=np.array(texts), vectors=vecs)
    np.savez(tmp_path / "corpus.npz", texts=np.array(texts[:n]), ids=np.array([str(i) for i in range(n)]))
    cfg = {"type": "hip_sharded_flat_vectorstore", "embedding": {"type": "table_embeddings", "path": str(tmp_path / "table.npz")},
           "corpus_path": str(tmp_path / "corpus.npz"), "backend": "gloo", "one_device": True}
    (tmp_path / "store.json").write_text(json.dumps(cfg))
    mp.spawn(_sharded_worker, args=(2, 29671, str(tmp_path / "store.json"), str(tmp_path)), nprocs=2, join=True)
    single = HipFlatVectorStore(TableEmbeddings(texts, vecs))
    single.add_texts(texts[:n], ids=[str(i) for i in range(n)])
    r = VectorStoreRetriever(single)
    queries = [f"query {i}" for i in range(300)]
    want_one = [[[d_.id, s] for d_, s in single.similarity_search_with_score(q, k=25)] for q in queries[:6]]
    want_many = [[d_.id for d_ in r.invoke(q, k=25)] for q in queries]
    for rank in range(2):
        got = json.loads((tmp_path / f"r{rank}.json").read_text())
        assert got["one"] == want_one, f"rank {rank}: (id, score) lists differ from the single store's"
        assert got["many"] == want_many, f"rank {rank}"


def test_2560_query_vectors_in_one_call_and_many_threads_keep_their_own_answers():
    """answers were handed out as views of a 6-slot pinned ring.  A call of more than 1536 queries kept
    seven chunks' views while chunks 7, 8 … were copied over the first ones; more than three threads in a batched call
    overwrote each other.  Staging slots are now owned by their search handle (engine._PinnedPool): ten chunks in one
    call, then eight threads at once, all equal to chunk-by-chunk searches — and an answer returned earlier does not
    change under later searches."""
    import torch

    from rag_arc_amd.encapsulation.database.vector_db import HipFlatVectorStore

    emb = _SynthEmbeddings(128)
    store = HipFlatVectorStore(emb, coalesce=False)
    n = 60_000
    store.add_texts([f"doc{i}" for i in range(n)], ids=[str(i) for i in range(n)])
    q = emb._rows(0, 2560, 4321)
    want_s, want_r = [], []
    for s0 in range(0, 2560, 256):
        s, r = store.index.search(q[s0:s0 + 256], 20)
        want_s.append(s), want_r.append(r)
    want_s, want_r = np.concatenate(want_s), np.concatenate(want_r)
    assert len({tuple(r) for r in want_r.tolist()}) > 2000        # (the queries really have different answers)
    got_s, got_r = store.batch_search_by_vector(q, 20)
    assert np.array_equal(got_r, want_r) and np.array_equal(got_s.view(np.uint32), want_s.view(np.uint32))
    one_s, one_r = store.batch_search_by_vector(q[:100], 20)       # one chunk: the caller owns these arrays
    keep_s, keep_r = one_s.copy(), one_r.copy()
    for s0 in range(256, 2560, 256):
        store.batch_search_by_vector(q[s0:s0 + 256], 20)
    assert np.array_equal(one_r, keep_r) and np.array_equal(one_s, keep_s)
    # eight threads, 768 queries each (three chunks per call), all at once
    out, errors = {}, []

    def work(t):
        try:
            torch.cuda.set_device(0)
            out[t] = store.batch_search_by_vector(q[t * 256:t * 256 + 768], 20)
        except Exception as exc:  # noqa: BLE001
            errors.append(exc)

    threads = [threading.Thread(target=work, args=(t,)) for t in range(8)]
    for th in threads:
        th.start()
    for th in threads:
        th.join()
    assert not errors, errors
    for t in range(8):
        assert np.array_equal(out[t][1], want_r[t * 256:t * 256 + 768]), t
        assert np.array_equal(out[t][0].view(np.uint32), want_s[t * 256:t * 256 + 768].view(np.uint32)), t
    assert store.index._pins.allocated <= 24, store.index._pins.allocated     # slots come back: the pool stays small


def test_an_index_with_pipelined_contexts_frees_its_rows_when_dropped():
    """search_async on the fp16-scan path runs over two internal search contexts that share the index's row tensors.  They see
    their index (and each other) through weak references: dropping the index frees its HBM at once — no cyclic-collector pass
    needed — and a handle that outlives the index still answers."""
    import gc

    import torch

    from rag_arc_amd.hip.engine import FlatIndexF16

    gc.collect()
    torch.cuda.empty_cache()
    gc.disable()
    try:
        base = torch.cuda.memory_allocated()
        g = torch.Generator(device="cuda").manual_seed(1)
        idx = FlatIndexF16(256, growable=False, scan="mfma16")                      # the fp16-scan path, whatever k
        idx.add(torch.randn((400_000, 256), device="cuda", generator=g))            # 200 MB of rows
        q = torch.randn((64, 256), device="cuda", generator=g)
        want = idx.search_device(q, 10)[0].clone()
        h1, h2 = idx.search_async(q, 10), idx.search_async(q, 10)
        assert idx._pair is not None                                               # the pipelined contexts exist
        assert torch.equal(h1.result()[0], want)
        held = torch.cuda.memory_allocated() - base
        assert held > 150 << 20
        del idx, h1
        left = torch.cuda.memory_allocated() - base
        assert torch.equal(h2.result()[0], want)                                   # the handle (and through it ONE context) lives on
        del h2
        torch.cuda.synchronize()
        assert torch.cuda.memory_allocated() - base < 8 << 20, (held, left, torch.cuda.memory_allocated() - base)
    finally:
        gc.enable()
