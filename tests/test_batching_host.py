"""Host logic of the batched / coalesced / sharded paths through the registered backend, on CPU with
the oracle standing in for the HIP index: concurrent one-query callers share scans, batch_invoke equals invoke query by
query, and a 2-rank (gloo) sharded store answers exactly like a single store."""
import os
import sys
import threading
import time

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

from tests.helpers import HashEmbeddings, OracleFusion, OracleIndex
from rag_arc_amd.core.retrieval.dense import VectorStoreRetriever
from rag_arc_amd.core.retrieval.multipath import MultiPathRetriever
from rag_arc_amd.encapsulation.database.vector_db import HipFlatVectorStore

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


class _SlowOracleIndex(OracleIndex):
    """An oracle index whose scan takes a while (so that callers pile up behind it) and counts its launches."""
    delay = 0.05
    gate = None      # a threading.Event the FIRST scan waits for (set once every caller thread is running): the test does not
                     # depend on how fast a loaded machine starts threads

    def __init__(self, *a, **kw):
        super().__init__(*a, **kw)
        self.launches, self.batch_sizes = 0, []

    def search(self, queries, k):
        self.launches += 1
        self.batch_sizes.append(len(queries))
        if self.gate is not None and self.launches == 1:
            self.gate.wait(timeout=30)
        time.sleep(self.delay)
        return super().search(queries, k)


def _run_gated(store, n, call):
    """n caller threads; the first scan is held until all of them are running (and have had `delay` to enqueue)."""
    started, lock = [0], threading.Lock()
    store.index.gate = threading.Event()

    def wrapped(i):
        with lock:
            started[0] += 1
            if started[0] == n:
                store.index.gate.set()
        call(i)

    threads = [threading.Thread(target=wrapped, args=(i,)) for i in range(n)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    store.index.gate = None


class _BatchingHashEmbeddings(HashEmbeddings):
    """A provider whose embed_query is embed_documents([text])[0] and that says so (embed_queries)."""
    def __init__(self, dim=384):
        super().__init__(dim)
        self.calls = []

    def embed_queries(self, texts):
        self.calls.append(len(texts))
        return self.embed_documents(list(texts))


def _store(n=400, dim=64, emb=None, engine=_SlowOracleIndex, **kw):
    emb = emb or HashEmbeddings(dim)
    store = HipFlatVectorStore(emb, engine_factory=lambda d, m, dev: engine(d, m, dev), **kw)
    store.add_texts([f"text number {i}" for i in range(n)], ids=[str(i) for i in range(n)])
    return store


def test_concurrent_one_query_callers_share_scans():
    store = _store()
    queries = [f"what about {i}?" for i in range(64)]
    ks = [1 + (i * 7) % 23 for i in range(64)]                      # every caller its own k: served as a prefix of the batch's
    got = [None] * 64

    def call(i):
        got[i] = store.similarity_search_with_score(queries[i], k=ks[i])

    _run_gated(store, 64, call)
    launches, served = store.coalesced_launches
    assert served == 64 and launches <= 3, (launches, store.index.batch_sizes)      # first caller alone, the rest together
    plain = _store(coalesce=False, engine=OracleIndex)
    for i in range(64):
        want = plain.similarity_search_with_score(queries[i], k=ks[i])
        assert [(d.id, s) for d, s in got[i]] == [(d.id, s) for d, s in want]
    # a lone caller is not delayed and not batched
    one = store.similarity_search("solo", k=3)
    assert len(one) == 3 and store.coalesced_launches == (launches + 1, 65)


def test_the_leader_waits_for_company_only_after_a_shared_launch():
    """window 0: a sequential caller is never delayed (the launch before it served one caller); after a launch that served
    several, the next idle-index leader waits while callers keep arriving (a quiet 0.2 ms ends the wait; at most 3/4 of that
    launch's duration)."""
    import time

    store = _store(n=50, dim=16)
    store.index.delay = 0.0
    co = store._coalescer
    t0 = time.perf_counter()
    for i in range(40):
        store.similarity_search(f"alone {i}", k=2)
    assert co.last_batch == 1 and time.perf_counter() - t0 < 2.0
    store.index.delay = 0.02                                         # a 20 ms scan: the leader may wait up to 15 ms, in 0.2 ms steps
    got = [None] * 48
    _run_gated(store, 48, lambda i: got.__setitem__(i, store.similarity_search(f"burst {i}", k=2)))
    assert co.last_batch > 1 and all(len(g) == 2 for g in got)
    waited = []
    real_wait = co.cv.wait
    co.cv.wait = lambda timeout=None: (waited.append(timeout), real_wait(timeout))[1]
    try:
        store.similarity_search("right after the burst", k=2)       # idle index, but the last launch was shared: waits
    finally:
        co.cv.wait = real_wait
    assert waited and 0 < waited[0] <= co.QUIET_S
    assert co.last_batch == 1                                        # ... and that one was alone again: no wait next time


def test_more_than_256_waiters_take_several_launches_and_window_mode():
    store = _store(n=50, dim=16, coalesce_window_us=20_000)         # 20 ms window: even the first caller waits for company
    store.index.delay = 0.0
    vecs = [HashEmbeddings(16)._one(f"q{i}").tolist() for i in range(300)]
    got = [None] * 300
    # the first scan is held until all 300 callers are running, so the pile behind it is > 256 whatever the machine's load
    _run_gated(store, 300, lambda i: got.__setitem__(i, store.similarity_search_by_vector(vecs[i], k=2)))
    launches, served = store.coalesced_launches
    assert served == 300 and max(store.index.batch_sizes) <= 256 and 2 <= launches <= 12, (launches, store.index.batch_sizes)
    assert sum(store.index.batch_sizes) == 300 and all(len(g) == 2 for g in got)


def test_text_payloads_are_embedded_once_per_batch():
    emb = _BatchingHashEmbeddings(64)
    store = _store(emb=emb)
    got = [None] * 40
    _run_gated(store, 40, lambda i: got.__setitem__(i, store.similarity_search(f"question {i}", k=4)))
    assert sum(emb.calls) == 40 and len(emb.calls) <= 3             # one encoder call per launch, not per caller
    plain = _store(coalesce=False, engine=OracleIndex)
    for i in range(40):
        assert [d.id for d in got[i]] == [d.id for d in plain.similarity_search(f"question {i}", k=4)]


def test_a_failing_scan_reaches_every_waiter():
    class Boom(_SlowOracleIndex):
        def search(self, queries, k):
            time.sleep(0.02)
            raise RuntimeError("scan failed")

    store = _store(engine=Boom)
    errors = []

    def call(i):
        try:
            store.similarity_search(f"q{i}", k=2)
        except RuntimeError as exc:
            errors.append(str(exc))

    threads = [threading.Thread(target=call, args=(i,)) for i in range(12)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    assert errors == ["scan failed"] * 12
    assert store._coalescer.busy is False and store._coalescer.queue == []


def test_batch_invoke_equals_invoke_query_by_query():
    store = _store(engine=OracleIndex)
    queries = [f"topic {i}" for i in range(9)]
    for search_type, kw in (("similarity", {}), ("mmr", {}), ("similarity_score_threshold", {"search_kwargs": {"score_threshold": 0.0}})):
        r = VectorStoreRetriever(store, search_type=search_type, **kw)
        want = [[d.id for d in r.invoke(q, k=6)] for q in queries]
        assert [[d.id for d in docs] for docs in r.batch_invoke(queries, k=6)] == want
    r = VectorStoreRetriever(store)
    assert [[d.id for d in docs] for docs in r.batch_invoke(queries)] == [[d.id for d in r.invoke(q)] for q in queries]   # default k = 5
    assert r.batch_invoke([]) == []
    other = VectorStoreRetriever(_store(n=300, dim=32, engine=OracleIndex))
    mp_r = MultiPathRetriever([r, other], fusion_method=OracleFusion(), top_k_per_retriever=20)
    want = [[d.id for d in mp_r.invoke(q, top_k=7)] for q in queries]
    assert [[d.id for d in docs] for docs in mp_r.batch_invoke(queries, top_k=7)] == want

    class Failing(VectorStoreRetriever):
        def batch_invoke(self, inputs, **kwargs):
            raise RuntimeError("down")

        def _get_relevant_documents(self, query, **kwargs):
            raise RuntimeError("down")

    mp_f = MultiPathRetriever([r, Failing(store)], fusion_method=OracleFusion(), top_k_per_retriever=20)
    assert [[d.id for d in docs] for docs in mp_f.batch_invoke(queries[:3], top_k=5)] == [[d.id for d in mp_f.invoke(q, top_k=5)] for q in queries[:3]]

    # ONE query a retriever cannot answer costs that query's list from that retriever — as under invoke() — not the whole
    # batch's: the retriever's batch call fails, the fan-out falls back to it query by query
    class OneBad(VectorStoreRetriever):
        def batch_invoke(self, inputs, **kwargs):
            if any("poison" in q for q in inputs):
                raise RuntimeError("cannot batch")
            return super().batch_invoke(inputs, **kwargs)

        def _get_relevant_documents(self, query, **kwargs):
            if "poison" in query:
                raise RuntimeError("cannot answer")
            return super()._get_relevant_documents(query, **kwargs)

    mixed = queries[:2] + ["poison pill"] + queries[2:4]
    mp_b = MultiPathRetriever([OneBad(store), other], fusion_method=OracleFusion(), top_k_per_retriever=20)
    want = [[d.id for d in mp_b.invoke(q, top_k=6)] for q in mixed]
    got = [[d.id for d in docs] for docs in mp_b.batch_invoke(mixed, top_k=6)]
    assert got == want and all(len(g) == 6 for g in got)                 # the poisoned query still gets the other path's answer
    # a dense retriever on its own: the batch equals invoke() element by element up to the query that raises
    with pytest.raises(RuntimeError):
        OneBad(store).batch_invoke(mixed)


# ------------------------------------------------------------------------------------------- sharded store, 2 ranks over gloo
def _cpu_merge(ids, scores, k):
    from oracle import cpu_ref

    i, s = cpu_ref.topk_merge(ids.numpy(), scores.numpy(), k)
    return torch.from_numpy(i), torch.from_numpy(s)


def _shard_worker(rank, world, port, out_dir):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    import torch.distributed as dist

    from tests.helpers import HashEmbeddings, OracleIndex
    from rag_arc_amd.core.retrieval.dense import VectorStoreRetriever
    from rag_arc_amd.encapsulation.database.vector_db.hip_sharded import HipShardedFlatVectorStore

    dist.init_process_group("gloo", rank=rank, world_size=world)
    store = HipShardedFlatVectorStore(HashEmbeddings(48), engine_factory=lambda d, m, dev: OracleIndex(d, m, dev), merge_fn=_cpu_merge)
    store.add_texts([f"passage {i}" for i in range(301)], ids=[str(i) for i in range(301)])
    store.add_texts([f"late passage {i}" for i in range(57)])          # a second block: ids derived from global row numbers
    assert store.shard == (rank, world, (151 if rank == 0 else 150) + (29 if rank == 0 else 28), 358)
    r = VectorStoreRetriever(store)
    queries = [f"query {i}" for i in range(11)]
    one = [[(d.id, d.content) for d in r.invoke(q, k=40)] for q in queries]
    many = [[(d.id, d.content) for d in docs] for docs in r.batch_invoke(queries, k=40)]
    scored = [[(d.id, s) for d, s in store.similarity_search_with_score(q, k=400)] for q in queries[:2]]   # k > ntotal: all 358 rows
    import pickle
    with open(os.path.join(out_dir, f"r{rank}.pkl"), "wb") as fh:
        pickle.dump((one, many, scored), fh)
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_sharded_store_equals_single_store(tmp_path):
    import pickle
    import uuid

    mp.spawn(_shard_worker, args=(2, 29541, str(tmp_path)), nprocs=2, join=True)
    single = HipFlatVectorStore(HashEmbeddings(48), engine_factory=lambda d, m, dev: OracleIndex(d, m, dev))
    single.add_texts([f"passage {i}" for i in range(301)], ids=[str(i) for i in range(301)])
    single.add_texts([f"late passage {i}" for i in range(57)],
                     ids=[str(uuid.uuid5(uuid.NAMESPACE_OID, f"rarc-row-{301 + i}")) for i in range(57)])
    r = VectorStoreRetriever(single)
    queries = [f"query {i}" for i in range(11)]
    want = [[(d.id, d.content) for d in r.invoke(q, k=40)] for q in queries]
    want_scored = [[(d.id, s) for d, s in single.similarity_search_with_score(q, k=400)] for q in queries[:2]]
    for rank in range(2):
        with open(tmp_path / f"r{rank}.pkl", "rb") as fh:
            one, many, scored = pickle.load(fh)
        assert one == want and many == want, f"rank {rank}"
        assert scored == want_scored and len(scored[0]) == 358


def test_256_coroutines_share_scans_without_a_thread_each():
    """`ainvoke` from one event loop: the callers enqueue and await, ONE worker thread answers them in a few scans — with
    the answers of `invoke`, the async quirks of the retriever kept (no default k, no truncation: dense.py:176-218), and a
    caller whose query cannot be answered failing alone."""
    import asyncio

    class Picky(_BatchingHashEmbeddings):
        def embed_queries(self, texts):
            if any("poison" in t for t in texts):
                raise KeyError("poison")
            return super().embed_queries(texts)

        def embed_query(self, text):
            if "poison" in text:
                raise KeyError("poison")
            return super().embed_query(text)

    emb = Picky(64)
    store = _store(emb=emb)
    store.index.delay = 0.02
    r = VectorStoreRetriever(store)
    queries = [f"question {i}" for i in range(256)]
    before_threads = threading.active_count()

    async def storm():
        peak = [0]

        async def one(q):
            out = await r.ainvoke(q, k=6)
            peak[0] = max(peak[0], threading.active_count())
            return out

        return await asyncio.gather(*[one(q) for q in queries]), peak[0]

    got, peak = asyncio.run(storm())
    assert [[d.id for d in docs] for docs in got] == [[d.id for d in r.invoke(q, k=6)] for q in queries]
    assert store.index.launches - 256 <= 6, store.index.batch_sizes[:8]        # (the 256 invokes above ran one scan each)
    assert peak <= before_threads + 2, "the async twins started a thread per caller"
    assert store._afront.served == 256 and store._afront.launches <= 6

    async def mixed():
        return await asyncio.gather(*[r.ainvoke(q, k=3) for q in ["question 1", "poison pill", "question 2"]], return_exceptions=True)

    a, b, c = asyncio.run(mixed())
    assert [d.id for d in a] == [d.id for d in r.invoke("question 1", k=3)] and isinstance(b, KeyError)
    assert [d.id for d in c] == [d.id for d in r.invoke("question 2", k=3)]
    # a provider that cannot batch keeps the reference's route (a pool thread per call), same answers
    plain = _store(engine=OracleIndex)
    rp = VectorStoreRetriever(plain)
    assert [d.id for d in asyncio.run(rp.ainvoke("question 9", k=4))] == [d.id for d in rp.invoke("question 9", k=4)]


def test_async_front_never_leaves_a_future_unresolved():
    """anything that is not an ordinary Exception inside the worker (here a BaseException out of the scan) used to
    kill the worker after the batch had left the queue — its callers awaited for ever.  Every caller of the batch is told,
    and a later caller gets a new worker."""
    import asyncio

    class Abort(BaseException):
        pass

    class Fatal(_SlowOracleIndex):
        armed = True

        def search(self, queries, k):
            if Fatal.armed:
                raise Abort("not an Exception")
            return super().search(queries, k)

    store = _store(emb=_BatchingHashEmbeddings(64), engine=Fatal)
    store.index.delay = 0.0
    r = VectorStoreRetriever(store)

    async def burst():
        return await asyncio.wait_for(asyncio.gather(*[r.ainvoke(f"question {i}", k=3) for i in range(5)],
                                                     return_exceptions=True), timeout=20)

    got = asyncio.run(burst())
    assert all(isinstance(g, RuntimeError) and "async search front failed" in str(g) for g in got), got
    assert all(isinstance(g.__cause__, Abort) for g in got)
    Fatal.armed = False
    docs = asyncio.run(asyncio.wait_for(r.ainvoke("question 1", k=3), timeout=20))
    assert [d.id for d in docs] == [d.id for d in r.invoke("question 1", k=3)]
