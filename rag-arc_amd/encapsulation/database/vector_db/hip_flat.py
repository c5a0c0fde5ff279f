"""HBM-resident exact vector store — the MI355X answer to the reference's
"TODO: needs GPU support to go faster" (encapsulation/database/vector_db/VectorStore_Faiss.py:14).

Same behaviour as FaissVectorStore for index_type "flat" with metric "cosine" or "ip"
(VectorStore_Faiss.py:65-512): embed -> fp32 -> (L2-normalise) -> add; query -> fp32 ->
(normalise) -> k = min(k, ntotal) -> search -> (Document, float(score)).  Rows are stored as fp16
in HBM and scored by the HIP kernels behind `FlatIndexF16`; scores are the canonical fp32 inner
products defined in DESIGN.md, ties ordered by insertion index.  IVF / HNSW / L2 are out of scope
(SURVEY.md §8 a4).  There is no CPU fallback: without a GPU the default engine raises.
"""
import logging
import os
import pickle
import threading
import time
import uuid
from typing import Any, Callable, List, Optional, Sequence, Tuple

import numpy as np

from ....core.utils.data_model import Document
from .base import VectorStore
from .docstore import ColumnarDocstore, rows_from_dicts

logger = logging.getLogger(__name__)


def _default_engine(dim: int, metric: str, device: int, storage: str = "f16", **engine_kwargs):
    from ....hip.engine import FlatIndexF16

    return FlatIndexF16(dim, metric=metric, device=device, storage=storage, **engine_kwargs)


def _mmr_select(scored, embeddings, query_embedding, k, lambda_mult=0.5):
    """Greedy maximal-marginal-relevance pick (reference: VectorStore_Faiss.py:16-62)."""
    if k >= len(scored):
        return [d for d, _ in scored]
    emb = np.asarray(embeddings, dtype=np.float64)
    qv = np.asarray(query_embedding, dtype=np.float64)
    chosen, rest = [0], list(range(1, len(scored)))
    while len(chosen) < k and rest:
        best, best_val = None, None
        for i in rest:
            redundancy = max(0, max(float(np.dot(emb[j], emb[i])) for j in chosen))
            val = lambda_mult * float(np.dot(qv, emb[i])) - (1 - lambda_mult) * redundancy
            if best_val is None or val > best_val:
                best, best_val = i, val
        chosen.append(best)
        rest.remove(best)
    return [scored[i][0] for i in chosen]


def _shard_files(folder: str, index_name: str) -> List[str]:
    """Every shard file of `index_name` in the folder, whatever layout wrote it."""
    import re

    pat = re.compile(re.escape(index_name) + r"(\.r\d+of\d+)?\.rarc$")
    return [os.path.join(folder, f) for f in sorted(os.listdir(folder)) if pat.fullmatch(f)]


def _belongs_to(path: str, index_name: str, world: int) -> bool:
    """Is `path` one of the files a save with `world` ranks writes?"""
    base = os.path.basename(path)
    if world == 1:
        return base == f"{index_name}.rarc"
    return any(base == f"{index_name}.r{r}of{world}.rarc" for r in range(world))


class _QueryCoalescer:
    """Gathers the queries of CONCURRENT callers into one scan.

    The reference issues one query per `index.search` (VectorStore_Faiss.py:258-263) and its `ainvoke` runs every call on
    a thread of its own (core/retrieval/base.py:82-96), so a busy service holds many one-query searches at once — each a
    full pass over the corpus.  Here a caller that finds the index idle launches at once (no added latency, alone if
    nobody else is waiting); callers that arrive WHILE a scan is running queue up, and whoever is woken first takes the
    whole queue — up to 256 queries — through ONE scan.  256 threads calling together cost two or three scans instead
    of 256.  `window_s` > 0 additionally lets an idle-index leader wait that long for company; with window_s = 0 the leader
    still waits when the launch before this one served more than one caller (a burst is likely under way) — for as long as
    callers keep arriving, at most 3/4 of the last launch's duration: at 100M rows 256 threads then share ONE scan instead
    of 6 + 250.  A lone sequential caller never waits.

    An exact top-k under a total order (score desc, id asc) is a prefix of the top-k' for k' > k, so one launch with the
    largest k of the batch serves every caller."""

    def __init__(self, run_batch: Callable, max_batch: int = 256, window_s: float = 0.0):
        self.run_batch, self.max_batch, self.window_s = run_batch, int(max_batch), float(window_s)
        self.cv = threading.Condition()
        self.queue: list = []
        self.busy = False
        self.launches = 0        # scans issued
        self.served = 0          # queries answered
        self.last_batch = 1      # callers served by the previous launch, and how long it took
        self.last_scan_s = 0.0

    QUIET_S = 200e-6         # a leader that expects company stops waiting after this long without a new caller
    BURST_SHARE = 0.75       # ... and in any case after this share of the last launch's duration

    class _Item:
        __slots__ = ("payload", "k", "result", "error", "done")

        def __init__(self, payload, k):
            self.payload, self.k, self.result, self.error, self.done = payload, k, None, None, False

    def submit(self, payload, k: int):
        """payload: a query text or a query vector; returns that query's [(Document, score)], at most k of them (the
        leader of a batch turns the whole answer into Documents in one native call: a waiter wakes up to a finished list)."""
        item = self._Item(payload, int(k))
        while True:
            with self.cv:
                if item not in self.queue and not item.done:
                    self.queue.append(item)
                while not item.done and self.busy:
                    self.cv.wait()
                if item.done:
                    break
                self.busy = True                       # this caller leads the next launch
                if self.window_s > 0:
                    if len(self.queue) < self.max_batch:
                        self.cv.wait(timeout=self.window_s)
                elif self.last_batch > 1:
                    # a burst is likely under way (the launch before this one served several callers): collect it — wait
                    # while callers keep arriving (a quiet interval of QUIET_S ends the wait), for at most three quarters
                    # of what the last launch took: 256 python threads need ~20 ms to get through their own prologues,
                    # a 100M-row scan is 27 ms — one scan of 256 beats a scan of six followed by a scan of 250
                    t_end = time.perf_counter() + self.BURST_SHARE * self.last_scan_s
                    quiet = min(self.QUIET_S, self.BURST_SHARE * self.last_scan_s)
                    while len(self.queue) < self.max_batch and time.perf_counter() < t_end:
                        n0 = len(self.queue)
                        self.cv.wait(timeout=quiet)
                        if len(self.queue) == n0:
                            break
                batch = self.queue[: self.max_batch]   # FIFO (the leader's own item is among them unless > max_batch queued)
                del self.queue[: self.max_batch]
            t_launch = time.perf_counter()
            try:
                results = self.run_batch([it.payload for it in batch], max(it.k for it in batch))
                for it, pairs in zip(batch, results):
                    it.result = pairs[: it.k]
            except BaseException as exc:  # noqa: BLE001 - every waiter of the batch must learn of it
                if len(batch) == 1 or not isinstance(exc, Exception):
                    for it in batch:
                        it.error = exc
                else:
                    # one caller's bad query (an unknown text, a vector of the wrong width) must fail THAT caller only, as
                    # it would have without the coalescer: the batch is answered again, one query per launch
                    for it in batch:
                        try:
                            it.result = self.run_batch([it.payload], it.k)[0][: it.k]
                        except BaseException as exc_one:  # noqa: BLE001
                            it.error = exc_one
            finally:
                with self.cv:
                    self.launches += 1
                    self.served += len(batch)
                    self.last_batch, self.last_scan_s = len(batch), time.perf_counter() - t_launch
                    for it in batch:
                        it.done = True
                    self.busy = False
                    self.cv.notify_all()
            if item.done:
                break
        if item.error is not None:
            raise item.error
        return item.result


class _RowsGuard:
    """Readers-writer guard over a store's rows AND its row -> Document list.  A search holds it shared from its launch until
    its answer has become Documents; add_texts / delete hold it exclusively — a delete renumbers every later row, and an
    answer whose row numbers were taken before it must not be looked up in the list after it.  (The reference is not
    thread-safe at all — plain dicts, framework/register.py — but this store invites concurrent callers: the coalescer, the
    async front, two-caller pipelining.)  Writers are re-entrant (delete -> rebuild -> add_texts) and have priority."""

    def __init__(self):
        self.cv = threading.Condition()
        self.readers = 0
        self.writer: Optional[int] = None
        self.depth = 0
        self.waiting_writers = 0
        self.held = threading.local()       # .n: how many times THIS thread holds the guard shared (nested searches)

    class _Shared:
        def __init__(self, g):
            self.g = g

        def __enter__(self):
            g = self.g
            with g.cv:
                me = threading.get_ident()
                if g.writer == me:          # the writer's own searches (a rebuild calling search): already exclusive
                    g.depth += 1
                    return
                n = getattr(g.held, "n", 0)
                if n:                       # a reader's nested search: waiting for a queued writer here would wait for itself
                    g.held.n = n + 1
                    return
                while g.writer is not None or g.waiting_writers:
                    g.cv.wait()
                g.readers += 1
                g.held.n = 1

        def __exit__(self, *exc):
            g = self.g
            with g.cv:
                if g.writer == threading.get_ident():
                    g.depth -= 1
                    return
                g.held.n -= 1
                if g.held.n:
                    return
                g.readers -= 1
                if g.readers == 0:
                    g.cv.notify_all()

    class _Exclusive:
        def __init__(self, g):
            self.g = g

        def __enter__(self):
            g = self.g
            with g.cv:
                me = threading.get_ident()
                if g.writer == me:
                    g.depth += 1
                    return
                g.waiting_writers += 1
                while g.writer is not None or g.readers:
                    g.cv.wait()
                g.waiting_writers -= 1
                g.writer, g.depth = me, 1

        def __exit__(self, *exc):
            g = self.g
            with g.cv:
                g.depth -= 1
                if g.depth == 0:
                    g.writer = None
                    g.cv.notify_all()

    def shared(self):
        return self._Shared(self)

    def exclusive(self):
        return self._Exclusive(self)


class _AsyncFront:
    """The coalescer for coroutines: callers enqueue (payload, k, future) and await; one worker thread, started when there is
    work and gone after a second without any, answers up to `max_batch` of them per scan."""

    IDLE_S = 1.0
    QUIET_S = 200e-6         # the worker stops collecting a burst after this long without a new caller
    BURST_MIN_S = 2e-3       # ... and after max(this, 3/4 of the last launch's duration) in any case

    def __init__(self, run_batch: Callable, max_batch: int = 256):
        self.run_batch, self.max_batch = run_batch, int(max_batch)
        self.cv = threading.Condition()
        self.queue: list = []
        self.worker: Optional[threading.Thread] = None
        self.launches = 0
        self.served = 0
        self.last_batch = 1      # callers served by the previous launch, and how long it took
        self.last_scan_s = 0.0

    async def submit(self, payload, k: int):
        import asyncio

        loop = asyncio.get_running_loop()
        fut = loop.create_future()
        with self.cv:
            self.queue.append((payload, int(k), fut, loop))
            if self.worker is None or not self.worker.is_alive():
                self.worker = threading.Thread(target=self._work, name="rarc-async-front", daemon=True)
                self.worker.start()
            self.cv.notify()
        return await fut

    @staticmethod
    def _deliver(fut, result, error) -> None:
        if fut.cancelled():
            return
        if error is not None:
            fut.set_exception(error)
        else:
            fut.set_result(result)

    @classmethod
    def _deliver_many(cls, items) -> None:
        for fut, res, err in items:
            cls._deliver(fut, res, err)

    def _work(self) -> None:
        while True:
            with self.cv:
                if not self.queue:
                    self.cv.wait(timeout=self.IDLE_S)
                if not self.queue:
                    self.worker = None
                    return
                if len(self.queue) < self.max_batch and (len(self.queue) > 1 or self.last_batch > 1):
                    # a burst is arriving (the event loop is still running its callers up to their awaits; the interpreter may
                    # hand this thread the GIL in the middle of that): collect it — wait while callers keep coming, a quiet
                    # QUIET_S ends the wait.  A lone awaiting caller after a lone one is launched at once.
                    t_end = time.perf_counter() + max(self.BURST_MIN_S, 0.75 * self.last_scan_s)
                    while len(self.queue) < self.max_batch and time.perf_counter() < t_end:
                        n0 = len(self.queue)
                        self.cv.wait(timeout=self.QUIET_S)
                        if len(self.queue) == n0:
                            break
                batch = self.queue[: self.max_batch]
                del self.queue[: self.max_batch]
            try:
                self._serve(batch)
            except BaseException as exc:  # noqa: BLE001 - the batch's callers have been told (_serve's finally); the worker
                # stays: a thread that is on its way out still reads as alive to submit(), which would then queue a
                # caller for nobody
                logger.error("async search front: a batch failed outside the per-query handling: %r", exc)

    def _serve(self, batch) -> None:
        """Answer one dequeued batch.  Whatever happens in here — a BaseException out of the scan, a loop object that
        raises something other than RuntimeError, a bug in the delivery code — every caller of the batch gets an answer or
        an exception: a future nobody resolves is a coroutine that hangs for ever."""
        settled: set = set()
        failure: Optional[BaseException] = None
        try:
            t_launch = time.perf_counter()
            outcomes = []
            try:
                results = self.run_batch([p for p, _, _, _ in batch], max(k for _, k, _, _ in batch))
                outcomes = [(pairs[:k], None) for (_, k, _, _), pairs in zip(batch, results)]
            except Exception:  # noqa: BLE001 - one caller's bad query fails that caller only: again, one by one
                for p, k, _, _ in batch:
                    try:
                        outcomes.append((self.run_batch([p], k)[0][:k], None))
                    except Exception as exc_one:  # noqa: BLE001
                        outcomes.append((None, exc_one))
            self.launches += 1
            self.served += len(batch)
            self.last_batch, self.last_scan_s = len(batch), time.perf_counter() - t_launch
            per_loop: dict = {}          # ONE wake-up per event loop, not one per future (each is a write to the loop's pipe)
            for (_, _, fut, loop), (res, err) in zip(batch, outcomes):
                per_loop.setdefault(loop, []).append((fut, res, err))
            for loop, items in per_loop.items():
                try:
                    loop.call_soon_threadsafe(self._deliver_many, items)
                except RuntimeError:          # the callers' loop is closed: nobody is waiting any more
                    pass
                settled.update(id(fut) for fut, _, _ in items)
        except BaseException as exc:  # noqa: BLE001
            failure = exc
            raise
        finally:
            left = [(fut, loop) for _, _, fut, loop in batch if id(fut) not in settled]
            if left:
                err = RuntimeError(f"the async search front failed while serving this batch: {failure!r}")
                err.__cause__ = failure
                for fut, loop in left:
                    try:
                        loop.call_soon_threadsafe(self._deliver, fut, None, err)
                    except BaseException:  # noqa: BLE001 - that caller's loop is gone (or broken): nobody to tell
                        pass


class HipFlatVectorStore(VectorStore):
    def __init__(self, embedding, metric: str = "cosine", normalize_L2: bool = False, index_type: str = "flat",
                 device: int = 0, engine_factory: Optional[Callable] = None, storage: str = "f16",
                 coalesce: bool = True, coalesce_window_us: float = 0.0, capacity: int = 0, max_rows: int = 0,
                 growable: Optional[bool] = None, **kwargs: Any):
        super().__init__(**kwargs)
        # row storage of the default engine (hip/engine.py): `capacity` rows backed right away, `growable` = rows in a
        # virtual-memory arena that grows in place up to `max_rows` (0 = what the device could hold) — appending never
        # copies the rows already there (the reference's index.add, VectorStore_Faiss.py:199-202, grows a std::vector)
        self._engine_kwargs = {key: v for key, v in (("capacity", int(capacity)), ("max_rows", int(max_rows)),
                                                     ("growable", growable)) if v}
        # concurrent one-query callers (the reference's only calling pattern) share scans: see _QueryCoalescer
        self._coalescer = (_QueryCoalescer(self._run_query_batch, 256, coalesce_window_us * 1e-6) if coalesce else None)
        if storage not in ("f16", "f8", "f32"):
            raise ValueError(f"unsupported row storage: {storage}")
        # "f16"; "f8": e4m3fn bytes + one scale per row (half the HBM footprint); "f32": the reference's own fp32
        # rows (VectorStore_Faiss.py:170) — returned scores carry no storage rounding
        self.storage = storage
        if index_type != "flat":
            raise ValueError(f"unsupported index type: {index_type} (exact flat scan only)")
        if metric not in ("cosine", "ip"):
            raise ValueError(f"unsupported metric: {metric}")
        self.embedding = embedding
        self.metric, self.normalize_L2, self.index_type, self.device = metric, normalize_L2, index_type, device
        self._engine_factory = engine_factory or _default_engine
        self.index = None  # created on first add, like the reference
        self.docstore: dict = {}
        self._i2d: Optional[dict] = {}       # index_to_docstore_id (a property: rebuilt from the row list when asked for)
        # id -> slot, for delete(): a row's SLOT is the ordinal it was inserted with and never changes; its row number is
        # the slot minus the deleted slots before it (rows only ever move down over holes).  An id used for several rows
        # maps to a list.  None = not built yet (a loaded / adopted store builds it at its first delete).
        self._slot_of_id: Optional[dict] = {}
        self._dead_slots = np.zeros(0, dtype=np.int64)
        self._next_slot = 0
        # row -> Document as a SEQUENCE (docstore.py): what a whole answer is mapped through in one native call.  A list of
        # the very objects the two dicts above hold, rebuilt from them whenever it may have fallen behind (a re-used id,
        # dicts assigned from outside); or a ColumnarDocstore for corpus-scale stores (adopt()), the dicts then stay empty
        self._row_docs = []
        self._row_docs_stale = False
        self._guard = _RowsGuard()
        self.timing: Optional[dict] = None   # set to {} to have the batch entry points add up their host-side phases (bench.py)

    # ------------------------------------------------------------------ helpers
    def _engine_metric(self) -> str:
        return "cosine" if (self.metric == "cosine" or self.normalize_L2) else "ip"

    @property
    def ntotal(self) -> int:
        return 0 if self.index is None else self.index.ntotal

    # ------------------------------------------------------------------ ingestion
    def add_texts(self, texts, metadatas=None, *, ids=None, **kwargs: Any) -> List[str]:
        texts = list(texts)
        if not texts:
            return []
        with self._guard.exclusive():
            return self._add_texts_locked(texts, metadatas, ids)

    def _add_texts_locked(self, texts, metadatas, ids) -> List[str]:
        if hasattr(self.embedding, "embed_documents_device"):
            vectors = self.embedding.embed_documents_device(texts)   # stays in HBM: fp32 [n][d] device tensor
        else:
            vectors = np.array(self.embedding.embed_documents(texts)).astype(np.float32)
        if ids is None:
            ids = [str(uuid.uuid4()) for _ in texts]
        elif len(ids) != len(texts):
            raise ValueError("number of ids must match number of texts")
        if metadatas is None:
            metadatas = [{} for _ in texts]
        elif len(metadatas) != len(texts):
            raise ValueError("number of metadatas must match number of texts")
        if self.index is None:
            self.index = self._make_engine(vectors.shape[1])
        start = self.index.ntotal
        self.index.add(vectors)
        self._remember(start, texts, metadatas, ids)
        return list(ids)

    @property
    def index_to_docstore_id(self) -> dict:
        """row -> document id (VectorStore_Faiss.py:97).  The row LIST is what searches go through; this dict is its
        reference-shaped twin, materialised when somebody asks (the pickle of save_local, user code) — a delete renumbers
        every later row, which costs nothing until then."""
        if self._i2d is None:
            seq = self._row_docs
            self._i2d = {r: d.id for r, d in enumerate(seq)} if isinstance(seq, list) else {}
        return self._i2d

    @index_to_docstore_id.setter
    def index_to_docstore_id(self, value: dict) -> None:
        self._i2d = value
        self._row_docs_stale = True         # dicts handed in from outside: the row list follows them at the next search
        self._slot_of_id = None

    def _remember(self, start: int, texts, metadatas, ids) -> None:
        """docstore + index_to_docstore_id entries of rows start.. (VectorStore_Faiss.py:205-208), and the row list."""
        if isinstance(self._row_docs, ColumnarDocstore):
            raise ValueError("this store holds a columnar (read-only) docstore: it cannot be added to")
        rows_ok = not self._row_docs_stale and len(self._row_docs) == start
        if not rows_ok:
            self._slot_of_id = None
        slots = self._slot_of_id
        for i, (text, meta, doc_id) in enumerate(zip(texts, metadatas, ids)):
            if doc_id in self.docstore and rows_ok:
                # an id used again: its EARLIER rows now answer with the new Document too (docstore[id] is overwritten) —
                # the dicts become the truth, the row list is rebuilt from them at the next search
                _ = self.index_to_docstore_id
                rows_ok, slots, self._slot_of_id = False, None, None
            doc = self.docstore[doc_id] = Document(content=text, metadata=meta, id=doc_id)
            if self._i2d is not None:
                self._i2d[start + i] = doc_id
            if rows_ok:
                self._row_docs.append(doc)
                if slots is not None:
                    slots[doc_id] = self._next_slot + i
        self._next_slot += len(texts)
        self._row_docs_stale = not rows_ok

    def _docs_by_row(self):
        """The row-indexed docstore sequence, brought up to date with the dicts if need be."""
        seq = self._row_docs
        if isinstance(seq, list) and self._row_docs_stale:
            seq = self._row_docs = rows_from_dicts(self.docstore, self.index_to_docstore_id)
            self._row_docs_stale = False
        return seq

    def _rows_of_ids(self, ids):
        """(current row numbers, slots) of the documents with these ids (each id names exactly one row: see delete)."""
        seq = self._docs_by_row()
        if self._slot_of_id is None:               # a loaded / adopted store: slots = today's row numbers
            self._slot_of_id = {d.id: r for r, d in enumerate(seq)}
            self._dead_slots, self._next_slot = np.zeros(0, dtype=np.int64), len(seq)
        slots = np.fromiter((self._slot_of_id[i] for i in ids), dtype=np.int64, count=len(ids))
        return slots - np.searchsorted(self._dead_slots, slots), slots

    def adopt(self, index, row_docs) -> "HipFlatVectorStore":
        """Corpus-scale construction: take an engine whose rows are already in HBM (loaded from a shard file, generated,
        ingested in slabs) together with the row-indexed docstore of those rows (a list of Documents or a
        ColumnarDocstore).  The reference has no counterpart — its only way in is add_texts (VectorStore_Faiss.py:156-210)."""
        if len(row_docs) != index.ntotal:
            raise ValueError(f"{len(row_docs)} documents for {index.ntotal} rows")
        self.index = index
        self.docstore, self._i2d, self._slot_of_id = {}, None, None
        if isinstance(row_docs, ColumnarDocstore):
            self._row_docs = row_docs
        else:
            self._row_docs = list(row_docs)
            self.docstore = {d.id: d for d in self._row_docs}
            if len(self.docstore) != len(self._row_docs):
                raise ValueError("adopt(): document ids must be distinct")
        self._row_docs_stale = False
        return self

    # ------------------------------------------------------------------ search
    def similarity_search(self, query: str, k: int = 4, **kwargs: Any) -> List[Document]:
        return [d for d, _ in self.similarity_search_with_score(query, k, **kwargs)]

    def similarity_search_with_score(self, query: str, k: int = 4, **kwargs: Any) -> List[Tuple[Document, float]]:
        if self.ntotal == 0:
            return []
        if self._coalescer is not None and self._batch_embedder() is not None:
            # text goes into the queue: the leader embeds the whole batch in one encoder call, then scans once
            return self._coalescer.submit(query, min(k, self.ntotal))
        return self.similarity_search_by_vector_with_score(self.embedding.embed_query(query), k, **kwargs)

    # -- async twins without a thread per call ------------------------------------------------------------------------
    # The reference's async twins run the sync method on a fresh ThreadPoolExecutor per call (VectorStoreBase.py:250-256):
    # 256 concurrent `ainvoke`s are 256 threads, and python wakes them one at a time (measured: 24 ms for 256 one-query
    # answers over a 1M-row index whose scan takes 0.5 ms).  Here a coroutine leaves its query in a queue and awaits a
    # future; ONE worker thread takes whatever has gathered — up to 256 queries — through one encoder call and one scan and
    # resolves the futures.  Same answers as the sync path (the same _run_query_batch), no thread per caller.
    async def asimilarity_search_with_score(self, *args: Any, **kwargs: Any) -> List[Tuple[Document, float]]:
        query = args[0] if args else kwargs.get("query")
        k = args[1] if len(args) > 1 else kwargs.get("k", 4)
        if self._coalescer is None or not isinstance(query, str) or self._batch_embedder() is None or len(args) > 2:
            return await super().asimilarity_search_with_score(*args, **kwargs)
        if self.ntotal == 0:
            return []
        return await self._async_front().submit(query, min(int(k), self.ntotal))

    async def asimilarity_search(self, query: str, k: int = 4, **kwargs: Any) -> List[Document]:
        if self._coalescer is None or self._batch_embedder() is None:
            return await super().asimilarity_search(query, k, **kwargs)
        return [d for d, _ in await self.asimilarity_search_with_score(query, k)]

    def _async_front(self) -> "_AsyncFront":
        front = self.__dict__.get("_afront")
        if front is None:
            front = self.__dict__.setdefault("_afront", _AsyncFront(self._run_query_batch))
        return front

    def similarity_search_by_vector(self, embedding: List[float], k: int = 4, **kwargs: Any) -> List[Document]:
        return [d for d, _ in self.similarity_search_by_vector_with_score(embedding, k, **kwargs)]

    def similarity_search_by_vector_with_score(self, embedding, k: int = 4, **kwargs: Any):
        if self.ntotal == 0:
            return []
        k = min(k, self.ntotal)
        if self._coalescer is not None:
            return self._coalescer.submit(np.asarray(embedding, dtype=np.float32), k)
        qv = np.array([embedding]).astype(np.float32)
        with self._guard.shared():
            scores, rows = self.index.search(qv, k)
            return self._to_documents(scores[0], rows[0])

    def _to_documents(self, scores, rows) -> List[Tuple[Document, float]]:
        """[(Document, float(score))] of one query's answer, row -1 skipped (VectorStore_Faiss.py:265-272)."""
        return self._map_batch(np.asarray(scores)[None, :], np.asarray(rows)[None, :], True)[0]

    def _map_batch(self, scores, rows, with_scores: bool):
        """A whole answer (scores fp32 [nq][k], rows int64 [nq][k]) -> per query [(Document, float)] or [Document], in one
        native call over the row-indexed docstore (csrc/hostmap.c)."""
        from ....hip import hostmap

        rows = np.ascontiguousarray(rows, dtype=np.int64)
        nq, k = rows.shape
        seq = self._docs_by_row()
        if isinstance(seq, ColumnarDocstore):
            seq = seq.columns()
        if with_scores:
            return hostmap.load().rows_to_pairs(seq, rows, np.ascontiguousarray(scores, dtype=np.float32), nq, k)
        return hostmap.load().rows_to_docs(seq, rows, nq, k)

    # -- batches: many queries, one scan (extension over the reference, which only has nq = 1) ----------------------
    def _batch_embedder(self) -> Optional[Callable]:
        """A function texts -> vectors that is the provider's embed_query applied to each text, if the provider has one
        (`embed_queries`); None otherwise (each caller then embeds its own query before queueing)."""
        return getattr(self.embedding, "embed_queries_device", None) or getattr(self.embedding, "embed_queries", None)

    def _run_query_batch(self, payloads: Sequence, k: int):
        """One scan for a batch of queries given as texts and / or vectors; returns [(Document, score)] x k per query."""
        texts = [i for i, p in enumerate(payloads) if isinstance(p, str)]
        if texts and len(texts) == len(payloads) and hasattr(self.embedding, "embed_queries_device"):
            q = self.embedding.embed_queries_device([payloads[i] for i in texts])        # stays in HBM
        else:
            vecs: list = list(payloads)
            if texts:
                emb = self._batch_embedder()([payloads[i] for i in texts])
                emb = emb.cpu().numpy() if hasattr(emb, "cpu") else emb
                for i, v in zip(texts, emb):
                    vecs[i] = np.asarray(v, dtype=np.float32)
            q = np.stack([np.asarray(v, dtype=np.float32) for v in vecs])
        with self._guard.shared():
            scores, rows = self.index.search(q, min(k, self.ntotal))
            return self._map_batch(scores, rows, True)

    def _search_chunks(self, q, k: int):
        """(scores fp32 [n][k], rows int64 [n][k]) numpy pairs, one per 256 queries of q (device tensor or array), in order.
        Chunk i+1 is on the GPU while chunk i is collected, copied out (pinned) and — in the caller — mapped to Documents.

        A yielded pair is a zero-copy VIEW of its search handle's pinned staging slot: it is the consumer's for the body
        of its loop iteration only.  The slot goes back to the index's pool when the generator is resumed (or closed);
        a consumer that keeps a pair must copy it (batch_search_by_vector does)."""
        idx = self.index
        k = min(int(k), self.ntotal)
        if not hasattr(idx, "search_async"):            # an engine with the numpy surface only
            yield idx.search(q, k)
            return
        step, pending = 256, None
        view = lambda h: h.host_view() if hasattr(h, "host_view") else h.host()
        done = lambda h: h.release() if hasattr(h, "release") else None
        try:
            for s0 in range(0, len(q), step):
                nxt = idx.search_async(q[s0:s0 + step], k, to_host=True)
                if pending is not None:
                    try:
                        yield view(pending)
                    finally:
                        done(pending)
                pending = nxt
            if pending is not None:
                last, pending = pending, None
                try:
                    yield view(last)
                finally:
                    done(last)
        finally:
            if pending is not None:     # the consumer stopped early: the launch in flight still has to let go of its slot
                done(pending)

    def _embed_batch(self, queries: List[str]):
        """Query texts -> vectors: a device tensor when the provider can keep them in HBM, else a float32 array."""
        if hasattr(self.embedding, "embed_queries_device"):
            return self.embedding.embed_queries_device(queries)
        if hasattr(self.embedding, "embed_queries"):
            return np.asarray(self.embedding.embed_queries(queries), dtype=np.float32)
        return np.asarray([self.embedding.embed_query(t) for t in queries], dtype=np.float32)

    def _tick(self, key: str, t0: float) -> float:
        """Add the time since t0 to self.timing[key] (when a caller asked for the breakdown); returns now."""
        import time

        now = time.perf_counter()
        if self.timing is not None:
            self.timing[key] = self.timing.get(key, 0.0) + (now - t0)
        return now

    def batch_search_by_vector(self, embeddings, k: int = 4):
        """Many query vectors, one scan per 256.  Returns (scores fp32 [nq][k], row indices int64 [nq][k])."""
        if self.ntotal == 0:
            nq = len(embeddings)
            return np.zeros((nq, 0), np.float32), np.zeros((nq, 0), np.int64)
        q = embeddings if hasattr(embeddings, "is_cuda") else np.asarray(embeddings, dtype=np.float32)
        nq, kk, at = len(q), min(int(k), self.ntotal), 0
        scores, rows = np.empty((nq, kk), np.float32), np.empty((nq, kk), np.int64)
        for sc, rw in self._search_chunks(q, k):        # (views of pinned staging: copied out before the next chunk is asked for)
            scores[at: at + len(sc)], rows[at: at + len(rw)] = sc, rw
            at += len(sc)
        return scores, rows

    def _batch_answers(self, queries: Sequence[str], k: int, with_scores: bool):
        import time

        queries = list(queries)
        if self.ntotal == 0 or not queries:
            return [[] for _ in queries]
        t0 = time.perf_counter()
        q = self._embed_batch(queries)
        t0 = self._tick("embed_s", t0)
        out: list = []
        with self._guard.shared():
            for scores, rows in self._search_chunks(q, k):
                t0 = self._tick("search_s", t0)          # launch + wait + pinned copy-out (the next chunk is already scanning)
                out.extend(self._map_batch(scores, rows, with_scores))
                t0 = self._tick("map_s", t0)
        return out

    def batch_similarity_search_with_score(self, queries: Sequence[str], k: int = 4, **kwargs: Any):
        """similarity_search_with_score for a list of queries: one encoder call (when the provider can batch), one scan per
        256 queries, each scan running while the answer before it becomes Documents.  Element i equals
        similarity_search_with_score(queries[i], k)."""
        return self._batch_answers(queries, k, True)

    def batch_similarity_search(self, queries: Sequence[str], k: int = 4, **kwargs: Any) -> List[List[Document]]:
        return self._batch_answers(queries, k, False)

    @property
    def coalesced_launches(self) -> Tuple[int, int]:
        """(scans issued, queries answered) by the coalescing front so far."""
        c = self._coalescer
        return (0, 0) if c is None else (c.launches, c.served)

    def max_marginal_relevance_search(self, query: str, k: int = 4, fetch_k: int = 20, lambda_mult: float = 0.5,
                                      **kwargs: Any) -> List[Document]:
        if self.ntotal == 0:
            return []
        return self.max_marginal_relevance_search_by_vector(self.embedding.embed_query(query), k, fetch_k,
                                                            lambda_mult, **kwargs)

    def max_marginal_relevance_search_by_vector(self, embedding, k: int = 4, fetch_k: int = 20,
                                                lambda_mult: float = 0.5, **kwargs: Any) -> List[Document]:
        """MMR as the reference does it (VectorStore_Faiss.py:276-372): fetch_k nearest, their texts
        re-embedded, greedy selection.  `reembed=False` (extension) takes the
        candidates' vectors from the HBM-resident rows instead of running the encoder on fetch_k texts
        again: the stored rows are those very embeddings after normalisation, rounded to the storage
        format (a relative 5e-4 for fp16), so the selection can differ from the reference's only where
        two candidates' MMR values are within that rounding."""
        reembed = kwargs.pop("reembed", True)
        if self.ntotal == 0:
            return []
        qv32 = np.array([embedding]).astype(np.float32)
        with self._guard.shared():
            scores, rows = self.index.search(qv32, min(fetch_k, self.ntotal))
            keep = [(float(sc), int(r)) for sc, r in zip(scores[0], rows[0]) if r != -1]
            by_row = self._docs_by_row()
            scored = [(by_row[r], sc) for sc, r in keep]
            if not scored:
                return []
            # (the row numbers are good for as long as the guard is held: the resident rows are read under it)
            if not reembed and getattr(self.index, "lib", None) is not None:
                # candidates stay in HBM: gather the resident rows, run the greedy selection in rarc_mmr_select
                order = self._mmr_on_device([r for _, r in keep], embedding, k, lambda_mult)
                return [scored[i][0] for i in order]
            if not reembed:
                cand = self._stored_vectors([r for _, r in keep])
        if reembed:
            cand = np.array([self.embedding.embed_query(d.content) for d, _ in scored])  # re-embedded, as the reference
        qv = np.array(embedding)
        if self.normalize_L2 or self.metric == "cosine":
            qv = qv / np.linalg.norm(qv)
            cand = cand / np.linalg.norm(cand, axis=1, keepdims=True)
        return _mmr_select(scored, cand.tolist(), qv.tolist(), k, lambda_mult)

    def _mmr_on_device(self, rows: List[int], embedding, k: int, lambda_mult: float) -> List[int]:
        """Selection order (indices into `rows`) of the MMR loop, computed on the device from the resident rows."""
        import torch

        from ....hip import binding as B

        idx = self.index
        if k >= len(rows):
            return list(range(len(rows)))
        dev = idx.rows.device
        sel = torch.as_tensor(rows, dtype=torch.long, device=dev)
        r = idx.rows[sel]
        if getattr(idx, "storage", "f16") == "f8":
            cand = r.view(torch.float8_e4m3fn).to(torch.float32) * idx.row_scales[sel][:, None]
        else:
            cand = r.to(torch.float32)
        cand = cand[:, : idx.dim].contiguous()
        q = torch.as_tensor(np.asarray(embedding, dtype=np.float64), device=dev)
        n, d = cand.shape
        work = torch.empty(int(idx.lib.rarc_mmr_workspace_doubles(n, d)), dtype=torch.float64, device=dev)
        out = torch.empty(min(k, n), dtype=torch.int32, device=dev)
        norm = 1 if (self.normalize_L2 or self.metric == "cosine") else 0
        with torch.cuda.device(dev):
            B.check(idx.lib.rarc_mmr_select(cand.data_ptr(), d, q.data_ptr(), n, d, norm, int(k), float(lambda_mult),
                                            work.data_ptr(), out.data_ptr(), torch.cuda.current_stream(dev).cuda_stream),
                    "rarc_mmr_select")
        return out.cpu().tolist()

    def _stored_vectors(self, rows: List[int]) -> np.ndarray:
        """Rows of the resident index as float64 [n][dim] (fp8 rows decoded and scaled)."""
        import torch

        idx = self.index
        sel = torch.as_tensor(rows, dtype=torch.long, device=idx.rows.device) if hasattr(idx.rows, "device") else rows
        r = idx.rows[sel]
        if getattr(idx, "storage", "f16") == "f8":
            vals = r.view(torch.float8_e4m3fn).to(torch.float32) * idx.row_scales[sel][:, None]
        else:
            vals = r.to(torch.float32) if hasattr(r, "to") else np.asarray(r).view(np.float16).astype(np.float32)
        vals = vals.cpu().numpy() if hasattr(vals, "cpu") else np.asarray(vals)
        return vals[:, : idx.dim].astype(np.float64)

    # ------------------------------------------------------------------ maintenance
    def delete(self, ids: Optional[List[str]] = None, **kwargs: Any) -> Optional[bool]:
        """FaissVectorStore.delete (VectorStore_Faiss.py:374-415): None clears the store, an unknown id anywhere in the list
        -> False and nothing changes, otherwise the documents go and True comes back.  The reference rebuilds the index by
        embedding every surviving text again; the surviving embeddings are in HBM already, so here the rows are compacted
        in place (rarc_compact_rows: stable, the survivors keep their order — exactly the rows a rebuild from the docstore
        would produce, without the encoder and without its batch-shape rounding noise) and the docstore follows."""
        with self._guard.exclusive():
            return self._delete_locked(ids)

    def _delete_locked(self, ids: Optional[List[str]]) -> Optional[bool]:
        if ids is None:
            self.docstore.clear()
            self._forget_rows()
            if self.index is not None:
                self.index.reset()
            return True
        if not ids:
            return True
        if any(i not in self.docstore for i in ids):
            return False
        if isinstance(self._row_docs, ColumnarDocstore):
            raise ValueError("this store holds a columnar (read-only) docstore: documents cannot be deleted from it")
        if len(self.docstore) != self.ntotal or not hasattr(self.index, "remove_rows"):
            return self._delete_by_rebuild(ids)
        gone = list(dict.fromkeys(ids))
        rows, slots = self._rows_of_ids(gone)
        self._remove_index_rows(rows)
        order = np.argsort(rows)
        rows, slots = rows[order], slots[order]
        for i in gone:
            del self.docstore[i]
            del self._slot_of_id[i]
        seq, kept, prev = self._row_docs, [], 0
        for r in rows.tolist():                    # the row list without the holes, slice by slice
            kept.extend(seq[prev:r])
            prev = r + 1
        kept.extend(seq[prev:])
        self._row_docs = kept
        self._dead_slots = np.union1d(self._dead_slots, slots)
        self._i2d = None                           # (every later row was renumbered: rebuilt when somebody asks for it)
        return True

    def _remove_index_rows(self, rows: np.ndarray) -> None:
        self.index.remove_rows(rows)

    def _forget_rows(self) -> None:
        self._row_docs, self._row_docs_stale, self._i2d = [], False, {}
        self._slot_of_id, self._dead_slots, self._next_slot = {}, np.zeros(0, dtype=np.int64), 0

    def _delete_by_rebuild(self, ids) -> bool:
        """The reference's own way (VectorStore_Faiss.py:390-413): reset, then add the surviving documents — one per ID, in
        docstore order — again.  Taken when an id names several rows (the rebuild collapses them into one: compaction
        cannot reproduce that) or the engine cannot compact."""
        drop = set(ids)
        keep = [d for i, d in self.docstore.items() if i not in drop]
        self.docstore.clear()
        self._forget_rows()
        if self.index is not None:
            self.index.reset()
        if keep:
            self.add_texts([d.content for d in keep], [d.metadata for d in keep], ids=[d.id for d in keep])
        return True

    def get_by_ids(self, ids: List[str]) -> List[Document]:
        if isinstance(self._row_docs, ColumnarDocstore):
            rows = [self._row_docs.row_of(i) for i in ids]
            return [self._row_docs[r] for r in rows if r is not None]
        return [self.docstore[i] for i in ids if i in self.docstore]

    def _select_relevance_score_fn(self):
        if self.metric == "cosine" or self.normalize_L2:
            return self._cosine_relevance_score_fn
        if self.metric == "ip":
            return self._max_inner_product_relevance_score_fn
        raise ValueError(f"unsupported metric: {self.metric}")

    def _make_engine(self, dim: int):
        if self._engine_factory is _default_engine:
            return _default_engine(dim, self._engine_metric(), self.device, self.storage, **self._engine_kwargs)
        if self.storage == "f16":  # (custom factories keep their three-argument signature)
            return self._engine_factory(dim, self._engine_metric(), self.device)
        return self._engine_factory(dim, self._engine_metric(), self.device, self.storage)

    # ------------------------------------------------------------------ persistence
    def _shard_layout(self) -> Tuple[int, int]:
        """(rank, world size) of this process: a single-GPU store is rank 0 of 1."""
        return 0, 1

    def _barrier(self) -> None:
        """All ranks of a sharded store meet here; nothing to do for one process."""

    def _local_engine(self):
        """The engine that holds THIS process's rows (the sharded store wraps it)."""
        return self.index

    def _local_blocks(self):
        """Id map of the local rows, [(first global id, count)]; None = one block at the engine's id_base."""
        return None

    def save_local(self, folder_path: str, index_name: str = "index") -> None:
        """`<index_name>.rarc` (hip/shardfile.py: 64-byte header, page-aligned raw rows, fp8 row scales, id map) + the
        pickled docstore and parameters, as the reference's `<index_name>.faiss` + `.pkl` (VectorStore_Faiss.py:432-450).
        The rows stream HBM -> pinned ring -> file inside the library (rarc_device_to_file): no host copy of the corpus
        exists, whatever its size.  The sharded store writes one file per rank (`<index_name>.r<rank>of<world>.rarc`)."""
        from ....hip import shardfile as SF

        os.makedirs(folder_path, exist_ok=True)
        rank, world = self._shard_layout()
        mine = SF.shard_path(folder_path, index_name, rank, world)
        fsync = bool(getattr(self, "durable_saves", False))
        # Order: every rank writes its new file aside-and-renamed, then the ranks AGREE on the outcome (a rank that failed —
        # no space, an I/O error — makes every rank raise instead of leaving the others in a barrier), then rank 0
        # replaces the pickle, and only after that do files of an EARLIER layout go: a failure at any point leaves the
        # previous save loadable.  (An earlier save under the SAME layout is overwritten file by file, as the reference
        # overwrites its index file in place, VectorStore_Faiss.py:438.)
        problem: Optional[BaseException] = None
        try:
            eng = self._local_engine()
            if eng is not None and self.ntotal:
                self.last_save_stats = eng.save_shard(mine, blocks=self._local_blocks(), rank=rank, world=world,
                                                      global_ntotal=self.ntotal, fsync=fsync)
            elif os.path.exists(mine):
                # the reference always rewrites the index file; an empty index must not leave the rows of an earlier save
                # behind for load_local to pick up next to an empty docstore
                os.unlink(mine)
        except BaseException as exc:  # noqa: BLE001 - reported to every rank below, then re-raised here
            problem = exc
        if not self._all_ranks_ok(problem is None):
            raise RuntimeError(f"save_local: rank {rank} of {world} could not write its shard file"
                               + (f": {problem}" if problem is not None else " (another rank failed)")) from problem
        if rank == 0:
            meta = {"docstore": self.docstore, "index_to_docstore_id": self.index_to_docstore_id,
                    "index_type": self.index_type, "metric": self.metric, "normalize_L2": self.normalize_L2,
                    "storage": self.storage, "world": world, "ntotal": self.ntotal,
                    # a columnar docstore travels as its columns (the dicts above are empty then)
                    "row_docs": self._row_docs if isinstance(self._row_docs, ColumnarDocstore) else None,
                    # what the searches so far have taught the index about this corpus (engine: sticky candidate capacity)
                    "cand_cap": int(getattr(self._local_engine(), "cand_cap", 0) or 0) if self.index is not None else 0,
                    # ... and whether that capacity was ever put to the test (a search overflowed and it grew, or warm_up ran)
                    "cand_cap_settled": bool(self.index is not None and (getattr(self._local_engine(), "cand_cap_grown", 0)
                                                                         or getattr(self._local_engine(), "warmed_up", False)))}
            tmp = os.path.join(folder_path, f"{index_name}.pkl.tmp")
            with open(tmp, "wb") as fh:
                pickle.dump(meta, fh)
                if fsync:
                    fh.flush()
                    os.fsync(fh.fileno())
            os.replace(tmp, os.path.join(folder_path, f"{index_name}.pkl"))
            if fsync:
                dfd = os.open(folder_path, os.O_RDONLY)
                try:
                    os.fsync(dfd)
                finally:
                    os.close(dfd)
            for old in _shard_files(folder_path, index_name):       # files of an earlier save under ANOTHER layout
                if not _belongs_to(old, index_name, world):
                    os.unlink(old)
        self._barrier()

    def _all_ranks_ok(self, ok: bool) -> bool:
        """Did every rank of a sharded store succeed?  (One process: its own outcome.)"""
        return ok

    @classmethod
    def load_local(cls, folder_path: str, embeddings, index_name: str = "index", **kwargs: Any):
        """Counterpart of FaissVectorStore.load_local (VectorStore_Faiss.py:452-482).  The rows stream file -> pinned ring ->
        HBM (rarc_file_to_device).  A save made under ANOTHER layout loads too — eight rank files into one GPU, one file
        into eight ranks: every process takes a contiguous range of the global ids and reads exactly those rows
        (shardfile.plan_reshard); search results do not depend on the layout."""
        from ....hip import shardfile as SF

        with open(os.path.join(folder_path, f"{index_name}.pkl"), "rb") as fh:
            meta = pickle.load(fh)
        kwargs.setdefault("storage", meta.get("storage", "f16"))
        warm_up = bool(kwargs.pop("warm_up", True))
        store = cls(embedding=embeddings, index_type=meta["index_type"], metric=meta["metric"],
                    normalize_L2=meta["normalize_L2"], **kwargs)
        store.docstore, store.index_to_docstore_id = meta["docstore"], meta["index_to_docstore_id"]   # (marks the row list stale)
        if meta.get("row_docs") is not None:
            store._row_docs, store._row_docs_stale = meta["row_docs"], False
        saved_world = int(meta.get("world", 1))
        files = [SF.shard_path(folder_path, index_name, r, saved_world) for r in range(saved_world)]
        files = [f for f in files if os.path.exists(f)]
        if not files:
            return store            # an empty index was saved (no shard file), as `index = None` in the reference (:467)
        headers = [SF.read_header(f) for f in files]
        for f, h in zip(files, headers):
            if h.storage != store.storage:
                raise ValueError(f"{f}: stored as {h.storage}, store configured for {store.storage}")
        total_saved = sum(h.n_rows for h in headers)
        if "ntotal" in meta and total_saved != int(meta["ntotal"]):
            raise ValueError(f"{folder_path}: the shard files hold {total_saved} rows, the store was saved with {meta['ntotal']} "
                             f"(a rank's file is missing?)")
        store.index = store._make_engine(headers[0].dim)
        rank, world = store._shard_layout()
        eng = store._local_engine()
        stats = []
        if world == saved_world and len(files) == saved_world:
            h = headers[rank]                       # same layout: this rank's own file, rows and id map as saved
            stats.append(eng.load_shard(files[rank], header=h))
            blocks, total = list(h.blocks), (h.global_ntotal if h.global_ntotal >= 0 else total_saved)
        else:
            segments, blocks, total = SF.plan_reshard([h.blocks for h in headers], rank, world)
            run: list = []                           # consecutive segments of one file go through one native call
            for seg in segments + [None]:
                if run and (seg is None or seg[0] != run[0][0]):
                    f = run[0][0]
                    stats.append(eng.load_shard(files[f], row_ranges=[(a, c) for _, a, c in run], header=headers[f]))
                    run = []
                if seg is not None:
                    run.append(seg)
        store._adopt_loaded(blocks, total)
        store.last_load_stats = stats
        saved_cap = int(meta.get("cand_cap", 0) or 0)
        if saved_cap > int(getattr(eng, "cand_cap", 0) or 0):
            # a capacity the saved index had grown to is not learnt a second time — as far as THIS device can hold its
            # workspace (a tenth of its memory at most: the file may come from a larger part)
            import torch

            lib, total = getattr(eng, "lib", None), 0
            if lib is not None and torch.cuda.is_available():
                total = int(torch.cuda.get_device_properties(eng.device).total_memory)
            while lib is not None and saved_cap > eng.cand_cap and int(lib.rarc_search_workspace_bytes(saved_cap)) > total // 10:
                saved_cap //= 2
            eng.cand_cap = max(int(eng.cand_cap), saved_cap)
        if not meta.get("cand_cap_settled", False) and warm_up and hasattr(eng, "warm_up") and eng.ntotal:
            # the file carries no learnt capacity (its index never overflowed and was never warmed up): learn it now, not
            # in the first user batch — one search of 64 stored rows on corpora the default capacity fits
            eng.warm_up()
        return store

    def warm_up(self, k: int = 100) -> int:
        """Search a sample of the stored rows as queries until the engine's candidate capacity has settled
        (FlatIndexF16.warm_up): on clustered corpora the first user batch then costs what every later one does.
        Optional; every rank of a sharded store calls it (no collective inside).  Returns the capacity doublings."""
        eng = self._local_engine() if self.index is not None else None
        return int(eng.warm_up(k)) if eng is not None and hasattr(eng, "warm_up") else 0

    def _adopt_loaded(self, blocks, total: int) -> None:
        """After load_local filled the local engine: a single-GPU store's rows ARE the global ids."""
        if blocks and (len(blocks) != 1 or blocks[0][0] != getattr(self.index, "id_base", 0)):
            raise ValueError("a single-GPU store holds one block of rows starting at its id base")

    @classmethod
    def from_texts(cls, texts, embedding, metadatas=None, *, ids=None, **kwargs: Any):
        store = cls(embedding=embedding, **kwargs)
        store.add_texts(texts, metadatas=metadatas, ids=ids)
        return store

    @classmethod
    async def afrom_texts(cls, texts, embedding, metadatas=None, *, ids=None, **kwargs: Any):
        store = cls(embedding=embedding, **kwargs)
        await store.aadd_texts(texts, metadatas=metadatas, ids=ids)
        return store
