"""Row-sharded form of HipFlatVectorStore: one process per GPU, every rank holds a slice of the rows.

The reference has one process and one index (VectorStore_Faiss.py:110-148); BASELINE config 4 / 5 put the 100M-row
corpus on the eight GPUs of a node.  This store is what a registered backend builds under `torchrun` (WORLD_SIZE > 1):
every rank runs the SAME program and issues the SAME calls in the same order (SPMD — the collective inside a search is
entered by all ranks together); every rank returns the SAME, globally merged answer, so whichever rank serves the
caller has it.

  add_texts   all ranks get the whole list; rank r embeds and stores rows [lo_r, hi_r) of it (hip.sharded.shard_range),
              every rank keeps the whole docstore (id -> Document: host memory, as in the reference)
  save_local  every rank streams ITS rows into `<name>.r<rank>of<world>.rarc` (rank 0 also writes the docstore pickle);
  load_local  every rank streams its rows back — from its own file when the world size is the one that saved, else from
              whichever files hold its range of the global ids (hip/shardfile.plan_reshard)
  search      replicated query -> local exact top-k (global ids, canonical scores) -> ONE all-gather of (id, score)
              records over RCCL / xGMI -> merge by (score desc, id asc): bit-identical to the single-shard answer for
              every world size (canonical scores do not depend on the sharding)
Concurrent callers are NOT coalesced here (threads would interleave the ranks' collectives differently on each rank);
`batch_similarity_search*` is the way to put many queries into one scan.
"""
from typing import Any, Callable, List, Optional

import numpy as np

from .hip_flat import HipFlatVectorStore


class _ShardedIndex:
    """What HipFlatVectorStore expects of `self.index` (add / search / reset / ntotal), over a local engine that holds
    this rank's rows plus the exchange of hip.sharded.ShardedFlatSearch."""

    def __init__(self, local, group=None, merge_fn: Optional[Callable] = None):
        import torch

        from ....hip.sharded import ShardedFlatSearch

        self.torch = torch
        self.local = local
        self.exchange = ShardedFlatSearch(self, group=group, merge_fn=merge_fn)
        self.world, self.rank = self.exchange.world, self.exchange.rank
        self.ntotal = 0                      # GLOBAL row count
        self._blocks: List[tuple] = []       # (first global id, local row count) per add, in order
        self._l2g = None                     # int64 tensor [local rows]: global id of each local row (increasing)
        self.dim = local.dim

    # -- rows ---------------------------------------------------------------------------------------------------
    def add_block(self, vectors_local, first_global: int, n_block: int) -> None:
        """This rank's slice of a block of `n_block` new rows whose global ids start at ntotal; the slice starts at
        global id `first_global`."""
        n_local = int(vectors_local.shape[0])
        if n_local:
            self.local.add(vectors_local)
            self._blocks.append((int(first_global), n_local))
            self._l2g = None
        self.ntotal += int(n_block)

    def reset(self) -> None:
        self.local.reset()
        self._blocks, self._l2g, self.ntotal = [], None, 0

    def remove_rows(self, global_rows) -> None:
        """Delete rows by GLOBAL id (every rank is handed the same list): this rank compacts the ones it holds
        (FlatIndexF16.remove_rows), and every block's first id drops by the holes before it — the global numbering stays
        "position among the surviving rows", as on one GPU.  No collective."""
        holes = np.unique(np.asarray(global_rows, dtype=np.int64).reshape(-1))
        if holes.size == 0:
            return
        if holes[0] < 0 or holes[-1] >= self.ntotal:
            raise IndexError(f"rows to remove must lie in [0, {self.ntotal})")
        local_holes, blocks, at = [], [], 0
        for g0, n in self._blocks:
            a, b = np.searchsorted(holes, g0), np.searchsorted(holes, g0 + n)
            local_holes.append(holes[a:b] - g0 + at)
            if n - (b - a) > 0:
                blocks.append((int(g0 - a), int(n - (b - a))))
            at += n
        mine = np.concatenate(local_holes) if local_holes else np.zeros(0, dtype=np.int64)
        if mine.size:
            self.local.remove_rows(mine)
        self._blocks, self._l2g = blocks, None
        self.ntotal -= int(holes.size)

    def _global_ids(self, device):
        t = self.torch
        if self._l2g is None or self._l2g.device != device:
            parts = [t.arange(g0, g0 + n, dtype=t.int64) for g0, n in self._blocks]
            self._l2g = (t.cat(parts) if parts else t.zeros(0, dtype=t.int64)).to(device)
        return self._l2g

    # -- search -------------------------------------------------------------------------------------------------
    def search_device(self, queries, k: int):
        """Local top-k with GLOBAL ids, padded to k with (-1, -inf) when this shard holds fewer than k rows (the form the
        exchange gathers).  Local rows are in increasing global-id order, so the local tie order (row ascending) is the
        global one (id ascending)."""
        t = self.torch
        n_local = self.local.ntotal
        if hasattr(self.local, "search_device"):
            nq = int(queries.shape[0]) if hasattr(queries, "shape") and len(queries.shape) == 2 else 1
            if n_local:
                ids, sc = self.local.search_device(queries, min(k, n_local))
            else:
                dev = self.local.device
                ids = t.empty((nq, 0), dtype=t.int64, device=dev)
                sc = t.empty((nq, 0), dtype=t.float32, device=dev)
        else:  # an engine with the numpy surface only (test doubles)
            q = np.asarray(queries, dtype=np.float32)
            if n_local:
                sc_h, ids_h = self.local.search(q, min(k, n_local))
                ids, sc = t.from_numpy(np.ascontiguousarray(ids_h)), t.from_numpy(np.ascontiguousarray(sc_h))
            else:
                ids, sc = t.empty((q.shape[0], 0), dtype=t.int64), t.empty((q.shape[0], 0), dtype=t.float32)
        if ids.shape[1]:
            l2g = self._global_ids(ids.device)
            base = int(getattr(self.local, "id_base", 0))
            ids = t.where(ids >= 0, l2g[(ids - base).clamp(min=0)], ids)
        if ids.shape[1] < k:
            pad = k - ids.shape[1]
            ids = t.cat([ids, t.full((ids.shape[0], pad), -1, dtype=t.int64, device=ids.device)], dim=1)
            sc = t.cat([sc, t.full((sc.shape[0], pad), float("-inf"), dtype=t.float32, device=sc.device)], dim=1)
        return ids.contiguous(), sc.contiguous()

    def search(self, queries, k: int):
        """(scores fp32 [nq][k], global ids int64 [nq][k]) as numpy, merged over all ranks — identical on every rank."""
        ids, sc = self.exchange.search_device(queries, k)
        return self._to_host(ids, sc)

    def _to_host(self, ids, sc):
        if hasattr(self.local, "to_host") and ids.is_cuda:
            return self.local.to_host(ids, sc)
        return sc.cpu().numpy(), ids.cpu().numpy()

    def search_async(self, queries, k: int, to_host: bool = False):
        """Enqueue this rank's scan now; `.host()` later runs the exchange (every rank collects its handles in the order
        it made them: the collectives line up) and returns the merged answer as numpy."""
        if not hasattr(self.local, "search_async") or self.local.ntotal < k:
            return _Ready(self.search(queries, k))         # (short shards take the padded synchronous path)
        return _PendingShard(self, self.local.search_async(queries, k), k)

    @property
    def max_norm(self):
        return self.local.max_norm


class _Ready:
    def __init__(self, answer):
        self.answer = answer

    def host(self):
        return self.answer


class _PendingShard:
    """A local scan in flight; host() = status check (+ rare repair) -> global ids -> exchange -> numpy."""

    def __init__(self, index: _ShardedIndex, handle, k: int):
        self.index, self.handle, self.k = index, handle, k

    def host(self):
        idx, t = self.index, self.index.torch
        ids, sc = self.handle.result()
        l2g = idx._global_ids(ids.device)
        base = int(getattr(idx.local, "id_base", 0))
        ids = t.where(ids >= 0, l2g[(ids - base).clamp(min=0)], ids)
        ids, sc = idx.exchange._exchange(ids.contiguous(), sc.contiguous(), self.k)
        return idx._to_host(ids, sc)


class HipShardedFlatVectorStore(HipFlatVectorStore):
    def __init__(self, embedding, *args: Any, group=None, merge_fn: Optional[Callable] = None, **kwargs: Any):
        kwargs["coalesce"] = False        # SPMD: see the module docstring
        super().__init__(embedding, *args, **kwargs)
        self._group, self._merge_fn = group, merge_fn

    def _make_engine(self, dim: int):
        return _ShardedIndex(super()._make_engine(dim), group=self._group, merge_fn=self._merge_fn)

    @property
    def shard(self):
        """(rank, world size, rows held by this rank, rows in total)."""
        if self.index is None:
            return (0, 1, 0, 0)
        return (self.index.rank, self.index.world, self.index.local.ntotal, self.index.ntotal)

    def add_texts(self, texts, metadatas=None, *, ids=None, **kwargs: Any) -> List[str]:
        import uuid

        from ....hip.sharded import shard_range

        texts = list(texts)
        if not texts:
            return []
        with self._guard.exclusive():
            return self._add_texts_sharded(texts, metadatas, ids)

    def _add_texts_sharded(self, texts, metadatas, ids) -> List[str]:
        import uuid

        from ....hip.sharded import shard_range

        if ids is None:
            # every rank must name the documents alike: ids are derived from the global row numbers, not drawn at random
            start = self.ntotal
            ids = [str(uuid.uuid5(uuid.NAMESPACE_OID, f"rarc-row-{start + i}")) for i in range(len(texts))]
        elif len(ids) != len(texts):
            raise ValueError("number of ids must match number of texts")
        if metadatas is None:
            metadatas = [{} for _ in texts]
        elif len(metadatas) != len(texts):
            raise ValueError("number of metadatas must match number of texts")
        if self.index is None:
            probe = self.embedding.embed_documents(texts[:1])
            self.index = self._make_engine(len(probe[0]))
        lo, hi = shard_range(len(texts), self.index.rank, self.index.world)
        mine = texts[lo:hi]
        if mine and hasattr(self.embedding, "embed_documents_device"):
            vectors = self.embedding.embed_documents_device(mine)
        elif mine:
            vectors = np.array(self.embedding.embed_documents(mine)).astype(np.float32)
        else:
            vectors = np.zeros((0, self.index.dim), np.float32)
        start = self.index.ntotal
        self.index.add_block(vectors, start + lo, len(texts))
        self._remember(start, texts, metadatas, ids)
        return list(ids)

    def max_marginal_relevance_search_by_vector(self, embedding, k: int = 4, fetch_k: int = 20,
                                                lambda_mult: float = 0.5, **kwargs: Any):
        if not kwargs.get("reembed", True):
            raise NotImplementedError("reembed=False gathers candidate rows from one GPU's HBM; the sharded store re-embeds "
                                      "the candidates' texts, as the reference does")
        return super().max_marginal_relevance_search_by_vector(embedding, k, fetch_k, lambda_mult, **kwargs)

    # -- persistence: one shard file per rank (hip_flat.save_local / load_local do the work) ----------------------------
    def _shard_layout(self):
        import torch.distributed as dist

        if self.index is not None:
            return self.index.rank, self.index.world
        if dist.is_available() and dist.is_initialized():
            return dist.get_rank(self._group), dist.get_world_size(self._group)
        return 0, 1

    def _barrier(self) -> None:
        import torch.distributed as dist

        if dist.is_available() and dist.is_initialized() and dist.get_world_size(self._group) > 1:
            dist.barrier(group=self._group)

    def _all_ranks_ok(self, ok: bool) -> bool:
        """All-reduce of the ranks' outcomes: a failing rank must not leave the others waiting in a barrier."""
        import torch
        import torch.distributed as dist

        if not (dist.is_available() and dist.is_initialized() and dist.get_world_size(self._group) > 1):
            return ok
        on_gpu = dist.get_backend(self._group) == "nccl"
        flag = torch.tensor([0 if ok else 1], dtype=torch.int32, device=torch.device("cuda", self.device) if on_gpu else "cpu")
        dist.all_reduce(flag, op=dist.ReduceOp.MAX, group=self._group)
        return int(flag.item()) == 0

    def _local_engine(self):
        return None if self.index is None else self.index.local

    def _local_blocks(self):
        return list(self.index._blocks)

    def _adopt_loaded(self, blocks, total: int) -> None:
        """load_local filled the local engine with this rank's rows: their global ids and the global row count."""
        idx = self.index
        if sum(c for _, c in blocks) != idx.local.ntotal:
            raise ValueError("the id map of the loaded shard does not cover the rows that were loaded")
        idx._blocks = [(int(g), int(c)) for g, c in blocks if c]
        idx._l2g = None
        idx.ntotal = int(total)
