"""Row-sharded flat search: one process per GPU, local top-k, one all-gather, merge.

The reference has no distributed code at all; this is the MI355X-native way to
scale `IndexFlatIP.search` (encapsulation/database/vector_db/VectorStore_Faiss.py:263) past one
GPU: shard g of G holds rows [g*ceil(N/G), (g+1)*ceil(N/G)), queries are replicated, every rank
returns (global id, canonical score)[nq][k] for its shard, ONE RCCL all-gather of nq*k*12 bytes per
rank moves them over xGMI (latency-bound: 307 KB per rank at nq=256, k=100), and a tiny merge
kernel keeps the k best by (score desc, id asc).  Canonical scores do not depend on the sharding,
so the merged result is bit-identical for every G.
"""
from __future__ import annotations

from typing import Callable, Optional, Tuple


def shard_range(n_total: int, rank: int, world: int) -> Tuple[int, int]:
    """Rows [lo, hi) owned by `rank` (contiguous blocks of ceil(n_total / world) rows)."""
    if world < 1 or not (0 <= rank < world):
        raise ValueError("bad rank/world")
    per = -(-n_total // world)
    lo = min(n_total, rank * per)
    return lo, min(n_total, lo + per)


def split_range(n_items: int, rank: int, world: int) -> Tuple[int, int]:
    """Items [lo, hi) of a replicated batch that `rank` computes when the batch is split evenly (the online
    query encoder of BASELINE config 5: each rank embeds its slice, one all-gather rebuilds the batch)."""
    return shard_range(n_items, rank, world)


def pack_results(torch, ids, scores):
    """(ids int64 [nq][k], scores fp32 [nq][k]) -> int32 [nq][k][3] (one collective instead of two)."""
    out = torch.empty(ids.shape + (3,), dtype=torch.int32, device=ids.device)
    out[..., :2] = ids.contiguous().view(torch.int32).view(ids.shape + (2,))
    out[..., 2] = scores.contiguous().view(torch.int32)
    return out


def unpack_results(torch, packed):
    ids = packed[..., :2].contiguous().view(torch.int64).view(packed.shape[:-1])
    scores = packed[..., 2].contiguous().view(torch.float32)
    return ids, scores


class ShardedFlatSearch:
    """Wraps a local index (anything with search_device(q, k) -> (ids, scores) device tensors whose
    ids are already global) and a torch.distributed process group."""

    def __init__(self, local_index, group=None, merge_fn: Optional[Callable] = None, force_collective: bool = False):
        import torch
        import torch.distributed as dist

        self.torch, self.dist = torch, dist
        self.local = local_index
        self.group = group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0
        self.merge_fn = merge_fn or self._hip_merge
        self.force_collective = force_collective  # run gather + merge even with a single rank (testing)
        self.exchange_events = None  # measure_exchange(True): [(start, stop)] event pairs around pack/gather/merge

    def measure_exchange(self, on: bool = True) -> None:
        """Bracket every exchange (pack kernel, all-gather, merge kernel) with a pair of events on the compute
        stream; exchange_ms() returns the summed time.  Measurement hook for bench.py."""
        self.exchange_events = [] if on else None

    def exchange_ms(self) -> float:
        ev = self.exchange_events or []
        if ev:
            ev[-1][1].synchronize()
        return float(sum(a.elapsed_time(b) for a, b in ev))

    def _hip_merge(self, ids, scores, k):
        """ids/scores: [G][nq][k] device tensors -> [nq][k] via rarc_topk_merge."""
        from . import binding as B

        t = self.torch
        lib = B.load_library()
        G, nq, kk = ids.shape
        out_i = t.empty((nq, k), dtype=t.int64, device=ids.device)
        out_s = t.empty((nq, k), dtype=t.float32, device=ids.device)
        B.check(lib.rarc_topk_merge(ids.contiguous().data_ptr(), scores.contiguous().data_ptr(), G, nq, kk,
                                    out_i.data_ptr(), out_s.data_ptr(),
                                    t.cuda.current_stream(ids.device).cuda_stream), "rarc_topk_merge")
        return out_i, out_s

    def search_async(self, queries, k: int):
        """Pipelined form: enqueue the local scan now; `finish(handle)` does the status check, the
        all-gather and the merge.  Lets batch i+1 scan while batch i is collected."""
        return self.local.search_async(queries, k)

    def finish(self, handle, k: int):
        ids, scores = handle.result()
        return self._exchange(ids, scores, k)

    def search_device(self, queries, k: int):
        ids, scores = self.local.search_device(queries, k)
        return self._exchange(ids, scores, k)

    def _exchange(self, ids, scores, k: int):
        t = self.torch
        if self.world == 1 and not self.force_collective:
            return ids, scores
        if self.merge_fn == self._hip_merge and ids.is_cuda:
            # device path: one pack kernel, ONE collective, one merge kernel that reads the packed lists
            from . import binding as B

            if self.exchange_events is not None:
                ev = (t.cuda.Event(enable_timing=True), t.cuda.Event(enable_timing=True))
                ev[0].record()
                self.exchange_events.append(ev)

            lib = B.load_library()
            nq, kk = ids.shape
            st = t.cuda.current_stream(ids.device).cuda_stream
            gathered = t.empty((self.world, nq, kk, 3), dtype=t.int32, device=ids.device)
            mine = t.empty((nq, kk, 3), dtype=t.int32, device=ids.device)
            B.check(lib.rarc_pack_results(ids.contiguous().data_ptr(), scores.contiguous().data_ptr(), nq, kk,
                                          mine.data_ptr(), st), "rarc_pack_results")
            self.dist.all_gather_into_tensor(gathered.view(self.world * nq, kk, 3), mine, group=self.group)
            if k != kk:
                raise ValueError("merge width must equal the per-shard list width")
            out_i = t.empty((nq, k), dtype=t.int64, device=ids.device)
            out_s = t.empty((nq, k), dtype=t.float32, device=ids.device)
            B.check(lib.rarc_topk_merge_packed(gathered.data_ptr(), self.world, nq, kk, out_i.data_ptr(),
                                               out_s.data_ptr(), st), "rarc_topk_merge_packed")
            if self.exchange_events is not None:
                self.exchange_events[-1][1].record()
            return out_i, out_s
        mine = pack_results(t, ids, scores)
        # concatenated layout ([world*nq][k][3]) is the form both RCCL and gloo accept
        gathered = t.empty((self.world * mine.shape[0],) + tuple(mine.shape[1:]), dtype=mine.dtype,
                           device=mine.device)
        self.dist.all_gather_into_tensor(gathered, mine.contiguous(), group=self.group)
        g_ids, g_scores = unpack_results(t, gathered.view((self.world,) + tuple(mine.shape)))
        return self.merge_fn(g_ids, g_scores, k)
