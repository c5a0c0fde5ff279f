"""Result fusion (reference: core/utils/Fusion.py:9-76).

RRFusion.fuse semantics kept bit for bit: ranks are re-numbered 1.. inside every input list (the
inputs are mutated, as in the reference); score(content) = sum over occurrences — lists in order,
positions in order — of 1.0 / (k + rank) in fp64; the Document kept for a content is the LAST one
seen; order = score descending, ties in first-insertion order; output ranks 1...

The arithmetic runs in the rarc_rrf_fuse HIP kernel (rag-arc_amd/csrc/fuse.hip): contents are
mapped to integer keys in first-seen order on the host, the kernel returns keys + fp64 scores.
`fuse_ids` is the batched form used when retrieval results are already ids on the device.
"""
from abc import ABC, abstractmethod
from dataclasses import dataclass
from typing import List

from .data_model import Document


@dataclass
class RetrievalResult:
    document: Document
    score: float
    rank: int = 0


class FusionMethod(ABC):
    @abstractmethod
    def fuse(self, results: List[List[RetrievalResult]], top_k: int) -> List[RetrievalResult]:
        """Merge the per-retriever result lists into one ranked list of at most top_k entries."""


class HipRRFusion(FusionMethod):
    """Reciprocal-rank fusion on the MI355X."""

    def __init__(self, k: float = 60.0, device: int = 0):
        self.k = k
        self.device = device

    # -- batched, id-keyed ------------------------------------------------------------------
    def fuse_ids(self, keys, lengths, top_k: int):
        """keys: int64 [nq][n_lists][max_len] device tensor, lengths: int32 [nq][n_lists].
        Returns (fused keys int64 [nq][top_k], scores fp64 [nq][top_k], counts int32 [nq])."""
        import torch

        from ...hip import binding as B

        lib = B.load_library()
        nq, n_lists, max_len = keys.shape
        out_k = torch.full((nq, max(top_k, 1)), -1, dtype=torch.int64, device=keys.device)
        out_s = torch.zeros((nq, max(top_k, 1)), dtype=torch.float64, device=keys.device)
        out_n = torch.zeros(nq, dtype=torch.int32, device=keys.device)
        B.check(lib.rarc_rrf_fuse(keys.contiguous().data_ptr(), lengths.contiguous().data_ptr(), nq, n_lists,
                                  max_len, float(self.k), int(top_k), out_k.data_ptr(), out_s.data_ptr(),
                                  out_n.data_ptr(), torch.cuda.current_stream(keys.device).cuda_stream),
                "rarc_rrf_fuse")
        return out_k[:, :top_k], out_s[:, :top_k], out_n

    # -- the FusionMethod contract --------------------------------------------------------------
    def fuse(self, results: List[List[RetrievalResult]], top_k: int) -> List[RetrievalResult]:
        import torch

        for one in results:
            for pos, item in enumerate(one):
                item.rank = pos + 1
        key_of, doc_of = {}, {}
        max_len = max((len(one) for one in results), default=0)
        if max_len == 0 or top_k <= 0:
            return []
        table = [[0] * max_len for _ in results]
        for li, one in enumerate(results):
            for pi, item in enumerate(one):
                content = item.document.content
                key = key_of.setdefault(content, len(key_of))
                doc_of[key] = item.document  # last occurrence wins
                table[li][pi] = key
        dev = torch.device("cuda", self.device)
        keys = torch.tensor([table], dtype=torch.int64, device=dev)
        lens = torch.tensor([[len(one) for one in results]], dtype=torch.int32, device=dev)
        fk, fs, fn = self.fuse_ids(keys, lens, min(top_k, len(key_of)))
        n = int(fn[0].item())
        fk, fs = fk[0, :n].tolist(), fs[0, :n].tolist()
        return [RetrievalResult(document=doc_of[key], score=score, rank=i + 1)
                for i, (key, score) in enumerate(zip(fk, fs))]


    def fuse_many(self, batch: List[List[List[RetrievalResult]]], top_k: int) -> List[List[RetrievalResult]]:
        """fuse() for many queries in ONE kernel launch: batch[q] is the list of per-retriever result lists of query q.
        Element q equals fuse(batch[q], top_k) (same side effects on the input ranks)."""
        import torch

        if not batch:
            return []
        n_lists = max(len(one) for one in batch)
        max_len = max((len(lst) for one in batch for lst in one), default=0)
        if max_len == 0 or top_k <= 0 or n_lists == 0:
            return [[] for _ in batch]
        table = [[[0] * max_len for _ in range(n_lists)] for _ in batch]
        lens = [[0] * n_lists for _ in batch]
        docs: List[dict] = []
        for qi, results in enumerate(batch):
            key_of, doc_of = {}, {}
            for li, one in enumerate(results):
                lens[qi][li] = len(one)
                for pi, item in enumerate(one):
                    item.rank = pi + 1
                    key = key_of.setdefault(item.document.content, len(key_of))
                    doc_of[key] = item.document  # last occurrence wins
                    table[qi][li][pi] = key
            docs.append(doc_of)
        dev = torch.device("cuda", self.device)
        fk, fs, fn = self.fuse_ids(torch.tensor(table, dtype=torch.int64, device=dev),
                                   torch.tensor(lens, dtype=torch.int32, device=dev), min(top_k, n_lists * max_len))
        fk, fs, fn = fk.cpu().tolist(), fs.cpu().tolist(), fn.cpu().tolist()
        return [[RetrievalResult(document=docs[qi][key], score=score, rank=i + 1)
                 for i, (key, score) in enumerate(zip(fk[qi][: fn[qi]], fs[qi][: fn[qi]]))] for qi in range(len(batch))]


    def fuse_docs_many(self, batch: List[List[List[Document]]], top_k: int) -> List[List[Document]]:
        """The documents of fuse_many(...) without the RetrievalResult wrappers either side: batch[q][l] is retriever l's
        answer to query q as a plain list of Documents (position = rank - 1), the result is [r.document for r in
        fuse(...)] per query.  Content keys and the last-one-wins Document table are built in one native pass
        (csrc/hostmap.c: rrf_tables), the sums and the order in rarc_rrf_fuse, the picks in pick_docs — the per-item python
        work of fuse_many (51,200 wrappers + dict operations for 256 queries x 2 lists x 100) is what a batch cost."""
        import numpy as np
        import torch

        from ...hip import hostmap

        if not batch:
            return []
        n_lists = max(len(one) for one in batch)
        max_len = max((len(lst) for one in batch for lst in one), default=0)
        if max_len == 0 or top_k <= 0 or n_lists == 0:
            return [[] for _ in batch]
        H = hostmap.load()
        keys_b, lens_b, docs = H.rrf_tables(batch, n_lists, max_len)
        dev = torch.device("cuda", self.device)
        nq = len(batch)
        keys = torch.from_numpy(np.frombuffer(keys_b, dtype=np.int64).reshape(nq, n_lists, max_len).copy()).to(dev)
        lens = torch.from_numpy(np.frombuffer(lens_b, dtype=np.int32).reshape(nq, n_lists).copy()).to(dev)
        width = min(top_k, n_lists * max_len)
        fk, _, fn = self.fuse_ids(keys, lens, width)
        return H.pick_docs(docs, fk.contiguous().cpu().numpy(), fn.cpu().numpy(), width)


# the name the reference exports
RRFusion = HipRRFusion
