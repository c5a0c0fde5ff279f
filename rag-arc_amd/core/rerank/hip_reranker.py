"""Yes/no-logit reranker (reference: core/rerank/Reranker_Qwen3.py:6-75).

The reference scores each (query, document) pair with a causal LM and keeps only two numbers per
pair: the last-position logits of the tokens "no" and "yes" (fp16).  Everything after that —
p_yes = exp(log_softmax([no, yes])[1]) through fp16 tensors, stable descending sort, optional
top-k — is done here by the rarc_rerank_order HIP kernel.  The LM forward itself is supplied as
`logit_fn(query, [contents]) -> (z_no, z_yes)` (fp16 arrays); no model weights ship with this repo,
so LM-forward parity is outside what can be pinned.
"""
from typing import Callable, Optional, Sequence, Tuple

from ..utils.data_model import Document
from .base import RerankerBase


class TableLogits:
    """`logit_fn` backed by a table of logits computed offline: (query, doc content) -> (z_no, z_yes)."""

    def __init__(self, queries, docs, z_no, z_yes):
        import numpy as np

        self.qi = {str(q): i for i, q in enumerate(queries)}
        self.di = {str(d): i for i, d in enumerate(docs)}
        self.z_no, self.z_yes = np.asarray(z_no, dtype=np.float16), np.asarray(z_yes, dtype=np.float16)
        if self.z_no.shape != (len(self.qi), len(self.di)) or self.z_yes.shape != self.z_no.shape:
            raise ValueError("z_no / z_yes must be [n_queries][n_docs]")

    @classmethod
    def from_npz(cls, path: str) -> "TableLogits":
        import numpy as np

        with np.load(path, allow_pickle=False) as d:
            return cls([str(x) for x in d["queries"]], [str(x) for x in d["docs"]], d["z_no"], d["z_yes"])

    def __call__(self, query: str, contents: Sequence[str]):
        q = self.qi[query]
        cols = [self.di[c] for c in contents]
        return self.z_no[q, cols], self.z_yes[q, cols]


class HipLogitReranker(RerankerBase):
    def __init__(self, logit_fn: Callable[[str, Sequence[str]], Tuple], instruction: Optional[str] = None,
                 device: int = 0):
        super().__init__()
        self.logit_fn = logit_fn
        self.instruction = instruction or "Given the user query, retrieval the relevant passages"
        self.device = device

    def format_instruction(self, instruction, query, doc):
        return f"<Instruct>: {instruction or self.instruction}\n<Query>: {query}\n<Document>: {doc}"

    def score_order(self, z_no, z_yes):
        """fp16 logits [nq][n] (device or host) -> (p_yes fp16 [nq][n], permutation int32 [nq][n])."""
        import torch

        from ...hip import binding as B

        lib = B.load_library()
        dev = torch.device("cuda", self.device)
        zn = torch.as_tensor(z_no, dtype=torch.float16).to(dev).contiguous()
        zy = torch.as_tensor(z_yes, dtype=torch.float16).to(dev).contiguous()
        if zn.ndim == 1:
            zn, zy = zn[None, :], zy[None, :]
        nq, n = zn.shape
        scores = torch.empty((nq, n), dtype=torch.float16, device=dev)
        perm = torch.empty((nq, n), dtype=torch.int32, device=dev)
        B.check(lib.rarc_rerank_order(zn.data_ptr(), zy.data_ptr(), nq, n, scores.data_ptr(), perm.data_ptr(),
                                      torch.cuda.current_stream(dev).cuda_stream), "rarc_rerank_order")
        return scores, perm

    def rerank(self, query: str, documents: list[Document], k: int = None, batch_size: int = 8, **kwargs):
        if not documents:
            return []
        z_no, z_yes = [], []
        for s in range(0, len(documents), batch_size):  # same batching as the reference's LM calls
            a, b = self.logit_fn(query, [d.content for d in documents[s:s + batch_size]])
            z_no.extend(list(a))
            z_yes.extend(list(b))
        _, perm = self.score_order(z_no, z_yes)
        return self.apply_order(documents, perm[0].tolist(), k)
