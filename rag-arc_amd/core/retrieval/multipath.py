"""Fan-out over several retrievers + fusion (reference: core/retrieval/mutipath.py:8-127).

Retrievers run one after another with k = top_k_per_retriever (every other kwarg, `top_k`
included, is handed down as is); a retriever that raises is reported with print() and contributes
an empty list; if every list is empty the answer is []; otherwise fuse(..., top_k) and unwrap.
Like the reference, __init__ does not call BaseRetriever.__init__ (no search_kwargs attribute).
"""
from typing import Any, List, Optional

from ..utils.data_model import Document
from ..utils.fusion import FusionMethod, RetrievalResult
from .base import BaseRetriever


class MultiPathRetriever(BaseRetriever):
    def __init__(self, retrievers: List[BaseRetriever], fusion_method: Optional[FusionMethod] = None,
                 top_k_per_retriever: int = 50):
        self.retrievers = retrievers
        if fusion_method is None:
            from ..utils.fusion import RRFusion

            fusion_method = RRFusion()
        self.fusion_method = fusion_method
        self.top_k_per_retriever = top_k_per_retriever

    def _get_relevant_documents(self, query: str, **kwargs: Any) -> List[Document]:
        top_k = kwargs.get("top_k", 10)
        gathered: List[List[RetrievalResult]] = []
        for retriever in self.retrievers:
            try:
                docs = retriever.invoke(query, **{**kwargs, "k": self.top_k_per_retriever})
                gathered.append([RetrievalResult(document=d, score=getattr(d, "score", 1.0), rank=i + 1)
                                 for i, d in enumerate(docs)])
            except Exception as exc:  # noqa: BLE001 - a failing path must not sink the others
                print(f"retriever {type(retriever).__name__} failed: {exc}")
                gathered.append([])
        if not gathered or all(len(one) == 0 for one in gathered):
            return []
        return [r.document for r in self.fusion_method.fuse(gathered, top_k)]

    def batch_invoke(self, inputs: List[str], **kwargs: Any) -> List[List[Document]]:
        """invoke() for a list of queries: every retriever answers the whole list (in one pass where it can batch), then
        ALL queries are fused in one kernel launch (`fuse_many`) when the fusion method has one.  Element i equals
        invoke(inputs[i], **kwargs) — same kwargs hand-down, same swallow-and-report of a failing retriever."""
        inputs = list(inputs)
        top_k = kwargs.get("top_k", 10)
        fuse_docs_many = getattr(self.fusion_method, "fuse_docs_many", None)
        if fuse_docs_many is not None:
            # the fusion method takes plain Document lists: no RetrievalResult per hit on the way in or out
            answers: List[List[List[Document]]] = []
            for retriever in self.retrievers:
                answers.append(self._batch_of(retriever, inputs, kwargs))
            gathered_docs = [[lists[qi] for lists in answers] for qi in range(len(inputs))]
            live = [qi for qi, g in enumerate(gathered_docs) if g and not all(len(one) == 0 for one in g)]
            out_docs: List[List[Document]] = [[] for _ in inputs]
            if live:
                for qi, docs in zip(live, fuse_docs_many([gathered_docs[qi] for qi in live], top_k)):
                    out_docs[qi] = docs
            return out_docs
        per_retriever: List[List[List[RetrievalResult]]] = []
        for retriever in self.retrievers:
            try:
                lists = retriever.batch_invoke(inputs, **{**kwargs, "k": self.top_k_per_retriever})
                per_retriever.append([[RetrievalResult(document=d, score=getattr(d, "score", 1.0), rank=i + 1)
                                       for i, d in enumerate(docs)] for docs in lists])
            except Exception as exc:  # noqa: BLE001
                # one bad query must cost what it costs under invoke(): that query's list from this retriever, nothing else
                print(f"retriever {type(retriever).__name__} failed on the batch ({exc}): answering it query by query")
                lists = []
                for q in inputs:
                    try:
                        docs = retriever.invoke(q, **{**kwargs, "k": self.top_k_per_retriever})
                        lists.append([RetrievalResult(document=d, score=getattr(d, "score", 1.0), rank=i + 1)
                                      for i, d in enumerate(docs)])
                    except Exception as exc_q:  # noqa: BLE001
                        print(f"retriever {type(retriever).__name__} failed: {exc_q}")
                        lists.append([])
                per_retriever.append(lists)
        gathered = [[lists[qi] for lists in per_retriever] for qi in range(len(inputs))]
        live = [qi for qi, g in enumerate(gathered) if g and not all(len(one) == 0 for one in g)]
        out: List[List[Document]] = [[] for _ in inputs]
        fuse_many = getattr(self.fusion_method, "fuse_many", None)
        if fuse_many is not None and live:
            for qi, fused in zip(live, fuse_many([gathered[qi] for qi in live], top_k)):
                out[qi] = [r.document for r in fused]
        else:
            for qi in live:
                out[qi] = [r.document for r in self.fusion_method.fuse(gathered[qi], top_k)]
        return out

    def _batch_of(self, retriever: BaseRetriever, inputs: List[str], kwargs) -> List[List[Document]]:
        """One retriever's answers to the whole list as plain lists (same hand-down and failure handling as above)."""
        try:
            return [list(docs) for docs in retriever.batch_invoke(inputs, **{**kwargs, "k": self.top_k_per_retriever})]
        except Exception as exc:  # noqa: BLE001
            print(f"retriever {type(retriever).__name__} failed on the batch ({exc}): answering it query by query")
            lists: List[List[Document]] = []
            for q in inputs:
                try:
                    lists.append(list(retriever.invoke(q, **{**kwargs, "k": self.top_k_per_retriever})))
                except Exception as exc_q:  # noqa: BLE001
                    print(f"retriever {type(retriever).__name__} failed: {exc_q}")
                    lists.append([])
            return lists

    def add_retriever(self, retriever: BaseRetriever) -> None:
        self.retrievers.append(retriever)

    def remove_retriever(self, name: str) -> None:
        for i, retriever in enumerate(self.retrievers):
            if type(retriever).__name__ == name:
                self.retrievers.pop(i)
                break

    def set_fusion_method(self, fusion_method: FusionMethod) -> None:
        self.fusion_method = fusion_method
