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

    def add_retriever(self, retriever: BaseRetriever) -> None:
        self.retrievers.append(retriever)

    def remove_retriever(self, name: str) -> None:
        for i, retriever in enumerate(self.retrievers):
            if type(retriever).__name__ == name:
                self.retrievers.pop(i)
                break

    def set_fusion_method(self, fusion_method: FusionMethod) -> None:
        self.fusion_method = fusion_method
