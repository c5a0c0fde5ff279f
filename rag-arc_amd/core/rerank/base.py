"""Reranker contract (reference: core/rerank/base.py:5-27): `rerank(query, documents, **kw)` returns
the same Document objects in a new order, best first.  Instantiating the base directly only warns,
as in the reference."""
import warnings
from abc import ABC, abstractmethod
from typing import List, Sequence

from ..utils.data_model import Document


class RerankerBase(ABC):
    def __init__(self):
        if type(self) is RerankerBase:
            warnings.warn("RerankerBase is abstract: subclass it and implement rerank()", UserWarning)

    @abstractmethod
    def rerank(self, query: str, documents: List[Document], **kwargs) -> List[Document]:
        raise NotImplementedError("subclasses implement rerank()")

    @staticmethod
    def apply_order(documents: Sequence[Document], order: Sequence[int], k: int = None) -> List[Document]:
        """Reorder `documents` by a permutation (as produced by a scoring backend), optionally keep k."""
        picked = [documents[i] for i in order]
        return picked if k is None else picked[:k]
