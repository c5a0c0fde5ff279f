"""Reranker contract (reference: core/rerank/base.py:5-27)."""
import warnings
from abc import ABC, abstractmethod

from ..utils.data_model import Document


class RerankerBase(ABC):
    def __init__(self):
        if type(self) is RerankerBase:
            warnings.warn("RerankerBase is abstract; subclass it and implement rerank()", UserWarning)

    @abstractmethod
    def rerank(self, query: str, documents: list[Document], **kwargs) -> list[Document]:
        """Return the same Document objects, reordered (best first)."""
        raise NotImplementedError
