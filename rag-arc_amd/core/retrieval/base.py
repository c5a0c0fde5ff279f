"""Retriever contract (reference: core/retrieval/base.py:8-100).

query string in, ordered Documents out.  `invoke` / `ainvoke` are the entry points; subclasses
implement `_get_relevant_documents`; unless they also provide a native coroutine, `ainvoke` runs
the sync method on a fresh ThreadPoolExecutor (so a backend may be entered from any thread —
the HIP engine takes a lock and selects its device per call for that reason).
"""
import asyncio
from abc import ABC, abstractmethod
from concurrent.futures import ThreadPoolExecutor
from functools import partial
from typing import Any, List

from ..utils.data_model import Document


class BaseRetriever(ABC):
    def __init__(self, **kwargs):
        self.search_kwargs = kwargs.get("search_kwargs", {})
        self.tags = kwargs.get("tags")
        self.metadata = kwargs.get("metadata")

    # -- entry points ------------------------------------------------------------------------
    def invoke(self, input: str, **kwargs: Any) -> List[Document]:
        return self._get_relevant_documents(input, **kwargs)

    async def ainvoke(self, input: str, **kwargs: Any) -> List[Document]:
        return await self._aget_relevant_documents(input, **kwargs)

    def batch_invoke(self, inputs: List[str], **kwargs: Any) -> List[List[Document]]:
        """Extension over the reference (which has no batch entry point): element i equals invoke(inputs[i], **kwargs).
        Retrievers over a batched backend override this to answer the whole list in one pass."""
        return [self.invoke(q, **kwargs) for q in inputs]

    # -- to implement ------------------------------------------------------------------------
    @abstractmethod
    def _get_relevant_documents(self, query: str, **kwargs: Any) -> List[Document]:
        ...

    async def _aget_relevant_documents(self, query: str, **kwargs: Any) -> List[Document]:
        work = partial(self._get_relevant_documents, query, **kwargs)
        with ThreadPoolExecutor() as pool:
            return await asyncio.get_event_loop().run_in_executor(pool, work)

    def get_name(self) -> str:
        return type(self).__name__
