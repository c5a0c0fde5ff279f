"""Retriever contract (reference: core/retrieval/base.py:8-100).

A retriever maps a query string to an ordered list of Documents.  Callers use invoke / ainvoke;
implementations provide _get_relevant_documents (and optionally a native async twin; the default
one runs the sync method on a throw-away thread pool, so backends may be entered from any thread).
"""
import asyncio
from abc import ABC, abstractmethod
from concurrent.futures import ThreadPoolExecutor
from typing import Any, List

from ..utils.data_model import Document


class BaseRetriever(ABC):
    def __init__(self, **kwargs):
        self.search_kwargs = kwargs.get("search_kwargs", {})
        self.tags = kwargs.get("tags")
        self.metadata = kwargs.get("metadata")

    def invoke(self, input: str, **kwargs: Any) -> List[Document]:
        return self._get_relevant_documents(input, **kwargs)

    async def ainvoke(self, input: str, **kwargs: Any) -> List[Document]:
        return await self._aget_relevant_documents(input, **kwargs)

    @abstractmethod
    def _get_relevant_documents(self, query: str, **kwargs: Any) -> List[Document]:
        ...

    async def _aget_relevant_documents(self, query: str, **kwargs: Any) -> List[Document]:
        loop = asyncio.get_event_loop()
        with ThreadPoolExecutor() as pool:
            return await loop.run_in_executor(pool, lambda: self._get_relevant_documents(query, **kwargs))

    def get_name(self) -> str:
        return type(self).__name__
