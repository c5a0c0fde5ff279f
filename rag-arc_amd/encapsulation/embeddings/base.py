"""Embedding provider contract (reference: core/file_management/embeddings/base.py:7-61; the
reference's encapsulation/embeddings/ is the empty slot providers are meant to move into).

Two abstract methods; the async twins run them on a throw-away thread pool exactly like the
reference, so providers must tolerate being entered from arbitrary threads."""
import asyncio
from abc import ABC, abstractmethod
from concurrent.futures import ThreadPoolExecutor
from typing import List


def _off_thread(fn, *args):
    return asyncio.get_event_loop().run_in_executor(ThreadPoolExecutor(), fn, *args)


class Embeddings(ABC):
    def __init__(self, **kwargs):
        pass

    @abstractmethod
    def embed_documents(self, texts: List[str]) -> List[List[float]]:
        """texts -> one list of python floats per text."""

    @abstractmethod
    def embed_query(self, text: str) -> List[float]:
        """a single query text -> list of python floats."""

    async def aembed_documents(self, texts: List[str]) -> List[List[float]]:
        return await _off_thread(self.embed_documents, texts)

    async def aembed_query(self, text: str) -> List[float]:
        return await _off_thread(self.embed_query, text)
