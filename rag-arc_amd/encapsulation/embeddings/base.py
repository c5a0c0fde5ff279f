"""Embedding provider contract (reference: core/file_management/embeddings/base.py:7-61; the
reference's encapsulation/embeddings/ is the empty slot providers are meant to move into)."""
import asyncio
from abc import ABC, abstractmethod
from concurrent.futures import ThreadPoolExecutor
from typing import List


class Embeddings(ABC):
    def __init__(self, **kwargs):
        pass

    @abstractmethod
    def embed_documents(self, texts: List[str]) -> List[List[float]]:
        """One embedding (list of python floats) per text."""

    @abstractmethod
    def embed_query(self, text: str) -> List[float]:
        """Embedding of a single query text."""

    async def aembed_documents(self, texts: List[str]) -> List[List[float]]:
        return await asyncio.get_event_loop().run_in_executor(ThreadPoolExecutor(), self.embed_documents, texts)

    async def aembed_query(self, text: str) -> List[float]:
        return await asyncio.get_event_loop().run_in_executor(ThreadPoolExecutor(), self.embed_query, text)
