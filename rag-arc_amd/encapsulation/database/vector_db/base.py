"""Vector store contract (reference: encapsulation/database/vector_db/VectorStoreBase.py:45-627).

The whole public surface is mirrored: add / delete / get, the `search` / `asearch` dispatchers, every search entry
point with its async twin, from_texts / from_documents and their async twins, as_retriever.  Relevance scores keep the
reference's arithmetic exactly, including its quirk: for cosine stores relevance = 1.0 - score although `score` already
is a similarity (VectorStoreBase.py:263-266), so a score_threshold keeps the LEAST similar documents.

One deliberate difference, pinned by tests/golden/surface.json: the reference's async twins hand their keyword arguments
to `run_in_executor`, which takes none, and pass `add_texts`' keyword-only `ids` positionally — `asimilarity_search(q,
k, top_k=9)`, `adelete(ids, soft=True)`, `aadd_texts(..., ids=...)`, `afrom_texts(..., ids=...)` all die with TypeError
there (VectorStoreBase.py:125-137, :158-163, :250-256, :617-620).  Here a twin returns what its sync method returns.
"""
import asyncio
import logging
import math
import warnings
from abc import ABC, abstractmethod
from concurrent.futures import ThreadPoolExecutor
from functools import partial
from typing import Any, Callable, Iterable, List, Optional, Tuple

from ....core.utils.data_model import Document

logger = logging.getLogger(__name__)


async def _in_pool(fn, *args, **kwargs):
    return await asyncio.get_event_loop().run_in_executor(ThreadPoolExecutor(), partial(fn, *args, **kwargs))


class VectorStore(ABC):
    def __init__(self, **kwargs: Any):
        pass

    # ------------------------------------------------------------------ ingestion
    def add_texts(self, texts: Iterable[str], metadatas: Optional[List[dict]] = None, *,
                  ids: Optional[List[str]] = None, **kwargs: Any) -> List[str]:
        raise NotImplementedError

    def add_documents(self, documents: List[Document], **kwargs: Any) -> List[str]:
        if "ids" not in kwargs:
            ids = [d.id for d in documents]
            if any(ids):
                kwargs["ids"] = ids
        return self.add_texts([d.content for d in documents], [d.metadata for d in documents], **kwargs)

    async def aadd_texts(self, texts, metadatas=None, *, ids=None, **kwargs):
        return await _in_pool(self.add_texts, texts, metadatas, ids=ids, **kwargs)

    async def aadd_documents(self, documents: List[Document], **kwargs: Any) -> List[str]:
        return await _in_pool(self.add_documents, documents, **kwargs)

    def delete(self, ids: Optional[List[str]] = None, **kwargs: Any) -> Optional[bool]:
        raise NotImplementedError

    async def adelete(self, ids: Optional[List[str]] = None, **kwargs: Any) -> Optional[bool]:
        return await _in_pool(self.delete, ids, **kwargs)

    def get_by_ids(self, ids: List[str]) -> List[Document]:
        raise NotImplementedError(f"{self.__class__.__name__} does not support get_by_ids")

    async def aget_by_ids(self, ids: List[str]) -> List[Document]:
        return await _in_pool(self.get_by_ids, ids)

    # ------------------------------------------------------------------ dispatch by search type (VectorStoreBase.py:184-232)
    _SEARCH_TYPES = "'similarity', 'similarity_score_threshold' or 'mmr'"

    def search(self, query: str, search_type: str, **kwargs: Any) -> List[Document]:
        if search_type == "similarity":
            return self.similarity_search(query, **kwargs)
        if search_type == "similarity_score_threshold":
            return [doc for doc, _ in self.similarity_search_with_relevance_scores(query, **kwargs)]
        if search_type == "mmr":
            return self.max_marginal_relevance_search(query, **kwargs)
        raise ValueError(f"search_type {search_type} is not allowed; expected {self._SEARCH_TYPES}")

    async def asearch(self, query: str, search_type: str, **kwargs: Any) -> List[Document]:
        if search_type == "similarity":
            return await self.asimilarity_search(query, **kwargs)
        if search_type == "similarity_score_threshold":
            return [doc for doc, _ in await self.asimilarity_search_with_relevance_scores(query, **kwargs)]
        if search_type == "mmr":
            return await self.amax_marginal_relevance_search(query, **kwargs)
        raise ValueError(f"search_type {search_type} is not allowed; expected {self._SEARCH_TYPES}")

    # ------------------------------------------------------------------ search
    @abstractmethod
    def similarity_search(self, query: str, k: int = 4, **kwargs: Any) -> List[Document]:
        ...

    async def asimilarity_search(self, query: str, k: int = 4, **kwargs: Any) -> List[Document]:
        return await _in_pool(self.similarity_search, query, k, **kwargs)

    def similarity_search_with_score(self, *args: Any, **kwargs: Any) -> List[Tuple[Document, float]]:
        raise NotImplementedError

    async def asimilarity_search_with_score(self, *args: Any, **kwargs: Any):
        return await _in_pool(self.similarity_search_with_score, *args, **kwargs)

    def similarity_search_by_vector(self, embedding: List[float], k: int = 4, **kwargs: Any) -> List[Document]:
        raise NotImplementedError

    async def asimilarity_search_by_vector(self, embedding: List[float], k: int = 4, **kwargs: Any) -> List[Document]:
        return await _in_pool(self.similarity_search_by_vector, embedding, k, **kwargs)

    def max_marginal_relevance_search(self, query: str, k: int = 4, fetch_k: int = 20, lambda_mult: float = 0.5,
                                      **kwargs: Any) -> List[Document]:
        raise NotImplementedError

    def max_marginal_relevance_search_by_vector(self, embedding: List[float], k: int = 4, fetch_k: int = 20,
                                                lambda_mult: float = 0.5, **kwargs: Any) -> List[Document]:
        raise NotImplementedError

    async def amax_marginal_relevance_search_by_vector(self, embedding: List[float], k: int = 4, fetch_k: int = 20,
                                                       lambda_mult: float = 0.5, **kwargs: Any) -> List[Document]:
        return await _in_pool(self.max_marginal_relevance_search_by_vector, embedding, k, fetch_k, lambda_mult, **kwargs)

    async def amax_marginal_relevance_search(self, query: str, k: int = 4, fetch_k: int = 20,
                                             lambda_mult: float = 0.5, **kwargs: Any) -> List[Document]:
        return await _in_pool(self.max_marginal_relevance_search, query, k, fetch_k, lambda_mult, **kwargs)

    # ------------------------------------------------------------------ relevance scores
    @staticmethod
    def _euclidean_relevance_score_fn(distance: float) -> float:
        return 1.0 - distance / math.sqrt(2)

    @staticmethod
    def _cosine_relevance_score_fn(distance: float) -> float:
        return 1.0 - distance

    @staticmethod
    def _max_inner_product_relevance_score_fn(distance: float) -> float:
        if distance > 0:
            return 1.0 - distance
        return -1.0 * distance

    def _select_relevance_score_fn(self) -> Callable[[float], float]:
        raise NotImplementedError

    def _similarity_search_with_relevance_scores(self, query: str, k: int = 4, **kwargs: Any):
        fn = self._select_relevance_score_fn()
        return [(doc, fn(score)) for doc, score in self.similarity_search_with_score(query, k, **kwargs)]

    async def _asimilarity_search_with_relevance_scores(self, query: str, k: int = 4, **kwargs: Any):
        fn = self._select_relevance_score_fn()
        return [(doc, fn(score)) for doc, score in await self.asimilarity_search_with_score(query, k, **kwargs)]

    @staticmethod
    def _filter_relevance(pairs, score_threshold):
        if any(s < 0.0 or s > 1.0 for _, s in pairs):
            warnings.warn(f"relevance scores must lie in [0, 1], got {pairs}", stacklevel=3)
        if score_threshold is not None:
            pairs = [(d, s) for d, s in pairs if s >= score_threshold]
            if not pairs:
                logger.warning("no document passed the relevance threshold %s", score_threshold)
        return pairs

    def similarity_search_with_relevance_scores(self, query: str, k: int = 4, **kwargs: Any):
        thr = kwargs.pop("score_threshold", None)
        return self._filter_relevance(self._similarity_search_with_relevance_scores(query, k=k, **kwargs), thr)

    async def asimilarity_search_with_relevance_scores(self, query: str, k: int = 4, **kwargs: Any):
        thr = kwargs.pop("score_threshold", None)
        return self._filter_relevance(await self._asimilarity_search_with_relevance_scores(query, k=k, **kwargs), thr)

    # ------------------------------------------------------------------ construction
    @classmethod
    @abstractmethod
    def from_texts(cls, texts: List[str], embedding, metadatas: Optional[List[dict]] = None, *,
                   ids: Optional[List[str]] = None, **kwargs: Any) -> "VectorStore":
        ...

    @classmethod
    def from_documents(cls, documents: List[Document], embedding, **kwargs: Any) -> "VectorStore":
        if "ids" not in kwargs:
            ids = [d.id for d in documents]
            if any(ids):
                kwargs["ids"] = ids
        return cls.from_texts([d.content for d in documents], embedding, [d.metadata for d in documents], **kwargs)

    @classmethod
    async def afrom_documents(cls, documents: List[Document], embedding, **kwargs: Any) -> "VectorStore":
        if "ids" not in kwargs:
            ids = [d.id for d in documents]
            if any(ids):
                kwargs["ids"] = ids
        return await cls.afrom_texts([d.content for d in documents], embedding, metadatas=[d.metadata for d in documents], **kwargs)

    @classmethod
    async def afrom_texts(cls, texts: List[str], embedding, metadatas: Optional[List[dict]] = None, *,
                          ids: Optional[List[str]] = None, **kwargs: Any) -> "VectorStore":
        if ids is not None:
            kwargs["ids"] = ids
        return await _in_pool(cls.from_texts, texts, embedding, metadatas, **kwargs)

    # ------------------------------------------------------------------ retriever view (VectorStoreBase.py:613-627)
    def _get_retriever_tags(self) -> List[str]:
        tags = [self.__class__.__name__]
        emb = getattr(self, "embeddings", None) or getattr(self, "embedding", None)
        if emb:
            tags.append(emb.__class__.__name__)
        return tags

    def as_retriever(self, **kwargs: Any):
        """The reference imports its retriever from a module that does not exist (`rag_arc.core.search...`,
        VectorStoreBase.py:624): the call cannot succeed there.  Here it returns the mirror's VectorStoreRetriever."""
        from ....core.retrieval.dense import VectorStoreRetriever

        tags = kwargs.pop("tags", None) or [] + self._get_retriever_tags()
        return VectorStoreRetriever(self, tags=tags, **kwargs)
