"""Vector-store backed retriever (reference: core/retrieval/dense.py:13-380).

search_type: "similarity" | "similarity_score_threshold" | "mmr".  Sync path: k defaults to 5 and
the result is cut to k; the async path neither defaults k nor truncates (reference quirk, kept).
Unknown keyword arguments (e.g. the `top_k` that MultiPathRetriever leaks down) are passed on to
the store untouched.  Errors are logged and re-raised.
"""
import logging
from typing import Any, Dict, List, Optional

from ..utils.data_model import Document
from .base import BaseRetriever

logger = logging.getLogger(__name__)

_SEARCH_TYPES = ("similarity", "similarity_score_threshold", "mmr")


class VectorStoreRetriever(BaseRetriever):
    allowed_search_types = _SEARCH_TYPES

    def __init__(self, vectorstore, **kwargs):
        self.vectorstore = vectorstore
        self.search_type = kwargs.get("search_type", "similarity")
        self.search_kwargs = kwargs.get("search_kwargs", {})
        self._validate_search_config()
        super().__init__(**kwargs)

    def _validate_search_config(self) -> None:
        if self.search_type not in self.allowed_search_types:
            raise ValueError(f"search_type '{self.search_type}' is not allowed; valid: {self.allowed_search_types}")
        if self.search_type == "similarity_score_threshold":
            thr = self.search_kwargs.get("score_threshold")
            if thr is None or not isinstance(thr, (int, float)) or not (0 <= thr <= 1):
                raise ValueError("'similarity_score_threshold' needs search_kwargs['score_threshold'] in [0, 1]")

    # ------------------------------------------------------------------ sync
    def _get_relevant_documents(self, query: str, **kwargs: Any) -> List[Document]:
        params: Dict[str, Any] = {**self.search_kwargs, **kwargs}
        k = params.get("k", getattr(self, "k", 5))
        params["k"] = k
        try:
            if self.search_type == "similarity":
                docs = self.vectorstore.similarity_search(query, **params)
            elif self.search_type == "similarity_score_threshold":
                docs = [d for d, _ in self.vectorstore.similarity_search_with_relevance_scores(query, **params)]
            elif self.search_type == "mmr":
                docs = self.vectorstore.max_marginal_relevance_search(query, **params)
            else:
                raise ValueError(f"unsupported search type: {self.search_type}")
            return docs[:k]
        except Exception as exc:
            logger.error("retrieval failed: %s", exc)
            raise

    def batch_invoke(self, inputs: List[str], **kwargs: Any) -> List[List[Document]]:
        """invoke() for a list of queries in ONE pass of the store (one encoder call, one scan per 256 queries) when the
        store can batch (`batch_similarity_search`: HipFlatVectorStore and its sharded form) and the search type is
        "similarity"; otherwise query by query.  Element i equals invoke(inputs[i], **kwargs)."""
        params: Dict[str, Any] = {**self.search_kwargs, **kwargs}
        k = params.get("k", getattr(self, "k", 5))
        params["k"] = k
        batched = getattr(self.vectorstore, "batch_similarity_search", None)
        if self.search_type != "similarity" or batched is None:
            return [self.invoke(q, **kwargs) for q in inputs]
        try:
            return [docs[:k] for docs in batched(list(inputs), **params)]
        except Exception as exc:
            # element i must equal invoke(inputs[i]): a query that cannot be answered raises from ITS invoke, the ones
            # before it are not lost to a caller that catches per query (MultiPathRetriever.batch_invoke does)
            logger.error("batched retrieval failed (%s): answering query by query", exc)
            return [self.invoke(q, **kwargs) for q in inputs]

    # ------------------------------------------------------------------ async
    async def _aget_relevant_documents(self, query: str, **kwargs: Any) -> List[Document]:
        params: Dict[str, Any] = {**self.search_kwargs, **kwargs}
        try:
            if self.search_type == "similarity":
                return await self.vectorstore.asimilarity_search(query, **params)
            if self.search_type == "similarity_score_threshold":
                pairs = await self.vectorstore.asimilarity_search_with_relevance_scores(query, **params)
                return [d for d, _ in pairs]
            if self.search_type == "mmr":
                return await self.vectorstore.amax_marginal_relevance_search(query, **params)
            raise ValueError(f"unsupported search type: {self.search_type}")
        except Exception as exc:
            logger.error("async retrieval failed: %s", exc)
            raise

    # ------------------------------------------------------------------ passthroughs (dense.py:219-330: logged, re-raised)
    def add_documents(self, documents: List[Document], **kwargs: Any) -> List[str]:
        try:
            ids = self.vectorstore.add_documents(documents, **kwargs)
            logger.info("added %d documents to the vector store", len(documents))
            return ids
        except Exception as exc:
            logger.error("adding documents failed: %s", exc)
            raise

    async def aadd_documents(self, documents: List[Document], **kwargs: Any) -> List[str]:
        try:
            ids = await self.vectorstore.aadd_documents(documents, **kwargs)
            logger.info("added %d documents to the vector store (async)", len(documents))
            return ids
        except Exception as exc:
            logger.error("adding documents failed (async): %s", exc)
            raise

    def delete_documents(self, ids: Optional[List[str]] = None, **kwargs: Any) -> Optional[bool]:
        """ids = None deletes everything (dense.py:255-274)."""
        try:
            result = self.vectorstore.delete(ids, **kwargs)
            logger.info("deleted %s documents", len(ids) if ids else "all")
            return result
        except Exception as exc:
            logger.error("deleting documents failed: %s", exc)
            raise

    async def adelete_documents(self, ids: Optional[List[str]] = None, **kwargs: Any) -> Optional[bool]:
        try:
            result = await self.vectorstore.adelete(ids, **kwargs)
            logger.info("deleted %s documents (async)", len(ids) if ids else "all")
            return result
        except Exception as exc:
            logger.error("deleting documents failed (async): %s", exc)
            raise

    def get_by_ids(self, ids: List[str]) -> List[Document]:
        try:
            return self.vectorstore.get_by_ids(ids)
        except Exception as exc:
            logger.error("get_by_ids failed: %s", exc)
            raise

    async def aget_by_ids(self, ids: List[str]) -> List[Document]:
        try:
            return await self.vectorstore.aget_by_ids(ids)
        except Exception as exc:
            logger.error("get_by_ids failed (async): %s", exc)
            raise

    def get_vectorstore_info(self) -> Dict[str, Any]:
        info = {"vectorstore_class": self.vectorstore.__class__.__name__, "search_type": self.search_type,
                "search_kwargs": self.search_kwargs, "allowed_search_types": list(self.allowed_search_types)}
        if hasattr(self.vectorstore, "embeddings") and self.vectorstore.embeddings:
            info["embedding_class"] = self.vectorstore.embeddings.__class__.__name__
        elif hasattr(self.vectorstore, "embedding"):
            info["embedding_class"] = self.vectorstore.embedding.__class__.__name__
        return info

    def get_name(self) -> str:
        return f"{self.vectorstore.__class__.__name__}Retriever"

    def update_search_params(self, **kwargs: Any) -> None:
        """Every keyword lands in search_kwargs — `search_type` too, besides switching the type (dense.py:357-370)."""
        self.search_kwargs.update(kwargs)
        if "search_type" in kwargs:
            self.search_type = kwargs["search_type"]
            self._validate_search_config()

    def __repr__(self) -> str:
        return (f"{self.__class__.__name__}(vectorstore={self.vectorstore.__class__.__name__}, "
                f"search_type='{self.search_type}', search_kwargs={self.search_kwargs})")
