"""Lookup-table embedding provider: texts whose vectors were computed offline (.npz with `texts`
and `vectors`).  Unknown texts raise KeyError.  For pipelines whose embeddings come from elsewhere (a caller that
keeps its own encoder and hands over vectors) and for the plumbing tests; the encoder on the MI355X is
`hip_bert.HipBertEmbeddings` (config tag `hip_bert_embeddings`)."""
from typing import Dict, List, Sequence

import numpy as np

from .base import Embeddings


class TableEmbeddings(Embeddings):
    def __init__(self, texts: Sequence[str], vectors, **kwargs):
        super().__init__(**kwargs)
        vectors = np.asarray(vectors, dtype=np.float32)
        if len(texts) != vectors.shape[0]:
            raise ValueError("texts and vectors differ in length")
        self._table: Dict[str, np.ndarray] = {t: v for t, v in zip(texts, vectors)}
        self._row: Dict[str, int] = {t: i for i, t in enumerate(texts)}     # (a text listed twice: its last vector, as above)
        self._vectors = vectors
        self._device_table = None
        self.dim = int(vectors.shape[1])

    @classmethod
    def from_npz(cls, path: str) -> "TableEmbeddings":
        data = np.load(path, allow_pickle=False)
        return cls([str(t) for t in data["texts"]], data["vectors"])

    def embed_documents(self, texts: List[str]) -> List[List[float]]:
        return [self._table[t.replace("\n", " ")].tolist() for t in texts]

    def embed_query(self, text: str) -> List[float]:
        return self.embed_documents([text])[0]

    # -- batch forms (what HipFlatVectorStore's batch entry points and its query coalescer look for) --------------------
    def _rows(self, texts: Sequence[str]) -> List[int]:
        return [self._row[t.replace("\n", " ")] for t in texts]

    def embed_queries(self, texts: Sequence[str]) -> np.ndarray:
        """embed_query of every text as one float32 array [n][dim] (no python floats in between)."""
        return self._vectors[self._rows(texts)]

    def to_device(self, device: int = 0) -> "TableEmbeddings":
        """Keep the table in HBM: `embed_queries_device` then gathers query vectors there (no H2D per batch)."""
        import torch

        self._device_table = torch.from_numpy(self._vectors).to(torch.device("cuda", device))
        self.embed_queries_device = self._embed_queries_device
        self.embed_documents_device = self._embed_queries_device
        return self

    def _embed_queries_device(self, texts: Sequence[str]):
        import torch

        tab = self._device_table
        return tab[torch.as_tensor(self._rows(texts), dtype=torch.long).to(tab.device, non_blocking=True)]
