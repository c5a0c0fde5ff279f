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
        self.dim = int(vectors.shape[1])

    @classmethod
    def from_npz(cls, path: str) -> "TableEmbeddings":
        data = np.load(path, allow_pickle=False)
        return cls([str(t) for t in data["texts"]], data["vectors"])

    def embed_documents(self, texts: List[str]) -> List[List[float]]:
        return [self._table[t.replace("\n", " ")].tolist() for t in texts]

    def embed_query(self, text: str) -> List[float]:
        return self.embed_documents([text])[0]
