"""Embedding-distance chunking with the cosines computed on the MI355X.

Mirror of the semantic part of the reference's splitter module (core/file_management/chunker/spliter.py):
  cosine_similarity(X, Y)            :307-332   float64 cosine matrix, non-finite entries -> 0
  combine_sentences                  :335-351   each sentence joined with `buffer_size` neighbours either side
  calculate_cosine_distances         :354-371   1 - cosine of consecutive combined-sentence embeddings
  SemanticChunker                    :374-534   regex sentence split -> embed -> distances -> breakpoints -> chunks
The arithmetic (rarc_cosine_matrix_f32 / rarc_adjacent_cosine_distance_f32, csrc/prep.hip) takes fp32 rows — what an
embedding provider returns — and sums in float64; when the provider can hand its embeddings over on the device
(`embed_documents_device`, as HipBertEmbeddings does) they never visit the host.  Everything else here is string
and list handling on the host.  There is no CPU fallback: without the HIP library these functions raise.
"""
from __future__ import annotations

import re
from typing import Any, Dict, List, Optional, Sequence, Tuple

import numpy as np

from ....hip import binding as B

BREAKPOINT_DEFAULTS: Dict[str, float] = {"percentile": 95, "standard_deviation": 3, "interquartile": 1.5, "gradient": 95}


def _device_rows(torch, data):
    """fp32 [n][d] tensor on the current ROCm device from lists / numpy / a tensor."""
    if not torch.cuda.is_available():
        raise B.RarcError("no ROCm device visible: the chunker's cosine kernels have no CPU fallback")
    if isinstance(data, torch.Tensor):
        t = data.to("cuda", torch.float32)
    else:
        t = torch.from_numpy(np.ascontiguousarray(np.asarray(data, dtype=np.float32))).to("cuda")
    if t.dim() != 2:
        raise ValueError(f"expected a matrix of row vectors, got shape {tuple(t.shape)}")
    return t.contiguous()


def cosine_similarity(X, Y) -> np.ndarray:
    """All-pairs cosine of the rows of X and Y as a float64 [len(X)][len(Y)] array (spliter.py:307-332)."""
    if len(X) == 0 or len(Y) == 0:
        return np.array([])
    import torch

    x, y = _device_rows(torch, X), _device_rows(torch, Y)
    if x.shape[1] != y.shape[1]:
        raise ValueError(f"Number of columns in X and Y must be the same. X has shape {tuple(x.shape)} "
                         f"and Y has shape {tuple(y.shape)}.")
    out = torch.empty((x.shape[0], y.shape[0]), dtype=torch.float64, device=x.device)
    B.check(B.load_library().rarc_cosine_matrix_f32(x.data_ptr(), x.stride(0), x.shape[0], y.data_ptr(), y.stride(0),
                                                    y.shape[0], x.shape[1], out.data_ptr(),
                                                    torch.cuda.current_stream().cuda_stream))
    return out.cpu().numpy()


def device_cosine_distances(embeddings) -> List[float]:
    """1 - cosine(e_i, e_{i+1}) for consecutive rows, one kernel launch; python floats like the reference's."""
    import torch

    e = _device_rows(torch, embeddings)
    if e.shape[0] < 2:
        return []
    out = torch.empty(e.shape[0] - 1, dtype=torch.float64, device=e.device)
    B.check(B.load_library().rarc_adjacent_cosine_distance_f32(e.data_ptr(), e.stride(0), e.shape[0], e.shape[1],
                                                               out.data_ptr(), torch.cuda.current_stream().cuda_stream))
    return out.cpu().tolist()


def combine_sentences(sentences: List[dict], buffer_size: int = 1) -> List[dict]:
    """sentences[i]["combined_sentence"] = sentences i-buffer_size .. i+buffer_size joined by single spaces."""
    texts = [s["sentence"] for s in sentences]
    for i, entry in enumerate(sentences):
        entry["combined_sentence"] = " ".join(texts[max(0, i - buffer_size): i + buffer_size + 1])
    return sentences


def calculate_cosine_distances(sentences: List[dict]) -> Tuple[List[float], List[dict]]:
    """Distances between consecutive "combined_sentence_embedding" entries; stored as "distance_to_next" too."""
    distances = device_cosine_distances([s["combined_sentence_embedding"] for s in sentences]) if len(sentences) > 1 else []
    for entry, dist in zip(sentences, distances):
        entry["distance_to_next"] = dist
    return distances, sentences


class SemanticChunker:
    """Splits text where the embedding distance between neighbouring sentence windows jumps (spliter.py:374-534)."""

    def __init__(self, embeddings, buffer_size: int = 1, add_start_index: bool = False,
                 breakpoint_threshold_type: str = "percentile", breakpoint_threshold_amount: Optional[float] = None,
                 number_of_chunks: Optional[int] = None, sentence_split_regex: str = r"(?<=[.?!])\s+",
                 min_chunk_size: Optional[int] = None):
        self.embeddings = embeddings
        self.buffer_size = buffer_size
        self._add_start_index = add_start_index
        self.breakpoint_threshold_type = breakpoint_threshold_type
        self.breakpoint_threshold_amount = (BREAKPOINT_DEFAULTS[breakpoint_threshold_type]
                                            if breakpoint_threshold_amount is None else breakpoint_threshold_amount)
        self.number_of_chunks = number_of_chunks
        self.sentence_split_regex = sentence_split_regex
        self.min_chunk_size = min_chunk_size

    # ---- thresholds (host numpy on a handful of floats) ----
    def _calculate_breakpoint_threshold(self, distances: Sequence[float]) -> Tuple[float, Sequence[float]]:
        kind, amount = self.breakpoint_threshold_type, self.breakpoint_threshold_amount
        if kind == "percentile":
            return float(np.percentile(distances, amount)), distances
        if kind == "standard_deviation":
            return float(np.mean(distances) + amount * np.std(distances)), distances
        if kind == "interquartile":
            q1, q3 = np.percentile(distances, [25, 75])
            return np.mean(distances) + amount * (q3 - q1), distances
        if kind == "gradient":
            slope = np.gradient(distances, range(0, len(distances)))
            return float(np.percentile(slope, amount)), slope
        raise ValueError(f"Got unexpected `breakpoint_threshold_type`: {kind}")

    def _threshold_from_clusters(self, distances: Sequence[float]) -> float:
        if self.number_of_chunks is None:
            raise ValueError("This should never be called if `number_of_chunks` is None.")
        # percentile falls linearly from 100 (one chunk) to 0 (as many chunks as distances)
        n = len(distances)
        want = max(min(self.number_of_chunks, n), 1.0)
        pct = 100.0 if n == 1.0 else (100.0 / (1.0 - n)) * (want - n)
        return float(np.percentile(distances, min(max(pct, 0), 100)))

    def _calculate_sentence_distances(self, single_sentences_list: List[str]) -> Tuple[List[float], List[dict]]:
        sentences = combine_sentences([{"sentence": s, "index": i} for i, s in enumerate(single_sentences_list)],
                                      self.buffer_size)
        windows = [s["combined_sentence"] for s in sentences]
        on_device = getattr(self.embeddings, "embed_documents_device", None)
        if on_device is not None:            # embeddings stay in HBM: encoder output -> distance kernel
            dev = on_device(windows)
            distances = device_cosine_distances(dev)
            host = dev.float().cpu().tolist()
            for entry, emb in zip(sentences, host):
                entry["combined_sentence_embedding"] = emb
            for entry, dist in zip(sentences, distances):
                entry["distance_to_next"] = dist
            return distances, sentences
        for entry, emb in zip(sentences, self.embeddings.embed_documents(windows)):
            entry["combined_sentence_embedding"] = emb
        return calculate_cosine_distances(sentences)

    def split_text(self, text: str) -> List[str]:
        pieces = re.split(self.sentence_split_regex, text)
        if len(pieces) == 1 or (self.breakpoint_threshold_type == "gradient" and len(pieces) == 2):
            return pieces                    # np.percentile / np.gradient need more than that
        distances, sentences = self._calculate_sentence_distances(pieces)
        if self.number_of_chunks is not None:
            threshold, series = self._threshold_from_clusters(distances), distances
        else:
            threshold, series = self._calculate_breakpoint_threshold(distances)
        chunks: List[str] = []
        first = 0
        for cut in (i for i, v in enumerate(series) if v > threshold):
            body = " ".join(s["sentence"] for s in sentences[first: cut + 1])
            if self.min_chunk_size is not None and len(body) < self.min_chunk_size:
                continue                     # too short: keep growing the same chunk
            chunks.append(body)
            first = cut + 1
        if first < len(sentences):
            chunks.append(" ".join(s["sentence"] for s in sentences[first:]))
        return chunks
