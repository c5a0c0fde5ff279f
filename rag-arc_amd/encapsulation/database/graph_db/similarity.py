"""All pairs of embeddings whose cosine reaches a threshold, on the MI355X (csrc/pairs.hip, `rarc_similar_pairs`).

What it replaces in the reference (encapsulation/database/graph_db/Base_Neo4j.py:559-583):

    embeddings_array = np.array(embeddings)
    similarity_matrix = cosine_similarity(embeddings_array)          # sklearn: n x n float64
    for i in range(len(embeddings)):
        for j in range(i + 1, len(embeddings)):
            if similarity_matrix[i][j] >= similarity_threshold:      # 0.95
                ... CREATE (e1)-[:SIMILAR {similarity: float(similarity_matrix[i][j])}]->(e2)

`similar_pairs(embeddings, 0.95)` returns those (i, j, similarity) triples in that loop's order.  The matrix is never formed
(the score GEMM nominates, an exact double-precision pass decides), so 100,000 entities cost tens of milliseconds instead of
80 GB of float64 and 5e9 interpreter steps.  No CPU fallback: without the HIP library this raises."""
from __future__ import annotations

import ctypes
from typing import List, Tuple

import numpy as np

from ....hip import binding as B

MAX_DIM = 4096


def similar_pairs(embeddings, threshold: float = 0.95, device: int = 0, as_arrays: bool = False):
    """[(i, j, cosine)] for every i < j with cosine(embeddings[i], embeddings[j]) >= threshold, ordered by (i, j).
    embeddings: [n][d] array-like of floats (the reference holds python lists) or a torch tensor (used where it is).
    as_arrays=True returns (pairs int64 [m][2], cosines float64 [m]) instead — half a million pairs are 0.1 s of python
    objects otherwise."""
    import torch

    if not torch.cuda.is_available():
        raise B.RarcError("no ROCm device visible: the HIP backend has no CPU fallback")
    lib = B.load_library()
    dev = torch.device("cuda", device)
    if isinstance(embeddings, torch.Tensor):
        x = embeddings.to(device=dev, dtype=torch.float32)
    else:
        arr = np.asarray(embeddings, dtype=np.float32)
        if arr.ndim != 2:
            if arr.size == 0:
                return (np.zeros((0, 2), np.int64), np.zeros(0, np.float64)) if as_arrays else []
            raise ValueError(f"expected [n][d] embeddings, got an array of shape {arr.shape}")
        x = torch.from_numpy(np.ascontiguousarray(arr)).to(dev)
    if x.ndim != 2:
        raise ValueError(f"expected [n][d] embeddings, got a tensor of shape {tuple(x.shape)}")
    n, d = int(x.shape[0]), int(x.shape[1])
    if n < 2:
        return (np.zeros((0, 2), np.int64), np.zeros(0, np.float64)) if as_arrays else []
    if not 1 <= d <= MAX_DIM:
        raise B.RarcError(f"embeddings of {d} dimensions: the all-pairs kernel takes 1..{MAX_DIM}")
    if not 0.0 < float(threshold) <= 1.0:
        raise ValueError("threshold must lie in (0, 1]")
    x = x.contiguous()
    cand_cap, out_cap = 256, max(1024, 4 * n)
    n_pad = (n + 255) // 256 * 256
    count = torch.zeros(1, dtype=torch.int64, device=dev)
    flags = torch.zeros(1, dtype=torch.int32, device=dev)
    with torch.cuda.device(dev):
        while True:
            ws = torch.empty(int(lib.rarc_similar_pairs_workspace_bytes(n, d, cand_cap)), dtype=torch.uint8, device=dev)
            pairs = torch.empty((out_cap, 2), dtype=torch.int64, device=dev)
            scores = torch.empty(out_cap, dtype=torch.float64, device=dev)
            B.check(lib.rarc_similar_pairs(x.data_ptr(), x.stride(0), n, d, ctypes.c_double(float(threshold)), ws.data_ptr(),
                                           ws.numel(), cand_cap, pairs.data_ptr(), scores.data_ptr(), out_cap, count.data_ptr(),
                                           flags.data_ptr(), torch.cuda.current_stream(dev).cuda_stream), "rarc_similar_pairs")
            f, found = int(flags.item()), int(count.item())
            if f == 0:
                break
            if f & 1:
                if cand_cap >= n_pad:
                    raise B.RarcError("rarc_similar_pairs flagged a nomination list that holds every row")
                cand_cap = min(4 * cand_cap, n_pad)
            if f & 2:
                out_cap = max(found, 2 * out_cap)       # (an overflowed list hid nominations: the count may still grow)
        p = pairs[:found].cpu().numpy()
        s = scores[:found].cpu().numpy()
    order = np.lexsort((p[:, 1], p[:, 0]))
    p, s = p[order], s[order]
    if as_arrays:
        return p, s
    return list(zip(p[:, 0].tolist(), p[:, 1].tolist(), s.tolist()))
