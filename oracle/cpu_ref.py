"""cpu_ref.py — CPU ORACLE, python side.  TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module; the
product package (rag-arc_amd/) never does and fails loudly without its HIP library.

What is restated here, and how it is pinned:

* flat inner-product search / L2 normalisation (faiss arithmetic reached from
  encapsulation/database/vector_db/VectorStore_Faiss.py:150-154, :170-202, :258-272):
  thin ctypes wrappers over oracle/rarc_oracle.c.  faiss is not vendored in /root/reference and not
  installed here; PINNED instead to the reference's own float64 cosine (spliter.py:326-332) run on
  64 x 4096 fp16-representable vectors at d = 384 / 768 (tests/golden/cosine_pin_d*.npz, 1e-5; see
  the header of rarc_oracle.c and DESIGN.md §2).  A float64 numpy cross-check (`flat_search_f64`)
  bounds the canonical fp32 scores to the 1e-5 the north star asks for on any input.
* BERT forward (`bert_forward_f32`, float32 or float64) and the reranker's LM forward
  (`qwen3_last_logits_f32`): numpy graphs PINNED against transformers.BertModel /
  transformers.Qwen3ForCausalLM with seeded weights (tests/test_oracle_golden.py).
* reciprocal-rank fusion (core/utils/Fusion.py:45-76), the relevance-score quirk
  (encapsulation/database/vector_db/VectorStoreBase.py:263-266) and the reranker's
  score->order step (core/rerank/Reranker_Qwen3.py:41-49, :70-74): pure python / numpy below,
  PINNED against outputs of the reference's own importable code, committed under tests/golden/
  by tests/golden/make_golden.py.
* all-pairs cosine >= threshold (`similar_pairs_f64`: the entity de-duplication of
  encapsulation/database/graph_db/Base_Neo4j.py:559-583, which calls scikit-learn's cosine_similarity — a
  dependency, not vendored): float64 numpy restatement, PINNED against scikit-learn itself (installed here:
  tests/test_similar_pairs_host.py) and against tests/golden/similar_pairs.json.
"""
from __future__ import annotations

import ctypes
import os
import subprocess
from collections import defaultdict
from typing import Hashable, List, Sequence, Tuple

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "_build", "librarc_oracle.so")
_lib = None


def build(force: bool = False) -> str:
    """Compile rarc_oracle.c with gcc (oracle/Makefile).  Returns the .so path."""
    if force or not os.path.exists(_LIB_PATH):
        subprocess.run(["make", "-C", _HERE] + (["-B"] if force else []), check=True,
                       stdout=subprocess.DEVNULL)
    return _LIB_PATH


def lib() -> ctypes.CDLL:
    global _lib
    if _lib is None:
        build()
        L = ctypes.CDLL(_LIB_PATH)
        if not L.oracle_cpu_ok():
            raise RuntimeError("oracle needs an x86-64 CPU with AVX2 + FMA + F16C")
        L.oracle_canon_dot_f16.restype = ctypes.c_float
        L.oracle_canon_dot_f16_scalar.restype = ctypes.c_float
        L.oracle_synth_ppnd.restype = ctypes.c_double
        L.oracle_synth_ppnd.argtypes = [ctypes.c_double]
        L.oracle_synth_log.restype = ctypes.c_double
        L.oracle_synth_log.argtypes = [ctypes.c_double]
        L.oracle_synth_val.restype = ctypes.c_int32
        L.oracle_synth_val.argtypes = [ctypes.c_uint64, ctypes.c_uint64, ctypes.c_uint32]
        _lib = L
    return _lib


def _p(a: np.ndarray):
    return a.ctypes.data_as(ctypes.c_void_p)


def padded_dim(d: int, align: int = 128) -> int:
    return ((d + align - 1) // align) * align


# ----------------------------------------------------------------------------- normalise / ingest
def normalize_L2(x: np.ndarray) -> np.ndarray:
    """faiss.normalize_L2 (VectorStore_Faiss.py:153): returns a normalised fp32 copy."""
    y = np.array(x, dtype=np.float32, order="C", copy=True)
    if y.ndim != 2:
        raise ValueError("expected a 2-d array")
    if y.size:
        lib().oracle_normalize_rows_f32(_p(y), ctypes.c_int64(y.shape[1]), ctypes.c_int64(y.shape[0]),
                                        ctypes.c_int(y.shape[1]))
    return y


def ingest_f16(x: np.ndarray, normalize: bool = True, d_pad: int | None = None) -> Tuple[np.ndarray, np.ndarray]:
    """add_texts side (VectorStore_Faiss.py:170-202) for an fp16 flat index: returns
    (rows as uint16 bit patterns [n][d_pad], squared norms of the stored rows [n])."""
    x = np.ascontiguousarray(x, dtype=np.float32)
    n, d = x.shape
    d_pad = d_pad or padded_dim(d)
    out = np.zeros((n, d_pad), dtype=np.uint16)
    n2 = np.zeros(n, dtype=np.float32)
    if n:
        lib().oracle_ingest_f16(_p(x), ctypes.c_int64(d), _p(out), ctypes.c_int(d_pad), _p(n2),
                                ctypes.c_int64(n), ctypes.c_int(d), ctypes.c_int(1 if normalize else 0))
    return out, n2


def ingest_f8(x: np.ndarray, normalize: bool = True, d_pad: int | None = None):
    """add_texts side for an fp8 (e4m3fn + per-row fp32 scale) flat index — BASELINE config 5's storage.
    Returns (bytes uint8 [n][d_pad], scales fp32 [n], squared norms of the stored rows [n]).
    d_pad defaults to the next multiple of 256 (the fp8 scan reads 16-byte chunks of 16 values)."""
    x = np.ascontiguousarray(x, dtype=np.float32)
    n, d = x.shape
    d_pad = d_pad or padded_dim(d, 256)
    out = np.zeros((n, d_pad), dtype=np.uint8)
    sc = np.ones(n, dtype=np.float32)
    n2 = np.zeros(n, dtype=np.float32)
    if n:
        lib().oracle_ingest_f8(_p(x), ctypes.c_int64(d), _p(out), ctypes.c_int(d_pad), _p(sc), _p(n2),
                               ctypes.c_int64(n), ctypes.c_int(d), ctypes.c_int(1 if normalize else 0))
    return out, sc, n2


def f8_decode(b: np.ndarray) -> np.ndarray:
    """e4m3fn bytes -> fp32 (exact)."""
    L = lib()
    L.oracle_f8_decode.restype = ctypes.c_float
    lut = np.array([L.oracle_f8_decode(ctypes.c_uint8(i)) for i in range(256)], dtype=np.float32)
    return lut[np.asarray(b, dtype=np.uint8)]


def f8_encode(x: np.ndarray) -> np.ndarray:
    """fp32 -> e4m3fn bytes: round to nearest even, saturating at 448 (scalar loop: small inputs only)."""
    L = lib()
    L.oracle_f8_encode.restype = ctypes.c_uint8
    flat = np.asarray(x, dtype=np.float32).ravel()
    return np.array([L.oracle_f8_encode(ctypes.c_float(float(v))) for v in flat], dtype=np.uint8).reshape(np.shape(x))


def flat_search_f8(corpus_u8: np.ndarray, scales: np.ndarray, q32: np.ndarray, k: int, id_base: int = 0):
    """IndexFlatIP.search over fp8 rows: score = scale[r] * canonical fp32 dot(q, decoded bytes).
    Ties: id ascending.  Returns (ids int64 [nq][k], scores fp32 [nq][k], threads used)."""
    corpus_u8 = np.ascontiguousarray(corpus_u8, dtype=np.uint8)
    scales = np.ascontiguousarray(scales, dtype=np.float32)
    n, d_pad = corpus_u8.shape
    q32 = np.ascontiguousarray(q32, dtype=np.float32)
    if q32.shape[1] != d_pad:
        q32 = pad_queries(q32, d_pad)
    nq = q32.shape[0]
    ids = np.full((nq, k), -1, dtype=np.int64)
    sc = np.full((nq, k), -np.inf, dtype=np.float32)
    L = lib()
    L.oracle_flat_search_f8.restype = ctypes.c_int
    nt = L.oracle_flat_search_f8(_p(corpus_u8), _p(scales), ctypes.c_int64(n), ctypes.c_int(d_pad), _p(q32),
                                 ctypes.c_int(nq), ctypes.c_int(k), ctypes.c_int64(id_base), _p(ids), _p(sc))
    return ids, sc, int(nt)


def ingest_f32(x: np.ndarray, normalize: bool = True, d_pad: int | None = None):
    """add_texts side for fp32 storage — the reference's own (VectorStore_Faiss.py:170-202: astype(float32),
    normalize_L2, index.add): returns (fp32 rows [n][d_pad], squared norms [n])."""
    x = np.ascontiguousarray(x, dtype=np.float32)
    n, d = x.shape
    d_pad = d_pad or padded_dim(d)
    out = np.zeros((n, d_pad), dtype=np.float32)
    n2 = np.zeros(n, dtype=np.float32)
    if n:
        lib().oracle_ingest_f32(_p(x), ctypes.c_int64(d), _p(out), ctypes.c_int(d_pad), _p(n2), ctypes.c_int64(n),
                                ctypes.c_int(d), ctypes.c_int(1 if normalize else 0))
    return out, n2


def flat_search_f32(corpus_f32: np.ndarray, q32: np.ndarray, k: int, id_base: int = 0):
    """IndexFlatIP.search over fp32 rows (VectorStore_Faiss.py:262-263): canonical fp32 scores, (score desc, id asc)."""
    corpus_f32 = np.ascontiguousarray(corpus_f32, dtype=np.float32)
    n, d_pad = corpus_f32.shape
    qp = pad_queries(q32, d_pad)
    nq = qp.shape[0]
    ids = np.full((nq, k), -1, dtype=np.int64)
    sc = np.full((nq, k), -np.inf, dtype=np.float32)
    nt = lib().oracle_flat_search_f32(_p(corpus_f32), ctypes.c_int64(n), ctypes.c_int(d_pad), _p(qp), ctypes.c_int(nq),
                                      ctypes.c_int(k), ctypes.c_int64(id_base), _p(ids), _p(sc))
    return ids, sc, int(nt)


def blas_threads() -> int:
    """Threads the BLAS behind numpy's matmul will use (threadpoolctl when it is installed, else the CPU count)."""
    try:
        from threadpoolctl import threadpool_info

        n = [int(i.get("num_threads", 0)) for i in threadpool_info() if i.get("user_api") == "blas"]
        if n:
            return max(n)
    except Exception:  # noqa: BLE001
        pass
    return os.cpu_count() or 1


def flat_search_blas_f32(corpus_f32: np.ndarray, q32: np.ndarray, k: int, chunk_rows: int = 131072, workers: int | None = None):
    """The CPU BASELINE SURVEY.md §8(d) defines — "numpy fp32 `normalize` -> `D @ q` -> top-k" — i.e. what the reference's path
    (VectorStore_Faiss.py:258-263: fp32 query, `index.search` on an IndexFlatIP) costs when its inner products come from the
    host's BLAS, which is what faiss itself does for batches of 20 queries and more (sgemm over row blocks + a per-query
    selection).  corpus_f32 [n][d] (rows already normalised, fp32 as faiss stores them), q32 [nq][d] normalised queries.
    Scores are sgemm's (its summation order, not the canonical one): NOT the parity oracle — tests hold it to the canonical
    oracle within 1e-5 and to the same ids wherever gaps exceed that.  Row chunks of `chunk_rows`: S = Q @ D_chunkᵀ [nq][chunk]
    (BLAS, all its threads), then each query's top-k of the chunk by argpartition (a thread pool over the query rows: numpy
    releases the GIL inside partition), merged into the running top-k.  Returns (ids int64 [nq][k], scores fp32 [nq][k]) ordered
    by (score desc, id asc), and the BLAS thread count."""
    from concurrent.futures import ThreadPoolExecutor

    D = np.ascontiguousarray(corpus_f32, dtype=np.float32)
    Q = np.ascontiguousarray(q32, dtype=np.float32)
    n, nq = D.shape[0], Q.shape[0]
    k = min(int(k), n)
    workers = workers or min(os.cpu_count() or 1, 64, max(1, nq))
    best_s = np.full((nq, 0), -np.inf, np.float32)
    best_i = np.zeros((nq, 0), np.int64)

    def select(args):
        S, lo, hi, kk = args
        part = np.argpartition(-S[lo:hi], kk - 1, axis=1)[:, :kk]
        return part

    with ThreadPoolExecutor(max_workers=workers) as pool:
        for c0 in range(0, n, chunk_rows):
            c1 = min(n, c0 + chunk_rows)
            S = Q @ D[c0:c1].T                                    # sgemm
            kk = min(k, c1 - c0)
            if nq >= 2 * workers:
                step = -(-nq // workers)
                parts = list(pool.map(select, [(S, lo, min(nq, lo + step), kk) for lo in range(0, nq, step)]))
                part = np.concatenate(parts)
            else:
                part = select((S, 0, nq, kk))
            cand_s = np.take_along_axis(S, part, axis=1)
            best_s = np.concatenate([best_s, cand_s], axis=1)
            best_i = np.concatenate([best_i, part.astype(np.int64) + c0], axis=1)
            if best_s.shape[1] > 4 * k:                           # keep the running lists short
                keep = np.argpartition(-best_s, k - 1, axis=1)[:, :k]
                best_s, best_i = np.take_along_axis(best_s, keep, axis=1), np.take_along_axis(best_i, keep, axis=1)
    order = np.lexsort((best_i, -best_s.astype(np.float64)), axis=1)[:, :k]       # (score desc, id asc)
    return np.take_along_axis(best_i, order, axis=1), np.take_along_axis(best_s, order, axis=1), blas_threads()


def pad_queries(q: np.ndarray, d_pad: int) -> np.ndarray:
    q = np.ascontiguousarray(q, dtype=np.float32)
    out = np.zeros((q.shape[0], d_pad), dtype=np.float32)
    out[:, : q.shape[1]] = q
    return out


# ----------------------------------------------------------------------------------- flat search
def flat_search_f16(corpus_u16: np.ndarray, q32: np.ndarray, k: int, id_base: int = 0) -> Tuple[np.ndarray, np.ndarray, int]:
    """IndexFlatIP.search (VectorStore_Faiss.py:263) over fp16 rows with the canonical fp32 scorer.
    Ties: id ascending.  Returns (ids int64 [nq][k], scores fp32 [nq][k], threads used)."""
    corpus_u16 = np.ascontiguousarray(corpus_u16, dtype=np.uint16)
    n, d_pad = corpus_u16.shape
    q32 = np.ascontiguousarray(q32, dtype=np.float32)
    if q32.shape[1] != d_pad:
        q32 = pad_queries(q32, d_pad)
    nq = q32.shape[0]
    ids = np.full((nq, k), -1, dtype=np.int64)
    sc = np.full((nq, k), -np.inf, dtype=np.float32)
    L = lib()
    L.oracle_flat_search_f16.restype = ctypes.c_int
    nt = L.oracle_flat_search_f16(_p(corpus_u16), ctypes.c_int64(n), ctypes.c_int(d_pad), _p(q32),
                                  ctypes.c_int(nq), ctypes.c_int(k), ctypes.c_int64(id_base), _p(ids), _p(sc))
    return ids, sc, int(nt)


def flat_search_f64(corpus_u16: np.ndarray, q32: np.ndarray, k: int) -> Tuple[np.ndarray, np.ndarray]:
    """float64 numpy cross-check (small sizes): scores = q @ D^T in float64, (score desc, id asc)."""
    D = corpus_u16.view(np.float16).astype(np.float64)
    q = np.asarray(q32, dtype=np.float64)
    if q.shape[1] != D.shape[1]:
        q = np.pad(q, ((0, 0), (0, D.shape[1] - q.shape[1])))
    S = q @ D.T
    order = np.lexsort((np.arange(S.shape[1])[None, :].repeat(S.shape[0], 0), -S), axis=1)[:, :k]
    return order.astype(np.int64), np.take_along_axis(S, order, axis=1)


def score_rows_f16(corpus_u16: np.ndarray, qv: np.ndarray, rows: np.ndarray) -> np.ndarray:
    corpus_u16 = np.ascontiguousarray(corpus_u16, dtype=np.uint16)
    rows = np.ascontiguousarray(rows, dtype=np.int64)
    qv = np.ascontiguousarray(qv, dtype=np.float32)
    out = np.zeros(rows.shape[0], dtype=np.float32)
    lib().oracle_score_rows_f16(_p(corpus_u16), ctypes.c_int(corpus_u16.shape[1]), _p(qv), _p(rows),
                                ctypes.c_int(rows.shape[0]), _p(out))
    return out


def topk_merge(ids: np.ndarray, scores: np.ndarray, k: int) -> Tuple[np.ndarray, np.ndarray]:
    """Merge [G][nq][k] per-shard results into [nq][k]: score desc, id asc; id -1 entries ignored."""
    G, nq, kk = ids.shape
    I = np.transpose(ids, (1, 0, 2)).reshape(nq, G * kk)
    S = np.transpose(scores, (1, 0, 2)).reshape(nq, G * kk).astype(np.float32)
    out_i = np.full((nq, k), -1, dtype=np.int64)
    out_s = np.full((nq, k), -np.inf, dtype=np.float32)
    for q in range(nq):
        valid = I[q] >= 0
        ii, ss = I[q][valid], S[q][valid]
        order = np.lexsort((ii, -ss.astype(np.float64)))[:k]
        out_i[q, : order.size] = ii[order]
        out_s[q, : order.size] = ss[order]
    return out_i, out_s


# ------------------------------------------------------------------------------------- synthetic
def synth_ppnd(p: float) -> float:
    """The generator's inverse normal CDF (Wichura AS 241 in exactly rounded double operations)."""
    return float(lib().oracle_synth_ppnd(float(p)))


def synth_val(seed: int, row: int, col: int) -> int:
    """One N(0,1) draw of the counter-based generator as the integer rint(z * 2^20)."""
    return int(lib().oracle_synth_val(seed, row, col))


def synth_rows_f16(n: int, d: int, first_row: int = 0, seed: int = 1234, d_pad: int | None = None) -> np.ndarray:
    d_pad = d_pad or padded_dim(d)
    out = np.zeros((n, d_pad), dtype=np.uint16)
    if n:
        lib().oracle_synth_rows_f16(_p(out), ctypes.c_int(d_pad), ctypes.c_int(d), ctypes.c_int64(first_row),
                                    ctypes.c_int64(n), ctypes.c_uint64(seed))
    return out


def synth_rows_f32(n: int, d: int, first_row: int = 0, seed: int = 4321) -> np.ndarray:
    out = np.zeros((n, d), dtype=np.float32)
    if n:
        lib().oracle_synth_rows_f32(_p(out), ctypes.c_int64(d), ctypes.c_int(d), ctypes.c_int64(first_row),
                                    ctypes.c_int64(n), ctypes.c_uint64(seed))
    return out


# ---- the chunker's float64 cosines (core/file_management/chunker/spliter.py:307-371) ----
def _lane_sums(prod: np.ndarray) -> np.ndarray:
    """Sum the last axis in the order of rarc_cosine_pairs_kernel: 64 lanes, lane l adds elements l, l+64, ... in
    ascending order, then a butterfly over lane distances 32, 16, ..., 1 (lane 0's value)."""
    d = prod.shape[-1]
    pad = (-d) % 64
    if pad:
        prod = np.concatenate([prod, np.zeros(prod.shape[:-1] + (pad,), np.float64)], axis=-1)
    steps = prod.reshape(prod.shape[:-1] + (-1, 64))
    lanes = np.zeros(prod.shape[:-1] + (64,), np.float64)
    for m in range(steps.shape[-2]):
        lanes = lanes + steps[..., m, :]
    idx = np.arange(64)
    for off in (32, 16, 8, 4, 2, 1):
        lanes = lanes + lanes[..., idx ^ off]
    return lanes[..., 0]


def cosine_matrix_f64(x: np.ndarray, y: np.ndarray) -> np.ndarray:
    """spliter.cosine_similarity's numpy branch (:326-332) on fp32 rows: dot / (|x| |y|) in float64, NaN/inf -> 0."""
    x64, y64 = np.asarray(x, np.float32).astype(np.float64), np.asarray(y, np.float32).astype(np.float64)
    dot = _lane_sums(x64[:, None, :] * y64[None, :, :])
    nx, ny = _lane_sums(x64 * x64), _lane_sums(y64 * y64)
    with np.errstate(divide="ignore", invalid="ignore"):
        sim = dot / (np.sqrt(nx)[:, None] * np.sqrt(ny)[None, :])
    sim[~np.isfinite(sim)] = 0.0
    return sim


def adjacent_cosine_distances(x: np.ndarray) -> np.ndarray:
    """calculate_cosine_distances (:354-371): 1 - cosine(x[i], x[i+1])."""
    x64 = np.asarray(x, np.float32).astype(np.float64)
    if x64.shape[0] < 2:
        return np.zeros(0, np.float64)
    a, b = x64[:-1], x64[1:]
    with np.errstate(divide="ignore", invalid="ignore"):
        sim = _lane_sums(a * b) / (np.sqrt(_lane_sums(a * a)) * np.sqrt(_lane_sums(b * b)))
    sim[~np.isfinite(sim)] = 0.0
    return 1.0 - sim


# ------------------------------------------------------------------------------------------- RRF
def similar_pairs_f64(embeddings, threshold: float = 0.95) -> List[Tuple[int, int, float]]:
    """The entity de-duplication arithmetic of the reference's graph store (encapsulation/database/graph_db/Base_Neo4j.py:
    559-583): `similarity_matrix = cosine_similarity(np.array(embeddings))`, then `for i ... for j in range(i + 1, n): if
    similarity_matrix[i][j] >= 0.95` — returned as [(i, j, float(similarity))] in that loop's order.  cosine_similarity is
    scikit-learn's (a dependency of the reference, not vendored): rows divided by their Euclidean norms (a zero row stays
    zero), then X_n @ X_n.T, all in float64 — restated here; PINNED against sklearn.metrics.pairwise.cosine_similarity itself
    where that is installed (tests/test_similar_pairs_host.py) and through tests/golden/similar_pairs.json."""
    x = np.array(embeddings, dtype=np.float64)
    if x.ndim != 2 or x.shape[0] < 2:
        return []
    norms = np.sqrt(np.einsum("ij,ij->i", x, x))
    norms[norms == 0.0] = 1.0                       # sklearn.preprocessing.normalize: zero rows are left alone
    xn = x / norms[:, None]
    sim = xn @ xn.T
    out = []
    n = x.shape[0]
    for i in range(n):
        row = sim[i]
        for j in np.nonzero(row[i + 1:] >= threshold)[0]:
            out.append((i, int(i + 1 + j), float(row[i + 1 + j])))
    return out


def rrf_fuse(lists: Sequence[Sequence[Hashable]], rrf_k: float = 60.0, top_k: int = 10) -> List[Tuple[Hashable, float]]:
    """RRFusion.fuse (core/utils/Fusion.py:45-76) on bare keys (the reference keys on
    `document.content`): rank = position + 1 in every list; score[key] += 1.0 / (k + rank) in list
    order then position order (python float = fp64, first add is 0.0 + x); stable sort descending
    (ties keep first-insertion order); top_k.  Returns [(key, score)]."""
    scores = defaultdict(float)
    for one in lists:
        for i, key in enumerate(one):
            scores[key] += 1.0 / (rrf_k + (i + 1))
    ranked = sorted(scores.items(), key=lambda kv: kv[1], reverse=True)
    return ranked[:top_k]


def cosine_relevance(score: float) -> float:
    """VectorStoreBase._cosine_relevance_score_fn (VectorStoreBase.py:263-266): 1.0 - score, applied
    by the reference to what is already a similarity (quirk kept)."""
    return 1.0 - score


def max_inner_product_relevance(score: float) -> float:
    """VectorStoreBase._max_inner_product_relevance_score_fn (VectorStoreBase.py:268-273)."""
    if score > 0:
        return 1.0 - score
    return -1.0 * score


# --------------------------------------------------------------------------------------- rerank
def rerank_logsoftmax_yes_f16(z_no: np.ndarray, z_yes: np.ndarray) -> np.ndarray:
    """log_softmax([no, yes])[1] as an fp16 tensor (math in fp32): (yes - max) - log(sum exp(x - max))."""
    zn = np.asarray(z_no, dtype=np.float16).astype(np.float32)
    zy = np.asarray(z_yes, dtype=np.float16).astype(np.float32)
    m = np.maximum(zn, zy)
    s = np.exp(zn - m, dtype=np.float32) + np.exp(zy - m, dtype=np.float32)
    return ((zy - m) - np.log(s, dtype=np.float32)).astype(np.float16)


def rerank_scores_f16(z_no: np.ndarray, z_yes: np.ndarray) -> np.ndarray:
    """Reranker_Qwen3.py:41-49 on fp16 logits: exp(log_softmax([no, yes])[1]) where the
    log_softmax output and the exp output are fp16 tensors (math in fp32).  Returns float16.
    expf/logf are not bit-identical across libm / SLEEF / ocml, so a last-place difference in the
    fp16 log-softmax (amplified by exp) is the parity limit of this step; ORDER is what is pinned."""
    ls = rerank_logsoftmax_yes_f16(z_no, z_yes)
    return np.exp(ls.astype(np.float32), dtype=np.float32).astype(np.float16)


def stable_desc_order(scores: np.ndarray) -> np.ndarray:
    """Reranker_Qwen3.py:70-72: list.sort(key=score, reverse=True) — stable, ties keep input order."""
    s = np.asarray(scores, dtype=np.float64)
    return np.argsort(-s, kind="stable").astype(np.int32)


# --------------------------------------------------------------------------------------- encoder
def bert_forward_f32(sd, input_ids, lengths, num_heads, eps=1e-12, normalize=True, pooling="cls", dtype=np.float32):
    """BERT encoder forward in float32 numpy, CLS pooling (what SentenceTransformer.encode computes
    for bge-style models reached from huggingface.py:122-126).  `sd` = HuggingFace BertModel state
    dict (numpy arrays).  Pinned against transformers.BertModel in tests/test_oracle_golden.py.
    dtype=np.float64 runs the same graph in float64 on the same (fp32) weights: the yardstick two fp32-class
    forwards are measured against (tests of the fp32 encoder mode)."""
    from math import sqrt

    try:
        from scipy.special import erf
    except Exception:  # pragma: no cover
        import math
        erf = np.vectorize(math.erf)
    g = lambda k: np.asarray(sd[k], dtype=np.float32).astype(dtype)
    ids = np.asarray(input_ids)
    n, L = ids.shape
    lens = np.asarray(lengths)

    def ln(x, w, b):
        mu = x.mean(-1, keepdims=True)
        var = ((x - mu) ** 2).mean(-1, keepdims=True)
        return (x - mu) / np.sqrt(var + eps) * w + b

    x = g("embeddings.word_embeddings.weight")[ids] + g("embeddings.position_embeddings.weight")[None, :L] \
        + g("embeddings.token_type_embeddings.weight")[0]
    x = ln(x, g("embeddings.LayerNorm.weight"), g("embeddings.LayerNorm.bias"))
    H = x.shape[-1]
    dh = H // num_heads
    mask = np.where(np.arange(L)[None, :] < lens[:, None], 0.0, -np.inf).astype(dtype)   # [n][L] keys
    i = 0
    while f"encoder.layer.{i}.attention.self.query.weight" in sd:
        p = f"encoder.layer.{i}."
        lin = lambda t, name: t @ g(p + name + ".weight").T + g(p + name + ".bias")
        q = lin(x, "attention.self.query").reshape(n, L, num_heads, dh).transpose(0, 2, 1, 3)
        k = lin(x, "attention.self.key").reshape(n, L, num_heads, dh).transpose(0, 2, 1, 3)
        v = lin(x, "attention.self.value").reshape(n, L, num_heads, dh).transpose(0, 2, 1, 3)
        s = q @ k.transpose(0, 1, 3, 2) / sqrt(dh) + mask[:, None, None, :]
        s = s - s.max(-1, keepdims=True)
        pr = np.exp(s)
        pr = pr / pr.sum(-1, keepdims=True)
        ctx = (pr @ v).transpose(0, 2, 1, 3).reshape(n, L, H)
        x = ln(lin(ctx, "attention.output.dense") + x, g(p + "attention.output.LayerNorm.weight"),
               g(p + "attention.output.LayerNorm.bias"))
        h = lin(x, "intermediate.dense")
        h = 0.5 * h * (1.0 + erf(h / np.sqrt(2.0)))
        x = ln(lin(h.astype(dtype), "output.dense") + x, g(p + "output.LayerNorm.weight"), g(p + "output.LayerNorm.bias"))
        i += 1
    if pooling == "mean":   # sentence-transformers Pooling(mode_mean_tokens): mean over the attention-masked tokens
        m = (np.arange(L)[None, :] < lens[:, None]).astype(dtype)[:, :, None]
        cls = ((x * m).sum(1) / m.sum(1)).astype(dtype)
    else:
        cls = x[:, 0, :].astype(dtype)
    if dtype != np.float32:
        return cls / np.linalg.norm(cls, axis=1, keepdims=True) if normalize else cls
    return normalize_L2(cls) if normalize else cls


def mpnet_relative_position_bucket(relative_position, num_buckets=32, max_distance=128):
    """transformers 4.x MPNetEncoder.relative_position_bucket (models/mpnet/modeling_mpnet.py; third-party dependency of the
    reference's default checkpoint, sentence-transformers/all-mpnet-base-v2, huggingface.py:6), restated in numpy with its
    float32 arithmetic: relative_position = key - query; half the buckets per direction, exact below num_buckets/4, then
    logarithmic up to max_distance.  Pinned to the transformers function over every offset in tests/test_mpnet_oracle.py."""
    import math

    rp = np.asarray(relative_position, dtype=np.int64)
    n = -rp
    nb = num_buckets // 2
    ret = (n < 0).astype(np.int64) * nb
    n = np.abs(n)
    max_exact = nb // 2
    is_small = n < max_exact
    with np.errstate(divide="ignore"):
        val = np.log(n.astype(np.float32) / np.float32(max_exact)) / np.float32(math.log(max_distance / max_exact)) \
            * np.float32(nb - max_exact)
    val = np.where(np.isfinite(val), val, 0).astype(np.float32)
    val_if_large = np.minimum(max_exact + val.astype(np.int64), nb - 1)
    return ret + np.where(is_small, n, val_if_large)


def mpnet_rel_bias_table(rel_weight, span):
    """[heads][2*span - 1] table of MPNetEncoder.compute_position_bias: entry [h][key - query + span - 1] =
    relative_attention_bias.weight[bucket(key - query)][h] (the bias depends on key - query only)."""
    w = np.asarray(rel_weight, dtype=np.float32)                     # [num_buckets][heads]
    delta = np.arange(-(span - 1), span)
    return np.ascontiguousarray(w[mpnet_relative_position_bucket(delta, num_buckets=w.shape[0])].T)


def mpnet_forward_f32(sd, input_ids, lengths, num_heads, eps=1e-5, normalize=True, pooling="mean", dtype=np.float32):
    """MPNet encoder forward in numpy (transformers MPNetModel, the architecture of the reference's DEFAULT embedding model,
    huggingface.py:6), mean pooling over the real tokens + L2 normalisation as the checkpoint's sentence-transformers
    modules do.  `sd`: MPNetModel state dict (numpy arrays, no prefix).  Differences from BERT: no token types, position
    ids start at padding_idx + 1 = 2 for the real tokens (right-padded batches), separate q / k / v / o projections under
    attention.attn, and ONE relative-position bias (32 buckets x heads, shared by all layers) added to the scaled scores.
    Pinned against transformers.MPNetModel in tests/test_mpnet_oracle.py."""
    from math import sqrt

    from scipy.special import erf

    g = lambda k: np.asarray(sd[k], dtype=np.float32).astype(dtype)
    ids = np.asarray(input_ids)
    n, L = ids.shape
    lens = np.asarray(lengths)

    def ln(x, w, b):
        mu = x.mean(-1, keepdims=True)
        var = ((x - mu) ** 2).mean(-1, keepdims=True)
        return (x - mu) / np.sqrt(var + eps) * w + b

    x = g("embeddings.word_embeddings.weight")[ids] + g("embeddings.position_embeddings.weight")[None, 2:L + 2]
    x = ln(x, g("embeddings.LayerNorm.weight"), g("embeddings.LayerNorm.bias"))
    H = x.shape[-1]
    dh = H // num_heads
    mask = np.where(np.arange(L)[None, :] < lens[:, None], 0.0, -np.inf).astype(dtype)   # [n][L] keys
    tab = mpnet_rel_bias_table(sd["encoder.relative_attention_bias.weight"], L).astype(dtype)   # [heads][2L-1]
    idx = np.arange(L)[None, :] - np.arange(L)[:, None] + L - 1                                  # [query][key]
    bias = tab[:, idx]                                                                          # [heads][L][L]
    i = 0
    while f"encoder.layer.{i}.attention.attn.q.weight" in sd:
        p = f"encoder.layer.{i}."
        lin = lambda t, name: t @ g(p + name + ".weight").T + g(p + name + ".bias")
        q = lin(x, "attention.attn.q").reshape(n, L, num_heads, dh).transpose(0, 2, 1, 3)
        k = lin(x, "attention.attn.k").reshape(n, L, num_heads, dh).transpose(0, 2, 1, 3)
        v = lin(x, "attention.attn.v").reshape(n, L, num_heads, dh).transpose(0, 2, 1, 3)
        s = q @ k.transpose(0, 1, 3, 2) / sqrt(dh) + bias[None] + mask[:, None, None, :]
        s = s - s.max(-1, keepdims=True)
        pr = np.exp(s)
        pr = pr / pr.sum(-1, keepdims=True)
        ctx = (pr @ v).transpose(0, 2, 1, 3).reshape(n, L, H)
        x = ln(lin(ctx, "attention.attn.o") + x, g(p + "attention.LayerNorm.weight"), g(p + "attention.LayerNorm.bias"))
        h = lin(x, "intermediate.dense")
        h = 0.5 * h * (1.0 + erf(h / np.sqrt(2.0)))
        x = ln(lin(h.astype(dtype), "output.dense") + x, g(p + "output.LayerNorm.weight"), g(p + "output.LayerNorm.bias"))
        i += 1
    if pooling == "mean":
        m = (np.arange(L)[None, :] < lens[:, None]).astype(dtype)[:, :, None]
        out = ((x * m).sum(1) / m.sum(1)).astype(dtype)
    else:
        out = x[:, 0, :].astype(dtype)
    if dtype != np.float32:
        return out / np.linalg.norm(out, axis=1, keepdims=True) if normalize else out
    return normalize_L2(out) if normalize else out


def random_mpnet_state_dict(hidden, layers, heads, inter, vocab=1000, max_pos=130, seed=0, scale=0.05, buckets=32):
    """Seeded synthetic MPNetModel weights (HuggingFace names, no prefix)."""
    rng = np.random.default_rng(seed)
    r = lambda *shape: (rng.standard_normal(shape) * scale).astype(np.float32)
    sd = {"embeddings.word_embeddings.weight": r(vocab, hidden) * 4, "embeddings.position_embeddings.weight": r(max_pos, hidden),
          "embeddings.LayerNorm.weight": 1 + r(hidden), "embeddings.LayerNorm.bias": r(hidden),
          "encoder.relative_attention_bias.weight": r(buckets, heads) * 10}
    for i in range(layers):
        p = f"encoder.layer.{i}."
        for nm in ("q", "k", "v", "o"):
            sd[p + f"attention.attn.{nm}.weight"], sd[p + f"attention.attn.{nm}.bias"] = r(hidden, hidden), r(hidden)
        sd[p + "attention.LayerNorm.weight"], sd[p + "attention.LayerNorm.bias"] = 1 + r(hidden), r(hidden)
        sd[p + "intermediate.dense.weight"], sd[p + "intermediate.dense.bias"] = r(inter, hidden), r(inter)
        sd[p + "output.dense.weight"], sd[p + "output.dense.bias"] = r(hidden, inter), r(hidden)
        sd[p + "output.LayerNorm.weight"], sd[p + "output.LayerNorm.bias"] = 1 + r(hidden), r(hidden)
    return sd


def random_bert_state_dict(hidden, layers, heads, inter, vocab=1000, max_pos=128, seed=0, scale=0.05):
    """Seeded synthetic BertModel weights (HuggingFace names); LayerNorm weights near 1."""
    rng = np.random.default_rng(seed)
    r = lambda *shape: (rng.standard_normal(shape) * scale).astype(np.float32)
    sd = {"embeddings.word_embeddings.weight": r(vocab, hidden) * 4, "embeddings.position_embeddings.weight": r(max_pos, hidden),
          "embeddings.token_type_embeddings.weight": r(2, hidden),
          "embeddings.LayerNorm.weight": 1 + r(hidden), "embeddings.LayerNorm.bias": r(hidden)}
    for i in range(layers):
        p = f"encoder.layer.{i}."
        for nme in ("attention.self.query", "attention.self.key", "attention.self.value", "attention.output.dense"):
            sd[p + nme + ".weight"], sd[p + nme + ".bias"] = r(hidden, hidden), r(hidden)
        sd[p + "intermediate.dense.weight"], sd[p + "intermediate.dense.bias"] = r(inter, hidden), r(inter)
        sd[p + "output.dense.weight"], sd[p + "output.dense.bias"] = r(hidden, inter), r(hidden)
        for nme in ("attention.output.LayerNorm", "output.LayerNorm"):
            sd[p + nme + ".weight"], sd[p + nme + ".bias"] = 1 + r(hidden), r(hidden)
    return sd


# --------------------------------------------------------------------------------- reranker LM forward
def qwen3_last_logits_f32(sd, cfg, input_ids, attention_mask, token_ids):
    """Last-position logits of a Qwen3-style causal LM at `token_ids` (e.g. [no_id, yes_id]) in float32 numpy —
    what Qwen3Reranker.compute_logits reads from `self.lm(**inputs).logits[:, -1, :]`
    (core/rerank/Reranker_Qwen3.py:41-49) for a LEFT-padded batch (:29-39).  `sd`: HuggingFace Qwen3ForCausalLM
    state dict (numpy); `cfg`: dict(num_attention_heads, num_key_value_heads, head_dim, rms_norm_eps, rope_theta).
    Positions run 0..L-1 over the padded sequence (the reference's forward passes no position_ids).
    Pinned against transformers.Qwen3ForCausalLM in tests/test_oracle_golden.py."""
    g = lambda k: np.asarray(sd[k], dtype=np.float32)
    ids = np.asarray(input_ids)
    mask = np.asarray(attention_mask).astype(bool)
    n, L = ids.shape
    nq, nkv, dh = cfg["num_attention_heads"], cfg["num_key_value_heads"], cfg["head_dim"]
    eps, theta = float(cfg.get("rms_norm_eps", 1e-6)), float(cfg.get("rope_theta", 1e6))

    def rms(x, w):
        return x / np.sqrt((x.astype(np.float32) ** 2).mean(-1, keepdims=True) + eps) * w

    inv_freq = (1.0 / theta ** (np.arange(0, dh, 2, dtype=np.float32) / np.float32(dh))).astype(np.float32)
    ang = np.arange(L, dtype=np.float32)[:, None] * inv_freq[None, :]
    cos, sin = np.cos(np.concatenate([ang, ang], -1)), np.sin(np.concatenate([ang, ang], -1))     # [L][dh]

    def rope(x):  # x [n][heads][L][dh]; rotate_half convention
        x1, x2 = x[..., : dh // 2], x[..., dh // 2:]
        return x * cos + np.concatenate([-x2, x1], -1) * sin

    causal = np.tril(np.ones((L, L), dtype=bool))
    allow = causal[None, :, :] & mask[:, None, :]                     # [n][query][key]
    bias = np.where(allow, 0.0, -np.inf).astype(np.float32)
    x = g("model.embed_tokens.weight")[ids]
    i = 0
    while f"model.layers.{i}.self_attn.q_proj.weight" in sd:
        p = f"model.layers.{i}."
        h = rms(x, g(p + "input_layernorm.weight"))
        q = (h @ g(p + "self_attn.q_proj.weight").T).reshape(n, L, nq, dh)
        k = (h @ g(p + "self_attn.k_proj.weight").T).reshape(n, L, nkv, dh)
        v = (h @ g(p + "self_attn.v_proj.weight").T).reshape(n, L, nkv, dh)
        q = rope(rms(q, g(p + "self_attn.q_norm.weight")).transpose(0, 2, 1, 3))
        k = rope(rms(k, g(p + "self_attn.k_norm.weight")).transpose(0, 2, 1, 3))
        v = v.transpose(0, 2, 1, 3)
        rep = nq // nkv
        k, v = np.repeat(k, rep, axis=1), np.repeat(v, rep, axis=1)
        s = q @ k.transpose(0, 1, 3, 2) / np.sqrt(np.float32(dh)) + bias[:, None, :, :]
        s = s - np.where(np.isfinite(s.max(-1, keepdims=True)), s.max(-1, keepdims=True), 0.0)
        pr = np.exp(s)
        den = pr.sum(-1, keepdims=True)
        pr = np.where(den > 0, pr / np.where(den > 0, den, 1.0), 0.0)   # a padding query attends to nothing
        ctx = (pr @ v).transpose(0, 2, 1, 3).reshape(n, L, nq * dh)
        x = x + ctx @ g(p + "self_attn.o_proj.weight").T
        h = rms(x, g(p + "post_attention_layernorm.weight"))
        gate, up = h @ g(p + "mlp.gate_proj.weight").T, h @ g(p + "mlp.up_proj.weight").T
        x = x + ((gate / (1.0 + np.exp(-gate))) * up) @ g(p + "mlp.down_proj.weight").T
        i += 1
    last = rms(x[:, -1, :], g("model.norm.weight"))
    head = g("lm_head.weight") if "lm_head.weight" in sd else g("model.embed_tokens.weight")
    return (last @ head[np.asarray(token_ids)].T).astype(np.float32)


def random_qwen3_state_dict(hidden, layers, n_q, n_kv, head_dim, inter, vocab=1000, seed=0, scale=0.05, tie=True):
    """Seeded synthetic Qwen3ForCausalLM weights (HuggingFace names); norm weights near 1."""
    rng = np.random.default_rng(seed)
    r = lambda *shape: (rng.standard_normal(shape) * scale).astype(np.float32)
    sd = {"model.embed_tokens.weight": r(vocab, hidden) * 4, "model.norm.weight": 1 + r(hidden)}
    for i in range(layers):
        p = f"model.layers.{i}."
        sd[p + "self_attn.q_proj.weight"] = r(n_q * head_dim, hidden)
        sd[p + "self_attn.k_proj.weight"] = r(n_kv * head_dim, hidden)
        sd[p + "self_attn.v_proj.weight"] = r(n_kv * head_dim, hidden)
        sd[p + "self_attn.o_proj.weight"] = r(hidden, n_q * head_dim)
        sd[p + "self_attn.q_norm.weight"], sd[p + "self_attn.k_norm.weight"] = 1 + r(head_dim), 1 + r(head_dim)
        sd[p + "mlp.gate_proj.weight"], sd[p + "mlp.up_proj.weight"] = r(inter, hidden), r(inter, hidden)
        sd[p + "mlp.down_proj.weight"] = r(hidden, inter)
        sd[p + "input_layernorm.weight"], sd[p + "post_attention_layernorm.weight"] = 1 + r(hidden), 1 + r(hidden)
    sd["lm_head.weight"] = sd["model.embed_tokens.weight"] if tie else r(vocab, hidden)
    return sd
