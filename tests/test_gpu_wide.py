"""Arbitrary dimension and k: faiss.IndexFlatIP takes any d and any k
(VectorStore_Faiss.py:101-115, :262), the reference's OpenAI embeddings are 1536- / 3072-d
(encapsulation/llm/openai_llm.py:139-161).  Rows beyond 1024 padded dimensions and k beyond 1024 take the wide path
(csrc/wide.hip: score GEMM in chunks -> select against a rigorous threshold -> canonical finalize).

Parity: ids AND scores bit-exact against the oracle (the same canonical fp32 inner product, ties by id) at
d in {1536, 2048, 3072, 4096} and k in {1500, 5000} on >= 200k rows; small / ragged shapes; ties; near-duplicate clusters
(the candidate capacity grows); fp32 storage; inner-product metric; through the store."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def hip():
    import torch

    assert torch.cuda.is_available(), "gpu tests need a ROCm device"
    from rag_arc_amd.hip import engine

    return engine


def _oracle(oracle, X, Q, k, metric="cosine", storage="f16"):
    qn = oracle.normalize_L2(Q) if metric == "cosine" else Q
    if storage == "f32":
        rows = oracle.ingest_f32(X, normalize=(metric == "cosine"))[0]
        return oracle.flat_search_f32(rows, qn, k)[:2]
    rows, _ = oracle.ingest_f16(X, normalize=(metric == "cosine"))
    I, D, _ = oracle.flat_search_f16(rows, qn, k)
    return I, D


def _same(D, I, ref_I, ref_D):
    return np.array_equal(I, ref_I) and np.array_equal(D.view(np.uint32), ref_D.view(np.uint32))


@pytest.mark.parametrize("d,k", [(1536, 100), (2048, 1500), (3072, 10), (4096, 100), (1030, 7)])
def test_wide_rows_match_the_oracle_on_200k_rows(hip, oracle, d, k):
    rng = np.random.default_rng(d + k)
    n = 200_000 if d <= 2048 else 120_000
    X = rng.standard_normal((n, d)).astype(np.float32)
    Q = rng.standard_normal((40, d)).astype(np.float32)
    idx = hip.FlatIndexF16(d, metric="cosine")
    assert idx.wide
    for s0 in range(0, n, 50_000):
        idx.add(X[s0:s0 + 50_000])
    D, I = idx.search(Q, k)
    assert _same(D, I, *_oracle(oracle, X, Q, k)), (d, k)


@pytest.mark.parametrize("d,k", [(768, 1500), (384, 5000), (1536, 5000)])
def test_large_k_matches_the_oracle(hip, oracle, d, k):
    rng = np.random.default_rng(k + d)
    n = 200_000
    X = rng.standard_normal((n, d)).astype(np.float32)
    Q = rng.standard_normal((9, d)).astype(np.float32)
    idx = hip.FlatIndexF16(d, metric="cosine")
    idx.add(X)
    D, I = idx.search(Q, k)
    assert _same(D, I, *_oracle(oracle, X, Q, k)), (d, k)
    if d <= 1024:     # the same index still answers small k through the register-resident scans, and the two paths agree
        D2, I2 = idx.search(Q, 100)
        assert np.array_equal(I2, I[:, :100]) and np.array_equal(D2.view(np.uint32), D[:, :100].view(np.uint32))
    handle = idx.search_async(Q, k)
    Dh, Ih = handle.host()
    assert np.array_equal(Ih, I) and np.array_equal(Dh.view(np.uint32), D.view(np.uint32))


@pytest.mark.parametrize("n,d,nq,k", [(1, 1536, 1, 1), (127, 1152, 3, 127), (129, 2048, 256, 50), (2049, 1536, 300, 2049),
                                      (5000, 3072, 17, 1025)])
def test_small_and_ragged_shapes(hip, oracle, n, d, nq, k):
    rng = np.random.default_rng(n + d)
    X = (rng.standard_normal((n, d)) * 3).astype(np.float32)
    Q = rng.standard_normal((nq, d)).astype(np.float32)
    idx = hip.FlatIndexF16(d, metric="cosine")
    idx.add(X)
    D, I = idx.search(Q, k)
    assert _same(D, I, *_oracle(oracle, X, Q, k)), (n, d, nq, k)
    D5, I5 = idx.search(Q, min(n + 5, 8192))          # k > ntotal: the tail is (-inf, -1), like faiss
    assert np.array_equal(I5[:, :min(n, k)], I[:, :min(n, k)]) and (I5[:, n:] == -1).all() and np.isneginf(D5[:, n:]).all()


def test_ties_and_near_duplicate_clusters_grow_the_capacity(hip, oracle):
    """40,000 copies of 4 distinct rows + noise-level perturbations: every query's k-th best score is shared by thousands of
    rows (inside the error margin): the candidate lists fill up, the search is answered again with more room, and the order
    among equal scores is id ascending."""
    rng = np.random.default_rng(4)
    d = 1536
    base = rng.standard_normal((4, d)).astype(np.float32)
    X = np.repeat(base, 10_000, axis=0)
    X[::3] += (rng.standard_normal((len(X[::3]), d)) * 1e-4).astype(np.float32)
    rng.shuffle(X)
    Q = np.concatenate([base[:2], rng.standard_normal((3, d)).astype(np.float32)])
    idx = hip.FlatIndexF16(d, metric="cosine")
    idx.add(X)
    D, I = idx.search(Q, 300)
    assert _same(D, I, *_oracle(oracle, X, Q, 300))


def test_fp32_storage_and_inner_product(hip, oracle):
    rng = np.random.default_rng(8)
    n, d = 60_000, 1536
    X = (rng.standard_normal((n, d)) * np.exp(rng.standard_normal((n, 1)))).astype(np.float32)
    Q = rng.standard_normal((12, d)).astype(np.float32)
    a = hip.FlatIndexF16(d, metric="cosine", storage="f32")
    a.add(X)
    D, I = a.search(Q, 60)
    assert _same(D, I, *_oracle(oracle, X, Q, 60, storage="f32"))
    b = hip.FlatIndexF16(d, metric="ip")
    b.add(X)
    D, I = b.search(Q, 60)
    assert _same(D, I, *_oracle(oracle, X, Q, 60, metric="ip"))
    with pytest.raises(hip.B.RarcError):
        hip.FlatIndexF16(d, storage="f8")
    with pytest.raises(hip.B.RarcError):
        hip.FlatIndexF16(4100)
    with pytest.raises(hip.B.RarcError):
        a.search(Q, 9000)


def test_store_with_3072_dimensional_embeddings(hip, tmp_path):
    """The registered store over text-embedding-3-large-sized vectors: add, search, delete (compaction), save / load."""
    from rag_arc_amd.encapsulation.database.vector_db import HipFlatVectorStore
    from tests.helpers import HashEmbeddings

    emb = HashEmbeddings(3072)
    texts = [f"wide document {i}" for i in range(3000)]
    store = HipFlatVectorStore.from_texts(texts, emb, ids=[f"w{i}" for i in range(3000)])
    got = store.similarity_search_with_score(texts[1234], k=5)
    assert got[0][0].id == "w1234" and abs(got[0][1] - 1.0) < 1e-3
    assert store.delete(["w1234", "w0"]) is True
    assert store.similarity_search(texts[1234], k=1)[0].id != "w1234" and store.similarity_search(texts[2999], k=1)[0].id == "w2999"
    many = store.batch_similarity_search(texts[10:400], k=3)
    assert [m[0].id for m in many] == [f"w{i}" for i in range(10, 400)]
    store.save_local(str(tmp_path / "wide"))
    again = HipFlatVectorStore.load_local(str(tmp_path / "wide"), emb)
    assert [(d.id, s) for d, s in again.similarity_search_with_score(texts[77], k=9)] == \
        [(d.id, s) for d, s in store.similarity_search_with_score(texts[77], k=9)]


def test_what_the_backend_refuses_is_refused_loudly_and_before_any_launch():
    """faiss's IndexFlatIP takes any d and any k (VectorStore_Faiss.py:101-115, :262).  This backend answers d_pad <= 4096 and
    k <= 8192 on fp16 / fp32 rows, d_pad <= 1024 and k <= 1024 on fp8 rows; everything else is REFUSED — RarcUnsupported (a
    RarcError and a NotImplementedError), raised at construction or before the first kernel, the message naming the
    configuration that does answer ("What the backend refuses").  Never a truncated or approximate answer."""
    import torch

    from rag_arc_amd.hip import binding as B
    from rag_arc_amd.hip.engine import FlatIndexF16

    assert issubclass(B.RarcUnsupported, B.RarcError) and issubclass(B.RarcUnsupported, NotImplementedError)
    with pytest.raises(B.RarcUnsupported, match="wide path"):
        FlatIndexF16(1536, storage="f8")                              # fp8 rows wider than 1024 padded dimensions
    with pytest.raises(B.RarcUnsupported):
        FlatIndexF16(4200)                                            # beyond 4096 padded dimensions, any storage
    g = torch.Generator(device="cuda").manual_seed(3)
    x = torch.randn((9000, 256), device="cuda", generator=g)
    q = torch.randn((3, 256), device="cuda", generator=g)
    f8 = FlatIndexF16(256, storage="f8")
    f8.add(x)
    assert f8.search(q, 1000)[1].shape == (3, 1000)                   # the register-resident scan's whole range
    with pytest.raises(B.RarcUnsupported, match="'f16' or 'f32'"):
        f8.search(q, 2000)
    with pytest.raises(B.RarcUnsupported):
        f8.search_async(q, 2000)
    f16 = FlatIndexF16(256)
    f16.add(x)
    assert f16.search(q, 8192)[1].shape == (3, 8192)                  # the wide path's whole range
    with pytest.raises(B.RarcUnsupported, match="8192"):
        f16.search(q, 8500)


def test_inner_product_scores_beyond_the_fp16_range(hip, oracle):
    """metric "ip" takes rows and queries of any scale (faiss does), and the wide path stores its first chunk's
    scores as fp16.  Rows of norm ~60 against queries of norm ~2000 score up to 1e5 > 65504: an infinite first-chunk score
    would set an infinite threshold and drop every better row of the later chunks.  The engine scales such queries by a power
    of two on the way in and the scores back on the way out (both exact): ids and score bits are the oracle's — with the best
    rows planted in the LAST chunk."""
    rng = np.random.default_rng(99)
    n, d, k = 90_000, 1536, 50
    X = rng.standard_normal((n, d)).astype(np.float32) * 1.5               # ||row|| ~ 59
    Q = rng.standard_normal((7, d)).astype(np.float32) * 50.0              # ||q|| ~ 1960
    for j in range(7):                                                     # the best row of every query sits at the very end
        X[n - 1 - j] = Q[j] / np.linalg.norm(Q[j]) * 60.0
    idx = hip.FlatIndexF16(d, metric="ip")
    idx.add(X)
    D, I = idx.search(Q, k)
    ref_I, ref_D = _oracle(oracle, X, Q, k, metric="ip")
    assert float(ref_D.max()) > 65504.0 and np.isfinite(D).all()
    assert [int(I[j, 0]) for j in range(7)] == [n - 1 - j for j in range(7)]
    assert _same(D, I, ref_I, ref_D)
