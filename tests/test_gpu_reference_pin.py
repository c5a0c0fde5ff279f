"""The HIP index against numbers the REFERENCE produced (tests/golden/cosine_pin_d*.npz: the float64 cosine matrix
of spliter.cosine_similarity, core/file_management/chunker/spliter.py:326-332, on 64 x 4096 fp16-representable
vectors).  The inputs are exact in every storage format, so there is one true answer for all of them:
  * metric "ip" over fp16 rows (both scan kernels): score / (|x| |y|) within 1e-5 of the reference's cosine, the
    reference's own ip ranking wherever its gaps exceed the tolerance;
  * metric "cosine" over fp32 rows (storage="f32", the reference's own storage): normalise + search as
    VectorStore_Faiss.py:150-154,258-263 does, scores within 1e-5 (measured: < 2e-6), the reference's top-100;
    and bit-identical to the oracle's fp32 search;
  * BASELINE config 1 (10k x 384 fp32, cosine top-10) through the fp32 storage."""
import numpy as np
import pytest

from tests.helpers import check_against_pin, pin_reference

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("d", [384, 768])
@pytest.mark.parametrize("scan", ["q8", "mfma16"])
def test_ip_over_fp16_rows_matches_reference_numbers(d, scan):
    from rag_arc_amd.hip.engine import FlatIndexF16

    X, Y, cos, xn, yn = pin_reference(d)
    idx = FlatIndexF16(d, metric="ip", scan=scan)
    idx.add(Y)                                                    # stored exactly: the values are fp16 numbers
    assert np.array_equal(idx.rows.cpu().numpy()[:, :d].astype(np.float32), Y)
    D, I = idx.search(X, 100)
    for b in range(X.shape[0]):
        got_cos = D[b].astype(np.float64) / (xn[b] * yn[I[b]])
        assert np.max(np.abs(got_cos - cos[b][I[b]])) < 1e-5
    scale = float(np.max(xn)) * float(np.max(yn))
    worst, n_set, n_ord = check_against_pin(cos * xn[:, None] * yn[None, :] / scale, I, D / np.float32(scale), tol=1e-5)
    assert n_set >= 50 and n_ord >= 10


@pytest.mark.parametrize("d", [384, 768])
def test_cosine_over_fp32_rows_matches_reference_numbers(oracle, d):
    from rag_arc_amd.hip.engine import FlatIndexF16

    X, Y, cos, _, _ = pin_reference(d)
    idx = FlatIndexF16(d, metric="cosine", storage="f32")
    idx.add(Y[:1500])                                             # two appends: the image and the bounds follow
    idx.add(Y[1500:])
    D, I = idx.search(X, 100)
    worst, n_set, n_ord = check_against_pin(cos, I, D, tol=1e-5)
    assert worst < 2e-6 and n_set >= 50 and n_ord >= 10
    rows32, _ = oracle.ingest_f32(Y, normalize=True)
    assert np.array_equal(idx.rows.cpu().numpy().view(np.uint32), rows32.view(np.uint32))   # stored rows bit-identical
    rI, rD, _ = oracle.flat_search_f32(rows32, oracle.normalize_L2(X), 100)
    assert np.array_equal(I, rI) and np.array_equal(D.view(np.uint32), rD.view(np.uint32))
    assert len(idx.last_repaired) == 0
    # every pair of the matrix, through k = 996 sweeps is overkill: the exact verify scan agrees for a few queries
    ids_d, sc_d = idx.search_device(X, 100)
    assert sum(idx.verify_query(X, b, ids_d, sc_d) for b in (0, 17, 63)) == 0


def test_config1_10k_x_384_fp32_cosine_top10(oracle):
    """BASELINE.json config 1's shape (the reference's CPU-runnable case) with the reference's storage: ids and
    scores bit-identical to the oracle, scores within 1e-5 of float64."""
    from rag_arc_amd.hip.engine import FlatIndexF16

    rng = np.random.default_rng(384)
    Xc = rng.standard_normal((10_000, 384)).astype(np.float32)
    Q = rng.standard_normal((32, 384)).astype(np.float32)
    idx = FlatIndexF16(384, metric="cosine", storage="f32")
    idx.add(Xc)
    D, I = idx.search(Q, 10)
    rows32, _ = oracle.ingest_f32(Xc, normalize=True)
    rI, rD, _ = oracle.flat_search_f32(rows32, oracle.normalize_L2(Q), 10)
    assert np.array_equal(I, rI) and np.array_equal(D.view(np.uint32), rD.view(np.uint32))
    c64 = (Q.astype(np.float64) / np.linalg.norm(Q.astype(np.float64), axis=1, keepdims=True)) @ \
          (Xc.astype(np.float64) / np.linalg.norm(Xc.astype(np.float64), axis=1, keepdims=True)).T
    for b in range(32):
        assert np.max(np.abs(c64[b][I[b]] - D[b])) < 1e-5
        assert set(I[b].tolist()) == set(np.argsort(-c64[b])[:10].tolist())


def test_fp32_storage_wide_rows_and_persistence(oracle, tmp_path):
    """1024-d fp32 rows (finalize stages 4 KiB rows with two waves), ip metric with mixed norms, save/load."""
    from rag_arc_amd.hip.engine import FlatIndexF16

    rng = np.random.default_rng(7)
    Y = rng.standard_normal((30_000, 1000)).astype(np.float32) * np.exp(rng.uniform(-2, 1, (30_000, 1))).astype(np.float32)
    Q = rng.standard_normal((9, 1000)).astype(np.float32)
    idx = FlatIndexF16(1000, metric="ip", storage="f32")
    idx.add(Y)
    D, I = idx.search(Q, 64)
    rows32, _ = oracle.ingest_f32(Y, normalize=False)
    rI, rD, _ = oracle.flat_search_f32(rows32, Q, 64)
    assert np.array_equal(I, rI) and np.array_equal(D.view(np.uint32), rD.view(np.uint32))
    # save / load: the fp32 rows come back bit for bit (their fp16 scan image is rebuilt on the device), same answer
    import torch

    path = str(tmp_path / "wide.rarc")
    st = idx.save_shard(path)
    assert st["bytes"] == 30_000 * 1024 * 4
    again = FlatIndexF16(1000, metric="ip", storage="f32")
    again.load_shard(path)
    assert again.ntotal == 30_000 and torch.equal(again.rows.view(torch.int32), idx.rows.view(torch.int32))
    D2, I2 = again.search(Q, 64)
    assert np.array_equal(I2, rI) and np.array_equal(D2.view(np.uint32), rD.view(np.uint32))
