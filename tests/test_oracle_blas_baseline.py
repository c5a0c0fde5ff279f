"""The CPU baseline SURVEY.md §8(d) defines (numpy fp32 normalise -> sgemm in row chunks -> argpartition top-k:
oracle/cpu_ref.flat_search_blas_f32, what bench.py's cpu_baseline times) against the canonical-order oracle: the same answers
up to sgemm's summation order — scores within 1e-5 (north_star's tolerance; measured 2e-7), ids identical wherever the oracle's
neighbouring scores are further apart than twice the largest score difference."""
import numpy as np

from oracle import cpu_ref


def test_blas_baseline_agrees_with_the_canonical_oracle():
    rng = np.random.default_rng(7)
    for n, d, nq, k, chunk in ((30_000, 384, 33, 10, 4096), (50_000, 768, 64, 100, 131072), (300, 128, 5, 400, 128)):
        X = rng.standard_normal((n, d)).astype(np.float32)
        Q = rng.standard_normal((nq, d)).astype(np.float32)
        rows, _ = cpu_ref.ingest_f32(X)
        Qn = cpu_ref.normalize_L2(Q)
        ref_i, ref_s = cpu_ref.flat_search_f32(rows, Qn, min(k, n))[:2]
        got_i, got_s, threads = cpu_ref.flat_search_blas_f32(rows[:, :d], Qn, k, chunk_rows=chunk)
        assert threads >= 1 and got_i.shape == ref_i.shape == (nq, min(k, n))
        worst = float(np.abs(got_s - ref_s).max())
        assert worst <= 1e-5
        for q in range(nq):
            diff = np.nonzero(got_i[q] != ref_i[q])[0]
            for j in diff:       # a swap only between rows whose oracle scores are closer than the two paths' disagreement
                lo, hi = max(0, j - 1), min(ref_s.shape[1] - 1, j + 1)
                assert min(abs(ref_s[q, j] - ref_s[q, lo]) if lo != j else 1.0, abs(ref_s[q, j] - ref_s[q, hi]) if hi != j else 1.0) <= 2 * worst + 1e-7
            assert set(got_i[q].tolist()) == set(ref_i[q].tolist()) or len(diff) <= 2


def test_blas_baseline_single_query_and_small_k():
    rng = np.random.default_rng(8)
    X = cpu_ref.normalize_L2(rng.standard_normal((5000, 64)).astype(np.float32))
    q = cpu_ref.normalize_L2(rng.standard_normal((1, 64)).astype(np.float32))
    i, s, _ = cpu_ref.flat_search_blas_f32(X, q, 1)
    full = (X @ q[0])
    assert i[0, 0] == int(np.argmax(full)) and abs(float(s[0, 0]) - float(full.max())) < 1e-6
