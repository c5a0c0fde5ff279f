"""GPU parity: HIP flat search (through the C-ABI) vs the CPU oracle — ids AND scores bit-exact.

Reference path being replaced: FaissVectorStore.add_texts / similarity_search_by_vector_with_score
(encapsulation/database/vector_db/VectorStore_Faiss.py:170-202, :258-272).
"""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def hip():
    import torch

    assert torch.cuda.is_available(), "gpu tests need a ROCm device"
    from rag_arc_amd.hip import engine

    return engine


def _data(n, d, nq, seed):
    rng = np.random.default_rng(seed)
    return (rng.standard_normal((n, d)).astype(np.float32) * 3.0,
            rng.standard_normal((nq, d)).astype(np.float32) * 0.5)


def _check(hip, oracle, X, Q, k, metric="cosine", scan="q8"):
    n, d = X.shape
    idx = hip.FlatIndexF16(d, metric=metric, scan=scan)
    idx.add(X)
    # ingest parity: stored rows are bit-identical to the oracle's
    ref_rows, ref_n2 = oracle.ingest_f16(X, normalize=(metric == "cosine"))
    if n:
        got_rows = idx.rows.cpu().numpy().view(np.uint16)
        assert np.array_equal(got_rows, ref_rows)
    kk = min(k, n) if n else k
    D, I = idx.search(Q, kk) if n else (np.zeros((Q.shape[0], 0)), np.zeros((Q.shape[0], 0)))
    if n == 0:
        return idx
    qn = oracle.normalize_L2(Q) if metric == "cosine" else Q
    ref_I, ref_D, _ = oracle.flat_search_f16(ref_rows, qn, kk)
    assert np.array_equal(I, ref_I), f"ids differ (n={n} d={d} k={kk})"
    assert np.array_equal(D.view(np.uint32), ref_D.view(np.uint32)), "scores not bit-identical"
    # and within 1e-5 of float64 truth (north-star tolerance)
    i64, s64 = oracle.flat_search_f64(ref_rows, qn, kk)
    assert np.max(np.abs(s64 - D) / np.maximum(1.0, np.abs(s64))) < 1e-5
    return idx


@pytest.mark.parametrize("n,d,nq,k", [
    (1, 384, 1, 1), (31, 384, 3, 10), (32, 768, 5, 10), (33, 768, 256, 10), (100, 768, 7, 100),
    (1000, 100, 9, 10), (5000, 768, 256, 100), (4097, 384, 64, 50), (20000, 768, 300, 100),
])
def test_search_matches_oracle(hip, oracle, n, d, nq, k):
    X, Q = _data(n, d, nq, seed=n * 7 + d)
    _check(hip, oracle, X, Q, k)                    # int8 prefilter scan (default)
    _check(hip, oracle, X, Q, k, scan="mfma16")     # fp16 MFMA scan + certificate


@pytest.mark.parametrize("n,d,nq,k", [(3000, 1024, 40, 10), (20000, 1024, 256, 100), (777, 896, 5, 20)])
def test_search_wide_rows_q8(hip, oracle, n, d, nq, k):
    """d up to 1024 (bge-large, BASELINE config 5's dimension) runs on the int8-prefilter scan."""
    X, Q = _data(n, d, nq, seed=n + d)
    _check(hip, oracle, X, Q, k)


def test_q8_overflow_is_flagged_and_repaired(hip, oracle):
    """A candidate buffer far too small for the batch: every query overflows, is flagged and repaired."""
    X, Q = _data(50_000, 768, 16, seed=77)
    idx = hip.FlatIndexF16(768, cand_cap=512, scan="q8")       # x8 for k = 900: 16 slots per (workgroup, query)
    idx.add(X)
    D, I = idx.search(Q, 900)
    assert len(idx.last_repaired) == 16
    rows, _ = oracle.ingest_f16(X)
    rI, rD, _ = oracle.flat_search_f16(rows, oracle.normalize_L2(Q), 900)
    assert np.array_equal(I, rI) and np.array_equal(D.view(np.uint32), rD.view(np.uint32))


def test_inner_product_metric(hip, oracle):
    X, Q = _data(3000, 768, 17, seed=5)
    _check(hip, oracle, X, Q, 20, metric="ip")


def test_ties_break_by_id(hip, oracle):
    # 600 copies of 5 distinct rows: every score is tied 120 ways; order must be id ascending
    rng = np.random.default_rng(3)
    base = rng.standard_normal((5, 768)).astype(np.float32)
    X = np.tile(base, (600, 1))
    Q = rng.standard_normal((4, 768)).astype(np.float32)
    _check(hip, oracle, X, Q, 100)


def test_zero_rows_and_zero_query(hip, oracle):
    X, Q = _data(500, 384, 4, seed=11)
    X[7] = 0.0
    X[123] = 0.0
    Q[2] = 0.0
    _check(hip, oracle, X, Q, 10)


def test_medium_c2_shape(hip, oracle):
    """C2 shape at 1/5 scale: 200k x 768, 256 queries, k=100."""
    X, Q = _data(200_000, 768, 256, seed=42)
    idx = _check(hip, oracle, X, Q, 100)
    # the exactness certificate should hold for (nearly) all queries on random data
    assert len(getattr(idx, "last_repaired", [])) <= 2


def test_config2_full_size_auto_scan(hip, oracle):
    """BASELINE config 2 at its real size: 1M x 768 fp16, 256 queries, k = 100, scan="auto" (which picks the fp16
    MFMA scan at this shard size) — ids and scores bit-identical to the oracle (0.6 s of CPU on the GPU box)."""
    import torch
    from rag_arc_amd.hip import binding as B

    lib = B.load_library()
    n, d, nq, k = 1_000_000, 768, 256, 100
    rows = torch.empty((n, d), dtype=torch.float16, device="cuda")
    B.check(lib.rarc_synth_rows_f16(rows.data_ptr(), d, d, 0, n, 1234, 0))
    q = torch.empty((nq, d), dtype=torch.float32, device="cuda")
    B.check(lib.rarc_synth_rows_f32(q.data_ptr(), d, d, 0, nq, 4321, 0))
    idx = hip.FlatIndexF16(d, scan="auto")
    idx.add_rows_f16(rows, 1.001)
    assert not idx._use_q8(k)
    D, I = idx.search(q, k)
    ref_I, ref_D, _ = oracle.flat_search_f16(rows.cpu().numpy().view(np.uint16), oracle.normalize_L2(q.cpu().numpy()), k)
    assert np.array_equal(I, ref_I) and np.array_equal(D.view(np.uint32), ref_D.view(np.uint32))
    assert len(idx.last_repaired) <= 2
    idx.scan = "q8"                                      # and through the int8-prefilter scan at the same size
    D8, I8 = idx.search(q, k)
    assert np.array_equal(I8, ref_I) and np.array_equal(D8.view(np.uint32), ref_D.view(np.uint32))


def test_split_scan_large_shard(hip, oracle):
    """2.2M rows (>= 65536 tiles): the int8 scan runs as two launches around the exact mid-scan pass that
    tightens the thresholds (scan_q8.hip); ids and scores must still equal the oracle's (fp8 rows: test_gpu_fp8_corpus.py)."""
    import torch
    from rag_arc_amd.hip import binding as B

    lib = B.load_library()
    n, d, nq, k = 2_200_000, 128, 48, 50
    rows_h = oracle.synth_rows_f16(n, d)                       # unit rows, storage format
    q = oracle.synth_rows_f32(nq, d)
    idx = hip.FlatIndexF16(d, scan="q8")
    idx.load_rows(rows_h.view(np.float16), 1.001)
    D, I = idx.search(q, k)
    ref_I, ref_D, _ = oracle.flat_search_f16(rows_h, oracle.normalize_L2(q), k)
    assert np.array_equal(I, ref_I) and np.array_equal(D.view(np.uint32), ref_D.view(np.uint32))
    assert len(idx.last_repaired) == 0
    # same data ordered by similarity to query 0, best rows LAST, then best rows FIRST (everything the mid-scan
    # pass sees is below / above what follows)
    sims = (rows_h.view(np.float16)[:, :d].astype(np.float32) @ oracle.normalize_L2(q)[0])
    for order in (np.argsort(sims, kind="stable"), np.argsort(-sims, kind="stable")):
        r2 = np.ascontiguousarray(rows_h[order])
        idx2 = hip.FlatIndexF16(d, scan="q8")
        idx2.load_rows(r2.view(np.float16), 1.001)
        D2, I2 = idx2.search(q, k)
        w_I, w_D, _ = oracle.flat_search_f16(r2, oracle.normalize_L2(q), k)
        assert np.array_equal(I2, w_I) and np.array_equal(D2.view(np.uint32), w_D.view(np.uint32))
        del idx2
        torch.cuda.empty_cache()


def test_search_async_over_256_queries(hip, oracle):
    """The pipelined form takes any batch: 600 queries go out as three 256-query launches, one result."""
    import torch

    X, Q = _data(6000, 384, 600, seed=5)
    idx = hip.FlatIndexF16(384)
    idx.add(X)
    h = idx.search_async(torch.from_numpy(Q).cuda(), 10)
    ai, asc = h.result()
    si, ssc = idx.search_device(Q, 10)
    assert torch.equal(ai, si) and torch.equal(asc.view(torch.int32), ssc.view(torch.int32))
    rows, _ = oracle.ingest_f16(X)
    rI, rD, _ = oracle.flat_search_f16(rows, oracle.normalize_L2(Q), 10)
    assert np.array_equal(ai.cpu().numpy(), rI) and np.array_equal(asc.cpu().numpy().view(np.uint32), rD.view(np.uint32))
    assert idx.verify_query(Q, 599, ai, asc) == 0            # a row of the third launch


def test_incremental_add_and_growth(hip, oracle):
    X, Q = _data(3000, 768, 8, seed=9)
    idx = hip.FlatIndexF16(768)
    for s in range(0, 3000, 700):
        idx.add(X[s:s + 700])
    ref_rows, _ = oracle.ingest_f16(X)
    D, I = idx.search(Q, 10)
    ref_I, ref_D, _ = oracle.flat_search_f16(ref_rows, oracle.normalize_L2(Q), 10)
    assert np.array_equal(I, ref_I) and np.array_equal(D.view(np.uint32), ref_D.view(np.uint32))


def test_synth_generator_matches_oracle(hip, oracle):
    import torch
    from rag_arc_amd.hip import binding as B

    lib = B.load_library()
    out = torch.zeros((1000, 768), dtype=torch.float16, device="cuda")
    B.check(lib.rarc_synth_rows_f16(out.data_ptr(), 768, 768, 12345, 1000, 1234, 0))
    ref = oracle.synth_rows_f16(1000, 768, first_row=12345, seed=1234)
    assert np.array_equal(out.cpu().numpy().view(np.uint16), ref)
    out32 = torch.zeros((64, 384), dtype=torch.float32, device="cuda")
    B.check(lib.rarc_synth_rows_f32(out32.data_ptr(), 384, 384, 5, 64, 4321, 0))
    assert np.array_equal(out32.cpu().numpy(), oracle.synth_rows_f32(64, 384, first_row=5, seed=4321))


def test_verify_finds_nothing_on_exact_answer(hip, oracle):
    X, Q = _data(50_000, 768, 16, seed=77)
    idx = hip.FlatIndexF16(768)
    idx.add(X)
    ids, sc = idx.search_device(Q, 100)
    before = ids.clone()
    for qi in (0, 5, 15):
        assert idx.verify_query(Q, qi, ids, sc) == 0
    assert bool((ids == before).all())


def test_async_pipeline_gives_identical_results(hip, oracle):
    X, Q = _data(30_000, 768, 256, seed=21)
    idx = hip.FlatIndexF16(768)
    idx.add(X)
    want_i, want_s = idx.search_device(Q, 50)
    handles = [idx.search_async(Q[i * 64:(i + 1) * 64], 50) for i in range(4)]     # four batches in flight
    for i, h in enumerate(handles):
        ids, sc = h.result()
        assert bool((ids == want_i[i * 64:(i + 1) * 64]).all()) and bool((sc == want_s[i * 64:(i + 1) * 64]).all())
    # forced repair through the deferred path (fp16 scan: k' == k leaves the certificate no margin)
    idx = hip.FlatIndexF16(768, scan="mfma16")
    idx.add(X)
    idx.kprime_for = lambda k: k
    h1, h2 = idx.search_async(Q[:16], 50), idx.search_async(Q[16:32], 50)
    for h, lo in ((h1, 0), (h2, 16)):
        ids, sc = h.result()
        assert len(h.repaired) == 16
        assert bool((ids == want_i[lo:lo + 16]).all()) and bool((sc == want_s[lo:lo + 16]).all())


def test_l2norm_rows_kernel_matches_oracle(hip, oracle):
    """rarc_l2norm_rows_f32 == faiss.normalize_L2 restated (VectorStore_Faiss.py:150-154): bit-exact,
    zero rows untouched, in place and strided."""
    import torch
    from rag_arc_amd.hip import binding as B

    lib = B.load_library()
    rng = np.random.default_rng(8)
    for n, d in ((1, 8), (37, 100), (1000, 384), (300, 768)):
        x = (rng.standard_normal((n, d)) * rng.uniform(0.01, 50, (n, 1))).astype(np.float32)
        if n > 5:
            x[3] = 0.0
        want = oracle.normalize_L2(x)
        dx = torch.from_numpy(x).cuda()
        out = torch.empty_like(dx)
        B.check(lib.rarc_l2norm_rows_f32(dx.data_ptr(), d, out.data_ptr(), d, n, d, 0))
        assert np.array_equal(out.cpu().numpy().view(np.uint32), want.view(np.uint32))
        B.check(lib.rarc_l2norm_rows_f32(dx.data_ptr(), d, dx.data_ptr(), d, n, d, 0))          # in place
        assert np.array_equal(dx.cpu().numpy().view(np.uint32), want.view(np.uint32))
    wide = torch.from_numpy(rng.standard_normal((50, 200)).astype(np.float32)).cuda()           # ld > d
    out = torch.zeros((50, 128), dtype=torch.float32, device="cuda")
    B.check(lib.rarc_l2norm_rows_f32(wide.data_ptr(), 200, out.data_ptr(), 128, 50, 96, 0))
    assert np.array_equal(out.cpu().numpy()[:, :96].view(np.uint32),
                          oracle.normalize_L2(wide.cpu().numpy()[:, :96].copy()).view(np.uint32))
    assert not out.cpu().numpy()[:, 96:].any()


def test_neighbors_above_threshold_matches_numpy(hip, oracle):
    """All-pairs cosine with a cut-off (the reference's entity de-duplication pattern): every stored row
    scoring >= threshold for each query, and nothing else — including a cluster larger than the first k."""
    rng = np.random.default_rng(12)
    X = rng.standard_normal((3000, 256)).astype(np.float32)
    c = rng.standard_normal(256).astype(np.float32)
    X[100:300] = c + 0.05 * rng.standard_normal((200, 256)).astype(np.float32)     # 200 near-duplicates
    idx = hip.FlatIndexF16(256)
    idx.add(X)
    Q = np.stack([c, X[5], rng.standard_normal(256).astype(np.float32)])
    got = idx.neighbors_above(Q, 0.95, k_cap=64)                                    # cluster (200) > k_cap (64)
    rows, _ = oracle.ingest_f16(X)
    ref_i, ref_s, _ = oracle.flat_search_f16(rows, oracle.normalize_L2(Q), 900)
    for j in range(3):
        m = ref_s[j] >= 0.95
        assert np.array_equal(got[j][0], ref_i[j][m]) and np.array_equal(got[j][1].view(np.uint32), ref_s[j][m].view(np.uint32))
    assert len(got[0][0]) == 200 and len(got[1][0]) == 1 and len(got[2][0]) == 0


@pytest.mark.parametrize("n,d,nq,k", [(33, 768, 5, 10), (5000, 768, 256, 100), (20000, 1024, 64, 100), (4097, 200, 64, 50)])
def test_shadow_image_scan_matches_oracle(hip, oracle, n, d, nq, k):
    """shadow=True: the prefilter reads the int8 image written at ingest instead of converting fp16 rows on
    every search; answers are the same bits (appends in two steps exercise the image's growth)."""
    X, Q = _data(n, d, nq, seed=n + 3 * d)
    idx = hip.FlatIndexF16(d, shadow=True)
    idx.add(X[: n // 3])
    idx.add(X[n // 3:])
    D, I = idx.search(Q, min(k, n))
    rows, _ = oracle.ingest_f16(X, d_pad=idx.d_pad)
    rI, rD, _ = oracle.flat_search_f16(rows, oracle.normalize_L2(Q), min(k, n))
    assert np.array_equal(I, rI) and np.array_equal(D.view(np.uint32), rD.view(np.uint32))
    plain = hip.FlatIndexF16(d, scan="q8")
    plain.add(X)
    D2, I2 = plain.search(Q, min(k, n))
    assert np.array_equal(I, I2) and np.array_equal(D.view(np.uint32), D2.view(np.uint32))


def test_verification_reports_a_truncated_check_as_such(hip):
    """The pair check gives a thread 64 look-ups per query; a thread owns the rows congruent to it modulo 524,288.  A tiled
    corpus can put more than 64 of one query's answer rows on one thread: the check is then INCOMPLETE, and says so —
    it used to report a correct answer as wrong."""
    import torch

    from rag_arc_amd.hip import binding as B

    free, _ = torch.cuda.mem_get_info()
    if free < 16 * (1 << 30):
        pytest.skip("needs 16 GiB of free HBM")
    stride, d = 2048 * 256, 128
    n = 70 * stride + 1000
    rows = torch.zeros((((n + 31) // 32) * 32, d), dtype=torch.float16, device="cuda")
    rows[1::2, 1] = 1.0                                   # filler: unit rows orthogonal to the query (score 0)
    rows[0::2, 2] = 1.0
    special = torch.arange(70, device="cuda") * stride + 12345
    rows[special] = 0
    rows[special, 0] = 1.0                                # 70 copies of the query itself, all on thread 12345's rows
    idx = hip.FlatIndexF16(d, metric="cosine")
    idx.add_rows_f16(rows, 1.0, n_valid=n)
    q = torch.zeros((1, d), dtype=torch.float32, device="cuda")
    q[0, 0] = 1.0
    ids, sc = idx.search_device(q, 60)                    # 60 of the 70 tied rows, lowest ids first: 60 look-ups, complete
    assert ids[0].tolist() == special[:60].tolist() and bool((sc == 1.0).all())
    assert idx.verify_batch(q, ids, sc, detail=True) == (0, 0)
    ids, sc = idx.search_device(q, 70)                    # all 70: one thread meets 70 rows at or above the k-th entry
    assert ids[0].tolist() == special.tolist()
    with pytest.raises(B.RarcError, match="could not complete"):
        idx.verify_batch(q, ids, sc)
    del idx, rows
    torch.cuda.empty_cache()


@pytest.mark.parametrize("storage", ["f16", "f8", "f32"])
def test_batched_exact_verification(hip, oracle, storage):
    """rarc_verify_batch: an exact answer has no row beating its k-th entry, for every query of the batch at once; a
    corrupted answer (k-th entry swapped for the worst row of the shard) is caught, with the count an exhaustive
    scan would find."""
    import torch

    rng = np.random.default_rng(77)
    n, d, nq, k = 120_000, 384, 21, 30
    X = rng.standard_normal((n, d)).astype(np.float32)
    Q = rng.standard_normal((nq, d)).astype(np.float32)
    idx = hip.FlatIndexF16(d, storage=storage)
    idx.add(X)
    ids, sc = idx.search_device(torch.from_numpy(Q).cuda(), k)
    assert idx.verify_batch(Q, ids, sc) == 0
    assert idx.verify_batch(Q, ids, sc, which=[0, 1, 2, 9, 20]) == 0
    bad_i, bad_s = ids.clone(), sc.clone()
    worst_i, worst_s = idx.search_device(torch.from_numpy(-Q).cuda(), 1)      # the row most OPPOSITE to each query
    for qi in (3, 4, 17):
        bad_i[qi, k - 1] = worst_i[qi, 0]
        bad_s[qi, k - 1] = -worst_s[qi, 0]
    # the planted entry is the query's WORST row: every other row beats it, k - 1 of them already listed in the answer
    assert idx.verify_batch(Q, bad_i, bad_s, detail=True)[0] == 3 * (n - k)
    assert idx.verify_batch(Q, bad_i, bad_s, which=[4], detail=True)[0] == n - k and idx.verify_batch(Q, bad_i, bad_s, which=[5, 6]) == 0
    # every returned (id, score) pair is checked, not only the k-th: a score one ulp off, an id swapped for a
    # row that does not belong, and two entries exchanged (ids and scores no longer paired) are each reported
    for corrupt in ("score", "id", "swap"):
        c_i, c_s = ids.clone(), sc.clone()
        if corrupt == "score":
            c_s[2, 5] = torch.nextafter(c_s[2, 5], c_s[2, 5] + 1)
            c_s[7, 0] = torch.nextafter(c_s[7, 0], c_s[7, 0] - 1)
        elif corrupt == "id":
            c_i[2, 5] = worst_i[2, 0]
            c_i[7, 0] = worst_i[7, 0]
        else:
            c_i[2, 5], c_i[2, 6] = ids[2, 6], ids[2, 5]
            c_i[7, 0], c_i[7, 1] = ids[7, 1], ids[7, 0]
        beating, wrong = idx.verify_batch(Q, c_i, c_s, detail=True)
        assert wrong == (4 if corrupt == "swap" else 2), (corrupt, beating, wrong)
        assert idx.verify_batch(Q, c_i, c_s, which=[0, 1, 3]) == 0


def test_twin_search_context_two_streams(oracle):
    """twin(): a second search context over the same rows (own query block, workspace, side stream).  Alternating
    batches between index and twin gives the answers of the index alone; the twin is read-only and goes stale when
    the parent changes."""
    import torch

    from rag_arc_amd.hip import binding as B
    from rag_arc_amd.hip.engine import FlatIndexF16

    rng = np.random.default_rng(5)
    n, d, k = 70_000, 384, 20
    X = rng.standard_normal((n, d)).astype(np.float32)
    idx = FlatIndexF16(d)
    idx.add(X)
    tw = idx.twin()
    batches = [torch.from_numpy(rng.standard_normal((64 + 16 * i, d)).astype(np.float32)).cuda() for i in range(6)]
    want = [idx.search_device(q, k) for q in batches]
    pend = [(idx if i % 2 == 0 else tw).search_async(q, k) for i, q in enumerate(batches)]      # all six in flight
    for (wi, ws), h in zip(want, pend):
        gi, gs = h.result()
        assert torch.equal(gi, wi) and torch.equal(gs.view(torch.int32), ws.view(torch.int32))
    rows, _ = oracle.ingest_f16(X, normalize=True)
    oi, osc, _ = oracle.flat_search_f16(rows, oracle.normalize_L2(batches[1].cpu().numpy()), k)
    gi, gs = tw.search_device(batches[1], k)
    assert np.array_equal(gi.cpu().numpy(), oi) and np.array_equal(gs.cpu().numpy().view(np.uint32), osc.view(np.uint32))
    with pytest.raises(B.RarcError):
        tw.add(X[:8])
    idx.add(X[:40])
    with pytest.raises(B.RarcError):
        tw.search_async(batches[0], k)
    tw2 = idx.twin()
    assert tw2.ntotal == n + 40 and torch.equal(tw2.search_device(batches[0], k)[0], idx.search_device(batches[0], k)[0])
