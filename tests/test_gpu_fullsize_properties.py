"""Size-independent properties at sizes the CPU oracle cannot finish quickly (5M rows here; bench.py
runs the same verification at its full 100M size):
  * the exact repair scan finds no row beating the returned k-th entry (top-k is exact),
  * returned scores are the canonical scores of the returned rows (oracle re-scores just those rows),
  * rows are ordered (score desc, id asc) and unique,
  * splitting the corpus into shards + merge gives the identical answer (sharding invariance)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def test_five_million_rows(oracle):
    import torch

    from rag_arc_amd.hip import binding as B
    from rag_arc_amd.hip.engine import FlatIndexF16
    from rag_arc_amd.hip.sharded import ShardedFlatSearch, shard_range

    lib = B.load_library()
    N, D, NQ, K = 5_000_000, 768, 256, 100
    rows = torch.empty((N, D), dtype=torch.float16, device="cuda")
    B.check(lib.rarc_synth_rows_f16(rows.data_ptr(), D, D, 0, N, 1234, 0))
    q = torch.empty((NQ, D), dtype=torch.float32, device="cuda")
    B.check(lib.rarc_synth_rows_f32(q.data_ptr(), D, D, 0, NQ, 4321, 0))
    idx = FlatIndexF16(D)
    idx.add_rows_f16(rows, 1.001)
    ids, sc = idx.search_device(q, K)
    assert idx.last_repaired == []                                   # certificate held for every query
    I, S = ids.cpu().numpy(), sc.cpu().numpy()
    # ordering + uniqueness
    assert np.all(S[:, :-1] >= S[:, 1:])
    ties = S[:, :-1] == S[:, 1:]
    assert np.all(I[:, :-1][ties] < I[:, 1:][ties])
    assert all(len(set(r)) == K for r in I)
    # scores are the canonical scores of those rows
    qn = oracle.normalize_L2(q.cpu().numpy())
    for b in (0, 17, 255):
        sub = rows[torch.from_numpy(I[b]).cuda()].cpu().numpy().view(np.uint16)
        want = oracle.score_rows_f16(sub, oracle.pad_queries(qn[b:b + 1], D)[0], np.arange(K))
        assert np.array_equal(want.view(np.uint32), S[b].view(np.uint32))
    # exactness: nothing in 5M rows beats the k-th entry
    for b in (0, 17, 100, 255):
        assert idx.verify_query(q, b, ids, sc) == 0
    assert idx.verify_batch(q, ids, sc) == 0                          # ... for all 256 queries in one pass
    # sharding invariance: 3 shards with id_base + HIP merge == single shard
    parts_i, parts_s = [], []
    for g in range(3):
        lo, hi = shard_range(N, g, 3)
        sh = FlatIndexF16(D, id_base=lo)
        cap = ((hi - lo + 31) // 32) * 32
        buf = torch.zeros((cap, D), dtype=torch.float16, device="cuda")
        buf[: hi - lo].copy_(rows[lo:hi])
        sh.add_rows_f16(buf[: hi - lo] if cap == hi - lo else buf, 1.001)
        sh.ntotal = hi - lo
        i, s = sh.search_device(q, K)
        parts_i.append(i)
        parts_s.append(s)
    m = ShardedFlatSearch.__new__(ShardedFlatSearch)
    m.torch = torch
    mi, ms = m._hip_merge(torch.stack(parts_i), torch.stack(parts_s), K)
    assert torch.equal(mi, ids) and torch.equal(ms.view(torch.int32), sc.view(torch.int32))


@pytest.mark.parametrize("storage,N,D", [("f8", 5_000_000, 1024), ("f32", 1_000_000, 384)])
def test_full_size_properties_other_storage_formats(oracle, storage, N, D):
    """The same size-independent properties for fp8 (BASELINE config 5's storage) and fp32 (the reference's own) rows:
    exact-rescan finds nothing better, order and uniqueness, scores = canonical scores of the returned rows, and
    sharding + merge invariance."""
    import torch

    from rag_arc_amd.hip import binding as B
    from rag_arc_amd.hip.engine import FlatIndexF16
    from rag_arc_amd.hip.sharded import ShardedFlatSearch, shard_range

    lib = B.load_library()
    NQ, K = 256, 100
    q = torch.empty((NQ, D), dtype=torch.float32, device="cuda")
    B.check(lib.rarc_synth_rows_f32(q.data_ptr(), D, D, 0, NQ, 4321, 0))

    def build(lo, hi):
        idx = FlatIndexF16(D, storage=storage, id_base=lo, capacity=hi - lo)
        buf = torch.empty((1 << 20, D), dtype=torch.float32, device="cuda")
        for s0 in range(lo, hi, 1 << 20):
            m = min(1 << 20, hi - s0)
            B.check(lib.rarc_synth_rows_f32(buf.data_ptr(), D, D, s0, m, 1234, 0))
            idx.add(buf[:m])
        return idx

    idx = build(0, N)
    ids, sc = idx.search_device(q, K)
    assert idx.last_repaired == []
    I, S = ids.cpu().numpy(), sc.cpu().numpy()
    assert np.all(S[:, :-1] >= S[:, 1:])
    ties = S[:, :-1] == S[:, 1:]
    assert np.all(I[:, :-1][ties] < I[:, 1:][ties])
    assert all(len(set(r)) == K for r in I)
    qn = oracle.normalize_L2(q.cpu().numpy())
    for b in (0, 131, 255):
        sel = torch.from_numpy(I[b]).cuda()
        if storage == "f8":
            sub, scl = idx.rows[sel].cpu().numpy(), idx.row_scales[sel].cpu().numpy()
            o_i, o_s, _ = oracle.flat_search_f8(sub, scl, qn[b:b + 1], K)
            assert np.array_equal(np.sort(o_s[0])[::-1].view(np.uint32), S[b].view(np.uint32))
        else:
            sub = idx.rows[sel].cpu().numpy()
            o_i, o_s, _ = oracle.flat_search_f32(sub, qn[b:b + 1], K)
            assert np.array_equal(o_s[0].view(np.uint32), S[b].view(np.uint32))
    for b in (0, 77, 255):
        assert idx.verify_query(q, b, ids, sc) == 0
    assert idx.verify_batch(q, ids, sc) == 0
    del idx
    torch.cuda.empty_cache()
    parts_i, parts_s = [], []
    for g in range(3):
        lo, hi = shard_range(N, g, 3)
        sh = build(lo, hi)
        i, s = sh.search_device(q, K)
        parts_i.append(i)
        parts_s.append(s)
        del sh
    m = ShardedFlatSearch.__new__(ShardedFlatSearch)
    m.torch = torch
    mi, ms = m._hip_merge(torch.stack(parts_i), torch.stack(parts_s), K)
    assert torch.equal(mi, ids) and torch.equal(ms.view(torch.int32), sc.view(torch.int32))
