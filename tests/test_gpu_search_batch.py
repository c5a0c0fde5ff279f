"""rarc_search_batch (ABI 600) called the way INTEGRATION.md's stub calls it — one foreign call for VectorStore_Faiss.py:258-263 —
against the two-call form (rarc_prep_queries + rarc_search_f16) on the same buffers: same ids, same score bits, same status
words; the status buffer needs no zeroing by the caller; the any-flag word arrives in pinned host memory (zero on a clean
batch, non-zero when a query overflowed); a gate event that has completed lets the scan through."""
import ctypes

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("scan,n", [("mfma16", 150_000), ("q8", 400_000)])
def test_one_call_equals_two_calls(scan, n):
    import torch

    from rag_arc_amd.hip import binding as B
    from rag_arc_amd.hip.engine import FlatIndexF16

    lib = B.load_library()
    d, nq, k = 384, 70, 50
    g = torch.Generator(device="cuda").manual_seed(n)
    idx = FlatIndexF16(d, scan=scan)
    idx.add(torch.randn((n, d), device="cuda", generator=g))
    q = torch.randn((nq, d), device="cuda", generator=g)
    ws = idx._workspace(k)
    qb = idx._qbuf["qblock"]
    use_q8 = idx._use_q8(k)
    qm = idx._qmeta.data_ptr() if use_q8 else 0
    kp = k if use_q8 else idx.kprime_for(k)
    st = torch.cuda.current_stream().cuda_stream
    # two calls, status zeroed by the caller
    ids2 = torch.empty((nq, k), dtype=torch.int64, device="cuda")
    sc2 = torch.empty((nq, k), dtype=torch.float32, device="cuda")
    status2 = torch.zeros(257, dtype=torch.int32, device="cuda")
    B.check(lib.rarc_prep_queries(q.data_ptr(), d, nq, d, idx.d_pad, 1, 1.0, qm, qb.data_ptr(), st))
    B.check(lib.rarc_search_f16(idx._rows.data_ptr(), idx.ntotal, idx.d_pad, qm, qb.data_ptr(), nq, k, kp, 0, -1.0, 1.0,
                                ids2.data_ptr(), sc2.data_ptr(), status2.data_ptr(), ws.data_ptr(), ws.numel(), idx._cap_eff, st))
    # one call: status full of garbage, flag word in pinned memory preset to garbage, a completed event as gate
    ids1, sc1 = torch.empty_like(ids2), torch.empty_like(sc2)
    status1 = torch.full((257,), -1, dtype=torch.int32, device="cuda")
    flag = torch.full((1,), 77, dtype=torch.int32).pin_memory()
    gate = torch.cuda.Event()
    gate.record()
    gate.synchronize()
    b = B.SearchBatch(idx._rows.data_ptr(), 0, 0, idx.ntotal, idx.d_pad, qm, q.data_ptr(), d, nq, d, 1, 1.0, qb.data_ptr(),
                      k, kp, 0, -1.0, 1.0, ids1.data_ptr(), sc1.data_ptr(), status1.data_ptr(), flag.data_ptr(),
                      ws.data_ptr(), ws.numel(), idx._cap_eff, gate.cuda_event)
    B.check(lib.rarc_search_batch(ctypes.byref(b), st), "rarc_search_batch")
    torch.cuda.synchronize()
    assert torch.equal(ids1, ids2) and torch.equal(sc1.view(torch.int32), sc2.view(torch.int32))
    assert torch.equal(status1[:nq], status2[:nq]) and int(status1[256]) == int(status2[256])
    assert int(flag[0]) == (1 if int(status2[256]) else 0)
    D, I = idx.search(q, k)                              # and both equal the engine's own answer (which repairs what is flagged)
    if int(status2[256]) == 0:
        assert np.array_equal(I, ids1.cpu().numpy())


def test_flag_word_reports_an_overflowing_query():
    """Forty thousand copies of one row: every copy ties at the top, the candidate lists overflow, the query is flagged — the
    pinned flag word says so, the per-query words name it, and the engine's repair still returns the exact answer."""
    import torch

    from rag_arc_amd.hip import binding as B
    from rag_arc_amd.hip.engine import FlatIndexF16

    lib = B.load_library()
    d, k = 256, 20
    g = torch.Generator(device="cuda").manual_seed(5)
    x = torch.randn((300_000, d), device="cuda", generator=g)
    x[100_000:140_000] = x[7]
    idx = FlatIndexF16(d, scan="q8", cand_cap=4096 * 4)
    idx.add(x)
    q = torch.cat([x[7:8] + 1e-3 * torch.randn((1, d), device="cuda", generator=g), torch.randn((3, d), device="cuda", generator=g)])
    ws, qb = idx._workspace(k), idx._qbuf["qblock"]
    ids, sc = torch.empty((4, k), dtype=torch.int64, device="cuda"), torch.empty((4, k), dtype=torch.float32, device="cuda")
    status = torch.empty(257, dtype=torch.int32, device="cuda")
    flag = torch.zeros(1, dtype=torch.int32).pin_memory()
    b = B.SearchBatch(idx._rows.data_ptr(), 0, 0, idx.ntotal, idx.d_pad, idx._qmeta.data_ptr(), q.data_ptr(), d, 4, d, 1, 1.0,
                      qb.data_ptr(), k, k, 0, -1.0, 1.0, ids.data_ptr(), sc.data_ptr(), status.data_ptr(), flag.data_ptr(),
                      ws.data_ptr(), ws.numel(), idx._cap_eff, None)
    B.check(lib.rarc_search_batch(ctypes.byref(b), torch.cuda.current_stream().cuda_stream), "rarc_search_batch")
    torch.cuda.synchronize()
    words = status[:4].cpu().tolist()
    assert (int(flag[0]) != 0) == any(words) == (int(status[256]) != 0)
    D, I = idx.search(q, k)                               # the engine repairs whatever was flagged
    assert set(I[0].tolist()) <= ({7} | set(range(100_000, 140_000))) and float(D[0, 0]) > 0.99
