"""BASELINE.json config 3 shape, end to end on the device and checked against the oracle:
dense top-100 (fp16 scan + canonical rescore) -> reranker score->order on seeded fp16 logits ->
RRF with a supplied lexical rank list (~30 % overlap), 256 queries per batch, ids bit-exact."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def test_dense_rerank_rrf_batch(oracle):
    import torch

    from rag_arc_amd.core.rerank import HipLogitReranker
    from rag_arc_amd.core.utils import HipRRFusion
    from rag_arc_amd.hip import binding as B
    from rag_arc_amd.hip.engine import FlatIndexF16

    lib = B.load_library()
    N, D, NQ, K = 200_000, 768, 256, 100
    rows = torch.zeros((N, D), dtype=torch.float16, device="cuda")
    B.check(lib.rarc_synth_rows_f16(rows.data_ptr(), D, D, 0, N, 1234, 0))
    q = torch.zeros((NQ, D), dtype=torch.float32, device="cuda")
    B.check(lib.rarc_synth_rows_f32(q.data_ptr(), D, D, 0, NQ, 4321, 0))
    idx = FlatIndexF16(D)
    idx.add_rows_f16(rows, 1.001)
    ids, _ = idx.search_device(q, K)                                      # [256][100] on device

    # reranker logits: seeded fp16 pairs per (query, doc) — no LM weights offline
    rng = np.random.default_rng(99)
    zn = (rng.standard_normal((NQ, K)) * 3).astype(np.float16)
    zy = (rng.standard_normal((NQ, K)) * 3).astype(np.float16)
    scores, perm = HipLogitReranker(lambda *_: None).score_order(zn, zy)
    reranked = torch.gather(ids, 1, perm.long())                          # dense list in reranked order

    # supplied lexical list: 30 dense hits + 70 other ids, shuffled (seed 777)
    r2 = np.random.default_rng(777)
    ids_h = ids.cpu().numpy()
    lex = np.zeros((NQ, K), np.int64)
    for b in range(NQ):
        over = r2.choice(ids_h[b], 30, replace=False)
        rest = r2.choice(np.setdiff1d(np.arange(N), ids_h[b])[:5000], 70, replace=False)
        row = np.concatenate([over, rest])
        r2.shuffle(row)
        lex[b] = row
    keys = torch.stack([reranked, torch.from_numpy(lex).cuda()], dim=1).contiguous()       # [256][2][100]
    lens = torch.full((NQ, 2), K, dtype=torch.int32, device="cuda")
    fk, fs, fn = HipRRFusion().fuse_ids(keys, lens, K)

    # ---- oracle pipeline on the host ----
    rows_h = rows.cpu().numpy().view(np.uint16)
    o_ids, _, _ = oracle.flat_search_f16(rows_h, oracle.normalize_L2(q.cpu().numpy()), K)
    assert np.array_equal(ids_h, o_ids)
    o_scores = oracle.rerank_scores_f16(zn, zy)
    got_scores = scores.cpu().numpy()
    fk, fs, fn = fk.cpu().numpy(), fs.cpu().numpy(), fn.cpu().numpy()
    for b in range(NQ):
        order = oracle.stable_desc_order(got_scores[b])                   # order of ITS fp16 scores (expf last place)
        assert perm[b].cpu().numpy().tolist() == order.tolist()
        want = oracle.rrf_fuse([o_ids[b][order].tolist(), lex[b].tolist()], 60.0, K)
        assert fn[b] == len(want)
        assert fk[b, : fn[b]].tolist() == [k for k, _ in want]
        assert fs[b, : fn[b]].tolist() == [s for _, s in want]
    assert (got_scores == o_scores).mean() > 0.98
