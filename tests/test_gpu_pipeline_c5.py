"""BASELINE.json config 5 shape on one GPU, end to end on the device: online query embedding (encoder
forward on MFMA, bge-large geometry: hidden 1024, 16 heads, ffn 4096; two layers of seeded weights) ->
L2-normalise -> int8-prefilter scan over a 1024-d fp16 shard -> canonical rescore -> RRF with a
supplied lexical rank list.

Two parity boundaries (round 3):
  * FROM TOKEN IDS — the oracle runs the whole path on the host (numpy fp32 BERT forward -> search -> RRF) and
    the encoder runs at the reference's precision (precision="fp32"): embeddings within 1e-5 (L2), every returned
    score within 1e-5 of the oracle's score of that row, and ids / fused order IDENTICAL on every query whose
    oracle gaps exceed 2e-6, and on EVERY query the device's i-th row is one the oracle scores within 2e-6 of its own
    i-th row (two fp32 forwards cannot agree on the order of rows closer than their own rounding noise);
  * GIVEN THE EMBEDDINGS — ids, scores and the fused order equal the oracle run on the device's embeddings, bit for
    bit, for all 256 queries."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def test_encoder_scan_rrf_1024d(oracle):
    import torch

    from rag_arc_amd.core.utils import HipRRFusion
    from rag_arc_amd.encapsulation.embeddings.hip_bert import HipBertEncoder
    from rag_arc_amd.hip import binding as B
    from rag_arc_amd.hip.engine import FlatIndexF16

    lib = B.load_library()
    H, HEADS, FFN, NQ, L, N, K = 1024, 16, 4096, 256, 32, 300_000, 100
    sd = oracle.random_bert_state_dict(H, 2, HEADS, FFN, vocab=1000, max_pos=64, seed=5)
    enc = HipBertEncoder(sd, num_heads=HEADS, precision="fp32")
    rng = np.random.default_rng(5)
    tok = rng.integers(1, 1000, (NQ, L)).astype(np.int32)
    lens = rng.integers(4, L + 1, NQ).astype(np.int32)
    for r, l in enumerate(lens):
        tok[r, l:] = 0
    q = enc.forward(tok, lens, normalize=True)                            # fp32 [256][1024] on the device
    assert q.shape == (NQ, H) and bool(torch.isfinite(q).all())
    o_emb = oracle.bert_forward_f32(sd, tok, lens, HEADS, normalize=True)  # the host's fp32 forward from the same token ids
    d_emb = np.linalg.norm(q.cpu().numpy().astype(np.float64) - o_emb.astype(np.float64), axis=1)
    assert d_emb.max() <= 1e-5, d_emb.max()

    rows = torch.zeros((N, H), dtype=torch.float16, device="cuda")
    B.check(lib.rarc_synth_rows_f16(rows.data_ptr(), H, H, 0, N, 1234, 0))
    # plant each query's own direction (the ORACLE's embedding: an input both sides share) into a few rows so that
    # the top of the lists is not noise
    rows[torch.arange(NQ, device="cuda") * 7 + 11] = torch.from_numpy(o_emb).cuda().half()
    idx = FlatIndexF16(H)                                                 # d_pad 1024 -> int8-prefilter scan
    assert idx._use_q8()
    idx.add_rows_f16(rows, 1.001)
    ids, sc = idx.search_device(q, K)

    # the same corpus in config 5's storage format (fp8 e4m3fn + per-row scale), against its own oracle
    idx8 = FlatIndexF16(H, storage="f8")
    idx8.add(rows.float())
    ids8, sc8 = idx8.search_device(q, K)
    b8, s8, _ = oracle.ingest_f8(rows.float().cpu().numpy())
    o8_ids, o8_sc, _ = oracle.flat_search_f8(b8, s8, oracle.normalize_L2(q.cpu().numpy()), K)
    assert np.array_equal(ids8.cpu().numpy(), o8_ids)
    assert np.array_equal(sc8.cpu().numpy().view(np.uint32), o8_sc.view(np.uint32))
    assert (o8_ids[:, 0] == np.arange(NQ) * 7 + 11).all()

    q_h = q.cpu().numpy()
    rows_h = rows.cpu().numpy().view(np.uint16)
    o_ids, o_sc, _ = oracle.flat_search_f16(rows_h, oracle.normalize_L2(q_h), K)
    assert np.array_equal(ids.cpu().numpy(), o_ids)
    assert np.array_equal(sc.cpu().numpy().view(np.uint32), o_sc.view(np.uint32))
    assert (o_ids[:, 0] == np.arange(NQ) * 7 + 11).all()                  # the planted rows come first

    # ---- from token ids: the oracle's own embeddings through the oracle's search (fp16 and fp8 storage) ----
    TOL, GAP = 1e-5, 2e-6      # score tolerance (north star); rows the oracle separates by more than GAP keep their order
    t_ids, t_sc, _ = oracle.flat_search_f16(rows_h, oracle.normalize_L2(o_emb), K + 8)
    t8_ids, t8_sc, _ = oracle.flat_search_f8(b8, s8, oracle.normalize_L2(o_emb), K + 8)
    same16 = np.zeros(NQ, bool)
    for name, g_ids, g_sc, w_ids, w_sc in (("f16", ids.cpu().numpy(), sc.cpu().numpy(), t_ids, t_sc),
                                           ("f8", ids8.cpu().numpy(), sc8.cpu().numpy(), t8_ids, t8_sc)):
        same = safe = 0
        worst = drift = 0.0
        for b in range(NQ):
            pos = {int(r): j for j, r in enumerate(w_ids[b])}
            for i in range(K):
                j = pos.get(int(g_ids[b][i]))
                assert j is not None, f"{name} query {b}: row {g_ids[b][i]} is not among the oracle's top-{K + 8}"
                worst = max(worst, abs(float(g_sc[b][i]) - float(w_sc[b][j])))
                # the device's i-th row is the oracle's i-th row, or one the oracle scores within GAP of it
                drift = max(drift, abs(float(w_sc[b][j]) - float(w_sc[b][i])))
            identical = g_ids[b].tolist() == w_ids[b][:K].tolist()
            same += identical
            if np.min(w_sc[b][:K] - w_sc[b][1:K + 1]) > GAP:                # every gap of the oracle's top-101 is resolvable
                safe += 1
                assert identical, f"{name} query {b}: ids differ from the oracle run from token ids"
            if name == "f16":
                same16[b] = identical
        print(f"C5-FROM-TOKENS {name}: max |score - oracle score| {worst:.2e}; order drift {drift:.2e}; ids identical on {same}/{NQ} "
              f"queries, on all {safe} whose oracle gaps exceed {GAP:g}")
        assert worst <= TOL and drift <= GAP and safe >= NQ // 4

    r2 = np.random.default_rng(777)
    lex = np.zeros((NQ, K), np.int64)
    for b in range(NQ):
        over = r2.choice(o_ids[b], 30, replace=False)
        rest = r2.choice(np.setdiff1d(np.arange(5000), o_ids[b]), 70, replace=False)
        row = np.concatenate([over, rest])
        r2.shuffle(row)
        lex[b] = row
    keys = torch.stack([ids, torch.from_numpy(lex).cuda()], dim=1).contiguous()            # [256][2][100]
    lens2 = torch.full((NQ, 2), K, dtype=torch.int32, device="cuda")
    fk, fs, fn = (x.cpu().numpy() for x in HipRRFusion().fuse_ids(keys, lens2, K))
    for b in range(NQ):
        want = oracle.rrf_fuse([o_ids[b].tolist(), lex[b].tolist()], 60.0, K)
        assert fn[b] == len(want)
        assert fk[b, : fn[b]].tolist() == [k for k, _ in want]
        assert fs[b, : fn[b]].tolist() == [s for _, s in want]
        if same16[b]:      # end to end from token ids: the oracle's dense list is the same list, so is the fused one
            assert [k for k, _ in oracle.rrf_fuse([t_ids[b][:K].tolist(), lex[b].tolist()], 60.0, K)] == fk[b, : fn[b]].tolist()
