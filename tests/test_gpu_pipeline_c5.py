"""BASELINE.json config 5 shape on one GPU, end to end on the device: online query embedding (encoder
forward on MFMA, bge-large geometry: hidden 1024, 16 heads, ffn 4096; two layers of seeded weights) ->
L2-normalise -> int8-prefilter scan over a 1024-d fp16 shard -> canonical rescore -> RRF with a
supplied lexical rank list.  Parity boundary for the search and fusion is the embedding the encoder
produced (the encoder itself is checked against its fp32 oracle in test_gpu_encoder.py): ids and
scores of the dense stage and the fused order must equal the CPU oracle run on those embeddings."""
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
    enc = HipBertEncoder(sd, num_heads=HEADS)
    rng = np.random.default_rng(5)
    tok = rng.integers(1, 1000, (NQ, L)).astype(np.int32)
    lens = rng.integers(4, L + 1, NQ).astype(np.int32)
    for r, l in enumerate(lens):
        tok[r, l:] = 0
    q = enc.forward(tok, lens, normalize=True)                            # fp32 [256][1024] on the device
    assert q.shape == (NQ, H) and bool(torch.isfinite(q).all())

    rows = torch.zeros((N, H), dtype=torch.float16, device="cuda")
    B.check(lib.rarc_synth_rows_f16(rows.data_ptr(), H, H, 0, N, 1234, 0))
    # plant each query's own direction into a few rows so that the top of the lists is not noise
    rows[torch.arange(NQ, device="cuda") * 7 + 11] = q.half()
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
