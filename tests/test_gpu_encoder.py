"""GPU parity of the encoder forward (csrc/encoder.hip, through the C-ABI) vs the float32 oracle
(oracle.bert_forward_f32, itself pinned against transformers.BertModel).

Tolerance (floating point, stated here as the contract): weights, activations and GEMM operands are
fp16 with fp32 accumulation, so the L2-normalised embedding is compared with
  max |Δ| <= 4e-3  and  cosine(hip, oracle) >= 0.9995   per sequence.
"""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _check(oracle, H, layers, heads, I, n_seq, L, seed):
    from rag_arc_amd.encapsulation.embeddings.hip_bert import HipBertEncoder

    sd = oracle.random_bert_state_dict(H, layers, heads, I, vocab=800, max_pos=256, seed=seed)
    enc = HipBertEncoder(sd, num_heads=heads, precision="fp16")
    rng = np.random.default_rng(seed)
    ids = rng.integers(1, 800, (n_seq, L)).astype(np.int32)
    lens = rng.integers(1, L + 1, n_seq).astype(np.int32)
    lens[0] = L
    for r, l in enumerate(lens):
        ids[r, l:] = 0
    got = enc.forward(ids, lens, normalize=True).cpu().numpy()
    # the oracle sees the same fp16-rounded weights (storage format), computes in fp32
    sd16 = {k: v.astype(np.float16).astype(np.float32) for k, v in sd.items()}
    want = oracle.bert_forward_f32(sd16, ids, lens, heads, normalize=True)
    assert got.shape == want.shape
    assert np.max(np.abs(got - want)) <= 4e-3
    cos = np.sum(got * want, axis=1) / (np.linalg.norm(got, axis=1) * np.linalg.norm(want, axis=1))
    assert cos.min() >= 0.9995


@pytest.mark.parametrize("H,layers,heads,I,n_seq,L", [
    (128, 1, 2, 256, 3, 8),        # head_dim 64, tiny
    (128, 2, 4, 256, 5, 24),       # head_dim 32 (bge-small style heads)
    (384, 2, 12, 1536, 9, 32),     # bge-small shape, 2 layers
    (256, 1, 4, 512, 2, 100),      # keys span four 32-key tiles, ragged last query block
    (128, 1, 4, 256, 6, 200),      # head_dim 32, seven query blocks x seven key tiles, ragged lengths
    (256, 1, 4, 512, 3, 65),       # head_dim 64, one key past a tile boundary
    (256, 1, 4, 2048, 4, 32),      # FFN2 runs split-K x4 (fp32 partials summed by the LayerNorm kernel)
    (768, 1, 12, 3072, 4, 32),     # bge-base layer: LayerNorm rows span both 8-column chunks of a lane
    (1024, 1, 16, 4096, 8, 16),    # bge-large layer: out-proj split x2, FFN2 split x4
])
def test_encoder_matches_oracle(oracle, H, layers, heads, I, n_seq, L):
    _check(oracle, H, layers, heads, I, n_seq, L, seed=H + L)


@pytest.mark.parametrize("name,H,layers,heads,I", [("bge-base", 768, 12, 12, 3072), ("bge-large", 1024, 24, 16, 4096)])
def test_full_depth_encoder_and_induced_score_error(oracle, name, H, layers, heads, I):
    """Full depth (12 / 24 layers of seeded weights) against the fp32 oracle: how far the fp16 forward drifts with
    depth, and what that does to the cosine scores of a 100k-row scan.  |score(hip emb, d) - score(oracle emb, d)| <=
    ||hip emb - oracle emb|| for unit rows, so the embedding distance IS the end-to-end score-error bound; the test
    also measures the actual score differences over the top-100 of every query.  The figures are printed (and
    quoted in DESIGN.md / INTEGRATION.md as the encoder's contract): the 1e-5 of the north star holds for the
    search given the embeddings; the fp16 encoder itself is a 1e-3-class approximation of an fp32 one."""
    from rag_arc_amd.encapsulation.embeddings.hip_bert import HipBertEncoder
    from rag_arc_amd.hip.engine import FlatIndexF16

    n_seq, L = 16, 32
    sd = oracle.random_bert_state_dict(H, layers, heads, I, vocab=800, max_pos=64, seed=layers)
    enc = HipBertEncoder(sd, num_heads=heads, precision="fp16")
    rng = np.random.default_rng(layers)
    ids = rng.integers(1, 800, (n_seq, L)).astype(np.int32)
    lens = rng.integers(4, L + 1, n_seq).astype(np.int32)
    for r, l in enumerate(lens):
        ids[r, l:] = 0
    got = enc.forward(ids, lens, normalize=True).cpu().numpy()
    sd16 = {k: v.astype(np.float16).astype(np.float32) for k, v in sd.items()}
    want = oracle.bert_forward_f32(sd16, ids, lens, heads, normalize=True)
    max_abs = float(np.max(np.abs(got - want)))
    dist = float(np.max(np.linalg.norm(got.astype(np.float64) - want.astype(np.float64), axis=1)))
    cos = np.sum(got * want, axis=1) / (np.linalg.norm(got, axis=1) * np.linalg.norm(want, axis=1))
    # induced error on a scan: same 100k unit rows searched with both sets of embeddings
    idx = FlatIndexF16(H, metric="cosine")
    idx.add(oracle.synth_rows_f32(100_000, H, seed=1234))
    Dg, Ig = idx.search(got, 100)
    Dw, Iw = idx.search(want, 100)
    recall = float(np.mean([len(np.intersect1d(Ig[b], Iw[b])) / 100.0 for b in range(n_seq)]))
    sdiff = 0.0
    for b in range(n_seq):
        common, ig, iw = np.intersect1d(Ig[b], Iw[b], return_indices=True)
        sdiff = max(sdiff, float(np.max(np.abs(Dg[b][ig] - Dw[b][iw]))))
    print(f"ENCODER-DEPTH {name}: layers={layers} max|d_emb|={max_abs:.2e} max||d_emb||2={dist:.2e} "
          f"min cos={cos.min():.6f} max|d_score| (top-100, 100k rows)={sdiff:.2e} recall@100 vs fp32 embeddings={recall:.4f}")
    assert max_abs <= 4e-3 and cos.min() >= 0.9995
    assert sdiff <= dist + 1e-6                      # Cauchy-Schwarz: the embedding distance bounds every score error
    assert sdiff <= 4e-3


def test_fp16_encoder_long_sequences_512(oracle):
    """seq_len 512 — the attention kernel's limit (encoder.hip: seq_len <= 512) and bge's max_position_embeddings:
    sixteen 32-key tiles x sixteen query blocks, ragged lengths."""
    from rag_arc_amd.encapsulation.embeddings.hip_bert import HipBertEncoder

    H, heads, I, n_seq, L = 256, 4, 1024, 3, 512
    sd = oracle.random_bert_state_dict(H, 2, heads, I, vocab=800, max_pos=512, seed=512)
    enc = HipBertEncoder(sd, num_heads=heads, precision="fp16")
    rng = np.random.default_rng(512)
    ids = rng.integers(1, 800, (n_seq, L)).astype(np.int32)
    lens = np.array([512, 481, 7], np.int32)
    for r, l in enumerate(lens):
        ids[r, l:] = 0
    got = enc.forward(ids, lens, normalize=True).cpu().numpy()
    sd16 = {k: v.astype(np.float16).astype(np.float32) for k, v in sd.items()}
    want = oracle.bert_forward_f32(sd16, ids, lens, heads, normalize=True)
    assert np.max(np.abs(got - want)) <= 4e-3
    cos = np.sum(got * want, axis=1) / (np.linalg.norm(got, axis=1) * np.linalg.norm(want, axis=1))
    assert cos.min() >= 0.9995


def test_token_ids_and_lengths_are_validated(oracle):
    from rag_arc_amd.encapsulation.embeddings.hip_bert import HipBertEncoder

    sd = oracle.random_bert_state_dict(128, 1, 2, 256, vocab=300, max_pos=64, seed=2)
    enc = HipBertEncoder(sd, num_heads=2, precision="fp16")
    ok = np.ones((2, 8), np.int32)
    for bad_ids, bad_lens in ((np.full((2, 8), 300, np.int32), None), (np.full((2, 8), -1, np.int32), None),
                              (ok, [0, 8]), (ok, [9, 1]), (ok, [1])):
        with pytest.raises(ValueError):
            enc.forward(bad_ids, bad_lens)
    # a single odd-length sequence runs as 4 x 32 tokens (length-masked padding), with the same result
    one = np.array([[5, 6, 7, 8, 9, 10, 11]], np.int32)
    a = enc.forward(one, [7]).cpu().numpy()
    sd16 = {k: v.astype(np.float16).astype(np.float32) for k, v in sd.items()}
    want = oracle.bert_forward_f32(sd16, one, np.array([7]), 2, normalize=True)
    assert np.max(np.abs(a - want)) <= 4e-3


def test_mean_pooling_matches_oracle(oracle):
    from rag_arc_amd.encapsulation.embeddings.hip_bert import HipBertEncoder

    sd = oracle.random_bert_state_dict(384, 2, 12, 1536, vocab=500, max_pos=64, seed=21)
    enc = HipBertEncoder(sd, num_heads=12, pooling="mean", precision="fp16")
    rng = np.random.default_rng(21)
    ids = rng.integers(1, 500, (7, 24)).astype(np.int32)
    lens = np.array([24, 1, 5, 17, 24, 9, 2], np.int32)
    for r, l in enumerate(lens):
        ids[r, l:] = 0
    sd16 = {k: v.astype(np.float16).astype(np.float32) for k, v in sd.items()}
    for norm in (True, False):
        got = enc.forward(ids, lens, normalize=norm).cpu().numpy()
        want = oracle.bert_forward_f32(sd16, ids, lens, 12, normalize=norm, pooling="mean")
        assert np.max(np.abs(got - want)) <= (4e-3 if norm else 2e-2)
        cos = np.sum(got * want, axis=1) / (np.linalg.norm(got, axis=1) * np.linalg.norm(want, axis=1))
        assert cos.min() >= 0.9995


def test_gemm_kernel_alone(oracle):
    """A=I check with asymmetric W (catches transposes), then random data with bias + GELU."""
    import torch

    from rag_arc_amd.hip import binding as B

    lib = B.load_library()
    M, N, K = 256, 384, 128
    rng = np.random.default_rng(1)
    A = np.zeros((M, K), np.float16)
    A[np.arange(K), np.arange(K)] = 1.0                       # rows 0..K-1 form the identity
    W = rng.standard_normal((N, K)).astype(np.float16)
    bias = np.zeros(N, np.float16)
    dA, dW, db = (torch.from_numpy(x).cuda() for x in (A, W, bias))
    dC = torch.empty((M, N), dtype=torch.float16, device="cuda")
    B.check(lib.rarc_enc_gemm(dA.data_ptr(), dW.data_ptr(), db.data_ptr(), dC.data_ptr(), M, N, K, 0, 0))
    C = dC.cpu().numpy()
    assert np.array_equal(C[:K], W.T) and not C[K:].any()
    A = (rng.standard_normal((M, K)) * 0.5).astype(np.float16)
    bias = rng.standard_normal(N).astype(np.float16)
    dA, db = torch.from_numpy(A).cuda(), torch.from_numpy(bias).cuda()
    B.check(lib.rarc_enc_gemm(dA.data_ptr(), dW.data_ptr(), db.data_ptr(), dC.data_ptr(), M, N, K, 1, 0))
    ref = A.astype(np.float64) @ W.astype(np.float64).T + bias.astype(np.float64)
    from scipy.special import erf
    ref = 0.5 * ref * (1 + erf(ref / np.sqrt(2)))
    assert np.max(np.abs(dC.cpu().numpy().astype(np.float64) - ref)) < 2e-2 * max(1.0, np.abs(ref).max() / 8)


@pytest.mark.parametrize("M,N,K,act", [
    (8192, 4096, 64, 0),     # 256x256 ping-pong tiles, a single k tile (prologue + last-tile path only)
    (8192, 4096, 192, 1),    # three k tiles: steady, second-to-last and last schedules, GELU epilogue
    (8192, 4096, 448, 0),    # odd number of k tiles (buffer parity)
    (8192, 1024, 192, 1),    # 256x128 ping-pong tiles (3-slot ring), the minimum of three k tiles
    (8192, 1024, 448, 0),    # ring wraps twice, tail schedules
    (4096, 2304, 320, 0),    # N not a multiple of 256 -> 256x128 tiles, ragged last round
    (8192, 256, 256, 0),     # too few tiles for either: 256-row kernel of the older pipeline
    (1024, 3072, 1024, 0),   # small batch: 128x128 ping-pong tiles (4-slot ring), 16 k tiles
    (1024, 4096, 256, 1),    # ... the minimum of four k tiles, GELU
    (512, 1024, 320, 0),     # ... five k tiles, a quarter of the CUs busy
    (1024, 1024, 192, 0),    # K too short for the ring: one-barrier kernel
    (256 * 37, 256 * 7, 192, 0),   # 259 tiles of 256x256 on 256 persistent workgroups: three of them take a second tile
    (256 * 19, 256 * 27, 128, 1),  # 513 tiles: two full rounds + one, XCD shares of 65 / 64 tiles, GELU
    (256 * 50, 256 * 16, 64, 0),   # 800 tiles (the reranker LM's down projection count), single k tile per tile
    (256 * 3, 256 * 100, 256, 0),  # N >> M: 300 tiles, 8-deep row groups of a 3-row tile grid (partial group only)
    (256 * 70, 256 * 4, 320, 1),   # 280 tiles = one round + 24: the short tail (6 tile rows) is cut off and runs as its own GEMM
    (256 * 200, 256 * 4, 128, 0),  # the reranker LM's projection shape: 800 tiles = 3 rounds + 32
    (256 * 88, 256 * 4, 256, 1),   # 352 tiles = one round + 96: between a quarter and half a round -> the tail runs as one
                                   # round of 256 x 128 tiles (192 workgroups), GELU
    (256 * 80, 256 * 8, 192, 0),   # 640 tiles = two rounds + 128: exactly half a round cut off (256 workgroups of 256 x 128)
])
def test_gemm_large_tile_kernels(oracle, M, N, K, act):
    """Every large-shape GEMM path against a float64 product on a transposition-detecting operand pair
    (random asymmetric A, W; position-dependent bias); fp16 output rounding is the only error allowed."""
    import torch

    from rag_arc_amd.hip import binding as B

    lib = B.load_library()
    g = torch.Generator(device="cuda").manual_seed(M + N + K)
    a = (torch.randn((M, K), device="cuda", generator=g) * 0.5).half()
    w = (torch.randn((N, K), device="cuda", generator=g) * 0.1).half()
    bias = (torch.arange(N, device="cuda", dtype=torch.float32) / N - 0.5).half()
    c = torch.full((M, N), float("nan"), dtype=torch.float16, device="cuda")
    B.check(lib.rarc_enc_gemm(a.data_ptr(), w.data_ptr(), bias.data_ptr(), c.data_ptr(), M, N, K, act, 0))
    ref = a.double() @ w.double().T + bias.double()
    if act:
        ref = 0.5 * ref * (1 + torch.erf(ref / np.sqrt(2)))
    err = (c.double() - ref).abs()
    assert not torch.isnan(c).any()
    # half-ulp of fp16 at the value's magnitude, plus fp32 accumulation noise
    assert bool((err <= ref.abs() * 2.0 ** -10 + 2e-3).all()), float(err.max())


def test_embeddings_provider_contract(oracle):
    from rag_arc_amd.encapsulation.embeddings.hip_bert import HipBertEmbeddings, HipBertEncoder

    sd = oracle.random_bert_state_dict(128, 1, 2, 256, vocab=300, max_pos=64, seed=2)
    tok = lambda text: [1 + (ord(c) % 250) for c in text][:40]
    emb = HipBertEmbeddings(HipBertEncoder(sd, num_heads=2, precision="fp16"), tok, batch_size=4)
    texts = ["alpha", "a much longer piece of text\nwith a newline", "b", "gamma delta", "e"]
    vecs = emb.embed_documents(texts)
    assert len(vecs) == 5 and all(len(v) == 128 and isinstance(v[0], float) for v in vecs)
    assert all(abs(np.linalg.norm(v) - 1.0) < 1e-3 for v in vecs)
    q = emb.embed_query("gamma delta")
    assert np.allclose(q, vecs[3], atol=2e-3)                      # batch composition does not matter
    assert np.allclose(emb.embed_query("x\ny"), emb.embed_query("x y"), atol=1e-6)   # newline -> space
    # sentence-transformers prompts (huggingface.py:26-37): the prompt string goes in front of the text
    enc = emb.encoder
    pq = HipBertEmbeddings(enc, tok, batch_size=4, prompts={"query": "ask: ", "passage": "doc: "}, default_prompt_name="query")
    assert np.allclose(pq.embed_query("gamma"), emb.embed_query("ask: gamma"), atol=1e-6)
    pp = HipBertEmbeddings(enc, tok, batch_size=4, prompts={"query": "ask: ", "passage": "doc: "}, default_prompt_name="query",
                           prompt_name="passage")
    assert np.allclose(pp.embed_query("gamma"), emb.embed_query("doc: gamma"), atol=1e-6)
    with pytest.raises(ValueError):
        HipBertEmbeddings(enc, tok, prompts={"query": "q"}, prompt_name="nope")


def test_store_ingests_device_embeddings_without_round_trip(oracle):
    """add_texts with a HIP embedding provider: vectors go encoder -> ingest kernel on the device and the
    store answers exactly like one fed through python lists."""
    from rag_arc_amd.encapsulation.database.vector_db import HipFlatVectorStore
    from rag_arc_amd.encapsulation.embeddings.hip_bert import HipBertEmbeddings, HipBertEncoder

    sd = oracle.random_bert_state_dict(128, 1, 2, 256, vocab=300, max_pos=64, seed=4)
    tok = lambda text: [1 + (ord(c) * 7 + i) % 250 for i, c in enumerate(text)][:48]
    emb = HipBertEmbeddings(HipBertEncoder(sd, num_heads=2, precision="fp16"), tok, batch_size=64)
    texts = [f"chunk {i} about topic {i % 17}" for i in range(500)]
    fast = HipFlatVectorStore.from_texts(texts, emb, ids=[str(i) for i in range(500)])

    class ListsOnly:            # same provider, device fast path hidden
        embed_documents = staticmethod(emb.embed_documents)
        embed_query = staticmethod(emb.embed_query)

    slow = HipFlatVectorStore.from_texts(texts, ListsOnly(), ids=[str(i) for i in range(500)])
    for q in ("chunk 3 about topic 3", "topic 5", "zzz"):
        a, b = fast.similarity_search_with_score(q, k=10), slow.similarity_search_with_score(q, k=10)
        assert [(d.id, s) for d, s in a] == [(d.id, s) for d, s in b]
    assert fast.similarity_search("chunk 42 about topic 8", k=1)[0].content == "chunk 42 about topic 8"


@pytest.mark.parametrize("heads,dh,n_seq,L", [(3, 64, 5, 64), (2, 64, 4, 100), (4, 32, 6, 200), (2, 64, 3, 512), (5, 32, 2, 33)])
def test_shared_attention_kernel_is_bit_identical_to_the_per_wave_kernel(monkeypatch, heads, dh, n_seq, L):
    """Round 4: sequences of more than one query block take rarc_attention_mfma_shared_kernel (key / value tiles staged once per
    workgroup).  Same MFMAs, same operands, same order: its output must equal the per-wave kernel's (RARC_ENC_ATTN=wave) to the
    bit — ragged lengths, lengths of 1 and L, workgroups with idle waves (L = 33, 100), head_dim 32 and 64."""
    import torch

    from rag_arc_amd.hip import binding as B

    lib = B.load_library()
    H = heads * dh
    g = torch.Generator(device="cuda"); g.manual_seed(L + heads)
    qkv = (torch.randn((n_seq * L, 3 * H), generator=g, device="cuda") * 1.5).half()
    lens = torch.randint(1, L + 1, (n_seq,), generator=g, device="cuda").int()
    lens[0] = L
    lens[-1] = 1
    outs = []
    for mode in ("shared", "wave"):
        monkeypatch.setenv("RARC_ENC_ATTN", mode)
        ctx = torch.full((n_seq * L, H), float("nan"), dtype=torch.float16, device="cuda")
        B.check(lib.rarc_enc_attention(qkv.data_ptr(), lens.data_ptr(), n_seq, L, H, heads, ctx.data_ptr(),
                                       torch.cuda.current_stream().cuda_stream), "attention")
        torch.cuda.synchronize()
        outs.append(ctx.view(torch.int16).cpu().numpy().reshape(n_seq, L, H))
    for b in range(n_seq):      # (rows past a sequence's length are padding queries: both kernels compute them, compare all)
        assert np.array_equal(outs[0][b], outs[1][b]), f"sequence {b} (len {int(lens[b])}) differs"
    # and against a plain fp32 softmax(QK^T / sqrt(d)) V of the same fp16 inputs, real rows only
    q, k, v = (qkv.float().view(n_seq, L, 3, heads, dh)[:, :, i].permute(0, 2, 1, 3) for i in range(3))
    s = q @ k.transpose(-1, -2) / dh ** 0.5
    mask = torch.arange(L, device="cuda")[None, None, None, :] >= lens[:, None, None, None]
    ref = (torch.softmax(s.masked_fill(mask, float("-inf")), -1) @ v).permute(0, 2, 1, 3).reshape(n_seq, L, H)
    got = torch.from_numpy(outs[0]).view(torch.float16).float().cuda()
    for b in range(n_seq):
        n = int(lens[b])
        assert float((got[b, :n] - ref[b, :n]).abs().max()) <= 4e-3


def test_mpnet_forward_same_bits_through_both_attention_kernels(oracle, monkeypatch):
    """... and with MPNet's relative-position bias (only reachable through the forward)."""
    from rag_arc_amd.encapsulation.embeddings.hip_bert import HipBertEncoder

    sd = oracle.random_mpnet_state_dict(128, 2, 2, 256, vocab=300, max_pos=514, seed=5)
    enc = HipBertEncoder(sd, num_heads=2, layer_norm_eps=1e-5, precision="fp16", pooling="mean")
    assert enc.model_type == "mpnet"
    rng = np.random.default_rng(5)
    ids = rng.integers(5, 300, (6, 150)).astype(np.int32)
    lens = np.array([150, 1, 33, 64, 97, 128], np.int32)
    for r, l in enumerate(lens):
        ids[r, l:] = 1
    a = enc.forward(ids, lens).cpu().numpy()
    monkeypatch.setenv("RARC_ENC_ATTN", "wave")
    b = enc.forward(ids, lens).cpu().numpy()
    assert np.array_equal(a, b)
