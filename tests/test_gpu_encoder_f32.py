"""GPU parity of the encoder at the REFERENCE's precision (precision="fp32", csrc/encoder_f32.hip) — the mode
HuggingFaceEmbeddings runs in (`SentenceTransformer(model_name, **model_kwargs)` loads fp32,
core/file_management/embeddings/huggingface.py:96-98,122-126).

Yardstick: the oracle graph in float64 on the same fp32 weights (`oracle.bert_forward_f32(..., dtype=np.float64)`).
Two honest fp32 forwards (numpy here, torch in the reference) differ from it — and from each other — by fp32 rounding
noise; the HIP forward must sit in that class.  Tolerances (floating point, stated as the contract):
  per GEMM element      |C - C64| <= 2^-19 * sum_k |a||w| and rms error <= 4x that of a plain fp32 GEMM (split operands carry 22 bits; fp32 accumulation over 3K/16 MFMA steps)
  embedding (L2-normalised, any depth <= 24 layers)   ||e_hip - e_f64||_2 <= 1e-5   and   <= 2x the numpy-fp32 forward's own distance + 3e-7
      (measured: 3.3e-6 at 24 layers of bge-large geometry, where numpy fp32 is at 3.7e-6)
  induced cosine-score error over the top-100 of a 100k-row scan   <= 1e-5   (the north star's figure),
  top-100 sets identical wherever the oracle's 100/101 gap exceeds 2e-5.
"""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _tokens(rng, n_seq, L, vocab):
    ids = rng.integers(1, vocab, (n_seq, L)).astype(np.int32)
    lens = rng.integers(1, L + 1, n_seq).astype(np.int32)
    lens[0] = L
    for r, l in enumerate(lens):
        ids[r, l:] = 0
    return ids, lens


@pytest.mark.parametrize("M,N,K", [
    (256, 384, 128),        # one-barrier / small ping-pong kernels (split-K kernels with one slice)
    (1024, 3072, 1024),     # 128 x 128 ping-pong tiles
    (512, 1024, 4096),      # long K' = 12288
    (8192, 1024, 1024),     # 256 x 128 ping-pong tiles, fp32 epilogue
    (8192, 4096, 256),      # 256 x 256 persistent tiles, fp32 epilogue in two 32-column halves
    (256 * 70, 1024, 128),  # 280 tiles of 256 x 256: one round + a cut-off tail that runs as its own GEMM (fp32 row offset)
])
def test_split_gemm_is_fp32_class(M, N, K):
    import torch

    from rag_arc_amd.hip import binding as B

    lib = B.load_library()
    g = torch.Generator(device="cuda").manual_seed(M + N + K)
    a = torch.randn((M, K), device="cuda", generator=g)
    w = torch.randn((N, K), device="cuda", generator=g) * 0.04
    # rows / columns of very different magnitude, a few outlier elements, one zero row each: the per-row scales
    a[3] *= 1e-4; a[5] *= 3e3; a[7] = 0; a[11, 17] = 900.0; a[:, 5] *= 50.0
    w[2] *= 1e-3; w[9] *= 40.0; w[13] = 0; w[21, 3] = 7.0
    bias = torch.randn(N, device="cuda", generator=g)
    a3 = torch.empty((M, 3 * K), dtype=torch.float16, device="cuda")
    ra = torch.empty(M, dtype=torch.float32, device="cuda")
    w3 = torch.empty((N, 3 * K), dtype=torch.float16, device="cuda")
    rw = torch.empty(N, dtype=torch.float32, device="cuda")
    c = torch.full((M, N), float("nan"), dtype=torch.float32, device="cuda")
    B.check(lib.rarc_enc32_split_rows(a.data_ptr(), M, K, a3.data_ptr(), ra.data_ptr(), 0))
    B.check(lib.rarc_enc32_split_weight(w.data_ptr(), N, K, w3.data_ptr(), rw.data_ptr(), 0))
    # the split images reproduce the operands to 2^-22 (relative to each element, down to 2^-39 of the row maximum)
    lo, hi = a3[:, :K].double(), a3[:, K:2 * K].double()
    assert torch.equal(a3[:, K:2 * K], a3[:, 2 * K:])
    back = (hi + lo) * ra.double()[:, None]
    assert bool(((back - a.double()).abs() <= a.double().abs() * 2.0 ** -22 + a.double().abs().amax(1, keepdim=True) * 2.0 ** -38).all())
    assert float(ra[7]) == 1.0 and float(rw[13]) == 1.0                       # zero rows: scale 1
    whi, wlo = w3[:, :K].double(), w3[:, K:2 * K].double()
    assert torch.equal(w3[:, :K], w3[:, 2 * K:])
    wb = (whi + wlo) * rw.double()[:, None]
    assert bool(((wb - w.double()).abs() <= w.double().abs() * 2.0 ** -22 + w.double().abs().amax(1, keepdim=True) * 2.0 ** -38).all())
    B.check(lib.rarc_enc32_gemm(a3.data_ptr(), ra.data_ptr(), w3.data_ptr(), rw.data_ptr(), bias.data_ptr(), c.data_ptr(), M, N, K, 0))
    ref = a.double() @ w.double().T + bias.double()
    mag = a.double().abs() @ w.double().abs().T + bias.double().abs()
    err = (c.double() - ref).abs()
    assert not torch.isnan(c).any()
    assert bool((err <= mag * 2.0 ** -19).all()), float((err / mag.clamp_min(1e-300)).max())
    # and it is as good as a plain fp32 GEMM on the same operands (torch / the vendor library), within a small factor
    e32 = ((a @ w.T + bias).double() - ref).abs()
    print(f"SPLIT-GEMM {M}x{N}x{K}: max rel err (vs sum|a||w|) split {float((err / mag).max()):.2e}  torch fp32 {float((e32 / mag).max()):.2e};"
          f" rms split {float(err.pow(2).mean().sqrt()):.2e} torch fp32 {float(e32.pow(2).mean().sqrt()):.2e}")
    assert float(err.pow(2).mean().sqrt()) <= 4.0 * float(e32.pow(2).mean().sqrt()) + 1e-9


def _forward_pair(oracle, H, layers, heads, I, n_seq, L, seed, pooling="cls", vocab=800, max_pos=None):
    from rag_arc_amd.encapsulation.embeddings.hip_bert import HipBertEncoder

    sd = oracle.random_bert_state_dict(H, layers, heads, I, vocab=vocab, max_pos=max_pos or max(64, L), seed=seed)
    enc = HipBertEncoder(sd, num_heads=heads, pooling=pooling, precision="fp32")
    ids, lens = _tokens(np.random.default_rng(seed), n_seq, L, vocab)
    got = enc.forward(ids, lens, normalize=True).cpu().numpy()
    w64 = oracle.bert_forward_f32(sd, ids, lens, heads, normalize=True, pooling=pooling, dtype=np.float64)
    w32 = oracle.bert_forward_f32(sd, ids, lens, heads, normalize=True, pooling=pooling)
    d_hip = np.linalg.norm(got.astype(np.float64) - w64, axis=1)
    d_np = np.linalg.norm(w32.astype(np.float64) - w64, axis=1)
    return got, w32, w64, d_hip, d_np


@pytest.mark.parametrize("H,layers,heads,I,n_seq,L", [
    (128, 1, 2, 256, 3, 8),        # head_dim 64, tiny
    (128, 2, 4, 256, 5, 24),       # head_dim 32
    (384, 2, 12, 1536, 9, 32),     # bge-small shape: rows of 1536 take the 4-chunk row kernels
    (256, 1, 4, 512, 2, 100),      # two 64-key tiles, ragged last query block
    (128, 1, 4, 256, 6, 200),      # four query blocks x four key tiles, ragged lengths
    (768, 2, 12, 3072, 4, 32),     # bge-base layer
    (1024, 2, 16, 4096, 8, 16),    # bge-large layer: the widest rows the row kernels take
])
def test_fp32_encoder_is_in_the_fp32_class(oracle, H, layers, heads, I, n_seq, L):
    got, w32, w64, d_hip, d_np = _forward_pair(oracle, H, layers, heads, I, n_seq, L, seed=H + L)
    print(f"ENC32 H={H} layers={layers} L={L}: ||hip - f64|| max {d_hip.max():.2e}   ||numpy32 - f64|| max {d_np.max():.2e}"
          f"   max|hip - numpy32| {np.abs(got - w32).max():.2e}")
    assert d_hip.max() <= 1e-5
    assert d_hip.max() <= 2.0 * d_np.max() + 3e-7


def test_fp32_encoder_mean_pooling_and_no_normalisation(oracle):
    from rag_arc_amd.encapsulation.embeddings.hip_bert import HipBertEncoder

    sd = oracle.random_bert_state_dict(384, 2, 12, 1536, vocab=500, max_pos=64, seed=21)
    enc = HipBertEncoder(sd, num_heads=12, pooling="mean", precision="fp32")
    rng = np.random.default_rng(21)
    ids = rng.integers(1, 500, (7, 24)).astype(np.int32)
    lens = np.array([24, 1, 5, 17, 24, 9, 2], np.int32)
    for r, l in enumerate(lens):
        ids[r, l:] = 0
    for norm in (True, False):
        got = enc.forward(ids, lens, normalize=norm).cpu().numpy()
        want = oracle.bert_forward_f32(sd, ids, lens, 12, normalize=norm, pooling="mean", dtype=np.float64)
        rel = np.linalg.norm(got - want, axis=1) / np.linalg.norm(want, axis=1)
        assert rel.max() <= 2e-5, rel.max()


@pytest.mark.parametrize("name,H,layers,heads,I", [("bge-base", 768, 12, 12, 3072), ("bge-large", 1024, 24, 16, 4096)])
def test_full_depth_fp32_encoder_and_induced_score_error(oracle, name, H, layers, heads, I):
    """at 12 and 24 layers the embeddings of this mode induce max |d score| <= 1e-5 over the top-100
    of a 100k-row scan, and the same top-100 as the fp32 oracle's embeddings on every gap-safe query."""
    from rag_arc_amd.hip.engine import FlatIndexF16

    n_seq, L = 16, 32
    got, w32, w64, d_hip, d_np = _forward_pair(oracle, H, layers, heads, I, n_seq, L, seed=layers)
    idx = FlatIndexF16(H, metric="cosine", storage="f32")
    idx.add(oracle.synth_rows_f32(100_000, H, seed=1234))
    Dg, Ig = idx.search(got, 100)
    Dw, Iw = idx.search(w32, 101)
    sdiff, safe = 0.0, 0
    for b in range(n_seq):
        common, ig, iw = np.intersect1d(Ig[b], Iw[b][:100], return_indices=True)
        sdiff = max(sdiff, float(np.max(np.abs(Dg[b][ig] - Dw[b][iw]))))
        if Dw[b][99] - Dw[b][100] > 2e-5:
            safe += 1
            assert set(Ig[b].tolist()) == set(Iw[b][:100].tolist()), f"query {b}: top-100 differs from the fp32 oracle's"
    print(f"ENC32-DEPTH {name}: layers={layers} ||hip - f64|| max {d_hip.max():.2e}  ||numpy32 - f64|| max {d_np.max():.2e}  "
          f"max|d_emb| vs numpy32 {np.abs(got - w32).max():.2e}  max|d_score| (top-100, 100k rows) {sdiff:.2e}  gap-safe queries {safe}/{n_seq}")
    assert d_hip.max() <= 1e-5 and d_hip.max() <= 2.0 * d_np.max() + 3e-7
    assert sdiff <= 1e-5
    assert safe >= n_seq // 2


def test_fp32_encoder_long_sequences_512(oracle):
    """seq_len 512 (the kernels' limit, the checkpoint's max_position_embeddings): eight key tiles, eight query blocks."""
    got, w32, w64, d_hip, d_np = _forward_pair(oracle, 256, 2, 4, 1024, 3, 512, seed=512, max_pos=512)
    print(f"ENC32 L=512: ||hip - f64|| max {d_hip.max():.2e}  ||numpy32 - f64|| max {d_np.max():.2e}")
    assert d_hip.max() <= 1e-5 and d_hip.max() <= 2.0 * d_np.max() + 3e-7


def test_fp16_weights_give_identical_fp32_forward_with_zero_low_halves(oracle):
    """A state dict that is already fp16-exact (what the fp16 mode stores) has lo = 0 in every weight split; the fp32
    mode then still beats the fp16 mode's error by orders of magnitude (activations stay fp32)."""
    from rag_arc_amd.encapsulation.embeddings.hip_bert import HipBertEncoder

    sd = oracle.random_bert_state_dict(256, 2, 4, 512, vocab=300, max_pos=64, seed=3)
    sd16 = {k: v.astype(np.float16).astype(np.float32) for k, v in sd.items()}
    ids, lens = _tokens(np.random.default_rng(3), 6, 32, 300)
    w64 = oracle.bert_forward_f32(sd16, ids, lens, 4, normalize=True, dtype=np.float64)
    e32 = HipBertEncoder(sd16, num_heads=4, precision="fp32").forward(ids, lens).cpu().numpy()
    e16 = HipBertEncoder(sd16, num_heads=4, precision="fp16").forward(ids, lens).cpu().numpy()
    d32 = np.linalg.norm(e32 - w64, axis=1).max()
    d16 = np.linalg.norm(e16 - w64, axis=1).max()
    print(f"fp16-exact weights: ||fp32 mode - f64|| {d32:.2e}   ||fp16 mode - f64|| {d16:.2e}")
    assert d32 <= 2e-5 and d16 >= 20 * d32


def test_split_attention_against_the_fp32_mfma_kernel(oracle, monkeypatch):
    """Round 4: at head_dim 64 attention runs on the fp16 MFMA over (hi, lo) split operands.  Same forward through that kernel
    and through the exact-product fp32-MFMA kernel it replaces (RARC_E32_ATTN=mfma32), on weights that stress the split: sharp
    softmaxes (q / k weights x 5), value rows spread over six decades (per-block scales), ragged lengths, five key tiles and
    five query blocks (a workgroup with three idle waves).  Both must sit in the fp32 class against float64, and next to each other."""
    from rag_arc_amd.encapsulation.embeddings.hip_bert import HipBertEncoder

    H, layers, heads, I, L = 256, 2, 4, 1024, 160
    sd = oracle.random_bert_state_dict(H, layers, heads, I, vocab=400, max_pos=L, seed=77)
    rng = np.random.default_rng(77)
    for i in range(layers):
        p = f"encoder.layer.{i}.attention.self."
        sd[p + "query.weight"] = sd[p + "query.weight"] * 5.0
        sd[p + "key.weight"] = sd[p + "key.weight"] * 5.0
        f = np.exp2(np.round(rng.uniform(-10, 10, H))).astype(np.float32)      # value column d scaled by f[d] (six decades) ...
        sd[p + "value.weight"] = sd[p + "value.weight"] * f[:, None]
        sd[p + "value.bias"] = sd[p + "value.bias"] * f
        po = f"encoder.layer.{i}.attention.output.dense.weight"
        sd[po] = sd[po] / f[None, :]                                           # ... and undone by the output projection
    ids, lens = _tokens(rng, 7, L, 400)
    lens[:] = [160, 1, 33, 97, 128, 159, 64]
    for r, l in enumerate(lens):
        ids[r, l:] = 0
    enc = HipBertEncoder(sd, num_heads=heads, precision="fp32")
    split = enc.forward(ids, lens, normalize=True).cpu().numpy()
    monkeypatch.setenv("RARC_E32_ATTN", "mfma32")
    exact = enc.forward(ids, lens, normalize=True).cpu().numpy()
    monkeypatch.delenv("RARC_E32_ATTN")
    again = enc.forward(ids, lens, normalize=True).cpu().numpy()
    w64 = oracle.bert_forward_f32(sd, ids, lens, heads, normalize=True, dtype=np.float64)
    w32 = oracle.bert_forward_f32(sd, ids, lens, heads, normalize=True)
    d_split = np.linalg.norm(split - w64, axis=1).max()
    d_exact = np.linalg.norm(exact - w64, axis=1).max()
    d_np = np.linalg.norm(w32 - w64, axis=1).max()
    print(f"ENC32-ATTN split vs f64 {d_split:.2e}   fp32-MFMA vs f64 {d_exact:.2e}   numpy32 vs f64 {d_np:.2e}   "
          f"max|split - fp32-MFMA| {np.abs(split - exact).max():.2e}")
    assert np.array_equal(split, again)                       # the switch is read per call; the kernel is deterministic
    assert not np.array_equal(split, exact)                   # ... and the two kernels really are different code
    assert d_split <= 1e-5 and d_split <= 2.0 * max(d_np, d_exact) + 3e-7
    assert np.abs(split - exact).max() <= 2e-6


@pytest.mark.parametrize("n_seq", [128, 32, 24])
@pytest.mark.parametrize("stress", [False, True])
def test_ffn1_gelu_fused_into_the_gemm_epilogue(oracle, monkeypatch, stress, n_seq):
    """Round 4: on batches that fill the chip FFN1 runs with bias + GELU + the split of its output fused into the GEMM's
    epilogue; the scale of that split is fixed before the GEMM from a bound on the row (||x|| ||W_j|| + |b_j|), not from the
    row's maximum.  Same forward with the fusion off (RARC_E32_FUSE_GELU=0: fp32 product + row pass): both in the fp32 class
    against float64 and next to each other.  `stress`: FFN1 rows over eight decades, a hot input channel, large biases — the
    bound overshoots the true row maximum by many more binades than on ordinary weights.
    Round 5: batches of 192..256 tiles of 128 x 128 (n_seq 32 and 24 here; FFN1 of bge-large at 768..1024 tokens) take the same
    epilogue in the small-batch kernel (rarc_gemm128pp_f16_kernel<5>)."""
    from rag_arc_amd.encapsulation.embeddings.hip_bert import HipBertEncoder

    H, layers, heads, I, L = 256, 2, 4, 1024, 128          # n_seq 128: 16384 tokens = 64 x 4 tiles of 256 x 256; 32 / 24: 256 / 192 tiles of 128 x 128
    sd = oracle.random_bert_state_dict(H, layers, heads, I, vocab=500, max_pos=L, seed=91)
    rng = np.random.default_rng(91)
    if stress:
        for i in range(layers):
            p = f"encoder.layer.{i}."
            f = np.exp2(np.round(rng.uniform(-13, 13, I))).astype(np.float32)       # FFN1 row j scaled by f[j] ...
            sd[p + "intermediate.dense.weight"] = sd[p + "intermediate.dense.weight"] * f[:, None]
            sd[p + "intermediate.dense.bias"] = sd[p + "intermediate.dense.bias"] * f * 4.0
            sd[p + "output.dense.weight"] = sd[p + "output.dense.weight"] / np.maximum(f, 1.0)[None, :]   # (keeps FFN2's output finite)
            sd[p + "attention.output.LayerNorm.weight"][7] = 25.0                     # a hot channel into FFN1
    ids, lens = _tokens(rng, n_seq, L, 500)
    enc = HipBertEncoder(sd, num_heads=heads, precision="fp32")
    fused = enc.forward(ids, lens, normalize=True).cpu().numpy()
    monkeypatch.setenv("RARC_E32_FUSE_GELU", "0")
    plain = enc.forward(ids, lens, normalize=True).cpu().numpy()
    monkeypatch.delenv("RARC_E32_FUSE_GELU")
    assert np.array_equal(fused, enc.forward(ids, lens, normalize=True).cpu().numpy())
    assert not np.array_equal(fused, plain), "the fused path did not engage (or is the unfused code)"
    w64 = oracle.bert_forward_f32(sd, ids, lens, heads, normalize=True, dtype=np.float64)
    w32 = oracle.bert_forward_f32(sd, ids, lens, heads, normalize=True)
    d_f, d_p, d_np = (np.linalg.norm(v - w64, axis=1).max() for v in (fused, plain, w32))
    print(f"ENC32-FUSE stress={stress}: fused vs f64 {d_f:.2e}   unfused vs f64 {d_p:.2e}   numpy32 vs f64 {d_np:.2e}   "
          f"max|fused - unfused| {np.abs(fused - plain).max():.2e}")
    assert d_f <= 1e-5 and d_f <= 2.0 * max(d_np, d_p) + 3e-7
    assert np.abs(fused - plain).max() <= 2e-6
