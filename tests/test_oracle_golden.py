"""The oracle's python restatements vs outputs of the reference's own code (tests/golden/*.json)."""
import numpy as np

from tests.helpers import golden, unhex


def test_rrf_matches_reference_bit_for_bit(oracle):
    for case in golden("rrf.json"):
        got = oracle.rrf_fuse(case["lists"], case["k"], case["top_k"])
        want = [(f["content"], unhex(f["score_hex"])) for f in case["fused"]]
        assert [c for c, _ in got] == [c for c, _ in want]
        assert [s for _, s in got] == [s for _, s in want]  # exact fp64 equality
        assert [f["rank"] for f in case["fused"]] == list(range(1, len(want) + 1))


def test_relevance_functions_match_reference(oracle):
    rel = golden("relevance.json")
    for f in rel["fns"]:
        s = unhex(f["score_hex"])
        assert oracle.cosine_relevance(s) == unhex(f["cosine_hex"])
        assert oracle.max_inner_product_relevance(s) == unhex(f["ip_hex"])


def test_cosine_scores_within_1e5_of_reference_float64(oracle):
    """Secondary arithmetic pin: the reference's own numpy cosine (spliter.py:326-332, float64) on
    fp16-representable inputs vs the oracle's canonical fp32 path."""
    g = golden("cosine.json")
    X = np.array(g["X_f16_bits"], dtype=np.uint16).view(np.float16).astype(np.float32)
    Y = np.array(g["Y_f16_bits"], dtype=np.uint16).view(np.float16).astype(np.float32)
    want = np.array([[unhex(v) for v in row] for row in g["cos_hex"]])
    rows, _ = oracle.ingest_f16(Y, normalize=True)       # corpus side: normalise -> fp16
    qn = oracle.normalize_L2(X)
    ids, scores, _ = oracle.flat_search_f16(rows, qn, Y.shape[0])
    got = np.zeros_like(want)
    for q in range(X.shape[0]):
        got[q, ids[q]] = scores[q]
    # fp16 storage of the normalised rows costs ~2^-12 relative per element; stay well inside 2e-3 absolute
    assert np.max(np.abs(got - want)) < 2e-3
    # with the SAME stored rows the canonical fp32 scorer is within 1e-5 of float64
    i64, s64 = oracle.flat_search_f64(rows, qn, Y.shape[0])
    assert np.array_equal(ids, i64)
    assert np.max(np.abs(scores - s64)) < 1e-5


def test_canonical_dot_avx_equals_scalar_definition(oracle):
    import ctypes
    rng = np.random.default_rng(1)
    L = oracle.lib()
    for d in (8, 128, 384, 768):
        q = rng.standard_normal(d).astype(np.float32)
        r = rng.standard_normal(d).astype(np.float16).view(np.uint16)
        a = L.oracle_canon_dot_f16(oracle._p(q), oracle._p(r), ctypes.c_int(d))
        b = L.oracle_canon_dot_f16_scalar(oracle._p(q), oracle._p(r), ctypes.c_int(d))
        assert np.float32(a).tobytes() == np.float32(b).tobytes()
    assert L.oracle_selftest() == 0


def test_normalize_and_search_edge_cases(oracle):
    x = np.zeros((3, 16), np.float32)
    x[1, 3] = 2.0
    y = oracle.normalize_L2(x)
    assert np.array_equal(y[0], x[0]) and y[1, 3] == 1.0           # zero rows untouched
    rows, n2 = oracle.ingest_f16(np.eye(4, 8, dtype=np.float32))
    ids, sc, _ = oracle.flat_search_f16(rows, np.ones((1, 8), np.float32), 6)
    assert ids.tolist() == [[0, 1, 2, 3, -1, -1]]                  # ties by id, -1 padding like faiss
    assert np.isneginf(sc[0, 4:]).all()
    empty_ids, empty_sc, _ = oracle.flat_search_f16(np.zeros((0, 128), np.uint16), np.ones((2, 128), np.float32), 3)
    assert (empty_ids == -1).all()


def test_rerank_restatement_matches_torch(oracle):
    """The reference runs its fp16 LM on a GPU, where half log_softmax / exp are evaluated in fp32 and
    rounded to fp16 once (torch's CPU half kernels round more often, so they are not the model)."""
    import torch
    rng = np.random.default_rng(5)
    zn = (rng.standard_normal(2000) * 4).astype(np.float16)
    zy = (rng.standard_normal(2000) * 4).astype(np.float16)
    t = torch.stack([torch.from_numpy(zn), torch.from_numpy(zy)], dim=1).float()
    want_ls = torch.nn.functional.log_softmax(t, dim=1)[:, 1].half()
    got_ls = oracle.rerank_logsoftmax_yes_f16(zn, zy)
    diff = np.abs(got_ls.astype(np.float64) - want_ls.numpy().astype(np.float64))
    ulp = np.maximum(np.abs(want_ls.numpy().astype(np.float64)) * 2.0 ** -10, 2.0 ** -24)
    assert np.all(diff <= ulp)                                  # libm vs SLEEF: last place at most
    same = diff == 0
    assert same.mean() > 0.99
    got = oracle.rerank_scores_f16(zn, zy)
    want = want_ls.float().exp().half().numpy()
    dp = np.abs(got.astype(np.float64) - want.astype(np.float64))
    assert np.all(dp[same] <= np.maximum(np.abs(want[same].astype(np.float64)) * 2.0 ** -10, 2.0 ** -24))
    order = oracle.stable_desc_order(got)
    pairs = sorted(zip(range(len(got)), got.tolist()), key=lambda p: p[1], reverse=True)  # python stable sort
    assert order.tolist() == [i for i, _ in pairs]


def test_topk_merge_is_sharding_invariant(oracle):
    rows = oracle.synth_rows_f16(3000, 128)
    q = oracle.synth_rows_f32(5, 128)
    full_i, full_s, _ = oracle.flat_search_f16(rows, q, 10)
    parts_i, parts_s = [], []
    for lo, hi in ((0, 1000), (1000, 2000), (2000, 3000)):
        i, s, _ = oracle.flat_search_f16(rows[lo:hi], q, 10, id_base=lo)
        parts_i.append(i)
        parts_s.append(s)
    mi, ms = oracle.topk_merge(np.stack(parts_i), np.stack(parts_s), 10)
    assert np.array_equal(mi, full_i) and np.array_equal(ms.view(np.uint32), full_s.view(np.uint32))


def test_bert_oracle_matches_transformers(oracle):
    """Pins oracle.bert_forward_f32 against transformers.BertModel (fp32, seeded weights, padding mask)."""
    import torch
    from transformers import BertConfig, BertModel

    H, Lyr, heads, I = 128, 2, 4, 256
    sd = oracle.random_bert_state_dict(H, Lyr, heads, I, vocab=500, max_pos=64, seed=3)
    cfg = BertConfig(vocab_size=500, hidden_size=H, num_hidden_layers=Lyr, num_attention_heads=heads, intermediate_size=I,
                     max_position_embeddings=64, hidden_act="gelu", layer_norm_eps=1e-12, attn_implementation="eager")
    model = BertModel(cfg, add_pooling_layer=False).eval()
    missing = model.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()}, strict=False)
    assert not [m for m in missing.missing_keys if "position_ids" not in m]
    rng = np.random.default_rng(0)
    ids = rng.integers(1, 500, (5, 24))
    lens = np.array([24, 7, 1, 16, 24])
    for r, l in enumerate(lens):
        ids[r, l:] = 0
    att = (np.arange(24)[None, :] < lens[:, None]).astype(np.int64)
    with torch.no_grad():
        hs = model(input_ids=torch.from_numpy(ids), attention_mask=torch.from_numpy(att)).last_hidden_state
    want = hs[:, 0].numpy()
    got = oracle.bert_forward_f32(sd, ids, lens, heads, normalize=False)
    assert np.max(np.abs(got - want)) < 2e-5
    # mean pooling as sentence-transformers' Pooling(mode_mean_tokens): masked sum / token count
    m = torch.from_numpy(att).float()[:, :, None]
    want_mean = ((hs * m).sum(1) / m.sum(1)).numpy()
    got_mean = oracle.bert_forward_f32(sd, ids, lens, heads, normalize=False, pooling="mean")
    assert np.max(np.abs(got_mean - want_mean)) < 2e-5


# ---- tight pin of the flat-search arithmetic: the reference's float64 numbers on 64 x 4096 vectors ---------------
import pytest


@pytest.mark.parametrize("d", [384, 768])
def test_flat_search_pinned_to_reference_float64(oracle, d):
    """spliter.cosine_similarity (the reference's own float64 numpy arithmetic, spliter.py:326-332) run on 64 x 4096
    fp16-representable vectors (tests/golden/cosine_pin_d*.npz).  The inputs are exact in every storage format, so:
    raw inner product over fp16 rows / float64 norms, and cosine over fp32 rows (normalise -> search, exactly the
    reference's VectorStore_Faiss call sequence), must both land within 1e-5 of the reference's numbers, with the
    reference's own top-100 wherever its gaps exceed the tolerance."""
    from tests.helpers import check_against_pin, pin_reference

    X, Y, cos, xn, yn = pin_reference(d)
    # (a) metric "ip", fp16 storage: rows stored as they are (exact), canonical fp32 dot; reference ip = cos*|x|*|y|
    rows16, _ = oracle.ingest_f16(Y, normalize=False)
    assert np.array_equal(rows16[:, :d].view(np.float16).astype(np.float32), Y)
    ids, sc, _ = oracle.flat_search_f16(rows16, X, 100)
    ip_ref = cos * xn[:, None] * yn[None, :]
    scale = float(np.max(xn)) * float(np.max(yn))            # compare in cosine units: divide by the norm product
    for b in range(X.shape[0]):
        got_cos = sc[b].astype(np.float64) / (xn[b] * yn[ids[b]])
        assert np.max(np.abs(got_cos - cos[b][ids[b]])) < 1e-5
    worst, n_set, n_ord = check_against_pin(ip_ref / scale, ids, sc / np.float32(scale), tol=1e-5)
    assert n_set >= 50 and n_ord >= 10
    # (b) metric "cosine", fp32 storage: normalise rows and queries in fp32 (faiss.normalize_L2), canonical dot
    rows32, _ = oracle.ingest_f32(Y, normalize=True)
    ids2, sc2, _ = oracle.flat_search_f32(rows32, oracle.normalize_L2(X), 100)
    worst2, n_set2, n_ord2 = check_against_pin(cos, ids2, sc2, tol=1e-5)
    assert worst2 < 2e-6 and n_set2 >= 50 and n_ord2 >= 10
    # every pair, not just the top: all 4096 scores per query
    idsA, scA, _ = oracle.flat_search_f32(rows32, oracle.normalize_L2(X), Y.shape[0])
    full = np.zeros_like(cos)
    for b in range(X.shape[0]):
        full[b, idsA[b]] = scA[b]
    assert np.max(np.abs(full - cos)) < 2e-6


def test_qwen3_last_logits_oracle_matches_transformers(oracle):
    """The reranker's LM forward, restated in numpy, against transformers.Qwen3ForCausalLM with the same seeded
    weights at a reduced config, left-padded batch: last-position logits at two token ids (the reference reads
    exactly these, core/rerank/Reranker_Qwen3.py:41-49)."""
    torch = pytest.importorskip("torch")
    transformers = pytest.importorskip("transformers")
    if not hasattr(transformers, "Qwen3ForCausalLM"):
        pytest.skip("this transformers has no Qwen3")
    H, LAYERS, NQ, NKV, DH, I, V = 256, 2, 4, 2, 64, 512, 1000
    sd = oracle.random_qwen3_state_dict(H, LAYERS, NQ, NKV, DH, I, vocab=V, seed=3)
    cfg = transformers.Qwen3Config(vocab_size=V, hidden_size=H, intermediate_size=I, num_hidden_layers=LAYERS,
                                   num_attention_heads=NQ, num_key_value_heads=NKV, head_dim=DH, rms_norm_eps=1e-6,
                                   rope_theta=1e6, tie_word_embeddings=True, attention_bias=False, max_position_embeddings=512)
    model = transformers.Qwen3ForCausalLM(cfg).eval()
    model.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()})
    rng = np.random.default_rng(3)
    n, L = 5, 40
    ids = rng.integers(5, V, (n, L))
    mask = np.ones((n, L), np.int64)
    for r, pad in enumerate((0, 7, 33, 39, 16)):
        mask[r, :pad] = 0
        ids[r, :pad] = 0
    with torch.no_grad():
        want = model(input_ids=torch.from_numpy(ids), attention_mask=torch.from_numpy(mask)).logits[:, -1, :][:, [11, 42]].numpy()
    got = oracle.qwen3_last_logits_f32(sd, dict(num_attention_heads=NQ, num_key_value_heads=NKV, head_dim=DH,
                                                rms_norm_eps=1e-6, rope_theta=1e6), ids, mask, [11, 42])
    assert np.max(np.abs(got - want)) < 2e-4, np.max(np.abs(got - want))
