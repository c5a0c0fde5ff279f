"""The reranker's LM forward on the GPU (csrc/decoder.hip through rarc_lm_yes_no_logits) against the float32 oracle
(oracle.qwen3_last_logits_f32, itself pinned to transformers.Qwen3ForCausalLM): seeded weights at reduced configs,
left-padded batches.  Tolerance (fp16 weights / activations / residual stream, fp32 accumulation, as the reference's
torch.float16 model): |logit - oracle| <= 3e-2 on logits of magnitude ~1, and the reference's final score
p_yes = sigmoid(z_yes - z_no) within 1e-2."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _run(oracle, H, LAYERS, NQ, NKV, DH, I, V, n, L, pads, seed, tol_logit=3e-2, tol_p=1e-2):
    from rag_arc_amd.core.rerank import HipCausalLM

    sd = oracle.random_qwen3_state_dict(H, LAYERS, NQ, NKV, DH, I, vocab=V, seed=seed)
    lm = HipCausalLM(sd, NQ, NKV, DH, rms_norm_eps=1e-6, rope_theta=1e6)
    rng = np.random.default_rng(seed)
    ids = rng.integers(5, V, (n, L))
    mask = np.ones((n, L), np.int64)
    for r in range(n):
        p = pads[r % len(pads)]
        mask[r, :p] = 0
        ids[r, :p] = 0
    no_id, yes_id = 11, 42
    got = lm.yes_no_logits(ids, mask, no_id, yes_id).float().cpu().numpy()
    sd16 = {k: np.asarray(v, np.float32).astype(np.float16).astype(np.float32) for k, v in sd.items()}
    want = oracle.qwen3_last_logits_f32(sd16, dict(num_attention_heads=NQ, num_key_value_heads=NKV, head_dim=DH,
                                                   rms_norm_eps=1e-6, rope_theta=1e6), ids, mask, [no_id, yes_id])
    err = np.max(np.abs(got - want))
    p_got = 1.0 / (1.0 + np.exp(-(got[:, 1] - got[:, 0])))
    p_want = 1.0 / (1.0 + np.exp(-(want[:, 1] - want[:, 0])))
    print(f"LM-FWD H={H} layers={LAYERS} heads={NQ}/{NKV} dh={DH} L={L}: max|dlogit|={err:.2e} "
          f"(|logit| up to {np.abs(want).max():.2f}) max|dp_yes|={np.max(np.abs(p_got - p_want)):.2e}")
    assert err <= tol_logit and np.max(np.abs(p_got - p_want)) <= tol_p
    return got


@pytest.mark.parametrize("H,LAYERS,NQ,NKV,DH,I,n,L,pads", [
    (256, 2, 4, 2, 64, 512, 5, 40, (0, 7, 33, 39, 16)),        # head_dim 64, grouped K/V, ragged left padding
    (256, 1, 2, 2, 128, 256, 4, 32, (0, 31, 1, 12)),           # head_dim 128 (every Qwen3 reranker's), no grouping
    (512, 3, 8, 2, 64, 1024, 3, 100, (0, 60, 99)),             # four q heads per kv head; keys span four tiles
    (1024, 2, 16, 8, 128, 3072, 8, 64, (0, 5, 20, 63)),        # Qwen3-Reranker-0.6B's layer geometry, two layers
    (256, 2, 4, 2, 64, 2048, 32, 64, (0, 9, 40, 63)),          # 2048 tokens: SwiGLU runs as the gate/up GEMM's epilogue
    (384, 2, 6, 2, 64, 512, 4, 96, (0, 33, 64, 95)),           # three q heads per kv head x three query blocks: 9 units
                                                               # per (sequence, kv head), attention workgroups of 4 + 4 + 1
    (256, 1, 2, 1, 128, 256, 4, 160, (0, 1, 100, 159)),        # head_dim 128, 2 x 5 units: workgroups of 4 + 4 + 2, five key tiles
    (256, 2, 4, 2, 64, 512, 2, 1024, (0, 700)),                # long prompts: 32 query blocks x up to 32 key tiles
    (256, 2, 4, 2, 64, 2048, 16, 136, (0, 9, 100, 135)),       # 2176 tokens = 17 x 128: the GEMMs run on 2304 rows (a zero
                                                               # 128-row pad), sequence length not a multiple of 32
])
def test_yes_no_logits_match_oracle(oracle, H, LAYERS, NQ, NKV, DH, I, n, L, pads):
    _run(oracle, H, LAYERS, NQ, NKV, DH, I, 1000, n, L, pads, seed=H + L)


def test_bench_geometry_full_depth_and_vocabulary(oracle):
    """The geometry bench.py's reranker leg runs (Qwen3-Reranker-0.6B: 28 layers, hidden 1024, 16 query / 8 key-value
    heads of 128, ffn 3072, vocabulary 151 669 with tied embeddings), eight left-padded pairs of 64 tokens, against the
    fp32 oracle (0.45 TFLOP on the host).  Tolerance at this depth (fp16 residual stream, error grows ~ sqrt(depth)):
    |logit - oracle| <= 1e-1 on logits of magnitude ~10, p_yes within 3e-2."""
    _run(oracle, 1024, 28, 16, 8, 128, 3072, 151_669, 8, 64, (0, 5, 20, 63, 1, 33, 48, 11), seed=28, tol_logit=1e-1, tol_p=3e-2)


def test_max_length_4096_tokens(oracle):
    """One pair at the reference's max_length (Reranker_Qwen3.py:7: 4096 tokens), 0.6B layer geometry, four layers:
    128 query blocks x up to 128 key tiles per head, rotary angles up to 4095 rad."""
    _run(oracle, 1024, 4, 16, 8, 128, 3072, 2000, 1, 4096, (0,), seed=4096, tol_logit=5e-2, tol_p=2e-2)
    _run(oracle, 1024, 2, 16, 8, 128, 3072, 2000, 1, 4096, (1500,), seed=4097, tol_logit=5e-2, tol_p=2e-2)   # left padded


@pytest.mark.parametrize("H,LAYERS,NQ,NKV,DH,I", [(256, 2, 4, 2, 64, 512), (1024, 2, 16, 8, 128, 3072), (384, 3, 6, 2, 64, 512)])
def test_shared_prefix_equals_whole_prompts(oracle, H, LAYERS, NQ, NKV, DH, I):
    """Prompts that share a beginning (the (query, document) prompts of a reranker: three queries here, each with its own
    prefix length, 5-9 documents each): the prefix runs through the LM once per query (rarc_lm_prefix_kv), the remainders
    attend to its cached k | v rows — and the logits are those of the whole prompts run the reference's way (oracle), at
    the tolerance of the plain path, and within fp16 noise of the plain HIP path."""
    import torch

    from rag_arc_amd.core.rerank import HipCausalLM

    V = 1000
    sd = oracle.random_qwen3_state_dict(H, LAYERS, NQ, NKV, DH, I, vocab=V, seed=H + 7)
    lm = HipCausalLM(sd, NQ, NKV, DH, rms_norm_eps=1e-6, rope_theta=1e6)
    rng = np.random.default_rng(H)
    plens, P = [75, 33, 96], 96                                # prefix lengths (left padded to P), one of them unpadded
    prefixes = [rng.integers(5, V, n).tolist() for n in plens]
    pairs, owner = [], []
    for qi, ndoc in enumerate((9, 5, 6)):
        for _ in range(ndoc):
            pairs.append(rng.integers(5, V, int(rng.integers(1, 140))).tolist())      # remainder: 1 .. 139 tokens
            owner.append(qi)
    n = len(pairs)
    Ls = -(-max(len(p) for p in pairs) // 32) * 32
    pre = np.zeros((4, P), np.int32)                          # 3 prefixes + a dummy row: 4 x 96 = 384 tokens
    pstart = np.full(4, P - 1, np.int32)
    for qi, p in enumerate(prefixes):
        pre[qi, P - len(p):] = p
        pstart[qi] = P - len(p)
    ids = np.zeros((n, Ls), np.int32)
    start = np.zeros(n, np.int32)
    for r, p in enumerate(pairs):
        ids[r, Ls - len(p):] = p
        start[r] = Ls - len(p)
    dev = lm.device
    handle = lm.prefix_kv_device(torch.from_numpy(pre).to(dev), torch.from_numpy(pstart).to(dev))
    got = lm.yes_no_logits_device(torch.from_numpy(ids).to(dev), torch.from_numpy(start).to(dev), 11, 42, prefix=handle,
                                  prefix_of=torch.tensor(owner, dtype=torch.int32, device=dev)).float().cpu().numpy()
    # the whole prompts, left padded, through the oracle and through the plain HIP path
    full = [prefixes[o] + p for o, p in zip(owner, pairs)]
    L = -(-max(len(f) for f in full) // 32) * 32
    f_ids, f_mask = np.zeros((n, L), np.int64), np.zeros((n, L), np.int64)
    for r, f in enumerate(full):
        f_ids[r, L - len(f):] = f
        f_mask[r, L - len(f):] = 1
    sd16 = {k: np.asarray(v, np.float32).astype(np.float16).astype(np.float32) for k, v in sd.items()}
    want = oracle.qwen3_last_logits_f32(sd16, dict(num_attention_heads=NQ, num_key_value_heads=NKV, head_dim=DH, rms_norm_eps=1e-6,
                                                   rope_theta=1e6), f_ids, f_mask, [11, 42])
    plain = lm.yes_no_logits(f_ids, f_mask, 11, 42).float().cpu().numpy()
    sig = lambda z: 1.0 / (1.0 + np.exp(-(z[:, 1] - z[:, 0])))
    print(f"LM-PREFIX H={H} layers={LAYERS}: prefixed vs oracle max|dlogit|={np.abs(got - want).max():.2e}, plain vs oracle "
          f"{np.abs(plain - want).max():.2e}, prefixed vs plain {np.abs(got - plain).max():.2e}; max|dp_yes| vs oracle {np.abs(sig(got) - sig(want)).max():.2e}")
    assert np.abs(got - want).max() <= 3e-2 and np.abs(sig(got) - sig(want)).max() <= 1e-2
    assert np.abs(got - plain).max() <= 3e-2
    # list-level entry used by rerank(): longest common prefix found on the host, same numbers
    z = lm.yes_no_logits_shared_prefix([prefixes[0] + p for p, o in zip(pairs, owner) if o == 0], 11, 42)
    assert np.abs(z.astype(np.float32) - want[[i for i, o in enumerate(owner) if o == 0]]).max() <= 3e-2


@pytest.mark.parametrize("H,LAYERS,NQ,NKV,DH,I,n,L,P", [
    (256, 2, 4, 2, 64, 512, 8, 96, 0),            # head_dim 64, 2 x 3 units on 8 waves
    (1024, 2, 16, 8, 128, 3072, 8, 160, 96),      # 0.6B layer geometry: 2 x 5 units, 96 prefix + 160 own keys = the 288-key limit... minus one tile
    (384, 2, 6, 2, 64, 512, 4, 192, 64),          # three q heads per kv head x six query blocks: 18 units, three rounds of the deal
    (256, 1, 2, 1, 128, 256, 4, 256, 32),         # 2 x 8 units: exactly two rounds; 288 keys = the most head_dim 128 holds
])
def test_resident_attention_equals_streaming_bit_for_bit(oracle, monkeypatch, H, LAYERS, NQ, NKV, DH, I, n, L, P):
    """Short sequences run the LDS-resident attention kernel (every key of a (sequence, K/V head) prepared once, the units dealt
    to 8 waves), long ones the streaming kernel; RARC_LM_ATTN=stream forces the latter.  Same arithmetic in the same order:
    the logits must be IDENTICAL, with and without a shared prefix, for ragged left padding."""
    import torch

    from rag_arc_amd.core.rerank import HipCausalLM

    V = 1000
    sd = oracle.random_qwen3_state_dict(H, LAYERS, NQ, NKV, DH, I, vocab=V, seed=H + L)
    lm = HipCausalLM(sd, NQ, NKV, DH, rms_norm_eps=1e-6, rope_theta=1e6)
    dev = lm.device
    rng = np.random.default_rng(L + P)
    ids = rng.integers(5, V, (n, L)).astype(np.int32)
    start = np.array([0, L - 1, 7, L // 2, 33, 1, L - 40, 64][:n], np.int32)
    for r in range(n):
        ids[r, :start[r]] = 0
    kw = {}
    if P:
        npre = 128 // P if 128 % P == 0 else 4                  # n_prefix * P must be a multiple of 128
        while (npre * P) % 128:
            npre += 1
        pre = rng.integers(5, V, (npre, P)).astype(np.int32)
        pstart = np.array([(0, P - 1, 5, P // 2)[i % 4] for i in range(npre)], np.int32)
        for i in range(npre):
            pre[i, :pstart[i]] = 0
        handle = lm.prefix_kv_device(torch.from_numpy(pre).to(dev), torch.from_numpy(pstart).to(dev))
        kw = dict(prefix=handle, prefix_of=torch.tensor([(i % (npre + 1)) - 1 for i in range(n)], dtype=torch.int32, device=dev))
    run = lambda: lm.yes_no_logits_device(torch.from_numpy(ids).to(dev), torch.from_numpy(start).to(dev), 11, 42, **kw).cpu()
    monkeypatch.delenv("RARC_LM_ATTN", raising=False)
    resident = run()
    monkeypatch.setenv("RARC_LM_ATTN", "stream")
    streaming = run()
    assert torch.isfinite(resident.float()).all()
    assert torch.equal(resident.view(torch.int16), streaming.view(torch.int16))


@pytest.mark.parametrize("n,L,with_oracle", [(32, 256, True), (68, 256, False), (128, 137, False)])
def test_rmsnorm_folded_into_the_projections(oracle, monkeypatch, n, L, with_oracle):
    """Batches large enough for the 256-row tile kernels run without RMSNorm passes: the output / down projections add into the
    residual stream in their epilogue, q|k|v and gate|up multiply the stream by norm-folded weights and scale their rows by
    rsqrt(mean(x²) + eps) (RarcLmLayer.qkv_w_folded / gate_up_w_folded; RARC_LM_FUSE_NORM=0 keeps the separate passes).
    Same function with one rounding moved, so: (a) the fp32 oracle at the tolerance of the plain path, (b) the two paths
    within fp16 noise of each other.  Qwen3-Reranker-0.6B's layer geometry, three layers; 8192 tokens (256 x 256 and 256 x 128
    tile kernels), 17 408 tokens (68 row tiles: a cut-off tail on the 128 x 128 kernel, its row scales offset), 17 536 = 128 x 137
    tokens (an odd multiple of 128: the GEMMs run on a zero-padded 17 664 rows, x's padding rows included)."""
    import torch

    from rag_arc_amd.core.rerank import HipCausalLM

    H, LAYERS, NQ, NKV, DH, I, V = 1024, 3, 16, 8, 128, 3072, 1000
    sd = oracle.random_qwen3_state_dict(H, LAYERS, NQ, NKV, DH, I, vocab=V, seed=n + L)
    lm = HipCausalLM(sd, NQ, NKV, DH, rms_norm_eps=1e-6, rope_theta=1e6)
    rng = np.random.default_rng(n * L)
    ids = rng.integers(5, V, (n, L)).astype(np.int32)
    start = rng.integers(0, L // 2, n).astype(np.int32)
    start[0] = 0
    for r in range(n):
        ids[r, :start[r]] = 0
    assert (n * L) % 128 == 0
    dev = lm.device
    run = lambda: lm.yes_no_logits_device(torch.from_numpy(ids).to(dev), torch.from_numpy(start).to(dev), 11, 42).float().cpu().numpy()
    monkeypatch.delenv("RARC_LM_FUSE_NORM", raising=False)
    fused = run()
    monkeypatch.setenv("RARC_LM_FUSE_NORM", "0")
    plain = run()
    sig = lambda z: 1.0 / (1.0 + np.exp(-(z[:, 1] - z[:, 0])))
    assert np.isfinite(fused).all()
    d = np.abs(fused - plain).max()
    print(f"LM-FOLD n={n} L={L}: folded vs separate norm passes max|dlogit|={d:.2e} (|logit| up to {np.abs(plain).max():.2f}), "
          f"max|dp_yes|={np.abs(sig(fused) - sig(plain)).max():.2e}")
    assert 0 < d <= 4e-2 and np.abs(sig(fused) - sig(plain)).max() <= 1e-2      # different roundings (d > 0: the folded path did run)
    if with_oracle:
        mask = (np.arange(L)[None, :] >= start[:, None]).astype(np.int64)
        sd16 = {k: np.asarray(v, np.float32).astype(np.float16).astype(np.float32) for k, v in sd.items()}
        want = oracle.qwen3_last_logits_f32(sd16, dict(num_attention_heads=NQ, num_key_value_heads=NKV, head_dim=DH, rms_norm_eps=1e-6,
                                                       rope_theta=1e6), ids.astype(np.int64) * mask, mask, [11, 42])
        e_f, e_p = np.abs(fused - want).max(), np.abs(plain - want).max()
        print(f"LM-FOLD vs oracle: folded {e_f:.2e}, separate {e_p:.2e}; max|dp_yes| {np.abs(sig(fused) - sig(want)).max():.2e}")
        # (logits reach 17 here: one fp16 ulp is 1.6e-2; the plain path itself sits at 3.4e-2)
        assert e_f <= 5e-2 and e_f <= 1.5 * e_p + 1e-2 and np.abs(sig(fused) - sig(want)).max() <= 1e-2


def test_reranker_end_to_end_matches_reference_steps(oracle):
    """rerank(): prompt format, truncation, prefix / suffix ids, left padding, batches of 8, p_yes in fp16, stable
    descending order — the steps of core/rerank/Reranker_Qwen3.py:23-75 with a toy tokenizer."""
    from rag_arc_amd.core.rerank import HipCausalLM, HipQwen3Reranker
    from rag_arc_amd.core.utils.data_model import Document

    H, LAYERS, NQ, NKV, DH, I, V = 256, 2, 4, 2, 64, 512, 1000
    sd = oracle.random_qwen3_state_dict(H, LAYERS, NQ, NKV, DH, I, vocab=V, seed=8)
    lm = HipCausalLM(sd, NQ, NKV, DH)
    tok = lambda text: [5 + (ord(c) * 31 + i) % 990 for i, c in enumerate(text)]
    prefix, suffix = [1, 2, 3], [4, 9]
    rr = HipQwen3Reranker(lm, tok, yes_id=42, no_id=11, prefix_ids=prefix, suffix_ids=suffix, max_length=96)
    docs = [Document(content=f"passage {i} " + "x" * (i * 7 % 50), metadata={}, id=str(i)) for i in range(19)]
    out = rr.rerank("what is a vector index?", docs, k=7)
    # oracle: same token sequences through the fp32 forward, fp16 score arithmetic, stable sort
    pairs = [rr.format_instruction(None, "what is a vector index?", d.content) for d in docs]
    room = 96 - len(prefix) - len(suffix)
    seqs = [prefix + tok(p)[:room] + suffix for p in pairs]
    assert max(map(len, seqs)) == 96                                     # truncation is exercised
    scores = []
    for s0 in range(0, len(seqs), 8):
        chunk = seqs[s0:s0 + 8]
        L = max(map(len, chunk))
        ids = np.zeros((len(chunk), L), np.int64)
        mask = np.zeros((len(chunk), L), np.int64)
        for r, s in enumerate(chunk):
            ids[r, L - len(s):] = s
            mask[r, L - len(s):] = 1
        sd16 = {k: np.asarray(v, np.float32).astype(np.float16).astype(np.float32) for k, v in sd.items()}
        z = oracle.qwen3_last_logits_f32(sd16, dict(num_attention_heads=NQ, num_key_value_heads=NKV, head_dim=DH), ids, mask, [11, 42])
        scores.extend(oracle.rerank_scores_f16(z[:, 0].astype(np.float16), z[:, 1].astype(np.float16)).tolist())
    got_scores = rr.compute_scores([("what is a vector index?", d.content) for d in docs])     # one call, like rerank(): shared prefix
    plain = HipQwen3Reranker(lm, tok, yes_id=42, no_id=11, prefix_ids=prefix, suffix_ids=suffix, max_length=96, share_prefix=False)
    plain_scores = []
    for s0 in range(0, len(docs), 8):                                                          # the reference's batches of 8, no sharing
        plain_scores.extend(plain.compute_scores([("what is a vector index?", d.content) for d in docs[s0:s0 + 8]]))
    assert np.max(np.abs(np.array(plain_scores) - np.array(scores, dtype=np.float64))) < 1e-2
    assert np.max(np.abs(np.array(got_scores) - np.array(scores, dtype=np.float64))) < 1e-2
    assert [d.id for d in plain.rerank("what is a vector index?", docs, k=7)] == \
        [docs[i].id for i in oracle.stable_desc_order(np.array(plain_scores))[:7]]
    want_order = oracle.stable_desc_order(np.array(got_scores))          # the order of ITS scores, stable
    assert [d.id for d in out] == [docs[i].id for i in want_order[:7]]
    assert len(out) == 7 and all(isinstance(d, Document) for d in out)


def test_qwen3_reranker_from_json_registry(oracle, tmp_path):
    """JSON -> Register.register -> HipQwen3Reranker: weights from .safetensors, tokenizer from a tokenizer.json (the
    `tokenizers` library), rerank() through the registered module equals the directly constructed one."""
    import json

    from safetensors.numpy import save_file
    from tokenizers import Tokenizer, models, pre_tokenizers

    from rag_arc_amd.config.app_registration import register_qwen3_reranker, registrator
    from rag_arc_amd.core.rerank import HipCausalLM, HipQwen3Reranker
    from rag_arc_amd.core.utils.data_model import Document

    words = ["<|endoftext|>", "[UNK]", "yes", "no", "<|im_start|>", "<|im_end|>", "system", "user", "assistant", "<think>", "</think>",
             "judge", "whether", "the", "document", "meets", "requirements", "query", "instruct", "passage", "about", "vector",
             "index", "search", "rank", "<", ">", ":", ".", ",", "\"", "a", "is", "what", "?"] + [f"w{i}" for i in range(200)]
    tok = Tokenizer(models.WordLevel({w: i for i, w in enumerate(words)}, unk_token="[UNK]"))
    tok.pre_tokenizer = pre_tokenizers.Whitespace()
    tok.save(str(tmp_path / "tokenizer.json"))
    H, LAYERS, NQ, NKV, DH, I, V = 256, 2, 4, 2, 64, 512, len(words)
    V = (V + 7) // 8 * 8
    sd = oracle.random_qwen3_state_dict(H, LAYERS, NQ, NKV, DH, I, vocab=V, seed=12)
    sd = {k: np.ascontiguousarray(v) for k, v in sd.items() if k != "lm_head.weight"}      # tied embeddings: no lm_head tensor
    save_file(sd, str(tmp_path / "model.safetensors"))
    (tmp_path / "rr.json").write_text(json.dumps({"type": "hip_qwen3_reranker", "weights_path": str(tmp_path / "model.safetensors"),
                                                 "tokenizer_path": str(tmp_path / "tokenizer.json"), "num_attention_heads": NQ,
                                                 "num_key_value_heads": NKV, "head_dim": DH, "max_length": 128}))
    register_qwen3_reranker(str(tmp_path / "rr.json"), "q3")
    rr = registrator.get_object("q3")
    docs = [Document(content=f"passage about w{i} vector index w{i + 1} .", metadata={}, id=str(i)) for i in range(11)]
    got = rr.rerank("what is a vector index ?", docs, k=5)
    direct = HipQwen3Reranker(HipCausalLM(sd, NQ, NKV, DH), lambda t: tok.encode(t, add_special_tokens=False).ids,
                              yes_id=tok.token_to_id("yes"), no_id=tok.token_to_id("no"), max_length=128,
                              pad_id=tok.token_to_id("<|endoftext|>"))
    want = direct.rerank("what is a vector index ?", docs, k=5)
    assert [d.id for d in got] == [d.id for d in want] and len(got) == 5
    scores = direct.compute_scores([("what is a vector index ?", d.content) for d in docs[:8]])
    assert len(scores) == 8 and all(0.0 <= s <= 1.0 for s in scores)


def test_qwen3_reranker_registry_with_byte_level_bpe_files(oracle, tmp_path):
    """The same through a byte-level BPE vocabulary (trained offline here; Qwen's is not on disk): tokenizer.json and the
    vocab.json + merges.txt pair both register, tokenise like the `tokenizers` library and give the same ranking."""
    import json

    from safetensors.numpy import save_file
    from tokenizers import AddedToken, Regex, Tokenizer, models, normalizers, pre_tokenizers, trainers

    from rag_arc_amd.config.app_registration import register_qwen3_reranker, registrator
    from rag_arc_amd.config.modules import HipQwen3RerankerConfig
    from rag_arc_amd.core.rerank import HipCausalLM, HipQwen3Reranker
    from rag_arc_amd.core.rerank.bpe import QWEN_PATTERN
    from rag_arc_amd.core.utils.data_model import Document

    docs = [Document(content=f"passage {i}: a vector index ranks {i * 37 % 11} rows; naïve café 北京!", metadata={}, id=str(i)) for i in range(9)]
    corpus = [HipQwen3Reranker.PREFIX, HipQwen3Reranker.SUFFIX, "yes no yes no", "<Instruct>: <Query>: <Document>:"] + [d.content for d in docs]
    tok = Tokenizer(models.BPE())
    tok.normalizer = normalizers.NFC()
    tok.pre_tokenizer = pre_tokenizers.Sequence([pre_tokenizers.Split(Regex(QWEN_PATTERN), behavior="isolated", invert=False),
                                                 pre_tokenizers.ByteLevel(add_prefix_space=False, use_regex=False)])
    tok.train_from_iterator(corpus * 4, trainers.BpeTrainer(vocab_size=420, initial_alphabet=pre_tokenizers.ByteLevel.alphabet(),
                                                            special_tokens=[], show_progress=False))
    first = tok.get_vocab_size()
    tok.add_special_tokens([AddedToken(t, special=True, normalized=False) for t in HipQwen3RerankerConfig.SPECIAL_TOKENS])
    assert tok.token_to_id("<|endoftext|>") == first and tok.token_to_id("yes") is not None and tok.token_to_id("no") is not None
    tok.save(str(tmp_path / "tokenizer.json"))
    spec = json.loads(tok.to_str())
    (tmp_path / "vocab.json").write_text(json.dumps(spec["model"]["vocab"]), encoding="utf-8")
    (tmp_path / "merges.txt").write_text("#version: 0.2\n" + "\n".join(" ".join(m) for m in spec["model"]["merges"]) + "\n", encoding="utf-8")
    H, LAYERS, NQ, NKV, DH, I = 256, 2, 4, 2, 64, 512
    V = (tok.get_vocab_size() + 7) // 8 * 8
    sd = {k: np.ascontiguousarray(v) for k, v in oracle.random_qwen3_state_dict(H, LAYERS, NQ, NKV, DH, I, vocab=V, seed=13).items()
          if k != "lm_head.weight"}
    save_file(sd, str(tmp_path / "model.safetensors"))
    common = {"type": "hip_qwen3_reranker", "weights_path": str(tmp_path / "model.safetensors"), "num_attention_heads": NQ,
              "num_key_value_heads": NKV, "head_dim": DH, "max_length": 160}
    (tmp_path / "a.json").write_text(json.dumps({**common, "tokenizer_path": str(tmp_path / "tokenizer.json")}))
    (tmp_path / "b.json").write_text(json.dumps({**common, "vocab_path": str(tmp_path / "vocab.json"), "merges_path": str(tmp_path / "merges.txt")}))
    register_qwen3_reranker(str(tmp_path / "a.json"), "q3_json")
    a = registrator.get_object("q3_json")
    direct = HipQwen3Reranker(HipCausalLM(sd, NQ, NKV, DH), lambda t: tok.encode(t, add_special_tokens=False).ids,
                              yes_id=tok.token_to_id("yes"), no_id=tok.token_to_id("no"), max_length=160,
                              pad_id=tok.token_to_id("<|endoftext|>"))
    assert a.impl.prefix_ids == direct.prefix_ids and a.impl.suffix_ids == direct.suffix_ids
    assert (a.impl.yes_id, a.impl.no_id, a.impl.pad_id) == (direct.yes_id, direct.no_id, direct.pad_id)
    q = "what's a vector index?"
    want = [d.id for d in direct.rerank(q, docs, k=6)]
    assert [d.id for d in a.rerank(q, docs, k=6)] == want
    # vocab.json + merges.txt: the special tokens take Qwen's fixed ids (151643 ...), which this toy embedding table does not
    # hold — the tokenisation of plain text and the yes / no ids are what the pair of files determines
    cfg_b = HipQwen3RerankerConfig(**json.loads((tmp_path / "b.json").read_text()))
    from rag_arc_amd.core.rerank.bpe import ByteLevelBPETokenizer

    tb = ByteLevelBPETokenizer.from_files(cfg_b.vocab_path, cfg_b.merges_path,
                                          {t: i for i, t in enumerate(cfg_b.SPECIAL_TOKENS, start=cfg_b.FIRST_SPECIAL_ID)})
    assert tb.encode(docs[3].content) == tok.encode(docs[3].content, add_special_tokens=False).ids
    assert tb.encode("<|im_end|>\n") == [cfg_b.FIRST_SPECIAL_ID + 2] + tok.encode("\n", add_special_tokens=False).ids
    assert tb.convert_tokens_to_ids("yes") == tok.token_to_id("yes")


@pytest.mark.parametrize("m,n,k", [(2048, 4096, 256), (4096, 4096, 128), (8192, 6144, 192), (256, 32768, 64 * 3),
                                   (256 * 134, 1024, 128),    # 536 tiles: a short last round that must NOT be cut off (its rows alone could not run the fused epilogue)
                                   (256 * 80, 2048, 256)])    # 640 tiles = two rounds + 128: half a round cut off, SwiGLU on the 256 x 128 kernel
def test_swiglu_epilogue_equals_gemm_then_swiglu(m, n, k):
    """rarc_enc_gemm act = 3 (gate / up columns interleaved in groups of 8 -> silu(gate)·up, [m][n/2]) against the
    plain GEMM's own output put through the same roundings on the host side: fp16 GEMM output, fp16 silu, fp16 product.
    Both the 256 x 256 and the 256 x 128 tile kernels are taken by these shapes."""
    import torch

    from rag_arc_amd.hip import binding as B

    lib = B.load_library()
    g = torch.Generator(device="cuda")
    g.manual_seed(m + n + k)
    a = (torch.randn((m, k), generator=g, device="cuda") * 0.5).half()
    w = (torch.randn((n, k), generator=g, device="cuda") * 0.2).half()
    bias = (torch.randn((n,), generator=g, device="cuda") * 0.3).half()
    plain = torch.empty((m, n), dtype=torch.float16, device="cuda")
    fused = torch.full((m, n // 2), float("nan"), dtype=torch.float16, device="cuda")
    st = torch.cuda.current_stream().cuda_stream
    B.check(lib.rarc_enc_gemm(a.data_ptr(), w.data_ptr(), bias.data_ptr(), plain.data_ptr(), m, n, k, 0, st))
    B.check(lib.rarc_enc_gemm(a.data_ptr(), w.data_ptr(), bias.data_ptr(), fused.data_ptr(), m, n, k, 3, st))
    gu = plain.view(m, n // 16, 2, 8).float()
    gate, up = gu[:, :, 0].reshape(m, n // 2), gu[:, :, 1].reshape(m, n // 2)
    want = ((gate / (1.0 + torch.exp(-gate))).half().float() * up).half()
    assert bool(torch.isfinite(fused).all())
    diff = (fused.float() - want.float()).abs()
    ulp = torch.clamp(want.float().abs(), min=6e-5) * 2.0 ** -10            # device __expf vs torch.exp: a last-bit matter
    assert bool((diff <= 2.0 * ulp).all()) and float((diff == 0).float().mean()) > 0.98
    # a shape the 256-row kernels do not take: refused loudly, not computed some other way
    small = torch.empty((128, 64), dtype=torch.float16, device="cuda")
    rc = lib.rarc_enc_gemm(a.data_ptr(), w.data_ptr(), bias.data_ptr(), small.data_ptr(), 128, 128, k, 3, st)
    assert rc != 0
