"""Parity on weights that LOOK like a checkpoint, not like N(0, 0.05²).

Every other encoder / LM parity test draws near-Gaussian, outlier-free weights.  Trained checkpoints are different in
ways that stress exactly the places where this build departs from a plain fp32 forward — the folded RMSNorm of the
reranker LM (GEMMs on the UN-normalised fp16 residual stream), the split-operand fp32-class encoder GEMMs, fp16 operands:

  * massive-activation channels: four hidden channels carry values ~100x the rest of the residual stream
  * LayerNorm / RMSNorm gammas up to 30 (and small ones, 0.1), betas off zero
  * weight rows whose magnitudes span 1e-4 .. 10 (per-output-row factors over five decades)
  * one massive-activation TOKEN (an embedding row 50x the others in those channels), present in every sequence

`checkpointify_*` below turns a seeded state dict into one with those statistics (outliers come in compensated pairs —
a gamma of 30 with the consumer's columns / 30, q rows x f with k rows / f ... — the way training leaves them, so the
function stays well conditioned while the tensors span five decades); the oracle (numpy, pinned to transformers) runs
the same weights.  Asserted: no inf / NaN, and the stated tolerances — the fp32-class encoder within 1e-5 (L2) of a
float64 forward (measured 2.4e-7 .. 4.3e-7, level with a numpy fp32 forward), the fp16 encoder within 4e-3 (measured
7e-4), and the LM within 2x what the REFERENCE's own fp16 forward (transformers.Qwen3ForCausalLM in torch.float16, run on
the host on the same weights) loses against the fp32 oracle, and within 3x the error measured when the test was written —
the suite's older LM bound, 1e-1 absolute, was wide enough to hide a precision cliff; there is none: folded-norm and
separate-norm paths sit at 7.1e-2 / 6.8e-2 where the reference's fp16 model sits at 5.7e-2."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

HOT = (3, 17, 100, 201)      # the massive-activation channels


def _factors(rng, n, lo, hi):
    return np.exp(rng.uniform(np.log(lo), np.log(hi), n)).astype(np.float32)


def _gamma(rng, n, hot=True):
    """Norm weights log-uniform in [0.3, 3]; with `hot`, two of the massive channels at 30 and one more at 12 — the
    CONSUMERS of that norm's output get the matching input columns divided by the same factors (_compensate), the way a
    trained checkpoint pairs an outlier gamma with small weights downstream."""
    g = _factors(rng, n, 0.3, 3.0)
    boost = np.ones(n, np.float32)
    if hot:
        boost[list(HOT)[:2]] = 30.0 / g[list(HOT)[:2]]
        boost[HOT[2] % n] = 12.0 / g[HOT[2] % n]
    return g * boost, boost


def _compensate(out, names, boost):
    for k in names:
        out[k] = (out[k] / boost[None, :]).astype(np.float32)


def checkpointify_bert(sd, seed, qkv=("attention.self.query", "attention.self.key", "attention.self.value"),
                       o="attention.output.dense", ln1="attention.output.LayerNorm"):
    """A BertModel (or, with the MPNet tensor names, MPNetModel) state dict with checkpoint-like statistics; see the
    module docstring.  Row scalings come in compensated pairs (q rows x f / k rows / f, v rows x f / o columns / f, FFN1
    rows x f / FFN2 columns / f), so weights span decades while the function stays well conditioned."""
    rng = np.random.default_rng(seed)
    out = {k: np.array(v, dtype=np.float32, copy=True) for k, v in sd.items()}
    H = out["embeddings.word_embeddings.weight"].shape[1]
    hot = [h for h in HOT if h < H]
    n_layers = 1 + max(int(k.split(".")[2]) for k in out if k.startswith("encoder.layer."))
    g, boost = _gamma(rng, H)
    out["embeddings.LayerNorm.weight"] = g
    out["embeddings.LayerNorm.bias"] = (out["embeddings.LayerNorm.bias"] * boost).astype(np.float32)
    for i in range(n_layers):
        p = f"encoder.layer.{i}."
        _compensate(out, [p + n + ".weight" for n in qkv], boost)                 # consumers of the previous LayerNorm
        f = _factors(rng, H, 0.05, 20.0)
        out[p + qkv[0] + ".weight"] *= f[:, None]
        out[p + qkv[0] + ".bias"] *= f
        out[p + qkv[1] + ".weight"] /= f[:, None]
        out[p + qkv[1] + ".bias"] /= f
        f = _factors(rng, H, 0.02, 40.0)
        out[p + qkv[2] + ".weight"] *= f[:, None]
        out[p + qkv[2] + ".bias"] *= f
        out[p + o + ".weight"] /= f[None, :]
        out[p + o + ".weight"][hot] *= 10.0                                       # the sub-layer writes large values into the hot channels
        out[p + o + ".bias"][hot] *= 10.0
        g, boost = _gamma(rng, H)
        out[p + ln1 + ".weight"] = g
        out[p + ln1 + ".bias"] = (out[p + ln1 + ".bias"] * boost).astype(np.float32)
        _compensate(out, [p + "intermediate.dense.weight"], boost)
        inter = out[p + "intermediate.dense.weight"].shape[0]
        f = _factors(rng, inter, 0.2, 5.0)
        f[rng.choice(inter, 3, replace=False)] = 2e-3                             # near-dead neurons: rows of magnitude 1e-4
        out[p + "intermediate.dense.weight"] *= f[:, None]
        out[p + "intermediate.dense.bias"] *= f
        out[p + "output.dense.weight"] /= np.maximum(f, 0.2)[None, :]
        if i + 1 < n_layers:                                                      # (not into the pooled output itself)
            out[p + "output.dense.weight"][hot] *= 10.0
            out[p + "output.dense.bias"][hot] *= 10.0
        g, boost = _gamma(rng, H, hot=(i + 1 < n_layers))                         # (the last norm keeps to the plain range)
        out[p + "output.LayerNorm.weight"] = g
        out[p + "output.LayerNorm.bias"] = (out[p + "output.LayerNorm.bias"] * boost).astype(np.float32)
    out["embeddings.word_embeddings.weight"][:, hot] *= 10.0
    out["embeddings.word_embeddings.weight"][7, hot] *= 20.0                      # token 7: the massive-activation token
    return out


def checkpointify_qwen3(sd, seed, hidden):
    rng = np.random.default_rng(seed)
    out = {k: np.array(v, dtype=np.float32, copy=True) for k, v in sd.items() if k != "lm_head.weight"}
    hot = [h for h in HOT if h < hidden]
    lm_head = out["model.embed_tokens.weight"] * 0.25            # untied head: the logits stay in the tens
    i = 0
    while f"model.layers.{i}.self_attn.q_proj.weight" in out:
        p = f"model.layers.{i}."
        g, boost = _gamma(rng, hidden)                           # exact in exact arithmetic: gamma x 30, consumer columns / 30
        out[p + "input_layernorm.weight"] = g
        _compensate(out, [p + f"self_attn.{n}_proj.weight" for n in "qkv"], boost)
        for n in ("q", "k"):                                     # (the per-head q / k norms follow: row scales only move magnitudes)
            out[p + f"self_attn.{n}_proj.weight"] *= _factors(rng, out[p + f"self_attn.{n}_proj.weight"].shape[0], 0.1, 10.0)[:, None]
            out[p + f"self_attn.{n}_norm.weight"] = _factors(rng, out[p + f"self_attn.{n}_norm.weight"].shape[0], 0.3, 4.0)
        nkv = out[p + "self_attn.v_proj.weight"].shape[0]
        rep = out[p + "self_attn.o_proj.weight"].shape[1] // nkv
        dh = out[p + "self_attn.q_norm.weight"].shape[0]
        f = _factors(rng, nkv, 0.02, 40.0)
        out[p + "self_attn.v_proj.weight"] *= f[:, None]
        f_o = np.repeat(f.reshape(-1, dh), rep, axis=0).reshape(-1)               # q heads of a group read the same v head
        out[p + "self_attn.o_proj.weight"] /= f_o[None, :]
        g, boost = _gamma(rng, hidden)
        out[p + "post_attention_layernorm.weight"] = g
        _compensate(out, [p + "mlp.gate_proj.weight", p + "mlp.up_proj.weight"], boost)
        inter = out[p + "mlp.up_proj.weight"].shape[0]
        f = _factors(rng, inter, 0.2, 5.0)
        f[rng.choice(inter, 3, replace=False)] = 2e-3
        out[p + "mlp.up_proj.weight"] *= f[:, None]
        out[p + "mlp.down_proj.weight"] /= np.maximum(f, 0.2)[None, :]
        if i < 2:                                                # massive activations enter the (never normalised) residual stream early
            out[p + "self_attn.o_proj.weight"][hot] *= 25.0
            out[p + "mlp.down_proj.weight"][hot] *= 25.0
        i += 1
    out["model.norm.weight"] = _factors(rng, hidden, 0.3, 3.0)
    out["model.embed_tokens.weight"][:, hot] *= 30.0
    out["model.embed_tokens.weight"][7, hot] *= 50.0             # token 7: the massive-activation token
    out["lm_head.weight"] = np.ascontiguousarray(lm_head)
    return out


def _bert_inputs(rng, n, L, vocab):
    ids = rng.integers(10, vocab, (n, L))
    lens = rng.integers(L // 2, L + 1, n)
    lens[0] = L
    for r in range(n):
        ids[r, int(rng.integers(1, lens[r]))] = 7           # the massive token, somewhere in every sequence
        ids[r, lens[r]:] = 0
    return ids.astype(np.int32), lens.astype(np.int32)


@pytest.mark.parametrize("H,LAYERS,HEADS,FFN", [(384, 6, 12, 1536), (1024, 4, 16, 4096)])
def test_encoder_on_checkpoint_like_weights(oracle, H, LAYERS, HEADS, FFN):
    from rag_arc_amd.encapsulation.embeddings.hip_bert import HipBertEncoder

    sd = checkpointify_bert(oracle.random_bert_state_dict(H, LAYERS, HEADS, FFN, vocab=500, max_pos=128, seed=H), seed=H + 1)
    rng = np.random.default_rng(H + 2)
    ids, lens = _bert_inputs(rng, 12, 96, 500)
    e64 = oracle.bert_forward_f32(sd, ids, lens, HEADS, normalize=True, dtype=np.float64)
    e32 = oracle.bert_forward_f32(sd, ids, lens, HEADS, normalize=True)
    honest = np.linalg.norm(e32.astype(np.float64) - e64, axis=1).max()        # what a plain fp32 forward loses on this data
    # the reference's precision class
    got32 = HipBertEncoder(sd, num_heads=HEADS, precision="fp32").forward(ids, lens).cpu().numpy()
    assert np.isfinite(got32).all()
    err32 = np.linalg.norm(got32.astype(np.float64) - e64, axis=1).max()
    print(f"CKPT-ENC H={H}: fp32-class ||e - e64|| = {err32:.2e} (numpy fp32 forward: {honest:.2e})")
    assert err32 <= max(1e-5, 4 * honest), (err32, honest)
    # the fp16 forward: its stated class (max |delta| <= 4e-3, cosine >= 0.9995) must survive the outliers too
    sd16 = {k: v.astype(np.float16).astype(np.float32) for k, v in sd.items()}
    want16 = oracle.bert_forward_f32(sd16, ids, lens, HEADS, normalize=True)
    got16 = HipBertEncoder(sd, num_heads=HEADS, precision="fp16").forward(ids, lens).cpu().numpy()
    assert np.isfinite(got16).all()
    d16 = np.abs(got16 - want16).max()
    cos = (got16 * want16).sum(1) / np.linalg.norm(got16, axis=1) / np.linalg.norm(want16, axis=1)
    print(f"CKPT-ENC H={H}: fp16 max|delta| = {d16:.2e}, min cosine = {cos.min():.6f}")
    assert d16 <= 4e-3 and cos.min() >= 0.9995


def test_mpnet_on_checkpoint_like_weights(oracle):
    from rag_arc_amd.encapsulation.embeddings.hip_bert import HipBertEncoder

    H, LAYERS, HEADS, FFN = 768, 3, 12, 3072          # all-mpnet-base-v2's layer geometry
    sd = oracle.random_mpnet_state_dict(H, LAYERS, HEADS, FFN, vocab=500, max_pos=130, seed=5)
    rng = np.random.default_rng(6)
    out = checkpointify_bert(sd, 7, qkv=("attention.attn.q", "attention.attn.k", "attention.attn.v"), o="attention.attn.o",
                             ln1="attention.LayerNorm")
    ids, lens = _bert_inputs(rng, 8, 64, 500)
    ids[ids == 0] = 1                                  # <pad> = 1 in MPNet's vocabulary
    e64 = oracle.mpnet_forward_f32(out, ids, lens, HEADS, eps=1e-5, normalize=True, pooling="mean", dtype=np.float64)
    e32 = oracle.mpnet_forward_f32(out, ids, lens, HEADS, eps=1e-5, normalize=True, pooling="mean")
    honest = np.linalg.norm(e32.astype(np.float64) - e64, axis=1).max()
    got = HipBertEncoder(out, num_heads=HEADS, layer_norm_eps=1e-5, pooling="mean", precision="fp32").forward(ids, lens).cpu().numpy()
    assert np.isfinite(got).all()
    err = np.linalg.norm(got.astype(np.float64) - e64, axis=1).max()
    print(f"CKPT-MPNET: fp32-class ||e - e64|| = {err:.2e} (numpy fp32 forward: {honest:.2e})")
    assert err <= max(1e-5, 4 * honest)


def _reference_fp16_logits(sd, H, LAYERS, NQ, NKV, DH, I, V, ids, mask, token_ids):
    """Last-position logits of transformers.Qwen3ForCausalLM loaded in torch.float16 — the reference's model class and dtype
    (core/rerank/Reranker_Qwen3.py:11-13, :41-49) — on the host.  None if this transformers / torch cannot run it."""
    try:
        import torch
        import transformers

        cfg = transformers.Qwen3Config(vocab_size=V, hidden_size=H, intermediate_size=I, num_hidden_layers=LAYERS,
                                       num_attention_heads=NQ, num_key_value_heads=NKV, head_dim=DH, rms_norm_eps=1e-6,
                                       rope_theta=1e6, tie_word_embeddings=False, attention_bias=False,
                                       max_position_embeddings=max(512, ids.shape[1]))
        model = transformers.Qwen3ForCausalLM(cfg).eval()
        model.load_state_dict({k: torch.from_numpy(np.ascontiguousarray(v, dtype=np.float32)) for k, v in sd.items()})
        model = model.to(torch.float16)
        with torch.no_grad():
            out = model(input_ids=torch.from_numpy(np.asarray(ids, np.int64)), attention_mask=torch.from_numpy(np.asarray(mask, np.int64)))
        return out.logits[:, -1, :][:, token_ids].float().numpy()
    except Exception as exc:  # noqa: BLE001
        print("reference fp16 forward unavailable:", exc)
        return None


# LM: |logit - oracle| measured on this data when the test was written (fp16 model; logits reach ~40 with these weights,
# one fp16 ulp there is 3e-2).  Bound = 3 x measured.
LM_CASES = [
    # H, LAYERS, NQ, NKV, DH, I, n, L, measured max |dlogit|, measured max |dp_yes|
    (256, 4, 4, 2, 64, 512, 6, 64, 2.2e-3, 7.7e-4),
    (1024, 6, 16, 8, 128, 3072, 8, 96, 1.7e-2, 1.6e-3),    # Qwen3-Reranker-0.6B's layer geometry
    (1024, 4, 16, 8, 128, 3072, 64, 128, 7.1e-2, 2.0e-2),  # 8192 tokens: the folded-RMSNorm path (no norm pass, GEMMs on the raw
                                                           # stream); the reference's own fp16 forward is 5.7e-2 from the oracle here
]


@pytest.mark.parametrize("H,LAYERS,NQ,NKV,DH,I,n,L,m_logit,m_p", LM_CASES)
def test_lm_on_checkpoint_like_weights(oracle, monkeypatch, H, LAYERS, NQ, NKV, DH, I, n, L, m_logit, m_p):
    import torch

    from rag_arc_amd.core.rerank import HipCausalLM

    V = 600
    sd = checkpointify_qwen3(oracle.random_qwen3_state_dict(H, LAYERS, NQ, NKV, DH, I, vocab=V, seed=H + L), seed=L, hidden=H)
    lm = HipCausalLM(sd, NQ, NKV, DH, rms_norm_eps=1e-6, rope_theta=1e6)
    rng = np.random.default_rng(H + n)
    ids = rng.integers(10, V, (n, L))
    mask = np.ones((n, L), np.int64)
    for r in range(n):
        p = int(rng.integers(0, L // 2)) if r % 3 else 0
        mask[r, :p] = 0
        ids[r, :p] = 0
        ids[r, int(rng.integers(p, L - 1))] = 7                    # the massive-activation token
    no_id, yes_id = 11, 42
    monkeypatch.delenv("RARC_LM_FUSE_NORM", raising=False)
    got = lm.yes_no_logits(ids, mask, no_id, yes_id).float().cpu().numpy()
    assert np.isfinite(got).all(), "inf / NaN in the fp16 residual stream"
    monkeypatch.setenv("RARC_LM_FUSE_NORM", "0")            # the same batch with the separate RMSNorm passes
    plain = lm.yes_no_logits(ids, mask, no_id, yes_id).float().cpu().numpy()
    monkeypatch.delenv("RARC_LM_FUSE_NORM", raising=False)
    sd16 = {k: np.asarray(v, np.float32).astype(np.float16).astype(np.float32) for k, v in sd.items()}
    n_or = min(n, 8)                                               # (the oracle is a numpy forward: 8 sequences are plenty)
    want = oracle.qwen3_last_logits_f32(sd16, dict(num_attention_heads=NQ, num_key_value_heads=NKV, head_dim=DH,
                                                   rms_norm_eps=1e-6, rope_theta=1e6), ids[:n_or], mask[:n_or], [no_id, yes_id])
    # the yardstick: the REFERENCE's own arithmetic on the same weights — transformers.Qwen3ForCausalLM in torch.float16
    # (Reranker_Qwen3.py:11-13 loads it that way), run on the host: how far an fp16 model sits from the fp32 oracle HERE
    ref16 = _reference_fp16_logits(sd, H, LAYERS, NQ, NKV, DH, I, V, ids[:n_or], mask[:n_or], [no_id, yes_id])
    ref_err = None if ref16 is None else float(np.abs(ref16 - want).max())
    err = np.abs(got[:n_or] - want).max()
    p_got = 1.0 / (1.0 + np.exp(-(got[:n_or, 1] - got[:n_or, 0]).astype(np.float64)))
    p_want = 1.0 / (1.0 + np.exp(-(want[:, 1] - want[:, 0]).astype(np.float64)))
    perr = np.abs(p_got - p_want).max()
    print(f"CKPT-LM H={H} layers={LAYERS} n={n} L={L}: max|dlogit| = {err:.3e} (|logit| up to {np.abs(want).max():.1f}), "
          f"max|dp_yes| = {perr:.3e}, relative {err / np.abs(want).max():.2e}; separate norm passes: "
          f"{np.abs(plain[:n_or] - want).max():.3e}; folded vs separate {np.abs(got - plain).max():.3e}")
    print(f"CKPT-LM   the reference's own fp16 forward (transformers, host) vs the oracle: {ref_err}")
    # an fp16 model's distance from the fp32 oracle depends on the weights (it was 2e-3 .. 7e-2 over these cases), so the
    # bound is taken from the reference's own fp16 forward on the SAME weights and inputs: the HIP forward must not be
    # further from the oracle than 2x that (with the plain path's 3e-2 as the floor) — a precision cliff of the folded-norm
    # GEMMs or of a massive channel would show as a multiple of it
    bound = max(3e-2, 2.0 * ref_err) if ref_err is not None else 1e-1
    assert err <= bound, (err, ref_err)
    assert np.abs(plain[:n_or] - want).max() <= bound
    if m_logit is not None:
        assert err <= 3 * m_logit and perr <= 3 * m_p
