"""The MPNet family on the GPU — the architecture of the reference's DEFAULT embedding model
(core/file_management/embeddings/huggingface.py:6: sentence-transformers/all-mpnet-base-v2; relative-position attention bias,
no token types, position ids from 2, mean pooling + L2 normalisation) through HipBertEncoder, against
oracle.mpnet_forward_f32 (itself pinned to transformers.MPNetModel, tests/test_mpnet_oracle.py).
Tolerances as for the BERT family: precision="fp32" (the reference's arithmetic) ||e_hip - e_f64||_2 <= 1e-5 per
embedding; precision="fp16" max|d| <= 4e-3 and cosine >= 0.9995."""
import json

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _batch(rng, n_seq, L, vocab):
    ids = rng.integers(4, vocab, (n_seq, L)).astype(np.int32)
    lens = rng.integers(1, L + 1, n_seq).astype(np.int32)
    lens[0] = L
    for r, l in enumerate(lens):
        ids[r, l:] = 1                                       # <pad>
    return ids, lens


@pytest.mark.parametrize("H,layers,heads,I,n_seq,L", [
    (128, 2, 2, 256, 5, 24),        # head_dim 64
    (128, 2, 4, 256, 4, 200),       # head_dim 32, seven key tiles: offsets beyond max_distance (the last bucket)
    (768, 2, 12, 3072, 4, 64),      # all-mpnet-base-v2's layer geometry, two layers
])
def test_mpnet_encoder_matches_oracle_both_precisions(oracle, H, layers, heads, I, n_seq, L):
    from rag_arc_amd.encapsulation.embeddings import hip_bert
    from rag_arc_amd.encapsulation.embeddings.hip_bert import HipBertEncoder

    sd = oracle.random_mpnet_state_dict(H, layers, heads, I, vocab=800, max_pos=514, seed=H + L)
    # the host's bucket table is the oracle's (which is transformers')
    assert np.array_equal(hip_bert._mpnet_rel_bias_table(sd["encoder.relative_attention_bias.weight"], 300),
                          oracle.mpnet_rel_bias_table(sd["encoder.relative_attention_bias.weight"], 300))
    ids, lens = _batch(np.random.default_rng(H + L), n_seq, L, 800)
    enc32 = HipBertEncoder(sd, num_heads=heads, layer_norm_eps=1e-5, pooling="mean", precision="fp32")
    assert enc32.model_type == "mpnet" and enc32.max_pos == 512
    got32 = enc32.forward(ids, lens, normalize=True).cpu().numpy().astype(np.float64)
    want64 = oracle.mpnet_forward_f32(sd, ids, lens, heads, eps=1e-5, normalize=True, pooling="mean", dtype=np.float64)
    d32 = np.linalg.norm(got32 - want64, axis=1).max()
    enc16 = HipBertEncoder(sd, num_heads=heads, layer_norm_eps=1e-5, pooling="mean", precision="fp16")
    got16 = enc16.forward(ids, lens, normalize=True).cpu().numpy()
    sd16 = {k: v.astype(np.float16).astype(np.float32) for k, v in sd.items()}
    want16 = oracle.mpnet_forward_f32(sd16, ids, lens, heads, eps=1e-5, normalize=True, pooling="mean")
    cos = np.sum(got16 * want16, axis=1) / (np.linalg.norm(got16, axis=1) * np.linalg.norm(want16, axis=1))
    print(f"MPNET H={H} layers={layers} heads={heads} L={L}: fp32 mode max ||e - e64|| = {d32:.2e}; fp16 mode max|d| = "
          f"{np.abs(got16 - want16).max():.2e} min cos = {cos.min():.6f}")
    assert d32 <= 1e-5
    assert np.abs(got16 - want16).max() <= 4e-3 and cos.min() >= 0.9995
    # the bias matters in this test: without it the embeddings move by far more than the tolerance
    sd0 = dict(sd)
    sd0["encoder.relative_attention_bias.weight"] = np.zeros_like(sd["encoder.relative_attention_bias.weight"])
    far = oracle.mpnet_forward_f32(sd0, ids, lens, heads, eps=1e-5, normalize=True, pooling="mean", dtype=np.float64)
    assert np.linalg.norm(far - want64, axis=1).max() > 1e-3


def test_mpnet_embeddings_from_json_registry(oracle, tmp_path):
    """`hip_bert_embeddings` built from JSON over an MPNet checkpoint (state dict under sentence-transformers' "0.auto_model."
    prefix, MPNet vocab with <s> / </s> / <pad>): texts in, the oracle's embeddings of the same token ids out."""
    from safetensors.numpy import save_file

    from rag_arc_amd.config.modules import HipBertEmbeddingsConfig

    words = ["<s>", "<pad>", "</s>", "<unk>", "[UNK]", "what", "is", "the", "capital", "of", "france", "paris", "?", ".", "a", "city",
             "in", "europe", "##s", "<mask>"]
    vp = tmp_path / "vocab.txt"
    vp.write_text("\n".join(words) + "\n")
    H, layers, heads, I = 128, 2, 2, 256
    sd = oracle.random_mpnet_state_dict(H, layers, heads, I, vocab=len(words), max_pos=66, seed=5)
    wp = tmp_path / "model.safetensors"
    save_file({"0.auto_model." + k: v for k, v in sd.items()}, str(wp))
    cfg = HipBertEmbeddingsConfig.model_validate(json.loads(json.dumps(dict(
        type="hip_bert_embeddings", weights_path=str(wp), vocab_path=str(vp), num_heads=heads, pooling="mean",
        layer_norm_eps=1e-5, max_length=64))))
    emb = cfg.build()
    texts = ["What is the capital of France?", "Paris is a city in Europe."]
    got = np.asarray(emb.embed_documents(texts), dtype=np.float64)
    vocab = {w: i for i, w in enumerate(words)}
    toks = [["<s>", "what", "is", "the", "capital", "of", "france", "?", "</s>"],
            ["<s>", "paris", "is", "a", "city", "in", "europe", ".", "</s>"]]
    ids = np.array([[vocab[t] for t in row] for row in toks])
    want = oracle.mpnet_forward_f32(sd, ids, np.array([9, 9]), heads, eps=1e-5, normalize=True, pooling="mean", dtype=np.float64)
    assert np.linalg.norm(got - want, axis=1).max() <= 1e-5
