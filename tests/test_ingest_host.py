"""Host side of the ingest path (no GPU): how HipBertEmbeddings cuts a run of tokenised texts into encoder calls, and
what a provider configured without pooling / layer_norm_eps takes from the checkpoint (an MPNet checkpoint
built with BERT's defaults gives embeddings that are silently not the reference's)."""
import json

import numpy as np
import pytest

from rag_arc_amd.encapsulation.embeddings.hip_bert import HipBertEmbeddings, checkpoint_defaults


def _bare(batch_size=None, max_batch_tokens=4096):
    e = HipBertEmbeddings.__new__(HipBertEmbeddings)
    e.batch_size, e.max_batch_tokens = batch_size, max_batch_tokens
    return e


def test_token_budget_batches_cover_everything_within_budget():
    rng = np.random.default_rng(0)
    lens = np.sort(rng.integers(1, 513, 3000))[::-1].astype(np.int32)       # longest first, as embed_documents orders them
    e = _bare(max_batch_tokens=131072)
    cuts = list(e._batches(lens))
    assert cuts[0][0] == 0 and cuts[-1][1] == 3000 and all(a[1] == b[0] for a, b in zip(cuts, cuts[1:]))
    for s, t in cuts:
        L = -(-int(lens[s:t].max()) // 32) * 32
        assert (t - s) * L <= 131072 or t - s == 1
    assert cuts[0][1] - cuts[0][0] == 256                                    # 256 sequences x 512 tokens fill the budget
    assert max(t - s for s, t in cuts) >= 600                                # shorter texts: more of them per call
    # a fixed batch size (sentence-transformers' encode(batch_size=...)) is honoured as given
    assert list(_bare(batch_size=32)._batches(lens[:70])) == [(0, 32), (32, 64), (64, 70)]
    # one over-long sequence still gets a call of its own
    assert list(_bare(max_batch_tokens=128)._batches(np.array([512, 40, 40], np.int32))) == [(0, 1), (1, 3)]


def test_checkpoint_defaults_follow_the_files_then_the_family(tmp_path):
    bert = {"embeddings.word_embeddings.weight": 0, "encoder.layer.0.attention.self.query.weight": 0}
    mpnet = {"0.auto_model.encoder.relative_attention_bias.weight": 0, "0.auto_model.embeddings.word_embeddings.weight": 0}
    d = checkpoint_defaults(None, bert)
    assert (d["model_type"], d["layer_norm_eps"], d["pooling"], d["force_normalize"]) == ("bert", 1e-12, "cls", False)
    d = checkpoint_defaults(None, mpnet)      # all-mpnet-base-v2, the reference's default model (huggingface.py:6)
    assert (d["model_type"], d["layer_norm_eps"], d["pooling"], d["force_normalize"]) == ("mpnet", 1e-5, "mean", True)
    # the checkpoint's own files win over the family
    w = tmp_path / "model.safetensors"
    w.write_bytes(b"")
    (tmp_path / "config.json").write_text(json.dumps({"layer_norm_eps": 1e-7, "num_attention_heads": 12}))
    (tmp_path / "1_Pooling").mkdir()
    (tmp_path / "1_Pooling" / "config.json").write_text(json.dumps({"pooling_mode_cls_token": False, "pooling_mode_mean_tokens": True}))
    (tmp_path / "modules.json").write_text(json.dumps([{"type": "sentence_transformers.models.Transformer"},
                                                       {"type": "sentence_transformers.models.Pooling"}]))
    d = checkpoint_defaults(str(w), bert)
    assert (d["layer_norm_eps"], d["pooling"], d["force_normalize"], d["num_heads"]) == (1e-7, "mean", False, 12)
    (tmp_path / "1_Pooling" / "config.json").write_text(json.dumps({"pooling_mode_max_tokens": True}))
    with pytest.raises(ValueError):
        checkpoint_defaults(str(w), bert)
