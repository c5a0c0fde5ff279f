"""A JSON app that selects the HIP encoder and the reranker as registered backends: JSON -> Register.register ->
module graph -> invoke, texts in, Documents out (the reference's provider slot,
core/file_management/embeddings/huggingface.py:85-98,116-126; registry pattern framework/module_test.py:47-101)."""
import json

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

WORDS = ["alpha", "beta", "gamma", "delta", "retrieval", "vector", "search", "index", "query", "document", "fusion", "rank",
         "embedding", "encoder", "corpus", "shard", "kernel", "memory", "score", "top"]


def test_json_app_with_hip_encoder_and_reranker(oracle, tmp_path):
    from safetensors.numpy import save_file

    from rag_arc_amd.config.app_registration import register_multipath_retriever, register_reranker, registrator
    from rag_arc_amd.encapsulation.embeddings.wordpiece import WordPieceTokenizer

    vocab = ["[PAD]", "[UNK]", "[CLS]", "[SEP]", "[MASK]"] + WORDS + ["##s", "##ing", ".", ","] + [str(i) for i in range(10)] + \
            ["##" + str(i) for i in range(10)]
    (tmp_path / "vocab.txt").write_text("\n".join(vocab) + "\n")
    H, LAYERS, HEADS, FFN = 128, 2, 2, 256
    sd = oracle.random_bert_state_dict(H, LAYERS, HEADS, FFN, vocab=len(vocab), max_pos=64, seed=9)
    save_file({k: np.ascontiguousarray(v) for k, v in sd.items()}, str(tmp_path / "model.safetensors"))
    rng = np.random.default_rng(9)
    texts = [" ".join(rng.choice(WORDS, 6)) + f" {i}." for i in range(300)]
    np.savez(tmp_path / "corpus.npz", texts=np.array(texts), ids=np.array([str(i) for i in range(300)]))
    emb = {"type": "hip_bert_embeddings", "weights_path": str(tmp_path / "model.safetensors"),
           "vocab_path": str(tmp_path / "vocab.txt"), "num_heads": HEADS, "batch_size": 64}
    vs = {"type": "hip_flat_vectorstore", "metric": "cosine", "storage": "f32", "embedding": emb,
          "corpus_path": str(tmp_path / "corpus.npz")}
    cfg = {"type": "multipath_retriever", "top_k_per_retriever": 20, "fusion": {"type": "rrf", "k": 60.0},
           "retrievers": [{"type": "vectorstore_retriever", "vectorstore": vs, "search_kwargs": {}}]}
    (tmp_path / "app.json").write_text(json.dumps(cfg))
    register_multipath_retriever(str(tmp_path / "app.json"), "enc_app")
    app = registrator.get_object("enc_app")
    out = app.invoke(texts[17], top_k=5)
    assert len(out) == 5 and out[0].content == texts[17]                    # a stored text retrieves itself first

    # the same embeddings from the fp32 oracle over the same tokenisation: top-5 agree (fp16 encoder, fp32 store)
    tok = WordPieceTokenizer.from_file(str(tmp_path / "vocab.txt"))
    ids = [tok(t) for t in texts + [texts[17]]]
    L = max(map(len, ids))
    arr = np.zeros((len(ids), L), np.int32)
    for r, t in enumerate(ids):
        arr[r, : len(t)] = t
    sd16 = {k: v.astype(np.float16).astype(np.float32) for k, v in sd.items()}
    ref = oracle.bert_forward_f32(sd16, arr, np.array([len(t) for t in ids]), HEADS, normalize=True)
    sims = ref[:-1] @ ref[-1]
    assert set(np.argsort(-sims)[:3].tolist()) <= {int(d.id) for d in out}

    # the reranker, registered from JSON too: logits table -> rarc_rerank_order -> stable descending order
    docs = out
    z = rng.standard_normal((1, len(docs))).astype(np.float16)
    np.savez(tmp_path / "logits.npz", queries=np.array([texts[17]]), docs=np.array([d.content for d in docs]),
             z_no=-z, z_yes=z)
    (tmp_path / "rr.json").write_text(json.dumps({"type": "hip_logit_reranker", "logits_path": str(tmp_path / "logits.npz")}))
    register_reranker(str(tmp_path / "rr.json"), "rr_app")
    rr = registrator.get_object("rr_app")
    got = rr.rerank(texts[17], docs)
    want = [docs[i] for i in oracle.stable_desc_order(oracle.rerank_scores_f16(-z[0], z[0]))]
    assert [d.content for d in got] == [d.content for d in want]
