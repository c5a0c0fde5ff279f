"""SURVEY §8 f2 — batched ingest: `from_texts` / `add_texts` with the MI355X encoder as the embedding provider
(reference: FaissVectorStore.add_texts, VectorStore_Faiss.py:156-210, from_texts :484-497; the provider behind it
HuggingFaceEmbeddings.embed_documents, core/file_management/embeddings/huggingface.py:105-134) against the ORACLE CHAIN

    WordPiece ids -> cpu_ref.bert_forward_f32 (numpy; pinned to transformers.BertModel) -> cpu_ref.ingest_f16 / ingest_f8
    -> cpu_ref.flat_search_*

* embeddings: ||e_hip - e_oracle64||_2 <= 1e-5 per text (fp32-class forward; measured ~1e-6)
* stored rows: bit-identical to the oracle's ingest of the SAME embeddings (the device embeddings handed to the oracle),
  and identical to the oracle chain's own rows wherever the two embeddings agree bit for bit
* ids: equal to the oracle chain's top-k on every query whose score gaps exceed the embedding error
* the same text embedded alone, in a 32-sequence call and in a token-budget call agrees to fp32 rounding (not bit for bit:
  tile shapes and split-K slabs follow the batch size — ADVICE r3)
"""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

WORDS = ["alpha", "beta", "gamma", "delta", "retrieval", "vector", "search", "index", "query", "document", "fusion", "rank",
         "embedding", "encoder", "corpus", "shard", "kernel", "memory", "score", "top", "wave", "tile", "fetch", "scan"]
VOCAB = ["[PAD]", "[UNK]", "[CLS]", "[SEP]", "[MASK]"] + WORDS + ["##s", "##ing", "##ed", ".", ",", "!"] + \
        [str(i) for i in range(10)] + ["##" + str(i) for i in range(10)]


def _texts(n, seed, lo=3, hi=40):
    rng = np.random.default_rng(seed)
    sfx = np.array(["", "s", "ing", "ed"])
    out = []
    for i in range(n):
        k = int(rng.integers(lo, hi))
        out.append(" ".join(w + s for w, s in zip(rng.choice(WORDS, k), rng.choice(sfx, k))) + f" {i}.")
    return out


def _oracle_embed(oracle, tok, sd, heads, texts, dtype=np.float32, pooling="cls"):
    rows = [tok(t.replace("\n", " ")) for t in texts]
    L = max(map(len, rows))
    ids = np.zeros((len(rows), L), np.int32)
    for r, row in enumerate(rows):
        ids[r, : len(row)] = row
    return oracle.bert_forward_f32(sd, ids, np.array([len(r) for r in rows]), heads, normalize=True, pooling=pooling, dtype=dtype)


@pytest.fixture(scope="module")
def provider(oracle):
    import torch

    assert torch.cuda.is_available()
    from rag_arc_amd.encapsulation.embeddings.hip_bert import HipBertEmbeddings, HipBertEncoder
    from rag_arc_amd.encapsulation.embeddings.wordpiece import WordPieceTokenizer

    H, LAYERS, HEADS, FFN = 384, 4, 12, 1536      # bge-small's layer geometry, 4 layers deep
    sd = oracle.random_bert_state_dict(H, LAYERS, HEADS, FFN, vocab=len(VOCAB), max_pos=128, seed=21)
    tok = WordPieceTokenizer({t: i for i, t in enumerate(VOCAB)}, max_length=128)
    enc = HipBertEncoder(sd, num_heads=HEADS, precision="fp32")
    return HipBertEmbeddings(enc, tok, max_length=128, pad_id=tok.pad), tok, sd, HEADS


@pytest.mark.parametrize("storage", ["f16", "f8"])
def test_from_texts_equals_the_oracle_chain(provider, oracle, storage):
    import torch

    from rag_arc_amd.encapsulation.database.vector_db import HipFlatVectorStore

    emb, tok, sd, heads = provider
    texts = _texts(1500, 3)
    queries = _texts(40, 99, 3, 12)
    store = HipFlatVectorStore.from_texts(texts, emb, ids=[str(i) for i in range(len(texts))], storage=storage)
    assert emb.last_stats["texts"] == 1500 and emb.last_stats["encoder_calls"] <= 3      # token-budget batches, not 47 x 32
    # (1) embeddings against the float64 oracle forward
    e_hip = emb.embed_documents_device(texts).cpu().numpy()
    e64 = _oracle_embed(oracle, tok, sd, heads, texts, dtype=np.float64)
    err = np.linalg.norm(e_hip.astype(np.float64) - e64, axis=1)
    assert err.max() <= 1e-5, err.max()
    # (2) stored rows = the oracle's ingest of those very embeddings, bit for bit
    rows_hip = store.index.rows.cpu().numpy()
    if storage == "f16":
        rows_ref, _ = oracle.ingest_f16(e_hip)
        assert np.array_equal(rows_hip.view(np.uint16), rows_ref)
    else:
        rows_ref, sc_ref, _ = oracle.ingest_f8(e_hip)
        assert np.array_equal(rows_hip, rows_ref)
        assert np.array_equal(store.index.row_scales.cpu().numpy().view(np.uint32), sc_ref.view(np.uint32))
    # ... and the oracle CHAIN's rows wherever its fp32 embeddings have the same bits
    e32 = _oracle_embed(oracle, tok, sd, heads, texts)
    same = np.all(e_hip.view(np.uint32) == e32.view(np.uint32), axis=1)
    chain_rows = oracle.ingest_f16(e32)[0] if storage == "f16" else oracle.ingest_f8(e32)[0]
    assert np.array_equal(rows_hip.view(chain_rows.dtype)[same], chain_rows[same])
    # (3) search: the store (query embedded on the device) against the chain (query embedded by the oracle)
    k = 10
    q32 = _oracle_embed(oracle, tok, sd, heads, queries)
    if storage == "f16":
        ref_I, ref_D, _ = oracle.flat_search_f16(chain_rows, oracle.normalize_L2(q32), k + 1)
    else:
        ref_I, ref_D, _ = oracle.flat_search_f8(chain_rows, oracle.ingest_f8(e32)[1], oracle.normalize_L2(q32), k + 1)
    got = store.batch_similarity_search_with_score(queries, k=k)
    checked = 0
    for qi, pairs in enumerate(got):
        gaps = ref_D[qi, :-1] - ref_D[qi, 1:]
        sc = np.array([s for _, s in pairs])
        assert np.abs(sc - ref_D[qi, :k]).max() < 3e-5            # embedding error of both sides (1e-5 each) + storage
        if gaps.min() > 1e-4:                                      # gap-safe: the order cannot hinge on that error
            assert [int(d.id) for d, _ in pairs] == ref_I[qi, :k].tolist()
            checked += 1
    assert checked >= 10, checked
    # one-at-a-time callers get the same documents as the batch
    assert [d.id for d in store.similarity_search(queries[0], k=k)] == [d.id for d, _ in got[0]]


def test_batching_does_not_change_an_embedding_beyond_rounding(provider, oracle):
    from rag_arc_amd.encapsulation.embeddings.hip_bert import HipBertEmbeddings

    emb, tok, sd, heads = provider
    texts = _texts(700, 5)
    big = emb.embed_documents_device(texts).cpu().numpy()                       # token-budget calls
    st32 = HipBertEmbeddings(emb.encoder, tok, max_length=128, batch_size=32, pad_id=tok.pad)
    small = st32.embed_documents_device(texts).cpu().numpy()                    # sentence-transformers' default batching
    assert st32.last_stats["encoder_calls"] == 22
    solo = np.array([emb.embed_query(t) for t in texts[:24]], dtype=np.float32)   # each alone
    assert np.linalg.norm(big - small, axis=1).max() < 2e-6
    assert np.linalg.norm(big[:24] - solo, axis=1).max() < 2e-6
    e64 = _oracle_embed(oracle, tok, sd, heads, texts[:24], dtype=np.float64)
    for arr in (big[:24], small[:24], solo):
        assert np.linalg.norm(arr - e64, axis=1).max() < 1e-5


def test_add_texts_in_pieces_equals_one_call(provider):
    import torch

    from rag_arc_amd.encapsulation.database.vector_db import HipFlatVectorStore

    emb, *_ = provider
    texts = _texts(900, 8)
    one = HipFlatVectorStore.from_texts(texts, emb, ids=[str(i) for i in range(900)])
    parts = HipFlatVectorStore(emb)
    for a in range(0, 900, 300):
        parts.add_texts(texts[a:a + 300], ids=[str(i) for i in range(a, a + 300)])
    # (the same texts in other batches: rows agree wherever the embeddings' bits do, and searches agree on documents)
    r1, r2 = one.index.rows, parts.index.rows
    assert (r1 == r2).all(dim=1).float().mean().item() > 0.5
    assert (r1.float() - r2.float()).abs().max().item() < 2e-3 * 1e-2
    q = texts[123]
    assert [d.id for d in one.similarity_search(q, k=5)][:1] == [d.id for d in parts.similarity_search(q, k=5)][:1] == ["123"]
