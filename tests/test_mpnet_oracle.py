"""The MPNet oracle (oracle/cpu_ref.mpnet_forward_f32 — the architecture of the reference's default embedding model,
core/file_management/embeddings/huggingface.py:6 `sentence-transformers/all-mpnet-base-v2`) pinned to transformers'
MPNetModel on seeded weights, its relative-position buckets pinned to transformers' function over every offset, and the
MPNet flavour of the WordPiece tokeniser pinned token for token to transformers' MPNetTokenizer."""
import numpy as np
import pytest

from oracle import cpu_ref as oracle


def test_relative_position_buckets_equal_transformers():
    torch = pytest.importorskip("torch")
    from transformers.models.mpnet.modeling_mpnet import MPNetEncoder

    delta = np.arange(-700, 701)
    want = MPNetEncoder.relative_position_bucket(torch.from_numpy(delta), num_buckets=32, max_distance=128).numpy()
    assert np.array_equal(oracle.mpnet_relative_position_bucket(delta), want)
    tab = oracle.mpnet_rel_bias_table(np.arange(32 * 4, dtype=np.float32).reshape(32, 4), 9)
    assert tab.shape == (4, 17) and tab[2, 8] == 2.0            # offset 0 -> bucket 0, head 2


@pytest.mark.parametrize("hidden,layers,heads,inter,n,L", [(64, 2, 2, 128, 3, 20), (128, 3, 4, 256, 4, 140)])
def test_mpnet_oracle_matches_transformers(hidden, layers, heads, inter, n, L):
    torch = pytest.importorskip("torch")
    from transformers import MPNetConfig, MPNetModel

    cfg = MPNetConfig(vocab_size=300, hidden_size=hidden, num_hidden_layers=layers, num_attention_heads=heads,
                      intermediate_size=inter, max_position_embeddings=L + 2, hidden_dropout_prob=0.0,
                      attention_probs_dropout_prob=0.0, layer_norm_eps=1e-5)
    torch.manual_seed(hidden + L)
    model = MPNetModel(cfg, add_pooling_layer=False).eval()
    with torch.no_grad():
        model.encoder.relative_attention_bias.weight.mul_(5.0)     # make the bias matter
    sd = {k: v.numpy() for k, v in model.state_dict().items()}
    rng = np.random.default_rng(L)
    lens = np.array([L, 1, L // 2, 7][:n])
    ids = rng.integers(4, 300, (n, L))
    for r in range(n):
        ids[r, lens[r]:] = 1                                       # <pad> = padding_idx
    mask = (np.arange(L)[None, :] < lens[:, None]).astype(np.int64)
    with torch.no_grad():
        hs = model(input_ids=torch.from_numpy(ids), attention_mask=torch.from_numpy(mask)).last_hidden_state.numpy()
    m = mask[:, :, None].astype(np.float32)
    want = (hs * m).sum(1) / m.sum(1)
    want /= np.linalg.norm(want, axis=1, keepdims=True)
    got = oracle.mpnet_forward_f32(sd, ids, lens, heads, eps=1e-5, normalize=True, pooling="mean")
    assert np.abs(got - want).max() < 2e-5


def test_wordpiece_with_mpnet_specials_equals_transformers(tmp_path):
    pytest.importorskip("torch")
    from transformers import MPNetTokenizer

    from rag_arc_amd.encapsulation.embeddings.wordpiece import WordPieceTokenizer

    words = ["<s>", "<pad>", "</s>", "<unk>", "[UNK]", "the", "quick", "brown", "fox", "jump", "##s", "##ed", "over", "lazy", "dog",
             ",", ".", "!", "a", "##b", "##c", "un", "##believ", "##able", "caf", "##e", "2024", "<mask>"]
    vp = tmp_path / "vocab.txt"
    vp.write_text("\n".join(words) + "\n")
    ref = MPNetTokenizer(str(vp))
    tok = WordPieceTokenizer.from_file(str(vp), cls_token="<s>", sep_token="</s>", pad_token="<pad>", unk_token="[UNK]",
                                       mask_token="<mask>")
    for text in ["The quick brown fox jumps over the lazy dog.", "Unbelievable, café! abc 2024 xyz", "", "  a  ,b. ",
                 "the <mask> fox", "the<mask>fox <MASK> a<s>a</s> <unk> a<unk>b [UNK]a <pad>a <PAD>", "  <mask>  .<mask>"]:
        assert tok(text) == ref(text)["input_ids"], text
