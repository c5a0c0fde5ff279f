"""The fp32-class encoder's QUERY PATH (csrc/encoder_f32.hip, round 6): forwards of 32 / 64 / 128 padded tokens — the call the
reference issues per search, `embed_query` = `embed_documents([text])[0]` (core/file_management/embeddings/huggingface.py:136-145,
called from VectorStore_Faiss.py:240) — run their projections as weight streams (rarc_e32_skinny_gemm_kernel) over the
fragment-major weight images instead of the 128 x 128 tile kernels.

Parity gate = the oracle, as for every encoder test: ||e - e_f64||_2 <= 1e-5 and within 2x of numpy's own fp32 forward.
Beside it: the query path and the tile path compute the same three split products per element and differ only in how the
k range is grouped into fp32 partial sums — bounded here at 2e-6 per embedding; and the path is deterministic (bits)."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _tokens(rng, n_seq, L, vocab, full_first=True):
    ids = rng.integers(1, vocab, (n_seq, L)).astype(np.int32)
    lens = rng.integers(1, L + 1, n_seq).astype(np.int32)
    if full_first:
        lens[0] = L
    for r, l in enumerate(lens):
        ids[r, l:] = 0
    return ids, lens


def _tile_path(enc, ids, lens):
    """The same forward through the 128 x 128 tile kernels: the host pads to a multiple of 128 tokens as it did before round 6
    (query_path off) and the library is told not to take the weight stream for a 128-token forward either."""
    os.environ["RARC_E32_QUERY"] = "0"
    enc.query_path = False
    try:
        return enc.forward(ids, lens, normalize=True).cpu().numpy()
    finally:
        enc.query_path = True
        del os.environ["RARC_E32_QUERY"]


@pytest.mark.parametrize("H,layers,heads,I,n_seq,L", [
    (1024, 3, 16, 4096, 1, 32),    # bge-large geometry, ONE 32-token query: 32 tokens, every projection at its own KR
    (1024, 2, 16, 4096, 1, 7),     # a 7-token query is 32 padded tokens
    (768, 3, 12, 3072, 1, 32),     # bge-base geometry (k ranges of 48 / 192 steps: six-slice groups, idle upper waves)
    (768, 2, 12, 3072, 2, 32),     # two queries: 64 tokens (MT = 2)
    (768, 2, 12, 3072, 4, 30),     # four queries: 128 tokens (MT = 4), ragged
    (384, 2, 12, 1536, 3, 20),     # bge-small geometry, head_dim 32 (the query attention's 32-dim form), three queries -> 128 tokens
    (384, 3, 12, 1536, 1, 32),     # bge-small geometry, ONE query (BASELINE config 1's model): per-head scales over 2 k steps
    (128, 2, 4, 256, 2, 9),        # head_dim 32, two short queries, ragged lengths
    (256, 2, 4, 512, 1, 64),       # one 64-token query: two query blocks in the attention
    (128, 1, 2, 256, 1, 100),      # one 100-token query: 128 tokens
])
def test_query_path_is_in_the_fp32_class_and_next_to_the_tile_path(oracle, H, layers, heads, I, n_seq, L):
    from rag_arc_amd.encapsulation.embeddings.hip_bert import HipBertEncoder

    vocab = 700
    sd = oracle.random_bert_state_dict(H, layers, heads, I, vocab=vocab, max_pos=max(64, 128), seed=H + L + n_seq)
    enc = HipBertEncoder(sd, num_heads=heads, precision="fp32")
    assert enc.query_path
    ids, lens = _tokens(np.random.default_rng(H + L), n_seq, L, vocab)
    got = enc.forward(ids, lens, normalize=True).cpu().numpy()
    again = enc.forward(ids, lens, normalize=True).cpu().numpy()
    assert np.array_equal(got.view(np.uint32), again.view(np.uint32))            # deterministic: fixed reduction orders
    w64 = oracle.bert_forward_f32(sd, ids, lens, heads, normalize=True, dtype=np.float64)
    w32 = oracle.bert_forward_f32(sd, ids, lens, heads, normalize=True)
    d_hip = np.linalg.norm(got.astype(np.float64) - w64, axis=1)
    d_np = np.linalg.norm(w32.astype(np.float64) - w64, axis=1)
    tile = _tile_path(enc, ids, lens)
    d_tile = np.linalg.norm(got.astype(np.float64) - tile.astype(np.float64), axis=1)
    print(f"QUERY-PATH H={H} layers={layers} {n_seq}x{L}: ||hip - f64|| {d_hip.max():.2e}  ||numpy32 - f64|| {d_np.max():.2e}  "
          f"||query - tile|| {d_tile.max():.2e}")
    assert d_hip.max() <= 1e-5
    assert d_hip.max() <= 2.0 * d_np.max() + 3e-7
    assert d_tile.max() <= 2e-6


def test_query_path_takes_exactly_the_small_forwards(oracle):
    """32 / 64 / 128 padded tokens go through the weight stream, everything else through the tile kernels; an encoder built
    without the images (query_path=False) never does, and refuses a 32-token forward_device as before."""
    import torch

    from rag_arc_amd.encapsulation.embeddings.hip_bert import HipBertEncoder
    from rag_arc_amd.hip import binding as B

    sd = oracle.random_bert_state_dict(256, 1, 4, 512, vocab=300, max_pos=128, seed=3)
    enc = HipBertEncoder(sd, num_heads=4, precision="fp32")
    plain = HipBertEncoder(sd, num_heads=4, precision="fp32", query_path=False)
    lib = B.load_library()

    def launches_of(encoder, n_seq, L):
        ids = torch.ones((n_seq, L), dtype=torch.int32, device="cuda")
        lens = torch.full((n_seq,), L, dtype=torch.int32, device="cuda")
        return encoder.forward_device(ids, lens)

    for n_seq, L in ((1, 32), (2, 32), (1, 64), (4, 32), (1, 128)):
        a = launches_of(enc, n_seq, L).cpu().numpy()
        if n_seq * L == 128:                       # the tile path takes 128 tokens too: same embeddings to rounding
            b = launches_of(plain, n_seq, L).cpu().numpy()
            assert np.abs(a - b).max() < 2e-6
        else:
            with pytest.raises(ValueError):
                launches_of(plain, n_seq, L)
    with pytest.raises(ValueError):
        launches_of(enc, 3, 32)                     # 96 tokens: neither path (forward() pads it to 128)
    ids = np.ones((3, 32), np.int32)
    assert enc.forward(ids).shape == (3, 256)
    assert lib.rarc_version() >= 600


def test_query_path_mpnet_family(oracle):
    """The reference's default checkpoint family (all-mpnet-base-v2, huggingface.py:6): relative-position bias, mean pooling."""
    from rag_arc_amd.encapsulation.embeddings.hip_bert import HipBertEncoder

    sd = oracle.random_mpnet_state_dict(768, 2, 12, 3072, vocab=600, max_pos=130, seed=77)
    enc = HipBertEncoder(sd, num_heads=12, pooling="mean", precision="fp32", layer_norm_eps=1e-5)
    ids, lens = _tokens(np.random.default_rng(5), 1, 32, 600)
    got = enc.forward(ids, lens, normalize=True).cpu().numpy()
    w64 = oracle.mpnet_forward_f32(sd, ids, lens, 12, normalize=True, pooling="mean", dtype=np.float64)
    assert np.linalg.norm(got.astype(np.float64) - w64, axis=1).max() <= 1e-5
    assert np.linalg.norm(got - _tile_path(enc, ids, lens), axis=1).max() <= 2e-6
