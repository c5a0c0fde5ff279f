"""Seeded sweeps over model geometries the hand-picked tests do not name — encoder (heads, head width, depth, pooling,
ragged lengths, odd sequence lengths) and reranker LM (grouped K/V ratios, head_dim 64 / 128, left padding, sequence
lengths off the 32-token grid, batch sizes off the multiples of four) — against the fp32 oracles, at the tolerances of
tests/test_gpu_encoder.py and tests/test_gpu_reranker_lm.py.  And concurrent callers: pool threads sharing one index."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
_FIRST, _LAST = (int(v) for v in os.environ.get("RARC_FUZZ_SEEDS", "0:8").split(":"))


@pytest.mark.parametrize("seed", range(_FIRST, _LAST))
def test_random_encoder_geometry(oracle, seed):
    from rag_arc_amd.encapsulation.embeddings.hip_bert import HipBertEncoder

    rng = np.random.default_rng(3000 + seed)
    dh = int(rng.choice([32, 64]))
    heads = int(rng.choice([2, 4, 6, 12]))
    H = heads * dh
    if H % 128:                      # the encoder takes hidden sizes on the 128 grid
        heads = 4 if dh == 32 else 2
        H = heads * dh
    layers, inter = int(rng.integers(1, 4)), int(rng.choice([256, 512, 768]))
    n, L = int(rng.integers(1, 20)), int(rng.integers(3, 130))
    pooling = ("cls", "mean")[seed % 2]
    sd = oracle.random_bert_state_dict(H, layers, heads, inter, vocab=500, max_pos=160, seed=seed)
    ids = rng.integers(1, 500, (n, L)).astype(np.int32)
    lens = rng.integers(1, L + 1, n).astype(np.int32)
    lens[0] = L
    for r, l in enumerate(lens):
        ids[r, l:] = 0
    enc = HipBertEncoder(sd, num_heads=heads, pooling=pooling, precision="fp16")
    got = enc.forward(ids, lens, normalize=True).cpu().numpy()
    sd16 = {k: v.astype(np.float16).astype(np.float32) for k, v in sd.items()}      # the storage format's rounding
    want = oracle.bert_forward_f32(sd16, ids, lens, heads, normalize=True, pooling=pooling)
    err = float(np.max(np.abs(got - want)))
    cos = float(np.min(np.sum(got * want, axis=1) / (np.linalg.norm(got, axis=1) * np.linalg.norm(want, axis=1))))
    assert err <= 4e-3 and cos >= 0.9995, (H, heads, layers, inter, n, L, pooling, err, cos)


@pytest.mark.parametrize("seed", range(_FIRST, _LAST))
def test_random_reranker_lm_geometry(oracle, seed):
    from rag_arc_amd.core.rerank import HipCausalLM

    rng = np.random.default_rng(4000 + seed)
    dh = int(rng.choice([64, 128]))
    nq, nkv = [(2, 1), (2, 2), (4, 2), (4, 4), (6, 2), (8, 2), (8, 8), (3, 1)][int(rng.integers(0, 8))]
    if ((nq + 2 * nkv) * dh) % 128 or (nq * dh) % 64:
        nq, nkv = 4, 2
    H, inter = int(rng.choice([128, 256, 384])), int(rng.choice([128, 256, 512]))
    layers = int(rng.integers(1, 4))
    n, L = int(rng.integers(1, 10)), int(rng.integers(5, 200))
    V = 400
    sd = oracle.random_qwen3_state_dict(H, layers, nq, nkv, dh, inter, vocab=V, seed=seed)
    lm = HipCausalLM(sd, nq, nkv, dh, rms_norm_eps=1e-6, rope_theta=1e6)
    ids = rng.integers(5, V, (n, L))
    mask = np.ones((n, L), np.int64)
    for r in range(n):
        p = int(rng.integers(0, L))          # at least one real token (the last)
        mask[r, :p] = 0
        ids[r, :p] = 0
    got = lm.yes_no_logits(ids, mask, 7, 9).float().cpu().numpy()
    sd16 = {k: np.asarray(v, np.float32).astype(np.float16).astype(np.float32) for k, v in sd.items()}
    want = oracle.qwen3_last_logits_f32(sd16, dict(num_attention_heads=nq, num_key_value_heads=nkv, head_dim=dh,
                                                   rms_norm_eps=1e-6, rope_theta=1e6), ids, mask, [7, 9])
    err = float(np.max(np.abs(got - want)))
    assert np.all(np.isfinite(got)) and err <= 3e-2, (H, layers, nq, nkv, dh, inter, n, L, err)


@pytest.mark.parametrize("seed", range(_FIRST, _LAST))
def test_random_prefixed_batches_resident_equals_streaming(oracle, monkeypatch, seed):
    """Random decoder geometry, random left padding, a random shared prefix (or none): the LDS-resident attention kernel and the
    streaming one (RARC_LM_ATTN=stream) must return the same bits — and finite ones."""
    import torch

    from rag_arc_amd.core.rerank import HipCausalLM

    rng = np.random.default_rng(5000 + seed)
    dh = int(rng.choice([64, 128]))
    nq, nkv = [(2, 1), (2, 2), (4, 2), (4, 4), (6, 2), (8, 2), (8, 8), (3, 1), (16, 8)][int(rng.integers(0, 9))]
    if ((nq + 2 * nkv) * dh) % 128 or (nq * dh) % 64:
        nq, nkv = 4, 2
    H, inter = int(rng.choice([128, 256])), int(rng.choice([128, 256]))
    layers = int(rng.integers(1, 3))
    P = int(rng.choice([0, 0, 16, 40, 64, 79, 96]))
    key_cap = 288 if dh == 128 else 544
    L = int(rng.integers(4, max(5, key_cap - P - 8)))
    n = int(rng.integers(1, 6)) * (128 // np.gcd(L, 128))          # n * L a multiple of 128
    if n * L > 40_000:
        n = 128 // np.gcd(L, 128)
    V = 300
    sd = oracle.random_qwen3_state_dict(H, layers, nq, nkv, dh, inter, vocab=V, seed=seed)
    lm = HipCausalLM(sd, nq, nkv, dh, rms_norm_eps=1e-6, rope_theta=1e6)
    dev = lm.device
    ids = rng.integers(5, V, (n, L)).astype(np.int32)
    start = rng.integers(0, L, n).astype(np.int32)
    start[0] = 0
    for r in range(n):
        ids[r, :start[r]] = 0
    kw = {}
    if P:
        npre = 128 // np.gcd(P, 128)
        pre = rng.integers(5, V, (npre, P)).astype(np.int32)
        pstart = rng.integers(0, P, npre).astype(np.int32)
        pstart[0] = 0
        for i in range(npre):
            pre[i, :pstart[i]] = 0
        handle = lm.prefix_kv_device(torch.from_numpy(pre).to(dev), torch.from_numpy(pstart).to(dev))
        kw = dict(prefix=handle, prefix_of=torch.from_numpy(rng.integers(-1, npre, n).astype(np.int32)).to(dev))
    run = lambda: lm.yes_no_logits_device(torch.from_numpy(ids).to(dev), torch.from_numpy(start).to(dev), 7, 9, **kw).cpu()
    monkeypatch.delenv("RARC_LM_ATTN", raising=False)
    a = run()
    monkeypatch.setenv("RARC_LM_ATTN", "stream")
    b = run()
    assert torch.isfinite(a.float()).all(), (dh, nq, nkv, H, layers, P, L, n)
    assert torch.equal(a.view(torch.int16), b.view(torch.int16)), (dh, nq, nkv, H, layers, P, L, n)


def test_pool_threads_share_one_index(oracle):
    """Callers may be pool threads (core/retrieval/base.py:92-96: a new ThreadPoolExecutor per ainvoke): eight threads
    search one index and one fp8 index at once, every answer equals the serial one."""
    from concurrent.futures import ThreadPoolExecutor

    from rag_arc_amd.hip.engine import FlatIndexF16

    rng = np.random.default_rng(77)
    n, d = 120_000, 256
    X = rng.standard_normal((n, d)).astype(np.float32)
    a, b = FlatIndexF16(d), FlatIndexF16(d, storage="f8")
    a.add(X)
    b.add(X)
    jobs = [(a if i % 2 == 0 else b, rng.standard_normal((int(rng.integers(1, 90)), d)).astype(np.float32), int(rng.choice([1, 10, 50])))
            for i in range(48)]
    serial = [idx.search(q, k) for idx, q, k in jobs]
    with ThreadPoolExecutor(8) as pool:
        threaded = list(pool.map(lambda j: j[0].search(j[1], j[2]), jobs))
    for (sd_, si), (td, ti) in zip(serial, threaded):
        assert np.array_equal(si, ti) and np.array_equal(sd_.view(np.uint32), td.view(np.uint32))
