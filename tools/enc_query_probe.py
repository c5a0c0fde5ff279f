"""Latency of the fp32-class forward at query sizes (1 / 2 / 4 sequences of 32 tokens): the query path's weight stream against the
tile kernels (RARC_E32_QUERY=0), bge-large and bge-base geometry, seeded weights.  PROBE_ITERS forwards back to back, wall clock
around a synchronize (what a caller of embed_query waits for is one forward: also printed from HIP events per forward)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from rag_arc_amd.encapsulation.embeddings.hip_bert import HipBertEncoder

dev = torch.device("cuda", 0)
ITERS = int(os.environ.get("PROBE_ITERS", 50))
GEOS = {"bge-large": (1024, 16, 4096, 24), "bge-base": (768, 12, 3072, 12), "bge-small": (384, 12, 1536, 12)}


def build(H, HEADS, FFN, LAYERS, VOCAB=30522):
    g = torch.Generator(device=dev); g.manual_seed(5)
    rnd = lambda *s: torch.randn(s, generator=g, device=dev) * 0.05
    sd = {"embeddings.word_embeddings.weight": rnd(VOCAB, H), "embeddings.position_embeddings.weight": rnd(512, H),
          "embeddings.token_type_embeddings.weight": rnd(2, H), "embeddings.LayerNorm.weight": 1.0 + rnd(H), "embeddings.LayerNorm.bias": rnd(H)}
    for i in range(LAYERS):
        p = f"encoder.layer.{i}."
        for nm, (o, c) in {"attention.self.query": (H, H), "attention.self.key": (H, H), "attention.self.value": (H, H),
                           "attention.output.dense": (H, H), "intermediate.dense": (FFN, H), "output.dense": (H, FFN)}.items():
            sd[p + nm + ".weight"], sd[p + nm + ".bias"] = rnd(o, c), rnd(o)
        for nm in ("attention.output.LayerNorm", "output.LayerNorm"):
            sd[p + nm + ".weight"], sd[p + nm + ".bias"] = 1.0 + rnd(H), rnd(H)
    return HipBertEncoder(sd, num_heads=HEADS, precision="fp32"), g


def timed(enc, tok, lens):
    for _ in range(3):
        enc.forward_device(tok, lens)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(ITERS):
        enc.forward_device(tok, lens)
    torch.cuda.synchronize()
    wall = (time.perf_counter() - t0) / ITERS
    one = []
    for _ in range(10):        # one forward at a time: launch -> done, as a caller of embed_query sees it
        torch.cuda.synchronize(); t1 = time.perf_counter()
        enc.forward_device(tok, lens)
        torch.cuda.synchronize(); one.append(time.perf_counter() - t1)
    return wall * 1e3, sorted(one)[len(one) // 2] * 1e3


for name in os.environ.get("PROBE_GEOS", "bge-large,bge-base").split(","):
    enc, g = build(*GEOS[name])
    for n, L in [(1, 32), (2, 32), (4, 32)] + ([(1, 64), (1, 128), (2, 64)] if os.environ.get("PROBE_LONG") else []):
        tok = torch.randint(1, 30522, (n, L), generator=g, device=dev).int()
        lens = torch.full((n,), L, dtype=torch.int32, device=dev)
        q_wall, q_one = timed(enc, tok, lens)
        # the tile kernels need a multiple of 128 tokens: the same sequences padded to 4 x 32 (what forward() did before round 6)
        n128 = 128 // L          # the tile kernels take multiples of 128 tokens
        pad = torch.cat([tok, torch.zeros((n128 - n, L), dtype=torch.int32, device=dev)]) if n < n128 else tok
        plens = torch.cat([lens, torch.ones(n128 - n, dtype=torch.int32, device=dev)]) if n < n128 else lens
        os.environ["RARC_E32_QUERY"] = "0"
        t_wall, t_one = timed(enc, pad, plens)
        del os.environ["RARC_E32_QUERY"]
        a, b = enc.forward_device(tok, lens), None
        os.environ["RARC_E32_QUERY"] = "0"; b = enc.forward_device(pad, plens)[:n]; del os.environ["RARC_E32_QUERY"]
        print(f"ENCQ {name} {n} x {L} tokens: query path {q_wall:.3f} ms back to back, {q_one:.3f} ms alone | tile kernels ({n128} x {L}) "
              f"{t_wall:.3f} / {t_one:.3f} ms | max |diff| {float((a - b).abs().max()):.2e}")
    del enc
    torch.cuda.empty_cache()
