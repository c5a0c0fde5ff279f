"""Six bge-large forwards of 256 x 32 tokens (seeded weights on the device) for rocprofv3 passes."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from rag_arc_amd.encapsulation.embeddings.hip_bert import HipBertEncoder
H, HEADS, FFN, LAYERS, VOCAB = 1024, 16, 4096, 24, 30522
L, NQ = int(os.environ.get("PROBE_TOKENS", 32)), int(os.environ.get("PROBE_SEQS", 256))
dev = torch.device("cuda", 0); g = torch.Generator(device=dev); g.manual_seed(5)
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
enc = HipBertEncoder(sd, num_heads=HEADS, precision=os.environ.get("RARC_ENC_PRECISION", "fp16"))
tok = torch.randint(1, VOCAB, (NQ, L), generator=g, device=dev).int()
lens = torch.full((NQ,), L, dtype=torch.int32, device=dev)
import time
for _ in range(2):
    enc.forward_device(tok, lens)
torch.cuda.synchronize(); t0 = time.time()
for _ in range(int(os.environ.get("PROBE_ITERS", 4))):
    enc.forward_device(tok, lens)
torch.cuda.synchronize()
print(f"ENC seqs={NQ} tokens={L} precision={enc.precision}: {(time.time() - t0) / int(os.environ.get('PROBE_ITERS', 4)) * 1e3:.3f} ms per forward (wall, {os.environ.get('PROBE_ITERS', 4)} back to back)")
