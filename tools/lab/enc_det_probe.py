"""Which forwards of the fp32-class encoder differ from which (debugging aid for tools/enc_det_soak.py)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from oracle import cpu_ref
from rag_arc_amd.encapsulation.embeddings.hip_bert import HipBertEncoder
H, heads, I, layers = int(os.environ.get("SOAK_H", 768)), int(os.environ.get("SOAK_HEADS", 12)), int(os.environ.get("SOAK_I", 3072)), int(os.environ.get("SOAK_LAYERS", 4))
PREC = os.environ.get("SOAK_PRECISION", "fp32")
L, NSEQ = int(os.environ.get("SOAK_LEN", 192)), int(os.environ.get("SOAK_SEQS", 48))
sd = cpu_ref.random_bert_state_dict(H, layers, heads, I, vocab=2000, max_pos=L, seed=13)
rng = np.random.default_rng(13)
lens = rng.integers(1, L + 1, NSEQ).astype(np.int32); lens[0] = L; lens[1] = 1
ids = rng.integers(1, 2000, (NSEQ, L)).astype(np.int32)
for r, l in enumerate(lens):
    ids[r, l:] = 0
enc = HipBertEncoder(sd, num_heads=heads, pooling="mean", precision=PREC)
tok = torch.from_numpy(ids).cuda(); ln = torch.from_numpy(lens).cuda()
REPS = int(os.environ.get("PROBE_REPS", 6))
outs = [enc.forward_device(tok, ln).clone() for _ in range(REPS)]
torch.cuda.synchronize()
ref = torch.stack(outs).median(dim=0).values          # the majority answer
bad = [(i, float((o - ref).abs().max())) for i, o in enumerate(outs) if not torch.equal(o, ref)]
print(f"[{PREC} H={H} layers={layers} {NSEQ}x{L}] quiet:   {len(bad)} of {REPS} forwards differ from the majority:", [(i, f"{d:.1e}") for i, d in bad][:8])
for i, _ in bad[:3]:
    rows = torch.nonzero((outs[i] - ref).abs().amax(dim=1) > 0).flatten().tolist()
    print("   forward", i, "rows", rows[:12], "lens", [int(lens[r]) for r in rows[:12]])
side = torch.cuda.Stream(); junk = torch.empty(1 << 28, dtype=torch.float32, device="cuda")
outs2 = []
for _ in range(REPS):
    with torch.cuda.stream(side):
        junk.mul_(1.0001)
    outs2.append(enc.forward_device(tok, ln).clone())
torch.cuda.synchronize()
bad = [(i, float((o - ref).abs().max())) for i, o in enumerate(outs2) if not torch.equal(o, ref)]
print(f"traffic: {len(bad)} of {REPS} forwards differ from the majority:", [(i, f"{d:.1e}") for i, d in bad][:8])
