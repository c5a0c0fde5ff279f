"""Soak of the fp32-class encoder forward (split-operand GEMMs + split-operand attention): one ragged batch through a 4-layer
bge-base-geometry model SOAK_REPS times while a side stream keeps HBM busy; every embedding compared bit for bit with the first."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from oracle import cpu_ref
from rag_arc_amd.encapsulation.embeddings.hip_bert import HipBertEncoder
reps = int(os.environ.get("SOAK_REPS", "300"))
H, heads, I, layers = 768, 12, 3072, 4
L, NSEQ = int(os.environ.get("SOAK_LEN", 192)), int(os.environ.get("SOAK_SEQS", 48))   # 128 x 256: FFN1 takes the fused-GELU GEMM
sd = cpu_ref.random_bert_state_dict(H, layers, heads, I, vocab=2000, max_pos=L, seed=13)
rng = np.random.default_rng(13)
lens = rng.integers(1, L + 1, NSEQ).astype(np.int32); lens[0] = L
for i_, v_ in ((1, 1), (2, 32), (3, 33)):      # (a single query or a pair — the query path's shapes — has no such rows)
    if i_ < NSEQ:
        lens[i_] = min(v_, L)
ids = rng.integers(1, 2000, (NSEQ, L)).astype(np.int32)
for r, l in enumerate(lens):
    ids[r, l:] = 0
enc = HipBertEncoder(sd, num_heads=heads, pooling="mean", precision="fp32")
tok = torch.from_numpy(ids).cuda(); ln = torch.from_numpy(lens).cuda()
first = enc.forward_device(tok, ln).clone()
if NSEQ * L <= 10000:
    w64 = cpu_ref.bert_forward_f32(sd, ids, lens, heads, normalize=True, pooling="mean", dtype=np.float64)
    print("max ||e - e64||:", float(np.linalg.norm(first.cpu().numpy() - w64, axis=1).max()))
side = torch.cuda.Stream(); junk = torch.empty(1 << 28, dtype=torch.float32, device="cuda")
bad = 0
for it in range(reps):
    with torch.cuda.stream(side):
        junk.mul_(1.0001)
    e = enc.forward_device(tok, ln)
    if not torch.equal(e.view(torch.int32), first.view(torch.int32)):
        bad += 1
        print("MISMATCH at", it, float((e - first).abs().max()))
torch.cuda.synchronize()
print(f"encoder soak ({NSEQ} x {L} tokens): {reps} forwards,", "all identical" if not bad else f"{bad} MISMATCHES")
