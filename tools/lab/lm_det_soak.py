"""Determinism soak of the LM forward (folded-norm path: row-scale, SwiGLU, residual + sums-of-squares epilogues on the 256-row tile GEMMs,
resident attention): the same batch many times, every pair of logits compared bit for bit with the first run.  Development tool."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from oracle import cpu_ref as oracle
from rag_arc_amd.core.rerank import HipCausalLM
H, LAYERS, NQ, NKV, DH, I, V = 1024, 4, 16, 8, 128, 3072, 3000
sd = oracle.random_qwen3_state_dict(H, LAYERS, NQ, NKV, DH, I, vocab=V, seed=7)
lm = HipCausalLM(sd, NQ, NKV, DH, rms_norm_eps=1e-6, rope_theta=1e6)
rng = np.random.default_rng(3)
n, L = int(os.environ.get("SOAK_SEQS", 640)), int(os.environ.get("SOAK_LEN", 200))
ids = rng.integers(5, V, (n, L)).astype(np.int32)
start = rng.integers(0, L // 2, n).astype(np.int32)
for r in range(n): ids[r, :start[r]] = 0
d_ids, d_start = torch.from_numpy(ids).to(lm.device), torch.from_numpy(start).to(lm.device)
noise = torch.empty(256 << 20, dtype=torch.uint8, device=lm.device); side = torch.cuda.Stream()
first, bad = None, 0
for it in range(int(os.environ.get("SOAK_REPS", 60))):
    if it % 2:
        with torch.cuda.stream(side): noise.add_(1)      # HBM traffic from another stream under the forward
    out = lm.yes_no_logits_device(d_ids, d_start, 11, 42).clone()
    torch.cuda.synchronize()
    if first is None: first = out
    elif not torch.equal(out.view(torch.int16), first.view(torch.int16)):
        bad += 1; print(f"MISMATCH at repetition {it}: {(out != first).sum().item()} logits differ")
assert bool(torch.isfinite(first.float()).all())
print("LM soak:", "clean" if not bad else f"{bad} MISMATCHES", f"({n} x {L} tokens, {LAYERS} layers)")
