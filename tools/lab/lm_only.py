"""The reranker LM forward alone (Qwen3-Reranker-0.6B geometry, seeded weights) for rocprofv3 passes:
PROBE_PAIRS prompts x PROBE_LEN tokens, PROBE_REPS calls."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
import bench
dev = torch.device("cuda", 0)
lm, _ = bench.build_reranker_lm(torch, dev, 0, want_host=False)
n, L, reps = int(os.environ.get("PROBE_PAIRS", 640)), int(os.environ.get("PROBE_LEN", 256)), int(os.environ.get("PROBE_REPS", 3))
g = torch.Generator(device=dev); g.manual_seed(1)
ids = torch.randint(10, bench.LM_GEOM["V"], (n, L), generator=g, device=dev).int()
start = torch.randint(0, L // 3, (n,), generator=g, device=dev).int()
lm.yes_no_logits_device(ids, start, 1, 2); torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(reps): z = lm.yes_no_logits_device(ids, start, 1, 2)
torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / reps
G = bench.LM_GEOM
per_tok = G["LAYERS"] * 2.0 * (G["H"] * (G["NQ"] + 2 * G["NKV"]) * G["DH"] + G["NQ"] * G["DH"] * G["H"] + 3 * G["H"] * G["I"])
print(f"LM forward {n} x {L} tokens: {dt*1e3:.2f} ms, {n/dt:.0f} pairs/s, {per_tok*n*L/dt/1e12:.0f} TF/s on projection flops (padded tokens)")
