"""where do the two attention kernels' context rows differ (development probe; library built with -DLM_DUMP_CTX)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from oracle import cpu_ref as oracle
from rag_arc_amd.core.rerank import HipCausalLM
H, LAYERS, NQ, NKV, DH, I, n, L = 1024, 1, 16, 8, 128, 3072, 8, 160
sd = oracle.random_qwen3_state_dict(H, LAYERS, NQ, NKV, DH, I, vocab=1000, seed=H + L)
lm = HipCausalLM(sd, NQ, NKV, DH, rms_norm_eps=1e-6, rope_theta=1e6)
dev = lm.device
rng = np.random.default_rng(L)
ids = rng.integers(5, 1000, (n, L)).astype(np.int32)
start = np.array([0, L - 1, 7, L // 2, 33, 1, L - 40, 64], np.int32)
for r in range(n): ids[r, :start[r]] = 0
f = lambda: lm.yes_no_logits_device(torch.from_numpy(ids).to(dev), torch.from_numpy(start).to(dev), 11, 42).float().cpu().numpy()
out = {}
for mode in ("stream", "resident"):
    if mode == "stream": os.environ["RARC_LM_ATTN"] = "stream"
    else: os.environ.pop("RARC_LM_ATTN", None)
    os.environ["RARC_LM_DUMP_CTX"] = f"/tmp/ctx_{mode}.bin"
    z = f()
    out[mode] = np.fromfile(f"/tmp/ctx_{mode}.bin", dtype=np.float16).reshape(n, L, NQ, DH).astype(np.float32)
d = np.abs(out["stream"] - out["resident"])
print("max diff", d.max(), "elements differing", int((d > 0).sum()), "of", d.size)
idx = np.argwhere(d.max(axis=3) > 0)
print("differing (seq, pos, head) count", len(idx))
for s_, p_, h_ in idx[:40]:
    print("seq", s_, "start", start[s_], "pos", p_, "qb", p_ // 32, "head", h_, "max", d[s_, p_, h_].max(), "n", int((d[s_, p_, h_] > 0).sum()))
