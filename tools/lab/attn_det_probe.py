"""run-to-run determinism of the two attention kernels (development probe)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from oracle import cpu_ref as oracle
from rag_arc_amd.core.rerank import HipCausalLM
H, LAYERS, NQ, NKV, DH, I, n, L = 1024, 2, 16, 8, 128, 3072, 8, 160
sd = oracle.random_qwen3_state_dict(H, LAYERS, NQ, NKV, DH, I, vocab=1000, seed=H + L)
lm = HipCausalLM(sd, NQ, NKV, DH, rms_norm_eps=1e-6, rope_theta=1e6)
dev = lm.device
rng = np.random.default_rng(7)
ids = rng.integers(5, 1000, (n, L)).astype(np.int32)
start = np.array([0, L - 1, 7, L // 2, 33, 1, L - 40, 64], np.int32)
for r in range(n): ids[r, :start[r]] = 0
f = lambda: lm.yes_no_logits_device(torch.from_numpy(ids).to(dev), torch.from_numpy(start).to(dev), 11, 42).float().cpu().numpy()
for mode in ("stream", "resident"):
    if mode == "stream": os.environ["RARC_LM_ATTN"] = "stream"
    else: os.environ.pop("RARC_LM_ATTN", None)
    outs = [f() for _ in range(6)]
    print(mode, "max run-to-run diff:", max(np.abs(o - outs[0]).max() for o in outs), outs[0][:, 0].round(4).tolist())
