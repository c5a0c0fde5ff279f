"""Encoder forward timing at the per-GPU share of a 256-query batch split over 8 GPUs (development tool)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
PREC = os.environ.get("RARC_ENC_PRECISION", "fp16")   # fp16 | fp32
from oracle import cpu_ref
from rag_arc_amd.encapsulation.embeddings.hip_bert import HipBertEncoder
for name, (H, Ly, hd, I), cases in [("bge-large", (1024, 24, 16, 4096), [(32, 32), (64, 32), (128, 32), (256, 32), (256, 64)]),
                                    ("bge-base", (768, 12, 12, 3072), [(32, 32), (256, 32)])]:
    sd = cpu_ref.random_bert_state_dict(H, Ly, hd, I, vocab=2000, max_pos=512, seed=1)
    enc = HipBertEncoder(sd, num_heads=hd, precision=PREC)
    for (B, L) in cases:
        ids = np.random.default_rng(0).integers(1, 2000, (B, L)).astype(np.int32)
        for _ in range(2): enc.forward(ids)
        torch.cuda.synchronize(); t = time.perf_counter(); R = 5
        for _ in range(R): enc.forward(ids)
        torch.cuda.synchronize(); dt = (time.perf_counter() - t) / R
        params = Ly * (4 * H * H + 2 * H * I)
        print(f"[{PREC}] {name} {B}x{L}: {dt*1e3:8.2f} ms  ({2*params*B*L/dt/1e12:.0f} TF/s on GEMM flops, {B/dt:.0f} seq/s)")
