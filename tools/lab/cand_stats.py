"""Candidate statistics of the int8-prefilter scan on the GPU box: R, eps8, candidates per query, fullest
(workgroup, query) segment, flagged queries, and time per batch.

    python tools/lab/cand_stats.py ROWS DIM STORAGE [K] [QUERY_KIND]      STORAGE: f16 | f8 | f32; QUERY_KIND: synth | clustered
"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from rag_arc_amd.hip import binding as B
from rag_arc_amd.hip.engine import FlatIndexF16

N, D, ST = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3]
K = int(sys.argv[4]) if len(sys.argv) > 4 else 100
KIND = sys.argv[5] if len(sys.argv) > 5 else "synth"
NQ = 256
lib = B.load_library()
dev = torch.device("cuda", 0)
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench
idx = bench.build_index(torch, lib, B, FlatIndexF16, 0, D, 0, N, storage=ST, scan="q8")
q = torch.zeros((NQ, D), dtype=torch.float32, device=dev)
B.check(lib.rarc_synth_rows_f32(q.data_ptr(), D, D, 0, NQ, 4321, 0))
if KIND == "collapsed":   # queries that are near-copies of one direction (what a seeded random encoder emits)
    q = q[:1] + 0.05 * q
torch.cuda.synchronize()
ids, sc = idx.search_device(q, K, repair=False)
torch.cuda.synchronize()
st = idx.last_status.cpu().numpy()
ws = idx._ws
HIST = 8192; CNT2 = HIST + 256 * 256 * 4
cnt2 = ws[CNT2:CNT2 + 256 * 256 * 4].view(torch.int32).view(256, 256).cpu().numpy()
cnt = cnt2.sum(0)
qb = idx._qbuf["qblock"]
n = 256 * idx.d_pad
eps16 = qb[n * 7: n * 7 + 1024].view(torch.float32).cpu().numpy()
eps8 = qb[n * 7 + 1024: n * 7 + 2048].view(torch.float32).cpu().numpy()
R = float(idx._qmeta[0].item())
sigma = 1.0 / np.sqrt(D)
print(f"CAND {N}x{D} {ST} k={K} {KIND}: R={R:.5f} eps8 mean={eps8.mean():.5f} ({eps8.mean() / sigma:.3f} sigma) eps16 mean={eps16.mean():.2e} "
      f"cand/query min={cnt.min()} mean={cnt.mean():.0f} max={cnt.max()} fullest segment={cnt2.max()} (cap {idx._cap_eff // 256}) "
      f"flagged={int((st != 0).sum())} kth score mean={float(sc[:, -1].mean()):.4f} ({float(sc[:, -1].mean()) / sigma:.2f} sigma)")
for it in range(2):
    torch.cuda.synchronize(); t = time.time()
    for _ in range(5):
        idx.search_device(q, K, repair=False)
    torch.cuda.synchronize(); dt = (time.time() - t) / 5
print(f"CAND   {dt * 1e3:.3f} ms/batch -> {NQ / dt:.0f} q/s")
