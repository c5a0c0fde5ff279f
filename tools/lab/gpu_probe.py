"""Scratch probe for the GPU box: timings + candidate statistics of the scan at C2 shape."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from rag_arc_amd.hip import binding as B
from rag_arc_amd.hip.engine import FlatIndexF16

N = int(os.environ.get("PROBE_ROWS", 1_000_000)); D = 768; NQ = 256; K = 100
lib = B.load_library()
dev = torch.device("cuda", 0)
rows = torch.zeros((N, D), dtype=torch.float16, device=dev)
B.check(lib.rarc_synth_rows_f16(rows.data_ptr(), D, D, 0, N, 1234, 0))
q = torch.zeros((NQ, D), dtype=torch.float32, device=dev)
B.check(lib.rarc_synth_rows_f32(q.data_ptr(), D, D, 0, NQ, 4321, 0))
torch.cuda.synchronize()
idx = FlatIndexF16(D, "cosine")
idx.add_rows_f16(rows, 1.001)
ids, sc = idx.search_device(q, K, repair=False)
torch.cuda.synchronize()
st = idx.last_status.cpu().numpy()
print("status: uncertain %d overflow %d" % (int((st & 1).astype(bool).sum()), int((st & 2).astype(bool).sum())))
print("status nonzero:", int((idx.last_status != 0).sum()), "repaired:", getattr(idx, "last_repaired", None))
ws = idx._ws
HIST = 8192; CNT2 = HIST + 256 * 256 * 4
cnt2 = ws[CNT2:CNT2 + 256 * 256 * 4].view(torch.int32).view(256, 256).cpu().numpy()
cnt = cnt2.sum(0); print("max per (wg,query) segment:", cnt2.max())
thr = ws[0:1024].view(torch.float32).cpu().numpy()
print("cand count per query: min %d mean %.0f max %d" % (cnt.min(), cnt.mean(), cnt.max()))
print("final thr: min %.4f mean %.4f max %.4f ; kth score mean %.4f" % (thr.min(), thr.mean(), thr.max(), float(sc[:, -1].mean())))
for it in range(3):
    torch.cuda.synchronize(); t = time.time()
    for _ in range(10):
        ids, sc = idx.search_device(q, K, repair=False)
    torch.cuda.synchronize(); dt = (time.time() - t) / 10
    print("search: %.1f us/batch  -> %.0f q/s, %.2f TB/s algorithmic" % (dt * 1e6, NQ / dt, N * D * 2 / dt / 1e12))
# exactness property at full size: verify a few queries
ids, sc = idx.search_device(q, K, repair=False)
eps = idx._qbuf["eps"].cpu().numpy(); print("eps: mean %.2e max %.2e" % (eps.mean(), eps.max()))
hist = ws[HIST:HIST + 256 * 256 * 4].view(torch.int32).view(256, 256).cpu().numpy()
print("hist q0 nonzero bins:", [(int(b), int(c)) for b, c in enumerate(hist[0]) if c][-12:])
for qi in (0, 100, 255):
    print("verify q%d: rows beating k-th =" % qi, idx.verify_query(q, qi, ids, sc))
