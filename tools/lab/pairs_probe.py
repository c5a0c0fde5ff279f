"""Where the time of similar_pairs goes at n = 100k x 1024 (development tool): whole call vs the library call alone."""
import ctypes, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from rag_arc_amd.hip import binding as B
from rag_arc_amd.encapsulation.database.graph_db import similar_pairs
lib = B.load_library(); dev = torch.device("cuda", 0)
n, d = int(os.environ.get("PROBE_N", 100_000)), int(os.environ.get("PROBE_DIM", 1024))
x = torch.empty((n, d), dtype=torch.float32, device=dev)
B.check(lib.rarc_synth_rows_f32(x.data_ptr(), d, d, 0, n, 777, 0))
x[1::50] = x[0::50][: len(x[1::50])] * 1.5 + 0.001 * torch.randn_like(x[1::50])
for _ in range(2): similar_pairs(x, 0.95)
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(10): p = similar_pairs(x, 0.95)
torch.cuda.synchronize(); print(f"similar_pairs whole call: {(time.perf_counter() - t0) / 10 * 1e3:.2f} ms, {len(p)} pairs")
cap, out_cap = 256, 4 * n
ws = torch.empty(int(lib.rarc_similar_pairs_workspace_bytes(n, d, cap)), dtype=torch.uint8, device=dev)
pairs = torch.empty((out_cap, 2), dtype=torch.int64, device=dev); scores = torch.empty(out_cap, dtype=torch.float64, device=dev)
count = torch.zeros(1, dtype=torch.int64, device=dev); flags = torch.zeros(1, dtype=torch.int32, device=dev)
def call():
    B.check(lib.rarc_similar_pairs(x.data_ptr(), x.stride(0), n, d, ctypes.c_double(0.95), ws.data_ptr(), ws.numel(), cap, pairs.data_ptr(),
                                   scores.data_ptr(), out_cap, count.data_ptr(), flags.data_ptr(), torch.cuda.current_stream().cuda_stream))
for _ in range(2): call()
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(10): call()
torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 10
print(f"rarc_similar_pairs alone: {dt * 1e3:.2f} ms, count {int(count.item())}, flags {int(flags.item())}")
t0 = time.perf_counter(); w2 = torch.empty(ws.numel(), dtype=torch.uint8, device=dev); torch.cuda.synchronize(); print(f"torch.empty(ws {ws.numel() >> 20} MiB): {(time.perf_counter() - t0) * 1e3:.2f} ms")
import cProfile, pstats
pr = cProfile.Profile(); pr.enable()
for _ in range(10): similar_pairs(x, 0.95)
pr.disable(); pstats.Stats(pr).sort_stats("tottime").print_stats(8)
# many pairs: 100k entities in 10k groups of 10 near-identical ones -> 450k pairs
g = torch.Generator(device=dev); g.manual_seed(7)
base = torch.randn((n // 10, d), generator=g, device=dev)
xd = base.repeat_interleave(10, dim=0) + 0.01 * torch.randn((n // 10 * 10, d), generator=g, device=dev)
for _ in range(2): pd = similar_pairs(xd, 0.95)
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(5): pd = similar_pairs(xd, 0.95, as_arrays=True)[0]
torch.cuda.synchronize(); print(f"groups of 10: {len(pd)} pairs, {(time.perf_counter() - t0) / 5 * 1e3:.1f} ms per call (as arrays)")
