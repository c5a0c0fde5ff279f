"""Runs a few searches (for rocprofv3 --pmc passes).  PROBE_ROWS, PROBE_DIM, PROBE_STORAGE (f16 | f8), PROBE_ITERS."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from rag_arc_amd.hip import binding as B
from rag_arc_amd.hip.engine import FlatIndexF16
N = int(os.environ.get("PROBE_ROWS", 1_000_000)); D = int(os.environ.get("PROBE_DIM", 768)); NQ = 256; K = 100
ST = os.environ.get("PROBE_STORAGE", "f16")
lib = B.load_library(); dev = torch.device("cuda", 0)
idx = bench.build_index(torch, lib, B, FlatIndexF16, 0, D, 0, N, storage=ST)
q = torch.zeros((NQ, D), dtype=torch.float32, device=dev)
B.check(lib.rarc_synth_rows_f32(q.data_ptr(), D, D, 0, NQ, 4321, 0))
import ctypes
idx.search_device(q, K, repair=False)          # warm-up (allocations)
torch.cuda.synchronize()
iters = int(os.environ.get("PROBE_ITERS", 3))
B.check(lib.rarc_profile_begin(64 * iters + 16))
for _ in range(iters):
    idx.search_device(q, K, repair=False)
torch.cuda.synchronize()
tot, n = ctypes.c_double(0), ctypes.c_int(0)
B.check(lib.rarc_profile_end(ctypes.byref(tot), ctypes.byref(n)))
if n.value == 0:      # (rows wider than 1024 dims take rarc_search_wide: its kernels are not in the library's scan brackets)
    print(f"SCAN rows={N} dim={D} storage={ST}: wide path, {iters} searches (no scan-kernel brackets)"); sys.exit(0)
esize = {"f16": 2, "f8": 1}[ST]
gb = N * idx.d_pad * esize / 1e9
print(f"SCAN rows={N} dim={D} storage={ST} abl={os.environ.get('RARC_Q8_ABL', '0')}: {tot.value / iters:.3f} ms per scan "
      f"({n.value // iters} launches), {gb / (tot.value / iters * 1e-3) / 1e3:.2f} TB/s of stored bytes, "
      f"{2.0 * NQ * N * idx.d_pad / (tot.value / iters * 1e-3) / 1e15:.2f} POPS")
