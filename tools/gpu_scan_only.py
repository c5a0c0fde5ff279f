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
for _ in range(int(os.environ.get("PROBE_ITERS", 3))):
    idx.search_device(q, K, repair=False)
torch.cuda.synchronize()
