"""Runs a few searches at C2 shape (for rocprofv3 --pmc passes)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from rag_arc_amd.hip import binding as B
from rag_arc_amd.hip.engine import FlatIndexF16
N = int(os.environ.get("PROBE_ROWS", 1_000_000)); D = 768; NQ = 256; K = 100
lib = B.load_library(); dev = torch.device("cuda", 0)
rows = torch.zeros((N, D), dtype=torch.float16, device=dev)
B.check(lib.rarc_synth_rows_f16(rows.data_ptr(), D, D, 0, N, 1234, 0))
q = torch.zeros((NQ, D), dtype=torch.float32, device=dev)
B.check(lib.rarc_synth_rows_f32(q.data_ptr(), D, D, 0, NQ, 4321, 0))
idx = FlatIndexF16(D, "cosine"); idx.add_rows_f16(rows, 1.001)
for _ in range(int(os.environ.get("PROBE_ITERS", 3))):
    idx.search_device(q, K, repair=False)
torch.cuda.synchronize()
