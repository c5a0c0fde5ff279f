import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from rag_arc_amd.core.utils import HipRRFusion
f = HipRRFusion()
for nq, n in ((256, 100), (256, 10), (1, 100), (256, 400), (32, 100)):
    keys = torch.randint(0, 1000, (nq, 2, n), device="cuda")
    lens = torch.full((nq, 2), n, dtype=torch.int32, device="cuda")
    for _ in range(3): f.fuse_ids(keys, lens, min(100, n))
    torch.cuda.synchronize(); e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    from rag_arc_amd.hip import binding as B
    lib = B.load_library()
    out_k = torch.empty((nq, 100), dtype=torch.int64, device="cuda"); out_s = torch.empty((nq, 100), dtype=torch.float64, device="cuda"); out_n = torch.empty(nq, dtype=torch.int32, device="cuda")
    e0.record()
    for _ in range(20):
        lib.rarc_rrf_fuse(keys.data_ptr(), lens.data_ptr(), nq, 2, n, 60.0, min(100, n), out_k.data_ptr(), out_s.data_ptr(), out_n.data_ptr(), 0)
    e1.record(); torch.cuda.synchronize()
    print(f"RRF nq={nq} n={n}: {e0.elapsed_time(e1) / 20 * 1e3:.1f} us per call")
