"""rarc_search_wide against the shard size (development tool): ms per 256-query batch at d = 1536, k = 100, n = 100k .. 10M,
next to what the score GEMM alone would take at ~0.95 PF."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from rag_arc_amd.hip import binding as B
from rag_arc_amd.hip.engine import FlatIndexF16
lib = B.load_library(); dev = torch.device("cuda", 0)
d, k = int(os.environ.get("PROBE_DIM", 1536)), int(os.environ.get("PROBE_K", 100))
q = torch.empty((256, d), dtype=torch.float32, device=dev)
B.check(lib.rarc_synth_rows_f32(q.data_ptr(), d, d, 0, 256, 4321, 0))
for n in [int(v) for v in os.environ.get("PROBE_SIZES", "100000,300000,1000000,3000000,10000000").split(",")]:
    buf = torch.empty((n, B.padded_dim(d)), dtype=torch.float16, device=dev)
    B.check(lib.rarc_synth_rows_f16(buf.data_ptr(), buf.shape[1], d, 0, n, 1234, 0))
    idx = FlatIndexF16(d, metric="cosine", growable=False)
    idx.add_rows_f16(buf, 1.001)
    for _ in range(3): idx.search_device(q, k)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    reps = 20 if n <= 1_000_000 else 8
    for _ in range(reps): idx.search_device(q, k)
    torch.cuda.synchronize(); ms = (time.perf_counter() - t0) / reps * 1e3
    print(f"WIDE n={n:9d} d={d} k={k}: {ms:7.3f} ms per batch   (GEMM alone at 0.95 PF: {2.0 * 256 * n * buf.shape[1] / 0.95e15 * 1e3:6.3f} ms)")
    del idx, buf
