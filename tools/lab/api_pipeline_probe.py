#!/usr/bin/env python3
"""Where a 2048-query batch_invoke spends its time, chunk by chunk (launch / wait + copy-out / mapping), at PROBE_ROWS rows."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import bench
from rag_arc_amd.hip import binding as B
from rag_arc_amd.hip.engine import FlatIndexF16
from rag_arc_amd.encapsulation.database.vector_db.docstore import ColumnarDocstore
from rag_arc_amd.encapsulation.database.vector_db.hip_flat import HipFlatVectorStore

N = int(os.environ.get("PROBE_ROWS", 100_000_000)); D = 768; K = 100
lib = B.load_library(); dev = torch.device("cuda", 0)
idx = bench.build_index(torch, lib, B, FlatIndexF16, 0, D, 0, N)
store = HipFlatVectorStore(embedding=None).adopt(idx, ColumnarDocstore.decimal(N))
q = torch.zeros((2048, D), dtype=torch.float32, device=dev)
B.check(lib.rarc_synth_rows_f32(q.data_ptr(), D, D, 0, 2048, 4321, 0))
for rep in range(2):
    torch.cuda.synchronize(); t_all = time.perf_counter()
    marks, pending = [], None
    out = []
    for s0 in range(0, 2048, 256):
        t0 = time.perf_counter(); nxt = idx.search_async(q[s0:s0 + 256], K, to_host=True); t1 = time.perf_counter()
        if pending is not None:
            sc, rows = pending.host(); t2 = time.perf_counter()
            out.extend(store._map_batch(sc, rows, False)); t3 = time.perf_counter()
            marks.append((t1 - t0, t2 - t1, t3 - t2))
        pending = nxt
    sc, rows = pending.host(); out.extend(store._map_batch(sc, rows, False))
    torch.cuda.synchronize(); total = time.perf_counter() - t_all
    print(f"rep {rep}: total {total*1e3:.1f} ms; per chunk (launch, wait, map) ms:", [tuple(round(x * 1e3, 2) for x in m) for m in marks])
import cProfile, pstats, gc
for rep in range(2):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    ans = store.batch_search_by_vector(q, K)
    torch.cuda.synchronize(); print(f"store.batch_search_by_vector(2048): {(time.perf_counter() - t0) * 1e3:.1f} ms")
    del ans
pr = cProfile.Profile(); pr.enable()
ans = store.batch_search_by_vector(q, K)
pr.disable(); pstats.Stats(pr).sort_stats("tottime").print_stats(10)
