#!/usr/bin/env python3
"""What the cyclic collector costs a 256 x 100 answer (development tool): the mapping call alone, the young collection that
follows it, and the same after gc.freeze(); PROBE_ROWS rows, columnar docstore."""
import gc, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
import bench
from rag_arc_amd.hip import binding as B
from rag_arc_amd.hip.engine import FlatIndexF16
from rag_arc_amd.encapsulation.database.vector_db.docstore import ColumnarDocstore
from rag_arc_amd.encapsulation.database.vector_db.hip_flat import HipFlatVectorStore

N = int(os.environ.get("PROBE_ROWS", 10_000_000)); D = 768; K = 100
lib = B.load_library(); dev = torch.device("cuda", 0)
idx = bench.build_index(torch, lib, B, FlatIndexF16, 0, D, 0, N)
store = HipFlatVectorStore(embedding=None).adopt(idx, ColumnarDocstore.decimal(N))
q = torch.zeros((256, D), dtype=torch.float32, device=dev)
B.check(lib.rarc_synth_rows_f32(q.data_ptr(), D, D, 0, 256, 4321, 0))
sc, rows = idx.search_async(q, K, to_host=True).host()
sc, rows = np.array(sc), np.array(rows)
def one(label):
    ts = []
    for _ in range(12):
        c0 = gc.get_count(); s0 = [g["collections"] for g in gc.get_stats()]
        t0 = time.perf_counter(); ans = store._map_batch(sc, rows, False); t1 = time.perf_counter()
        x = [[] for _ in range(8)]                      # the first tracked allocations after the call: trigger what is due
        t2 = time.perf_counter()
        s1 = [g["collections"] for g in gc.get_stats()]
        ts.append((t1 - t0, t2 - t1, tuple(b - a for a, b in zip(s0, s1)), c0))
        del ans
    print(label, "tracked objects in the process:", len(gc.get_objects()))
    for m, g, colls, c0 in ts[2:]:
        print(f"   map {m * 1e3:6.2f} ms   next allocations {g * 1e3:6.2f} ms   collections run (gen0, gen1, gen2) {colls}   count before {c0}")
one("default:")
gc.collect(); gc.freeze()
one("after gc.freeze():")
gc.unfreeze(); gc.disable()
one("gc disabled:")
