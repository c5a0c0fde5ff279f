"""Does running alternate batches on two streams (two query blocks / workspaces over the same rows) hide the small kernels
around the scan?  1M x 768, batch 256, k = 100 (development probe)."""
import copy, os, sys, threading, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import bench
from rag_arc_amd.hip import binding as B
from rag_arc_amd.hip.engine import FlatIndexF16
lib = B.load_library(); dev = torch.device("cuda", 0)
N = int(os.environ.get("PROBE_ROWS", 1_000_000)); D = 768; K = 100
idx = bench.build_index(torch, lib, B, FlatIndexF16, 0, D, 0, N, storage="f16")
q = torch.empty((256, D), dtype=torch.float32, device=dev)
B.check(lib.rarc_synth_rows_f32(q.data_ptr(), D, D, 0, 256, 4321, 0))
twin = copy.copy(idx); twin._ws = None; twin._qbuf = None; twin._lock = threading.RLock() if isinstance(idx._lock, type(threading.RLock())) else threading.Lock()
def run(objs, streams, steps):
    pend = [None] * len(objs)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for i in range(steps):
        s = i % len(objs)
        if pend[s] is not None: pend[s].result()
        with torch.cuda.stream(streams[s]):
            pend[s] = objs[s].search_async(q, K)
    for p in pend:
        if p is not None: p.result()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / steps
s0 = torch.cuda.current_stream(); sa, sb = torch.cuda.Stream(), torch.cuda.Stream()
ref = idx.search_device(q, K)
for name, objs, streams in (("one stream, two in flight", [idx, idx], [s0, s0]), ("two streams", [idx, twin], [sa, sb]), ("one stream again", [idx, idx], [s0, s0]), ("two streams again", [idx, twin], [sa, sb])):
    run(objs, streams, 20)
    dt = run(objs, streams, 400)
    print(f"PROBE {name}: {dt*1e3:.4f} ms per batch, {256/dt:.0f} q/s")
i2, s2 = twin.search_device(q, K)
print("PROBE twin answers equal:", bool(torch.equal(ref[0], i2) and torch.equal(ref[1].view(torch.int32), s2.view(torch.int32))))
