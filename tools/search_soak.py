"""Soak: the same batch searched many times over one shard, every answer compared bit for bit with the first
(a race in the scan's counted waits or in the finalize would show as a differing id or score).  Development tool."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from rag_arc_amd.hip import binding as B
from rag_arc_amd.hip.engine import FlatIndexF16
lib = B.load_library()
rows = int(os.environ.get("SOAK_ROWS", "12500000")); reps = int(os.environ.get("SOAK_REPS", "300"))
for storage, dim, n in (("f16", 768, rows), ("f8", 1024, rows // 2)):
    idx = bench.build_index(torch, lib, B, FlatIndexF16, 0, dim, 0, n, storage=storage)
    q = torch.empty((256, dim), dtype=torch.float32, device="cuda")
    B.check(lib.rarc_synth_rows_f32(q.data_ptr(), dim, dim, 0, 256, 4321, 0), "synth")
    bad = 0
    for k in (10, 100):
        i0, s0 = idx.search_device(q, k)
        i0, s0 = i0.clone(), s0.clone()
        for it in range(reps):
            i1, s1 = idx.search_device(q, k)
            if not (torch.equal(i0, i1) and torch.equal(s0.view(torch.int32), s1.view(torch.int32))):
                bad += 1
                print(f"MISMATCH storage={storage} k={k} it={it}")
        print(f"{storage} {n}x{dim} k={k}: {reps} searches identical, repaired={len(idx.last_repaired)}" if not bad else "...")
    del idx
    torch.cuda.empty_cache()
# round 6: the small-shard path — two pipelined contexts on two streams, rarc_search_batch, idle waves at nq < 256 — and the host-direct
# answers of search(): 1M x 768, search_async two in flight, every answer compared with the first; then nq = 1 / 40 / 200 through search()
idx = bench.build_index(torch, lib, B, FlatIndexF16, 0, 768, 0, 1_000_000)
q = torch.empty((256, 768), dtype=torch.float32, device="cuda")
B.check(lib.rarc_synth_rows_f32(q.data_ptr(), 768, 768, 0, 256, 4321, 0), "synth")
i0, s0 = idx.search_device(q, 100)
pending, n_bad = [], 0
for it in range(3 * reps):
    pending.append(idx.search_async(q, 100))
    if len(pending) > 2:
        i1, s1 = pending.pop(0).result()
        n_bad += not (torch.equal(i0, i1) and torch.equal(s0.view(torch.int32), s1.view(torch.int32)))
for h in pending:
    i1, s1 = h.result()
    n_bad += not (torch.equal(i0, i1) and torch.equal(s0.view(torch.int32), s1.view(torch.int32)))
print(f"f16 1000000x768 k=100, two pipelined contexts: {3 * reps} searches, {n_bad} differ from the single-context answer")
bad += n_bad
i0h, s0h = i0.cpu().numpy(), s0.cpu().numpy()
for nq in (1, 40, 200):
    n_bad = 0
    for it in range(reps):
        D, I = idx.search(q[:nq], 100)
        n_bad += not ((I == i0h[:nq]).all() and (D.view("uint32") == s0h[:nq].view("uint32")).all())
    print(f"f16 1000000x768 k=100, search() of {nq} queries (idle waves, answer written to pinned memory by the finalize): {reps} searches, {n_bad} differ")
    bad += n_bad
print("soak:", "clean" if not bad else "MISMATCHES")
