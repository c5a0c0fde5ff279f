"""Determinism of the search paths at sizes the other soaks do not visit: 1M rows (fp16 scan, hybrid int8 path), 3M rows (cascade's
first cut), clustered 2M rows (finalize's radix select and banded rescore): the same batch REPS times, every answer bit for bit."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from rag_arc_amd.hip.engine import FlatIndexF16
REPS = int(os.environ.get("PROBE_REPS", 150))
dev = torch.device("cuda", 0)
g = torch.Generator(device=dev); g.manual_seed(11)
def check(name, idx, q, k):
    i0, s0 = idx.search_device(q, k); i0, s0 = i0.clone(), s0.clone()
    bad = 0
    for _ in range(REPS):
        i1, s1 = idx.search_device(q, k)
        bad += int(not (torch.equal(i0, i1) and torch.equal(s0.view(torch.int32), s1.view(torch.int32))))
    print(f"{name}: {REPS} searches, {bad} differ; repaired last {len(idx.last_repaired)}")
for n, d, scan, k in ((1_000_000, 768, "auto", 100), (1_000_000, 768, "q8", 100), (3_000_000, 384, "auto", 10), (1_000_000, 1024, "q8", 500)):
    idx = FlatIndexF16(d, scan=scan, capacity=n)
    for s0 in range(0, n, 500_000):
        idx.add(torch.randn((min(500_000, n - s0), d), generator=g, device=dev))
    q = torch.randn((256, d), generator=g, device=dev)
    check(f"isotropic {n}x{d} scan={scan} k={k}", idx, q, k)
    del idx; torch.cuda.empty_cache()
n, d, nc = 2_000_000, 256, 50
centers = torch.nn.functional.normalize(torch.randn((nc, d), generator=g, device=dev), dim=1)
mk = lambda m: centers[torch.randint(0, nc, (m,), generator=g, device=dev)] + 0.3 / d ** 0.5 * torch.randn((m, d), generator=g, device=dev)
for storage in ("f16", "f8"):
    idx = FlatIndexF16(d, scan="q8" if storage == "f16" else "auto", storage=storage, capacity=n)
    for s0 in range(0, n, 500_000):
        idx.add(mk(500_000))
    idx.warm_up()
    check(f"clustered {n}x{d} {storage} k=100", idx, mk(256), 100)
    check(f"clustered {n}x{d} {storage} k=900", idx, mk(64), 900)
    del idx; torch.cuda.empty_cache()
