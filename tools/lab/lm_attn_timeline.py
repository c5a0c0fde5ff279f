"""s_memtime timeline of one attention workgroup (wave 0) inside a full LM forward: build librarc with -DLM_ATTN_TIMELINE
(tools/build_variant.sh) and run with RARC_LIBRARY pointing at it."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
import bench
from rag_arc_amd.hip import binding as B
dev = torch.device("cuda", 0)
lm, _ = bench.build_reranker_lm(torch, dev, 0, want_host=False)
n, L = int(os.environ.get("PROBE_PAIRS", 640)), int(os.environ.get("PROBE_LEN", 256))
g = torch.Generator(device=dev); g.manual_seed(1)
ids = torch.randint(10, bench.LM_GEOM["V"], (n, L), generator=g, device=dev).int()
start = torch.zeros(n, dtype=torch.int32, device=dev)
for _ in range(2): lm.yes_no_logits_device(ids, start, 1, 2)
torch.cuda.synchronize()
lib = B.load_library()
buf = (ctypes.c_ulonglong * 512)()
lib.rarc_lm_debug_timeline(buf, 512)
t = np.array(buf[:64], dtype=np.int64)
n = int(np.argmax(t == 0)) if (t == 0).any() else len(t)
t = t[:n] - t[0]
print("stamps (shader cycles)")
print(t.tolist())
print("deltas:", np.diff(t).tolist())
