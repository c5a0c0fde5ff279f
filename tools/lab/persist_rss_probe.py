"""Where does the host memory of a shard save / load go?  VmRSS / VmHWM at each step (one GPU)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
def vm(f):
    for line in open("/proc/self/status"):
        if line.startswith(f + ":"): return int(line.split()[1]) // 1024
def smaps():
    # biggest resident mappings
    cur, out = None, []
    for line in open("/proc/self/smaps"):
        p = line.split()
        if "-" in p[0] and len(p) >= 5 and p[0][0] in "0123456789abcdef" and ":" not in p[0]:
            cur = " ".join(p[5:]) or "[anon]"
        elif p[0] == "Rss:":
            out.append((int(p[1]) // 1024, cur))
    agg = {}
    for r, n in out: agg[n] = agg.get(n, 0) + r
    return sorted(((r, n) for n, r in agg.items()), reverse=True)[:8]
import torch, bench
from rag_arc_amd.hip import binding as B
from rag_arc_amd.hip.engine import FlatIndexF16
lib = B.load_library(); torch.zeros(1, device="cuda")
print("after init", vm("VmRSS"), vm("VmHWM"))
n = int(os.environ.get("ROWS", 2_000_000))
idx = bench.build_index(torch, lib, B, FlatIndexF16, 0, 768, 0, n)
print("after build", vm("VmRSS"), vm("VmHWM"))
st = idx._io_staging(); print("after pinned ring", st.numel() >> 20, "MiB:", vm("VmRSS"), vm("VmHWM"))
d = os.environ.get("DIR", "/tmp/rssprobe"); os.makedirs(d, exist_ok=True)
for threads, direct in ((1, True), (8, True), (8, False)):
    s = idx.save_shard(d + "/p.rarc", threads=threads, direct=direct)
    print(f"save threads={threads} direct={direct}: {s['gb_per_s']:.1f} GB/s rss", vm("VmRSS"), "hwm", vm("VmHWM"))
    i2 = FlatIndexF16(768, metric="cosine", device=0)
    s = i2.load_shard(d + "/p.rarc", threads=threads, direct=direct)
    print(f"load threads={threads} direct={direct}: {s['gb_per_s']:.1f} GB/s rss", vm("VmRSS"), "hwm", vm("VmHWM"))
    del i2
print(smaps())
os.unlink(d + "/p.rarc")
