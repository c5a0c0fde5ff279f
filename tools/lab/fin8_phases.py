"""Phase stamps of rarc_finalize_q8_kernel's workgroup 0 (100 MHz ticks -> us) on a 12.5M x 768 shard: sets the library's g_fin8_dbg to a
device buffer, runs searches, prints stamp differences.  Development probe."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from rag_arc_amd.hip import binding as B
from rag_arc_amd.hip.engine import FlatIndexF16
lib = B.load_library()
n, d = int(os.environ.get("PROBE_ROWS", 12_500_000)), 768
idx = FlatIndexF16(d, capacity=n)
buf = torch.zeros((1 << 20, d), dtype=torch.float16, device="cuda")
for s0 in range(0, n, 1 << 20):
    m = min(1 << 20, n - s0)
    B.check(lib.rarc_synth_rows_f16(buf.data_ptr(), d, d, s0, m, 1234, 0))
    idx.add_rows_f16(buf[:m].clone(), 1.001, n_valid=m) if s0 == 0 else idx.add_rows_f16(buf[:m].clone(), 1.001, n_valid=m)
q = torch.zeros((256, d), dtype=torch.float32, device="cuda")
B.check(lib.rarc_synth_rows_f32(q.data_ptr(), d, d, 0, 256, 4321, 0))
dbg = torch.zeros(2048, dtype=torch.int64, device="cuda")
ptr = ctypes.c_void_p.in_dll(lib, "g_fin8_dbg")
for _ in range(3):
    idx.search_device(q, 100)
ptr.value = dbg.data_ptr()
for rep in range(3):
    dbg.zero_()
    idx.search_device(q, 100)
    torch.cuda.synchronize()
    h = dbg.cpu().tolist()
    st = h[:8]
    print("FIN8 stamps (us from start):", [round((x - st[0]) / 100.0, 1) for x in st], " ne1 =", h[8], " ne =", h[9])
    ends = [h[18 + 4 * qq] for qq in range(256)]
    print("   all 256 workgroups: last end - first start = %.1f us; workgroup 0 ran %.1f us" % ((max(ends) - st[0]) / 100.0, (st[7] - st[0]) / 100.0))
ptr.value = None
