"""s_memtime per k tile of one workgroup's stream in the seamless 256 x 256 GEMM kernel (library built with -DG256S_TIMELINE:
tools/build_variant_enc.sh).  Prints shader cycles per k tile, output tile by output tile."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from rag_arc_amd.hip import binding as B
os.environ["RARC_GEMM_SEAM"] = "1"
lib = B.load_library()
M, N, K = int(os.environ.get("PROBE_M", 51200)), int(os.environ.get("PROBE_N", 4096)), int(os.environ.get("PROBE_K", 1024))
a = torch.randn((M, K), device="cuda").half(); w = (torch.randn((N, K), device="cuda") * 0.05).half(); z = torch.zeros(N, device="cuda").half()
c = torch.empty((M, N), device="cuda", dtype=torch.float16)
st = torch.cuda.current_stream().cuda_stream
for _ in range(3): lib.rarc_enc_gemm_zero_bias(a.data_ptr(), w.data_ptr(), z.data_ptr(), c.data_ptr(), M, N, K, 0, st)
torch.cuda.synchronize()
buf = (ctypes.c_ulonglong * 1024)()
lib.rarc_gemm_debug_timeline.restype = ctypes.c_int
lib.rarc_gemm_debug_timeline(buf, 1024)
KT = K // 64
t = np.array(buf[:], dtype=np.int64)
n = int(np.argmax(t == 0)) if (t == 0).any() else len(t)
d = np.diff(t[:n])
print("k tiles stamped:", n, "KT =", KT)
for i in range(0, len(d), KT):
    print("tile", i // KT, d[i:i + KT].tolist(), "sum", int(d[i:i + KT].sum()))
