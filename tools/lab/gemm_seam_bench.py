"""rarc_enc_gemm (tile kernel with a seam between output tiles) vs rarc_enc_gemm_zero_bias (seamless stream of k tiles) vs
torch.matmul (hipBLASLt) at the reranker LM's shapes and along K (development tool; profiles/r03_gemm_seamless.txt)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from rag_arc_amd.hip import binding as B
lib = B.load_library()
M = int(os.environ.get("PROBE_M", 51200))
def run(fn, R=10):
    for _ in range(3): fn()
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(R): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t) / R
shapes = [(4096, 1024, 0), (1024, 2048, 0), (6144, 1024, 3), (1024, 3072, 0)] + [(4096, k, 0) for k in (256, 512, 1024, 2048, 4096)]
for (N, K, act) in shapes:
    a = torch.randn((M, K), device="cuda").half(); w = (torch.randn((N, K), device="cuda") * 0.05).half(); b = torch.zeros(N, device="cuda").half()
    if os.environ.get("PROBE_ZERO") == "1":   # same instruction stream, no operand toggling: what the clock does without the MFMA power
        a.zero_(); w.zero_()
    c = torch.empty((M, N // 2 if act == 3 else N), device="cuda", dtype=torch.float16)
    st = torch.cuda.current_stream().cuda_stream
    t_seam = run(lambda: lib.rarc_enc_gemm(a.data_ptr(), w.data_ptr(), b.data_ptr(), c.data_ptr(), M, N, K, act, st))
    t_less = run(lambda: lib.rarc_enc_gemm_zero_bias(a.data_ptr(), w.data_ptr(), b.data_ptr(), c.data_ptr(), M, N, K, act, st))
    t_t = run(lambda: torch.matmul(a, w.T))
    fl = 2.0 * M * N * K
    print(f"M={M} N={N} K={K} act={act}: seam {t_seam*1e6:8.1f} us {fl/t_seam/1e12:7.1f} TF/s | seamless {t_less*1e6:8.1f} us {fl/t_less/1e12:7.1f} TF/s"
          f" ({(t_seam/t_less-1)*100:+.1f} %) | torch {t_t*1e6:8.1f} us {fl/t_t/1e12:7.1f} TF/s", flush=True)
