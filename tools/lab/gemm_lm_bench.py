"""Timing of rarc_enc_gemm at the reranker LM's shapes (51 200 tokens) and a torch (hipBLASLt) yardstick (development tool).
RARC_GEMM_T256_MIN picks where the 256 x 256 tile kernel takes over from the 256 x 128 one."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from rag_arc_amd.hip import binding as B
lib = B.load_library()
M = int(os.environ.get("PROBE_M", 51200))
for (N, K) in [(4096, 1024), (1024, 2048), (6144, 1024), (1024, 3072)]:
    a = torch.randn((M, K), device="cuda").half(); w = (torch.randn((N, K), device="cuda") * 0.05).half(); b = torch.zeros(N, device="cuda").half()
    c = torch.empty((M, N), device="cuda", dtype=torch.float16)
    def run(fn, R=10):
        for _ in range(2): fn()
        torch.cuda.synchronize(); t = time.perf_counter()
        for _ in range(R): fn()
        torch.cuda.synchronize(); return (time.perf_counter() - t) / R
    dt = run(lambda: lib.rarc_enc_gemm(a.data_ptr(), w.data_ptr(), b.data_ptr(), c.data_ptr(), M, N, K, 0, 0))
    dt_t = run(lambda: torch.matmul(a, w.T))
    print(f"M={M} N={N} K={K}: rarc {dt*1e6:8.1f} us {2*M*N*K/dt/1e12:6.1f} TF/s | torch {dt_t*1e6:8.1f} us {2*M*N*K/dt_t/1e12:6.1f} TF/s")
