"""Timing of rarc_enc_gemm at encoder shapes (development tool)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from rag_arc_amd.hip import binding as B
lib = B.load_library()
for (M, N, K, act) in [(8192, 3072, 1024, 0), (8192, 1024, 1024, 0), (8192, 4096, 1024, 1), (8192, 1024, 4096, 0), (8192, 1152, 384, 0), (8192, 1536, 384, 1)]:
    a = torch.randn((M, K), device="cuda").half(); w = (torch.randn((N, K), device="cuda") * 0.05).half(); b = torch.zeros(N, device="cuda").half()
    c = torch.empty((M, N), device="cuda", dtype=torch.float16)
    for _ in range(3): B.check(lib.rarc_enc_gemm(a.data_ptr(), w.data_ptr(), b.data_ptr(), c.data_ptr(), M, N, K, act, 0))
    torch.cuda.synchronize(); t = time.perf_counter(); R = 20
    for _ in range(R): lib.rarc_enc_gemm(a.data_ptr(), w.data_ptr(), b.data_ptr(), c.data_ptr(), M, N, K, act, 0)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t) / R
    ref = (a.float() @ w.float().T)
    if act: ref = torch.nn.functional.gelu(ref)
    err = (c.float() - ref).abs().max().item()
    print(f"M={M} N={N} K={K} act={act}: {dt*1e6:8.1f} us  {2*M*N*K/dt/1e12:6.1f} TF/s  maxerr {err:.3e}")
