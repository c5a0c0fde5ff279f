import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from rag_arc_amd.hip import binding as B
lib = B.load_library()
torch.manual_seed(0)
for (M, N, K, act) in [(4096, 6144, 64, 0), (4096, 6144, 192, 1), (8192, 3072, 1024, 0), (8192, 4096, 1024, 1), (2048, 12288, 448, 0)]:
    a = torch.randn((M, K), device="cuda").half(); w = (torch.randn((N, K), device="cuda") * 0.05).half(); b = torch.randn(N, device="cuda").half()
    for rep in range(3):
        c = torch.zeros((M, N), device="cuda", dtype=torch.float16)
        B.check(lib.rarc_enc_gemm(a.data_ptr(), w.data_ptr(), b.data_ptr(), c.data_ptr(), M, N, K, act, 0))
        torch.cuda.synchronize()
        ref = a.float() @ w.float().T + b.float()
        if act: ref = torch.nn.functional.gelu(ref)
        err = (c.float() - ref).abs().max().item()
        print(f"M={M} N={N} K={K} act={act} rep={rep}: maxerr {err:.3e}  ref max {ref.abs().max().item():.2f}")
