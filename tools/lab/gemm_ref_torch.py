"""What the vendor library reaches on the encoder's GEMM shapes (a yardstick for rarc_enc_gemm; development tool)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from rag_arc_amd.hip import binding as B
lib = B.load_library()
dev = torch.device("cuda", 0)
for (M, N, K) in [(8192, 3072, 1024), (8192, 1024, 1024), (8192, 4096, 1024), (8192, 1024, 4096), (16384, 4096, 1024),
                  (1024, 3072, 1024), (1024, 4096, 1024), (1024, 1024, 4096)]:
    a = torch.randn(M, K, device=dev, dtype=torch.float16); w = torch.randn(N, K, device=dev, dtype=torch.float16) * 0.05
    b = torch.zeros(N, device=dev, dtype=torch.float16); c = torch.empty(M, N, device=dev, dtype=torch.float16)
    def t_ours():
        B.check(lib.rarc_enc_gemm(a.data_ptr(), w.data_ptr(), b.data_ptr(), c.data_ptr(), M, N, K, 0, torch.cuda.current_stream().cuda_stream))
    def t_torch():
        torch.nn.functional.linear(a, w, b)
    res = []
    for f in (t_ours, t_torch):
        for _ in range(5): f()
        torch.cuda.synchronize(); e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20): f()
        e1.record(); torch.cuda.synchronize()
        res.append(e0.elapsed_time(e1) / 20 * 1e3)
    fl = 2.0 * M * N * K
    print(f"M={M:6d} N={N:5d} K={K:5d}: ours {res[0]:7.1f} us ({fl/res[0]/1e6:6.0f} TF/s)   torch/hipBLASLt {res[1]:7.1f} us ({fl/res[1]/1e6:6.0f} TF/s)")
