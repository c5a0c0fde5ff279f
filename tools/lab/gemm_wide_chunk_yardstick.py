"""The wide path's chunk GEMM shapes (M rows x 256 queries x K = 1536) : rarc_enc_gemm_zero_bias vs hipBLASLt (development tool)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from rag_arc_amd.hip import binding as B
lib = B.load_library(); dev = torch.device("cuda", 0)
K, N = int(os.environ.get("PROBE_K", 1536)), 256
w = (torch.randn(N, K, device=dev) * 0.05).half(); bias = torch.zeros(N, device=dev).half()
for M in (16384, 32768, 65536, 131072, 262144, 1048576):
    a = (torch.randn(M, K, device=dev) * 0.05).half(); c = torch.empty(M, N, device=dev, dtype=torch.float16)
    def ours(): B.check(lib.rarc_enc_gemm_zero_bias(a.data_ptr(), w.data_ptr(), bias.data_ptr(), c.data_ptr(), M, N, K, 0, torch.cuda.current_stream().cuda_stream))
    def blas(): torch.nn.functional.linear(a, w)
    res = []
    for f in (ours, blas):
        for _ in range(5): f()
        torch.cuda.synchronize(); e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(30): f()
        e1.record(); torch.cuda.synchronize(); res.append(e0.elapsed_time(e1) / 30 * 1e3)
    fl = 2.0 * M * N * K
    print(f"M={M:8d}: rarc_enc_gemm_zero_bias {res[0]:8.1f} us ({fl / res[0] / 1e6:6.0f} TF/s)   hipBLASLt {res[1]:8.1f} us ({fl / res[1] / 1e6:6.0f} TF/s)")
