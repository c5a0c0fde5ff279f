"""What the vendor library reaches on the fp32-class encoder's small-batch GEMM shapes (K' = 3K split operands, fp16 in):
a yardstick for rarc_gemm128pp + split-K at 32 x 32 and 1 x 32(->128) tokens (development tool)."""
import sys, torch
dev = torch.device("cuda", 0)
for M in (128, 1024):
    for name, N, K in (("qkv", 3072, 3072), ("out", 1024, 3072), ("ffn1", 4096, 3072), ("ffn2", 1024, 12288)):
        a = torch.randn(M, K, device=dev, dtype=torch.float16); w = torch.randn(N, K, device=dev, dtype=torch.float16) * 0.05
        for out_dtype in (torch.float16,):
            f = lambda: torch.nn.functional.linear(a, w)
            for _ in range(10): f()
            torch.cuda.synchronize(); e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(50): f()
            e1.record(); torch.cuda.synchronize()
            us = e0.elapsed_time(e1) / 50 * 1e3
            print(f"M={M:5d} {name:5s} N={N:5d} K'={K:6d}: hipBLASLt {us:7.1f} us ({2.0*M*N*K/us/1e6:6.0f} TF/s)")
