"""The 256 x 256 tile kernel WITHOUT a seam between output tiles (csrc/encoder.hip: rarc_gemm256s_f16_kernel, reached through
rarc_enc_gemm_zero_bias on large shapes) against the kernel with the seam (rarc_enc_gemm with a zero bias; RARC_GEMM_SEAM=0
in a child process would give the same): same operands, same k order per output element, fp32 accumulation -> the SAME
bits.  Shapes cover: one tile per workgroup (no successor), several tiles per workgroup (the seam proper), a ragged last
round (some workgroups have one tile more), the cut-off tail, K = 256 (the shortest stream: four k tiles) and the SwiGLU
epilogue; and a torch fp32 product as the outside yardstick."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu



@pytest.mark.parametrize("m,n,k,act", [
    (256 * 16, 4096, 256, 0),        # 256 tiles: one per workgroup, four k tiles
    (256 * 40, 4096, 1024, 0),       # 640 tiles: 2.5 rounds -> streams of 2 and 3 tiles (q|k|v shape of the reranker LM)
    (256 * 100, 1024, 2048, 0),      # 400 tiles: output projection shape
    (256 * 70, 1024, 3072, 0),       # 280 tiles = one round + a cut-off tail (runs as its own GEMM), down projection shape
    (256 * 24, 6144, 1024, 3),       # gate|up with the SwiGLU epilogue, 576 tiles
    (256 * 33, 2048, 320, 3),        # five k tiles, 264 tiles (ragged: 8 workgroups carry two)
])
def test_seamless_kernel_equals_the_kernel_with_a_seam(monkeypatch, m, n, k, act):
    import torch

    monkeypatch.setenv("RARC_GEMM_SEAM", "1")     # every eligible shape of rarc_enc_gemm_zero_bias takes the seamless kernel

    from rag_arc_amd.hip import binding as B

    lib = B.load_library()
    g = torch.Generator(device="cuda"); g.manual_seed(m + n + k)
    a = (torch.randn((m, k), device="cuda", generator=g) * 0.5).half()
    w = (torch.randn((n, k), device="cuda", generator=g) * 0.05).half()
    zero = torch.zeros(n, dtype=torch.float16, device="cuda")
    nc = n // 2 if act == 3 else n
    c_seam = torch.full((m, nc), 7.0, dtype=torch.float16, device="cuda")
    c_less = torch.full((m, nc), -7.0, dtype=torch.float16, device="cuda")
    st = torch.cuda.current_stream().cuda_stream
    B.check(lib.rarc_enc_gemm(a.data_ptr(), w.data_ptr(), zero.data_ptr(), c_seam.data_ptr(), m, n, k, act, st))
    B.check(lib.rarc_enc_gemm_zero_bias(a.data_ptr(), w.data_ptr(), zero.data_ptr(), c_less.data_ptr(), m, n, k, act, st))
    torch.cuda.synchronize()
    assert torch.isfinite(c_less.float()).all()
    bad = (c_seam.view(torch.int16) != c_less.view(torch.int16))
    assert not bad.any(), f"{int(bad.sum())} elements differ, first rows {torch.nonzero(bad.any(dim=1)).flatten()[:8].tolist()}"
    if act == 0:    # and it is the product: fp32 yardstick on a sample of rows
        rows = torch.arange(0, m, max(1, m // 64), device="cuda")
        ref = a[rows].float() @ w.float().T
        assert (c_less[rows].float() - ref).abs().max() <= 2e-2 * max(1.0, float(ref.abs().max()))
